"""Index sharding of an environment population over the GPUs of one node.

Environments never read each other's state (reference: cartpole.rs:251-348,
mountain_car.rs:293-330, lunar_lander.rs:919-1167 touch only `self`), so multi-GPU is pure
index partitioning: rank g of G owns the contiguous block [g*N/G, (g+1)*N/G) and creates its
handle with env_id_base = block start.  Per-env random streams are keyed by the GLOBAL env id,
so results are bit-identical for any G.  There is no data-path collective; the optional
all-gather of observations (a learner that wants every observation on every rank) is the only
RCCL call and is kept off the step path.
"""
from collections import namedtuple

Shard = namedtuple("Shard", ["rank", "world", "start", "count"])


def shard_range(n_total, world, rank):
    """Contiguous block of rank `rank`: sizes differ by at most one, blocks tile [0, n_total)."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError(f"bad shard ({rank} of {world})")
    start = (n_total * rank) // world
    end = (n_total * (rank + 1)) // world
    return Shard(rank, world, start, end - start)


def mixed_population(n_per_gpu):
    """BASELINE config 5 / SURVEY §8d C5: per GPU 1/2 CartPole, 1/4 MountainCar, 1/4 LunarLander."""
    return {"cartpole": n_per_gpu // 2, "mountain_car": n_per_gpu // 4, "lunar_lander": n_per_gpu // 4}


def population_plan(workload, world, rank, scaling="weak", n_per_gpu=1 << 20, n_total=8 << 20):
    """The handles rank `rank` of `world` creates: [(family, n_envs, env_id_base), ...].

    Global env ids are laid out [family][rank][local index].  scaling = "weak": every rank steps n_per_gpu envs (split by
    mixed_population for the mixed workload), so the node total grows with `world` (BASELINE configs[1] / configs[4]
    per-GPU load).  scaling = "strong": the node total n_total (default 8 388 608, BASELINE configs[4]) is fixed and
    each family's population is cut into `world` contiguous blocks by shard_range.  Either way per-env random streams
    are keyed by the global id, so the results of an env do not depend on `world`."""
    if scaling not in ("weak", "strong"):
        raise ValueError("scaling must be 'weak' or 'strong'")
    if workload == "mixed":
        pop = mixed_population(n_per_gpu if scaling == "weak" else n_total)
    else:
        pop = {workload: n_per_gpu if scaling == "weak" else n_total}
    plan, base = [], 0
    for family, cnt in pop.items():
        if scaling == "weak":
            plan.append((family, cnt, base + rank * cnt))
            base += world * cnt
        else:
            sh = shard_range(cnt, world, rank)
            plan.append((family, sh.count, base + sh.start))
            base += cnt
    return plan


def all_reduce_episode_count(local_count, device=None):
    """Optional 8-byte all-reduce (SUM) of the per-rank finished-episode counters (VecEnv.episode_count()) for logging;
    torch.distributed default group ("nccl" = RCCL on ROCm, "gloo" on CPU).  Not on the step path."""
    import torch
    import torch.distributed as dist

    t = torch.tensor([int(local_count)], dtype=torch.int64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return int(t[0])


def all_gather_observations(obs_local, world):
    """Optional RCCL/gloo all-gather of SoA observations [obs_dim, n_local] -> [obs_dim, n_total]
    (equal shard sizes).  Uses torch.distributed's default group: backend "nccl" is RCCL over
    xGMI on ROCm, "gloo" on CPU.  Not on the step path."""
    import torch
    import torch.distributed as dist

    parts = [torch.empty_like(obs_local) for _ in range(world)]
    dist.all_gather(parts, obs_local.contiguous())
    return torch.cat(parts, dim=1)
