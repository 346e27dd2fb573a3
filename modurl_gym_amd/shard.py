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


def all_gather_observations(obs_local, world):
    """Optional RCCL/gloo all-gather of SoA observations [obs_dim, n_local] -> [obs_dim, n_total]
    (equal shard sizes).  Uses torch.distributed's default group: backend "nccl" is RCCL over
    xGMI on ROCm, "gloo" on CPU.  Not on the step path."""
    import torch
    import torch.distributed as dist

    parts = [torch.empty_like(obs_local) for _ in range(world)]
    dist.all_gather(parts, obs_local.contiguous())
    return torch.cat(parts, dim=1)
