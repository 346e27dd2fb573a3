"""Multi-rank (N > 1) path on CPU: index sharding + global-id RNG keying + the optional
observation all-gather, with the gloo backend at world_size 2.  The per-rank stepper here is
the CPU oracle (tests may use it); on the GPU box each rank drives its own mgym_env with
exactly the same shard plan (bench.py)."""
import os
import socket

import numpy as np
import pytest

from modurl_gym_amd.shard import mixed_population, population_plan, shard_range


def test_shard_ranges_tile_the_population():
    for n in (0, 1, 7, 1024, 1048576, 8388608 + 3):
        for world in (1, 2, 3, 4, 8):
            blocks = [shard_range(n, world, r) for r in range(world)]
            assert blocks[0].start == 0 and sum(b.count for b in blocks) == n
            for a, b in zip(blocks, blocks[1:]):
                assert a.start + a.count == b.start
            assert max(b.count for b in blocks) - min(b.count for b in blocks) <= 1
    with pytest.raises(ValueError):
        shard_range(10, 2, 2)
    assert mixed_population(1048576) == {"cartpole": 524288, "mountain_car": 262144, "lunar_lander": 262144}


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_total, q):
    import torch
    import torch.distributed as dist

    from modurl_gym_amd.shard import all_gather_observations
    from oracle import oracle as ora

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        sh = shard_range(n_total, world, rank)
        env = ora.OracleVec(ora.CARTPOLE, sh.count, seed=99, env_id_base=sh.start)
        env.reset()
        rng = np.random.default_rng(5)
        acts = rng.integers(0, 2, (20, n_total)).astype(np.uint32)
        obs = None
        for t in range(20):
            obs, rew, done, trunc = env.step(acts[t, sh.start:sh.start + sh.count])
            env.reset(mask=done | trunc)
        # bench.py's timing reduction: barrier, then MAX over ranks of the local elapsed time
        dist.barrier()
        t_local = torch.tensor([1.0 + rank], dtype=torch.float64)
        dist.all_reduce(t_local, op=dist.ReduceOp.MAX)
        full = all_gather_observations(torch.from_numpy(env.get_state()[:4].copy()), world)
        if rank == 0:
            q.put((full.numpy(), float(t_local[0])))
    finally:
        dist.destroy_process_group()


def test_two_rank_gloo_matches_single_rank():
    import torch.multiprocessing as mp

    from oracle import oracle as ora

    n_total, world = 4096, 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_total, q)) for r in range(world)]
    for p in procs:
        p.start()
    gathered, tmax = q.get(timeout=180)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert tmax == 2.0
    # single-rank run of the same population: identical, because streams are keyed by global env id
    env = ora.OracleVec(ora.CARTPOLE, n_total, seed=99, env_id_base=0)
    env.reset()
    acts = np.random.default_rng(5).integers(0, 2, (20, n_total)).astype(np.uint32)
    for t in range(20):
        _, _, done, trunc = env.step(acts[t])
        env.reset(mask=done | trunc)
    assert np.array_equal(gathered, env.get_state()[:4])


def test_population_plans_weak_and_strong():
    # weak (bench.py default): every rank the BASELINE per-GPU load, ids laid out [family][rank][local]
    for world in (1, 2, 4, 8):
        seen = []
        for rank in range(world):
            plan = population_plan("mixed", world, rank, "weak", n_per_gpu=1 << 20)
            assert [(f, c) for f, c, _ in plan] == [("cartpole", 524288), ("mountain_car", 262144), ("lunar_lander", 262144)]
            seen += [(f, b, c) for f, c, b in plan]
        for fam in ("cartpole", "mountain_car", "lunar_lander"):   # each family's blocks tile a contiguous id range
            blocks = sorted((b, c) for f, b, c in seen if f == fam)
            for (b0, c0), (b1, _) in zip(blocks, blocks[1:]):
                assert b0 + c0 == b1
        assert sorted(b for _, b, _ in seen)[0] == 0
    # strong: the node total is fixed (8 388 608 mixed: BASELINE configs[4]) and cut into `world` blocks per family
    for world in (1, 2, 3, 8):
        tot = {}
        for rank in range(world):
            for fam, cnt, base in population_plan("mixed", world, rank, "strong", n_total=8 << 20):
                tot.setdefault(fam, []).append((base, cnt))
        assert {f: sum(c for _, c in v) for f, v in tot.items()} == {"cartpole": 4 << 20, "mountain_car": 2 << 20, "lunar_lander": 2 << 20}
        flat = sorted(x for v in tot.values() for x in v)
        assert flat[0][0] == 0 and all(b0 + c0 == b1 for (b0, c0), (b1, _) in zip(flat, flat[1:]))
    assert population_plan("cartpole", 4, 3, "strong", n_total=10) == [("cartpole", 3, 7)]


def _strong_worker(rank, world, port, n_total, q):
    import torch
    import torch.distributed as dist

    from modurl_gym_amd.shard import all_gather_observations, all_reduce_episode_count
    from oracle import oracle as ora

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        (fam, cnt, base), = population_plan("cartpole", world, rank, "strong", n_total=n_total)
        env = ora.OracleVec(ora.CARTPOLE, cnt, seed=11, env_id_base=base)
        env.reset()
        acts = np.random.default_rng(6).integers(0, 2, (30, n_total)).astype(np.uint32)
        finished = 0
        for t in range(30):
            _, _, done, trunc = env.step(acts[t, base:base + cnt])
            m = done | trunc
            finished += int(m.sum())
            env.reset(mask=m)
        total = all_reduce_episode_count(finished)                   # the optional 8-byte all-reduce
        full = all_gather_observations(torch.from_numpy(env.get_state()[:4].copy()), world)
        if rank == 0:
            q.put((full.numpy(), total))
    finally:
        dist.destroy_process_group()


def test_two_rank_strong_scaling_plan_matches_single_rank():
    import torch.multiprocessing as mp

    from oracle import oracle as ora

    n_total, world = 4096, 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_strong_worker, args=(r, world, port, n_total, q)) for r in range(world)]
    for p in procs:
        p.start()
    gathered, episodes = q.get(timeout=180)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    env = ora.OracleVec(ora.CARTPOLE, n_total, seed=11, env_id_base=0)
    env.reset()
    acts = np.random.default_rng(6).integers(0, 2, (30, n_total)).astype(np.uint32)
    finished = 0
    for t in range(30):
        _, _, done, trunc = env.step(acts[t])
        m = done | trunc
        finished += int(m.sum())
        env.reset(mask=m)
    assert np.array_equal(gathered, env.get_state()[:4]) and episodes == finished and finished > 0
