"""BASELINE configs[4] (mixed CartPole + MountainCar + LunarLander, 8 388 608 envs on 8 GPUs) as ONE rank sees it:
the three handles built exactly as bench.py builds them — per-GPU population mixed_population(1 << 20), global env ids
laid out [family][rank][local index] — for world = 2, rank = 1, fused auto-reset, stepped side by side for 40 steps.
CartPole and MountainCar are compared with the oracle over the whole shard, bit for bit; LunarLander (oracle: ~1e5
env-steps/s per core) on eight 512-env blocks spread over the shard, driven for 110 steps so that contacts, time-of-impact
sub-steps, crashes and fused resets occur.  The oracle envs carry the same GLOBAL ids, so this also checks the id layout."""
import numpy as np
import pytest

import modurl_gym_amd as mg
from oracle import oracle as ora

pytestmark = pytest.mark.gpu

WORLD, RANK, SEED = 2, 1, 0x5EED0001 + 3


def handles():
    pop = mg.mixed_population(1 << 20)
    kinds = {"cartpole": (mg.CARTPOLE, ora.CARTPOLE, 2), "mountain_car": (mg.MOUNTAINCAR, ora.MOUNTAINCAR, 3),
             "lunar_lander": (mg.LUNARLANDER, ora.LUNARLANDER, 4)}
    out, base = {}, 0
    for name, cnt in pop.items():   # bench.py: Stepper(..., base + rank * cnt, ...); base += world * cnt
        kind, okind, nact = kinds[name]
        extra = dict(enable_wind=True) if name == "lunar_lander" else {}
        out[name] = (mg.VecEnv(kind, cnt, seed=SEED, env_id_base=base + RANK * cnt, auto_reset=True, **extra), okind, nact, cnt,
                     base + RANK * cnt, extra)
        base += WORLD * cnt
    return out


def test_config5_per_gpu_population_matches_oracle_with_global_ids():
    h = handles()
    assert [v[4] for v in h.values()] == [524288, 2 * 524288 + 262144, 2 * 524288 + 2 * 262144 + 262144]
    rng = np.random.default_rng(7)
    # ---- CartPole + MountainCar: whole shard, 40 steps, bit for bit
    for name in ("cartpole", "mountain_car"):
        env, okind, nact, n, gbase, _ = h[name]
        ref = ora.OracleVec(okind, n, seed=SEED, env_id_base=gbase)
        assert np.array_equal(env.reset(), ref.reset(nthreads=16)), name
        finished = 0
        for t in range(40):
            a = rng.integers(0, nact, n).astype(np.uint32)
            got, exp = env.step(a), ref.step(a, nthreads=16)
            for g, e, what in zip(got[1:], exp[1:], ("reward", "done", "truncated")):
                assert np.array_equal(g, e), f"{name} step {t}: {what}"
            m = exp[2] | exp[3]
            finished += int(m.sum())
            ref.reset(mask=m, nthreads=16)
            assert np.array_equal(got[0], ref.get_state()[: env.obs_dim]), f"{name} step {t}: observation after the fused reset"
        assert np.array_equal(env.get_state().view(np.uint32), ref.get_state().view(np.uint32)), name
        assert env.episode_count() == finished
        if name == "cartpole":
            assert finished > n
    # ---- LunarLander: the whole shard steps on the GPU, eight 512-env blocks of it are checked against oracles
    env, okind, nact, n, gbase, extra = h["lunar_lander"]
    blocks = [int(o) for o in np.linspace(0, n - 512, 8).astype(np.int64) // 64 * 64]
    refs = [ora.OracleVec(okind, 512, seed=SEED, env_id_base=gbase + o, **extra) for o in blocks]
    obs0 = env.reset()
    for o, r in zip(blocks, refs):
        assert np.array_equal(obs0[:, o:o + 512], r.reset(nthreads=8))
    episodes = 0
    for t in range(110):
        a = rng.integers(0, nact, n).astype(np.uint32)
        got = env.step(a)
        for o, r in zip(blocks, refs):
            eo, er, ed, et = r.step(a[o:o + 512], nthreads=8)
            ro = r.reset(ed, nthreads=8)
            eo = np.where(ed.astype(bool)[None, :], ro, eo)   # the engine's observation is the one after the fused reset
            assert np.array_equal(got[2][o:o + 512], ed) and np.array_equal(got[1][o:o + 512], er), f"lunar_lander step {t} block {o}"
            assert np.array_equal(got[0][:, o:o + 512], eo), f"lunar_lander step {t} block {o}: observation"
            episodes += int(ed.sum())
    assert episodes > 1000   # crashes / landings with contacts and resets happened in the checked blocks


def test_bench_steppers_in_graph_mode_on_three_streams_equal_the_oracles():
    """The figure `extra.mixed_configs4_this_n` comes from bench.py's own Stepper objects — one per family, each on a stream of its own,
    every step a hipGraph replay of 16 captured step launches with the fused auto-reset (LunarLander: staged resets inside the graph).
    This drives exactly those objects (smaller populations, same code) and compares the engines' final state blobs and finished-episode
    counts with oracles fed the same action ring.  Protocol: /root/reference src/testing.rs:65-134 (step, compare, reset when done)."""
    import torch

    import bench

    steps = 160    # 10 replays of the 16-step graph; long enough for LunarLander contacts, crashes and fused resets
    pop = {"cartpole": 65536, "mountain_car": 32768, "lunar_lander": 32768}
    kinds = {"cartpole": (ora.CARTPOLE, {}), "mountain_car": (ora.MOUNTAINCAR, {}), "lunar_lander": (ora.LUNARLANDER, dict(enable_wind=True))}
    streams = [torch.cuda.Stream(device=0) for _ in pop]
    base, steppers = 0, []
    for (name, cnt), st in zip(pop.items(), streams):
        steppers.append(bench.Stepper(mg, torch, name, cnt, 0, SEED, base + RANK * cnt, st, "fused", "auto"))
        base += WORLD * cnt
    assert all(s.launch == "graph" for s in steppers)
    for s in steppers:          # issue order of bench.py's mixed section: every family's replays queued, the streams overlap on the device
        s.run(steps)
    torch.cuda.synchronize()
    for s in steppers:
        s.env.sync()
    base = 0
    for s, (name, cnt) in zip(steppers, pop.items()):
        okind, extra = kinds[name]
        ref = ora.OracleVec(okind, cnt, seed=SEED, env_id_base=base + RANK * cnt, **extra)
        base += WORLD * cnt
        ref.reset(nthreads=16)
        ring = s.actions.cpu().numpy().astype(np.uint32)
        finished = 0
        for k in range(steps):
            _, _, d, tr = ref.step(ring[k % bench.RING], nthreads=16)
            m = (d | tr).astype(np.uint8)
            finished += int(m.sum())
            ref.reset(mask=m, nthreads=16)
        assert np.array_equal(s.env.get_state().view(np.uint32), ref.get_state().view(np.uint32)), f"{name}: state blob after {steps} graph-replayed steps"
        assert s.env.episode_count() == finished and (finished > 0 or name == "mountain_car"), name   # (a random policy never drives the car up the hill)
    for s in steppers:
        s.close()
