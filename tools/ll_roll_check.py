#!/usr/bin/env python3
"""mgym_rollout for LunarLander (ll_roll.h: one persistent launch) against the CPU oracle and against K x mgym_step (*GPU box*).

  tools/ll_roll_check.py check <n> <steps> <K> [auto_reset=1] [wind=1]   every word of every step vs the oracle
  tools/ll_roll_check.py time  <n> <K> [reps=20] [warm_steps=640]          ms per step-equivalent of mgym_rollout vs mgym_step
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import modurl_gym_amd as mg  # noqa: E402


def check(n, steps, K, auto=1, wind=1):
    from oracle import oracle as ora
    from test_gpu_lunar_soak import skilled_actions

    env = mg.VecEnv(mg.LUNARLANDER, n, seed=99, enable_wind=bool(wind), auto_reset=bool(auto))
    ref = ora.OracleVec(ora.LUNARLANDER, n, seed=99, enable_wind=bool(wind))
    assert np.array_equal(env.reset(), ref.reset(nthreads=16))
    rng = np.random.default_rng(1)
    t = 0
    episodes = 0
    while t < steps:
        acts, exps = [], []
        for _ in range(K):
            a = skilled_actions(rng, ref.get_state(), n)
            obs, rew, done, trunc = ref.step(a, nthreads=16)
            if auto:
                ro = ref.reset(done, nthreads=16)
                obs = np.where(done.astype(bool)[None, :], ro, obs)
            exps.append((obs, rew, done, trunc))
            acts.append(a)
            episodes += int(done.sum())
        gobs, grew, gdone, gtrunc = env.rollout(np.stack(acts))
        env.sync()
        for j in range(K):
            for g, e, nm in zip((gobs[j], grew[j], gdone[j], gtrunc[j]), exps[j], ("obs", "reward", "done", "truncated")):
                gv = g.view(np.uint32) if g.dtype == np.float32 else g
                ev = e.view(np.uint32) if e.dtype == np.float32 else e
                if not np.array_equal(gv, ev):
                    bad = np.argwhere(gv != ev)
                    print(f"MISMATCH step {t + j} (rollout step {j}): {nm} differs at {bad[0]} ({len(bad)} words): {g[tuple(bad[0])]!r} vs {e[tuple(bad[0])]!r}")
                    return 1
        # the engine-owned observation and the exported state after the call
        if not np.array_equal(env.observation().view(np.uint32), exps[-1][0].view(np.uint32)):
            print(f"MISMATCH after step {t + K}: mgym_observation differs from the last step's observation")
            return 1
        t += K
    gs, os_ = env.get_state(), ref.get_state()
    same = np.array_equal(gs.view(np.uint32), os_.view(np.uint32))
    if not same:
        # NaN payload of prev_shaping=None is the only word allowed to differ in representation
        bad = np.argwhere(gs.view(np.uint32) != os_.view(np.uint32))
        print(f"state blob differs in {len(bad)} words, first {bad[:5].tolist()}: {gs[tuple(bad[0])]!r} vs {os_[tuple(bad[0])]!r}")
    print(f"OK n={n} steps={t} K={K} auto={auto} wind={wind}: every word equals the oracle; {episodes} episodes finished; state blob equal: {same}")
    env.close()
    return 0


def timeit(n, K, reps=20, warm=640):
    import torch

    dev = "cuda:0"
    stream = torch.cuda.Stream(device=0)
    env = mg.VecEnv(mg.LUNARLANDER, n, seed=0x5EED0008, enable_wind=True, auto_reset=True, stream=stream.cuda_stream)
    g = torch.Generator(device=dev)
    g.manual_seed(5)
    acts = torch.randint(0, 4, (max(K, 16), n), generator=g, device=dev, dtype=torch.int32)
    rew = torch.empty((K, n), device=dev, dtype=torch.float32)
    dn = torch.zeros((K, n), device=dev, dtype=torch.uint8)
    tr = torch.zeros((K, n), device=dev, dtype=torch.uint8)
    stream.wait_stream(torch.cuda.current_stream(0))
    env.reset_device(None, None)
    for t in range(warm):
        env.step_device(acts[t % 16], None, rew[0], dn[0], tr[0])
    env.sync()
    out = {}
    for name in ("step", "rollout", "step", "rollout"):
        env.sync()
        env.timer_start()
        if name == "step":
            for r in range(reps):
                for k in range(K):
                    env.step_device(acts[k], None, rew[k], dn[k], tr[k])
        else:
            for r in range(reps):
                env.rollout_device(acts, K, None, rew, dn, tr)
        ms = env.timer_stop()
        env.sync()
        out.setdefault(name, []).append(ms / (reps * K))
    print(f"n={n} K={K}: ms per step-equivalent: mgym_step {['%.4f' % v for v in out['step']]}  mgym_rollout {['%.4f' % v for v in out['rollout']]}"
          f"  -> {n / (min(out['rollout']) * 1e-3):.4g} env-steps/s")
    env.close()


if __name__ == "__main__":
    mode = sys.argv[1]
    if mode == "check":
        sys.exit(check(*[int(x) for x in sys.argv[2:]]))
    timeit(*[int(x) for x in sys.argv[2:]])
