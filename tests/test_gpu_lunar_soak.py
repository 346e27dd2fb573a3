"""Long-horizon LunarLander parity in the driver-run suite: the HIP engine (through the C ABI) against the CPU oracle over
thousands of steps with auto-reset, wind on, a policy with enough skill that landings (sleep, +100) occur as well as crashes
and fly-aways — EVERY observation word, reward, done flag bit-identical.  Once per block layout of the contact kernel
(32 lanes with the per-lane world in LDS, 64 lanes with it in registers / scratch), once with every velocity constraint
forced into the global workspace (both contact launches of the overlapped order then use their workspace slices at the
same time: ADVICE r2, lunar_lander.hip `vc_far_late`), and once stepping through mgym_rollout.

Protocol reference: /root/reference src/testing.rs:65-134 (step, compare, reset when done); the loop being soaked is
src/box_2d/lunar_lander.rs:919-1167 behind `world.step(1/50, 180, 60)` (:1066).
"""
import numpy as np
import pytest

import modurl_gym_amd as mg
from oracle import oracle as ora

pytestmark = pytest.mark.gpu


def skilled_actions(rng, state, n):
    """uniform random actions, every fourth env under a crude stabilising controller (so that some episodes end asleep on the pad)"""
    a = rng.integers(0, 4, n).astype(np.uint32)
    skilled = (np.arange(n) % 4 == 0)
    vy, ang, w = state[4], state[2], state[5]
    ctrl = np.where((np.abs(ang) > 0.05) | (np.abs(w) > 0.3), np.where(ang + 0.5 * w > 0, 3, 1), np.where(vy < -0.6, 2, 0)).astype(np.uint32)
    a[skilled] = ctrl[skilled]
    return a


def soak(n, steps, seed=99, rollout_k=0):
    env = mg.VecEnv(mg.LUNARLANDER, n, seed=seed, enable_wind=True, auto_reset=True)
    ref = ora.OracleVec(ora.LUNARLANDER, n, seed=seed, enable_wind=True)
    assert np.array_equal(env.reset(), ref.reset(nthreads=16))
    rng = np.random.default_rng(1)
    episodes = landed = crashed = 0
    t = 0
    while t < steps:
        k = rollout_k if rollout_k else 1
        acts, exps = [], []
        for _ in range(k):
            a = skilled_actions(rng, ref.get_state(), n)
            obs, rew, done, trunc = ref.step(a, nthreads=16)
            ro = ref.reset(done, nthreads=16)                       # the engine's observation is the one after the fused reset
            exps.append((np.where(done.astype(bool)[None, :], ro, obs), rew, done, trunc))
            acts.append(a)
            d = done.astype(bool)
            episodes += int(d.sum()); landed += int((rew[d] == 100.0).sum()); crashed += int((rew[d] == -100.0).sum())
        if rollout_k:
            gobs, grew, gdone, gtrunc = env.rollout(np.stack(acts))
            gots = [(gobs[j], grew[j], gdone[j], gtrunc[j]) for j in range(k)]
        else:
            gots = [env.step(acts[0])]
        for j, (got, exp) in enumerate(zip(gots, exps)):
            for g, e, nm in zip(got, exp, ("obs", "reward", "done", "truncated")):
                if not np.array_equal(g.view(np.uint32) if g.dtype == np.float32 else g, e.view(np.uint32) if e.dtype == np.float32 else e):
                    bad = np.argwhere(g != e)
                    raise AssertionError(f"step {t + j}: {nm} differs at {bad[0]} ({len(bad)} words): {g[tuple(bad[0])]!r} vs {e[tuple(bad[0])]!r}")
        t += k
    env.sync()
    env.close()
    return episodes, landed, crashed


@pytest.mark.parametrize("block", [32, 64])
def test_soak_4096_envs_1500_steps_auto_reset_wind_every_word_equals_the_oracle(block, monkeypatch):
    monkeypatch.setenv("MGYM_LL_GENERAL_BLOCK", str(block))
    episodes, landed, crashed = soak(4096, 1500)
    assert episodes > 20000 and landed > 20 and crashed > 10000   # the horizon really holds landings asleep as well as crashes


def test_soak_with_every_velocity_constraint_in_the_global_workspace(monkeypatch):
    # MGYM_LL_VC_NEAR=0: nothing stays in LDS, so the main and the late contact launch (which run at the same time in the
    # overlapped order) both work in their slices of LLDev::vc_far for every constraint of every island and sub-step
    monkeypatch.setenv("MGYM_LL_VC_NEAR", "0")
    episodes, landed, _ = soak(8192, 500, seed=7)
    assert episodes > 10000


@pytest.mark.parametrize("n", [1, 33, 97, 1000])
def test_ragged_populations_every_word_equals_the_oracle(n):
    # populations that fill neither a 32-lane contact block, a 64-lane free-flight wave nor the 1 024-env padding of the
    # engine's allocation (one 288-word record per env): 400 steps with the fused auto-reset, every word against the oracle
    episodes, _, _ = soak(n, 400, seed=1234 + n)
    assert episodes >= (1 if n == 1 else n)


@pytest.mark.parametrize("blocks", [0, 12, 40])
def test_soak_with_other_lanes_per_contact_block(blocks, monkeypatch):
    # the single-launch step deals its contact list out over MGYM_LL_CONTACT_BLOCKS blocks (default 900: 8 lanes per block at this
    # population); 0 = always 32 lanes per block, 12 / 40 = ~30 / ~9 lanes — the grouping must not change a word
    monkeypatch.setenv("MGYM_LL_CONTACT_BLOCKS", str(blocks))
    episodes, _, _ = soak(4096, 500, seed=4321)
    assert episodes > 6000
