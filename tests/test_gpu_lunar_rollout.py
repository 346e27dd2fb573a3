"""mgym_rollout for LunarLanderV3 as ONE persistent launch (modurl_gym_amd/csrc/ll_roll.h): K steps in which every environment advances
as soon as it is ready — free-flight environments resident in registers, contact-path environments through device queues in batches of
their own kind, finished ones through a reset queue.  Per-environment results must be those of K mgym_step calls and of the CPU oracle,
word for word (the loop being fused: /root/reference src/box_2d/lunar_lander.rs:919-1167 called K times, reset :727-917 in between).
"""
import os

import numpy as np
import pytest

import modurl_gym_amd as mg
from oracle import oracle as ora
from test_gpu_lunar_soak import skilled_actions, soak

pytestmark = pytest.mark.gpu


def words(a):
    return a.view(np.uint32) if a.dtype == np.float32 else a


# The free-flight helper waves (MGYM_LL_ROLL_HELPER=1) are off in the product: a launch with them hung once in ~75 stress runs.  Their parity cases stay in
# the file and run when MGYM_TEST_ROLL_HELPER=1 is set; the suite the driver runs must not be able to hang.
HELPER = 1 if os.environ.get("MGYM_TEST_ROLL_HELPER") == "1" else 0


@pytest.mark.parametrize("n,K,auto_reset,helper", [(4096, 8, True, 0), (4096, 16, True, HELPER), (1000, 24, False, HELPER), (97, 9, True, 0), (1, 8, True, HELPER), (8192, 12, True, HELPER)])
def test_rollout_equals_k_steps_word_for_word(n, K, auto_reset, helper, monkeypatch):
    # two handles with the same seed: one steps K times, the other makes one mgym_rollout call; then again from where they stand.
    # helper: with / without the free-flight helper waves beside the main launch (by default they come with populations from 163 840 envs)
    kw = dict(seed=321, enable_wind=True, auto_reset=auto_reset)
    monkeypatch.setenv("MGYM_LL_ROLL_HELPER", str(helper))
    monkeypatch.setenv("MGYM_LL_ROLLOUT_MIN_K", "8")   # (by default populations below 163 840 envs take rollouts shorter than 12 steps as K steps)
    a_env, b_env = mg.VecEnv(mg.LUNARLANDER, n, **kw), mg.VecEnv(mg.LUNARLANDER, n, **kw)
    monkeypatch.delenv("MGYM_LL_ROLL_HELPER")
    monkeypatch.delenv("MGYM_LL_ROLLOUT_MIN_K")
    assert a_env.info()["rollout"] == "persistent_launch" and a_env.info()["rollout_min_k"] == "8" and (int(b_env.info()["rollout_helper_blocks"]) > 0) == (helper == 1 and n >= 64)
    assert np.array_equal(a_env.reset(), b_env.reset())
    rng = np.random.default_rng(5)
    for rep in range(6):
        acts = rng.integers(0, 4, (K, n)).astype(np.uint32)
        got = b_env.rollout(acts)
        for k in range(K):
            exp = a_env.step(acts[k])
            for g, e, nm in zip((got[0][k], got[1][k], got[2][k], got[3][k]), exp, ("obs", "reward", "done", "truncated")):
                assert np.array_equal(words(g), words(e)), f"round {rep} step {k}: {nm} differs at {np.argwhere(words(g) != words(e))[:3].tolist()}"
        assert np.array_equal(words(a_env.observation()), words(b_env.observation()))
    assert np.array_equal(words(a_env.get_state()), words(b_env.get_state()))    # incl. episode and step counters
    assert a_env.episode_count() == b_env.episode_count()
    a_env.sync(), b_env.sync()
    a_env.close(), b_env.close()


@pytest.mark.parametrize("helper", [0, 1] if HELPER else [0])
def test_rollout_soak_every_word_equals_the_oracle(helper, monkeypatch):
    # 4 096 envs x 960 steps in rollouts of 16, skilled policy (landings asleep as well as crashes and fly-aways), fused auto-reset;
    # without and with the free-flight helper waves
    monkeypatch.setenv("MGYM_LL_ROLL_HELPER", str(helper))
    episodes, landed, crashed = soak(4096, 960, seed=77, rollout_k=16)
    assert episodes > 12000 and landed > 10 and crashed > 6000


def test_rollout_mixes_with_steps_resets_and_state_imports():
    # rollout, eager steps, a caller reset of some envs, a state import and another rollout: the persistent launch leaves the handle in the
    # state K steps would have left it in (next step's contact list, prepared resets that no longer fit, counters)
    n = 2048
    env = mg.VecEnv(mg.LUNARLANDER, n, seed=11, enable_wind=True, auto_reset=True)
    ref = ora.OracleVec(ora.LUNARLANDER, n, seed=11, enable_wind=True)
    assert np.array_equal(env.reset(), ref.reset(nthreads=8))
    rng = np.random.default_rng(2)

    def ref_steps(acts):
        out = []
        for a in acts:
            obs, rew, done, trunc = ref.step(a, nthreads=8)
            ro = ref.reset(done, nthreads=8)
            out.append((np.where(done.astype(bool)[None, :], ro, obs), rew, done, trunc))
        return out

    for phase in range(4):
        acts = np.stack([skilled_actions(rng, ref.get_state(), n) for _ in range(12)])   # (actions drawn from the state before the block: any policy will do)
        exp = ref_steps(acts)
        got = env.rollout(acts)
        for k in range(12):
            for g, e in zip((got[0][k], got[1][k], got[2][k], got[3][k]), exp[k]):
                assert np.array_equal(words(g), words(e)), f"phase {phase} rollout step {k}"
        for _ in range(5):   # eager steps on the same handle
            a = rng.integers(0, 4, n).astype(np.uint32)
            (e,) = ref_steps([a])
            g = env.step(a)
            for x, y in zip(g, e):
                assert np.array_equal(words(x), words(y)), f"phase {phase} eager step"
        if phase == 1:       # caller-side reset of a third of the population
            m = (np.arange(n) % 3 == 0).astype(np.uint8)
            assert np.array_equal(words(np.ascontiguousarray(env.reset(m)[:, m.astype(bool)])), words(np.ascontiguousarray(ref.reset(m, nthreads=8)[:, m.astype(bool)])))
        if phase == 2:       # checkpoint / restore through the Testable seam
            blob = env.get_state()
            env.set_state(blob), ref.set_state(blob)
    env.sync()
    env.close()


def test_rollout_without_outputs_and_short_rollouts_fall_back():
    n = 512
    env = mg.VecEnv(mg.LUNARLANDER, n, seed=3, enable_wind=False, auto_reset=True)
    twin = mg.VecEnv(mg.LUNARLANDER, n, seed=3, enable_wind=False, auto_reset=True)
    env.reset(), twin.reset()
    acts = np.random.default_rng(0).integers(0, 4, (14, n)).astype(np.uint32)
    da = mg.DeviceArray.from_numpy(acts, 0)
    env.rollout_device(da, 14, None, None, None, None)     # every output pointer NULL: only the state advances (14 >= rollout_min_k: the persistent launch)
    for k in range(14):
        twin.step(acts[k])
    assert np.array_equal(words(env.observation()), words(twin.observation()))
    short = np.random.default_rng(1).integers(0, 4, (3, n)).astype(np.uint32)   # K below rollout_min_k: K x mgym_step inside the engine
    g = env.rollout(short)
    for k in range(3):
        e = twin.step(short[k])
        assert all(np.array_equal(words(x[k]), words(y)) for x, y in zip(g, e))
    env.sync(), twin.sync()


def test_rollout_captured_into_a_graph_replays_like_eager_rollouts(monkeypatch):
    """mgym_graph_begin / _end around mgym_rollout: while the stream is being captured the engine records K steps instead of the persistent launch
    (replays of a captured persistent launch aborted inside the HIP runtime — 35 of 40 fresh processes at the end of round 4, unexplained; eager launches
    of the same kernel and captured steps do not); replays — state is read when the graph runs — must give the words of eager rollouts (the persistent
    launch) from the same state.  Actions and outputs are the captured device buffers, rewritten between replays."""
    n, K = 16384, 8
    kw = dict(seed=41, enable_wind=True, auto_reset=True)
    monkeypatch.setenv("MGYM_LL_ROLL_HELPER", str(HELPER))   # (with the helper waves only under MGYM_TEST_ROLL_HELPER=1, see above;
    monkeypatch.setenv("MGYM_LL_ROLLOUT_MIN_K", "8")          #  populations below 163 840 envs take rollouts shorter than 12 steps as K steps)
    g_env, e_env = mg.VecEnv(mg.LUNARLANDER, n, **kw), mg.VecEnv(mg.LUNARLANDER, n, **kw)
    monkeypatch.delenv("MGYM_LL_ROLL_HELPER")
    monkeypatch.delenv("MGYM_LL_ROLLOUT_MIN_K")
    assert (int(g_env.info()["rollout_helper_blocks"]) > 0) == (HELPER == 1)
    assert np.array_equal(g_env.reset(), e_env.reset())
    rng = np.random.default_rng(8)
    for t in range(64):   # until contacts, crashes and resets are frequent
        a = rng.integers(0, 4, n).astype(np.uint32)
        g_env.step(a), e_env.step(a)
    d_act = mg.DeviceArray.from_numpy(np.zeros((K, n), np.uint32))
    d_obs, d_rew = mg.DeviceArray((K, 8, n), np.float32), mg.DeviceArray((K, n), np.float32)
    d_done, d_trunc = mg.DeviceArray((K, n), np.uint8), mg.DeviceArray((K, n), np.uint8)
    graph = g_env.graph_capture(lambda: g_env.rollout_device(d_act, K, d_obs, d_rew, d_done, d_trunc))
    assert np.array_equal(words(g_env.get_state()), words(e_env.get_state()))   # capturing advanced nothing
    finished = 0
    for rep in range(4):
        acts = rng.integers(0, 4, (K, n)).astype(np.uint32)
        d_act.copy_from(acts)
        g_env.graph_launch(graph)
        g_env.sync()
        exp = e_env.rollout(acts)
        for g, e, nm in zip((d_obs.numpy(), d_rew.numpy(), d_done.numpy(), d_trunc.numpy()), exp, ("obs", "reward", "done", "truncated")):
            assert np.array_equal(words(g), words(e)), f"replay {rep}: {nm}"
        finished += int(exp[2].sum())
        if rep == 1:   # eager steps between replays
            a = rng.integers(0, 4, n).astype(np.uint32)
            for g, e in zip(g_env.step(a), e_env.step(a)):
                assert np.array_equal(words(g), words(e))
    assert finished > 500
    assert np.array_equal(words(g_env.get_state()), words(e_env.get_state()))
    g_env.graph_destroy(graph)
    g_env.close(), e_env.close()
