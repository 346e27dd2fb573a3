"""GPU parity tests proper: the HIP engine, called through the C ABI (libmgym.so), against
the CPU oracle on identical seeds/states/actions, and against the reference's golden vectors.

Bar (BASELINE.json north_star): obs/reward within 1e-5 relative f32, done/truncated and
discrete outputs bit-exact.  CartPole/MountainCar are in fact required to be BIT-IDENTICAL
here (same operation order, no contraction, libm-identical sin/cos), which subsumes the bar.
"""
import numpy as np
import pytest

import modurl_gym_amd as mg
from harness import VecAdapter, replay
from oracle import oracle as ora

pytestmark = pytest.mark.gpu

PAIRS = {"cartpole": (mg.CARTPOLE, ora.CARTPOLE, 2), "mountain_car": (mg.MOUNTAINCAR, ora.MOUNTAINCAR, 3)}


def both(name, n, seed=1, base=0, **kw):
    k, ok, nact = PAIRS[name] if name in PAIRS else (mg.MOUNTAINCAR_CONT, ora.MOUNTAINCAR_CONT, 0)
    okw = dict(kw)
    okw.pop("auto_reset", None)
    return mg.VecEnv(k, n, seed=seed, env_id_base=base, **kw), ora.OracleVec(ok, n, seed=seed, env_id_base=base, **okw), nact


def assert_same(got, exp, what=""):
    names = ("obs", "reward", "done", "truncated")
    for g, e, nm in zip(got, exp, names):
        if not np.array_equal(g, e):
            bad = np.argwhere(g != e)
            raise AssertionError(f"{what}{nm}: {len(bad)} mismatches, first at {bad[0]}: {g[tuple(bad[0])]} vs {e[tuple(bad[0])]}")


# ------------------------------------------------------------------ golden vectors --------
def test_cartpole_against_python_through_abi(golden):
    env = mg.VecEnv(mg.CARTPOLE, 1, seed=1)
    worst_obs, worst_rew = replay(VecAdapter(env, "cartpole"), golden("cartpole"))  # 1e-4/1e-4, testing.rs:42-45
    assert worst_obs < 2e-7 and worst_rew == 0.0


def test_mountain_car_against_python_through_abi(golden):
    env = mg.VecEnv(mg.MOUNTAINCAR, 1)
    worst_obs, worst_rew = replay(VecAdapter(env, "mountain_car"), golden("mountain_car"))
    assert worst_obs < 2e-7 and worst_rew == 0.0


# --------------------------------------------------- reference unit tests, same shape -----
def test_cartpole_like_reference_unit_tests():
    env = mg.CartPoleV1()  # cartpole.rs:365-390
    state = env.reset()
    assert state.shape == (4,)
    info = env.step(np.uint32(0))
    assert info.state.shape == (4,) and info.reward == 1.0 and not info.done
    env = mg.CartPoleV1()  # cartpole.rs:392-403 #[should_panic]: a [1]-shaped action is rejected
    env.reset()
    with pytest.raises(mg.InvalidActionError):
        env.step(np.array([1], np.uint32))
    with pytest.raises(mg.InvalidActionError):
        env.step(np.uint32(2))
    env = mg.CartPoleV1()  # cartpole.rs:405-434
    env.reset()
    info = env.step(np.uint32(1))
    assert info.reward == 1.0 and not info.done
    done = False
    for _ in range(50):
        done = env.step(np.uint32(1)).done
        if done:
            break
    assert done


def test_mountain_car_like_reference_unit_tests():
    env = mg.MountainCarV0()  # mountain_car.rs:347-372
    assert env.reset().shape == (2,)
    info = env.step(np.uint32(0))
    assert info.state.shape == (2,) and info.reward == -1.0 and not info.done and not info.truncated
    with pytest.raises(mg.InvalidActionError):  # mountain_car.rs:374-385
        env.step(np.uint32(3))
    assert env.action_space() == ("Discrete", 3)


# ----------------------------------------------------------------- oracle parity ----------
@pytest.mark.parametrize("name", ["cartpole", "mountain_car"])
@pytest.mark.parametrize("n", [1, 3, 5, 64, 1023, 1025, 65536])
def test_rollout_bit_identical_to_oracle(name, n):
    env, ref, nact = both(name, n, seed=42 + n)
    assert_same((env.reset(),), (ref.reset(),), "reset ")
    rng = np.random.default_rng(n)
    for t in range(120 if n <= 1025 else 60):
        a = rng.integers(0, nact, n).astype(np.uint32)
        got, exp = env.step(a), ref.step(a)
        assert_same(got, exp, f"step {t} ")
        mask = exp[2] | exp[3]
        if t % 3 == 0:  # reset what finished (two of three steps keep stepping finished envs: no auto-reset)
            env.reset(mask), ref.reset(mask)
            assert np.array_equal(env.observation(), ref.get_state()[: env.obs_dim]), f"masked reset {t}"
    assert np.array_equal(env.get_state().view(np.uint32), ref.get_state().view(np.uint32))


def test_empty_population():
    env = mg.VecEnv(mg.CARTPOLE, 0)
    assert env.reset().shape == (4, 0)
    obs, rew, done, trunc = env.step(np.zeros(0, np.uint32))
    assert obs.shape == (4, 0) and rew.shape == (0,)


def test_cartpole_config1_single_env_long_loop():
    # BASELINE configs[0]: CartPole-v1, 1 env, step() loop with resets — plumbing through the ABI
    env, ref, _ = both("cartpole", 1, seed=7)
    env.reset(), ref.reset()
    rng = np.random.default_rng(0)
    for t in range(1500):
        a = rng.integers(0, 2, 1).astype(np.uint32)
        got, exp = env.step(a), ref.step(a)
        assert_same(got, exp, f"step {t} ")
        if exp[2][0] or exp[3][0]:
            assert np.array_equal(env.reset(), ref.reset())


def test_cartpole_full_size_1mi_envs():
    # BASELINE configs[1]: 1 048 576 envs, f32 SoA — bit-exact vs the oracle over a short rollout,
    # plus size-independent invariants of the episode state machine.
    n = 1 << 20
    env, ref, _ = both("cartpole", n, seed=2024)
    assert np.array_equal(env.reset(), ref.reset())
    rng = np.random.default_rng(1)
    fell = np.zeros(n, bool)
    for t in range(30):
        a = rng.integers(0, 2, n).astype(np.uint32)
        got, exp = env.step(a, ), ref.step(a, nthreads=8)
        assert_same(got, exp, f"step {t} ")
        obs, rew, done, trunc = got
        # invariants (steps < 500): done is exactly the threshold predicate on the returned observation,
        # recomputed every step (cartpole.rs:291-294,310 — it is NOT sticky); reward is 1 unless the env
        # is past its first termination (:319-346)
        pred = (np.abs(obs[0]) > np.float32(2.4)) | (np.abs(obs[2]) > np.float32(12 * 2 * np.float32(np.pi) / 360))
        assert np.array_equal(done.astype(bool), pred) and not trunc.any()
        assert np.array_equal(rew == 0.0, pred & fell)
        fell |= pred
    assert 0.1 < fell.mean() < 1.0  # random policy: a good part of the population has fallen by step 30


def test_mountain_car_full_size_1mi_envs():
    n = 1 << 20  # BASELINE configs[2]
    env, ref, _ = both("mountain_car", n, seed=11)
    assert np.array_equal(env.reset(), ref.reset())
    rng = np.random.default_rng(2)
    for t in range(30):
        a = rng.integers(0, 3, n).astype(np.uint32)
        got, exp = env.step(a), ref.step(a, nthreads=8)
        assert_same(got, exp, f"step {t} ")
    obs = got[0]
    assert (obs[0] >= np.float32(-1.2)).all() and (obs[0] <= np.float32(0.6)).all() and (np.abs(obs[1]) <= np.float32(0.07)).all()
    assert (got[1] == -1.0).all() and not got[3].any()


def test_mountain_car_continuous_vs_oracle():
    n = 1 << 20  # BASELINE configs[2], second half — NOT in the reference: parity unpinned
    env, ref, _ = both("mountain_car_cont", n, seed=5)
    assert np.array_equal(env.reset(), ref.reset())
    rng = np.random.default_rng(3)
    for t in range(20):
        a = rng.uniform(-1.5, 1.5, n).astype(np.float32)
        got, exp = env.step(a), ref.step(a, nthreads=8)
        assert_same(got, exp, f"step {t} ")
    small = mg.VecEnv(mg.MOUNTAINCAR_CONT, 4)
    with pytest.raises(mg.InvalidActionError):
        small.step(np.array([0.0, np.nan, 0.0, 0.0], np.float32))


# ----------------------------------------------------- forced branches (rare paths) --------
def u32col(x):
    return np.array(x, np.uint32).view(np.float32)


@pytest.mark.parametrize("sb", [False, True])
@pytest.mark.parametrize("euler", [True, False])
def test_cartpole_all_branches_vs_oracle(sb, euler):
    n = 4096
    env, ref, _ = both("cartpole", n, seed=3, sutton_barto_reward=sb, is_euler=euler)
    env.reset(), ref.reset()
    rng = np.random.default_rng(9)
    s = ref.get_state()
    s[0] = rng.uniform(-2.6, 2.6, n)          # x around the +-2.4 threshold
    s[1] = rng.uniform(-3, 3, n)
    s[2] = rng.uniform(-0.25, 0.25, n)        # theta around +-0.2094
    s[3] = rng.uniform(-3, 3, n)
    s[4] = u32col(rng.integers(490, 503, n))  # steps around the 500-step truncation
    s[5] = u32col(rng.integers(-1, 3, n).astype(np.int32).view(np.uint32))  # sbt None / Some(k)
    ref.set_state(s), env.set_state(s)
    seen_trunc = seen_post = 0
    for t in range(12):
        a = rng.integers(0, 2, n).astype(np.uint32)
        got, exp = env.step(a), ref.step(a)
        assert_same(got, exp, f"step {t} ")
        seen_trunc += int(exp[3].sum())
        seen_post += int(((exp[2] == 1) & (exp[1] == (-1.0 if sb else 0.0))).sum())
    assert seen_trunc > 0 and seen_post > 0  # truncation-beats-termination and post-terminal paths were hit
    assert np.array_equal(env.get_state().view(np.uint32), ref.get_state().view(np.uint32))


def test_cartpole_episode_counter_wraps_at_2_pow_20_as_documented():
    # include/mgym.h (mgym_get_state): the episode number lives in 20 bits of the counter word; an env stepped across
    # its 1 048 576th reset draws the initial states of episodes 0, 1, ... again.  Checked against the oracle AND
    # against a fresh handle (whose first resets ARE episodes 0, 1, ...), so the claim does not rest on the oracle's
    # own masking.
    n = 4096
    env, ref, _ = both("cartpole", n, seed=77, auto_reset=True)
    fresh = mg.VecEnv(mg.CARTPOLE, n, seed=77)
    env.reset(), ref.reset()
    s = ref.get_state()
    s[6] = u32col(np.full(n, (1 << 20) - 1))   # the next reset of every env is number 2^20 - 1, the one after wraps to 0
    ref.set_state(s), env.set_state(s)
    assert np.array_equal(env.get_state()[6].view(np.uint32), np.full(n, (1 << 20) - 1, np.uint32))
    big = s.copy(); big[6] = u32col(np.full(n, (5 << 20) + 123)); env.set_state(big)   # imports are reduced modulo 2^20
    assert np.array_equal(env.get_state()[6].view(np.uint32), np.full(n, 123, np.uint32))
    env.set_state(s)
    rng = np.random.default_rng(5)
    resets = np.zeros(n, np.int64)
    first_after_wrap = np.full((4, n), np.nan, np.float32)
    for t in range(120):
        a = rng.integers(0, 2, n).astype(np.uint32)
        got, exp = env.step(a), ref.step(a)
        assert_same(got[1:], exp[1:], f"step {t} ")
        fin = (exp[2] | exp[3]).astype(bool)
        ref.reset(mask=fin.astype(np.uint8))
        obs = env.observation()
        assert np.array_equal(obs, ref.get_state()[:4]), f"step {t}: post-reset state"
        take = fin & (resets == 1)     # second reset after the import = episode number 0 again
        first_after_wrap[:, take] = obs[:, take]
        resets += fin
    assert (resets >= 2).sum() > n // 2
    ep = env.get_state()[6].view(np.uint32)
    assert np.array_equal(ep, ((1 << 20) - 1 + resets) % (1 << 20))
    ep0 = fresh.reset()                # a fresh handle's first reset is episode 0 of the same (seed, env id)
    hit = resets >= 2
    assert np.array_equal(first_after_wrap[:, hit], ep0[:, hit])
    fresh.close()


def test_cartpole_large_angles_use_table_reduction():
    # stepping long after termination spins theta up: exercises reduce_fast beyond pi/4 and the
    # |x| >= 120 table reduction of mgym_math.h on the device
    n = 8192
    env, ref, _ = both("cartpole", n, seed=4)
    rng = np.random.default_rng(10)
    s = ref.get_state()
    mag = np.concatenate([rng.uniform(0, 16, n // 2), 10.0 ** rng.uniform(np.log10(120), 6, n // 2)])
    s[2] = (mag * rng.choice([-1.0, 1.0], n)).astype(np.float32)
    s[3] = rng.uniform(-1, 1, n)
    ref.set_state(s), env.set_state(s)
    a = rng.integers(0, 2, n).astype(np.uint32)
    assert_same(env.step(a), ref.step(a), "large-angle ")


def test_mountain_car_clamps_wall_goal_vs_oracle():
    n = 4096
    for gv in (0.0, 0.03):
        env, ref, _ = both("mountain_car", n, seed=6, goal_velocity=gv)
        rng = np.random.default_rng(12)
        s = ref.get_state()
        s[0] = rng.choice([-1.2, -1.19, 0.49, 0.5, 0.59, 0.6, -0.5], n).astype(np.float32)
        s[1] = rng.uniform(-0.07, 0.07, n)
        ref.set_state(s), env.set_state(s)
        hit_goal = hit_wall = 0
        for t in range(8):
            a = rng.integers(0, 3, n).astype(np.uint32)
            got, exp = env.step(a), ref.step(a)
            assert_same(got, exp, f"gv={gv} step {t} ")
            hit_goal += int(exp[2].sum())
            hit_wall += int(((exp[0][0] == np.float32(-1.2)) & (exp[0][1] == 0.0)).sum())
        assert hit_goal > 0 and hit_wall > 0


# ----------------------------------------------------------- error behaviour, API edges -----
def test_invalid_action_is_reported_and_sticky_until_sync():
    env = mg.VecEnv(mg.CARTPOLE, 1000)
    env.reset()
    a = np.zeros(1000, np.uint32)
    a[777] = 2
    with pytest.raises(mg.InvalidActionError):
        env.step(a)
    env.step(np.zeros(1000, np.uint32))  # cleared by the failing sync; engine stays usable


def test_unaligned_caller_buffers_take_the_scalar_kernel():
    n = 1000
    env, ref, _ = both("cartpole", n, seed=8)
    env.reset(), ref.reset()
    a = np.random.default_rng(0).integers(0, 2, n).astype(np.uint32)
    big = mg.DeviceArray(n + 8, np.uint32)
    rew = mg.DeviceArray(n + 8, np.float32)
    done = mg.DeviceArray(n + 8, np.uint8)
    trunc = mg.DeviceArray(n + 8, np.uint8)
    obs = mg.DeviceArray(4 * n + 8, np.float32)
    big.copy_from(np.concatenate([[0], a, np.zeros(7)]).astype(np.uint32))
    env.step_device(big.ptr + 4, obs.ptr + 4, rew.ptr + 4, done.ptr + 1, trunc.ptr + 3)  # all misaligned
    env.sync()
    exp = ref.step(a)
    assert np.array_equal(obs.numpy()[1:1 + 4 * n].reshape(4, n), exp[0])
    assert np.array_equal(rew.numpy()[1:1 + n], exp[1])
    assert np.array_equal(done.numpy()[1:1 + n], exp[2]) and np.array_equal(trunc.numpy()[3:3 + n], exp[3])


def test_null_outputs_and_zero_copy_observation():
    n = 4096
    env, ref, _ = both("cartpole", n, seed=13)
    env.reset(), ref.reset()
    a = np.random.default_rng(1).integers(0, 2, n).astype(np.uint32)
    act = mg.DeviceArray.from_numpy(a)
    env.step_device(act)  # every output NULL: the 50 B/env-step configuration minus reward/flags
    env.sync()
    exp = ref.step(a)
    assert np.array_equal(env.observation(), exp[0])
    ptr, stride = env.observation_device()
    assert ptr % 4096 == 0 and stride % 1024 == 0 and stride >= n


def test_auto_reset_flag_equals_step_then_masked_reset():
    n = 20000
    for name in ("cartpole", "mountain_car"):
        env, ref, nact = both(name, n, seed=21, auto_reset=True)
        env.reset(), ref.reset()
        if name == "mountain_car":  # start near the goal so some envs finish
            s = ref.get_state(); s[0] = 0.48; s[1] = 0.03; ref.set_state(s); env.set_state(s)
        rng = np.random.default_rng(4)
        for t in range(60):
            a = rng.integers(0, nact, n).astype(np.uint32)
            got, exp = env.step(a), ref.step(a)
            assert_same(got[1:], exp[1:], f"{name} step {t} ")  # reward/done/trunc of the step itself
            ref.reset(mask=exp[2] | exp[3])
            st = ref.get_state()
            assert np.array_equal(env.observation(), st[: env.obs_dim]), f"{name} step {t}: post-reset obs"
        assert np.array_equal(env.get_state().view(np.uint32), ref.get_state().view(np.uint32))


def test_reset_done_device_matches_masked_reset():
    n = 5000
    env, ref, _ = both("cartpole", n, seed=30)
    env.reset(), ref.reset()
    rng = np.random.default_rng(6)
    act, rew = mg.DeviceArray(n, np.uint32), mg.DeviceArray(n, np.float32)
    done, trunc = mg.DeviceArray(n, np.uint8), mg.DeviceArray(n, np.uint8)
    for t in range(40):
        a = rng.integers(0, 2, n).astype(np.uint32)
        act.copy_from(a)
        env.step_device(act, None, rew, done, trunc)
        env.reset_done_device(done, trunc)
        env.sync()
        exp = ref.step(a)
        ref.reset(mask=exp[2] | exp[3])
        assert np.array_equal(done.numpy(), exp[2])
    assert np.array_equal(env.get_state().view(np.uint32), ref.get_state().view(np.uint32))


def test_sharding_is_invisible_global_env_ids():
    # results are bit-identical however the population is sharded
    n = 6000
    whole = mg.VecEnv(mg.CARTPOLE, n, seed=77)
    parts = [mg.VecEnv(mg.CARTPOLE, c, seed=77, env_id_base=s) for s, c in ((0, 1500), (1500, 3000), (4500, 1500))]
    a = np.random.default_rng(8).integers(0, 2, (25, n)).astype(np.uint32)
    assert np.array_equal(whole.reset(), np.concatenate([p.reset() for p in parts], axis=1))
    for t in range(25):
        w = whole.step(a[t])
        ps = [p.step(a[t][s:s + c]) for p, (s, c) in zip(parts, ((0, 1500), (1500, 3000), (4500, 1500)))]
        assert np.array_equal(w[0], np.concatenate([x[0] for x in ps], axis=1))
        m = w[2] | w[3]
        whole.reset(m)
        for p, (s, c) in zip(parts, ((0, 1500), (1500, 3000), (4500, 1500))):
            p.reset(m[s:s + c])
    assert np.array_equal(whole.get_state()[:4], np.concatenate([p.get_state()[:4] for p in parts], axis=1))


def test_graph_replay_equals_eager():
    n = 8192
    env, ref, _ = both("cartpole", n, seed=31)
    env.reset(), ref.reset()
    a = np.random.default_rng(11).integers(0, 2, n).astype(np.uint32)
    act, rew = mg.DeviceArray.from_numpy(a), mg.DeviceArray(n, np.float32)
    done, trunc = mg.DeviceArray(n, np.uint8), mg.DeviceArray(n, np.uint8)

    def body():
        env.step_device(act, None, rew, done, trunc)
        env.reset_done_device(done, trunc)

    g = env.graph_capture(body)
    for t in range(10):
        env.graph_launch(g)
        exp = ref.step(a)
        ref.reset(mask=exp[2] | exp[3])
    env.sync()
    env.graph_destroy(g)
    assert np.array_equal(env.get_state().view(np.uint32), ref.get_state().view(np.uint32))


# ------------------------------------------------------------------ fused K-step rollout -----
@pytest.mark.parametrize("auto_reset", [False, True])
def test_cartpole_rollout_equals_k_steps(auto_reset):
    # mgym_rollout(K) == K x mgym_step, bit for bit (fused rollout keeps state in registers)
    n, K = 8192, 24
    a = np.random.default_rng(5).integers(0, 2, (K, n)).astype(np.uint32)
    fused = mg.VecEnv(mg.CARTPOLE, n, seed=17, auto_reset=auto_reset)
    plain = mg.VecEnv(mg.CARTPOLE, n, seed=17, auto_reset=auto_reset)
    ref = ora.OracleVec(ora.CARTPOLE, n, seed=17)
    assert np.array_equal(fused.reset(), plain.reset())
    ref.reset()
    obs, rew, done, trunc = fused.rollout(a)
    for t in range(K):
        o, r, d, tr = plain.step(a[t])
        assert np.array_equal(rew[t], r) and np.array_equal(done[t], d) and np.array_equal(trunc[t], tr), f"step {t}"
        eo, er, ed, et = ref.step(a[t])
        assert np.array_equal(r, er) and np.array_equal(d, ed)
        if auto_reset:
            ref.reset(mask=ed | et)
            assert np.array_equal(obs[t], ref.get_state()[:4]), f"step {t}: post-reset observation"
        else:
            assert np.array_equal(obs[t], eo), f"step {t}"
    assert np.array_equal(fused.get_state().view(np.uint32), plain.get_state().view(np.uint32))
    assert np.array_equal(fused.get_state().view(np.uint32), ref.get_state().view(np.uint32))


def test_rollout_fallback_paths():
    # families / shapes without a fused kernel fall back to K plain steps with identical results
    n, K = 1001, 5      # n % 4 != 0 -> fallback
    env, ref = mg.VecEnv(mg.CARTPOLE, n, seed=3), ora.OracleVec(ora.CARTPOLE, n, seed=3)
    env.reset(), ref.reset()
    a = np.random.default_rng(6).integers(0, 2, (K, n)).astype(np.uint32)
    obs, rew, done, trunc = env.rollout(a)
    for t in range(K):
        eo, er, ed, et = ref.step(a[t])
        assert np.array_equal(obs[t], eo) and np.array_equal(rew[t], er) and np.array_equal(done[t], ed)
    mc, mref = mg.VecEnv(mg.MOUNTAINCAR, 512, seed=3), ora.OracleVec(ora.MOUNTAINCAR, 512, seed=3)
    mc.reset(), mref.reset()
    a = np.random.default_rng(7).integers(0, 3, (K, 512)).astype(np.uint32)
    obs, rew, done, trunc = mc.rollout(a)
    for t in range(K):
        eo, er, ed, et = mref.step(a[t])
        assert np.array_equal(obs[t], eo) and np.array_equal(rew[t], er)


@pytest.mark.parametrize("name,okind,kind", [("mountain_car", ora.MOUNTAINCAR, mg.MOUNTAINCAR), ("mountain_car_cont", ora.MOUNTAINCAR_CONT, mg.MOUNTAINCAR_CONT)])
def test_mountain_car_fused_rollout_equals_k_steps(name, okind, kind):
    n, K = 4096, 40
    rng = np.random.default_rng(8)
    a = rng.integers(0, 3, (K, n)).astype(np.uint32) if kind == mg.MOUNTAINCAR else rng.uniform(-1.2, 1.2, (K, n)).astype(np.float32)
    env, ref = mg.VecEnv(kind, n, seed=4, auto_reset=True), ora.OracleVec(okind, n, seed=4)
    env.reset(), ref.reset()
    s = ref.get_state(); s[0] = 0.4; s[1] = 0.03; ref.set_state(s); env.set_state(s)   # near the goal: resets happen
    obs, rew, done, trunc = env.rollout(a)
    n_done = 0
    for t in range(K):
        eo, er, ed, et = ref.step(a[t])
        assert np.array_equal(rew[t], er) and np.array_equal(done[t], ed) and not trunc[t].any(), f"step {t}"
        ref.reset(mask=ed)
        assert np.array_equal(obs[t], ref.get_state()[:2]), f"step {t}"
        n_done += int(ed.sum())
    assert n_done > 0
    assert np.array_equal(env.get_state().view(np.uint32), ref.get_state().view(np.uint32))


def test_observation_aos_is_the_transposed_soa_view():
    for kind, nact in ((mg.CARTPOLE, 2), (mg.MOUNTAINCAR, 3), (mg.LUNARLANDER, 4)):
        env = mg.VecEnv(kind, 1000, seed=2)
        env.reset()
        obs, *_ = env.step(np.random.default_rng(0).integers(0, nact, 1000).astype(np.uint32))
        aos = env.observation_aos()
        assert aos.shape == (1000, env.obs_dim) and np.array_equal(aos, obs.T)
        assert np.array_equal(env.observation(), obs)


def test_many_envs_of_one_wave_finish_at_once():
    """The fused auto-reset compacts a wave's finished envs into a 64-wide list pass by pass: force whole waves (256
    envs) to finish in the same step (list longer than one pass) next to waves with none, against the oracle."""
    n = 4096
    env, ref, nact = both("cartpole", n, seed=22, auto_reset=True)
    env.reset(), ref.reset()
    s = ref.get_state()
    s[0, 256:1024] = 3.0          # envs 256..1023: beyond x_threshold -> three full waves finish at once
    s[0, 2048:2048 + 70] = -3.0   # 70 consecutive envs of one wave (two passes), the rest of it alive
    s[4, 3000:3100] = np.array([499] * 100, np.uint32).view(np.float32)  # truncations
    ref.set_state(s), env.set_state(s)
    rng = np.random.default_rng(5)
    for t in range(3):
        a = rng.integers(0, nact, n).astype(np.uint32)
        got, exp = env.step(a), ref.step(a)
        assert_same(got[1:], exp[1:], f"step {t} ")
        if t == 0:
            assert exp[2][256:1024].all() and exp[3][3000:3100].all()
        ref.reset(mask=exp[2] | exp[3])
        assert np.array_equal(env.observation(), ref.get_state()[:4]), f"step {t}: post-reset obs"
    assert np.array_equal(env.get_state().view(np.uint32), ref.get_state().view(np.uint32))


def test_bench_variant_1mi_envs_graph_replay_fused_reset_vs_oracle():
    """Exactly what bench.py times (BASELINE configs[1]): 1 048 576 envs, MGYM_FLAG_AUTO_RESET, hipGraph of 16 step
    launches, obs_out = NULL, action ring of 16 columns — 32 steps (two replays) against the oracle: reward / done /
    truncated of the last step of each replay and the full state blob afterwards, bit for bit."""
    n, ring = 1 << 20, 16
    env, ref, _ = both("cartpole", n, seed=0x5EED0001, auto_reset=True)
    env.reset_device(), ref.reset(nthreads=8)
    rng = np.random.default_rng(3)
    acts = rng.integers(0, 2, (ring, n)).astype(np.uint32)
    d_act = [mg.DeviceArray.from_numpy(acts[k]) for k in range(ring)]
    rew, done, trunc = mg.DeviceArray(n, np.float32), mg.DeviceArray(n, np.uint8), mg.DeviceArray(n, np.uint8)
    g = env.graph_capture(lambda: [env.step_device(d_act[k], None, rew, done, trunc) for k in range(ring)])
    finished = 0
    for rep in range(2):
        env.graph_launch(g)
        env.sync()
        for k in range(ring):
            exp = ref.step(acts[k], nthreads=8)
            m = exp[2] | exp[3]
            finished += int(m.sum())
            ref.reset(mask=m, nthreads=8)
        assert np.array_equal(rew.numpy(), exp[1]) and np.array_equal(done.numpy(), exp[2]) and np.array_equal(trunc.numpy(), exp[3]), f"replay {rep}"
        assert np.array_equal(env.get_state().view(np.uint32), ref.get_state().view(np.uint32)), f"replay {rep}: state blob"
    env.graph_destroy(g)
    assert finished > n                      # every env restarted more than once on average
    assert env.episode_count() == finished   # the kernels' ballot/popcount reduction of the done mask


def test_full_size_16mi_envs_true_hbm_regime_vs_oracle():
    """16 777 216 envs (0.8 GB of state + outputs: beyond the 256 MiB Infinity Cache), fused auto-reset, 12 steps
    against the oracle bit for bit (state blob, reward, flags), plus the finished-episode count."""
    n, steps = 1 << 24, 12
    env, ref, _ = both("cartpole", n, seed=3, auto_reset=True)
    env.reset_device(), ref.reset(nthreads=16)
    rng = np.random.default_rng(2)
    acts = [rng.integers(0, 2, n).astype(np.uint32) for _ in range(4)]
    d_act = [mg.DeviceArray.from_numpy(a) for a in acts]
    rew, done, trunc = mg.DeviceArray(n, np.float32), mg.DeviceArray(n, np.uint8), mg.DeviceArray(n, np.uint8)
    total = 0
    for t in range(steps):
        env.step_device(d_act[t % 4], None, rew, done, trunc)
        env.sync()
        exp = ref.step(acts[t % 4], nthreads=16)
        assert np.array_equal(done.numpy(), exp[2]) and np.array_equal(rew.numpy(), exp[1]) and np.array_equal(trunc.numpy(), exp[3]), f"step {t}"
        m = exp[2] | exp[3]
        total += int(m.sum())
        ref.reset(mask=m, nthreads=16)
    assert np.array_equal(env.get_state().view(np.uint32), ref.get_state().view(np.uint32))
    assert total > 0 and env.episode_count() == total


def test_cartpole_fast_math_exhaustive_on_gpu():
    """mgym_selftest_cartpole_math: the exhaustive bit-identity enumerations of tests/native/cartpole_fast_check.cpp,
    evaluated by gfx950 kernels (hardware v_rcp_f32 / fused multiply-add / IEEE divide): all four counts must be 0."""
    import ctypes as C
    from modurl_gym_amd import _lib
    out = (C.c_uint64 * 5)()
    assert _lib.load().mgym_selftest_cartpole_math(0, out) == _lib.OK
    assert list(out) == [0, 0, 0, 0, 0], f"mismatches (sincos, x/M, n/d, step, lock-step sincos/cos below 120) = {list(out)}"


def test_episode_count_counts_finished_env_steps():
    for name in ("cartpole", "mountain_car"):
        n = 5000
        env, ref, nact = both(name, n, seed=9)           # no auto-reset: finished envs keep reporting done
        env.reset(), ref.reset()
        if name == "mountain_car":
            s = ref.get_state(); s[0] = 0.48; s[1] = 0.03; ref.set_state(s); env.set_state(s)
        rng = np.random.default_rng(1)
        total = 0
        for t in range(40):
            a = rng.integers(0, nact, n).astype(np.uint32)
            got, exp = env.step(a), ref.step(a)
            assert_same(got, exp, f"{name} step {t} ")
            total += int((exp[2] | exp[3]).sum())
        assert total > 0 and env.episode_count() == total, name


def policy_actions(policy_seed, call, K, n, base=0):
    """The documented on-device uniform policy of mgym_rollout_uniform, restated with the oracle's Philox."""
    out = np.zeros((K, n), np.uint32)
    for g0 in range(0, n, 4):
        gid = base + g0
        for blk in range((K + 31) // 32):
            w = ora.philox([gid & 0xFFFFFFFF, gid >> 32, call, 0x40000000 + blk], [policy_seed & 0xFFFFFFFF, policy_seed >> 32])
            for t in range(blk * 32, min(K, blk * 32 + 32)):
                for k in range(min(4, n - g0)):
                    out[t, g0 + k] = (w[k] >> (t & 31)) & 1
    return out


@pytest.mark.parametrize("auto_reset", [False, True])
def test_rollout_uniform_policy_equals_oracle_driven_by_the_same_bits(auto_reset):
    n, K, base = 2048, 70, 4096
    env = mg.VecEnv(mg.CARTPOLE, n, seed=17, env_id_base=base, auto_reset=auto_reset)
    ref = ora.OracleVec(ora.CARTPOLE, n, seed=17, env_id_base=base)
    env.reset(), ref.reset()
    for call in range(2):
        acts, obs, rew, done, trunc = env.rollout_uniform(0xABCDEF12345, K)
        exp_a = policy_actions(0xABCDEF12345, call, K, n, base)
        assert np.array_equal(acts, exp_a), f"call {call}: drawn actions"
        assert 0.45 < acts.mean() < 0.55
        for t in range(K):
            eo, er, ed, et = ref.step(exp_a[t])
            assert np.array_equal(rew[t], er) and np.array_equal(done[t], ed) and np.array_equal(trunc[t], et), f"call {call} step {t}"
            if auto_reset:
                ref.reset(mask=ed | et)
                assert np.array_equal(obs[t], ref.get_state()[:4])
            else:
                assert np.array_equal(obs[t], eo)
    assert np.array_equal(env.get_state().view(np.uint32), ref.get_state().view(np.uint32))
    with pytest.raises(mg.MgymError):
        mg.VecEnv(mg.CARTPOLE, 8, env_id_base=2).rollout_uniform(1, 4)   # env_id_base must be a multiple of 4
    with pytest.raises(mg.MgymError):
        mg.VecEnv(mg.LUNARLANDER, 8).rollout_uniform(1, 4)               # classic control only


@pytest.mark.parametrize("kind,okind", [(mg.MOUNTAINCAR, ora.MOUNTAINCAR), (mg.MOUNTAINCAR_CONT, ora.MOUNTAINCAR_CONT)])
def test_rollout_uniform_policy_for_mountain_car_equals_oracle_driven_by_the_same_draws(kind, okind):
    """mgym_rollout_uniform for the MountainCar families (include/mgym.h): Discrete(3) — 16-bit halves of the policy stream's words, action (3 h) >> 16;
    Box(-1, 1) — the top 24 bits u of a word, force u 2^-23 - 1.  The draws are restated here with the oracle's Philox; the oracle stepped with them
    (fused auto-reset) must give the engine's words."""
    n, K, base, seed = 1024, 37, 8192, 0x1234ABCD5678
    cont = kind == mg.MOUNTAINCAR_CONT
    env = mg.VecEnv(kind, n, seed=19, env_id_base=base, auto_reset=True)
    ref = ora.OracleVec(okind, n, seed=19, env_id_base=base)
    env.reset(), ref.reset()
    for call in range(2):
        acts, obs, rew, done, trunc = env.rollout_uniform(seed, K)
        exp = np.zeros((K, n), np.uint32)
        for g0 in range(0, n, 4):
            gid = base + g0
            for t in range(K):
                w = ora.philox([gid & 0xFFFFFFFF, gid >> 32, call, 0x40000000 + (t if cont else t // 2)], [seed & 0xFFFFFFFF, seed >> 32])
                for k in range(4):
                    if cont:
                        exp[t, g0 + k] = np.float32(np.float32(int(w[k]) >> 8) * np.float32(2.0 ** -23) - np.float32(1.0)).view(np.uint32)
                    else:
                        h = (int(w[k]) >> 16) if (t & 1) else (int(w[k]) & 0xFFFF)
                        exp[t, g0 + k] = (h * 3) >> 16
        assert np.array_equal(acts, exp), f"call {call}: drawn actions"
        if cont:
            f = acts.view(np.float32)
            assert f.min() >= -1.0 and f.max() < 1.0 and abs(float(f.mean())) < 0.02
        else:
            assert acts.max() == 2 and all(0.31 < float((acts == j).mean()) < 0.36 for j in range(3))
        for t in range(K):
            a = exp[t].view(np.float32) if cont else exp[t]
            eo, er, ed, et = ref.step(a)
            m = (ed | et).astype(np.uint8)
            ref.reset(mask=m)
            assert np.array_equal(rew[t], er) and np.array_equal(done[t], ed) and np.array_equal(trunc[t], et), f"call {call} step {t}"
            assert np.array_equal(obs[t], ref.get_state()[:2]), f"call {call} step {t}: observation"
    assert np.array_equal(env.get_state().view(np.uint32), ref.get_state().view(np.uint32))


def test_discrete_actions_must_be_integers():
    env = mg.VecEnv(mg.CARTPOLE, 4)
    env.reset()
    with pytest.raises(mg.InvalidActionError):
        env.step(np.array([0.0, 1.9, 1.0, 0.0], np.float32))   # would silently truncate to 1
    with pytest.raises(mg.InvalidActionError):
        env.step(np.array([True, False, True, False]))
    env.step(np.array([0, 1, 1, 0], np.int64))                 # any integer dtype is fine


@pytest.mark.parametrize("auto_reset", [True, False])
def test_rollout_linear_policy_equals_a_stepping_loop_with_the_same_weights_and_the_oracle(auto_reset):
    """mgym_rollout_linear (SURVEY 8f-1 policy hook, 8f-4 the step either side of the path): the in-kernel policy
    a = (((w0 x + w1 x_dot) + w2 theta) + w3 theta_dot) + b > 0, evaluated on the observation the env holds before the step, must give the actions,
    observations, rewards and flags of a loop that computes the same expression in f32 (numpy: each operation rounded, nothing fused) and calls
    mgym_step, and of the oracle driven by those actions.  Caller loop it fuses: cartpole.rs:251-348 with the policy in front."""
    n, K = 8192, 48
    pol = np.array([0.1, 0.5, -1.0, -1.0, -0.003], np.float32)   # pushes the wrong way: episodes end within a few dozen steps
    fused = mg.VecEnv(mg.CARTPOLE, n, seed=21, auto_reset=auto_reset)
    loop = mg.VecEnv(mg.CARTPOLE, n, seed=21, auto_reset=auto_reset)
    ref = ora.OracleVec(ora.CARTPOLE, n, seed=21)
    obs = loop.reset()
    assert np.array_equal(fused.reset(), obs) and np.array_equal(ref.reset(nthreads=8), obs)
    for rnd in range(3):
        acts, gobs, grew, gdone, gtrunc = fused.rollout_linear(pol, K)
        for t in range(K):
            lin = ((pol[0] * obs[0] + pol[1] * obs[1]) + pol[2] * obs[2]) + pol[3] * obs[3]     # float32 arrays: every operation rounds to f32
            a = ((lin + pol[4]) > 0).astype(np.uint32)
            assert np.array_equal(acts[t], a), f"round {rnd} step {t}: actions"
            o, r, d, tr = loop.step(a)
            eo, er, ed, et = ref.step(a, nthreads=8)
            if auto_reset:
                m = (ed | et).astype(np.uint8)
                ro = ref.reset(mask=m, nthreads=8)
                eo = np.where(m.astype(bool)[None, :], ro, eo)
            for g, e, x, nm in zip((gobs[t], grew[t], gdone[t], gtrunc[t]), (o, r, d, tr), (eo, er, ed, et), ("obs", "reward", "done", "truncated")):
                assert np.array_equal(g, e), f"round {rnd} step {t}: {nm} vs the stepping loop"
                assert np.array_equal(g, x), f"round {rnd} step {t}: {nm} vs the oracle"
            obs = o
    assert fused.episode_count() == loop.episode_count() > n // 2   # most envs fell at least once
    with pytest.raises(mg.MgymError):
        mg.VecEnv(mg.LUNARLANDER, 8).rollout_linear(np.zeros(9, np.float32), 4)   # classic control only


@pytest.mark.parametrize("kind,okind", [(mg.MOUNTAINCAR, ora.MOUNTAINCAR), (mg.MOUNTAINCAR_CONT, ora.MOUNTAINCAR_CONT)])
def test_rollout_linear_policy_for_mountain_car_equals_a_stepping_loop_and_the_oracle(kind, okind):
    """mgym_rollout_linear for the MountainCar families (mountain_car.rs:293-330 with the policy in front): Discrete(3) — the index of the largest of three
    scores (w_j0 position + w_j1 velocity) + b_j, the first of equal ones; Box(-1, 1) — the one score as the force.  Same f32 expression in numpy (every
    operation rounded, nothing fused) driving mgym_step and the oracle must give the same actions, observations, rewards and flags.  The discrete policy
    is the energy-pumping one (push with the velocity), so cars reach the goal and fused resets happen."""
    n, K = 4096, 64
    cont = kind == mg.MOUNTAINCAR_CONT
    pol = np.array([0.0, 40.0, 0.01] if cont else [[0.0, -30.0, 0.0], [0.0, 0.0, 1e-4], [0.0, 30.0, 0.0]], np.float32)
    fused, loop = mg.VecEnv(kind, n, seed=33, auto_reset=True), mg.VecEnv(kind, n, seed=33, auto_reset=True)
    ref = ora.OracleVec(okind, n, seed=33)
    obs = loop.reset()
    assert np.array_equal(fused.reset(), obs) and np.array_equal(ref.reset(nthreads=8), obs)
    finished = 0
    for rnd in range(5):
        acts, gobs, grew, gdone, gtrunc = fused.rollout_linear(pol, K)
        for t in range(K):
            if cont:
                a = ((pol[0] * obs[0] + pol[1] * obs[1]) + pol[2]).astype(np.float32)
                assert np.array_equal(acts[t].view(np.float32).view(np.uint32), a.view(np.uint32)), f"round {rnd} step {t}: actions"
            else:
                s = [((pol[j, 0] * obs[0] + pol[j, 1] * obs[1]) + pol[j, 2]).astype(np.float32) for j in range(3)]
                a = np.where(s[1] > s[0], 1, 0).astype(np.uint32)
                a = np.where(s[2] > np.maximum(s[0], s[1]), 2, a).astype(np.uint32)
                assert np.array_equal(acts[t], a), f"round {rnd} step {t}: actions"
            o, r, d, tr = loop.step(a)
            eo, er, ed, et = ref.step(a, nthreads=8)
            m = (ed | et).astype(np.uint8)
            ro = ref.reset(mask=m, nthreads=8)
            eo = np.where(m.astype(bool)[None, :], ro, eo)
            finished += int(m.sum())
            for g, e, x, nm in zip((gobs[t], grew[t], gdone[t], gtrunc[t]), (o, r, d, tr), (eo, er, ed, et), ("obs", "reward", "done", "truncated")):
                assert np.array_equal(g, e), f"round {rnd} step {t}: {nm} vs the stepping loop"
                assert np.array_equal(g, x), f"round {rnd} step {t}: {nm} vs the oracle"
            obs = o
    assert finished > 0 and fused.episode_count() == loop.episode_count() == finished
