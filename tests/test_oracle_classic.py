"""CPU oracle vs the reference's golden vectors and known-answer tests (no GPU)."""
import numpy as np
import pytest

from harness import VecAdapter, replay
from oracle import oracle as ora


def u32col(x):
    return np.array(x, np.uint32).view(np.float32)


# ---------------------------------------------------------------- RNG KATs ----
def test_philox_known_answers():
    # Random123 kat_vectors: philox4x32 10 rounds
    assert ora.philox([0, 0, 0, 0], [0, 0]) == [0x6627E8D5, 0xE169C58D, 0xBC57AC4C, 0x9B00DBD8]
    assert ora.philox([0xFFFFFFFF] * 4, [0xFFFFFFFF] * 2) == [0x408F276D, 0x41C83B0E, 0xA20BC7C6, 0x6D5451FD]
    assert ora.philox([0x243F6A88, 0x85A308D3, 0x13198A2E, 0x03707344], [0xA4093822, 0x299F31D0]) == [
        0xD16CFE09, 0x94FDCCEB, 0x5001E420, 0x24126EA1]


def test_uniform_conversions():
    L = ora.lib()
    assert L.ora_u53(0, 0) == 0.0
    assert L.ora_u53(0xFFFFFFFF, 0xFFFFFFFF) == 1.0 - 2.0 ** -53
    assert L.ora_u23(0) == 0.0
    assert L.ora_u23(0xFFFFFFFF) == np.float32(1.0 - 2.0 ** -23)


# ---------------------------------------------------------------- CartPole ----
def test_cartpole_against_python(golden):
    # cartpole.rs:449-452, default tolerances testing.rs:42-45
    v = ora.OracleVec(ora.CARTPOLE, 1, seed=1)
    worst_obs, worst_rew = replay(VecAdapter(v, "cartpole"), golden("cartpole"))
    assert worst_obs < 2e-7 and worst_rew == 0.0  # f32 restatement of gymnasium's f64 math: <= 1 ulp


def test_cartpole_first_fixture_row_exact():
    v = ora.OracleVec(ora.CARTPOLE, 1)
    VecAdapter(v, "cartpole").reset_deterministic()
    obs, rew, done, trunc = v.step([0])
    assert np.allclose(obs[:, 0], [0.0, -0.19512194, 0.0, 0.29268292], rtol=0, atol=2e-8)  # f32 vs gymnasium f64: 1 ulp
    assert rew[0] == 1.0 and not done[0] and not trunc[0]


def test_cartpole_unit_kats():
    # cartpole.rs:365-390: shapes, reward 1, not done on the first step from a reset state
    v = ora.OracleVec(ora.CARTPOLE, 64, seed=7)
    obs = v.reset()
    assert obs.shape == (4, 64) and (np.abs(obs) <= 0.05).all()
    obs, rew, done, trunc = v.step(np.zeros(64, np.uint32))
    assert (rew == 1.0).all() and not done.any()
    # cartpole.rs:405-434: constant action 1 -> done within 51 steps
    v = ora.OracleVec(ora.CARTPOLE, 64, seed=8)
    v.reset()
    finished = np.zeros(64, bool)
    for _ in range(51):
        _, _, done, _ = v.step(np.ones(64, np.uint32))
        finished |= done.astype(bool)
    assert finished.all()


def test_cartpole_invalid_action():
    # cartpole.rs:392-403 (#[should_panic]); Discrete(2)
    v = ora.OracleVec(ora.CARTPOLE, 1)
    v.reset()
    with pytest.raises(ValueError):
        v.step([2])


def test_cartpole_truncation_beats_termination_and_post_terminal():
    # cartpole.rs:296-306: at steps_since_reset >= 500 -> (reward 1, done F, truncated T) even if terminated
    v = ora.OracleVec(ora.CARTPOLE, 2)
    v.reset()
    s = v.get_state()
    s[0:4, :] = 0.0
    s[0, 1] = 3.0  # env 1 is beyond x_threshold: would terminate
    s[4, :] = u32col([499, 499])
    v.set_state(s)
    _, rew, done, trunc = v.step([0, 1])
    assert rew.tolist() == [1.0, 1.0] and done.tolist() == [0, 0] and trunc.tolist() == [1, 1]
    s = v.get_state()
    assert s[5].view(np.int32).tolist() == [0, 0]  # sbt = Some(0) (:299)
    # cartpole.rs:319-346: first termination reward 1/done, then reward 0/done forever (no auto-reset)
    v = ora.OracleVec(ora.CARTPOLE, 1)
    v.reset()
    s = v.get_state()
    s[0:4, 0] = [2.39, 1.0, 0.0, 0.0]
    v.set_state(s)
    _, rew, done, _ = v.step([1])
    assert rew[0] == 1.0 and done[0] == 1
    for k in range(3):
        _, rew, done, _ = v.step([1])
        assert rew[0] == 0.0 and done[0] == 1
        assert v.get_state()[5].view(np.int32)[0] == min(k + 1, 2)  # the blob's sbt column saturates at Some(2) (only None/Some is observable)


def test_cartpole_sutton_barto_and_non_euler():
    v = ora.OracleVec(ora.CARTPOLE, 1, sutton_barto_reward=True)
    v.reset()
    _, rew, done, _ = v.step([0])
    assert rew[0] == 0.0 and not done[0]  # :311
    s = v.get_state()
    s[0:4, 0] = [2.39, 1.0, 0.0, 0.0]
    v.set_state(s)
    _, rew, done, _ = v.step([1])
    assert rew[0] == -1.0 and done[0]  # :322
    _, rew, done, _ = v.step([1])
    assert rew[0] == -1.0 and done[0]  # :337
    # non-Euler branch (:279-282) reproduced verbatim: x never advances
    v = ora.OracleVec(ora.CARTPOLE, 1, is_euler=False)
    v.reset()
    s = v.get_state()
    s[0:4, 0] = [0.1, 0.5, 0.02, -0.3]
    v.set_state(s)
    obs, _, _, _ = v.step([1])
    assert obs[0, 0] == np.float32(0.1)
    f = np.float32
    th, thd = f(0.02), f(-0.3)
    c, sn = np.cos(th, dtype=f), np.sin(th, dtype=f)
    temp = (f(10.0) + f(0.1) * f(0.5) * thd * thd * sn) / (f(0.1) + f(1.0))
    thacc = (f(9.8) * sn - c * temp) / (f(0.5) * (f(4.0) / f(3.0) - f(0.1) * c * c / (f(0.1) + f(1.0))))
    thd1 = thd + f(0.5) * f(0.02) * (thacc + temp)
    th1 = th + (f(0.02) * thd1 + f(0.5) * f(0.02) * f(0.02) * thacc)
    thd2 = thd1 + f(0.5) * f(0.02) * (thacc + temp)
    assert abs(obs[2, 0] - th1) < 1e-7 and abs(obs[3, 0] - thd2) < 1e-6


# -------------------------------------------------------------- MountainCar ----
def test_mountain_car_against_python(golden):
    v = ora.OracleVec(ora.MOUNTAINCAR, 1)
    worst_obs, worst_rew = replay(VecAdapter(v, "mountain_car"), golden("mountain_car"))
    assert worst_obs < 2e-7 and worst_rew == 0.0


def test_mountain_car_kats():
    v = ora.OracleVec(ora.MOUNTAINCAR, 1)
    VecAdapter(v, "mountain_car").reset_deterministic()
    obs, rew, done, trunc = v.step([1])
    assert obs[:, 0].tolist() == [np.float32(-0.0025), np.float32(-0.0025)]  # first fixture row
    assert rew[0] == -1.0 and not done[0] and not trunc[0]
    v = ora.OracleVec(ora.MOUNTAINCAR, 32, seed=3)
    obs = v.reset()
    assert ((obs[0] >= -0.6) & (obs[0] < -0.4)).all() and (obs[1] == 0).all()  # mountain_car.rs:279-291
    with pytest.raises(ValueError):
        v.step(np.full(32, 3, np.uint32))  # mountain_car.rs:374-385


def test_mountain_car_clamps_wall_goal():
    v = ora.OracleVec(ora.MOUNTAINCAR, 4)
    s = v.get_state()
    #            speed clamp   left wall     goal        goal but v<goal_velocity handled below
    s[0, :] = [-1.0, -1.19, 0.495, -0.5]
    s[1, :] = [0.0699, -0.05, 0.02, -0.0699]
    v.set_state(s)
    obs, rew, done, trunc = v.step([2, 0, 2, 0])
    assert obs[1, 0] == np.float32(0.07)                       # :304
    assert obs[0, 1] == np.float32(-1.2) and obs[1, 1] == 0.0  # :308-313
    assert done.tolist() == [0, 0, 1, 0] and (rew == -1.0).all() and not trunc.any()
    assert obs[1, 3] == np.float32(-0.07)
    v = ora.OracleVec(ora.MOUNTAINCAR, 1, goal_velocity=0.05)
    s = v.get_state()
    s[0, 0], s[1, 0] = 0.495, 0.02
    v.set_state(s)
    _, _, done, _ = v.step([2])
    assert not done[0]                                         # :318 velocity >= goal_velocity fails


def test_mountain_car_continuous_unpinned_semantics():
    # NOT in the reference (parity unpinned): gymnasium semantics
    v = ora.OracleVec(ora.MOUNTAINCAR_CONT, 3)
    s = v.get_state()
    s[0, :] = [0.0, 0.449, -1.19]
    s[1, :] = [0.0, 0.03, -0.06]
    v.set_state(s)
    obs, rew, done, trunc = v.step(np.array([2.0, 1.0, -1.0], np.float32))
    f = np.float32
    assert obs[1, 0] == f(0.0) + (f(1.0) * f(0.0015) + np.cos(f(0.0), dtype=f) * f(-0.0025))  # clamped force
    assert rew[0] == f(0.0) - f(2.0) * f(2.0) * f(0.1)   # penalty on the un-clamped action
    assert done.tolist() == [0, 1, 0] and rew[1] == f(100.0) - f(0.1)
    assert obs[0, 2] == f(-1.2) and obs[1, 2] == 0.0


def test_vec_run_equals_step_plus_masked_reset():
    """ora_vec_run (the C-side K-step loop bench.py times for BASELINE configs[0]) is exactly step + reset-on-finish."""
    import numpy as np
    from oracle import oracle as ora
    for kind, nact in ((ora.CARTPOLE, 2), (ora.MOUNTAINCAR, 3)):
        a, b = ora.OracleVec(kind, 48, seed=5), ora.OracleVec(kind, 48, seed=5)
        a.reset(), b.reset()
        acts = np.random.default_rng(1).integers(0, nact, (16, 48)).astype(np.uint32)
        finished = a.run(acts, 300)
        count = 0
        for t in range(300):
            _, _, d, tr = b.step(acts[t % 16])
            m = d | tr
            count += int(m.sum())
            if m.any():
                b.reset(mask=m)
        assert finished == count
        assert np.array_equal(a.get_state().view(np.uint32), b.get_state().view(np.uint32))


def test_cartpole_fast_math_is_bit_identical_exhaustively():
    """The cheaper instruction sequences of the CartPole kernels (modurl_gym_amd/csrc/cartpole_math.h, cartpole_step.h)
    against the reference-form arithmetic of cartpole.rs:264-271: every f32 input of the fused sin/cos (|y| < 0.75) and of
    x / total_mass, every reachable divisor of the theta-acceleration quotient under any 1-ulp reciprocal estimate, and the
    whole fast-form step on 7e7 random guard-admitted states.  ~10 s on 8 cores; no GPU."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    build = os.path.join(root, "tests", "native", "_build")
    os.makedirs(build, exist_ok=True)
    exe = os.path.join(build, "cartpole_fast_check")
    subprocess.run(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-fno-fast-math", "-mfma", "-pthread", "-o", exe,
                    os.path.join(root, "tests", "native", "cartpole_fast_check.cpp"), "-lm"], check=True)
    r = subprocess.run([exe, str(min(8, os.cpu_count() or 1)), "1"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    rows = {l.split()[0]: dict(kv.split("=") for kv in l.split()[1:]) for l in r.stdout.strip().splitlines()}
    assert int(rows["sincos_small"]["checked"]) == 2 * 0x3f400000 and int(rows["div_const"]["checked"]) > 3_000_000_000
    assert int(rows["div"]["checked"]) > 400_000_000 and int(rows["step"]["checked"]) > 50_000_000
    assert int(rows["sincos_u"]["checked"]) == 2 * 0x42f00000   # every f32 below 120, both signs
    for k in ("sincos_u", "sincos_small", "div_const", "div", "step"):
        assert int(rows[k]["mismatches"]) == 0, r.stdout
