"""Golden-file replay with teacher forcing — the reference's test protocol.

Restates /root/reference/src/testing.rs:65-134 so that the same check can be run
against (a) the CPU oracle and (b) the HIP engine through the C ABI:
  * before step i: if i == 0 or expected[i-1].done -> reset_deterministic();
    else set_state(expected[i-1].observation, expected[i-1].info)          (:73-87)
  * |reward - exp| <= reward_tol (:99-104); done ==, truncated == (:106-120);
    |obs_j - exp_j| < obs_tol (:123-133).
Default tolerances 1e-4 / 1e-4 (:42-45); LunarLander 5.0 / 0.2 (lunar_lander.rs:1649-1654).

An adapter exposes: reset_deterministic(), set_state(obs, info), step(action) ->
(obs[list], reward, done, truncated).
"""
import numpy as np


def replay(adapter, fixture, reward_tol=1e-4, obs_tol=1e-4, trace=None):
    """trace: optional list receiving (step, action, worst |obs error|, |reward error|) per step"""
    actions, expected = fixture["actions"], fixture["expected"]
    assert len(actions) > 0 and len(actions) == len(expected)
    adapter.reset_deterministic()
    worst_obs, worst_rew = 0.0, 0.0
    for i, action in enumerate(actions):
        if i == 0 or expected[i - 1]["done"]:
            adapter.reset_deterministic()
        else:
            adapter.set_state(expected[i - 1]["observation"], expected[i - 1].get("info"))
        obs, reward, done, trunc = adapter.step(action)
        exp = expected[i]
        # the reference compares in f32 (ExpectedOutput fields are f32, testing.rs:7-12)
        exp_rew = np.float32(exp["reward"])
        exp_obs = np.asarray(exp["observation"], np.float32)
        drew = abs(float(np.float32(reward) - exp_rew))
        assert drew <= reward_tol, f"step {i}: reward {reward} vs {exp['reward']} (obs {list(obs)} vs {exp['observation']})"
        assert bool(done) == exp["done"], f"step {i}: done {done} vs {exp['done']}"
        assert bool(trunc) == exp["truncated"], f"step {i}: truncated {trunc} vs {exp['truncated']}"
        dobs = np.abs(np.asarray(obs, np.float32) - exp_obs)
        assert (dobs < obs_tol).all(), f"step {i}: obs {list(obs)} vs {exp['observation']}"
        worst_obs = max(worst_obs, float(dobs.max()))
        if trace is not None:
            trace.append((i, action, float(dobs.max()), drew))
        worst_rew = max(worst_rew, drew)
    return worst_obs, worst_rew


class VecAdapter:
    """Drives a 1-env batched engine (oracle OracleVec or HIP VecEnv: same surface)
    through the Testable seam of the reference (testing.rs:15-18)."""

    def __init__(self, vec, kind):
        self.v, self.kind = vec, kind

    def reset_deterministic(self):
        if self.kind == "cartpole":
            # cartpole.rs:437-442: self.reset()? then state = zeros
            self.v.reset()
            s = self.v.get_state()
            s[0:4, :] = 0.0
            self.v.set_state(s)
        elif self.kind == "mountain_car":
            # mountain_car.rs:403-408: state = zeros, no reset()
            s = self.v.get_state()
            s[0:2, :] = 0.0
            self.v.set_state(s)
        elif self.kind == "lunar_lander":
            self.v.reset_deterministic()  # lunar_lander.rs:1249-1442
        else:
            raise NotImplementedError

    def set_state(self, obs, info):
        s = self.v.get_state()
        if self.kind == "lunar_lander":
            # lunar_lander.rs:1444-1554: raw physics values from `info`, leg contact flags; the blob's
            # words 0..19 carry exactly that (set_state ignores the rest)
            keys = [f"raw_{b}_{k}" for b in ("lander", "leg0", "leg1")
                    for k in ("pos_x", "pos_y", "angle", "vel_x", "vel_y", "angular_vel")]
            s[0:18, 0] = np.asarray([info[k] for k in keys], np.float32)
            s[18, 0] = 1.0 if info.get("leg0_contact", 0.0) > 0.5 else 0.0
            s[19, 0] = 1.0 if info.get("leg1_contact", 0.0) > 0.5 else 0.0
        else:
            s[0:len(obs), 0] = np.asarray(obs, np.float32)  # cartpole.rs:444-446 / mountain_car.rs:410-412
        self.v.set_state(s)

    def step(self, action):
        dt = np.float32 if self.kind == "mountain_car_cont" else np.uint32
        obs, rew, done, trunc = self.v.step(np.array([action], dt))
        return obs[:, 0].tolist(), float(rew[0]), bool(done[0]), bool(trunc[0])
