"""n_envs = 1 drop-in latency through the C ABI (upload action, launch, synchronise, download obs / reward / flags), the
shape of the Rust `Gym::step` shim: microseconds per step, against the CPU oracle's ns per step.  Run on the GPU box."""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
import modurl_gym_amd as mg
from oracle import oracle as ora

for name, cls, kind, nact in (("CartPoleV1", mg.CartPoleV1, ora.CARTPOLE, 2), ("MountainCarV0", mg.MountainCarV0, ora.MOUNTAINCAR, 3),
                              ("LunarLanderV3", mg.LunarLanderV3, ora.LUNARLANDER, 4)):
    env = cls()
    env.reset()
    rng = np.random.default_rng(0)
    acts = rng.integers(0, nact, 3000).astype(np.uint32)
    for a in acts[:200]:
        if env.step(a).done:
            env.reset()
    t0 = time.perf_counter()
    for a in acts[200:2200]:
        if env.step(a).done:
            env.reset()
    gpu_us = (time.perf_counter() - t0) / 2000 * 1e6
    ref = ora.OracleVec(kind, 1, seed=1)
    ref.reset()
    t0 = time.perf_counter()
    ref.run(acts[:16].reshape(16, 1), 200000)
    cpu_ns = (time.perf_counter() - t0) / 200000 * 1e9
    print(f"{name}: GPU engine n_envs=1 through the ABI (python ctypes host) {gpu_us:.1f} us/step; CPU oracle {cpu_ns:.0f} ns/step")
