"""Long LunarLander soak: GPU (through the C ABI) vs the CPU oracle over thousands of steps with resets.
Usage: python tools/ll_soak.py [n] [steps] [wind] [auto]   (run on the GPU box; not part of the test-suite)
auto = 1: the engine runs with MGYM_FLAG_AUTO_RESET (finished envs are reset inside mgym_step by the compacted
reset kernel) and the oracle does step + masked reset; observations are compared after the reset."""
import sys, time
import numpy as np
sys.path.insert(0, '.')
import modurl_gym_amd as mg
from oracle import oracle as ora
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3000
wind = bool(int(sys.argv[3])) if len(sys.argv) > 3 else True
auto = bool(int(sys.argv[4])) if len(sys.argv) > 4 else False
env = mg.VecEnv(mg.LUNARLANDER, n, seed=99, enable_wind=wind, auto_reset=auto)
ref = ora.OracleVec(ora.LUNARLANDER, n, seed=99, enable_wind=wind)
assert np.array_equal(env.reset(), ref.reset(nthreads=16))
rng = np.random.default_rng(1)
words = exact = 0; episodes = landed = 0; worst = 0.0; t0 = time.time()
for t in range(steps):
    # a policy with some skill so that landings (sleep, +100) occur as well as crashes
    s = ref.get_state()
    a = rng.integers(0, 4, n).astype(np.uint32)
    skilled = (np.arange(n) % 4 == 0)
    vy, ang, w = s[4], s[2], s[5]
    ctrl = np.where((np.abs(ang) > 0.05) | (np.abs(w) > 0.3), np.where(ang + 0.5 * w > 0, 3, 1), np.where(vy < -0.6, 2, 0)).astype(np.uint32)
    a[skilled] = ctrl[skilled]
    got, exp = env.step(a), ref.step(a, nthreads=16)
    if auto:  # the engine's observation is the one after the fused reset
        step_obs = exp[0].copy()
        ro = ref.reset(exp[2], nthreads=16)
        exp = (np.where(exp[2].astype(bool)[None, :], ro, step_obs), exp[1], exp[2], exp[3])
    for g, e in zip(got[:2], exp[:2]):
        d = np.abs(g - e); tol = 1e-5 * np.maximum(np.abs(g), np.abs(e)) + 1e-6
        if (d > tol).any():
            i = np.argwhere(d > tol)[0]
            print("MISMATCH at step", t, "index", i, g[tuple(i)], e[tuple(i)]); sys.exit(1)
        worst = max(worst, float(d.max()))
    if not np.array_equal(got[2], exp[2]):
        print("DONE MISMATCH at step", t, np.argwhere(got[2] != exp[2])[:5].ravel()); sys.exit(1)
    words += got[0].size; exact += int((got[0] == exp[0]).sum())
    done = exp[2].astype(bool); episodes += int(done.sum()); landed += int((exp[1][done] == 100.0).sum())
    if not auto and t % 3 == 0 and done.any():
        env.reset(exp[2]); ref.reset(exp[2], nthreads=16)
    if t % 500 == 0: print(f"step {t}: episodes {episodes} landed {landed} exact {exact}/{words} worst abs diff {worst:.3g} ({time.time()-t0:.0f}s)", flush=True)
print(f"SOAK OK: {n} envs x {steps} steps, episodes {episodes}, landed asleep {landed}, bit-identical obs words {exact}/{words} ({100.0*exact/words:.4f}%), worst abs diff {worst:.3g}")
