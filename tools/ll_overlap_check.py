"""Runs ON the GPU box: the overlapped launch order (MGYM_LL_OVERLAP=<mode>) against the sequential one, same seeds and
actions, every word of every output and of the exported state.  usage: python tools/ll_overlap_check.py [mode=2] [n=262144] [steps=150] [auto=1]"""
import os, sys
import numpy as np
sys.path.insert(0, '.')
import modurl_gym_amd as mg
mode = sys.argv[1] if len(sys.argv) > 1 else "2"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1 << 18
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 150
auto = bool(int(sys.argv[4])) if len(sys.argv) > 4 else True
os.environ["MGYM_LL_OVERLAP"] = mode
a_env = mg.VecEnv(mg.LUNARLANDER, n, seed=77, enable_wind=True, auto_reset=auto)
os.environ["MGYM_LL_OVERLAP"] = "0"
b_env = mg.VecEnv(mg.LUNARLANDER, n, seed=77, enable_wind=True, auto_reset=auto)
assert np.array_equal(a_env.reset(), b_env.reset())
rng = np.random.default_rng(8)
bad_steps = 0
for t in range(steps):
    a = rng.integers(0, 4, n).astype(np.uint32)
    got, exp = a_env.step(a), b_env.step(a)
    for g, e, name in zip(got, exp, ("obs", "reward", "done", "truncated")):
        gv = g.view(np.uint32) if g.dtype == np.float32 else g
        ev = e.view(np.uint32) if e.dtype == np.float32 else e
        if not np.array_equal(gv, ev):
            idx = np.argwhere(gv != ev)
            envs = np.unique(idx[:, -1])
            print(f"step {t}: {name} differs for {len(envs)} envs, first {envs[:8]}")
            if name == "obs":
                i = envs[0]
                print("   got", g[:, i], "\n   exp", e[:, i])
            bad_steps += 1
            break
    if bad_steps >= 3:
        break
sa, sb = a_env.get_state().view(np.uint32), b_env.get_state().view(np.uint32)
print("state equal:", np.array_equal(sa, sb), "| steps with differences:", bad_steps, "of", t + 1)
print("OVERLAP CHECK", "OK" if bad_steps == 0 and np.array_equal(sa, sb) else "FAILED")
