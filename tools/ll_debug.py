import sys, numpy as np
sys.path.insert(0, '.')
import modurl_gym_amd as mg
from oracle import oracle as ora
n=2048
env = mg.VecEnv(mg.LUNARLANDER, n, seed=77); ref = ora.OracleVec(ora.LUNARLANDER, n, seed=77)
env.reset(); ref.reset(nthreads=8)
rng = np.random.default_rng(3)
for t in range(260):
    a = rng.integers(0, 4, n).astype(np.uint32)
    sg0 = env.get_state(); so0 = ref.get_state()
    got, exp = env.step(a), ref.step(a, nthreads=8)
    bad = np.argwhere(got[0] != exp[0])
    if len(bad):
        print("step", t, "mismatching words", len(bad), "first", bad[:6].tolist())
        i = bad[0][1]
        print("env", i, "action", a[i], "gpu obs", got[0][:, i], "\n            oracle", exp[0][:, i])
        print("reward", got[1][i], exp[1][i], "done", got[2][i], exp[2][i])
        print("pre-state equal:", np.array_equal(sg0[:25, i].view(np.uint32), so0[:25, i].view(np.uint32)))
        print("pre-state gpu", sg0[:, i]); print("pre-state ora", so0[:, i])
        if not np.all(np.abs(got[0]-exp[0]) <= 1e-5*np.maximum(np.abs(got[0]),np.abs(exp[0]))+1e-6): break
    mask = exp[2]
    if t % 2 == 0 and mask.any():
        env.reset(mask); ref.reset(mask, nthreads=8)
