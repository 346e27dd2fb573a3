#!/usr/bin/env python3
"""(*GPU box*) LunarLander at a large population: the default order for that size (64-lane contact blocks, multi-stream, contact list by kind from 688 128 envs)
against the 32-lane single-launch step forced onto the same population — every output word of every step and the exported state.
usage: python tools/ll_layout_check.py [n=1048576] [steps=200]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import modurl_gym_amd as mg  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
a_env = mg.VecEnv(mg.LUNARLANDER, n, seed=61, enable_wind=True, auto_reset=True)
os.environ["MGYM_LL_GENERAL_BLOCK"] = "32"
b_env = mg.VecEnv(mg.LUNARLANDER, n, seed=61, enable_wind=True, auto_reset=True)
del os.environ["MGYM_LL_GENERAL_BLOCK"]
ia, ib = a_env.info(), b_env.info()
print("default:", {k: ia[k] for k in ("launch_order", "contact_block", "contact_list_by_kind")}, " forced:", {k: ib[k] for k in ("launch_order", "contact_block", "contact_list_by_kind")})
assert np.array_equal(a_env.reset(), b_env.reset())
rng = np.random.default_rng(3)
finished = 0
for t in range(steps):
    a = rng.integers(0, 4, n).astype(np.uint32)
    got, exp = a_env.step(a), b_env.step(a)
    for g, e, nm in zip(got, exp, ("obs", "reward", "done", "truncated")):
        gv = g.view(np.uint32) if g.dtype == np.float32 else g
        ev = e.view(np.uint32) if e.dtype == np.float32 else e
        if not np.array_equal(gv, ev):
            print(f"MISMATCH step {t}: {nm} differs for {int((gv != ev).sum())} words")
            sys.exit(1)
    finished += int(exp[2].sum())
same = np.array_equal(a_env.get_state().view(np.uint32), b_env.get_state().view(np.uint32))
print(f"OK n={n} steps={steps}: every output word equal; {finished} episodes finished; state blob equal: {same}")
sys.exit(0 if same else 1)
