#!/usr/bin/env python3
"""Timeline of one mgym_rollout launch from the wave trace (MGYM_LL_ROLL_TRACE=file): how many waves do what, per time bin."""
import struct
import sys

import numpy as np

NAMES = {1: "seed", 2: "touching", 3: "light", 4: "reset", 5: "free", 6: "idle", 7: "end", 8: "substeps"}
path, bin_us = sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 else 250.0
raw = open(path, "rb").read()
grid, tlen = struct.unpack("II", raw[:8])
ev = np.frombuffer(raw[8:], dtype=np.uint64).reshape(grid, tlen)
t_end = 0.0
rows = []
for w in range(grid):
    e = ev[w][ev[w] != 0]
    ts = (e >> np.uint64(8)).astype(np.float64) * 0.01   # us
    ks = (e & np.uint64(255)).astype(int)
    rows.append((ts, ks))
    if len(ts):
        t_end = max(t_end, ts[-1])
nb = int(t_end / bin_us) + 1
occ = np.zeros((9, nb))
for ts, ks in rows:
    for j in range(len(ts) - 1):
        a, b, k = ts[j], ts[j + 1], ks[j]
        ia, ib = int(a / bin_us), int(b / bin_us)
        for bi in range(ia, min(ib, nb - 1) + 1):
            lo, hi = max(a, bi * bin_us), min(b, (bi + 1) * bin_us)
            if hi > lo:
                occ[k, bi] += (hi - lo) / bin_us
    if len(ts):   # after its last event a wave has exited
        pass
print(f"{grid} waves, launch {t_end:.0f} us; waves per activity in bins of {bin_us:.0f} us (rest: exited)")
print("   t_us " + " ".join(f"{NAMES[k]:>8s}" for k in (1, 2, 3, 8, 4, 5, 6)))
for bi in range(nb):
    print(f"{bi * bin_us:7.0f} " + " ".join(f"{occ[k, bi]:8.0f}" for k in (1, 2, 3, 8, 4, 5, 6)))
