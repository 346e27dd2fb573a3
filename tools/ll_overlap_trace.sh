#!/bin/bash
# Runs ON the GPU box: kernel trace of a few LunarLander steps in the overlapped launch order; prints the timeline of
# one step (start / end of every kernel relative to the step's first kernel) — do the two streams really run side by side?
cd /tmp && export TMPDIR=/tmp
# usage: [ENVS=1048576] [MGYM_LL_BUCKET=..] tools/ll_overlap_trace.sh   (the engine's knobs pass through the environment)
ENVS=${ENVS:-262144}
rm -rf /tmp/llov
rocprofv3 --kernel-trace --output-format csv -d /tmp/llov -- python3 $GRAFT_REPO_ROOT/bench.py --workload lunar_lander --envs $ENVS --steps 8 --warmup 400 --launch eager --no-cpu-baseline --no-extra > /tmp/llov.log 2>&1
python3 - <<'PY'
import csv, glob
rows = []
for f in glob.glob("/tmp/llov/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void mgym::", ""), r.get("Queue_Id", "?")))
rows.sort()
ll = [r for r in rows if "ll_" in r[2] or "fillBuffer" in r[2]]
# the last two steps
tail = ll[-26:]
t0 = tail[0][0]
for r in tail:
    print("%9.1f us -> %9.1f us  (%8.1f us)  queue %s  %s" % ((r[0] - t0) / 1e3, (r[1] - t0) / 1e3, (r[1] - r[0]) / 1e3, r[3], r[2]))
PY
