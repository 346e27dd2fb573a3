#!/bin/bash
# Runs ON the GPU box: instruction-fetch counters of the LunarLander kernels (is the 136 KB contact kernel starved by the
# 64 KB instruction cache?).  Usage: tools/profile_ifetch.sh <tag> [env assignments...]
TAG=$1; shift
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/ifetch_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for kv in "$@"; do export $kv; done
rocprofv3 --pmc SQ_IFETCH SQ_IFETCH_LEVEL SQ_INSTS SQ_WAVE_CYCLES SQ_WAVES SQ_WAIT_ANY SQ_INST_LEVEL_VMEM SQ_BUSY_CYCLES --output-format csv -d $OUT/pmc -- python3 $REPO/bench.py --workload lunar_lander --steps 32 --warmup 640 --launch eager --no-cpu-baseline --no-extra > $OUT/log.txt 2>&1
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/pmc/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, c in acc.items():
    if "ll_" not in k: continue
    m = {n: sum(v[len(v)*3//4:]) / max(1, len(v) - len(v)*3//4) for n, v in c.items()}
    print(k[:70])
    for n, v in sorted(m.items()): print("   %-22s %14.1f" % (n, v))
    if m.get("SQ_IFETCH"): print("   -> mean fetch latency (IFETCH_LEVEL / IFETCH) = %.1f ; insts per fetch = %.2f ; fetches per wave = %.0f" % (m["SQ_IFETCH_LEVEL"] / m["SQ_IFETCH"], m["SQ_INSTS"] / m["SQ_IFETCH"], m["SQ_IFETCH"] / m["SQ_WAVES"]))
PY
