#!/bin/bash
# Runs ON the GPU box: resident waves per CU of the LunarLander contact kernel for 32- and 16-lane blocks
# (SQ_LEVEL_WAVES / SQ_BUSY_CU_CYCLES).  Do narrower blocks actually co-reside two per SIMD?
cd /tmp && export TMPDIR=/tmp
for blk in 32 16; do
  rm -rf /tmp/llocc_$blk
  MGYM_LL_GENERAL_BLOCK=$blk rocprofv3 --pmc SQ_LEVEL_WAVES SQ_BUSY_CU_CYCLES SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_CYCLES GRBM_GUI_ACTIVE --output-format csv -d /tmp/llocc_$blk -- python3 $GRAFT_REPO_ROOT/bench.py --workload lunar_lander --steps 32 --warmup 640 --launch eager --no-cpu-baseline --no-extra > /tmp/llocc_$blk.log 2>&1
  python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("/tmp/llocc_$blk/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "ll_contact" in r["Kernel_Name"] or "ll_free" in r["Kernel_Name"]:
            acc[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, c in acc.items():
    m = {n: sum(v[len(v)*3//4:]) / max(1, len(v) - len(v)*3//4) for n, v in c.items()}
    print("block=$blk", k, {n: round(v) for n, v in m.items()})
    if m.get("SQ_BUSY_CU_CYCLES"): print("   SQ_LEVEL_WAVES / SQ_BUSY_CU_CYCLES = %.2f ; SQ_WAVE_CYCLES / SQ_BUSY_CYCLES = %.2f" % (m["SQ_LEVEL_WAVES"] / m["SQ_BUSY_CU_CYCLES"], m["SQ_WAVE_CYCLES"] / m["SQ_BUSY_CYCLES"]))
PY
done
