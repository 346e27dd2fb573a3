#!/bin/bash
# Runs ON the GPU box: wait / issue counters of ONE role of the step kernel (tools/_ll_role_time_<variant>, see ll_role_time.sh), one
# rocprofv3 --pmc pass per counter group.  usage: tools/ll_role_pmc.sh "<binary suffixes>" [envs]
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
G1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU"
G2="SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_FLAT SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_LDS SQ_INSTS_FLAT SQ_INSTS_FLAT_LDS_ONLY SQ_INSTS_SALU"
G3="SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_THREAD_CYCLES_VALU"
for s in $1; do
  for g in 1 2 3; do
    eval "C=\$G$g"
    rm -rf /tmp/rolepmc_$g
    rocprofv3 --pmc $C --output-format csv -d /tmp/rolepmc_$g -- $REPO/tools/_ll_role_time_$s ${2:-262144} > /tmp/rolepmc.log 2>&1 || tail -5 /tmp/rolepmc.log
  done
  python3 - "$s" <<'PY'
import csv, glob, sys, collections
tot = collections.defaultdict(list)
for f in glob.glob("/tmp/rolepmc_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "ll_step_kernel" in r["Kernel_Name"]: tot[r["Counter_Name"]].append(float(r["Counter_Value"]))
print("== variant", sys.argv[1], "(ll_step_kernel, mean per launch)")
m = {k: sum(v) / len(v) for k, v in tot.items()}
for k in sorted(m): print("   %-28s %16.1f" % (k, m[k]))
if "SQ_WAVE_CYCLES" in m:
    wc = m["SQ_WAVE_CYCLES"]
    for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_FLAT", "SQ_ACTIVE_INST_VMEM", "SQ_ACTIVE_INST_SCA", "SQ_WAIT_INST_LDS", "SQ_INST_CYCLES_VMEM"):
        if k in m: print("   -> %-26s / wave cycles = %.3f" % (k, m[k] / wc))
PY
done
