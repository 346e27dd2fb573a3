#!/bin/bash
# Runs ON the GPU box: counted HBM bytes per LunarLander step (FETCH_SIZE x2 + WRITE_SIZE of the step kernel, separate --pmc passes) for
# several library builds.  usage: tools/ll_traffic_ab.sh "<lib suffixes>"   ("cur" = libmgym.so)
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
L=$REPO/modurl_gym_amd
cp $L/libmgym.so /tmp/libmgym_cur.so
trap 'cp /tmp/libmgym_cur.so $L/libmgym.so' EXIT   # the product library comes back whatever happens (an interrupted run must not leave a variant build in its place)
cd /tmp && export TMPDIR=/tmp
for s in $1; do
  [ "$s" = "cur" ] && cp /tmp/libmgym_cur.so $L/libmgym.so || cp $L/libmgym_$s.so $L/libmgym.so
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/lltr_$c
    rocprofv3 --pmc $c --output-format csv -d /tmp/lltr_$c -- python3 $REPO/bench.py --workload lunar_lander --steps 32 --warmup 640 --no-cpu-baseline --no-extra > /tmp/lltr.log 2>&1
  done
  python3 - "$s" <<'PY'
import csv, glob, sys, collections
tot = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    v = []
    for f in glob.glob(f"/tmp/lltr_{c}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == c and "ll_step_kernel" in r["Kernel_Name"]: v.append(float(r["Counter_Value"]))
    tot[c] = sum(v[len(v) * 3 // 4:]) / max(1, len(v) - len(v) * 3 // 4)
print("lib %-6s ll_step_kernel per launch: FETCH_SIZE x2 = %.1f MB, WRITE_SIZE = %.1f MB, sum %.1f MB" % (sys.argv[1], 2 * tot["FETCH_SIZE"] * 1024 / 1e6, tot["WRITE_SIZE"] * 1024 / 1e6, (2 * tot["FETCH_SIZE"] + tot["WRITE_SIZE"]) * 1024 / 1e6))
PY
done
cp /tmp/libmgym_cur.so $L/libmgym.so
