#!/bin/bash
# Runs ON the GPU box: duration of the LunarLander kernels as a function of the population (is the contact kernel's time
# set by its slowest wave — then it barely depends on n — or by cache footprint / throughput — then it scales with n?)
cd /tmp && export TMPDIR=/tmp
for n in 16384 65536 262144 1048576; do
  rm -rf /tmp/llsz_$n
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/llsz_$n -- python3 $GRAFT_REPO_ROOT/bench.py --workload lunar_lander --envs $n --steps 64 --warmup 640 --launch eager --no-cpu-baseline --no-extra > /tmp/llsz_$n.log 2>&1
  echo "n=$n: $(grep -h '^{"metric"' /tmp/llsz_$n.log | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.3f ms/step' % d['ms_per_step'])")"
  python3 - <<PY
import csv, glob
for f in glob.glob("/tmp/llsz_$n/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "mgym::ll_" in r["Name"] and int(r["Calls"]) > 10:
            print("   %-40s avg %9.1f us  max %9.1f us" % (r["Name"].split("(")[0].replace("void mgym::", ""), float(r["AverageNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
PY
done
