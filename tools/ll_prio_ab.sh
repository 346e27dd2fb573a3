#!/bin/bash
# (*GPU box*) priority of the helper stream (free-flight kernel + late contact launch) against the caller's stream (main contact kernel), by population
O=gpurun_out/prio_ab.txt; : > $O
run() { python bench.py --workload lunar_lander --envs $1 --steps 64 --warmup 640 --no-extra --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.4f ms/step %.4g env-steps/s' % (d['ms_per_step'], d['value']))" >> $O; }
for N in ${SIZES:-524288 786432 1048576}; do
  echo "== $N envs default (lowest)" >> $O; run $N || exit 1
  for P in 0 -1; do echo "== $N envs MGYM_LL_AUX_PRIO=$P" >> $O; MGYM_LL_AUX_PRIO=$P run $N || exit 1; done
done
echo "prio_ab rc=$?"
