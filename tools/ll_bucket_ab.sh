#!/bin/bash
# (*GPU box*) the contact list by kind in the multi-stream order, by population: ms per step, same box.
#   MGYM_LL_BUCKET 0 | 1 (touching end first, in blocks of its own) | 1 + MGYM_LL_KIND_SPLIT=1 (the two ends as launches of their own: 32- / 64-lane blocks)
O=gpurun_out/bucket_ab.txt; : > $O
run() { python bench.py --workload lunar_lander --envs $1 --steps 64 --warmup 640 --no-extra --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.4f ms/step %.4g env-steps/s' % (d['ms_per_step'], d['value']))" >> $O; }
for N in ${SIZES:-524288 655360 786432 1048576 2097152}; do
  echo "== $N envs MGYM_LL_BUCKET=0" >> $O; MGYM_LL_BUCKET=0 run $N || exit 1
  echo "== $N envs MGYM_LL_BUCKET=1" >> $O; MGYM_LL_BUCKET=1 run $N || exit 1
  echo "== $N envs MGYM_LL_BUCKET=1 MGYM_LL_KIND_SPLIT=1" >> $O; MGYM_LL_BUCKET=1 MGYM_LL_KIND_SPLIT=1 run $N || exit 1
done
echo "bucket_ab rc=$?"
