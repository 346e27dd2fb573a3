#!/bin/bash
# (*GPU box*) where the 64-lane multi-stream order overtakes the 32-lane single-launch step, and where the contact list by kind starts to pay
O=gpurun_out/block_threshold.txt; : > $O
run() { python bench.py --workload lunar_lander --envs $1 --steps 64 --warmup 640 --no-extra --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.4f ms/step %.4g env-steps/s' % (d['ms_per_step'], d['value']))" >> $O; }
for N in 262144 294912 327680 360448 393216 425984; do
  for B in 32 64; do echo "== $N envs MGYM_LL_GENERAL_BLOCK=$B" >> $O; MGYM_LL_GENERAL_BLOCK=$B run $N || exit 1; done
done
for N in 589824 655360 720896; do
  for B in 0 1; do echo "== $N envs MGYM_LL_BUCKET=$B" >> $O; MGYM_LL_BUCKET=$B run $N || exit 1; done
done
echo "block_threshold rc=$?"
