#!/bin/bash
# Runs ON the GPU box: is the contact kernel's occupancy throttled by the per-queue scratch ring (ROCr caps it; the kernel
# needs ~2.8 KB/lane = 182 KB per wave)?  Same configs with and without a larger HSA_SCRATCH_SINGLE_LIMIT.
run() { python bench.py --workload lunar_lander --steps 64 --warmup 640 --no-cpu-baseline --launch eager 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.3f ms/step' % d['ms_per_step'])"; }
for blk in 32 16 8; do
  for lim in default 4000000000; do
    if [ $lim = default ]; then r=$(MGYM_LL_GENERAL_BLOCK=$blk run); else r=$(HSA_SCRATCH_SINGLE_LIMIT=$lim HSA_SCRATCH_SINGLE_LIMIT_ASYNC=$lim MGYM_LL_GENERAL_BLOCK=$blk run); fi
    echo "general_block=$blk scratch_limit=$lim : $r"
  done
done
