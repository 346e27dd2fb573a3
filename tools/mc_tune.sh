#!/bin/bash
# Runs ON the GPU box: MountainCar step time, structured 16-byte kernel vs the round-1 kernel (MGYM_MC_OLD_KERNEL=1), interleaved.
pr() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%s us/step=%.3f frac=%.4f' % (sys.argv[1], d['ms_per_step']*1e3, d['roofline']['frac']))" "$1"; }
for rep in 1 2 3; do
  python bench.py --workload mountain_car --no-extra --no-cpu-baseline 2>/dev/null | pr "new rep=$rep"
  MGYM_MC_OLD_KERNEL=1 python bench.py --workload mountain_car --no-extra --no-cpu-baseline 2>/dev/null | pr "old rep=$rep"
done
python bench.py --workload mountain_car_cont --no-extra --no-cpu-baseline 2>/dev/null | pr "cont new"
MGYM_MC_OLD_KERNEL=1 python bench.py --workload mountain_car_cont --no-extra --no-cpu-baseline 2>/dev/null | pr "cont old"
