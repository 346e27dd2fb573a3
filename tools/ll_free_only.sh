#!/bin/bash
for rep in 1 2; do
for cfg in "-" "MGYM_LL_SINGLE_LAUNCH=0" "MGYM_LL_SINGLE_LAUNCH=0 MGYM_LL_FREE_OCC=1" "MGYM_LL_SINGLE_LAUNCH=0 MGYM_LL_FREE_OCC=3"; do
  [ "$cfg" = "-" ] && envs="" || envs="$cfg"
  v=$(env $envs python bench.py --workload lunar_lander --steps 32 --warmup 2 --no-cpu-baseline --no-extra --launch eager 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.3f ms/step' % d['ms_per_step'])")
  echo "[$cfg] first 32 steps after reset (all envs in free flight): $v"
done; done
