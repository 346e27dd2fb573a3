#!/bin/bash
# (*GPU box*) mgym_rollout with / without the free-flight helper waves: parity first, then ms per step-equivalent at 262 144 envs
O=gpurun_out/roll_helper_ab.txt; : > $O
MGYM_LL_ROLL_HELPER=1 timeout -k 10 600 python -m pytest tests/test_gpu_lunar_rollout.py -x -q > gpurun_out/roll_helper_tests.log 2>&1 || { tail -5 gpurun_out/roll_helper_tests.log; exit 1; }
tail -1 gpurun_out/roll_helper_tests.log >> $O
t() { echo "== $*" >> $O; env "$@" MGYM_LL_ROLL_STATS=1 timeout -k 10 300 python tools/ll_roll_check.py time ${N:-262144} $K 6 2>&1 | grep -E "^n=|helper" | tail -2 >> $O; }
for K in 64 16 8; do
  export K
  t MGYM_LL_ROLL_HELPER=0 || exit 1
  t MGYM_LL_ROLL_HELPER=1 || exit 1
  t MGYM_LL_ROLL_HELPER=1 MGYM_LL_ROLL_MAIN_PER_CU=2 MGYM_LL_ROLL_HELPER_PER_CU=4 || exit 1
  t MGYM_LL_ROLL_HELPER=1 MGYM_LL_ROLL_MAIN_PER_CU=4 MGYM_LL_ROLL_HELPER_PER_CU=2 || exit 1
done
echo "roll_helper_ab rc=$?"
