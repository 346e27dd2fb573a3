#!/bin/bash
# (*GPU box*) mgym_rollout with / without the free-flight helper waves by population and K: ms per step-equivalent (tools/ll_roll_check.py time)
O=gpurun_out/roll_helper_by_population.txt; : > $O
for N in ${SIZES:-32768 65536 98304 131072 196608}; do
  for K in ${KS:-8 16}; do
    for H in 0 1; do
      echo "== $N envs K=$K MGYM_LL_ROLL_HELPER=$H" >> $O
      MGYM_LL_ROLL_HELPER=$H timeout -k 10 300 python tools/ll_roll_check.py time $N $K 10 2>&1 | grep "^n=" >> $O || exit 1
    done
  done
done
echo "roll_helper_ab rc=$?"
