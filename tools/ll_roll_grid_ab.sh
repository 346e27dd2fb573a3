#!/bin/bash
# (*GPU box*) rollout launch: how many main waves per CU, with and without helper waves (a `make DIAG=1` build as modurl_gym_amd/libmgym_diag.so is swapped in;
# the product library is restored by a trap).  ms per step-equivalent at 262 144 envs.
L=modurl_gym_amd
cp $L/libmgym.so /tmp/libmgym_cur.so
trap 'cp /tmp/libmgym_cur.so $L/libmgym.so' EXIT
cp $L/libmgym_diag.so $L/libmgym.so
O=gpurun_out/roll_grid_ab.txt; : > $O
t() { echo "== $*" >> $O; env "$@" MGYM_LL_ROLL_STATS=1 timeout -k 10 300 python tools/ll_roll_check.py time 262144 $K 6 2>&1 | grep -E "^n=|helper" | tail -2 >> $O; }
for K in 64 16; do export K
  t MGYM_LL_ROLL_HELPER=0 || exit 1
  t MGYM_LL_ROLL_HELPER=0 MGYM_LL_ROLL_GRID=768 || exit 1
  t MGYM_LL_ROLL_HELPER=0 MGYM_LL_ROLL_GRID=896 || exit 1
  t MGYM_LL_ROLL_HELPER=1 || exit 1
  t MGYM_LL_ROLL_HELPER=1 MGYM_LL_ROLL_HELPER_PER_CU=3 || exit 1
  t MGYM_LL_ROLL_HELPER=1 MGYM_LL_ROLL_HELPER_MIN=1024 || exit 1
done
echo "roll_grid_ab rc=$?"
