#!/bin/bash
# (*GPU box*) rollout tuning values with the helper waves on (a `make DIAG=1` build as modurl_gym_amd/libmgym_diag.so is swapped in; restored by a trap)
L=modurl_gym_amd
cp $L/libmgym.so /tmp/libmgym_cur.so
trap 'cp /tmp/libmgym_cur.so $L/libmgym.so' EXIT
cp $L/libmgym_diag.so $L/libmgym.so
O=gpurun_out/roll_sweep2.txt; : > $O
t() { echo -n "K=$K $* : " >> $O; env "$@" timeout -k 10 300 python tools/ll_roll_check.py time 262144 $K 6 2>&1 | grep "^n=" | sed 's/.*mgym_rollout//' >> $O; }
for K in ${KS:-8 16 64}; do export K
  t X=0 || exit 1
  t MGYM_LL_ROLL_KEEP=3 || exit 1
  t MGYM_LL_ROLL_KEEP=2 || exit 1
  t MGYM_LL_ROLL_KEEP=3 MGYM_LL_ROLL_RESIDENCY=3 || exit 1
  t MGYM_LL_ROLL_KEEP=3 MGYM_LL_ROLL_KEEP_MIN=16 || exit 1
  t MGYM_LL_ROLL_KEEP=3 MGYM_LL_ROLL_KEEP_MIN=28 || exit 1
  t MGYM_LL_ROLL_KEEP=3 MGYM_LL_ROLL_CONTACT_MIN=24 || exit 1
  t MGYM_LL_ROLL_KEEP=3 MGYM_LL_ROLL_TOI_MIN=32 || exit 1
done
echo "sweep rc=$?"
