#!/bin/bash
# (*GPU box*; needs a library built with `make -C modurl_gym_amd/csrc DIAG=1`: the tuning values are fixed in the product) mgym_rollout timing for a list of "VAR=value ..." settings (one line each on stdin or as arguments), K and n from the environment
mkdir -p gpurun_out/r4e
export MGYM_LL_ROLLOUT=1 MGYM_LL_ROLL_STATS=1
for cfg in "$@"; do
  echo "== $cfg"
  env $cfg timeout -k 10 200 python tools/ll_roll_check.py time ${LL_N:-262144} ${LL_K:-8} 2 2>&1 | grep -v amdgpu.ids | tail -2 || exit 1
done
