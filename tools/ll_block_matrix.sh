#!/bin/bash
# Runs ON the GPU box: LunarLander ms/step by population x lanes per contact block of the single-launch step kernel (MGYM_LL_GENERAL_BLOCK).
# usage: tools/ll_block_matrix.sh "<populations>" "<blocks>"
for n in ${1:-65536 131072 196608 262144}; do
  echo "== $n envs"
  cfgs=(); for b in ${2:-8 12 16 20 24 28 32}; do cfgs+=("MGYM_LL_GENERAL_BLOCK=$b"); done
  LL_TUNE_ENVS=$n bash tools/ll_env_ab.sh 1 "${cfgs[@]}"
done
