#!/bin/bash
# Runs ON the GPU box: LunarLander ms/step by population x MGYM_LL_CONTACT_BLOCKS (blocks the contact list of the single-launch step is dealt out over;
# 0 = always 32 lanes per block, -1 = the engine's choice by population).  usage: tools/ll_target_matrix.sh "<populations>" "<targets>"
for n in ${1:-131072 262144}; do
  echo "== $n envs"
  cfgs=(); for b in ${2:-0 -1 700 800 900 1000}; do cfgs+=("MGYM_LL_CONTACT_BLOCKS=$b"); done
  LL_TUNE_ENVS=$n bash tools/ll_env_ab.sh 1 "${cfgs[@]}"
done
