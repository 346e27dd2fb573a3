#!/bin/bash
# Runs ON the GPU box: LunarLander ms/step by population for the two contact-block layouts (32 lanes: single-launch step; 64 lanes:
# multi-stream order) with staged resets at their default and forced on.  usage: tools/ll_population_matrix.sh "<populations>"
for n in ${1:-262144 393216 524288 1048576}; do
  echo "== $n envs"
  LL_TUNE_ENVS=$n bash tools/ll_env_ab.sh 1 "MGYM_LL_GENERAL_BLOCK=32" "MGYM_LL_GENERAL_BLOCK=64" "MGYM_LL_GENERAL_BLOCK=32 MGYM_LL_STAGED_RESET=2" "MGYM_LL_GENERAL_BLOCK=64 MGYM_LL_STAGED_RESET=2" "MGYM_LL_GENERAL_BLOCK=32 MGYM_LL_STAGED_RESET=0" "MGYM_LL_GENERAL_BLOCK=64 MGYM_LL_STAGED_RESET=0"
done
