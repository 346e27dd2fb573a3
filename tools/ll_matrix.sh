#!/bin/bash
# Runs ON the GPU box: LunarLander ms/step over population x contact-kernel block x launch order (overlapped or sequential).
for n in "$@"; do
  for blk in 32 64; do
    for ov in 0 1; do
      echo -n "envs=$n overlap=$ov "
      LL_TUNE_ENVS=$n MGYM_LL_OVERLAP=$ov tools/ll_tune.sh "$blk 32 0 0"
    done
  done
done
