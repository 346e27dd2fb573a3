#!/bin/bash
# Runs ON the GPU box: the contact kernel takes the first F sub-steps itself, one follow-up launch (B-lane blocks) finishes the
# few envs that need more — overlapped order, eager.  "F B" pairs; "-" = default (everything in the contact kernel).
echo -n "default: "; tools/ll_tune.sh "0 32 0 0"
for cfg in "$@"; do
  set -- $cfg
  echo -n "first=$1 follow-up block=$2: "; MGYM_LL_TOI_FIRST=$1 tools/ll_tune.sh "0 $2 1 0"
done
echo -n "default: "; tools/ll_tune.sh "0 32 0 0"
