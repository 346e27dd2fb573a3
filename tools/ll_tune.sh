#!/bin/bash
# Runs ON the GPU box: LunarLander step time for combinations of the launch knobs (contact-kernel block, TOI-kernel
# block, number of TOI rounds; rounds = 0: the contact kernel runs the whole world.step).  One line per combination.
for cfg in "$@"; do
  set -- $cfg
  r=$(MGYM_LL_BUCKET=${4:-1} MGYM_LL_GENERAL_BLOCK=$1 MGYM_LL_TOI_BLOCK=$2 MGYM_LL_TOI_ROUNDS=$3 timeout -k 10 120 python bench.py --workload lunar_lander ${LL_TUNE_ENVS:+--envs $LL_TUNE_ENVS} --steps 64 --warmup 640 --no-cpu-baseline --launch eager 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.3f ms/step %.3e steps/s' % (d['ms_per_step'], d['value']))")
  echo "general_block=$1 toi_block=$2 rounds=$3 bucket=${4:-1} : $r"
done
