#!/bin/bash
# Runs ON the GPU box: LunarLander ms/step for several environment-variable settings of ONE library build, interleaved
# repetitions.  usage: tools/ll_env_ab.sh <reps> "VAR=a VAR2=b" "VAR=c" ...   ("-" = no setting); LL_TUNE_ENVS = population
reps=$1; shift
for r in $(seq $reps); do
  for cfg in "$@"; do
    [ "$cfg" = "-" ] && envs="" || envs="$cfg"
    v=$(env $envs timeout -k 10 120 python bench.py --workload lunar_lander ${LL_TUNE_ENVS:+--envs $LL_TUNE_ENVS} --steps 64 --warmup 640 --no-cpu-baseline --launch ${LL_LAUNCH:-eager} 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.3f ms/step %.3e steps/s' % (d['ms_per_step'], d['value']))")
    echo "[$cfg] : $v"
  done
done
