#!/bin/bash
# Runs ON the GPU box: LunarLander step time by launch mode (hipGraph replay / eager) x staged resets x overlap.
# (Inside a captured graph the engine resets directly whatever MGYM_LL_STAGED_RESET says, since the measurement recorded in
# profiles/r02_lunarlander/launch_modes.txt: staged resets replayed as a graph cost 2.00 ms per step against 1.49.)
run() { python bench.py --workload lunar_lander --no-cpu-baseline --no-extra "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.3f ms/step %.3e steps/s  [%s]' % (d['ms_per_step'], d['value'], d['config']['launch']))"; }
for st in 1 0; do for ov in 1 0; do
  echo -n "staged=$st overlap=$ov graph: "; MGYM_LL_STAGED_RESET=$st MGYM_LL_OVERLAP=$ov run --launch graph
  echo -n "staged=$st overlap=$ov eager: "; MGYM_LL_STAGED_RESET=$st MGYM_LL_OVERLAP=$ov run --launch eager
done; done
echo -n "staged=1 overlap=1 graph, 16 steps: "; run --launch graph --steps 16 --warmup 640
echo -n "staged=0 overlap=1 graph, 16 steps: "; MGYM_LL_STAGED_RESET=0 run --launch graph --steps 16 --warmup 640
