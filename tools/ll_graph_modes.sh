#!/bin/bash
# Runs ON the GPU box: LunarLander step time by launch mode — eager launches vs hipGraph replay (stream capture of the fused
# order, staged resets inside the graph or not) — over THREE instantiations each (a fresh process per line: a new capture, a new
# hipGraphInstantiate), 64 timed steps after 640 warm-up steps.  usage: tools/ll_graph_modes.sh [envs]
run() { python bench.py --workload lunar_lander ${1:+--envs $1} --steps 64 --warmup 640 --no-cpu-baseline --no-extra "${@:2}" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.3f ms/step %.3e steps/s  [%s]' % (d['ms_per_step'], d['value'], d['config']['launch']))"; }
n=$1
for rep in 1 2 3; do
  echo -n "instantiation $rep  eager:                          "; run "$n" --launch eager
  echo -n "instantiation $rep  graph, staged resets in graph:  "; run "$n" --launch graph
  echo -n "instantiation $rep  graph, captured steps reset directly (MGYM_LL_STAGED_IN_GRAPH=0): "; MGYM_LL_STAGED_IN_GRAPH=0 run "$n" --launch graph
done
echo -n "one graph of exactly 16 step launches: "; python bench.py --workload lunar_lander ${n:+--envs $n} --steps 16 --warmup 640 --no-cpu-baseline --no-extra --launch graph 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.3f ms/step [%s]' % (d['ms_per_step'], d['config']['launch']))"
