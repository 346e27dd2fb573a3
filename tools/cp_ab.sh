#!/bin/bash
# Runs ON the GPU box: A/B of libmgym.so against libmgym_base.so on CartPole (interleaved repetitions), with knobs from the environment.
pr() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%s us/step=%.3f frac=%.4f' % (sys.argv[1], d['ms_per_step']*1e3, d['roofline']['frac']))" "$1"; }
L=modurl_gym_amd
cp $L/libmgym.so /tmp/libmgym_cur.so
for rep in 1 2 3; do
  cp $L/libmgym_base.so $L/libmgym.so; python bench.py --no-extra --no-cpu-baseline $@ 2>/dev/null | pr "base rep=$rep"
  cp /tmp/libmgym_cur.so $L/libmgym.so
  for p in 1 2; do MGYM_CARTPOLE_PASSES=$p python bench.py --no-extra --no-cpu-baseline $@ 2>/dev/null | pr "cur passes=$p rep=$rep"; done
done
