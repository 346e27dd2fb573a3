#!/bin/bash
# Runs ON the GPU box: CartPole step time for the step kernel's block sizes (MGYM_CARTPOLE_BLOCK), interleaved
# repetitions so that box / clock drift shows up as spread within a config rather than as a difference between configs.
pr() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%s us/step=%.3f frac=%.4f' % (sys.argv[1], d['ms_per_step']*1e3, d['roofline']['frac']))" "$1"; }
for rep in 1 2 3; do
  for b in 256 128 64; do
    MGYM_CARTPOLE_BLOCK=$b python bench.py --no-extra --no-cpu-baseline 2>/dev/null | pr "1Mi block=$b rep=$rep"
  done
done
for b in 256 128 64; do
  MGYM_CARTPOLE_BLOCK=$b python bench.py --no-extra --no-cpu-baseline --envs 8388608 --steps 400 --warmup 64 2>/dev/null | pr "8Mi block=$b"
done
