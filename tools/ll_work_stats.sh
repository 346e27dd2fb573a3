#!/bin/bash
# CPU only: statistics of the LunarLander contact path from the KERNEL SOURCE compiled for the host (the same build as
# tests/test_ll_host.py, plus counters): cached contacts per env-step, time-of-impact rounds, and how many of the 180
# velocity sweeps a sub-step really runs before its state repeats.  usage: tools/ll_work_stats.sh [envs=2048] [steps=600] [wind=0]
set -e
cd "$(dirname "$0")/.."
python3 -c "from oracle import oracle as o; o.build()"
mkdir -p tests/native/_build
g++ -O2 -std=c++17 -ffp-contract=off -fno-fast-math -DLL_HOST_STATS -o tests/native/_build/ll_host_stats tests/native/ll_host_check.cpp \
    -Loracle/_build -loracle -Wl,-rpath,$PWD/oracle/_build -lm
tests/native/_build/ll_host_stats ${1:-2048} ${2:-600} ${3:-0} 0 | grep -v "^   \|^-- "
