#!/bin/bash
# Runs ON the GPU box (via gpurun): rocprofv3 kernel trace + PMC passes of bench.py.
# Usage: tools/profile_gpu.sh <tag> [bench args...]; writes gpurun_out/prof_<tag>/{trace,fetch,write}
# Counters are collected in their own runs (FETCH_SIZE and WRITE_SIZE do not fit one pass;
# never combined with --sys-trace et al.).  The program itself follows `--` (python3 bench.py).
set -e
TAG=$1; shift
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 480 --warmup 32 --no-cpu-baseline --no-extra $@"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $REPO/bench.py $ARGS > $OUT/trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 $REPO/bench.py $ARGS > $OUT/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 $REPO/bench.py $ARGS > $OUT/write.log 2>&1
cd $REPO
python3 tools/summarize_prof.py $OUT > $OUT/summary.txt
cat $OUT/summary.txt
# keep the merge small: drop the per-dispatch traces, keep stats + summary
find $OUT -name "*kernel_trace.csv" -size +2M -delete
find $OUT -name "*counter_collection.csv" -size +2M -delete
