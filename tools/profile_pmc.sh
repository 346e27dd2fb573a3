#!/bin/bash
# Runs ON the GPU box (via gpurun): rocprofv3 kernel trace + separate PMC passes of bench.py.
# Usage: tools/profile_pmc.sh <tag> <bench args...>; writes gpurun_out/pmc_<tag>/{trace,sqA,sqB,sqC,fetch,write}
# Each counter group is collected in its own run (8 SQ slots per pass; FETCH_SIZE and WRITE_SIZE do not
# fit one pass; --pmc is never combined with the sys/hip/hsa trace domains).  The program itself follows
# `--` (python3 bench.py): no env/bash hop between rocprofv3 and the process that initialises the GPU.
set -e
TAG=$1; shift
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--no-cpu-baseline --no-extra $@"
SQA="SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE"
SQB="SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_FLAT SQ_INSTS_SMEM SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_TRANS_F32"
SQC="SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $REPO/bench.py $ARGS > $OUT/trace.log 2>&1
echo "trace done"
rocprofv3 --pmc $SQA --output-format csv -d $OUT/sqA -- python3 $REPO/bench.py $ARGS > $OUT/sqA.log 2>&1
echo "sqA done"
rocprofv3 --pmc $SQB --output-format csv -d $OUT/sqB -- python3 $REPO/bench.py $ARGS > $OUT/sqB.log 2>&1
echo "sqB done"
rocprofv3 --pmc $SQC --output-format csv -d $OUT/sqC -- python3 $REPO/bench.py $ARGS > $OUT/sqC.log 2>&1
echo "sqC done"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 $REPO/bench.py $ARGS > $OUT/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 $REPO/bench.py $ARGS > $OUT/write.log 2>&1
echo "fetch/write done"
cd $REPO
python3 tools/summarize_pmc.py $OUT > $OUT/summary.txt
cat $OUT/summary.txt
# keep the merge small: drop the per-dispatch tables, keep stats + summary
find $OUT -name "*kernel_trace.csv" -size +1M -delete
find $OUT -name "*counter_collection.csv" -size +1M -delete
