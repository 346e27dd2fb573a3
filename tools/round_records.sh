#!/bin/bash
# (*GPU box*) the non-counter records of a round: the whole GPU test suite, the bench lines kept under profiles/, and the rollout's
# own wave trace / per-wave statistics.  Every step is joined with && — the first failure ends the call.
R=${1:-r04}
O=gpurun_out/records_$R
mkdir -p $O
python -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1 &&
python bench.py > $O/bench_default.json 2> $O/bench_default.err &&
python bench.py --steps 20 --warmup 5 > $O/bench_driver_config.json 2> $O/bench_driver_config.err &&
python bench.py --workload lunar_lander --no-extra > $O/bench_ll.json 2> $O/bench_ll.err &&
python bench.py --workload lunar_lander --ll-rollout 16 --no-extra > $O/bench_ll_rollout_K16.json 2> $O/bench_ll_rollout_K16.err &&
python bench.py --workload lunar_lander --ll-rollout 64 --no-extra > $O/bench_ll_rollout_K64.json 2> $O/bench_ll_rollout_K64.err &&
python bench.py --workload lunar_lander --envs 1048576 --no-extra > $O/bench_ll_1mi.json 2> $O/bench_ll_1mi.err &&
python bench.py --workload mixed --no-extra > $O/bench_mixed.json 2> $O/bench_mixed.err &&
python bench.py --workload mountain_car --no-extra > $O/bench_mountain_car.json 2> $O/bench_mountain_car.err &&
python bench.py --workload mountain_car_cont --no-extra > $O/bench_mountain_car_cont.json 2> $O/bench_mountain_car_cont.err &&
python bench.py --envs 33554432 --no-extra > $O/bench_32mi.json 2> $O/bench_32mi.err &&
for K in 8 16 64; do
  MGYM_LL_ROLL_TRACE=$O/trace_K$K.bin MGYM_LL_ROLL_STATS=1 python tools/ll_roll_check.py time 262144 $K 4 > $O/rollout_stats_K$K.txt 2>&1 &&
  python tools/ll_roll_trace.py $O/trace_K$K.bin 250 > $O/rollout_timeline_K$K.txt 2>&1 && rm -f $O/trace_K$K.bin || exit 1
done &&
for N in 65536 131072 262144 524288 1048576; do
  MGYM_LL_ROLLOUT=1 python tools/ll_roll_check.py time $N 16 10 >> $O/rollout_population.txt 2>&1 &&
  MGYM_LL_ROLLOUT=1 python tools/ll_roll_check.py time $N 64 4 >> $O/rollout_population.txt 2>&1 || exit 1
done &&
python tools/ll_roll_check.py check 65536 1024 64 > $O/soak_rollout.txt 2>&1 &&
python tools/ll_roll_check.py check 16384 1536 16 >> $O/soak_rollout.txt 2>&1 &&
python tools/ll_roll_check.py check 8192 1200 8 1 0 >> $O/soak_rollout.txt 2>&1
echo "round_records rc=$?"
