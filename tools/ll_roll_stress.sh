#!/bin/bash
# usage: [RUNS=20] [N=262144] [RUN_TIMEOUT=90] [MGYM_LL_ROLL_HELPER=..] tools/ll_roll_stress.sh   (the engine's knobs pass through the environment)
# (*GPU box*) repeated timing runs of mgym_rollout (each: 640 warm-up steps, then K x step and rollout phases alternating); stops at the first failure and keeps its output
O=gpurun_out/roll_stress.txt; : > $O
for r in $(seq 1 ${RUNS:-20}); do
  for K in 16 64; do
    MGYM_LL_ROLL_STATS=1 timeout -k 10 ${RUN_TIMEOUT:-90} python tools/ll_roll_check.py time ${N:-262144} $K $((K == 16 ? 10 : 4)) > gpurun_out/roll_stress_one.txt 2>&1
    rc=$?
    echo "run $r K=$K rc=$rc $(grep '^n=' gpurun_out/roll_stress_one.txt | sed 's/.*mgym_rollout//')" >> $O
    awk '/ll_rollout K=/{ split($0,a,"total "); split(a[2],b," "); if (b[1]+0 > 200000) print "SLOW LAUNCH: " substr($0,1,500) }' gpurun_out/roll_stress_one.txt >> $O
    if [ $rc -ne 0 ]; then cp gpurun_out/roll_stress_one.txt gpurun_out/roll_stress_failure.txt; echo "FAILED" >> $O; exit 1; fi
  done
done
echo "stress ok" >> $O
