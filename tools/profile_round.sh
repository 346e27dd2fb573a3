#!/bin/bash
# (*GPU box*) the counter records of a round: every workload through tools/profile_pmc.sh, in two gpurun calls (arg 1: a | b)
R=${2:-r04}
if [ "$1" = "a" ]; then
  bash tools/profile_pmc.sh ${R}_cp --workload cartpole --steps 512 --warmup 64 > gpurun_out/pmc_${R}_cp.log 2>&1 &&
  bash tools/profile_pmc.sh ${R}_cp32 --workload cartpole --envs 33554432 --steps 160 --warmup 32 > gpurun_out/pmc_${R}_cp32.log 2>&1 &&
  bash tools/profile_pmc.sh ${R}_mc --workload mountain_car --steps 512 --warmup 64 > gpurun_out/pmc_${R}_mc.log 2>&1 &&
  bash tools/profile_pmc.sh ${R}_mcc --workload mountain_car_cont --steps 512 --warmup 64 > gpurun_out/pmc_${R}_mcc.log 2>&1 &&
  bash tools/profile_pmc.sh ${R}_mc32 --workload mountain_car --envs 33554432 --steps 160 --warmup 32 > gpurun_out/pmc_${R}_mc32.log 2>&1
else
  bash tools/profile_pmc.sh ${R}_ll --workload lunar_lander --steps 128 --warmup 640 > gpurun_out/pmc_${R}_ll.log 2>&1 &&
  bash tools/profile_pmc.sh ${R}_ll_roll64 --workload lunar_lander --ll-rollout 64 --steps 256 --warmup 640 > gpurun_out/pmc_${R}_ll_roll64.log 2>&1 &&
  bash tools/profile_pmc.sh ${R}_ll_roll16 --workload lunar_lander --ll-rollout 16 --steps 128 --warmup 640 > gpurun_out/pmc_${R}_ll_roll16.log 2>&1 &&
  bash tools/profile_pmc.sh ${R}_ll_1mi --workload lunar_lander --envs 1048576 --steps 32 --warmup 640 > gpurun_out/pmc_${R}_ll_1mi.log 2>&1
fi
echo "profile_round $1 rc=$?"
