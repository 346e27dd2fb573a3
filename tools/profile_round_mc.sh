#!/bin/bash
# (*GPU box*) the MountainCar part of tools/profile_round.sh a (after a change to mountain_car.hip alone)
R=${1:-r04}
bash tools/profile_pmc.sh ${R}_mc --workload mountain_car --steps 512 --warmup 64 > gpurun_out/pmc_${R}_mc.log 2>&1 &&
bash tools/profile_pmc.sh ${R}_mcc --workload mountain_car_cont --steps 512 --warmup 64 > gpurun_out/pmc_${R}_mcc.log 2>&1 &&
bash tools/profile_pmc.sh ${R}_mc32 --workload mountain_car --envs 33554432 --steps 160 --warmup 32 > gpurun_out/pmc_${R}_mc32.log 2>&1
echo "profile_round_mc rc=$?"
