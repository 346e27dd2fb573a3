#!/bin/bash
# Runs ON the GPU box: how long do the roles of the single-launch step kernel take apart?  Builds four diagnostic binaries
# (LL_ROLE_MASK: 1 contact only, 2 free flight only, 7 all) next to the product and times ONE step from the same saved state for
# each (tools/_ll_role_time).  usage: tools/ll_role_time.sh [envs]
set -e
cd "$(dirname "$0")/.."
for m in 1 2 3 7; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math -Wno-unused-result -DLL_ROLE_MASK=$m tools/ll_role_time.hip -o tools/_ll_role_time_$m &
done
wait
for m in 7 1 2 3 7; do echo -n "roles mask $m: "; timeout -k 10 120 tools/_ll_role_time_$m ${1:-262144}; done
