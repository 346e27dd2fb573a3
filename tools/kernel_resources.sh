#!/bin/bash
# Per-kernel registers / scratch / LDS of one .hip file of the engine, from the compiler's resource-usage remarks.
#   tools/kernel_resources.sh lunar_lander.hip
cd "$(dirname "$0")/../modurl_gym_amd/csrc" || exit 1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -c "$1" -o /tmp/kres_$$.o \
    -Rpass-analysis=kernel-resource-usage 2>&1 |
  python3 -c '
import re, sys, subprocess
cur = None
for line in sys.stdin:
    m = re.search(r"remark:\s+(Function Name|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]): (\S+)", line)
    if not m: continue
    k, v = m.groups()
    if k == "Function Name":
        cur = subprocess.run(["c++filt", v], capture_output=True, text=True).stdout.strip().split("(")[0]
        print()
        print(f"{cur:56s}", end="")
    else:
        print(f" {k.split()[0]}={v}", end="")
print()
'
rm -f /tmp/kres_$$.o
