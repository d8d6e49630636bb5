#!/bin/bash
# usage: tools/kernel_resources.sh rsba_amd/csrc/<file>.hip  — prints VGPR/SGPR/scratch/occupancy/LDS per kernel
f=$1
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -Rpass-analysis=kernel-resource-usage -c "$f" -o /dev/null 2>&1 \
 | grep -E "remark:" | sed -E 's/.*remark: +//; s/ \[-Rpass.*//' \
 | awk '/Function Name/{name=$3} /^VGPRs:/{v=$2} /^AGPRs/{a=$2} /TotalSGPRs/{s=$2} /ScratchSize/{sc=$3} /Occupancy/{o=$4} /LDS Size/{printf "%-70s vgpr=%s agpr=%s sgpr=%s scratch=%s occ=%s lds=%s\n", name, v, a, s, sc, o, $4}'
