#!/bin/bash
# register / scratch / LDS use of every kernel of one source: tools/kernel_resources.sh airpose_amd/csrc/stem.hip [-DAP_F16 ...]
SRC=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -S --offload-device-only "$@" $SRC -o /tmp/kres.s 2>/dev/null
awk '/^; Kernel info:/{f=1} /^\s*\.amdhsa_kernel /{name=$2} /; NumVgprs:/{v=$3} /; NumAgprs:/{a=$3} /; ScratchSize:/{s=$3} /; Occupancy:/{o=$3} /; LDSByteSize:/{l=$3; printf "%-90s vgpr %3s agpr %3s scratch %4s lds %6s occ %s\n", substr(name,1,90), v, a, s, l, o}' /tmp/kres.s
