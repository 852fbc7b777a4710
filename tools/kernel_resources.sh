#!/bin/bash
# register / scratch / LDS use of every kernel of one source: tools/kernel_resources.sh airpose_amd/csrc/stem.hip [-DAP_F16 ...]
SRC=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -S --offload-device-only "$@" $SRC -o /tmp/kres.s 2>/dev/null
awk '/^[ \t]*\.type[ \t]+.*,@function/{name=$2; sub(/,@function/,"",name)} /; NumVgprs:/{v=$3} /; NumAgprs:/{a=$3} /; ScratchSize:/{s=$3} /; Occupancy:/{o=$3} /; LDSByteSize:/{l=$3; printf "%s vgpr %s agpr %s scratch %s lds %s occupancy %s\n", name, v, a, s, l, o}' /tmp/kres.s | c++filt | sed -e 's/(anonymous namespace):://g' | cut -c1-220
