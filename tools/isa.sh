#!/bin/bash
# usage: tools/isa.sh <file.hip>  -> /tmp/t/<name>.s + resource summary + flat-access count
cd /root/repo/4mc_amd/csrc && mkdir -p /tmp/t && f=$(basename $1 .hip)
hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../../include -I. -S --cuda-device-only $f.hip -o /tmp/t/$f.s -Rpass-analysis=kernel-resource-usage 2>&1 | grep -E "Function Name|VGPRs:|SGPRs:|Scratch|Occupancy|LDS Size|error" | sed 's/.*remark: //' | tr '\n' ' ' | sed 's/Function Name/\nFunction Name/g'; echo
echo "$f flat_ ops: $(grep -c 'flat_' /tmp/t/$f.s)"
