#!/bin/bash
# Register / scratch use of the kernels of one translation unit: tools/kernel_resources.sh adt_seq.hip [name filter]
cd "$(dirname "$0")/../adt_amd/csrc" || exit 1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -c "$1" -o /tmp/_kr.o -Rpass-analysis=kernel-resource-usage 2>&1 |
  awk '/Function Name/ {name=$5} /VGPRs:/ {v=$4} /ScratchSize/ {s=$5} /Occupancy/ {o=$5} /LDS Size/ {print name, "vgpr", v, "scratch", s, "occ", o, "lds", $6}' | grep -- "${2:-.}" | while read n rest; do echo "$(echo $n | c++filt | cut -c1-70) $rest"; done
