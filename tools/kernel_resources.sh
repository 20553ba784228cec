#!/bin/bash
# Developer tool: VGPRs / scratch / occupancy / LDS of every kernel of one translation unit (hipcc remarks), one line per kernel.
#   tools/kernel_resources.sh kpal_quads [extra hipcc flags]
unit=${1:-kpal_quads}; shift
cd "$(dirname "$0")/.."
hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fPIC -fvisibility=hidden -c kpal_amd/csrc/$unit.hip -o /dev/null \
      -Rpass-analysis=kernel-resource-usage "$@" 2>&1 | grep "remark:" | sed 's/ \[-Rpass-analysis=kernel-resource-usage\]//' |
  awk '/Function Name:/ {name=$NF} / VGPRs:/ {v=$NF} / AGPRs:/ {a=$NF} /ScratchSize/ {s=$NF} /Occupancy/ {o=$NF} /LDS Size/ {print name, "vgpr", v, "agpr", a, "scratch", s, "occ", o, "lds", $NF}' |
  while read n rest; do echo "$(echo $n | c++filt | cut -c1-120) $rest"; done
