#!/bin/bash
# A/B of whole-frame variants on ONE GPU box, alternating (the boxes differ by a few per cent among themselves: only pairs taken in one call compare).
# A variant is  name[:VAR=value[,VAR=value...]]  -- environment switches of the executor / kernel library, or GRANITE_LIB_DIR=lib_<x> for a
# library built with other flags (make -C granite_amd/csrc OUT=../lib_<x> EXTRA_<unit>="-D...").
# Usage (through gpurun):  bash tools/frame_ab.sh <tag> "<workload> ..." <variant> <variant> ... [-- extra bench.py arguments]
#   bash tools/frame_ab.sh lead "config3_4k_4096lights config4_4k_smaa_taa" lead3:GRANITE_HOST_LEAD_FRAMES=3 lead2:GRANITE_HOST_LEAD_FRAMES=2
#   bash tools/frame_ab.sh mfma config3_4k_4096lights base mfma:GRANITE_LIB_DIR=lib_mfma -- --steps 200 --warmup 20 --sustain-seconds 0
TAG=$1; WORKLOADS=$2; shift 2
VARIANTS=(); EXTRA=()
while [ $# -gt 0 ]; do if [ "$1" = "--" ]; then shift; EXTRA=("$@"); break; fi; VARIANTS+=("$1"); shift; done
O=gpurun_out/$TAG; mkdir -p $O
for round in 1 2; do
  for v in "${VARIANTS[@]}"; do
    name=${v%%:*}; settings=""; [ "$v" != "$name" ] && settings=${v#*:}
    for wl in $WORKLOADS; do
      ( IFS=,; for kv in $settings; do export "$kv"; done
        timeout 300 python bench.py --workload $wl --no-cpu-baseline "${EXTRA[@]}" > $O/bench_${wl}_$name.$round.json 2>/dev/null )
      python tools/bench_brief.py $O/bench_${wl}_$name.$round.json | sed "s/^/$name $wl /"
    done
  done
done
