#!/bin/bash
# Static ISA evidence for the lighting kernel: opcode classes per basic block of k_lighting<2, false>,
# the straight-line bodies of the point / spot walks marked as loops.  hipcc cross-compiles; no GPU needed.
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
S=$(mktemp /tmp/lighting.XXXX.s)
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -S --cuda-device-only "$ROOT/granite_amd/csrc/lighting.hip" -o "$S" 2>/dev/null
python "$ROOT/tools/isa_blocks.py" "$S" k_lightingILi2ELb0 12
rm -f "$S"
