#!/bin/bash
# four lanes per environment in the IK iteration, timed in isolation (GPU box): bash tools/probe/ik_lanes.sh [environments] [controller calls]
set -e
cd "$(dirname "$0")/../.."
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -disable-machine-licm -mllvm -disable-machine-sink -fno-signed-zeros -ffinite-math-only -Wno-unused-value -o /tmp/ik_lanes tools/probe/ik_lanes.hip
/tmp/ik_lanes "${1:-4096}" "${2:-35}"
/tmp/ik_lanes 1024 "${2:-35}"
