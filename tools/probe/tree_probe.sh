#!/bin/bash
# instruction statistics of gen_tree_solve<1> per source line: bash tools/probe/tree_probe.sh  (CPU only, ~1 min)
set -e
cd "$(dirname "$0")/../.."
O=/tmp/tree_probe; mkdir -p $O
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -gline-tables-only -mllvm -disable-machine-licm -mllvm -disable-machine-sink -fno-signed-zeros -ffinite-math-only \
  --cuda-device-only --no-gpu-bundle-output -c -o $O/probe.o tools/probe/tree_probe.hip
/opt/rocm/lib/llvm/bin/llvm-objdump -d -l $O/probe.o > $O/probe.s
python3 tools/probe/line_hist.py $O/probe.s "${1:-gen_tree_solve}"
