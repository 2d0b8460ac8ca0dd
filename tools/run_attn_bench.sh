#!/bin/bash
# VARIANTS="full -DFMI_ATT_SETPRIO ..." tools/run_attn_bench.sh [iters]
set -e
cd "$(dirname "$0")/.."
mkdir -p build
for v in ${VARIANTS:-full}; do
  echo "=== variant: $v"; [ "$v" = full ] && v=""
  /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=on -Wno-unused-value -Wno-unused-result -DFMI_ALT_KERNELS=1 $v tools/attn_bench.hip -o build/attn_bench
  ./build/attn_bench ${1:-20}
done
