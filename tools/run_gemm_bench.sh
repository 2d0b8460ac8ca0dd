#!/bin/bash
# Builds and runs GEMM micro-benchmark variants on the GPU box.
#   tools/run_gemm_bench.sh [iters]           full kernel only
#   VARIANTS="full -DFMI_ABLATE_NO_LOAD -DFMI_ABLATE_NO_MFMA" tools/run_gemm_bench.sh
set -e
cd "$(dirname "$0")/.."
mkdir -p build gpurun_out
F="-O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=on -Wno-unused-value -Wno-unused-result -DFMI_ALT_KERNELS=1"
for v in ${VARIANTS:-full}; do
  echo "=== variant: $v"
  [ "$v" = full ] && v=""
  /opt/rocm/bin/hipcc $F $v tools/gemm_bench.hip -o build/gemm_bench
  ./build/gemm_bench ${1:-10}
done
