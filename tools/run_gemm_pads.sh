#!/bin/bash
set -e
cd "$(dirname "$0")/.."
mkdir -p build
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=on -Wno-unused-value -Wno-unused-result -DFMI_ALT_KERNELS=1 tools/gemm_bench.hip -o build/gemm_bench
for pads in "0 0 0" "64 64 0" "64 0 0" "0 64 0" "64 64 64" "32 32 0" "128 128 0"; do ./build/gemm_bench 10 $pads | grep -v "txt\|proj img"; done
