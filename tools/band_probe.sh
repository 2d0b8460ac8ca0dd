#!/bin/bash
# A/B of the tile order's band height (TILE_BAND) on the launches whose tile-row count is not a multiple of 8 (M = 4608: 18 tile rows):
# time per launch stand-alone + FETCH_SIZE per launch, one build per band height.
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd "$ROOT"
mkdir -p build gpurun_out/band
F="-O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=on -Wno-unused-value -Wno-unused-result"
SH="${SHAPES:-4608,21504,3072;4608,3072,15360;4608,12288,3072;4608,3072,3072}"
for B in ${BANDS:-8 6 9 8 6}; do
  /opt/rocm/bin/hipcc $F -DFMI_TILE_BAND=$B tools/gemm_bench.hip -o build/gemm_bench_b$B
  echo "=== band $B"
  for E in store resid; do
    FMI_EPI=$E FMI_COLD_W=4 FMI_SHAPES="$SH" ./build/gemm_bench_b$B 20 | grep -E "TF" | sed "s/^/$E /" | cut -c1-150
  done
done
cd /tmp && export TMPDIR=/tmp
for B in 8 6; do
  for S in 4608,21504,3072 4608,3072,15360; do
    T=$(echo $S | tr , x)
    rm -rf /tmp/bp_$B_$T
    FMI_SHAPES="$S" rocprofv3 --pmc FETCH_SIZE -d /tmp/bp_${B}_$T -o pmc -- "$ROOT/build/gemm_bench_b$B" 3 > /dev/null 2>&1 || true
    DB=$(find /tmp/bp_${B}_$T -name "*.db" | head -1)
    echo "== band $B  $S  FETCH_SIZE KiB/dispatch (x2 = bytes)"; python "$ROOT/profiles/summarize_rocpd.py" pmc "$DB" FETCH_SIZE | grep -E "gemm_pp_kernel" | cut -c1-60,97-140
  done
done
