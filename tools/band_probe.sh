#!/bin/bash
# A/B of the tile order's band height on the launches whose tile-row count is not a multiple of 8 (M = 4608: 18 tile rows):
# time per launch stand-alone + FETCH_SIZE per launch; FMI_GEMM_BAND=<n> pins the height (launch_gemm: pick_tile_band), "auto" = the
# per-problem choice.  (The first version of this probe, whose output is in profiles/r03_band_probe.txt, compiled one binary per height.)
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd "$ROOT"
mkdir -p build gpurun_out/band
F="-O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=on -Wno-unused-value -Wno-unused-result -DFMI_ALT_KERNELS=1"
SH="${SHAPES:-4608,21504,3072;4608,3072,15360;4608,12288,3072;4608,3072,3072}"
/opt/rocm/bin/hipcc $F tools/gemm_bench.hip -o build/gemm_bench
pin() { if [ "$1" = auto ]; then env -u FMI_GEMM_BAND "${@:2}"; else env FMI_GEMM_BAND=$1 "${@:2}"; fi; }
for B in ${BANDS:-8 6 9 8 auto}; do
  echo "=== band $B"
  for E in store resid; do
    FMI_EPI=$E FMI_COLD_W=4 FMI_SHAPES="$SH" pin $B ./build/gemm_bench 20 | grep -E "TF" | sed "s/^/$E /" | cut -c1-150
  done
done
cd /tmp && export TMPDIR=/tmp
for B in 8 auto; do
  for S in 4608,21504,3072 4608,3072,15360; do
    T=$(echo $S | tr , x)
    rm -rf /tmp/bp_${B}_$T
    FMI_SHAPES="$S" pin $B rocprofv3 --pmc FETCH_SIZE -d /tmp/bp_${B}_$T -o pmc -- "$ROOT/build/gemm_bench" 3 > /dev/null 2>&1 || true
    DB=$(find /tmp/bp_${B}_$T -name "*.db" | head -1)
    echo "== band $B  $S  FETCH_SIZE KiB/dispatch (x2 = bytes)"; python "$ROOT/profiles/summarize_rocpd.py" pmc "$DB" FETCH_SIZE | grep -E "gemm_pp_kernel" | cut -c1-60,97-140
  done
done
