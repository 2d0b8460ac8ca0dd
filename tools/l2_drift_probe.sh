#!/bin/bash
# Does the L2-miss traffic of the ping-pong GEMM grow with the number of rounds a launch takes (tiles of an XCD's patch drifting apart)?
# One rocprofv3 --pmc FETCH_SIZE pass of tools/gemm_bench per shape (the kernel name is the same for every shape, so one run each).
#   gpurun -- 'bash tools/l2_drift_probe.sh'   -> gpurun_out/l2_drift/<shape>.txt  (KiB per dispatch as reported; x2 for bytes on gfx950)
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd "$ROOT"
mkdir -p build gpurun_out/l2_drift
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=on -Wno-unused-value -Wno-unused-result -DFMI_ALT_KERNELS=1 $PROBE_DEFS tools/gemm_bench.hip -o build/gemm_bench
cd /tmp && export TMPDIR=/tmp
for S in ${SHAPES:-4096,4096,3072 8192,8192,3072 4096,12288,3072 4608,21504,3072 4608,3072,15360}; do
  T=$(echo $S | tr , x)
  rm -rf /tmp/l2p_$T
  FMI_SHAPES="$S" rocprofv3 --pmc FETCH_SIZE -d /tmp/l2p_$T -o pmc -- "$ROOT/build/gemm_bench" 3 > "$ROOT/gpurun_out/l2_drift/$T.bench.txt" 2>&1 || tail -3 "$ROOT/gpurun_out/l2_drift/$T.bench.txt"
  DB=$(find /tmp/l2p_$T -name "*.db" | head -1)
  python "$ROOT/profiles/summarize_rocpd.py" pmc "$DB" FETCH_SIZE > "$ROOT/gpurun_out/l2_drift/$T.txt"
  echo "== $S"; grep -E "gemm_(pp|w4|bf16)_kernel" "$ROOT/gpurun_out/l2_drift/$T.txt" | cut -c1-60,97-140
  grep -E "TF" "$ROOT/gpurun_out/l2_drift/$T.bench.txt" | head -4
done
