#!/bin/bash
# What does each kind of work draw?  Runs a steady-state loop of one workload for a few seconds while sampling
# rocm-smi (socket power, shader clock) twice a second; prints the median of the samples taken while it ran.
#   tools/power_probe.sh            (on the GPU box; builds what it needs into build/)
cd "$(dirname "$0")/.."
mkdir -p build
F="-O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=on -Wno-unused-value -Wno-unused-result -DFMI_ALT_KERNELS=1"
[ -x build/gemm_bench ] || /opt/rocm/bin/hipcc $F tools/gemm_bench.hip -o build/gemm_bench
[ -x build/mfma_peak ] || /opt/rocm/bin/hipcc $F tools/mfma_peak.hip -o build/mfma_peak
[ -x build/load_rate ] || /opt/rocm/bin/hipcc $F tools/load_rate.hip -o build/load_rate
sample() {  # $1 = label, rest = command
  label=$1; shift
  ( "$@" > /tmp/pp_out.txt 2>&1 ) &
  pid=$!
  : > /tmp/pp_samples.txt
  sleep 1.0
  while kill -0 $pid 2>/dev/null; do
    rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Socket Graphics Package Power|sclk clock level" | tr '\n' ' ' >> /tmp/pp_samples.txt
    echo >> /tmp/pp_samples.txt
    sleep 0.4
  done
  python3 - "$label" <<'PY'
import re,sys,statistics
P=[];C=[]
for l in open('/tmp/pp_samples.txt'):
    m=re.search(r'Power \(W\): ([\d.]+)',l); c=re.search(r'\((\d+)Mhz\)',l)
    if m: P.append(float(m.group(1)))
    if c: C.append(int(c.group(1)))
tail=open('/tmp/pp_out.txt').read().strip().splitlines()[-1:] 
print(f"{sys.argv[1]:34s} samples {len(P):3d}  power median {statistics.median(P) if P else 0:7.1f} W (max {max(P) if P else 0:7.1f})   sclk median {statistics.median(C) if C else 0:6.0f} MHz   | {tail[0][:110] if tail else ''}")
PY
}
sample "idle (sleep)" sleep 3
sample "mfma only (register operands)" ./build/mfma_peak 600000
FMI_SHAPES="4096,4096,15360" sample "gemm 8-wave pp + 4-wave, 256 tiles" ./build/gemm_bench 1500
FMI_SHAPES="4608,21504,3072" sample "gemm linear1 (1512 tiles)" ./build/gemm_bench 1500
sample "loads only, L2-resident set" ./build/load_rate 300 64
sample "loads only, Infinity-Cache set" ./build/load_rate 100 512
sample "loads only, HBM set" ./build/load_rate 100 4096
