#!/bin/bash
# Collects the rocprofv3 evidence committed under profiles/ (run on the GPU box through gpurun):
#   tools/profile_round.sh <tag>        e.g. r01c      (BENCH_ARGS="--quant fp8" adds bench flags to every pass)
# 1. kernel trace + stats of the default bench command (whole 50-step image, no CPU baseline)
# 2./3. FETCH_SIZE and WRITE_SIZE in SEPARATE --pmc passes (PMC_STEPS denoise steps, default 10), never combined with traces
# 4. profiles/pmc_summary_<mode>.json rebuilt from 2./3. (tools/pmc_summary.py; mode = latest for the bf16 headline run, nf4, fp8)
set -e
TAG=${1:-r02x}
PMC_STEPS=${PMC_STEPS:-10}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d "$OUT/kt" -o kt -- python "$ROOT/bench.py" --no-cpu-baseline --no-secondary --no-live-traffic $BENCH_ARGS > "$OUT/bench_under_rocprof.json" 2> "$OUT/kt.err" || tail -5 "$OUT/kt.err"
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C -d "$OUT/pmc_$C" -o pmc -- python "$ROOT/bench.py" --no-cpu-baseline --no-secondary $BENCH_ARGS --no-profile-pass --denoise-steps $PMC_STEPS --steps 1 --warmup 0 > "$OUT/pmc_$C.log" 2>&1 || tail -5 "$OUT/pmc_$C.log"
done
cd "$ROOT"
KT=$(find "$OUT/kt" -name "*.db" | head -1)
python profiles/summarize_rocpd.py kernel "$KT" > "$OUT/${TAG}_kernel_stats.txt"
for C in FETCH_SIZE WRITE_SIZE; do
  DB=$(find "$OUT/pmc_$C" -name "*.db" | head -1)
  python profiles/summarize_rocpd.py pmc "$DB" $C > "$OUT/${TAG}_pmc_$(echo $C | tr A-Z a-z).txt"
done
# traffic per block-linear launch of this mode -> profiles/pmc_summary_<mode>.json (bench.py reads it: roofline.traffic of the line / of the leg)
MODE=latest
case "$BENCH_ARGS" in *"--quant nf4"*) MODE=nf4 ;; *"--quant fp8"*) MODE=fp8 ;; *"--quant int8"*) MODE=int8 ;; esac
python tools/pmc_summary.py "$TAG" "$OUT/${TAG}_pmc_fetch_size.txt" "$OUT/${TAG}_pmc_write_size.txt" "$PMC_STEPS" "" $MODE > /dev/null; cp profiles/pmc_summary_$MODE.json "$OUT/pmc_summary_$MODE.json"
tail -1 "$OUT/bench_under_rocprof.json" > "$OUT/${TAG}_bench_under_rocprof.json"
head -12 "$OUT/${TAG}_kernel_stats.txt"
head -6 "$OUT/${TAG}_pmc_fetch_size.txt"
head -6 "$OUT/${TAG}_pmc_write_size.txt"
# the rocpd databases are tens of MB each and gpurun merges at most 64 MiB back: keep only the summaries unless asked
[ -n "$KEEP_DB" ] || rm -rf "$OUT/kt" "$OUT/pmc_FETCH_SIZE" "$OUT/pmc_WRITE_SIZE"
