#!/bin/bash
# kernel trace of two images of the headline config + the per-dispatch attention table (tools/attn_dispatch_trace.py); run on the GPU box through gpurun:
#   tools/attn_trace.sh r06
set -e
TAG=${1:-r06}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/attn_trace_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d "$OUT/kt" -o kt -- python "$ROOT/bench.py" --steps 1 --warmup 1 --no-cpu-baseline --no-secondary --no-live-traffic --no-profile-pass $BENCH_ARGS > "$OUT/bench.json" 2> "$OUT/kt.err" || tail -5 "$OUT/kt.err"
cd "$ROOT"
KT=$(find "$OUT/kt" -name "*.db" | head -1)
python tools/attn_dispatch_trace.py "$KT" > "$OUT/${TAG}_attention_dispatch_trace.txt"
head -80 "$OUT/${TAG}_attention_dispatch_trace.txt"
[ -n "$KEEP_DB" ] || rm -rf "$OUT/kt"
