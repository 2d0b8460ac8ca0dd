#!/bin/bash
# In-model A/B of an environment switch, alternating processes on ONE box (box-to-box variance is +-2 %):
#   tools/env_ab.sh "FMI_LN_WAVE=0" [reps]      A = with the assignment, B = without; BENCH_ARGS adds bench flags (e.g. --quant fp8)
# The default bench line without its secondary legs and CPU baseline, one warm-up and one timed 50-step image per process.
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd "$ROOT"
SW="$1"
REPS=${2:-3}
OUT=gpurun_out/env_ab_$(echo "$SW$BENCH_ARGS" | tr -c 'A-Za-z0-9\n' _)
mkdir -p "$OUT"
ARGS="--no-cpu-baseline --no-secondary --no-live-traffic --steps 1 --warmup 1 $BENCH_ARGS"
for i in $(seq 1 $REPS); do
  env $SW python bench.py $ARGS > "$OUT/A_$i.json" 2> "$OUT/A_$i.err"
  python bench.py $ARGS > "$OUT/B_$i.json" 2> "$OUT/B_$i.err"
done
python - "$OUT" "$REPS" "$SW" <<'PY'
import json, sys
out, reps, sw = sys.argv[1], int(sys.argv[2]), sys.argv[3]
for tag, name in (("A", sw), ("B", "default")):
    for i in range(1, reps + 1):
        try:
            d = json.loads(open(f"{out}/{tag}_{i}.json").read().strip().splitlines()[-1])
            ph = d["phase_ms_per_denoise_step"]
            print(f"{name:24s} ms/step {d['ms_per_denoise_step']:6.2f}  GEMM {d['roofline']['achieved']:7.1f} TF  ln {ph['layernorm_mod']:6.3f}  qkv {ph['gemm_qkv']:6.3f}  proj {ph['gemm_proj']:6.3f}  mlp {ph['gemm_mlp']:6.3f}  attn {ph['attention']:6.3f}")
        except Exception as e:
            print(tag, i, "failed:", e)
PY
