#!/bin/bash
# In-model A/B of the per-problem band height (launch_gemm: pick_tile_band) against the fixed height 8 (FMI_GEMM_BAND=8), alternating
# processes on ONE box: the default bench line without its secondary legs, 50 denoise steps, one timed image per process.
#   gpurun -- 'bash tools/band_ab.sh [reps]'  -> gpurun_out/band_ab/{A,B}_<i>.json + a table on stdout
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd "$ROOT"
OUT=gpurun_out/band_ab
mkdir -p "$OUT"
REPS=${1:-3}
ARGS="--no-cpu-baseline --no-secondary --no-live-traffic --steps 1 --warmup 1 $BENCH_ARGS"
for i in $(seq 1 $REPS); do
  FMI_GEMM_BAND=8 python bench.py $ARGS > "$OUT/A_$i.json" 2> "$OUT/A_$i.err"
  python bench.py $ARGS > "$OUT/B_$i.json" 2> "$OUT/B_$i.err"
done
python - "$OUT" "$REPS" <<'PY'
import json, sys
out, reps = sys.argv[1], int(sys.argv[2])
for tag, name in (("A", "band 8 (fixed)"), ("B", "band picked per problem")):
    rows = []
    for i in range(1, reps + 1):
        try:
            d = json.loads(open(f"{out}/{tag}_{i}.json").read().strip().splitlines()[-1])
            ph = d["phase_ms_per_denoise_step"]
            rows.append((d["ms_per_denoise_step"], d["roofline"]["achieved"], ph["gemm_qkv"], ph["gemm_proj"], ph["gemm_mlp"], ph["attention"]))
        except Exception as e:
            print(tag, i, "failed:", e)
    for r in rows:
        print(f"{name:26s} ms/step {r[0]:6.2f}  GEMM {r[1]:7.1f} TF  qkv {r[2]:6.3f}  proj {r[3]:6.3f}  mlp {r[4]:6.3f}  attn {r[5]:6.3f}")
PY
