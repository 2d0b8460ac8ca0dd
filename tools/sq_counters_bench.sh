#!/bin/bash
# One rocprofv3 --pmc pass (8 SQ slots, no tracing) over a short bench.py run (BENCH_ARGS, default: 4 denoise steps of the headline config): per kernel of the
# denoise step, how busy is the matrix pipe while a wave is resident, how much of the wave's time issues instructions, LDS bank conflicts.
#   tools/sq_counters_bench.sh > gpurun_out/sq_bench.txt      (on the GPU box)
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=${OUT:-$ROOT/gpurun_out/sq_bench_pass}
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
CTRS="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
rocprofv3 --pmc $CTRS -d "$OUT" -o sq -- python "$ROOT/bench.py" --no-cpu-baseline --no-secondary --no-profile-pass --no-live-traffic --denoise-steps 4 --steps 1 --warmup 0 $BENCH_ARGS > "$OUT/run.log" 2>&1 || tail -5 "$OUT/run.log"
DB=$(find "$OUT" -name "*.db" | head -1)
python3 - "$DB" <<'PY'
import sqlite3, sys, collections, re
con = sqlite3.connect(sys.argv[1])
rows = con.execute("select kernel_name, counter_name, count(*), sum(value) from counters_collection group by kernel_name, counter_name").fetchall()
d = collections.defaultdict(dict); calls = {}
for k, c, n, v in rows:
    d[k][c] = v; calls[k] = n
print("# rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE")
print("# one pass, no tracing, over bench.py --denoise-steps 4 --steps 1 (sums over all dispatches of a kernel); w/SIMD = resident waves per SIMD the kernel runs with")
print(f"{'kernel':62s} {'calls':>5s} {'w/SIMD':>6s} {'mfma_busy':>9s} {'wait_any':>8s} {'wait_inst':>9s} {'active':>7s} {'wait_lds':>8s} {'lds_conflict/active':>19s}")
for k, c in sorted(d.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0)):
    if "fmi::" not in k or "SQ_WAVE_CYCLES" not in c or c["SQ_WAVE_CYCLES"] == 0: continue
    if not re.search(r"gemm|attention|layernorm|quantize", k): continue
    wps = 1 if re.search(r"attention_w(4|16|16l|32)_kernel|gemm_w4", k) else 2
    wc = c["SQ_WAVE_CYCLES"]
    busy = (c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / 4) / (wc / wps)
    name = re.sub(r"\(.*", "", k)[:62]
    print(f"{name:62s} {calls[k]:5d} {wps:6d} {busy:9.3f} {c['SQ_WAIT_ANY']/wc:8.3f} {c['SQ_WAIT_INST_ANY']/wc:9.3f} {c['SQ_ACTIVE_INST_ANY']/wc:7.3f} {c['SQ_WAIT_INST_LDS']/wc:8.3f} {c['SQ_LDS_BANK_CONFLICT']/max(c['SQ_LDS_IDX_ACTIVE'],1):19.4f}")
PY
