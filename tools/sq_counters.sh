#!/bin/bash
# One rocprofv3 --pmc pass (8 SQ slots, no tracing) over tools/bin/attn_bench: how busy is the matrix pipe while a wave is resident?
#   tools/sq_counters.sh > gpurun_out/sq_attn.txt       (on the GPU box; the binary is built here: tools/run_attn_bench.sh or hipcc ... -o tools/bin/attn_bench)
# mfma_busy = (SQ_VALU_MFMA_BUSY_CYCLES / 4) / (SQ_WAVE_CYCLES / waves per SIMD): MI355X_MICROARCH.md, rocprofv3 PMC slots.
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=${OUT:-$ROOT/gpurun_out/sq_pass}
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
CTRS="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
rocprofv3 --pmc $CTRS -d "$OUT" -o sq -- "$ROOT/tools/bin/attn_bench" 3 > "$OUT/run.log" 2>&1 || tail -5 "$OUT/run.log"
DB=$(find "$OUT" -name "*.db" | head -1)
python3 - "$DB" <<'PY'
import sqlite3, sys, collections, re
con = sqlite3.connect(sys.argv[1])
rows = con.execute("select kernel_name, counter_name, count(*), sum(value) from counters_collection group by kernel_name, counter_name").fetchall()
d = collections.defaultdict(dict); calls = {}
for k, c, n, v in rows:
    d[k][c] = v; calls[k] = n
print("# rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE")
print("# one pass, no tracing, over tools/bin/attn_bench 3 (all kernels, all shapes; sums over all dispatches of a kernel)")
print(f"{'kernel':58s} {'calls':>5s} {'w/SIMD':>6s} {'mfma_busy':>9s} {'wait_any':>8s} {'wait_inst':>9s} {'active':>7s} {'wait_lds':>8s} {'lds_conflict/active':>19s}")
for k, c in sorted(d.items()):
    if "attention" not in k or "SQ_WAVE_CYCLES" not in c: continue
    wps = 1 if re.search(r"attention_w(4|16|16l|32)_kernel", k) else 2
    wc = c["SQ_WAVE_CYCLES"]
    busy = (c["SQ_VALU_MFMA_BUSY_CYCLES"] / 4) / (wc / wps)
    name = re.sub(r"\(.*", "", k)[:58]
    print(f"{name:58s} {calls[k]:5d} {wps:6d} {busy:9.3f} {c['SQ_WAIT_ANY']/wc:8.3f} {c['SQ_WAIT_INST_ANY']/wc:9.3f} {c['SQ_ACTIVE_INST_ANY']/wc:7.3f} {c['SQ_WAIT_INST_LDS']/wc:8.3f} {c['SQ_LDS_BANK_CONFLICT']/max(c['SQ_LDS_IDX_ACTIVE'],1):19.4f}")
PY
