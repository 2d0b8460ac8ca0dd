#!/bin/bash
# Multi-GPU sweep of the batch-sharded path in ONE call: N = 1, 2, 4, 8 (capped by the devices the node shows), one JSON line each
# into gpurun_out/scale_sweep.jsonl, plus a summary table.  Every line is `python bench.py --gpus N` started plainly (bench.py starts its own
# ranks under torch.distributed.run on 127.0.0.1) — exactly what the driver's SCALE run does, so N = 1 of the sweep IS the BENCH line.
# Reported per N: value (images/s, whole job), ms_per_step (max over ranks), ms_per_step_per_rank, broadcast_s / broadcast_gib (weights
# from rank 0, in place, 1-GiB messages), gather_ms (u8 images to rank 0).  Efficiency is for the reader (and the driver) to compute.
#   tools/scale_sweep.sh [steps=3] [warmup=1] [extra bench.py flags ...]     e.g.  tools/scale_sweep.sh 3 1 --sequence-parallel
set -u
cd "$(dirname "$0")/.."
STEPS=${1:-3}; WARM=${2:-1}; shift 2 2>/dev/null || true
mkdir -p gpurun_out
OUT=gpurun_out/scale_sweep.jsonl
: > "$OUT"
NDEV=$(python -c 'import torch; print(torch.cuda.device_count())')
echo "[scale_sweep] $NDEV device(s) visible; steps=$STEPS warmup=$WARM extra='$*'" >&2
export HSA_ENABLE_IPC_MODE_LEGACY=${HSA_ENABLE_IPC_MODE_LEGACY:-0}
for N in 1 2 4 8; do
  if [ "$N" -gt "$NDEV" ]; then echo "[scale_sweep] N=$N skipped: only $NDEV device(s)" >&2; continue; fi
  FLAGS="--no-cpu-baseline --no-secondary --no-live-traffic"
  # a sequence-parallel sweep has no N = 1 form (one image on one device is the plain run)
  EXTRA="$*"; [ "$N" -eq 1 ] && EXTRA=$(echo "$EXTRA" | sed 's/--sequence-parallel//; s/--split-k//')
  timeout 3000 python bench.py --gpus "$N" --steps "$STEPS" --warmup "$WARM" $FLAGS $EXTRA 2> "gpurun_out/scale_sweep_N$N.err" | grep '^{' >> "$OUT" \
    || { echo "[scale_sweep] N=$N failed, see gpurun_out/scale_sweep_N$N.err" >&2; tail -5 "gpurun_out/scale_sweep_N$N.err" >&2; }
done
python - "$OUT" <<'PY'
import json, sys
rows = [json.loads(l) for l in open(sys.argv[1]) if l.startswith("{")]
print(f"{'N':>2} {'images/s':>9} {'ms/image':>9} {'per-rank ms/image':<44} {'bcast s':>8} {'GiB':>6} {'gather ms':>9}")
for r in rows:
    print(f"{r['n_gpus']:>2} {r['value']:>9.4f} {r['ms_per_step']:>9.1f} {str(r.get('ms_per_step_per_rank')):<44} {str(r.get('broadcast_s')):>8} "
          f"{str(r.get('broadcast_gib')):>6} {str(r.get('gather_ms')):>9}")
PY
