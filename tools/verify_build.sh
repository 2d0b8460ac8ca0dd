#!/bin/bash
# One gpurun call that checks the library built in this tree end to end: smoke(), the -m gpu suite, the default bench line.
#   gpurun --timeout 1500 -- 'bash tools/verify_build.sh <tag>'      -> gpurun_out/<tag>/{smoke.log,gpu_tests.log,bench.json,bench.err}
TAG=${1:-verify}
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
python -c "import __graft_entry__ as g; g.smoke()" > "$OUT/smoke.log" 2>&1; echo "smoke rc=$?"; tail -2 "$OUT/smoke.log"
timeout 900 python -m pytest tests -m gpu -x -q > "$OUT/gpu_tests.log" 2>&1; echo "tests rc=$?"; tail -3 "$OUT/gpu_tests.log"
timeout 600 python bench.py $BENCH_ARGS > "$OUT/bench.json" 2> "$OUT/bench.err"; echo "bench rc=$?"
python - "$OUT/bench.json" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("images/s", round(d["value"], 4), "ms/step", d["ms_per_denoise_step"], "GEMM TF", d["roofline"]["achieved"], "frac", d["roofline"]["frac"])
print({k: (v.get("value"), v.get("ms_per_denoise_step")) for k, v in d.get("secondary", {}).items() if isinstance(v, dict) and "value" in v})
PY
