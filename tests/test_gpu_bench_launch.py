"""`python bench.py --gpus 2` started PLAINLY (no torchrun, no WORLD_SIZE): bench.py must spawn its two ranks itself, run the
batch-sharded control flow (rank-0 weight generation, in-place state broadcast, barrier, max-over-ranks timing, gather) and
print one JSON line with n_gpus = 2.  The box has one GPU, so the two ranks share it through gloo (FMI_BENCH_BACKEND=gloo, the
documented debugging mode): this checks the launcher and the N > 1 control flow, not RCCL performance."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_gpus2_self_launch_gloo():
    env = dict(os.environ, FMI_BENCH_BACKEND="gloo")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--height", "256", "--width", "256",
           "--denoise-steps", "2", "--txt-tokens", "64", "--no-cpu-baseline", "--no-secondary", "--no-profile-pass"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    print({k: out[k] for k in ("n_gpus", "value", "broadcast_s", "broadcast_gib", "broadcast_messages", "gather_ms", "ms_per_step_per_rank", "backend")})
    assert out["n_gpus"] == 2 and out["rccl_ranks"] == 2 and out["backend"] == "gloo"
    assert out["output_ok"] and out["scaling"] == "weak" and out["config"]["global_batch"] == 2
    assert len(out["ms_per_step_per_rank"]) == 2 and out["ms_per_step"] == pytest.approx(max(out["ms_per_step_per_rank"]), rel=1e-3, abs=0.06)
    assert out["broadcast_gib"] > 20 and out["broadcast_messages"] <= 30
    assert "starting 2 ranks" in r.stderr
