#!/usr/bin/env python
"""Builds profiles/pmc_summary_latest.json (read by bench.py for roofline.traffic) from the two
per-counter summaries tools/profile_round.sh writes.

    tools/pmc_summary.py <tag> <fetch_size.txt> <write_size.txt> <denoise steps of the pmc pass> [kernel regex] [mode]

mode (default "latest" = the bf16 headline run) names the output: profiles/pmc_summary_<mode>.json; bench.py's secondary legs
read pmc_summary_nf4.json / pmc_summary_fp8.json the same way.

Traffic per launch = FETCH_SIZE x 2 + WRITE_SIZE (KiB as reported -> bytes), averaged over the
block-linear GEMM kernels weighted by dispatch count.  The x2 on FETCH_SIZE is the gfx950
correction of /opt/skills/guides/MI355X_MICROARCH.md (HBM section): a 16 B/lane stream's 128-B
requests are tallied as 64 B."""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def rows(path, rx):
    out = []
    for line in open(path):
        if line.startswith("#") or line.startswith("kernel "):
            continue
        m = re.match(r"^(.*\S)\s+(\d+)\s+([\d.]+)\s+([\d.]+)\s*$", line)
        if m and rx.search(m.group(1)):
            out.append((m.group(1), int(m.group(2)), float(m.group(3))))
    return out


def main():
    tag, fpath, wpath, steps = sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4])
    rx = re.compile(sys.argv[5] if len(sys.argv) > 5 and sys.argv[5] else r"fmi::gemm_(pp|w4|w4q)_kernel<(false|true|0|1|2|3)")
    mode = sys.argv[6] if len(sys.argv) > 6 else "latest"
    f, w = rows(fpath, rx), rows(wpath, rx)
    nf, nw = sum(c for _, c, _ in f), sum(c for _, c, _ in w)
    fa = sum(c * a for _, c, a in f) / max(nf, 1)
    wa = sum(c * a for _, c, a in w) / max(nw, 1)
    out = {
        "tag": tag,
        "kernels": sorted({n for n, _, _ in f}),
        "dispatches": nf,
        "source": f"profiles/{os.path.basename(fpath)} + profiles/{os.path.basename(wpath)}: separate rocprofv3 --pmc FETCH_SIZE / "
                  f"--pmc WRITE_SIZE passes over `bench.py --no-cpu-baseline --no-secondary --no-profile-pass --denoise-steps {steps} "
                  f"--steps 1 --warmup 0` (tools/profile_round.sh), averaged over the block-linear GEMM kernels weighted by dispatch count",
        "fetch_size_kib_avg_reported": round(fa, 1),
        "write_size_kib_avg_reported": round(wa, 1),
        "fetch_correction": "x2 (gfx950 FETCH_SIZE tallies the 128-B requests of a 16 B/lane stream as 64 B, MI355X_MICROARCH.md HBM section)",
        "traffic_bytes_per_launch": int((2 * fa + wa) * 1024),
        "note": "L2-miss side traffic (Infinity-Cache hits included); the QKV launches also write the head-major q/k and transposed v buffers from their epilogue",
    }
    out["mode"] = mode
    path = os.path.join(ROOT, "profiles", f"pmc_summary_{mode}.json")
    with open(path, "w") as fh:
        json.dump(out, fh, indent=1)
        fh.write("\n")
    print(json.dumps(out))


if __name__ == "__main__":
    main()
