#!/usr/bin/env python3
"""Per-dispatch view of the attention launches inside the denoise loop (VERDICT r5 "next" 4a): which of a step's 57 launches take 203 us and which 240+,
and what runs in front of them?   Input: the rocpd sqlite database of `rocprofv3 --kernel-trace -- python bench.py ...` (tools/attn_trace.sh).

For the LAST image of the run: per position in the step (0..18 double blocks, 19..56 single blocks) the mean / min / max duration over the 50 steps, the kernel
dispatched in front and the idle gap to it; then a histogram, the per-step totals (does the time drift along the image: clock / temperature?), and the same
split for the block-linear GEMM as a yardstick."""
import sqlite3
import sys
from collections import defaultdict


def main(db, kernel_pat="attention_w16l", per_step=57, steps=50):
    c = sqlite3.connect(db)
    cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
    need = {"name", "start", "end", "duration"}
    if not need <= set(cols):
        print("kernels view columns:", cols)
        raise SystemExit("unexpected rocpd schema")
    rows = list(c.execute("select name, start, end, duration from kernels order by start"))
    att = [i for i, r in enumerate(rows) if kernel_pat in r[0]]
    print(f"# {db}: {len(rows)} dispatches, {len(att)} of *{kernel_pat}*")
    n = per_step * steps
    if len(att) < n:
        raise SystemExit(f"fewer than {n} attention dispatches")
    last = att[-n:]
    pos = defaultdict(list)
    gaps = defaultdict(list)
    prev_name = {}
    step_tot = [0.0] * steps
    for k, i in enumerate(last):
        s, p = divmod(k, per_step)
        d = rows[i][3] / 1e3
        pos[p].append(d)
        step_tot[s] += d
        gaps[p].append((rows[i][1] - rows[i - 1][2]) / 1e3)
        prev_name[p] = rows[i - 1][0].replace("void ", "").replace("fmi::", "").replace("(anonymous namespace)::", "")[:64]
    alld = [d for v in pos.values() for d in v]
    print(f"# last image: {n} launches, mean {sum(alld) / len(alld):.1f} us, min {min(alld):.1f}, max {max(alld):.1f}")
    print(f"{'pos':>3s} {'kind':6s} {'mean_us':>8s} {'min_us':>8s} {'max_us':>8s} {'gap_us':>7s}  kernel in front")
    for p in range(per_step):
        v = pos[p]
        print(f"{p:3d} {'double' if p < 19 else 'single':6s} {sum(v) / len(v):8.1f} {min(v):8.1f} {max(v):8.1f} {sum(gaps[p]) / len(gaps[p]):7.1f}  {prev_name[p]}")
    dbl = [d for p in range(19) for d in pos[p]]
    sgl = [d for p in range(19, per_step) for d in pos[p]]
    print(f"# double-block launches: mean {sum(dbl) / len(dbl):.1f} us; single-block launches: mean {sum(sgl) / len(sgl):.1f} us")
    lo = int(min(alld) // 5 * 5)
    hist = defaultdict(int)
    for d in alld:
        hist[int(d // 5 * 5)] += 1
    print("# histogram (5 us bins): " + "  ".join(f"{b}:{hist[b]}" for b in sorted(hist)))
    print("# attention time per step (ms), steps 0..49: " + " ".join(f"{t / 1e3:.2f}" for t in step_tot))
    # yardstick: the dominant GEMM over the same window
    w0, w1 = rows[last[0]][1], rows[last[-1]][2]
    g = [r[3] / 1e3 for r in rows if "gemm_pp_kernel" in r[0] and w0 <= r[1] <= w1]
    if g:
        per = len(g) // steps
        gt = [sum(g[s * per:(s + 1) * per]) / 1e3 for s in range(steps)]
        print(f"# gemm_pp_kernel per step (ms), {per} launches each: " + " ".join(f"{t:.2f}" for t in gt))
    tot_busy = sum(r[3] for r in rows if w0 <= r[1] <= w1) / 1e6
    print(f"# window {(w1 - w0) / 1e6:.1f} ms wall, {tot_busy:.1f} ms of kernels")


if __name__ == "__main__":
    main(*sys.argv[1:2])
