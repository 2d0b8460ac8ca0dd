#!/usr/bin/env python
"""K sweep of one launch shape through the C-ABI: bf16 (fmi_linear_bf16), e4m3 and int8 (fmi_gemm_q8 on pre-quantised operands), plain bf16 store.
The slope over K is the K loop, the intercept is pipeline fill + epilogue of the launch's rounds of tiles — the split VERDICT r4 weak 9 asks about for the
8-bit kernels (their K loop is half as many 128-byte tiles long, so the intercept weighs double).   python tools/q8_ksweep.py [M N]"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from diffusion_rs_amd import _lib as L  # noqa: E402

lib = L.load()
M, N = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (4608, 21504)
p = lambda t: C.c_void_p(t.data_ptr())


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


rows = []
for K in (1024, 2048, 3072, 4096, 6144, 8192):
    x = torch.randn(M, K, device="cuda").to(torch.bfloat16)
    w = (torch.randn(N, K, device="cuda") * 0.02).to(torch.bfloat16)
    y = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    t = {"bf16": timeit(lambda: L.check(lib.fmi_linear_bf16(p(x), p(w), None, p(y), M, N, K, 0, None)))}
    for kind, nm in ((1, "e4m3"), (2, "int8")):
        xq, wq = torch.empty(M, K, dtype=torch.uint8, device="cuda"), torch.empty(N, K, dtype=torch.uint8, device="cuda")
        xs, ws = torch.empty(M, dtype=torch.float32, device="cuda"), torch.empty(N, dtype=torch.float32, device="cuda")
        q = lib.fmi_quantize_rows_fp8 if kind == 1 else lib.fmi_quantize_rows_i8
        L.check(q(p(x), M, K, p(xq), p(xs), None))
        L.check(q(p(w), N, K, p(wq), p(ws), None))
        t[nm] = timeit(lambda: L.check(lib.fmi_gemm_q8(p(xq), p(xs), p(wq), p(ws), None, p(y), M, N, K, kind, 0, None)))
    rows.append((K, t))
    print(f"{M} x {N} x {K:5d}   bf16 {t['bf16']:7.1f} us   e4m3 {t['e4m3']:7.1f} us   int8 {t['int8']:7.1f} us", flush=True)
tiles = -(-M // 256) * -(-N // 256)
rounds = tiles / 256
for nm, kb in (("bf16", 64), ("e4m3", 128), ("int8", 128)):
    ks = np.array([r[0] for r in rows], float)
    ts = np.array([r[1][nm] for r in rows], float)
    a, b = np.polyfit(ks, ts, 1)
    print(f"{nm}: {a * kb / np.ceil(rounds):.3f} us per K tile ({kb} k) and round of tiles, intercept {b:.1f} us per launch = {b / np.ceil(rounds):.1f} us per round "
          f"({tiles} tiles = {rounds:.2f} rounds); at K = 3072 the intercept is {b / (a * 3072 + b):.0%} of the launch")
