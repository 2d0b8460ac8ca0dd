#!/usr/bin/env python
"""Yardstick only (not on the product path): what does the vendor library (hipBLASLt behind torch.matmul) reach for the
FLUX block-linear shapes on this box, next to this library's kernel?  bf16, y = x @ W^T, 20 launches each."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from diffusion_rs_amd import _lib as L  # noqa: E402

lib = L.load()
shapes = [(4608, 21504, 3072, "single linear1"), (4608, 3072, 15360, "single linear2"), (4096, 9216, 3072, "double qkv img"),
          (4096, 12288, 3072, "double mlp1 img"), (4096, 3072, 12288, "double mlp2 img"), (4096, 4096, 15360, "256 tiles"), (8192, 8192, 8192, "8k cube"),
          # round 5: the launches the MODEL issues group the image and text streams — 4608 rows, 18 row tiles — not the 4096 of the image stream alone
          (4608, 9216, 3072, "double qkv img+txt"), (4608, 12288, 3072, "double mlp1 img+txt"), (4608, 3072, 12288, "double mlp2 img+txt")]
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
    return e0.elapsed_time(e1) / n


for M, N, K, name in shapes:
    x = torch.randn(M, K, device="cuda").to(torch.bfloat16)
    w = (torch.randn(N, K, device="cuda") * 0.02).to(torch.bfloat16)
    y = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    wt = w.t()
    t_lt = timeit(lambda: torch.matmul(x, wt, out=y))
    t_me = timeit(lambda: L.check(lib.fmi_linear_bf16(p(x), p(w), None, p(y), M, N, K, 0, None)))
    fl = 2.0 * M * N * K
    print(f"{name:18s} M={M:5d} N={N:5d} K={K:5d}   hipBLASLt {fl / t_lt / 1e9:7.1f} TF ({t_lt * 1e3:6.1f} us)   this library {fl / t_me / 1e9:7.1f} TF ({t_me * 1e3:6.1f} us)", flush=True)


# ---- the 8-bit GEMMs (round 4): this library's e4m3 / int8 kernels on pre-quantised operands (fmi_gemm_q8) next to the vendor library's fp8 GEMM behind
# torch._scaled_mm (row-wise scales where this torch build has them, else per-tensor) — same shapes, bf16 output
lib.fmi_quantize_rows_fp8.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
print()
for M, N, K, name in shapes:
    x = torch.randn(M, K, device="cuda").to(torch.bfloat16)
    w = (torch.randn(N, K, device="cuda") * 0.02).to(torch.bfloat16)
    y = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    res = {}
    for kind, nm in ((1, "e4m3"), (2, "int8")):
        xq, wq = torch.empty(M, K, dtype=torch.uint8, device="cuda"), torch.empty(N, K, dtype=torch.uint8, device="cuda")
        xs, ws = torch.empty(M, dtype=torch.float32, device="cuda"), torch.empty(N, dtype=torch.float32, device="cuda")
        q = lib.fmi_quantize_rows_fp8 if kind == 1 else lib.fmi_quantize_rows_i8
        L.check(q(p(x), M, K, p(xq), p(xs), None))
        L.check(q(p(w), N, K, p(wq), p(ws), None))
        res[nm] = timeit(lambda: L.check(lib.fmi_gemm_q8(p(xq), p(xs), p(wq), p(ws), None, p(y), M, N, K, kind, 0, None)))
        if kind == 1:
            x8, w8 = xq.view(torch.float8_e4m3fn), wq.view(torch.float8_e4m3fn)
            t_lt, how = None, ""
            for sa, sb, label in ((xs.view(M, 1), ws.view(1, N), "row-wise scales"), (torch.ones((), device="cuda"), torch.ones((), device="cuda"), "per-tensor scales")):
                try:
                    t_lt = timeit(lambda: torch._scaled_mm(x8, w8.t(), scale_a=sa, scale_b=sb, out_dtype=torch.bfloat16))
                    how = label
                    break
                except Exception as e:  # noqa: BLE001
                    how = f"unavailable ({type(e).__name__})"
    fl = 2.0 * M * N * K
    lt = f"{fl / t_lt / 1e9:7.1f} TF ({t_lt * 1e3:6.1f} us, {how})" if t_lt else how
    print(f"{name:18s} M={M:5d} N={N:5d} K={K:5d}   hipBLASLt fp8 {lt}   this library e4m3 {fl / res['e4m3'] / 1e9:7.1f} TF ({res['e4m3'] * 1e3:6.1f} us)   "
          f"int8 {fl / res['int8'] / 1e9:7.1f} TOPS ({res['int8'] * 1e3:6.1f} us)", flush=True)
