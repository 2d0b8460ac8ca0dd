"""The op-level seam (S2 / S3 of SURVEY 8b) honours the reference's native-library convention: "asynchronous on the passed stream, caller
allocates" (diffusion_rs_backend/src/bitsandbytes/ffi.rs:5-114, op.rs:204-228).  VERDICT r4 weak 3: fmi_sdpa_bf16 / fmi_sdpa_fp8qk /
fmi_linear_fp8 / fmi_linear_i8 used to hipMalloc, hipStreamSynchronize and hipFree on every call — a Rust host that binds ops::sdpa
(ops.rs:247-262) to fmi_sdpa_bf16 would have drained the stream 57 times per denoise step.

Checked here: (i) 57 attention calls (one per DiT block of a step) enqueued behind ~100 ms of other work return to the host while that
work is still running — an event recorded BEFORE the calls has not completed when the last call returns —, (ii) their results equal the
caller-owned-workspace form's bit for bit and are what the same call gives on an idle stream, (iii) the library-held scratch is reused
(free device memory does not shrink with the number of calls), (iv) the same for the 8-bit linears, (v) the workspace forms reject a
workspace that is too small, and fmi_sdpa_fp8qk_ws takes its power-of-two score factor as an integer.
"""
import ctypes as C
import json

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env():
    import torch
    from diffusion_rs_amd import _lib as L
    lib = L.load()
    L.check(lib.fmi_init(0))
    return dict(torch=torch, L=L, lib=lib)


def _p(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def _busy(torch, ms_target=100):
    """enqueue ~ms_target of matmuls on the current stream; returns (an event recorded behind them, a tensor that keeps them alive)"""
    a = torch.randn(8192, 8192, device="cuda", dtype=torch.bfloat16)
    b = torch.randn(8192, 8192, device="cuda", dtype=torch.bfloat16)
    c = torch.empty_like(a)
    torch.matmul(a, b, out=c)  # warm (library heuristics, workspace) before anything is timed
    torch.cuda.synchronize()
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(8):
        torch.matmul(a, b, out=c)
    t1.record()
    torch.cuda.synchronize()
    per = t0.elapsed_time(t1) / 8
    n = max(8, int(ms_target / max(per, 1e-3)))

    def go():
        for _ in range(n):
            torch.matmul(a, b, out=c)
        ev = torch.cuda.Event()
        ev.record()
        return ev

    return go, (a, b, c), n * per


def test_57_sdpa_calls_behind_a_long_kernel_do_not_block_the_host(env):
    torch, L, lib = env["torch"], env["L"], env["lib"]
    B, H, Lq, n_calls = 1, 24, 1024, 57
    g = torch.Generator(device="cuda").manual_seed(57)
    q = torch.randn(B, H, Lq, 128, device="cuda", generator=g).to(torch.bfloat16)
    k = torch.randn(B, H, Lq, 128, device="cuda", generator=g).to(torch.bfloat16)
    v = torch.randn(B, H, Lq, 128, device="cuda", generator=g).to(torch.bfloat16)
    scale = 1.0 / 128 ** 0.5
    # reference results: the caller-owned-workspace form on an idle stream
    nbytes = lib.fmi_sdpa_workspace_bytes(B, H, Lq)
    assert nbytes == B * H * 128 * Lq * 2
    ws = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
    ref = torch.empty(B, Lq, H * 128, device="cuda", dtype=torch.bfloat16)
    L.check(lib.fmi_sdpa_bf16_ws(_p(q), _p(k), _p(v), _p(ref), B, H, Lq, Lq, 128, scale, 1, _p(ws), nbytes, None))
    torch.cuda.synchronize()
    assert torch.isfinite(ref.float()).all()
    # a workspace one byte short is refused, nothing is launched
    assert lib.fmi_sdpa_bf16_ws(_p(q), _p(k), _p(v), _p(ref), B, H, Lq, Lq, 128, scale, 1, _p(ws), nbytes - 1, None) == -1
    assert b"workspace" in lib.fmi_last_error()
    # the first call of a size on a stream allocates the library's scratch block for it; every later one reuses it
    outs = [torch.full_like(ref, float("nan")) for _ in range(n_calls)]
    L.check(lib.fmi_sdpa_bf16(_p(q), _p(k), _p(v), _p(outs[0]), B, H, Lq, Lq, 128, scale, 1, None))
    torch.cuda.synchronize()
    go, keep, busy_ms = _busy(torch)
    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info()[0]
    import time
    ev_busy = go()
    t0 = time.perf_counter()
    for o in outs:
        L.check(lib.fmi_sdpa_bf16(_p(q), _p(k), _p(v), _p(o), B, H, Lq, Lq, 128, scale, 1, None))
    host_ms = (time.perf_counter() - t0) * 1e3
    still_running = not ev_busy.query()  # the work enqueued BEFORE the 57 calls has not finished: nobody waited for the stream
    ev_end = torch.cuda.Event()
    ev_end.record()
    torch.cuda.synchronize()
    free1 = torch.cuda.mem_get_info()[0]
    print(f"57 x fmi_sdpa_bf16 (H=24, L=1024) behind {busy_ms:.0f} ms of matmuls: the host got all calls back in {host_ms:.2f} ms, "
          f"earlier work still running at that point: {still_running}; free HBM before / after: {free0 >> 20} / {free1 >> 20} MiB (scratch {nbytes >> 20} MiB)")
    assert still_running and host_ms < 0.5 * busy_ms
    for o in outs:
        assert torch.equal(o.view(torch.int16), ref.view(torch.int16))
    assert free0 - free1 <= 4 * nbytes  # one block reused: not 57 scratch buffers
    # what the header promises must be findable in the binary's behaviour too: no entry point of this family synchronises
    del keep


def test_fp8qk_and_q8_linears_are_stream_ordered_and_equal_their_workspace_forms(env):
    torch, L, lib = env["torch"], env["L"], env["lib"]
    B, H, Lq = 1, 8, 640
    g = torch.Generator(device="cuda").manual_seed(8)
    q8 = torch.randn(B, H, Lq, 128, device="cuda", generator=g).to(torch.float8_e4m3fn)
    k8 = torch.randn(B, H, Lq, 128, device="cuda", generator=g).to(torch.float8_e4m3fn)
    v = torch.randn(B, H, Lq, 128, device="cuda", generator=g).to(torch.bfloat16)
    n = -4
    scale = float(np.float32(2.0 ** n) / np.float32(1.4426950408889634))  # built in f32: scale * log2(e) may be one ulp off 2^n
    nbytes = lib.fmi_sdpa_workspace_bytes(B, H, Lq)
    ws = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
    fb0 = json.loads(lib.fmi_device_info().decode())["fp8_attention_fallbacks"]
    o_int = torch.empty(B, Lq, H * 128, device="cuda", dtype=torch.bfloat16)
    o_flt, o_pool = torch.empty_like(o_int), torch.empty_like(o_int)
    L.check(lib.fmi_sdpa_fp8qk_ws(_p(q8), _p(k8), _p(v), _p(o_int), B, H, Lq, Lq, 128, 0.0, n, 1, _p(ws), nbytes, None))  # the exponent as an integer
    L.check(lib.fmi_sdpa_fp8qk_ws(_p(q8), _p(k8), _p(v), _p(o_flt), B, H, Lq, Lq, 128, scale, L.SDPA_NO_EXP2, 1, _p(ws), nbytes, None))
    L.check(lib.fmi_sdpa_fp8qk(_p(q8), _p(k8), _p(v), _p(o_pool), B, H, Lq, Lq, 128, scale, 1, None))
    torch.cuda.synchronize()
    assert torch.isfinite(o_int.float()).all()
    assert torch.equal(o_int.view(torch.int16), o_flt.view(torch.int16)) and torch.equal(o_int.view(torch.int16), o_pool.view(torch.int16))
    assert json.loads(lib.fmi_device_info().decode())["fp8_attention_fallbacks"] == fb0  # all three took the one-wave stream
    assert lib.fmi_sdpa_fp8qk_ws(_p(q8), _p(k8), _p(v), _p(o_int), B, H, Lq, Lq, 128, 0.0, 3, 1, _p(ws), nbytes, None) == -1  # 2^3 > 1: not a score factor
    # a factor that is NOT a power of two still works (8-wave kernel) and is counted
    L.check(lib.fmi_sdpa_fp8qk(_p(q8), _p(k8), _p(v), _p(o_pool), B, H, Lq, Lq, 128, 0.043, 1, None))
    torch.cuda.synchronize()
    assert json.loads(lib.fmi_device_info().decode())["fp8_attention_fallbacks"] == fb0 + 1
    # ... but a caller who PICKED an 8-wave kernel is not a fallback
    L.check(lib.fmi_set_attention_kernel(1))
    try:
        L.check(lib.fmi_sdpa_fp8qk(_p(q8), _p(k8), _p(v), _p(o_pool), B, H, Lq, Lq, 128, scale, 1, None))
        torch.cuda.synchronize()
    finally:
        L.check(lib.fmi_set_attention_kernel(5))
    assert json.loads(lib.fmi_device_info().decode())["fp8_attention_fallbacks"] == fb0 + 1

    # ---- the 8-bit linears
    M, N, K = 1024, 3072, 3072
    x = torch.randn(M, K, device="cuda", generator=g).to(torch.bfloat16)
    w = (torch.randn(N, K, device="cuda", generator=g) / K ** 0.5).to(torch.bfloat16)
    bias = torch.randn(N, device="cuda", generator=g).to(torch.bfloat16)
    go, keep, busy_ms = _busy(torch, 60)
    import time
    for kind, quant, lin, lin_ws in ((1, lib.fmi_quantize_rows_fp8, lib.fmi_linear_fp8, lib.fmi_linear_fp8_ws),
                                     (2, lib.fmi_quantize_rows_i8, lib.fmi_linear_i8, lib.fmi_linear_i8_ws)):
        wq = torch.empty(N, K, dtype=torch.uint8, device="cuda")
        wscale = torch.empty(N, dtype=torch.float32, device="cuda")
        L.check(quant(_p(w), N, K, _p(wq), _p(wscale), None))
        nb = lib.fmi_linear_q8_workspace_bytes(M, K)
        assert nb >= M * K + 4 * M
        wsp = torch.empty(nb, dtype=torch.uint8, device="cuda")
        y_ws = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
        L.check(lin_ws(_p(x), _p(wq), _p(wscale), _p(bias), _p(y_ws), M, N, K, 0, _p(wsp), nb, None))
        assert lin_ws(_p(x), _p(wq), _p(wscale), _p(bias), _p(y_ws), M, N, K, 0, _p(wsp), nb - 1, None) == -1
        ys = [torch.empty_like(y_ws) for _ in range(20)]
        L.check(lin(_p(x), _p(wq), _p(wscale), _p(bias), _p(ys[0]), M, N, K, 0, None))  # (first call of this size: allocates the scratch block)
        torch.cuda.synchronize()
        ev_busy = go()
        t0 = time.perf_counter()
        for y in ys:
            L.check(lin(_p(x), _p(wq), _p(wscale), _p(bias), _p(y), M, N, K, 0, None))
        host_ms = (time.perf_counter() - t0) * 1e3
        still_running = not ev_busy.query()
        torch.cuda.synchronize()
        print(f"20 x fmi_linear_{'fp8' if kind == 1 else 'i8'} ({M}x{N}x{K}) behind {busy_ms:.0f} ms of matmuls: host back in {host_ms:.2f} ms, earlier work still running: {still_running}")
        assert still_running
        ref = (x.float() @ w.float().T + bias.float())
        for y in ys:
            assert torch.equal(y.view(torch.int16), y_ws.view(torch.int16))
        rel = float((y_ws.float() - ref).norm() / ref.norm())
        assert rel <= (6e-2 if kind == 1 else 2e-2), rel
    del keep


def test_groupnorm_op_is_stream_ordered(env):
    """fmi_groupnorm_nhwc (nn/group_norm.rs:39-74 as one op) took its partial-sum scratch from hipMalloc, waited for the stream and freed it on every call until
    round 5; it now uses the same per-stream scratch as the attention / linear entries: 40 calls behind ~60 ms of other work return while that work is running,
    and the result is the idle-stream result."""
    import time
    torch, L, lib = env["torch"], env["L"], env["lib"]
    g = torch.Generator(device="cuda").manual_seed(3)
    B, HW, Cc, G = 1, 128 * 128, 512, 32
    x = torch.randn(B, HW, Cc, device="cuda", generator=g).to(torch.bfloat16)
    w = torch.randn(Cc, device="cuda", generator=g)
    b = torch.randn(Cc, device="cuda", generator=g)
    lib.fmi_groupnorm_nhwc.argtypes = [C.c_void_p] * 4 + [C.c_int] * 4 + [C.c_float, C.c_int, C.c_void_p]
    ref = torch.empty_like(x)
    L.check(lib.fmi_groupnorm_nhwc(_p(x), _p(w), _p(b), _p(ref), B, HW, Cc, G, 1e-6, 1, None))
    torch.cuda.synchronize()
    go, keep, ms = _busy(torch, 60)
    outs = [torch.empty_like(x) for _ in range(40)]
    ev = go()
    t0 = time.perf_counter()
    for o in outs:
        L.check(lib.fmi_groupnorm_nhwc(_p(x), _p(w), _p(b), _p(o), B, HW, Cc, G, 1e-6, 1, None))
    host_ms = (time.perf_counter() - t0) * 1e3
    still_running = not ev.query()
    torch.cuda.synchronize()
    print(f"40 x fmi_groupnorm_nhwc (128x128x512) behind {ms:.0f} ms of matmuls: host back in {host_ms:.2f} ms, earlier work still running: {still_running}")
    assert still_running and host_ms < 0.5 * ms
    for o in outs:
        assert torch.equal(o, ref)
