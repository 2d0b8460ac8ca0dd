"""int8 mode (fmi_flux_quantize_int8, round 4) through the C-ABI vs the oracle's restatement of the same recipe.

Like the fp8 mode this recipe is the library's own (the reference has no 8-bit MFMA path), restated in
oracle/flux_oracle.cpp (orc_quantize_rows_i8, orc_linear_i8, lin_blk mode 5).  Why it exists: tools/fp8_noise_study.py — an e4m3
operand is 2.65e-2 rms from its value whatever the scale granularity, a per-row int8 one 8.5e-3, at the same matrix-pipe rate.

Round 5: the operands that follow a GELU (the input of the double blocks' MLP-out, the gelu(mlp) segment of the single blocks' linear2 input) take
256 levels over their [min, max] instead of a grid symmetric around 0 (fp8.hip's header; oracle: orc_quantize_rows_i8_asym) — same bars.

Bars: quantisation bit-exact (codes and scales); the int8 MFMA's k map pinned by an EXACT integer GEMM (unit scales, sums below 256:
every output is an integer that bf16 holds); int8 GEMM vs the oracle on the same codes rel-L2 <= 2e-3 (bf16 output rounding only —
the sums are exact integers on both sides); model evaluation vs the int8 oracle <= 1e-2 (the bf16 path's bar), vs the f32 oracle
<= 2e-2 on the small synthetic model; latents after the Euler loop <= 3e-2.  The full-size bar (<= 3e-2 from f32 with the default
mask, FLUX.1-dev in full) is tests/test_gpu_production_shapes.py's.
"""
import ctypes as C

import numpy as np
import pytest

from tests.util import SMALL_FLUX, bf16_round, dev, flux_inputs, host, rel_l2

pytestmark = pytest.mark.gpu


def _p(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


@pytest.fixture(scope="module")
def env():
    import torch
    import diffusion_rs_amd as d
    from diffusion_rs_amd import _lib as L
    from oracle import oracle as orc
    return dict(torch=torch, d=d, L=L, lib=L.load(), orc=orc)


def gpu_quantize(env, x_bf16):
    torch, L, lib = env["torch"], env["L"], env["lib"]
    rows, K = x_bf16.shape
    q = torch.empty(rows, K, dtype=torch.int8, device="cuda")
    s = torch.empty(rows, dtype=torch.float32, device="cuda")
    L.check(lib.fmi_quantize_rows_i8(_p(x_bf16), rows, K, _p(q), _p(s), None))
    torch.cuda.synchronize()
    return q, s


@pytest.mark.parametrize("rows,K", [(1, 8), (5, 136), (33, 3072), (7, 15360), (300, 256), (2, 16384)])
def test_quantize_rows_i8_bit_exact(env, rows, K):
    torch, orc = env["torch"], env["orc"]
    rng = np.random.default_rng(rows * 131 + K)
    x = rng.standard_normal((rows, K)).astype(np.float32) * (10.0 ** rng.uniform(-3, 3, (rows, 1))).astype(np.float32)
    if rows > 2:
        x[1] = 0            # an all-zero token: scale 1e-30 / 127, codes 0
        x[2, ::3] *= 1e-4   # values far below the row maximum round to 0
    if rows > 4:            # exact ties of the rounding: x * (127 / absmax) = n + 0.5 -> round half to even
        x[4, :] = 0
        x[4, 0] = 127.0
        x[4, 1:8] = [0.5, 1.5, 2.5, -0.5, -1.5, -2.5, 126.5]
    x = bf16_round(x)
    q, s = gpu_quantize(env, dev(x, torch.bfloat16))
    rq, rs = orc.quantize_rows_i8(x)
    np.testing.assert_array_equal(s.cpu().numpy(), rs)
    np.testing.assert_array_equal(q.cpu().numpy(), rq)
    assert int(np.abs(rq.astype(np.int32)).max()) <= 127
    if rows > 4:
        assert rq[4, :8].tolist() == [127, 0, 2, 2, 0, -2, -2, 126]


@pytest.mark.parametrize("M,N,K", [(64, 256, 128), (300, 384, 1280), (257, 1024, 15360), (4608, 512, 3072)])
def test_int8_gemm_is_an_exact_integer_gemm(env, M, N, K):
    """Unit scales (every row of x and of w holds one +-127 at a position where the other operand is 0) and sums below 256: the output is
    the integer sum over k itself, exactly, in bf16.  Any error in which k a (lane, byte) of the MFMA operands carries — A and W must
    agree — or in the tile / K-tile walk shows as a wrong integer."""
    torch, L, lib = env["torch"], env["L"], env["lib"]
    rng = np.random.default_rng(M + N + K)
    x = np.zeros((M, K), np.float32)
    w = rng.integers(-1, 2, (N, K)).astype(np.float32) * rng.integers(1, 4, (N, K)).astype(np.float32)  # dense, values in -3..3
    for r in range(M):  # up to 80 non-zeros of +-1 per row of x: |sum| <= 240
        idx = rng.choice(np.arange(2, K), size=min(80, K - 2), replace=False)
        x[r, idx] = rng.choice([-1.0, 1.0], size=idx.size)
    x[:, 0], x[:, 1] = 127.0, 0.0
    w[:, 0], w[:, 1] = 0.0, 127.0 * rng.choice([-1.0, 1.0], size=N)
    xd, wd = dev(x, torch.bfloat16), dev(w, torch.bfloat16)
    wq, ws = gpu_quantize(env, wd)
    assert torch.equal(ws, torch.ones_like(ws)) and torch.equal(wq.float().cpu(), torch.from_numpy(w))
    y = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
    L.check(lib.fmi_linear_i8(_p(xd), _p(wq), _p(ws), None, _p(y), M, N, K, 0, None))
    torch.cuda.synchronize()
    ref = x.astype(np.float64) @ w.astype(np.float64).T
    assert np.abs(ref).max() <= 256
    np.testing.assert_array_equal(host(y).astype(np.float64), ref)


@pytest.mark.parametrize("M,N,K,epi", [(64, 256, 128, 0), (300, 384, 256, 0), (257, 1024, 1280, 1), (1000, 260, 384, 0), (16, 3072, 1024, 0), (130, 512, 15360, 0)])
def test_linear_i8_matches_oracle(env, M, N, K, epi):
    torch, L, lib, orc = env["torch"], env["L"], env["lib"], env["orc"]
    rng = np.random.default_rng(M + N + K)
    x = bf16_round(rng.standard_normal((M, K)).astype(np.float32) * (1 + 10 * (rng.random((M, 1)) < 0.1)).astype(np.float32))
    w = bf16_round((rng.standard_normal((N, K)) * 0.05).astype(np.float32))
    b = bf16_round(rng.standard_normal(N).astype(np.float32))
    xd, wd, bd = dev(x, torch.bfloat16), dev(w, torch.bfloat16), dev(b, torch.bfloat16)
    wq, ws = gpu_quantize(env, wd)
    y = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
    L.check(lib.fmi_linear_i8(_p(xd), _p(wq), _p(ws), _p(bd), _p(y), M, N, K, epi, None))
    torch.cuda.synchronize()
    ref = orc.linear_i8(x, w, b)
    f32 = orc.linear(x, w, b)
    if epi == 1:
        ref, f32 = orc.gelu(ref), orc.gelu(f32)
    err, noise = rel_l2(host(y), ref), rel_l2(ref, f32)
    # beyond the norm: element by element the two differ by bf16 rounding of the same f32 value (+ one ulp of f32 where the GPU contracts
    # the scale and bias into a fused multiply-add)
    close = np.abs(host(y) - ref) <= np.abs(ref) * 2.0 ** -8 + 1e-6 * np.abs(ref).max() if epi == 0 else None
    print(f"linear_i8 {M}x{N}x{K} epi={epi}: rel-L2 vs int8 oracle {err:.2e}; int8 recipe vs f32 linear {noise:.2e}")
    assert err <= 2e-3
    assert close is None or bool(close.all())
    assert noise <= 2.5e-2  # (the e4m3 recipe on the same operands: 3.7e-2 and up)


@pytest.mark.parametrize("kind", [1, 2], ids=["e4m3", "int8"])
def test_gemm_q8_on_prequantised_operands_equals_the_linear(env, kind):
    """fmi_gemm_q8 (both operands quantised by the caller) is the same launch as fmi_linear_fp8 / fmi_linear_i8 behind their internal row pass: bit-identical."""
    torch, L, lib = env["torch"], env["L"], env["lib"]
    M, N, K = 300, 520, 1280
    g = torch.Generator(device="cuda").manual_seed(5)
    x = torch.randn(M, K, device="cuda", generator=g).to(torch.bfloat16)
    w = (torch.randn(N, K, device="cuda", generator=g) * 0.05).to(torch.bfloat16)
    b = torch.randn(N, device="cuda", generator=g).to(torch.bfloat16)
    quant = lib.fmi_quantize_rows_fp8 if kind == 1 else lib.fmi_quantize_rows_i8
    linear = lib.fmi_linear_fp8 if kind == 1 else lib.fmi_linear_i8
    xq, wq = torch.empty(M, K, dtype=torch.uint8, device="cuda"), torch.empty(N, K, dtype=torch.uint8, device="cuda")
    xs, ws = torch.empty(M, dtype=torch.float32, device="cuda"), torch.empty(N, dtype=torch.float32, device="cuda")
    L.check(quant(_p(x), M, K, _p(xq), _p(xs), None))
    L.check(quant(_p(w), N, K, _p(wq), _p(ws), None))
    y1, y2 = torch.empty(M, N, dtype=torch.bfloat16, device="cuda"), torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
    for epi in (0, 1):
        L.check(linear(_p(x), _p(wq), _p(ws), _p(b), _p(y1), M, N, K, epi, None))
        L.check(lib.fmi_gemm_q8(_p(xq), _p(xs), _p(wq), _p(ws), _p(b), _p(y2), M, N, K, kind, epi, None))
        torch.cuda.synchronize()
        assert torch.equal(y1.view(torch.int16), y2.view(torch.int16))
    assert lib.fmi_gemm_q8(_p(xq), _p(xs), _p(wq), _p(ws), None, _p(y2), M, N, K, 3, 0, None) < 0   # no such kind
    assert lib.fmi_gemm_q8(_p(xq), None, _p(wq), _p(ws), None, _p(y2), M, N, K, 1, 0, None) < 0


def _post_gelu_rows(rng, rows, K, d0):
    """rows shaped like the library's post-GELU operands: a signed segment [0, d0) (attention rows) and gelu(h), h ~ N(0, sigma_row), behind it"""
    x = np.empty((rows, K), np.float32)
    x[:, :d0] = 0.3 * rng.standard_normal((rows, d0))
    h = rng.standard_normal((rows, K - d0)) * (10.0 ** rng.uniform(-1, 1, (rows, 1)))
    x[:, d0:] = 0.5 * h * (1.0 + np.tanh(0.7978845608 * h * (1.0 + 0.044715 * h * h)))
    return x.astype(np.float32)


@pytest.mark.parametrize("rows,K,d0", [(3, 16, 0), (64, 512, 0), (300, 15360, 3072), (7, 4096, 1024), (33, 12288, 0), (5, 16384, 16)])
def test_quantize_rows_i8_asym_bit_exact(env, rows, K, d0):
    """The int8 recipe's post-GELU form (round 5, fp8.hip's header): codes, step and offset bit for bit against the oracle — a constant row, an
    all-zero row, a front segment that dictates the step, ties of the rounding."""
    torch, L, lib, orc = env["torch"], env["L"], env["lib"], env["orc"]
    rng = np.random.default_rng(rows * 7 + K + d0)
    x = _post_gelu_rows(rng, rows, K, d0)
    if rows > 2:
        x[1] = 0.0                      # all zero: step 1e-30, codes -128 behind d0 (value = lo = 0), 0 in front
        x[2, d0:] = 0.75                # a constant offset segment: hi == lo
    if rows > 4 and d0:
        x[4, :d0] *= 50.0               # the front segment's absmax / 127 exceeds (hi - lo) / 255: it sets the step
    if rows > 5:                        # ties: (x - lo) / s = n + 0.5 exactly (lo = 0, hi = 255 -> s = 1)
        x[5, d0:] = 0.0
        x[5, d0:d0 + 8] = [255.0, 0.5, 1.5, 2.5, 126.5, 127.5, 254.5, 3.0]
        x[5, :d0] = 0.0
    x = bf16_round(x)
    xd = dev(x, torch.bfloat16)
    q = torch.empty(rows, K, dtype=torch.int8, device="cuda")
    sc = torch.empty(rows, dtype=torch.float32, device="cuda")
    of = torch.empty(rows, dtype=torch.float32, device="cuda")
    L.check(lib.fmi_quantize_rows_i8_asym(_p(xd), rows, K, d0, _p(q), _p(sc), _p(of), None))
    torch.cuda.synchronize()
    rq, rs, ro = orc.quantize_rows_i8_asym(x, d0)
    np.testing.assert_array_equal(sc.cpu().numpy(), rs)
    np.testing.assert_array_equal(of.cpu().numpy(), ro)
    np.testing.assert_array_equal(q.cpu().numpy(), rq)
    if rows > 5:
        assert rq[5, d0:d0 + 8].tolist() == [127, -128, -126, -126, -2, 0, 126, -125]  # round half to even, then - 128
    # the grid does what it is for: on post-GELU rows the reconstruction is about twice as close as the symmetric recipe's
    deq = rq.astype(np.float32) * rs[:, None]
    deq[:, d0:] += ro[:, None]
    sq, ss = orc.quantize_rows_i8(x)
    body = slice(6, None) if rows > 6 else slice(0, 1)
    e_asym, e_sym = rel_l2(deq[body], x[body]), rel_l2(sq[body].astype(np.float32) * ss[body, None], x[body])
    print(f"post-GELU rows {rows}x{K} (front segment {d0}): reconstruction rel-L2 {e_asym:.2e} on the offset grid, {e_sym:.2e} symmetric")
    if d0 == 0 and rows > 6:
        assert e_asym <= 0.65 * e_sym
    assert lib.fmi_quantize_rows_i8_asym(_p(xd), rows, K, K, _p(q), _p(sc), _p(of), None) < 0      # no offset segment left
    assert lib.fmi_quantize_rows_i8_asym(_p(xd), rows, K, 4, _p(q), _p(sc), _p(of), None) < 0      # d0 % 8


@pytest.mark.parametrize("M,N,K,d0,epi", [(300, 384, 512, 128, 0), (257, 512, 15360, 3072, 0), (1024, 3072, 12288, 0, 0), (64, 260, 1280, 0, 1)])
def test_gemm_i8_asym_matches_oracle(env, M, N, K, d0, epi):
    """quantise (offset form) -> column sums of the weight codes -> int8 GEMM with the offset term, against orc_linear_i8_asym on the same bf16
    inputs: the integer sums are exact on both sides, so the two differ by the bf16 rounding of the output only."""
    torch, L, lib, orc = env["torch"], env["L"], env["lib"], env["orc"]
    rng = np.random.default_rng(M + N + K + d0)
    x = bf16_round(_post_gelu_rows(rng, M, K, d0))
    w = bf16_round((rng.standard_normal((N, K)) * 0.05).astype(np.float32))
    b = bf16_round(rng.standard_normal(N).astype(np.float32))
    xd, wd, bd = dev(x, torch.bfloat16), dev(w, torch.bfloat16), dev(b, torch.bfloat16)
    wq, ws = gpu_quantize(env, wd)
    wsum = torch.empty(N, dtype=torch.float32, device="cuda")
    L.check(lib.fmi_rowsum_i8(_p(wq), _p(ws), N, K, d0, _p(wsum), None))
    xq = torch.empty(M, K, dtype=torch.int8, device="cuda")
    xs, xo = torch.empty(M, dtype=torch.float32, device="cuda"), torch.empty(M, dtype=torch.float32, device="cuda")
    L.check(lib.fmi_quantize_rows_i8_asym(_p(xd), M, K, d0, _p(xq), _p(xs), _p(xo), None))
    y = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
    L.check(lib.fmi_gemm_i8_asym(_p(xq), _p(xs), _p(xo), _p(wq), _p(ws), _p(wsum), _p(bd), _p(y), M, N, K, epi, None))
    torch.cuda.synchronize()
    np.testing.assert_array_equal(wsum.cpu().numpy(), (ws.cpu().numpy() * wq.cpu().numpy()[:, d0:].astype(np.int64).sum(1).astype(np.float32)))
    ref, sym, f32 = orc.linear_i8_asym(x, w, b, d0), orc.linear_i8(x, w, b), orc.linear(x, w, b)
    if epi == 1:
        ref, sym, f32 = orc.gelu(ref), orc.gelu(sym), orc.gelu(f32)
    err, noise, noise_sym = rel_l2(host(y), ref), rel_l2(ref - b, f32 - b), rel_l2(sym - b, f32 - b)
    print(f"gemm_i8_asym {M}x{N}x{K} d0={d0} epi={epi}: rel-L2 vs its oracle {err:.2e}; recipe vs f32 linear {noise:.2e} (all-symmetric rows: {noise_sym:.2e})")
    assert err <= 2e-3
    if epi == 0:
        assert noise < noise_sym  # what the offset grid is for
    assert lib.fmi_gemm_i8_asym(_p(xq), _p(xs), None, _p(wq), _p(ws), _p(wsum), _p(bd), _p(y), M, N, K, epi, None) < 0


def test_linear_i8_rejects_bad_shapes(env):
    torch, lib = env["torch"], env["lib"]
    x = torch.zeros(64, 256, dtype=torch.bfloat16, device="cuda")
    q = torch.zeros(256, 256, dtype=torch.int8, device="cuda")
    s = torch.ones(256, dtype=torch.float32, device="cuda")
    y = torch.zeros(64, 256, dtype=torch.bfloat16, device="cuda")
    assert lib.fmi_linear_i8(_p(x), _p(q), _p(s), None, _p(y), 64, 128, 256, 0, None) < 0   # N <= 128: no 8-bit tile shape
    assert lib.fmi_linear_i8(_p(x), _p(q), _p(s), None, _p(y), 64, 256, 192, 0, None) < 0   # K % 128
    assert lib.fmi_linear_i8(_p(x), _p(q), None, None, _p(y), 64, 256, 256, 0, None) < 0
    assert lib.fmi_linear_i8(_p(x), _p(q), _p(s), None, _p(y), 0, 256, 256, 0, None) == 0
    assert lib.fmi_quantize_rows_i8(_p(x), 4, 12, _p(q), _p(s), None) < 0                    # K % 8


MASKS = [0x33, 0x3f, 0x15, 0x0c, 0x01, 0x10]  # the default, every block linear, the LayerNorm-fed ones, the double blocks' MLP alone, and q|k|v of ONE block kind
# (0x01 / 0x10: e4m3 attention operands in the double blocks only / the single blocks only — ADVICE r4: the oracle gates them per block kind like the library)


@pytest.fixture(scope="module")
def models(env):
    d, orc = env["d"], env["orc"]
    sd = d.synth.flux_state_dict_numpy(SMALL_FLUX, seed=0)
    out = dict(sd=sd)
    for mask in MASKS:
        g = d.FluxModel(SMALL_FLUX)
        g.load_state_dict(sd)
        g.quantize_int8(mask)
        out[mask] = g
    out["gb"] = d.FluxModel(SMALL_FLUX)
    out["gb"].load_state_dict(sd)
    out["o8"] = orc.Flux(SMALL_FLUX)
    out["o8"].load(sd)
    out["of"] = orc.Flux(SMALL_FLUX)
    out["of"].load(sd)
    return out


@pytest.mark.parametrize("mask", MASKS)
@pytest.mark.parametrize("B,S_hw,T", [(1, (8, 12), 40), (2, (6, 6), 64), (1, (16, 16), 77), (2, (8, 16), 48), (1, (8, 8), 32)])
def test_flux_forward_int8_matches_int8_oracle(env, models, B, S_hw, T, mask):
    torch = env["torch"]
    img, ids, txt, txt_ids, y = flux_inputs(SMALL_FLUX, B, S_hw, T)
    t = np.linspace(0.9, 0.4, B).astype(np.float32)
    g = np.full(B, 3.5, np.float32)
    args = (dev(img), dev(ids), dev(txt, torch.bfloat16), dev(txt_ids), dev(t), dev(y), dev(g))
    got = host(models[mask].forward(*args))
    aligned = (S_hw[0] * S_hw[1]) % 16 == 0 and T % 16 == 0  # then every block takes the fused q|k|v epilogue and QK^T runs on e4m3 operands
    models["o8"].set_int8(True, mask, attention=aligned)  # (the oracle gates the attention operands on the mask per block kind, like the library)
    ref8 = models["o8"].forward(img, ids, txt, txt_ids, t, y, g)
    models["o8"].set_int8(False)
    ref = models["of"].forward(img, ids, txt, txt_ids, t, y, g)
    gotb = host(models["gb"].forward(*args))
    assert np.isfinite(got).all()
    e8, ef, eb = rel_l2(got, ref8), rel_l2(got, ref), rel_l2(gotb, ref)
    print(f"int8 forward mask 0x{mask:02x} B={B} S={S_hw} T={T}: vs int8 oracle {e8:.3e}; vs f32 oracle {ef:.3e} (bf16 path: {eb:.3e}); oracle int8 vs f32 {rel_l2(ref8, ref):.3e}")
    assert e8 <= 1e-2   # the bf16 path's bar against its oracle
    assert ef <= 2e-2
    assert not np.array_equal(got, gotb)  # the mode is on


def test_flux_denoise_int8(env, models):
    torch, d = env["torch"], env["d"]
    B, S_hw, T, steps = 1, (8, 8), 32, 4
    img, ids, txt, txt_ids, y = flux_inputs(SMALL_FLUX, B, S_hw, T, seed=7)
    g = np.full(B, 3.5, np.float32)
    sched = d.SchedulerConfig()
    ts = sched.get_timesteps(steps, sched.calculate_shift(S_hw[0] * S_hw[1]))
    models["o8"].set_int8(True, 0x33, attention=True)   # (8, 8) / 32: aligned -> e4m3 QK^T
    ref8 = models["o8"].denoise(img, ids, txt, txt_ids, y, g, ts)
    models["o8"].set_int8(False)
    args = (dev(img), dev(ids), dev(txt, torch.bfloat16), dev(txt_ids), dev(y), dev(g), ts)
    got = host(models[0x33].denoise(*args))
    again = host(models[0x33].denoise(*args))
    np.testing.assert_array_equal(got, again)   # deterministic
    err = rel_l2(got, ref8)
    print(f"int8 denoise {steps} steps: rel-L2 vs int8 oracle {err:.3e}")
    assert err <= 3e-2


def test_e4m3_attention_operands_in_bf16_mode_opt_in(env, models):
    """fmi_flux_set_fp8_attention(m, 2): bf16 block linears, q and k as e4m3 in front of QK^T (16-aligned token counts: every block).  Against the
    oracle with NO linear quantised and its attention on e4m3 operands (the same static scales); off again = the bf16 result bit for bit."""
    torch, d = env["torch"], env["d"]
    gb, o8, of = models["gb"], models["o8"], models["of"]
    B, S_hw, T = 2, (8, 16), 48
    img, ids, txt, txt_ids, y = flux_inputs(SMALL_FLUX, B, S_hw, T)
    t, g = np.linspace(0.9, 0.4, B).astype(np.float32), np.full(B, 3.5, np.float32)
    args = (dev(img), dev(ids), dev(txt, torch.bfloat16), dev(txt_ids), dev(t), dev(y), dev(g))
    base = host(gb.forward(*args))
    gb.set_fp8_attention(2)
    try:
        got = host(gb.forward(*args))
        name = "transformer_blocks.0.attn.norm_q.weight"   # a reload of a QkNorm weight must refresh the static scales (here: the same values)
        gb.set_tensor(name, models["sd"][name])
        np.testing.assert_array_equal(host(gb.forward(*args)), got)
    finally:
        gb.set_fp8_attention(1)
    np.testing.assert_array_equal(host(gb.forward(*args)), base)
    o8.set_int8(True, 0, attention=True, attention_everywhere=True)
    ref8 = o8.forward(img, ids, txt, txt_ids, t, y, g)
    o8.set_int8(False)
    ref = of.forward(img, ids, txt, txt_ids, t, y, g)
    e8, ef, eb = rel_l2(got, ref8), rel_l2(got, ref), rel_l2(base, ref)
    print(f"bf16 linears + e4m3 QK^T operands: vs its oracle {e8:.3e}; vs f32 oracle {ef:.3e} (bf16 attention operands: {eb:.3e}); oracle e4m3-QK vs f32 {rel_l2(ref8, ref):.3e}")
    assert not np.array_equal(got, base) and e8 <= 1e-2 and ef <= 2e-2
    with pytest.raises(d.FmiError):
        gb.set_fp8_attention(3)


def test_int8_mode_guards(env, models):
    d = env["d"]
    sd = models["sd"]
    m = d.FluxModel(SMALL_FLUX)
    with pytest.raises(d.FmiError):
        m.quantize_int8()             # tensors missing
    m.load_state_dict(sd)
    with pytest.raises(d.FmiError):
        m.quantize_int8(0)            # no linear named
    with pytest.raises(d.FmiError):
        m.quantize_int8(0x40)         # a bit that names nothing
    m.quantize_int8()
    m.quantize_int8(d.flux.INT8_DEFAULT_MASK)   # idempotent for the same mask
    with pytest.raises(d.FmiError):
        m.quantize_int8(0x3f)         # another mask: the codes are already chosen
    with pytest.raises(d.FmiError):
        m.quantize_fp8()              # one 8-bit form per handle
    name = "transformer_blocks.0.attn.to_q.weight"
    with pytest.raises(d.FmiError):
        m.set_tensor(name, sd[name])  # weights are frozen once quantised
    m.close()
    m = d.FluxModel(SMALL_FLUX)
    m.load_state_dict(sd)
    m.quantize_fp8()
    with pytest.raises(d.FmiError):
        m.quantize_int8()
    m.close()


def test_pipeline_int8_end_to_end(env):
    """ModelDType.I8 through the Pipeline: prompts -> u8 images, deterministic, close to the bf16 pipeline's."""
    torch, d = env["torch"], env["d"]
    from tests.util import SMALL_VAE
    cfg = dict(SMALL_FLUX)
    params = d.DiffusionGenerationParams(height=128, width=128, num_steps=3, guidance_scale=3.5)
    p8 = d.Pipeline.load(d.ModelSource.Synthetic("dev", seed=4, flux_cfg=cfg, vae_cfg=SMALL_VAE), dtype=d.ModelDType.I8)
    pb = d.Pipeline.load(d.ModelSource.Synthetic("dev", seed=4, flux_cfg=cfg, vae_cfg=SMALL_VAE))
    a = p8.generate_tensor(["a red fox"], params, seed=9).cpu().numpy()
    again = p8.generate_tensor(["a red fox"], params, seed=9).cpu().numpy()
    b = pb.generate_tensor(["a red fox"], params, seed=9).cpu().numpy()
    np.testing.assert_array_equal(a, again)
    assert a.dtype == np.uint8 and a.shape == b.shape
    diff = np.abs(a.astype(np.int32) - b.astype(np.int32))
    print(f"int8 pipeline vs bf16 pipeline: max |du8| {int(diff.max())}, mean {float(diff.mean()):.3f}, differing {float((diff > 0).mean()):.3f}")
    assert float((diff <= 8).mean()) >= 0.99


# ---------------------------------------------------------------------------------------------------------------- round 6: the smoothed recipe
@pytest.mark.parametrize("rows,K,d0", [(5, 64, -1), (64, 3072, -1), (300, 15360, 3072), (33, 12288, 0), (7, 4096, -1)])
def test_quantize_rows_i8_scaled_bit_exact(env, rows, K, d0):
    """fmi_quantize_rows_i8_scaled = the two row recipes on x[r, k] * col_scale[k] (one f32 product per element): codes, steps and offsets bit for bit
    against the oracle's quantisers applied to the same f32 products — with a col_scale that spans 2^-10 .. 2^10 like the smoothing factors do, and rows
    that carry outlier channels (what the factors are for)."""
    torch, L, lib, orc = env["torch"], env["L"], env["lib"], env["orc"]
    rng = np.random.default_rng(rows + K + max(d0, 0) + 17)
    x = _post_gelu_rows(rng, rows, K, max(d0, 0)) if d0 >= 0 else rng.standard_normal((rows, K)).astype(np.float32)
    out_ch = rng.choice(K, 6, replace=False)
    x[:, out_ch] *= 60.0
    x = bf16_round(x)
    vec = (2.0 ** rng.uniform(-10, 10, K)).astype(np.float32)
    vec[out_ch] = (1.0 / 60.0)
    xd, vd = dev(x, torch.bfloat16), dev(vec)
    q = torch.empty(rows, K, dtype=torch.int8, device="cuda")
    sc = torch.full((rows,), float("nan"), device="cuda")
    of = torch.full((rows,), float("nan"), device="cuda")
    L.check(lib.fmi_quantize_rows_i8_scaled(_p(xd), rows, K, d0, _p(vd), _p(q), _p(sc), _p(of), None))
    torch.cuda.synchronize()
    xs = (x * vec[None, :]).astype(np.float32)
    if d0 >= 0:
        rq, rs, ro = orc.quantize_rows_i8_asym(xs, d0)
        np.testing.assert_array_equal(of.cpu().numpy(), ro)
    else:
        rq, rs = orc.quantize_rows_i8(xs)
    np.testing.assert_array_equal(sc.cpu().numpy(), rs)
    np.testing.assert_array_equal(q.cpu().numpy(), rq)
    # a vector of ones = the plain entry points, bit for bit
    ones = dev(np.ones(K, np.float32))
    q1, s1 = torch.empty_like(q), torch.empty_like(sc)
    L.check(lib.fmi_quantize_rows_i8_scaled(_p(xd), rows, K, d0, _p(ones), _p(q1), _p(s1), _p(of), None))
    q2, s2, o2 = torch.empty_like(q), torch.empty_like(sc), torch.empty_like(sc)
    if d0 >= 0:
        L.check(lib.fmi_quantize_rows_i8_asym(_p(xd), rows, K, d0, _p(q2), _p(s2), _p(o2), None))
    else:
        L.check(lib.fmi_quantize_rows_i8(_p(xd), rows, K, _p(q2), _p(s2), None))
    torch.cuda.synchronize()
    assert torch.equal(q1, q2) and torch.equal(s1, s2)
    assert lib.fmi_quantize_rows_i8_scaled(_p(xd), rows, K, d0, None, _p(q), _p(sc), _p(of), None) < 0


def test_col_absmax(env):
    torch, L, lib = env["torch"], env["L"], env["lib"]
    rng = np.random.default_rng(5)
    for rows, K, ld in ((1, 8, 8), (300, 3072, 3072), (4608, 15360, 21504), (70000, 64, 72)):
        x = bf16_round(rng.standard_normal((rows, ld)).astype(np.float32) * (10.0 ** rng.uniform(-3, 3, (1, ld))).astype(np.float32))
        xd = dev(x, torch.bfloat16)
        am = torch.zeros(K, device="cuda")
        L.check(lib.fmi_col_absmax(_p(xd), rows // 2, K, ld, _p(am), None))                      # a running maximum: two calls over the two halves
        L.check(lib.fmi_col_absmax(_p(xd[rows // 2:]), rows - rows // 2, K, ld, _p(am), None))
        torch.cuda.synchronize()
        np.testing.assert_array_equal(am.cpu().numpy(), np.abs(x[:, :K]).max(0))
    assert lib.fmi_col_absmax(_p(xd), 4, 64, 60, _p(am), None) < 0


def test_flux_forward_int8_smoothed_on_outlier_channels(env):
    """The smoothed int8 recipe end to end at a small config WITH the outlier-channel profile (synth.apply_outlier_profile: 12 hidden channels whose AdaLN
    (1 + scale) is 30-100, massive residual channels, x 8 QkNorm dimensions): fmi_flux_calibrate_int8 + 3 evaluations + fmi_flux_quantize_int8 against
    (i) the f32 oracle — inside the 8-bit bar where the unsmoothed recipe is far outside —, (ii) the oracle's restatement of the smoothed recipe
    (orc_flux_set_calibration on the same calibration inputs), statistically as for the other 8-bit recipes, (iii) calibration leaves the bf16 results
    untouched, (iv) the guards."""
    torch, d, orc = env["torch"], env["d"], env["orc"]
    cfg = dict(SMALL_FLUX, num_attention_heads=4)  # D = 512: room for 12 outlier channels (2.3 %)
    D = 512
    sd = d.synth.flux_state_dict_numpy(cfg, seed=21)
    sd = {k: bf16_round(d.synth.apply_outlier_profile(k, v.copy(), D)) for k, v in sd.items()}
    B, S_hw, T = 1, (16, 16), 64
    img, ids, txt, txt_ids, y = flux_inputs(cfg, B, S_hw, T, seed=3)
    g = np.full(B, 3.5, np.float32)
    cal_ts = (0.95, 0.6, 0.2)
    t_eval = np.array([0.45], np.float32)
    mk = lambda tt: (dev(img), dev(ids), dev(txt, torch.bfloat16), dev(txt_ids), dev(np.array([tt], np.float32)), dev(y), dev(g))
    c_img, _, c_txt, _, c_y = flux_inputs(cfg, B, S_hw, T, seed=4)  # the calibration sees another sample than the evaluation
    mkc = lambda tt: (dev(c_img), dev(ids), dev(c_txt, torch.bfloat16), dev(txt_ids), dev(np.array([tt], np.float32)), dev(c_y), dev(g))
    of_ = orc.Flux(cfg)
    of_.load(sd)
    ref = of_.forward(img, ids, txt, txt_ids, t_eval, y, g)
    rows = {}
    for smooth in (False, True):
        m = d.FluxModel(cfg)
        try:
            m.load_state_dict(sd)
            base = host(m.forward(*mk(float(t_eval[0]))))
            if smooth:
                m.calibrate_int8(True)
                for tt in cal_ts:
                    m.forward(*mkc(tt))
                np.testing.assert_array_equal(host(m.forward(*mk(float(t_eval[0])))), base)  # recording does not change results
            m.quantize_int8()
            got = host(m.forward(*mk(float(t_eval[0]))))
            rows[smooth] = (rel_l2(got, ref), rel_l2(base, ref), got)
            if smooth:
                with pytest.raises(env["L"].FmiError):
                    m.calibrate_int8(True)  # the model already holds an 8-bit form
        finally:
            m.close()
    # the oracle's restatement: calibrate on the same evaluations (f32), then the int8 recipe with the same mask and e4m3 q / k
    o8 = orc.Flux(cfg)
    o8.load(sd)
    o8.set_calibration(1)
    for tt in cal_ts:
        o8.forward(c_img, ids, c_txt, txt_ids, np.array([tt], np.float32), c_y, g)
    o8.set_calibration(0)
    o8.set_int8(True, d.flux.INT8_DEFAULT_MASK, attention=True)
    ref8 = o8.forward(img, ids, txt, txt_ids, t_eval, y, g)
    o8.set_int8(False)
    o8.set_calibration(-1)
    o8.set_int8(True, d.flux.INT8_DEFAULT_MASK, attention=True)
    ref8_plain = o8.forward(img, ids, txt, txt_ids, t_eval, y, g)
    o8.set_int8(False)
    e_plain, e_bf16, _ = rows[False]
    e_sm, _, got_sm = rows[True]
    noise, noise_plain = rel_l2(ref8, ref), rel_l2(ref8_plain, ref)
    print(f"outlier-profile small model (D=512, 2 + 2 blocks): bf16 {e_bf16:.3e}; int8 unsmoothed {e_plain:.3e} (its oracle: {noise_plain:.3e}); "
          f"int8 SMOOTHED {e_sm:.3e} vs f32, {rel_l2(got_sm, ref8):.3e} vs the smoothed-recipe oracle (recipe noise {noise:.3e})")
    assert np.isfinite(got_sm).all() and not np.array_equal(got_sm, rows[False][2])
    # at 2 + 2 blocks the gates (~1e-2) keep every recipe's noise far below the bf16 path's own 1.6e-3, so the GPU rows can only be held to the bf16 bars; what the
    # smoothing buys shows in the ORACLE's recipe noise (f32 everywhere else) — and, at full depth, in tests/test_gpu_outlier_stats.py (1.9e-1 -> 2.4e-2)
    assert e_sm <= 1e-2 and rel_l2(got_sm, ref8) <= 1e-2
    assert noise <= 0.6 * noise_plain
    m2 = d.FluxModel(cfg)
    try:
        m2.load_state_dict(sd)
        m2.calibrate_int8(True)
        with pytest.raises(env["L"].FmiError):
            m2.quantize_int8()  # calibration on, no evaluation yet
        m2.calibrate_int8(False)
        m2.quantize_int8()      # dropped: the unsmoothed recipe, as before
        np.testing.assert_array_equal(host(m2.forward(*mk(float(t_eval[0])))), rows[False][2])
    finally:
        m2.close()
