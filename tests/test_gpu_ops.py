"""Operator-level parity: HIP kernels (through the C-ABI) vs the CPU oracle on identical inputs.

Tolerances (bf16 storage, f32 accumulate, vs the f32 oracle): rel-L2 <= 4e-3 for a single
GEMM / attention / norm; bit-exact for integer / LUT work (bnb dequant, pack/unpack, u8).
"""
import ctypes as C

import numpy as np
import pytest

from tests.util import bf16_round, dev, host, rel_l2

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env():
    import torch
    import diffusion_rs_amd as d
    from diffusion_rs_amd import _lib as L
    from oracle import oracle as orc
    lib = L.load()
    L.check(lib.fmi_init(0))
    return dict(torch=torch, d=d, L=L, lib=lib, orc=orc)


def _p(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


@pytest.mark.parametrize("M,N,K,epi,bias", [(256, 256, 64, 0, True), (300, 192, 256, 0, True), (1, 64, 128, 0, False), (515, 3072, 512, 1, True),
                                            (77, 64, 3072, 0, True), (1024, 520, 128, 2, True), (33, 4, 64, 0, True)])
def test_linear_bf16(env, M, N, K, epi, bias):
    torch, L, lib, orc = env["torch"], env["L"], env["lib"], env["orc"]
    rng = np.random.default_rng(M * 7 + N)
    x = bf16_round(rng.standard_normal((M, K)).astype(np.float32))
    w = bf16_round((rng.standard_normal((N, K)) / np.sqrt(K)).astype(np.float32))
    b = bf16_round(rng.standard_normal(N).astype(np.float32)) if bias else None
    ref = orc.linear(x, w, b)
    if epi == 1:
        ref = orc.gelu(ref)
    elif epi == 2:
        ref = orc.silu(ref)
    xd, wd = dev(x, torch.bfloat16), dev(w, torch.bfloat16)
    bd = dev(b, torch.bfloat16) if bias else None
    y = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device="cuda")
    L.check(lib.fmi_linear_bf16(_p(xd), _p(wd), _p(bd), _p(y), M, N, K, epi, None))
    torch.cuda.synchronize()
    got = host(y)
    assert np.isfinite(got).all()
    assert rel_l2(got, ref) <= 4e-3


@pytest.mark.parametrize("tiles_m,N", [(9, 768), (11, 512), (18, 1024), (29, 512), (33, 384), (13, 128)])
def test_linear_bf16_tile_order_band_heights(env, tiles_m, N):
    """launch_gemm picks the band height of the tile order per problem (pick_tile_band: 18 tile rows -> 6 + 6 + 6, 29 -> 6 x 4 + 5, 33 -> ..., a
    ragged last band, a partial last tile row): whatever it picks must be a bijection of the tiles — every 256 x 256 (or 256 x 128) tile of the
    output is written exactly once with the right rows and columns.  Rows carry their index in the data, so a swapped or duplicated tile
    cannot hide; compared per tile with the oracle's linear."""
    torch, L, lib, orc = env["torch"], env["L"], env["lib"], env["orc"]
    M, K = tiles_m * 256 - 37, 64  # the last tile row is partial
    rng = np.random.default_rng(tiles_m)
    x = bf16_round((rng.standard_normal((M, K)) + (np.arange(M)[:, None] % 17) * 0.25).astype(np.float32))
    w = bf16_round((rng.standard_normal((N, K)) / np.sqrt(K) + (np.arange(N)[:, None] % 5) * 0.125).astype(np.float32))
    ref = orc.linear(x, w, None)
    xd, wd = dev(x, torch.bfloat16), dev(w, torch.bfloat16)
    y = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device="cuda")
    L.check(lib.fmi_linear_bf16(_p(xd), _p(wd), None, _p(y), M, N, K, 0, None))
    torch.cuda.synchronize()
    got = host(y)
    assert np.isfinite(got).all()
    bn = 128 if N <= 128 else 256
    for tm in range(tiles_m):
        for tn in range((N + bn - 1) // bn):
            g, r = got[tm * 256:(tm + 1) * 256, tn * bn:(tn + 1) * bn], ref[tm * 256:(tm + 1) * 256, tn * bn:(tn + 1) * bn]
            assert rel_l2(g, r) <= 4e-3, (tm, tn)


@pytest.mark.parametrize("epi,M,N", [(1, 256, 64), (2, 256, 64), (1, 512, 256)])  # 256-wide: the ping-pong kernel's packed-pair GELU epilogue
def test_activation_epilogues_over_the_whole_range(env, epi, M, N):
    """GELU (v - v / (2^z + 1), `gelu_tanh` / `gelu_tanh4`) and SiLU (`v_rcp_f32` form) through an identity weight: exact zeros,
    the saturating tails on both sides, tiny and huge magnitudes — element by element against the oracle (core/op.rs:539-582,
    699-721, f32 arm) within one bf16 ulp of the result, finite for every finite input."""
    torch, L, lib, orc = env["torch"], env["L"], env["lib"], env["orc"]
    K = N
    special = np.array([0.0, -0.0, 1e-30, -1e-30, 1e-3, -1e-3, 0.5, -0.5, 1, -1, 2.5, -2.5, 4, -4, 6, -6, 9, -9, 17, -17, 40, -40, 100, -100,
                        1e4, -1e4, 3e38, -3e38], dtype=np.float32)
    rng = np.random.default_rng(epi)
    x = bf16_round(np.concatenate([np.tile(special, (M * K) // (2 * special.size)),
                                   (rng.standard_normal(M * K - special.size * ((M * K) // (2 * special.size))) * 3).astype(np.float32)]).reshape(M, K))
    w = np.eye(N, K, dtype=np.float32)
    ref = orc.gelu(x) if epi == 1 else orc.silu(x)
    xd, wd = dev(x, torch.bfloat16), dev(w, torch.bfloat16)
    y = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device="cuda")
    L.check(lib.fmi_linear_bf16(_p(xd), _p(wd), None, _p(y), M, N, K, epi, None))
    torch.cuda.synchronize()
    got = host(y)
    assert np.isfinite(got).all()
    tol = np.maximum(np.abs(ref) * 2.0 ** -7, 1e-6)  # one bf16 ulp (8 significant bits) of the reference, absolute floor for the vanishing tail
    bad = np.abs(got - ref) > tol
    assert not bad.any(), (x[bad][:8], got[bad][:8], ref[bad][:8])


def test_linear_rejects_bad_k(env):
    torch, L, lib = env["torch"], env["L"], env["lib"]
    x = torch.zeros((8, 100), dtype=torch.bfloat16, device="cuda")
    rc = lib.fmi_linear_bf16(_p(x), _p(x), None, _p(x), 8, 8, 100, 0, None)
    assert rc == -1 and b"multiple of 64" in lib.fmi_last_error()


@pytest.mark.parametrize("B,H,Lq,Lk", [(1, 2, 64, 64), (1, 2, 77, 77), (2, 3, 300, 300), (1, 1, 600, 600), (1, 2, 256, 1000)])
@pytest.mark.parametrize("token_major", [0, 1])
def test_sdpa_bf16(env, B, H, Lq, Lk, token_major):
    torch, L, lib, orc = env["torch"], env["L"], env["lib"], env["orc"]
    rng = np.random.default_rng(Lq + Lk + H)
    q = bf16_round(rng.standard_normal((B, H, Lq, 128)).astype(np.float32))
    k = bf16_round(rng.standard_normal((B, H, Lk, 128)).astype(np.float32))
    v = bf16_round(rng.standard_normal((B, H, Lk, 128)).astype(np.float32))
    scale = 1.0 / np.sqrt(128.0)
    ref = orc.sdpa(q, k, v, scale)
    o = torch.full((B, Lq, H * 128) if token_major else (B, H, Lq, 128), float("nan"), dtype=torch.bfloat16, device="cuda")
    qd, kd, vd = (dev(a, torch.bfloat16) for a in (q, k, v))
    L.check(lib.fmi_sdpa_bf16(_p(qd), _p(kd), _p(vd), _p(o), B, H, Lq, Lk, 128, scale, token_major, None))
    torch.cuda.synchronize()
    got = host(o)
    if token_major:
        got = got.reshape(B, Lq, H, 128).transpose(0, 2, 1, 3)
    assert np.isfinite(got).all()
    assert rel_l2(got, ref) <= 6e-3
    assert np.abs(got - ref).max() <= 3e-2


def test_sdpa_forced_rescale(env):
    """cdna guide rule 26: force the deferred-rescale branch — one key row spikes against one query
    row late in the sequence, so the running max jumps by far more than the threshold mid-stream."""
    torch, L, lib, orc = env["torch"], env["L"], env["lib"], env["orc"]
    rng = np.random.default_rng(5)
    B, H, Ln = 1, 1, 512
    q = bf16_round(0.3 * rng.standard_normal((B, H, Ln, 128)).astype(np.float32))
    k = bf16_round(0.3 * rng.standard_normal((B, H, Ln, 128)).astype(np.float32))
    v = bf16_round(rng.standard_normal((B, H, Ln, 128)).astype(np.float32))
    k[0, 0, 400] = bf16_round(q[0, 0, 17] * 40.0)  # raw q.k ~ 40*|q|^2 >> every other score of row 17
    scale = 1.0 / np.sqrt(128.0)
    ref = orc.sdpa(q, k, v, scale)
    o = torch.empty((B, H, Ln, 128), dtype=torch.bfloat16, device="cuda")
    qd, kd, vd = (dev(a, torch.bfloat16) for a in (q, k, v))
    L.check(lib.fmi_sdpa_bf16(_p(qd), _p(kd), _p(vd), _p(o), B, H, Ln, Ln, 128, scale, 0, None))
    torch.cuda.synchronize()
    got = host(o)
    assert np.abs(got - ref).max() <= 3e-2
    assert np.abs(got[0, 0, 17] - ref[0, 0, 17]).max() <= 3e-2


@pytest.mark.parametrize("rows,D", [(5, 256), (300, 3072), (64, 512)])
def test_layernorm_mod(env, rows, D):
    torch, L, lib, orc = env["torch"], env["L"], env["lib"], env["orc"]
    rng = np.random.default_rng(rows)
    x = (rng.standard_normal((rows, D)) * 3 + 0.5).astype(np.float32)
    sc = (0.1 * rng.standard_normal(D)).astype(np.float32)
    sh = (0.1 * rng.standard_normal(D)).astype(np.float32)
    ref = orc.layer_norm(x, None, None, 1e-6) * (sc + 1.0) + sh
    out = torch.empty((rows, D), dtype=torch.bfloat16, device="cuda")
    xd, scd, shd = dev(x), dev(sc), dev(sh)
    L.check(lib.fmi_layernorm_mod(_p(xd), _p(scd), _p(shd), _p(out), rows, D, 1e-6, None))
    torch.cuda.synchronize()
    assert rel_l2(host(out), ref) <= 3e-3
    # plain LN (no modulation)
    L.check(lib.fmi_layernorm_mod(_p(xd), None, None, _p(out), rows, D, 1e-6, None))
    torch.cuda.synchronize()
    assert rel_l2(host(out), orc.layer_norm(x, None, None, 1e-6)) <= 3e-3


_TD = {"f32": "float32", "f16": "float16", "bf16": "bfloat16"}


@pytest.mark.parametrize("odt", ["f32", "f16", "bf16"])
@pytest.mark.parametrize("qt", ["nf4", "fp4"])
@pytest.mark.parametrize("n,blocksize", [(64 * 64, 64), (3072 * 40, 64), (1000, 128), (4096 * 3 + 2, 4096), (31, 64), (3072 * 1024, 64), (2048 * 1024 + 8, 128)])
def test_dequant_4bit_bit_exact(env, odt, qt, n, blocksize):
    """Integer / LUT work: bit-exact vs the oracle (CUDA-kernel semantics of dequant.cu).  The two sizes above 2^20 elements
    take the streaming kernel in bf16 (the per-call expansion of the denoise loop), the others the general one."""
    torch, lib, orc = env["torch"], env["lib"], env["orc"]
    rng = np.random.default_rng(n)
    A = rng.integers(0, 256, (n + 1) // 2, dtype=np.uint8)
    absmax = (rng.random((n + blocksize - 1) // blocksize) * 3 + 0.01).astype(np.float32)
    ref = orc.dequantize_blockwise(None, A, absmax, blocksize, n, qt, odt)
    out = torch.zeros(n, dtype=getattr(torch, _TD[odt]), device="cuda")
    Ad, amd = dev(A), dev(absmax)
    getattr(lib, f"dequantize_blockwise_{odt}_{qt}")(None, _p(Ad), _p(amd), _p(out), blocksize, n, None)
    torch.cuda.synchronize()
    np.testing.assert_array_equal(host(out).view(np.uint32), ref.view(np.uint32))


@pytest.mark.parametrize("odt", ["f32", "f16", "bf16"])
def test_dequant_int8_bit_exact(env, odt):
    torch, lib, orc = env["torch"], env["lib"], env["orc"]
    rng = np.random.default_rng(3)
    n, blocksize = 5000, 256
    code = np.sort(rng.standard_normal(256)).astype(np.float32)
    A = rng.integers(0, 256, n, dtype=np.uint8)
    absmax = (rng.random((n + blocksize - 1) // blocksize) + 0.1).astype(np.float32)
    ref = orc.dequantize_blockwise(code, A, absmax, blocksize, n, "int8", odt)
    out = torch.zeros(n, dtype=getattr(torch, _TD[odt]), device="cuda")
    cd, Ad, amd = dev(code), dev(A), dev(absmax)
    getattr(lib, f"dequantize_blockwise_{odt}_int8")(_p(cd), _p(Ad), _p(amd), _p(out), blocksize, n, None)
    torch.cuda.synchronize()
    np.testing.assert_array_equal(host(out).view(np.uint32), ref.view(np.uint32))
    # LLM.int8 SCB path (the second shape is large enough for the streaming bf16 kernel of the denoise loop)
    for row, col in ((37, 100), (1024, 3072)):
        w = rng.integers(-128, 128, row * col, dtype=np.int8)
        scb = (rng.random(row) * 2).astype(np.float32)
        ref = orc.dequantize_8bit(w, scb, row, col, odt)
        out = torch.zeros(row * col, dtype=getattr(torch, _TD[odt]), device="cuda")
        wd, sd = dev(w), dev(scb)
        getattr(lib, f"dequantize_8bit_kernel_{odt}")(_p(wd), _p(sd), _p(out), row, col, row * col)
        torch.cuda.synchronize()
        np.testing.assert_array_equal(host(out).view(np.uint32), ref.view(np.uint32))


@pytest.mark.parametrize("qt", ["nf4", "fp4"])
@pytest.mark.parametrize("M,N,K,blocksize", [(300, 256, 256, 64), (64, 192, 512, 128), (1000, 3072, 128, 64)])
def test_linear_bnb4_fused(env, qt, M, N, K, blocksize):
    """Fused dequant-GEMM == BnbLinear::forward (dequantize_w to bf16, then matmul + bias)."""
    torch, L, lib, orc = env["torch"], env["L"], env["lib"], env["orc"]
    rng = np.random.default_rng(N + K)
    w = (rng.standard_normal((N, K)) / np.sqrt(K)).astype(np.float32)
    packed, absmax = orc.quantize_blockwise_4bit(w.ravel(), blocksize, qt)
    wdq = orc.dequantize_blockwise(None, packed, absmax, blocksize, N * K, qt, "bf16").reshape(N, K)
    x = bf16_round(rng.standard_normal((M, K)).astype(np.float32))
    b = bf16_round(rng.standard_normal(N).astype(np.float32))
    ref = orc.linear(x, wdq, b)
    y = torch.empty((M, N), dtype=torch.bfloat16, device="cuda")
    xd, pd, ad, bd = dev(x, torch.bfloat16), dev(packed), dev(absmax), dev(b, torch.bfloat16)
    L.check(lib.fmi_linear_bnb4_bf16(_p(xd), _p(pd), _p(ad), blocksize, {"fp4": 1, "nf4": 2}[qt], _p(bd), _p(y), M, N, K, 0, None))
    torch.cuda.synchronize()
    assert rel_l2(host(y), ref) <= 4e-3


@pytest.mark.parametrize("M,N,K,epi", [(300, 256, 256, 0), (64, 192, 512, 1), (1000, 3072, 128, 0), (4608, 3072, 3072, 0), (257, 260, 8256, 0)])
def test_linear_int8_fused_is_bit_identical_to_dequant_then_dense(env, M, N, K, epi):
    """LLM.int8 weights (BnbLinear::Int8, bitsandbytes/mod.rs:293-300: dequantize_8bit, then matmul) expanded inside the GEMM's
    weight-tile stage: the expansion reproduces w * SCB / 127 bit for bit (the division is a reciprocal multiply + one Newton step,
    correctly rounded for every finite input), so the product equals dequant-then-fmi_linear_bf16 exactly, and the oracle within
    the GEMM tolerance."""
    torch, L, lib, orc = env["torch"], env["L"], env["lib"], env["orc"]
    rng = np.random.default_rng(M + N + K)
    w = (rng.standard_normal((N, K)) / np.sqrt(K)).astype(np.float32)
    scb = np.abs(w).max(axis=1).astype(np.float32)
    w8 = np.clip(np.rint(w / scb[:, None] * 127.0), -127, 127).astype(np.int8)
    w8[0, :4] = [-128, 127, 0, -1]
    x = bf16_round(rng.standard_normal((M, K)).astype(np.float32))
    b = bf16_round(rng.standard_normal(N).astype(np.float32))
    xd, wd, sd, bd = dev(x, torch.bfloat16), dev(w8), dev(scb), dev(b, torch.bfloat16)
    y = torch.empty((M, N), dtype=torch.bfloat16, device="cuda")
    L.check(lib.fmi_linear_int8_bf16(_p(xd), _p(wd), _p(sd), _p(bd), _p(y), M, N, K, epi, None))
    wdq = torch.empty((N, K), dtype=torch.bfloat16, device="cuda")
    lib.dequantize_8bit_kernel_bf16(_p(wd), _p(sd), _p(wdq), N, K, N * K)
    y2 = torch.empty((M, N), dtype=torch.bfloat16, device="cuda")
    L.check(lib.fmi_linear_bf16(_p(xd), _p(wdq), _p(bd), _p(y2), M, N, K, epi, None))
    torch.cuda.synchronize()
    assert int((y.view(torch.int16) != y2.view(torch.int16)).sum()) == 0
    if epi == 0 and M * N * K <= 1000 * 3072 * 128:
        ref = orc.linear(x, orc.dequantize_8bit(w8.ravel(), scb, N, K, "bf16").reshape(N, K), b)
        assert rel_l2(host(y), ref) <= 4e-3


def test_pack_unpack_postprocess_bit_exact(env):
    torch, d, orc = env["torch"], env["d"], env["orc"]
    rng = np.random.default_rng(0)
    lat = rng.standard_normal((2, 16, 12, 20)).astype(np.float32)
    img, ids = d.pack_latents(dev(lat))
    rimg, rids = orc.pack_latents(lat)
    np.testing.assert_array_equal(host(img), rimg)
    np.testing.assert_array_equal(host(ids), rids)
    z = d.unpack_latents(img, 16, 12, 20, 0.3611, 0.1159)
    ref = orc.unpack_latents(rimg, 16, 12, 20) * np.float32(1.0 / 0.3611) + np.float32(0.1159)
    np.testing.assert_array_equal(host(z), ref.astype(np.float32))
    x = (rng.standard_normal((2, 3, 16, 24)) * 0.8).astype(np.float32)
    x[0, 0, 0, :4] = [np.nan, -5.0, 5.0, 1.0]
    np.testing.assert_array_equal(d.postprocess_u8(dev(x)).cpu().numpy(), orc.postprocess_u8(x))
    hwc = d.postprocess_u8(dev(x), interleave=True).cpu().numpy()
    np.testing.assert_array_equal(hwc, orc.postprocess_u8(x).transpose(0, 2, 3, 1))


def test_randn_is_seeded_normal(env):
    d = env["d"]
    a = host(d.randn_latents(2, 16, 64, 64, seed=7))
    b = host(d.randn_latents(2, 16, 64, 64, seed=7))
    c = host(d.randn_latents(1, 16, 64, 64, seed=7, first_sample=1))
    np.testing.assert_array_equal(a, b)
    np.testing.assert_array_equal(a[1], c[0])  # sample index, not batch position, selects the stream
    assert abs(a.mean()) < 0.02 and abs(a.std() - 1.0) < 0.02
    assert not np.array_equal(a[0], a[1])


# (96, 8, 37 * 53): 12 threads per pixel do not divide the grid stride (the per-iteration affine path of gn_apply) and the pixel count
# is not a multiple of the statistics chunk; (128, 32, 70001): many ragged chunks through the two-step combine and the parallel finalize
@pytest.mark.parametrize("C,G,HW,silu", [(64, 16, 100, 0), (128, 32, 4096, 1), (512, 32, 300, 1), (96, 8, 37 * 53, 1), (128, 32, 70001, 0)])
def test_groupnorm_nhwc(env, C, G, HW, silu):
    torch, L, lib, orc = env["torch"], env["L"], env["lib"], env["orc"]
    rng = np.random.default_rng(C)
    B = 2
    x = bf16_round((rng.standard_normal((B, C, HW)) * 2 + 0.3).astype(np.float32))
    w = (1 + 0.1 * rng.standard_normal(C)).astype(np.float32)
    b = (0.1 * rng.standard_normal(C)).astype(np.float32)
    ref = orc.group_norm(x, w, b, G, 1e-6)
    if silu:
        ref = orc.silu(ref)
    xd = dev(x.transpose(0, 2, 1).copy(), torch.bfloat16)
    out = torch.empty((B, HW, C), dtype=torch.bfloat16, device="cuda")
    wd, bd = dev(w), dev(b)
    L.check(lib.fmi_groupnorm_nhwc(_p(xd), _p(wd), _p(bd), _p(out), B, HW, C, G, 1e-6, silu, None))
    torch.cuda.synchronize()
    assert rel_l2(host(out).transpose(0, 2, 1), ref) <= 4e-3


@pytest.mark.parametrize("Cin,Cout,H,W,ks,up,resid", [(64, 64, 9, 7, 3, 0, False), (128, 192, 16, 16, 3, 0, True), (64, 128, 8, 12, 1, 0, False),
                                                      (64, 64, 6, 10, 3, 1, False), (128, 3, 20, 20, 3, 0, False)])
def test_conv2d_nhwc(env, Cin, Cout, H, W, ks, up, resid):
    torch, L, lib, orc = env["torch"], env["L"], env["lib"], env["orc"]
    rng = np.random.default_rng(Cin + Cout + H)
    B = 2
    x = bf16_round(rng.standard_normal((B, Cin, H, W)).astype(np.float32))
    w = bf16_round((rng.standard_normal((Cout, Cin, ks, ks)) / np.sqrt(Cin * ks * ks)).astype(np.float32))
    b = bf16_round(rng.standard_normal(Cout).astype(np.float32))
    xin = orc.upsample_nearest2d(x, 2 * H, 2 * W) if up else x
    ref = orc.conv2d(xin, w, b, pad=ks // 2)
    Ho, Wo = ref.shape[2], ref.shape[3]
    r = bf16_round(rng.standard_normal(ref.shape).astype(np.float32)) if resid else None
    if resid:
        ref = ref + r
    xd = dev(x.transpose(0, 2, 3, 1).copy(), torch.bfloat16)
    wd = dev(w.transpose(0, 2, 3, 1).copy(), torch.bfloat16)
    bd = dev(b, torch.bfloat16)
    rd = dev(r.transpose(0, 2, 3, 1).copy(), torch.bfloat16) if resid else None
    out = torch.full((B, Ho, Wo, Cout), float("nan"), dtype=torch.bfloat16, device="cuda")
    L.check(lib.fmi_conv2d_nhwc(_p(xd), _p(wd), _p(bd), _p(rd), _p(out), B, H, W, Cin, Cout, ks, up, None))
    torch.cuda.synchronize()
    got = host(out).transpose(0, 3, 1, 2)
    assert np.isfinite(got).all()
    assert rel_l2(got, ref) <= 4e-3


def test_empty_and_null_inputs_fail_cleanly(env):
    """Empty / null inputs return a negative fmi_status with a message (no crash, no launch): the C-ABI's
    equivalent of candle's shape errors.  Zero-length dequant is a no-op, like the CUDA launcher (grid 0)."""
    torch, L, lib, d = env["torch"], env["L"], env["lib"], env["d"]
    x = torch.zeros((64, 64), dtype=torch.bfloat16, device="cuda")
    assert lib.fmi_linear_bf16(_p(x), _p(x), None, _p(x), 0, 64, 64, 0, None) == 0  # zero rows: an empty result, nothing launched
    assert lib.fmi_linear_bf16(None, _p(x), None, _p(x), 64, 64, 64, 0, None) < 0
    q = torch.zeros((1, 1, 64, 128), dtype=torch.bfloat16, device="cuda")
    assert lib.fmi_sdpa_bf16(_p(q), _p(q), _p(q), _p(q), 1, 1, 0, 64, 128, C.c_float(1.0), 0, None) < 0
    assert lib.fmi_sdpa_bf16(_p(q), _p(q), _p(q), _p(q), 1, 1, 64, 64, 64, C.c_float(1.0), 0, None) < 0  # head dim != 128
    out = torch.full((4,), 7.0, dtype=torch.float32, device="cuda")
    a = torch.zeros((4,), dtype=torch.uint8, device="cuda")
    am = torch.ones((1,), dtype=torch.float32, device="cuda")
    lib.dequantize_blockwise_f32_nf4(None, _p(a), _p(am), _p(out), 64, 0, None)
    torch.cuda.synchronize()
    assert float(out.sum()) == 28.0  # untouched
    from tests.util import SMALL_FLUX, SMALL_VAE
    fm = d.FluxModel(SMALL_FLUX)
    with pytest.raises(d.FmiError):  # weights not loaded
        z = torch.zeros((1, 4, 64), device="cuda")
        fm.forward(z, torch.zeros((1, 4, 3), device="cuda"), torch.zeros((1, 4, SMALL_FLUX["joint_attention_dim"]), dtype=torch.bfloat16, device="cuda"),
                   torch.zeros((1, 4, 3), device="cuda"), torch.ones(1, device="cuda"), torch.zeros((1, SMALL_FLUX["pooled_projection_dim"]), device="cuda"),
                   torch.ones(1, device="cuda"))
    with pytest.raises(d.FmiError):
        fm.set_tensor("no.such.tensor", torch.zeros(3))
    with pytest.raises(d.FmiError):
        fm.set_tensor("x_embedder.weight", torch.zeros((3, 3)))  # wrong shape
    va = d.AutoEncoderKl(SMALL_VAE)
    with pytest.raises(d.FmiError):
        va.decode(torch.zeros((1, 16, 4, 4), device="cuda"))  # decoder weights missing


def test_gemm_4wave_kernel_is_bit_identical(tmp_path):
    """With FMI_GEMM_W4=1 eligible dense launches (N > 128: the f32 residual epilogue, or any epilogue at K >= 8192) go to the
    4-wave 128x128-per-wave kernel (v_mfma_f32_32x32x16_bf16), otherwise to the 8-wave ping-pong kernel (v_mfma_f32_16x16x32_bf16,
    the default since round 2).  Same accumulation order -> the outputs must be bit-identical — which also pins that the two MFMA
    shapes accumulate identically.  The switch is read when the library loads, so each setting runs in its own process."""
    import subprocess
    import sys
    script = r'''
import ctypes as C, sys, numpy as np, torch
sys.path.insert(0, ".")
from diffusion_rs_amd import _lib as L
lib = L.load()
p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
out = {}
g = torch.Generator(device="cuda").manual_seed(0)
for (M, N, K, epi) in [(300, 384, 256, 0), (1000, 260, 64, 0), (257, 1024, 1280, 1), (4608, 3072, 3072, 0), (512, 12288, 3072, 1), (4096, 3072, 12288, 0), (300, 512, 8192, 1), (257, 260, 8256, 0)]:
    x = torch.randn(M, K, device="cuda", generator=g).to(torch.bfloat16)
    w = (torch.randn(N, K, device="cuda", generator=g) * 0.05).to(torch.bfloat16)
    b = torch.randn(N, device="cuda", generator=g).to(torch.bfloat16)
    y = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    L.check(lib.fmi_linear_bf16(p(x), p(w), p(b), p(y), M, N, K, epi, None))
    torch.cuda.synchronize()
    out[f"{M}x{N}x{K}e{epi}"] = y.view(torch.int16).cpu().numpy()
# the residual-update GEMMs of a small FLUX (f32 read-modify-write epilogue, grouped img + txt launches)
import diffusion_rs_amd as d
from tests.util import SMALL_FLUX, dev, flux_inputs, host
gm = d.FluxModel(SMALL_FLUX)
gm.load_state_dict(d.synth.flux_state_dict_numpy(SMALL_FLUX, seed=0))
img, ids, txt, txt_ids, y = flux_inputs(SMALL_FLUX, 2, (8, 12), 40)
t = np.array([0.8, 0.5], np.float32)
g3 = np.full(2, 3.5, np.float32)
out["flux_forward"] = host(gm.forward(dev(img), dev(ids), dev(txt, torch.bfloat16), dev(txt_ids), dev(t), dev(y), dev(g3)))
np.savez(sys.argv[1], **out)
'''
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = {}
    from diffusion_rs_amd import _lib as L
    if not os.path.exists(L.ALT_LIB_PATH):
        pytest.skip("the test build of the library (libflux_mi355x_alt.so, make alt) is not there: the 4-wave kernel lives in it")
    for v in ("0", "1"):
        path = str(tmp_path / f"w4_{v}.npz")
        # "0": the PRODUCT library (8-wave ping-pong kernel); "1": the test build with the 4-wave kernel switched on — two binaries, the same bits
        env = dict(os.environ, FMI_GEMM_W4=v)
        if v == "1":
            env["FMI_LIB"] = L.ALT_LIB_PATH
        subprocess.run([sys.executable, "-c", script, path], check=True, cwd=root, env=env, timeout=600)
        res[v] = np.load(path)
    for k in res["0"].files:
        np.testing.assert_array_equal(res["0"][k], res["1"][k], err_msg=k)
