"""Pin the CPU oracle against the reference's own known-answer vectors.

The expected values in tests/golden/reference_kats.json are transcribed from the reference's
orphaned candle tests; comparisons use the reference's own convention (round to 4 decimals,
core/test_utils.rs:27-62) unless the reference asserts exact equality.
"""
import json
import os

import numpy as np
import pytest

from oracle import oracle as orc


@pytest.fixture(scope="module")
def kats(golden_dir):
    with open(os.path.join(golden_dir, "reference_kats.json")) as f:
        return json.load(f)


def r4(a):
    return np.round(np.asarray(a, np.float64), 4) + 0.0  # +0.0 folds -0.0


def eq4(got, exp):
    np.testing.assert_array_equal(r4(got), r4(exp))


def test_softmax(kats):
    k = kats["softmax_last_dim"]
    x = np.log(np.array(k["input_before_log"], np.float32))
    eq4(orc.softmax_last_dim(x), k["expected"])
    s = kats["softmax_stability"]
    np.testing.assert_array_equal(orc.softmax_last_dim(np.array([s["input"]], np.float32))[0], np.array(s["expected"], np.float32))


def test_rms_norm(kats):
    k = kats["rms_norm"]
    eq4(orc.rms_norm_slow(np.array(k["input"], np.float32), np.array(k["alpha"], np.float32), k["eps"]), k["expected"])


def test_layer_norm_ops(kats):
    k = kats["layer_norm_ops"]
    eq4(orc.layer_norm(np.array(k["input"], np.float32), k["alpha"], k["beta"], k["eps"]), k["expected"])


def test_layer_norm_module(kats):
    k = kats["layer_norm_module"]
    for case in k["cases"]:
        x = np.array(case["input"], np.float32)
        n = x.shape[-1]
        got = orc.layer_norm(x, [k["w"]] * n, [k["b"]] * n, k["eps"])
        if case["exact"]:
            np.testing.assert_array_equal(got, np.array(case["expected"], np.float32))
        else:
            eq4(got, case["expected"])
            # reference also checks mean == b and "std" == sqrt(w) (layer_norm.rs:41-53)
            mean = got.sum(-1, keepdims=True) / 3.0
            eq4(mean, [[[0.5], [0.5], [0.5]]])
            std = np.sqrt(((got - mean) ** 2).sum(-1, keepdims=True)) / 3.0
            eq4(std, [[[1.7321], [1.7321], [1.7321]]])


def test_layer_norm_fast_equals_slow():
    # nn/tests/ops.rs:143-163: fast vs slow path max diff < 1e-5 on (24,70,64) uniform data
    rng = np.random.default_rng(299792458)
    x = rng.random((24, 70, 64), dtype=np.float32)
    fast = orc.layer_norm(x, np.ones(64, np.float32), np.zeros(64, np.float32), 1e-5)
    mean = x.mean(-1, keepdims=True)
    xc = x - mean
    slow = xc / np.sqrt((xc * xc).mean(-1, keepdims=True) + 1e-5)
    assert np.abs(fast - slow).max() < 1e-5


def test_group_norm(kats):
    k = kats["group_norm"]
    x = np.array(k["input"], np.float32)
    w, b = np.ones(6, np.float32), np.zeros(6, np.float32)
    eq4(orc.group_norm(x, w, b, 2, k["eps"]), k["expected_g2"])
    eq4(orc.group_norm(x, w, b, 3, k["eps"]), k["expected_g3"])


def test_conv2d(kats):
    k = kats["conv2d"]
    t = np.array(k["t"], np.float32).reshape(k["t_shape"])
    w = np.array(k["w"], np.float32).reshape(k["w_shape"])
    res = orc.conv2d(t, w)
    assert res.shape == (1, 2, 3, 3)
    eq4(res.ravel(), k["expected_pad0"])
    res = orc.conv2d(t, w, dilation=2)
    assert res.shape == (1, 2, 1, 1)
    eq4(res.ravel(), k["expected_dilation2"])


def test_conv2d_small(kats):
    k = kats["conv2d_small"]
    t = np.array(k["t"], np.float32).reshape(k["t_shape"])
    w = np.array(k["w"], np.float32).reshape(k["w_shape"])
    eq4(orc.conv2d(t, w).ravel(), k["expected_pad0"])
    res = orc.conv2d(t, w, pad=2)
    assert list(res.shape) == k["expected_pad2_shape"]
    exp = np.zeros((7, 7))
    exp[2:5, 2:5] = np.array(k["expected_pad0"]).reshape(3, 3)
    eq4(res[0, 0], exp)
    k = kats["conv2d_smaller"]
    t = np.array(k["t"], np.float32).reshape(1, 1, 3, 3)
    eq4(orc.conv2d(t, np.ones((1, 1, 3, 3), np.float32)).ravel(), k["expected"])
    k = kats["conv2d_non_square"]
    t = np.array(k["t"], np.float32).reshape(k["t_shape"])
    w = np.array(k["w"], np.float32).reshape(k["w_shape"])
    eq4(orc.conv2d(t, w).ravel(), k["expected"])


def test_matmul_and_linear(kats):
    for c in kats["matmul"]["cases"]:
        a, b = np.array(c["a"], np.float32), np.array(c["b"], np.float32)
        # oracle linear computes x @ w.T, so pass w = b.T
        np.testing.assert_array_equal(orc.linear(a, b.T.copy()), np.array(c["expected"], np.float32))
    k = kats["linear_doctest"]
    np.testing.assert_array_equal(orc.linear(np.array(k["x"], np.float32), np.array(k["w"], np.float32)), np.array(k["expected"], np.float32))


def test_gelu_silu(kats):
    k = kats["gelu_silu"]
    x = np.array(k["input"], np.float32)
    eq4(orc.gelu(x), k["gelu"])
    eq4(orc.silu(x), k["silu"])


def test_upsample(kats):
    k = kats["upsample_nearest2d"]
    x = np.array(k["input"], np.float32).reshape(1, 1, 2, 3)
    np.testing.assert_array_equal(orc.upsample_nearest2d(x, 4, 6)[0, 0], np.array(k["expected"], np.float32))


def test_nf4_fp4_tables(kats):
    lut = np.array(kats["nf4_lut"]["values"], np.float32)
    # every nibble value with absmax 1.0 must give the LUT entry exactly (bit-exact f32)
    packed = np.array([(i << 4) | i for i in range(16)], np.uint8)
    out = orc.dequantize_blockwise(None, packed, np.ones(1, np.float32), 64, 32, "nf4")
    np.testing.assert_array_equal(out[0::2], lut)
    np.testing.assert_array_equal(out[1::2], lut)
    fp4 = np.array(kats["fp4_tree"]["abs_values_by_low3bits"], np.float32)
    out = orc.dequantize_blockwise(None, packed, np.ones(1, np.float32), 64, 32, "fp4")
    exp = np.array([(-1.0 if i & 8 else 1.0) * fp4[i & 7] for i in range(16)], np.float32)
    np.testing.assert_array_equal(out[0::2], exp)


def test_sdpa_matches_naive():
    # shape of nn/tests/sdpa.rs:4-37 ((4,3,4,64) f32).  That test compares two f32 paths with
    # sum(|ref-out|/|ref|) <= 5e-4; against an f64 reference the same sum is dominated by f32
    # round-off on near-zero outputs, so the oracle is held to max-abs 1e-5 instead.
    rng = np.random.default_rng(0)
    q, k, v = (rng.standard_normal((4, 3, 4, 64), dtype=np.float32) for _ in range(3))
    scale = 1.0 / np.sqrt(64.0)
    att = (q.astype(np.float64) * scale) @ k.astype(np.float64).transpose(0, 1, 3, 2)
    att = np.exp(att - att.max(-1, keepdims=True))
    att /= att.sum(-1, keepdims=True)
    ref = att @ v.astype(np.float64)
    got = orc.sdpa(q, k, v, scale)
    assert np.abs(ref - got).max() <= 1e-5


def test_division_by_127_as_reciprocal_plus_newton_step_is_correctly_rounded():
    """The fused LLM.int8 GEMM (gemm_bf16.hip: div127) computes x / 127 as q = x * RN(1/127); q + fma(-127, q, x) * RN(1/127).
    Checked exhaustively over all 2^32 f32 bit patterns when the kernel was written (0 mismatches for |x| >= 2^-118); this keeps
    a sampled version in the suite: every product of an int8 with 4096 random scales, and 2^22 random bit patterns."""
    rng = np.random.default_rng(127)
    r = np.float32(1.0) / np.float32(127.0)

    def fma(a, b, c):  # float64 holds the exact product of two f32 and the sum rounds once to f32 magnitude: an exact fma
        return (a.astype(np.float64) * b.astype(np.float64) + c.astype(np.float64)).astype(np.float32)

    def div127(x):
        q = (x * r).astype(np.float32)
        return fma(fma(np.full_like(x, -127.0), q, x), np.full_like(x, r), q)

    scb = (rng.random(4096) * 4).astype(np.float32)
    w = np.arange(-128, 128, dtype=np.float32)
    x = (w[:, None] * scb[None, :]).astype(np.float32).ravel()
    np.testing.assert_array_equal(div127(x).view(np.uint32), (x / np.float32(127.0)).astype(np.float32).view(np.uint32))
    bits = rng.integers(0, 1 << 32, 1 << 22, dtype=np.uint64).astype(np.uint32)
    x = bits.view(np.float32)
    x = x[np.isfinite(x) & (np.abs(x) >= np.float32(2.0) ** -100) & (np.abs(x) <= np.float32(2.0) ** 120)]
    np.testing.assert_array_equal(div127(x).view(np.uint32), (x / np.float32(127.0)).astype(np.float32).view(np.uint32))


def test_gemm_isa_paths_are_bit_identical():
    """The oracle's GEMM picks a 512-bit micro-kernel on hosts that have AVX-512 (round 5: speed only).  Every output element must be the
    same f32 FMA chain as in the AVX2 form — ragged rows, an odd number of 16-column panels, K not a multiple of the 256-deep block."""
    rng = np.random.default_rng(11)
    try:
        for (M, N, K) in ((1, 48, 64), (7, 3, 27), (97, 272, 300), (256, 1040, 513), (13, 16, 1024)):
            x = rng.standard_normal((M, K)).astype(np.float32)
            w = rng.standard_normal((N, K)).astype(np.float32)
            b = rng.standard_normal(N).astype(np.float32)
            orc.set_isa(1)
            assert orc.get_isa() == 256
            a = orc.linear(x, w, b)
            orc.set_isa(0)
            c = orc.linear(x, w, b)
            assert np.array_equal(a.view(np.uint32), c.view(np.uint32)), (M, N, K, orc.get_isa())
            ref = x.astype(np.float64) @ w.astype(np.float64).T + b
            assert np.abs(a - ref).max() <= 1e-4 * np.abs(ref).max()
        # few rows (M <= 4): computed straight from the W rows, no packing — the same chain per output
        for (M, N, K) in ((1, 100, 700), (4, 19, 256), (3, 64, 3072)):
            x = rng.standard_normal((M, K)).astype(np.float32)
            w = rng.standard_normal((N, K)).astype(np.float32)
            b = rng.standard_normal(N).astype(np.float32)
            orc.set_isa(3)  # AVX2 micro-kernel, packed path
            a = orc.linear(x, w, b)
            orc.set_isa(0)
            c = orc.linear(x, w, b)
            assert np.array_equal(a.view(np.uint32), c.view(np.uint32)), (M, N, K)
    finally:
        orc.set_isa(0)
