"""Row a4 (seedable get_noise, flux/sampling.rs:5-14): the oracle's Philox4x32-10 against Random123's own
known-answer vectors, and the layout of the stream `randn` draws from it (CPU only)."""
import json
import os

import numpy as np

from oracle import oracle as orc


def test_philox4x32_10_random123_known_answers(golden_dir):
    kat = json.load(open(os.path.join(golden_dir, "philox_kat.json")))["philox4x32_10"]
    assert len(kat) == 3
    for v in kat:
        ctr = np.array([int(x, 16) for x in v["ctr"]], np.uint32)
        key = np.array([int(x, 16) for x in v["key"]], np.uint32)
        out = orc.philox4x32_10(ctr, key)
        assert [f"{int(x):08x}" for x in out] == v["out"]


def test_stream_layout_and_sample_independence():
    a = orc.philox_u32(10, 3, seed=0x1234_5678_9ABC_DEF0, first_sample=5)
    # element e of sample b = word e % 4 of counter (e // 4, 0, sample, 0) under key (seed lo, seed hi)
    for b in range(3):
        for e in (0, 3, 4, 9):
            w = orc.philox4x32_10(np.array([e // 4, 0, 5 + b, 0], np.uint32), np.array([0x9ABCDEF0, 0x12345678], np.uint32))
            assert a[b, e] == w[e % 4]
    # a batch is the concatenation of its samples' streams (batch sharding: sample i on rank i % N draws the same numbers)
    np.testing.assert_array_equal(orc.philox_u32(10, 1, 0x123456789ABCDEF0, 6), a[1:2])
    # ragged tail: n_per_sample not a multiple of 4
    np.testing.assert_array_equal(orc.philox_u32(7, 1, 9)[0], orc.philox_u32(8, 1, 9)[0, :7])


def test_randn_moments_and_box_muller_pairs():
    z = orc.randn(1 << 18, 2, seed=299792458)
    assert abs(float(z.mean())) < 5e-3 and abs(float(z.std()) - 1.0) < 5e-3
    assert np.isfinite(z).all() and float(np.abs(z).max()) < 6.0  # u1 >= 2^-25 -> |z| <= sqrt(50 ln 2)
    w = orc.philox_u32(4, 1, 7)[0]
    f = lambda x: float((np.float32(int(x) >> 8) + np.float32(0.5)) * np.float32(2.0 ** -24))  # the device's f32 roundings
    u1, u2 = f(w[0]), f(w[1])
    zz = orc.randn(4, 1, 7)[0]
    ang = float(np.float32(6.283185307179586) * np.float32(u2))
    np.testing.assert_allclose(zz[:2], [np.sqrt(-2 * np.log(u1)) * np.cos(ang), np.sqrt(-2 * np.log(u1)) * np.sin(ang)], rtol=1e-6)


def test_c_philox_equals_the_numpy_statement():
    """orc_philox_u32 (flux_oracle.cpp; what the full-size fixtures draw 12e9 words from) against the numpy statement the KATs above pin."""
    for n, B, seed, first in ((1, 1, 0, 0), (7, 2, 0x123456789ABCDEF0, 5), (4099, 3, 42, 2**33 + 1), (64, 1, 2**64 - 1, 2**64 - 4)):
        np.testing.assert_array_equal(orc.philox_u32_c(n, B, seed, first), orc.philox_u32(n, B, seed, first))


def test_exact_synthetic_tensors_c_pass_equals_the_numpy_definition():
    """"Exact synthetic tensors" (diffusion-rs_amd/synth.py): value = bf16(f32(byte sum - 510) * coeff (+ offset)) from the Philox stream seeded by the
    tensor's name.  The numpy function is the definition; the oracle's one-pass C form (what tests/golden/gen_c2_trajectory_fixture.py loads the
    11.9e9 weights with) must give the same bits for every rule of the synthetic checkpoint, and the statistics must be what the rules say."""
    import diffusion_rs_amd as d
    S = d.synth
    raw = lambda n, seed: orc.philox_u32_c(n, 1, seed)[0]
    cases = [("transformer_blocks.3.attn.to_q.weight", (257, 33), "flux", 0.0, 0.02), ("transformer_blocks.3.attn.norm_q.weight", (128,), "flux", 1.0, 0.1),
             ("transformer_blocks.3.norm1.linear.weight", (96, 50), "flux", 0.0, 0.01), ("x_embedder.bias", (1001,), "flux", 0.0, 0.02),
             ("decoder.conv_in.weight", (8, 4, 3, 3), "vae", 0.0, 1.0 / 6.0), ("decoder.mid_block.resnets.0.norm1.weight", (64,), "vae", 1.0, 0.1),
             ("input.c2.t5", (3, 40, 7), "input", 0.0, 1.0)]
    for name, shape, fam, off, sc in cases:
        assert S.exact_rule(name, shape, fam) == (off, sc)
        a = S.exact_tensor_np(name, shape, raw, fam)
        n = int(np.prod(shape))
        bits = orc.exact_bf16(n, S.exact_seed(name), off, S.exact_coeff(sc))
        back = (bits.astype(np.uint32) << 16).view(np.float32).reshape(shape)
        np.testing.assert_array_equal(a, back)
        np.testing.assert_array_equal(a, S.to_bf16_f32(a))  # bf16-representable
    big = S.exact_tensor_np("w", (1 << 20,), raw, "flux")
    assert abs(float(big.mean())) < 1e-4 and abs(float(big.std()) / 0.02 - 1.0) < 5e-3 and float(np.abs(big).max()) <= 0.02 * 3.46
    # different names, different streams; the salt moves a stream too
    assert not np.array_equal(S.exact_tensor_np("a", (64,), raw), S.exact_tensor_np("b", (64,), raw))
    assert not np.array_equal(S.exact_tensor_np("a", (64,), raw), S.exact_tensor_np("a", (64,), raw, salt=1))
