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
