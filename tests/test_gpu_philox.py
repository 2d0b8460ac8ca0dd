"""Row a4 — the seedable replacement of get_noise (pipelines/flux/sampling.rs:5-14) against its checker: the device
Philox4x32-10 stream is integer work and must equal the oracle's (pinned by Random123's known answers,
tests/test_oracle_philox.py) bit for bit; the Box-Muller normals are f32 libm results and are compared with the
oracle's f64 evaluation within a stated ulp bound."""
import ctypes as C
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env():
    import torch
    from diffusion_rs_amd import _lib as L
    from oracle import oracle as orc
    lib = L.load()
    L.check(lib.fmi_init(0))
    return torch, L, lib, orc


def _u32(torch, L, lib, n, B, seed, first):
    out = torch.empty((B, n), dtype=torch.int32, device="cuda")
    L.check(lib.fmi_philox_u32(C.c_void_p(out.data_ptr()), n, B, seed, first, None))
    torch.cuda.synchronize()
    return out.cpu().numpy().view(np.uint32)


@pytest.mark.parametrize("n,B,seed,first", [(4096, 2, 0, 0), (1003, 3, 0x123456789ABCDEF0, 5), (16 * 128 * 128, 1, 299792458, 7), (1, 1, 1, 1 << 40)])
def test_device_philox_stream_is_bit_exact(env, n, B, seed, first):
    torch, L, lib, orc = env
    got = _u32(torch, L, lib, n, B, seed, first)
    np.testing.assert_array_equal(got, orc.philox_u32(n, B, seed, first))


def test_device_philox_reproduces_random123_known_answers(env, golden_dir):
    """The KAT counters straight through the device kernel: counter (c0, c1, c2, c3) = quad index c0 | c1 << 32 of sample
    c2 | c3 << 32; the huge quad indices are reached by offsetting the output pointer arithmetic on the host side, so only
    the all-zero vector can be requested directly — the others are covered through the oracle equality above plus this
    direct check of vector 0."""
    torch, L, lib, orc = env
    kat = json.load(open(os.path.join(golden_dir, "philox_kat.json")))["philox4x32_10"][0]
    got = _u32(torch, L, lib, 4, 1, 0, 0)[0]
    assert [f"{int(x):08x}" for x in got] == kat["out"]


def test_device_normals_match_oracle_within_ulps(env):
    torch, L, lib, orc = env
    import diffusion_rs_amd as d
    n = 16 * 64 * 64
    z = d.randn_latents(2, 16, 64, 64, seed=42, first_sample=3).cpu().numpy().reshape(2, n)
    ref = orc.randn(n, 2, 42, 3)
    # distance in units of f32 ulp of the reference value; the radius sqrt(-2 ln u) and the angle's sin / cos are three
    # libm calls of <= 1-2 ulp each and the product rounds once more; near a zero of sin / cos the absolute error of the
    # f32 angle (2^-24 * 2 pi) dominates, so the bound is max(8 ulp, 4e-7 absolute)
    ulp = np.spacing(np.abs(ref).astype(np.float32)).astype(np.float64)
    err = np.abs(z.astype(np.float64) - ref.astype(np.float64))
    bad = err > np.maximum(8 * ulp, 4e-7)
    print(f"normals: max err {err.max():.3e}, max ulp distance {(err / ulp).max():.1f} (median {(np.median(err / ulp)):.2f}), outside bound: {int(bad.sum())}")
    assert not bad.any()
    assert abs(float(z.mean())) < 0.01 and abs(float(z.std()) - 1.0) < 0.01
    # per-sample streams: a batch equals its samples drawn one by one (what batch sharding relies on)
    one = d.randn_latents(1, 16, 64, 64, seed=42, first_sample=4).cpu().numpy().reshape(n)
    np.testing.assert_array_equal(one, z[1])
