"""VAE decoder parity: AutoEncoderKl::decode on the GPU (NHWC bf16, MFMA implicit-GEMM convs)
vs the f32 CPU oracle; then the u8 post-process.  Tolerance: rel-L2 <= 2e-2 on the decoded
image, u8 |delta| <= 2 on >= 99% of pixels (bf16 activations through ~30 layers)."""
import numpy as np
import pytest

from tests.util import SMALL_VAE, dev, host, rel_l2

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("B,h,w", [(1, 8, 8), (2, 8, 16)])
def test_vae_decode_matches_oracle(B, h, w):
    import torch
    import diffusion_rs_amd as d
    from oracle import oracle as orc
    sd = d.synth.vae_state_dict_numpy(SMALL_VAE, seed=0)
    gv = d.AutoEncoderKl(SMALL_VAE)
    gv.load_state_dict(sd)
    ov = orc.Vae(SMALL_VAE)
    ov.load(sd)
    rng = np.random.default_rng(B * 10 + w)
    z = rng.standard_normal((B, 16, h, w)).astype(np.float32)
    ref = ov.decode(z)
    got = host(gv.decode(dev(z)))
    assert got.shape == ref.shape == (B, 3, 8 * h, 8 * w)
    assert np.isfinite(got).all()
    err = rel_l2(got, ref)
    print(f"vae decode: rel-L2 {err:.3e}, ref range [{ref.min():.2f},{ref.max():.2f}]")
    assert err <= 2e-2
    u_ref = orc.postprocess_u8(ref)
    u_got = d.postprocess_u8(torch.from_numpy(got).cuda()).cpu().numpy()
    diff = np.abs(u_ref.astype(np.int32) - u_got.astype(np.int32))
    frac = float((diff <= 2).mean())
    print(f"u8: max |d| {diff.max()}, frac<=2 {frac:.4f}")
    assert frac >= 0.99
