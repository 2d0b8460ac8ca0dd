"""Full-depth parity: ALL 19 double-stream + 38 single-stream blocks (Flux::forward, models/flux/model.rs:790-833)
through 8 Euler steps (Sampler::sample, pipelines/sampling.rs:25-48) against the f32 CPU oracle, at reduced width
(D = 512, 4 heads) so the oracle finishes in seconds.  This is the measurement of SURVEY §7 hard part (b): how far the
bf16-operand / f32-accumulate / f32-residual GPU path drifts from f32 over 57 blocks x N steps.  Weights are drawn so
that the gates are O(0.3) — every block changes the residual stream noticeably, unlike the default synthetic weights
whose gates are ~0.03 — otherwise depth would not be exercised.

Stated tolerances: one model evaluation rel-L2 <= 2e-2, latents after the loop rel-L2 <= 3e-2; the
measured figures are printed next to them."""
import numpy as np
import pytest

from tests.util import dev, flux_inputs, host, rel_l2

pytestmark = pytest.mark.gpu

DEEP = dict(in_channels=64, pooled_projection_dim=128, joint_attention_dim=256, num_attention_heads=4, num_layers=19,
            num_single_layers=38, guidance_embeds=True, axes_dim=[16, 56, 56], theta=10000)


@pytest.fixture(scope="module")
def deep():
    import torch
    import diffusion_rs_amd as d
    from oracle import oracle as orc
    sd = d.synth.flux_state_dict_numpy(DEEP, seed=3, w_std=0.03, mod_std=0.04, bias_std=0.01)
    gm = d.FluxModel(DEEP)
    gm.load_state_dict(sd)
    om = orc.Flux(DEEP)
    om.load(sd)
    return torch, d, gm, om


def test_full_depth_forward_matches_oracle(deep):
    torch, d, gm, om = deep
    img, ids, txt, txt_ids, y = flux_inputs(DEEP, 1, (16, 16), 64, seed=21)
    t = np.array([0.75], np.float32)
    g = np.array([3.5], np.float32)
    ref = om.forward(img, ids, txt, txt_ids, t, y, g)
    got = host(gm.forward(dev(img), dev(ids), dev(txt, torch.bfloat16), dev(txt_ids), dev(t), dev(y), dev(g)))
    err = rel_l2(got, ref)
    print(f"19+38 blocks, D=512, one evaluation: rel-L2 {err:.3e} (tolerance 2e-2); |pred| rms {np.sqrt((ref ** 2).mean()):.3f}")
    assert np.isfinite(got).all() and err <= 2e-2


def test_full_depth_euler_loop_drift(deep):
    torch, d, gm, om = deep
    B, S_hw, T, steps = 1, (16, 16), 64, 8
    img, ids, txt, txt_ids, y = flux_inputs(DEEP, B, S_hw, T, seed=22)
    g = np.full(B, 3.5, np.float32)
    from oracle import oracle as orc
    ts = list(orc.get_timesteps(steps, True, orc.calculate_shift(S_hw[0] * S_hw[1])))
    ref = om.denoise(img, ids, txt, txt_ids, y, g, ts)
    got = host(gm.denoise(dev(img), dev(ids), dev(txt, torch.bfloat16), dev(txt_ids), dev(y), dev(g), ts))
    err = rel_l2(got, ref)
    moved = rel_l2(ref, img)
    print(f"19+38 blocks x {steps} Euler steps: latent rel-L2 {err:.3e} (tolerance 3e-2); the loop moved the latent by {moved:.3f} of its norm")
    assert np.isfinite(got).all() and err <= 3e-2
    assert moved > 0.05  # the trajectory is not a no-op: depth and steps are exercised
    # drift per step stays bounded: the same loop cut at 4 steps is no worse
    ref4 = om.denoise(img, ids, txt, txt_ids, y, g, ts[:5])
    got4 = host(gm.denoise(dev(img), dev(ids), dev(txt, torch.bfloat16), dev(txt_ids), dev(y), dev(g), ts[:5]))
    print(f"  after 4 of the {steps} steps: {rel_l2(got4, ref4):.3e}")


def test_state_export_adopt_roundtrip(deep):
    """The flat weight state (multi-GPU broadcast unit): a second handle that adopts the exported layout and receives
    the arena bytes computes the same prediction bit for bit."""
    torch, d, gm, om = deep
    import ctypes as C
    from diffusion_rs_amd import _lib as L
    blob = gm.state_export()
    g2 = d.FluxModel(DEEP)
    assert len(g2.missing()) > 0
    g2.state_adopt(blob)
    assert g2.missing() == []
    for (ptr, n), src, dst in zip(gm.state_buffers(), gm.state_views(), g2.state_views()):  # the in-place views the broadcast uses
        assert (src is None) == (n == 0) == (dst is None)
        if n:
            assert src.numel() == dst.numel() == n and src.data_ptr() == ptr
            dst.copy_(src)
    torch.cuda.synchronize()
    img, ids, txt, txt_ids, y = flux_inputs(DEEP, 1, (8, 8), 32, seed=5)
    t = np.array([0.5], np.float32)
    g = np.array([3.5], np.float32)
    args = (dev(img), dev(ids), dev(txt, torch.bfloat16), dev(txt_ids), dev(t), dev(y), dev(g))
    np.testing.assert_array_equal(host(gm.forward(*args)), host(g2.forward(*args)))
    g2.close()
