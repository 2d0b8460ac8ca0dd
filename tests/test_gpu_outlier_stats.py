"""Off-Gaussian stress of the three modes (VERDICT r5 weak 3 / "next" 3): every tolerance of DESIGN 5 was measured on N(0, 0.02^2) weights, and per-token
8-bit grids are exactly the recipes that real DiT statistics break.  Here FLUX.1-dev at its TRUE width and depth (D = 3072, 19 + 38 blocks) gets the
outlier-channel profile of diffusion-rs_amd/synth.py — 12 hidden channels (0.39 %) whose AdaLN (1 + scale) is 30-100 in every block, 2 massive residual
channels, QkNorm weights with three x 3 dimensions — and ONE `Flux::forward` (model.rs:790-833) at 1024 image + 128 text tokens is compared with the f32
CPU oracle on the same weights, in bf16, in the int8 mode and in the e4m3 mode.  The table goes to DESIGN 5; bars:

  bf16  <= 2e-2   (the full-depth bar of the Gaussian checkpoint: bf16 is a floating-point format, outliers cost it nothing)
  int8  <= 3e-2   (the 8-bit bar) — with the per-channel smoothing of round 6 (fmi_flux_quantize_int8 calibrates it); the un-smoothed recipe is measured
                  next to it and is far outside
  e4m3  reported  (~1e-1 as on Gaussian weights: a floating-point grid does not care about the outliers; outside the 8-bit bar either way)

Measured (profiles/r06_outlier_study.txt, which also takes the profile apart piece by piece): bf16 3.7e-3, int8 unsmoothed 1.9e-1, int8 smoothed 2.55e-2 (calibrated on
another sample), e4m3 7.5e-2.
"""
import time

import numpy as np
import pytest

from tests.util import dev, host, rel_l2

pytestmark = pytest.mark.gpu
S_HW, T_TXT = (32, 32), 128  # a 512 x 512 image: 1024 image tokens


def _host_memory_gib():
    from tests.test_gpu_production_shapes import _host_memory_gib as f
    return f()


@pytest.fixture(scope="module")
def outlier():
    import torch
    import diffusion_rs_amd as d
    from oracle import oracle as orc
    if _host_memory_gib() < 40:
        print("\n!!! NOT RUN: the outlier-statistics stress needs 24 GB of host memory for the oracle's bf16 weight image !!!")
        pytest.skip("the host cannot hold the oracle's weights (24 GB as bf16 bits)")
    S = d.synth
    cfg = dict(d.FLUX_DEV)
    D = cfg["num_attention_heads"] * sum(cfg["axes_dim"])
    t0 = time.time()
    om = orc.Flux(cfg)

    def tensors():
        for name, shape in S.flux_tensor_shapes(cfg).items():
            yield name, S.apply_outlier_profile(name, S.exact_tensor_device(name, shape, "flux", salt=3), D)

    gm = d.FluxModel(cfg)
    for name, t in tensors():
        gm.set_tensor(name, t)
        om.set_tensor_bf16(name, t.view(torch.int16).cpu().numpy().view(np.uint16))
    gm.assert_complete()
    lat = S.exact_tensor_device("input.outlier.latent", (1, 16, 2 * S_HW[0], 2 * S_HW[1]), "input").float()
    t5 = S.exact_tensor_device("input.outlier.t5", (1, T_TXT, cfg["joint_attention_dim"]), "input")
    clip = S.exact_tensor_device("input.outlier.clip", (1, cfg["pooled_projection_dim"]), "input").float()
    img, ids = d.pack_latents(lat)
    txt_ids = torch.zeros((1, T_TXT, 3), device="cuda")
    t = torch.tensor([0.6], device="cuda")
    g = torch.tensor([3.5], device="cuda")
    args = (img, ids, t5, txt_ids, t, clip, g)
    t1 = time.time()
    ref = om.forward(host(img), host(ids), host(t5), host(txt_ids), host(t), host(clip), host(g))
    print(f"\noutlier-profile FLUX.1-dev (19 + 38 blocks) on the GPU and in the oracle in {t1 - t0:.0f} s; f32 oracle forward at {S_HW[0] * S_HW[1]} + {T_TXT} tokens {time.time() - t1:.0f} s; "
          f"|ref| rms {float(np.sqrt((ref.astype(np.float64) ** 2).mean())):.3f}")
    assert np.isfinite(ref).all()
    yield dict(torch=torch, d=d, orc=orc, gm=gm, om=om, args=args, ref=ref, tensors=tensors, cfg=cfg)
    gm.close()


def test_bf16_is_indifferent_to_outlier_channels(outlier):
    gm, ref = outlier["gm"], outlier["ref"]
    got = host(gm.forward(*outlier["args"]))
    err = rel_l2(got, ref)
    print(f"OUTLIER PROFILE, bf16: one Flux::forward vs the f32 oracle rel-L2 {err:.3e}  (Gaussian checkpoint at full size: 4.6e-3)")
    outlier["bf16"] = err
    assert np.isfinite(got).all() and err <= 2e-2


def test_8bit_modes_on_outlier_channels(outlier):
    d, ref, torch = outlier["d"], outlier["ref"], outlier["torch"]
    img, ids, t5, txt_ids, t, clip, g = outlier["args"]

    def calibrated_int8(m):
        # the smoothed recipe (round 6): 4 evaluations across the schedule record the per-channel absmax of every block linear's input, then quantise
        # on ANOTHER sample (latents, text, pooled vector) than the one evaluated below: the statistics have to carry over
        S, cfg = d.synth, outlier["cfg"]
        c_img, _ = d.pack_latents(S.exact_tensor_device("input.calib.latent", (1, 16, 2 * S_HW[0], 2 * S_HW[1]), "input").float())
        c_t5 = S.exact_tensor_device("input.calib.t5", (1, T_TXT, cfg["joint_attention_dim"]), "input")
        c_clip = S.exact_tensor_device("input.calib.clip", (1, cfg["pooled_projection_dim"]), "input").float()
        m.calibrate_int8(True)
        for tt in (1.0, 0.75, 0.5, 0.25):
            m.forward(c_img, ids, c_t5, txt_ids, torch.tensor([tt], device="cuda"), c_clip, g)
        m.quantize_int8()

    rows = []
    for label, quant in (("int8 UNSMOOTHED (the round-5 recipe), default mask, e4m3 q / k", lambda m: m.quantize_int8()),
                         ("int8 SMOOTHED (calibrated per-channel factors), default mask, e4m3 q / k", calibrated_int8), ("e4m3", lambda m: m.quantize_fp8())):
        m = d.FluxModel(outlier["cfg"])
        try:
            for name, tns in outlier["tensors"]():
                m.set_tensor(name, tns)
            quant(m)
            got = host(m.forward(*outlier["args"]))
            rows.append((label, rel_l2(got, ref), bool(np.isfinite(got).all())))
        finally:
            m.close()
    for label, err, fin in rows:
        print(f"OUTLIER PROFILE, {label}: rel-L2 vs the f32 oracle {err:.3e}{'' if fin else '  (NON-FINITE VALUES)'}")
    assert all(fin for _, _, fin in rows)
    outlier["int8_plain"], outlier["int8"], outlier["fp8"] = rows[0][1], rows[1][1], rows[2][1]
    assert rows[1][1] <= 3e-2, "the smoothed int8 mode leaves the 8-bit bar on the outlier-channel checkpoint"
