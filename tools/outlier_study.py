#!/usr/bin/env python3
"""Which piece of the outlier-channel profile (diffusion-rs_amd/synth.py) costs which mode what?  (GPU + the oracle on the box's host cores.)

FLUX.1-dev at full width and depth, one Flux::forward at 1024 + 128 tokens against the f32 oracle on the same weights, for several subsets of the profile
(AdaLN scale outliers "mod", the reading columns x 1/8 "cols", massive residual channels "res", QkNorm x gain "qk") and per mode: bf16; bf16 + e4m3 q / k;
int8 unsmoothed / smoothed (calibrated on a DIFFERENT sample: other latents, text and pooled vector, timesteps 1 / .75 / .5 / .25), each with bf16 and with e4m3 attention
operands; e4m3.

    python tools/outlier_study.py [--variants all] > profiles/r06_outlier_study.txt
"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tokens", default="32x32")
    ap.add_argument("--txt", type=int, default=128)
    ap.add_argument("--variants", default="none;mod,cols;mod,cols,res;qk:8;qk:3;mod,cols,res,qk:8;mod,cols,res,qk:3")
    a = ap.parse_args()
    import torch
    import diffusion_rs_amd as d
    from oracle import oracle as orc
    from tests.util import host, rel_l2
    S = d.synth
    cfg = dict(d.FLUX_DEV)
    D = 3072
    h2, w2 = (int(v) for v in a.tokens.split("x"))
    lat = S.exact_tensor_device("input.outlier.latent", (1, 16, 2 * h2, 2 * w2), "input").float()
    t5 = S.exact_tensor_device("input.outlier.t5", (1, a.txt, cfg["joint_attention_dim"]), "input")
    clip = S.exact_tensor_device("input.outlier.clip", (1, cfg["pooled_projection_dim"]), "input").float()
    img, ids = d.pack_latents(lat)
    txt_ids = torch.zeros((1, a.txt, 3), device="cuda")
    t = torch.tensor([0.6], device="cuda")
    g = torch.tensor([3.5], device="cuda")
    args = (img, ids, t5, txt_ids, t, clip, g)
    # the calibration sees ANOTHER sample (latents, text, pooled vector) at four other timesteps: the statistics must carry over
    c_img, _ = d.pack_latents(S.exact_tensor_device("input.calib.latent", (1, 16, 2 * h2, 2 * w2), "input").float())
    c_t5 = S.exact_tensor_device("input.calib.t5", (1, a.txt, cfg["joint_attention_dim"]), "input")
    c_clip = S.exact_tensor_device("input.calib.clip", (1, cfg["pooled_projection_dim"]), "input").float()
    print(f"# FLUX.1-dev 19 + 38 blocks, one Flux::forward at {h2 * w2} + {a.txt} tokens, rel-L2 vs the f32 oracle on the same weights")
    print(f"# {'profile':28s} {'bf16':>9s} {'bf16+qk8':>9s} {'i8':>9s} {'i8 qk-bf16':>10s} {'i8 smooth':>9s} {'i8s qk-bf16':>11s} {'e4m3':>9s} {'e4m3 qk-bf16':>12s}")
    for var in a.variants.split(";"):
        parts, gain = [], 8.0
        for tok in var.split(","):
            if tok.startswith("qk"):
                parts.append("qk")
                gain = float(tok.split(":")[1]) if ":" in tok else 8.0
            elif tok != "none":
                parts.append(tok)

        def tensors():
            for name, shape in S.flux_tensor_shapes(cfg).items():
                yield name, S.apply_outlier_profile(name, S.exact_tensor_device(name, shape, "flux", salt=3), D, parts=tuple(parts), qk_gain=gain)

        om = orc.Flux(cfg)
        for name, tn in tensors():
            om.set_tensor_bf16(name, tn.view(torch.int16).cpu().numpy().view(np.uint16))
        ref = om.forward(host(img), host(ids), host(t5), host(txt_ids), host(t), host(clip), host(g))
        del om
        row = []

        def run(prep):
            m = d.FluxModel(cfg)
            try:
                for name, tn in tensors():
                    m.set_tensor(name, tn)
                prep(m)
                return rel_l2(host(m.forward(*args)), ref)
            finally:
                m.close()

        def calib(m):
            m.calibrate_int8(True)
            for tt in (1.0, 0.75, 0.5, 0.25):
                m.forward(c_img, ids, c_t5, txt_ids, torch.tensor([tt], device="cuda"), c_clip, g)

        row.append(run(lambda m: None))
        row.append(run(lambda m: m.set_fp8_attention(2)))
        row.append(run(lambda m: m.quantize_int8()))
        row.append(run(lambda m: (m.set_fp8_attention(0), m.quantize_int8())))
        row.append(run(lambda m: (calib(m), m.quantize_int8())))
        row.append(run(lambda m: (m.set_fp8_attention(0), calib(m), m.quantize_int8())))
        row.append(run(lambda m: m.quantize_fp8()))
        row.append(run(lambda m: (m.set_fp8_attention(0), m.quantize_fp8())))
        print(f"  {var:28s} " + " ".join(f"{e:9.3e}" for e in row) + f"   |ref| rms {float(np.sqrt((ref.astype(np.float64) ** 2).mean())):.3f}", flush=True)


if __name__ == "__main__":
    main()
