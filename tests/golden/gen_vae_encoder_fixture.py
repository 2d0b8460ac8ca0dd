#!/usr/bin/env python
"""Generates tests/golden/torch_vae_encoder.npz: an independent PyTorch-CPU float64 derivation of
AutoEncoderKl::encode (Encoder::forward vaes/vae.rs:330-349 + Downsample :194-201 + DiagonalGaussian
:470-480) on the seeded synthetic SMALL_VAE weights; pins oracle/flux_oracle.cpp:orc_vae_encode.
Only inputs/outputs are stored.  Run in the authoring container: python tests/golden/gen_vae_encoder_fixture.py"""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import diffusion_rs_amd as d  # noqa: E402
from tests.util import SMALL_VAE  # noqa: E402

T64 = torch.float64


def vae_encode(sd, cfg, img, noise):
    G = cfg["norm_num_groups"]
    conv = lambda p, x, pad, stride=1: F.conv2d(x, sd[p + ".weight"], sd[p + ".bias"], padding=pad, stride=stride)
    gn = lambda p, x: F.group_norm(x, G, sd[p + ".weight"], sd[p + ".bias"], eps=1e-6)

    def res(p, x):
        h = conv(p + ".conv1", F.silu(gn(p + ".norm1", x)), 1)
        h = conv(p + ".conv2", F.silu(gn(p + ".norm2", h)), 1)
        return (conv(p + ".conv_shortcut", x, 0) if (p + ".conv_shortcut.weight") in sd else x) + h

    x = conv("encoder.conv_in", img, 1)
    nb = len(cfg["block_out_channels"])
    for lvl in range(nb):
        for i in range(cfg["layers_per_block"]):
            x = res(f"encoder.down_blocks.{lvl}.resnets.{i}", x)
        if lvl != nb - 1:
            x = conv(f"encoder.down_blocks.{lvl}.downsamplers.0.conv", F.pad(x, (0, 1, 0, 1)), 0, stride=2)
    x = res("encoder.mid_block.resnets.0", x)
    p = "encoder.mid_block.attentions.0"
    B, C, Hh, Ww = x.shape
    t = gn(p + ".group_norm", x).flatten(2).transpose(1, 2)
    q, k, v = (F.linear(t, sd[f"{p}.{n}.weight"], sd[f"{p}.{n}.bias"]) for n in ("to_q", "to_k", "to_v"))
    o = F.scaled_dot_product_attention(q[:, None], k[:, None], v[:, None])[:, 0]
    o = F.linear(o, sd[p + ".to_out.0.weight"], sd[p + ".to_out.0.bias"])
    x = x + o.transpose(1, 2).reshape(B, C, Hh, Ww)
    x = res("encoder.mid_block.resnets.1", x)
    mom = conv("encoder.conv_out", F.silu(gn("encoder.conv_norm_out", x)), 1)
    mean, logvar = mom.chunk(2, 1)
    return mom, mean + torch.exp(0.5 * logvar) * noise


def main():
    vsd = {k: torch.from_numpy(v).to(T64) for k, v in d.synth.vae_state_dict_numpy(SMALL_VAE, seed=0, encoder=True).items()}
    rng = np.random.default_rng(11)
    out = {}
    for tag, (H, W) in (("even", (32, 48)), ("odd", (24, 40))):  # "odd": an odd intermediate size (3 x 5 latent) exercises the one-sided padding
        img = rng.uniform(-1, 1, (1, 3, H, W)).astype(np.float32)
        h, w = H // 8, W // 8
        noise = rng.standard_normal((1, SMALL_VAE["latent_channels"], h, w)).astype(np.float32)
        mom, z = vae_encode(vsd, SMALL_VAE, torch.from_numpy(img).to(T64), torch.from_numpy(noise).to(T64))
        out[f"img_{tag}"], out[f"noise_{tag}"] = img, noise
        out[f"moments_{tag}"], out[f"z_{tag}"] = mom.numpy().astype(np.float32), z.numpy().astype(np.float32)
    path = os.path.join(ROOT, "tests", "golden", "torch_vae_encoder.npz")
    np.savez_compressed(path, **out)
    print({k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
