#!/usr/bin/env python
"""Generate tests/golden/torch_rederivation.npz — an INDEPENDENT PyTorch-CPU (float64)
re-derivation of the model-level semantics the reference has no test for (SURVEY §8c "unpinned"):
Flux::forward, the Euler sampler and the VAE decoder, written against torch.nn.functional
(layer_norm, scaled_dot_product_attention, gelu(tanh), conv2d, group_norm, interpolate) rather
than the op sequence of the oracle.  Weights/inputs come from seeded generators in
diffusion-rs_amd/synth.py and tests/util.py, so the fixture stores only seeds + expected outputs.
Run:  python tests/golden/gen_torch_fixtures.py     (CPU, a few seconds; no reference code involved —
the reference is Rust and cannot be imported.)"""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import diffusion_rs_amd as d  # noqa: E402
from tests.util import SMALL_FLUX, SMALL_VAE, flux_inputs  # noqa: E402

T64 = torch.float64


def lin(sd, p, x):
    return F.linear(x, sd[p + ".weight"], sd[p + ".bias"])


def rope_pe(ids, axes, theta):
    outs = []
    for a, dim in enumerate(axes):
        i = torch.arange(0, dim, 2, dtype=T64)
        inv = (1.0 / (float(theta) ** (i / dim))).to(torch.float32).to(T64)  # f32 inv_freq as model.rs:73
        outs.append(ids[..., a:a + 1].to(T64) * inv)
    f = torch.cat(outs, -1)  # (B, L, 64)
    return torch.cos(f), torch.sin(f)


def apply_rope(x, cos, sin):  # x (B,H,L,128)
    x0, x1 = x[..., 0::2], x[..., 1::2]
    c, s = cos[:, None], sin[:, None]
    return torch.stack([c * x0 - s * x1, s * x0 + c * x1], -1).flatten(-2)


def rms(x, w):
    return x / torch.sqrt((x * x).mean(-1, keepdim=True) + 1e-6) * w


def heads(x, H):
    B, L, D = x.shape
    return x.view(B, L, H, D // H).transpose(1, 2)


def attn(q, k, v, cos, sin):
    o = F.scaled_dot_product_attention(apply_rope(q, cos, sin), apply_rope(k, cos, sin), v)
    return o.transpose(1, 2).flatten(2)


def temb(t, dim=256):
    half = dim // 2
    fr = torch.exp(torch.arange(half, dtype=torch.float32) * np.float32(-np.log(10000.0) / half)).to(T64)
    a = (t.to(torch.float32) * 1000.0).to(T64)[:, None] * fr[None]
    return torch.cat([torch.cos(a), torch.sin(a)], -1)


def mlp_emb(sd, p, x):
    return lin(sd, p + ".linear_2", F.silu(lin(sd, p + ".linear_1", x)))


def flux_forward(sd, cfg, img, img_ids, txt, txt_ids, t, y, g):
    H = cfg["num_attention_heads"]
    D = H * 128
    cos, sin = rope_pe(torch.cat([txt_ids, img_ids], 1), cfg["axes_dim"], cfg["theta"])
    txt = lin(sd, "context_embedder", txt)
    img = lin(sd, "x_embedder", img)
    vec = mlp_emb(sd, "time_text_embed.timestep_embedder", temb(t))
    if cfg["guidance_embeds"]:
        vec = vec + mlp_emb(sd, "time_text_embed.guidance_embedder", temb(g))
    vec = vec + mlp_emb(sd, "time_text_embed.text_embedder", y)
    sv = F.silu(vec)
    Tn = txt.shape[1]
    ln = lambda x: F.layer_norm(x, (D,), eps=1e-6)
    for i in range(cfg["num_layers"]):
        p = f"transformer_blocks.{i}."
        im = lin(sd, p + "norm1.linear", sv)[:, None].chunk(6, -1)
        tm = lin(sd, p + "norm1_context.linear", sv)[:, None].chunk(6, -1)
        xi = ln(img) * (1 + im[1]) + im[0]
        xt = ln(txt) * (1 + tm[1]) + tm[0]
        qi, ki, vi = (heads(lin(sd, p + "attn." + n, xi), H) for n in ("to_q", "to_k", "to_v"))
        qt, kt, vt = (heads(lin(sd, p + "attn." + n, xt), H) for n in ("add_q_proj", "add_k_proj", "add_v_proj"))
        qi, ki = rms(qi, sd[p + "attn.norm_q.weight"]), rms(ki, sd[p + "attn.norm_k.weight"])
        qt, kt = rms(qt, sd[p + "attn.norm_added_q.weight"]), rms(kt, sd[p + "attn.norm_added_k.weight"])
        a = attn(torch.cat([qt, qi], 2), torch.cat([kt, ki], 2), torch.cat([vt, vi], 2), cos, sin)
        at, ai = a[:, :Tn], a[:, Tn:]
        img = img + im[2] * lin(sd, p + "attn.to_out.0", ai)
        img = img + im[5] * lin(sd, p + "ff.net.2", F.gelu(lin(sd, p + "ff.net.0.proj", ln(img) * (1 + im[4]) + im[3]), approximate="tanh"))
        txt = txt + tm[2] * lin(sd, p + "attn.to_add_out", at)
        txt = txt + tm[5] * lin(sd, p + "ff_context.net.2", F.gelu(lin(sd, p + "ff_context.net.0.proj", ln(txt) * (1 + tm[4]) + tm[3]), approximate="tanh"))
    x = torch.cat([txt, img], 1)
    for i in range(cfg["num_single_layers"]):
        p = f"single_transformer_blocks.{i}."
        m = lin(sd, p + "norm.linear", sv)[:, None].chunk(3, -1)
        xm = ln(x) * (1 + m[1]) + m[0]
        q, k, v = (heads(lin(sd, p + "attn." + n, xm), H) for n in ("to_q", "to_k", "to_v"))
        q, k = rms(q, sd[p + "attn.norm_q.weight"]), rms(k, sd[p + "attn.norm_k.weight"])
        a = attn(q, k, v, cos, sin)
        mlp = F.gelu(lin(sd, p + "proj_mlp", xm), approximate="tanh")
        x = x + m[2] * lin(sd, p + "proj_out", torch.cat([a, mlp], -1))
    x = x[:, Tn:]
    sc, sh = lin(sd, "norm_out.linear", sv)[:, None].chunk(2, -1)
    return lin(sd, "proj_out", ln(x) * (1 + sc) + sh)


def vae_decode(sd, cfg, z):
    G = cfg["norm_num_groups"]
    conv = lambda p, x, pad: F.conv2d(x, sd[p + ".weight"], sd[p + ".bias"], padding=pad)
    gn = lambda p, x: F.group_norm(x, G, sd[p + ".weight"], sd[p + ".bias"], eps=1e-6)

    def res(p, x):
        h = conv(p + ".conv1", F.silu(gn(p + ".norm1", x)), 1)
        h = conv(p + ".conv2", F.silu(gn(p + ".norm2", h)), 1)
        return (conv(p + ".conv_shortcut", x, 0) if (p + ".conv_shortcut.weight") in sd else x) + h

    x = conv("decoder.conv_in", z, 1)
    x = res("decoder.mid_block.resnets.0", x)
    p = "decoder.mid_block.attentions.0"
    B, C, Hh, Ww = x.shape
    t = gn(p + ".group_norm", x).flatten(2).transpose(1, 2)
    q, k, v = (F.linear(t, sd[f"{p}.{n}.weight"], sd[f"{p}.{n}.bias"]) for n in ("to_q", "to_k", "to_v"))
    o = F.scaled_dot_product_attention(q[:, None], k[:, None], v[:, None])[:, 0]
    o = F.linear(o, sd[p + ".to_out.0.weight"], sd[p + ".to_out.0.bias"])
    x = x + o.transpose(1, 2).reshape(B, C, Hh, Ww)
    x = res("decoder.mid_block.resnets.1", x)
    for lvl in range(4):
        for i in range(cfg["layers_per_block"] + 1):
            x = res(f"decoder.up_blocks.{lvl}.resnets.{i}", x)
        if lvl != 3:
            x = conv(f"decoder.up_blocks.{lvl}.upsamplers.0.conv", F.interpolate(x, scale_factor=2, mode="nearest"), 1)
    return conv("decoder.conv_out", F.silu(gn("decoder.conv_norm_out", x)), 1)


def main():
    out = {}
    sd = {k: torch.from_numpy(v).to(T64) for k, v in d.synth.flux_state_dict_numpy(SMALL_FLUX, seed=0).items()}
    img, ids, txt, txt_ids, y = flux_inputs(SMALL_FLUX, 1, (4, 6), 8, seed=42)
    t = np.array([0.65], np.float32)
    g = np.array([3.5], np.float32)
    a = [torch.from_numpy(v).to(T64) for v in (img, ids, txt, txt_ids)]
    pred = flux_forward(sd, SMALL_FLUX, a[0], a[1], a[2], a[3], torch.from_numpy(t), torch.from_numpy(y).to(T64), torch.from_numpy(g))
    out["flux_pred"] = pred.numpy().astype(np.float32)
    # 3 Euler steps with the dev schedule (sampling.rs:37-44)
    ts = d.SchedulerConfig().get_timesteps(3, d.SchedulerConfig().calculate_shift(24))
    x = a[0].clone()
    for tc, tp in zip(ts, ts[1:]):
        pr = flux_forward(sd, SMALL_FLUX, x, a[1], a[2], a[3], torch.tensor([np.float32(tc)]), torch.from_numpy(y).to(T64), torch.from_numpy(g))
        x = x + pr * float(np.float32(tp - tc))
    out["flux_denoised"] = x.numpy().astype(np.float32)
    out["timesteps"] = np.array(ts)
    vsd = {k: torch.from_numpy(v).to(T64) for k, v in d.synth.vae_state_dict_numpy(SMALL_VAE, seed=0).items()}
    z = np.random.default_rng(5).standard_normal((1, 16, 4, 4)).astype(np.float32)
    out["vae_image"] = vae_decode(vsd, SMALL_VAE, torch.from_numpy(z).to(T64)).numpy().astype(np.float32)
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "torch_rederivation.npz"), **out)
    print({k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
