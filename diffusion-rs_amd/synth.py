"""Synthetic FLUX / VAE weights (there are no checkpoints offline).

`flux_tensor_shapes` / `vae_tensor_shapes` enumerate exactly the diffusers tensor names the
reference's VarBuilder reads (model.rs:165-772, vae.rs:371-433) with their shapes, so the same
dictionaries drive the HIP library, the CPU oracle and the independent torch re-derivation.
"""
from collections import OrderedDict

import numpy as np


def flux_tensor_shapes(cfg) -> "OrderedDict[str, tuple]":
    D = cfg["num_attention_heads"] * sum(cfg["axes_dim"])
    M = 4 * D
    C = cfg["in_channels"]
    hd = sum(cfg["axes_dim"])
    t = OrderedDict()

    def lin(p, o, i):
        t[p + ".weight"] = (o, i)
        t[p + ".bias"] = (o,)

    lin("x_embedder", D, C)
    lin("context_embedder", D, cfg["joint_attention_dim"])
    lin("time_text_embed.timestep_embedder.linear_1", D, 256)
    lin("time_text_embed.timestep_embedder.linear_2", D, D)
    if cfg["guidance_embeds"]:
        lin("time_text_embed.guidance_embedder.linear_1", D, 256)
        lin("time_text_embed.guidance_embedder.linear_2", D, D)
    lin("time_text_embed.text_embedder.linear_1", D, cfg["pooled_projection_dim"])
    lin("time_text_embed.text_embedder.linear_2", D, D)
    for i in range(cfg["num_layers"]):
        p = f"transformer_blocks.{i}."
        lin(p + "norm1.linear", 6 * D, D)
        lin(p + "norm1_context.linear", 6 * D, D)
        for n in ("to_q", "to_k", "to_v", "add_q_proj", "add_k_proj", "add_v_proj", "to_out.0", "to_add_out"):
            lin(p + "attn." + n, D, D)
        for n in ("norm_q", "norm_k", "norm_added_q", "norm_added_k"):
            t[p + "attn." + n + ".weight"] = (hd,)
        lin(p + "ff.net.0.proj", M, D)
        lin(p + "ff.net.2", D, M)
        lin(p + "ff_context.net.0.proj", M, D)
        lin(p + "ff_context.net.2", D, M)
    for i in range(cfg["num_single_layers"]):
        p = f"single_transformer_blocks.{i}."
        lin(p + "norm.linear", 3 * D, D)
        for n in ("to_q", "to_k", "to_v"):
            lin(p + "attn." + n, D, D)
        t[p + "attn.norm_q.weight"] = (hd,)
        t[p + "attn.norm_k.weight"] = (hd,)
        lin(p + "proj_mlp", M, D)
        lin(p + "proj_out", D, D + M)
    lin("norm_out.linear", 2 * D, D)
    lin("proj_out", C, D)
    return t


def _std_for(name, w_std, mod_std):
    if "norm1.linear" in name or "norm1_context.linear" in name or ".norm.linear" in name or name.startswith("norm_out.linear"):
        return mod_std
    return w_std


def flux_state_dict_numpy(cfg, seed=0, w_std=0.02, mod_std=0.01, bias_std=0.02, norm_jitter=0.1, round_bf16=True):
    """Small-config weights for parity tests: Linear W~N(0,w_std^2) (modulation linears mod_std so
    1+scale stays near 1 and gates small), biases N(0,bias_std^2), QkNorm weights 1+N(0,norm_jitter^2).
    Values are rounded to bf16-representable f32 so oracle and GPU see identical weights."""
    rng = np.random.default_rng(seed)
    out = OrderedDict()
    for name, shape in flux_tensor_shapes(cfg).items():
        if name.endswith("norm_q.weight") or name.endswith("norm_k.weight") or name.endswith("norm_added_q.weight") or name.endswith("norm_added_k.weight"):
            a = 1.0 + norm_jitter * rng.standard_normal(shape)
        elif name.endswith(".bias"):
            a = bias_std * rng.standard_normal(shape)
        else:
            a = _std_for(name, w_std, mod_std) * rng.standard_normal(shape)
        a = a.astype(np.float32)
        out[name] = to_bf16_f32(a) if round_bf16 else a
    return out


def to_bf16_f32(a):
    """Round f32 to the nearest bf16 (RNE) and return as f32."""
    a = np.ascontiguousarray(a, np.float32)
    u = a.view(np.uint32).astype(np.uint64)
    r = ((u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000).astype(np.uint32)
    return r.view(np.float32).reshape(a.shape)


def fill_flux_random_device(model, seed=0, w_std=0.02, mod_std=0.01, bias_std=0.0, device="cuda"):
    """Full-size random weights generated on the GPU (bench): torch.randn is only the RNG here."""
    import torch
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    for name, shape in flux_tensor_shapes(model.cfg).items():
        if "norm_q.weight" in name or "norm_k.weight" in name or "norm_added" in name:
            t = torch.ones(shape, dtype=torch.bfloat16, device=device)
        elif name.endswith(".bias"):
            t = (torch.randn(shape, generator=g, device=device, dtype=torch.float32) * bias_std).to(torch.bfloat16) if bias_std else torch.zeros(
                shape, dtype=torch.bfloat16, device=device)
        else:
            t = torch.randn(shape, generator=g, device=device, dtype=torch.bfloat16)
            t.mul_(_std_for(name, w_std, mod_std))
        model.set_tensor(name, t)
        del t
    model.assert_complete()


def vae_tensor_shapes(cfg, encoder=False) -> "OrderedDict[str, tuple]":
    """Decoder tensors (vae.rs:371-433); with encoder=True also the encoder's (vae.rs:249-327) and quant_conv."""
    boc = list(cfg["block_out_channels"])
    t = OrderedDict()

    def conv(p, o, i, k):
        t[p + ".weight"] = (o, i, k, k)
        t[p + ".bias"] = (o,)

    def gn(p, c):
        t[p + ".weight"] = (c,)
        t[p + ".bias"] = (c,)

    def resnet(p, i, o):
        gn(p + ".norm1", i)
        conv(p + ".conv1", o, i, 3)
        gn(p + ".norm2", o)
        conv(p + ".conv2", o, o, 3)
        if i != o:
            conv(p + ".conv_shortcut", o, i, 1)

    block_in = boc[-1]
    conv("decoder.conv_in", block_in, cfg["latent_channels"], 3)
    resnet("decoder.mid_block.resnets.0", block_in, block_in)
    if cfg["mid_block_add_attention"]:
        p = "decoder.mid_block.attentions.0"
        gn(p + ".group_norm", block_in)
        for n in ("to_q", "to_k", "to_v", "to_out.0"):
            t[f"{p}.{n}.weight"] = (block_in, block_in)
            t[f"{p}.{n}.bias"] = (block_in,)
    resnet("decoder.mid_block.resnets.1", block_in, block_in)
    for lvl, block_out in enumerate(reversed(boc)):
        for i in range(cfg["layers_per_block"] + 1):
            resnet(f"decoder.up_blocks.{lvl}.resnets.{i}", block_in, block_out)
            block_in = block_out
        if lvl != 3:
            conv(f"decoder.up_blocks.{lvl}.upsamplers.0.conv", block_in, block_in, 3)
    gn("decoder.conv_norm_out", boc[0])
    conv("decoder.conv_out", cfg["out_channels"], boc[0], 3)
    if encoder:
        ch = boc[0]
        conv("encoder.conv_in", ch, cfg["in_channels"], 3)
        for lvl, block_out in enumerate(boc):
            for i in range(cfg["layers_per_block"]):
                resnet(f"encoder.down_blocks.{lvl}.resnets.{i}", ch, block_out)
                ch = block_out
            if lvl != len(boc) - 1:
                conv(f"encoder.down_blocks.{lvl}.downsamplers.0.conv", ch, ch, 3)
        resnet("encoder.mid_block.resnets.0", ch, ch)
        if cfg["mid_block_add_attention"]:
            p = "encoder.mid_block.attentions.0"
            gn(p + ".group_norm", ch)
            for n in ("to_q", "to_k", "to_v", "to_out.0"):
                t[f"{p}.{n}.weight"] = (ch, ch)
                t[f"{p}.{n}.bias"] = (ch,)
        resnet("encoder.mid_block.resnets.1", ch, ch)
        gn("encoder.conv_norm_out", ch)
        conv("encoder.conv_out", 2 * cfg["latent_channels"], ch, 3)
        if cfg.get("use_quant_conv", False):
            conv("quant_conv", 2 * cfg["latent_channels"], 2 * cfg["latent_channels"], 1)
    return t


def vae_state_dict_numpy(cfg, seed=0, round_bf16=True, encoder=False):
    """Conv W~N(0, 1/fan_in), small biases, GroupNorm w = 1+0.1N, b = 0.1N.  Decoder tensors come
    first, so encoder=True leaves their values unchanged."""
    rng = np.random.default_rng(seed)
    out = OrderedDict()
    for name, shape in vae_tensor_shapes(cfg, encoder).items():
        if len(shape) == 4 or len(shape) == 2:
            fan_in = int(np.prod(shape[1:]))
            a = rng.standard_normal(shape) / np.sqrt(fan_in)
        elif "norm" in name and name.endswith(".weight"):
            a = 1.0 + 0.1 * rng.standard_normal(shape)
        else:
            a = 0.1 * rng.standard_normal(shape) if "norm" in name else 0.02 * rng.standard_normal(shape)
        a = a.astype(np.float32)
        out[name] = to_bf16_f32(a) if round_bf16 else a
    return out


def fill_vae_random_device(vae, seed=0, device="cuda", encoder=False):
    import torch
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    for name, shape in vae_tensor_shapes(vae.cfg, encoder).items():
        if len(shape) in (2, 4):
            fan_in = 1
            for s in shape[1:]:
                fan_in *= s
            t = torch.randn(shape, generator=g, device=device, dtype=torch.float32) / (fan_in ** 0.5)
        elif "norm" in name and name.endswith(".weight"):
            t = torch.ones(shape, device=device, dtype=torch.float32)
        else:
            t = torch.zeros(shape, device=device, dtype=torch.float32)
        vae.set_tensor(name, t)
    return vae


# ------------------------------------------------------------------------------- exact synthetic tensors
# Synthetic tensors that are the SAME BITS wherever they are generated (the build container's CPU, the GPU box's device): a committed
# fixture computed by the CPU oracle in one place can then be compared with the HIP path in another without shipping 24 GB of weights.
# Element e of tensor `name` = offset + scale * z_e, z_e = (b0 + b1 + b2 + b3 - 510) / sqrt(21845) from the four bytes of word e of the
# Philox4x32-10 stream seeded by the tensor's name (integer work: bit-exact on every machine; the sum of four uniform bytes has mean 510,
# variance 4 (256^2 - 1) / 12 = 21845 and is Gaussian to ~3.4 sigma), evaluated as ONE f32 multiply by f32(scale / sqrt(21845)) and one f32
# add, then rounded to bf16 (RNE).  The raw words come from the caller: the library's fmi_philox_u32 on the device, the oracle's
# restatement on the CPU (tests/ only) — tests/test_gpu_philox.py holds them bit-equal.
EXACT_SALT = 0x46495854  # "FIXT"
_IH_VAR = 21845.0


def exact_seed(name, salt=0):
    import zlib
    return (((EXACT_SALT + int(salt)) & 0xFFFFFFFF) << 32) | zlib.crc32(name.encode())


def exact_rule(name, shape, family="flux", w_std=0.02, mod_std=0.01):
    """(offset, scale) of a tensor of the synthetic checkpoint, the rules of flux_state_dict_numpy / vae_state_dict_numpy."""
    if family == "flux":
        if name.endswith(("norm_q.weight", "norm_k.weight", "norm_added_q.weight", "norm_added_k.weight")):
            return 1.0, 0.1
        if name.endswith(".bias"):
            return 0.0, 0.02
        return 0.0, _std_for(name, w_std, mod_std)
    if family == "vae":
        if len(shape) in (2, 4):
            return 0.0, 1.0 / float(np.sqrt(int(np.prod(shape[1:]))))
        if "norm" in name and name.endswith(".weight"):
            return 1.0, 0.1
        return 0.0, (0.1 if "norm" in name else 0.02)
    if family == "input":  # N(0, 1) activations (latent noise, text embeddings)
        return 0.0, 1.0
    raise ValueError(family)


def exact_coeff(scale):
    return np.float32(float(scale) / float(np.sqrt(_IH_VAR)))


def exact_values_np(words, offset, scale):
    """u32 words -> f32 values that are bf16-representable (numpy side of the definition above)."""
    w = np.ascontiguousarray(words, np.uint32)
    s = ((w & np.uint32(0xFF)) + ((w >> np.uint32(8)) & np.uint32(0xFF)) + ((w >> np.uint32(16)) & np.uint32(0xFF)) + (w >> np.uint32(24))).astype(np.int32)
    v = (s - np.int32(510)).astype(np.float32) * exact_coeff(scale)
    if offset:
        v = v + np.float32(offset)
    return to_bf16_f32(v)


def exact_tensor_np(name, shape, raw_u32, family="flux", salt=0, **kw):
    """raw_u32(n, seed) -> (n,) u32 (tests pass the oracle's Philox).  Returns f32 of `shape`, every value bf16-representable."""
    n = int(np.prod(shape))
    off, sc = exact_rule(name, shape, family, **kw)
    return exact_values_np(raw_u32(n, exact_seed(name, salt)), off, sc).reshape(shape)


def exact_tensor_device(name, shape, family="flux", salt=0, device="cuda", **kw):
    """The same tensor on the GPU as bf16: words from fmi_philox_u32, the byte sum / multiply / add as separate torch ops (no contraction)."""
    import ctypes as C
    import torch
    from . import _lib as L
    lib = L.load()
    n = 1
    for d in shape:
        n *= int(d)
    off, sc = exact_rule(name, shape, family, **kw)
    w = torch.empty(n, dtype=torch.int32, device=device)
    L.check(lib.fmi_philox_u32(C.c_void_p(w.data_ptr()), n, 1, exact_seed(name, salt), 0, None))
    s = (w & 0xFF) + ((w >> 8) & 0xFF) + ((w >> 16) & 0xFF) + ((w >> 24) & 0xFF)
    del w
    v = (s - 510).to(torch.float32)
    del s
    v.mul_(float(exact_coeff(sc)))
    if off:
        v.add_(float(np.float32(off)))
    return v.to(torch.bfloat16).reshape(tuple(shape))


# ------------------------------------------------------------------------------- outlier-channel profile
# Real DiT checkpoints are not N(0, 0.02^2): the AdaLN modulation gives a handful of hidden channels a (1 + scale) of 30-100, so the operand of the
# q|k|v / MLP-in / linear1 GEMMs has a few channels two orders of magnitude above the rest; a couple of residual-stream channels carry "massive
# activations"; QkNorm weights have a few large dimensions.  Per-token 8-bit grids are exactly what such channels break.  This
# profile turns any synthetic checkpoint (numpy arrays or torch tensors: plain slicing and in-place arithmetic) into one with those statistics:
#   * OUT: 12 of the D hidden channels (0.39 %) with amplitudes 30 .. 100, alternating sign — added to the bias of the SCALE rows of every modulation
#     linear of the blocks (norm1 / norm1_context: rows D..2D and 4D..5D of [shift, scale, gate] x 2; single .norm: rows D..2D; model.rs:211-300), so
#     LN(x) * (1 + scale) + shift has those channels at 30-100x the others in every block and at every timestep;
#   * the weight columns that READ those channels (to_q/k/v, add_q/k/v_proj, ff.net.0.proj, ff_context.net.0.proj, single to_q/k/v, proj_mlp) x 1/8: an
#     outlier channel still contributes ~4-12x a normal one to an output (it stays an important channel), the outputs stay O(1);
#   * RES: 2 residual-stream channels with +-30 in the bias of x_embedder / context_embedder (massive activations: LayerNorm statistics are dominated by them);
#   * QkNorm weights (norm_q / norm_k / norm_added_q / norm_added_k): dims 5, 77, 100 of the 128 x 3 (q . k then weighs them 9x).  (x 8 — 64x in the score — makes
#     the softmax so peaked that the PROBLEM is ill-conditioned: the bf16 path alone is 0.42 from f32 on it, profiles/r06_outlier_study.txt; not a recipe matter.)
OUTLIER_SEED = 20260930


def outlier_channels(D):
    rng = np.random.default_rng(OUTLIER_SEED)
    ch = np.sort(rng.choice(D, 14, replace=False))
    out, res = ch[:12], ch[12:]
    amp = np.round(np.geomspace(30.0, 100.0, 12)) * np.where(np.arange(12) % 2 == 0, 1.0, -1.0)
    amp = amp[rng.permutation(12)]
    return [int(c) for c in out], [float(a) for a in amp], [int(c) for c in res]


def outlier_profile_touches(name):
    """does apply_outlier_profile edit this tensor at all?  (the full-size fixture generator skips the f32 round trip for the 11.5e9 weights it leaves alone)"""
    return (name.endswith(("norm1.linear.bias", "norm1_context.linear.bias")) or (".norm.linear.bias" in name and name.startswith("single_transformer_blocks."))
            or any(name.endswith(k + ".weight") for k in ("attn.to_q", "attn.to_k", "attn.to_v", "attn.add_q_proj", "attn.add_k_proj", "attn.add_v_proj", "ff.net.0.proj",
                                                          "ff_context.net.0.proj", "proj_mlp"))
            or name in ("x_embedder.bias", "context_embedder.bias") or name.endswith(("norm_q.weight", "norm_k.weight", "norm_added_q.weight", "norm_added_k.weight")))


def apply_outlier_profile(name, t, D, parts=("mod", "cols", "res", "qk"), qk_gain=3.0):
    """In-place on a numpy array or torch tensor holding tensor `name` of a FLUX checkpoint (already at its final dtype); returns t.
    `parts` / `qk_gain`: the pieces of the profile one at a time (tools/outlier_study.py)."""
    out, amp, res = outlier_channels(D)
    if "mod" not in parts:
        amp = [0.0] * len(amp)
    mod_double = name.endswith(("norm1.linear.bias", "norm1_context.linear.bias"))
    mod_single = ".norm.linear.bias" in name and name.startswith("single_transformer_blocks.")
    if mod_double or mod_single:
        for base in ((D, 4 * D) if mod_double else (D,)):
            for c, a in zip(out, amp):
                t[base + c] += a
    elif name.endswith(".weight") and any(name.endswith(k + ".weight") for k in (
            "attn.to_q", "attn.to_k", "attn.to_v", "attn.add_q_proj", "attn.add_k_proj", "attn.add_v_proj", "ff.net.0.proj", "ff_context.net.0.proj", "proj_mlp")):
        if "cols" in parts:
            for c in out:
                t[:, c] *= 0.125
    elif name in ("x_embedder.bias", "context_embedder.bias"):
        if "res" in parts:
            t[res[0]] += 30.0
            t[res[1]] -= 30.0
    elif name.endswith(("norm_q.weight", "norm_k.weight", "norm_added_q.weight", "norm_added_k.weight")):
        if "qk" in parts:
            for i in (5, 77, 100):
                t[i] *= qk_gain
    return t


NF4_CODE = [-1.0, -0.6961928009986877, -0.5250730514526367, -0.39491748809814453, -0.28444138169288635, -0.18477343022823334,
            -0.09105003625154495, 0.0, 0.07958029955625534, 0.16093020141124725, 0.24611230194568634, 0.33791524171829224,
            0.44070982933044434, 0.5626170039176941, 0.7229568362236023, 1.0]


def quantize_nf4_device(w, blocksize=64):
    """bitsandbytes-rule nf4 quantisation on the GPU (synthetic C3 weights): absmax per block, nearest
    code, high nibble first.  Returns (packed u8 (n/2), absmax f32 (n/blocksize))."""
    import torch
    flat = w.reshape(-1, blocksize).float()
    absmax = flat.abs().amax(1)
    xn = flat / absmax.clamp_min(1e-30)[:, None]
    code = torch.tensor(NF4_CODE, device=w.device)
    mid = (code[1:] + code[:-1]) / 2
    idx = torch.bucketize(xn, mid).to(torch.uint8).reshape(-1)
    packed = (idx[0::2] << 4) | idx[1::2]
    return packed.contiguous(), absmax.contiguous()


def is_block_linear(name):
    """Linears of the DiT blocks that take the fused 4-bit GEMM path (everything a bnb checkpoint
    quantises inside transformer_blocks / single_transformer_blocks except the modulation linears)."""
    return name.endswith(".weight") and "transformer_blocks." in name and "norm" not in name


# ------------------------------------------------------------------------------- text encoders
def t5_tensor_shapes(cfg) -> "OrderedDict[str, tuple]":
    """Tensor names/shapes T5EncoderModel::new reads (t5/mod.rs:614-627 and the loaders it calls)."""
    D, I, F = cfg["d_model"], cfg["num_heads"] * cfg["d_kv"], cfg["d_ff"]
    gated = cfg.get("feed_forward_proj", "relu") != "relu"
    out = OrderedDict()
    out["shared.weight"] = (cfg["vocab_size"], D)
    out["encoder.block.0.layer.0.SelfAttention.relative_attention_bias.weight"] = (cfg["relative_attention_num_buckets"], cfg["num_heads"])
    for l in range(cfg["num_layers"]):
        p = f"encoder.block.{l}.layer."
        out[p + "0.layer_norm.weight"] = (D,)
        for n in "qkv":
            out[p + f"0.SelfAttention.{n}.weight"] = (I, D)
        out[p + "0.SelfAttention.o.weight"] = (D, I)
        out[p + "1.layer_norm.weight"] = (D,)
        if gated:
            out[p + "1.DenseReluDense.wi_0.weight"] = (F, D)
            out[p + "1.DenseReluDense.wi_1.weight"] = (F, D)
        else:
            out[p + "1.DenseReluDense.wi.weight"] = (F, D)
        out[p + "1.DenseReluDense.wo.weight"] = (D, F)
    out["encoder.final_layer_norm.weight"] = (D,)
    return out


def clip_tensor_shapes(cfg) -> "OrderedDict[str, tuple]":
    """Tensor names/shapes ClipTextTransformer::new reads under vb.pp("text_model") (clip/text.rs:251-262)."""
    D, F = cfg["projection_dim"], cfg["intermediate_size"]
    out = OrderedDict()
    tm = "text_model."
    out[tm + "embeddings.token_embedding.weight"] = (cfg["vocab_size"], D)
    out[tm + "embeddings.position_embedding.weight"] = (cfg["max_position_embeddings"], D)
    for l in range(cfg["num_hidden_layers"]):
        p = tm + f"encoder.layers.{l}."
        for ln in ("layer_norm1", "layer_norm2"):
            out[p + ln + ".weight"] = (D,)
            out[p + ln + ".bias"] = (D,)
        for n in ("q_proj", "k_proj", "v_proj", "out_proj"):
            out[p + f"self_attn.{n}.weight"] = (D, D)
            out[p + f"self_attn.{n}.bias"] = (D,)
        out[p + "mlp.fc1.weight"] = (F, D)
        out[p + "mlp.fc1.bias"] = (F,)
        out[p + "mlp.fc2.weight"] = (D, F)
        out[p + "mlp.fc2.bias"] = (D,)
    out[tm + "final_layer_norm.weight"] = (D,)
    out[tm + "final_layer_norm.bias"] = (D,)
    return out


def text_state_dict_numpy(shapes, seed=0, round_bf16=True):
    """Seeded synthetic encoder weights: matrices ~N(0, 1/fan_in), embeddings ~N(0,1) (T5 relative
    bias ~N(0,1)), norm weights 1 + 0.1 N(0,1), biases 0.1 N(0,1); rounded to bf16 values."""
    rng = np.random.default_rng(seed)
    sd = OrderedDict()
    for name, shp in shapes.items():
        if len(shp) == 2 and ("embedding" in name or name == "shared.weight" or "relative_attention_bias" in name):
            a = rng.standard_normal(shp)
        elif len(shp) == 2:
            a = rng.standard_normal(shp) / np.sqrt(shp[1])
            if name.endswith("SelfAttention.q.weight"):
                a = a / 8.0  # T5 has no 1/sqrt(d_kv) in attention: its q init carries it (HF: std (d_model*d_kv)^-0.5)
        elif name.endswith("norm.weight") or "layer_norm1.weight" in name or "layer_norm2.weight" in name:
            a = 1.0 + 0.1 * rng.standard_normal(shp)
        else:
            a = 0.1 * rng.standard_normal(shp)
        a = a.astype(np.float32)
        sd[name] = to_bf16_f32(a) if round_bf16 else a
    return sd


def fill_text_random_device(model, seed=0, device="cuda"):
    """Random-init an encoder at full size directly on the GPU (T5-XXL: 4.7 B parameters)."""
    import torch
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    for name, shp in model.tensor_names().items():
        if len(shp) == 2 and ("embedding" in name or name == "shared.weight" or "relative_attention_bias" in name):
            t = torch.randn(shp, generator=g, device=device)
        elif len(shp) == 2:
            t = torch.randn(shp, generator=g, device=device) / shp[1] ** 0.5
            if name.endswith("SelfAttention.q.weight"):
                t = t / 8.0
        elif "norm" in name and name.endswith(".weight"):
            t = 1.0 + 0.1 * torch.randn(shp, generator=g, device=device)
        else:
            t = 0.1 * torch.randn(shp, generator=g, device=device)
        model.set_tensor(name, t.to(torch.bfloat16))
