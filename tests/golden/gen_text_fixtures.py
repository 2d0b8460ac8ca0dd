#!/usr/bin/env python
"""Generates tests/golden/text_encoders.npz: seeded random weights + token ids + the outputs of
HuggingFace transformers' T5EncoderModel / CLIPTextModel (torch CPU, float32) on them.

These are the implementations the reference says it follows (t5/mod.rs:3-4; clip/text.rs mirrors
modeling_clip.py), run here as an INDEPENDENT derivation that pins oracle/text_oracle.cpp — the
reference itself has no test for either encoder.  Only data is stored (inputs, weights, outputs).
Run in the authoring container:  python tests/golden/gen_text_fixtures.py
"""
import os

import numpy as np
import torch
from transformers import CLIPTextConfig, CLIPTextModel, T5Config, T5EncoderModel

T5_CFG = dict(vocab_size=96, d_model=64, d_kv=16, d_ff=128, num_layers=2, num_heads=4, relative_attention_num_buckets=32,
              relative_attention_max_distance=128, layer_norm_epsilon=1e-6, feed_forward_proj="gated-gelu")
CLIP_CFG = dict(vocab_size=120, projection_dim=64, intermediate_size=128, max_position_embeddings=32, num_hidden_layers=2, num_attention_heads=4)


def main():
    out = {}
    torch.manual_seed(0)
    t5 = T5EncoderModel(T5Config(**T5_CFG, is_encoder_decoder=False, use_cache=False, dropout_rate=0.0)).eval().float()
    with torch.no_grad():
        for n, p in t5.named_parameters():  # HF init is tiny for some tensors: use O(1/sqrt(fan_in)) weights, norm weights near 1
            if p.ndim == 2 and "relative_attention_bias" not in n and "shared" not in n:
                p.copy_(torch.randn_like(p) / p.shape[1] ** 0.5)
            elif p.ndim == 1:
                p.copy_(1.0 + 0.1 * torch.randn_like(p))
            else:
                p.copy_(torch.randn_like(p))
    g = torch.Generator().manual_seed(1)
    for tag, shape in (("short", (2, 24)), ("long", (1, 200))):  # "long" reaches the logarithmic buckets and the max-distance clamp
        ids = torch.randint(0, T5_CFG["vocab_size"], shape, generator=g)
        with torch.no_grad():
            h = t5(input_ids=ids).last_hidden_state
        out[f"t5_ids_{tag}"] = ids.numpy().astype(np.int32)
        out[f"t5_out_{tag}"] = h.numpy().astype(np.float32)
    sd = t5.state_dict()
    for k, v in sd.items():
        if k != "encoder.embed_tokens.weight":  # tied copy of shared.weight
            out["t5_w/" + k] = v.numpy().astype(np.float32)

    torch.manual_seed(2)
    cc = CLIPTextConfig(vocab_size=CLIP_CFG["vocab_size"], hidden_size=CLIP_CFG["projection_dim"], projection_dim=CLIP_CFG["projection_dim"],
                        intermediate_size=CLIP_CFG["intermediate_size"], max_position_embeddings=CLIP_CFG["max_position_embeddings"],
                        num_hidden_layers=CLIP_CFG["num_hidden_layers"], num_attention_heads=CLIP_CFG["num_attention_heads"], hidden_act="quick_gelu",
                        layer_norm_eps=1e-5, attention_dropout=0.0, eos_token_id=2)  # eos 2 = legacy pooling: argmax(input_ids), what the reference does
    clip = CLIPTextModel(cc).eval().float()
    with torch.no_grad():
        for n, p in clip.named_parameters():
            if p.ndim == 2 and "embedding" not in n:
                p.copy_(torch.randn_like(p) / p.shape[1] ** 0.5)
            elif "layer_norm" in n and n.endswith("weight"):
                p.copy_(1.0 + 0.1 * torch.randn_like(p))
            else:
                p.copy_(0.5 * torch.randn_like(p))
    ids = torch.randint(0, CLIP_CFG["vocab_size"] - 1, (3, 20), generator=g)
    eos = CLIP_CFG["vocab_size"] - 1
    for b, pos in enumerate((19, 7, 12)):  # EOS (largest id) then zero padding, as tokenize_and_pad produces
        ids[b, pos] = eos
        ids[b, pos + 1:] = 0
    with torch.no_grad():
        o = clip(input_ids=ids)
    out["clip_ids"] = ids.numpy().astype(np.int32)
    out["clip_hidden"] = o.last_hidden_state.numpy().astype(np.float32)
    out["clip_pooled"] = o.pooler_output.numpy().astype(np.float32)
    for k, v in clip.state_dict().items():
        if "position_ids" not in k:
            # checkpoint naming (what the reference's VarBuilder reads, flux/mod.rs:105: vb.pp("text_model")); newer
            # transformers flatten the wrapper module away
            out["clip_w/" + (k if k.startswith("text_model.") else "text_model." + k)] = v.numpy().astype(np.float32)
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "text_encoders.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes;", len(out), "arrays")
    print([k for k in out if k.startswith("clip_w/")][:6])


if __name__ == "__main__":
    main()
