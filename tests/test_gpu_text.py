"""Text encoders on the GPU (fmi_t5_forward / fmi_clip_forward) vs the f32 CPU
oracle (oracle/text_oracle.cpp, itself pinned to HuggingFace transformers in test_oracle_text.py)
on identical synthetic weights and token ids.

Stated tolerance: rel-L2 <= 1e-2 on the encoder outputs (bf16 operands, f32 accumulate, f32
residual stream — same bar as one Flux::forward evaluation)."""
import numpy as np
import pytest

from tests.util import host, rel_l2

pytestmark = pytest.mark.gpu

SMALL_T5 = dict(vocab_size=200, d_model=128, d_kv=64, d_ff=256, num_layers=3, num_heads=2, relative_attention_num_buckets=32,
                relative_attention_max_distance=128, layer_norm_epsilon=1e-6, feed_forward_proj="gated-gelu")
SMALL_CLIP = dict(vocab_size=300, projection_dim=128, intermediate_size=256, max_position_embeddings=77, num_hidden_layers=3, num_attention_heads=2)


def _pair(kind, cfg, seed=0):
    import diffusion_rs_amd as d
    from oracle import oracle as orc
    shapes = d.synth.t5_tensor_shapes(cfg) if kind == "t5" else d.synth.clip_tensor_shapes(cfg)
    sd = d.synth.text_state_dict_numpy(shapes, seed=seed)
    gm = (d.T5EncoderModel if kind == "t5" else d.ClipTextTransformer)(cfg)
    gm.load_state_dict(sd)
    om = (orc.T5 if kind == "t5" else orc.Clip)(cfg)
    om.load(sd)
    return d, gm, om, sd


@pytest.mark.parametrize("B,T", [(1, 17), (2, 64), (1, 300)])
def test_t5_matches_oracle(B, T):
    import torch
    d, gm, om, _ = _pair("t5", SMALL_T5)
    ids = np.random.default_rng(T).integers(0, SMALL_T5["vocab_size"], (B, T)).astype(np.int32)
    ids[:, T // 2:] = 0 if T == 64 else ids[:, T // 2:]  # zero padding as tokenize_and_pad / schnell's pad-to-256 produce
    ref = om.forward(ids)
    got32 = host(gm.forward(ids, dtype=torch.float32))
    got16 = host(gm.forward(torch.from_numpy(ids).cuda(), dtype=torch.bfloat16).float())
    err = rel_l2(got32, ref)
    print(f"T5 B={B} T={T}: rel-L2 {err:.3e} (bf16 out {rel_l2(got16, ref):.3e})")
    assert np.isfinite(got32).all() and err <= 1e-2
    assert rel_l2(got16, ref) <= 1.2e-2


@pytest.mark.parametrize("act", ["relu", "gated-silu"])
def test_t5_other_feed_forward_variants(act):
    import torch
    cfg = dict(SMALL_T5, feed_forward_proj=act, num_layers=2)
    d, gm, om, _ = _pair("t5", cfg, seed=3)
    ids = np.random.default_rng(5).integers(0, cfg["vocab_size"], (2, 40)).astype(np.int32)
    err = rel_l2(host(gm.forward(ids, dtype=torch.float32)), om.forward(ids))
    print(f"T5 {act}: rel-L2 {err:.3e}")
    assert err <= 1e-2


def test_t5_relative_bias_table_is_the_reference_bucket_function():
    """The device position bias is gathered through a host-computed bucket table; check that table
    against the oracle's restatement of t5/mod.rs:340-376 by running a 1-layer model whose output
    depends on the bias only through attention (long sequence: log buckets and the clamp)."""
    import torch
    cfg = dict(SMALL_T5, num_layers=1)
    d, gm, om, _ = _pair("t5", cfg, seed=9)
    ids = np.random.default_rng(1).integers(0, cfg["vocab_size"], (1, 400)).astype(np.int32)
    err = rel_l2(host(gm.forward(ids, dtype=torch.float32)), om.forward(ids))
    assert err <= 1e-2, err


def test_clip_matches_oracle():
    d, gm, om, _ = _pair("clip", SMALL_CLIP)
    rng = np.random.default_rng(0)
    ids = rng.integers(1, SMALL_CLIP["vocab_size"] - 1, (3, 33)).astype(np.int32)
    for b, pos in enumerate((32, 5, 20)):  # EOS = largest id, then zero padding
        ids[b, pos] = SMALL_CLIP["vocab_size"] - 1
        ids[b, pos + 1:] = 0
    rp, rh = om.forward(ids, return_hidden=True)
    gp, gh = gm.forward(ids, return_hidden=True)
    gp, gh = host(gp), host(gh)
    print(f"CLIP pooled rel-L2 {rel_l2(gp, rp):.3e}, hidden {rel_l2(gh, rh):.3e}")
    assert rel_l2(gh, rh) <= 1e-2 and rel_l2(gp, rp) <= 1e-2
    # pooled row really is the argmax(id) row of the hidden states
    for b, pos in enumerate((32, 5, 20)):
        np.testing.assert_array_equal(gp[b], gh[b, pos])


def test_encoders_reject_bad_input():
    import diffusion_rs_amd as d
    _, gm, _, _ = _pair("t5", dict(SMALL_T5, num_layers=1))
    with pytest.raises(d.FmiError):
        gm.forward(np.full((1, 8), SMALL_T5["vocab_size"], np.int32))  # id out of range
    with pytest.raises(d.FmiError):
        d.T5EncoderModel(dict(SMALL_T5, d_kv=32))
    _, gc, _, _ = _pair("clip", dict(SMALL_CLIP, num_hidden_layers=1))
    with pytest.raises(d.FmiError):
        gc.forward(np.zeros((1, 78), np.int32))  # longer than max_position_embeddings
    half = d.ClipTextTransformer(dict(SMALL_CLIP, num_hidden_layers=1))
    with pytest.raises(d.FmiError):
        half.forward(np.zeros((1, 4), np.int32))  # weights not loaded


def test_full_size_encoders_run_and_feed_flux_shapes():
    """T5-XXL (4.76 B parameters) and CLIP-L at their real configs: finite, deterministic, and the
    outputs have the shapes/dtypes Flux::forward takes as txt / y."""
    import torch
    import diffusion_rs_amd as d
    t5 = d.T5EncoderModel()
    d.synth.fill_text_random_device(t5, seed=0)
    ids = torch.randint(0, 32128, (1, 512), generator=torch.Generator().manual_seed(0)).int()
    a = t5.forward(ids)
    b = t5.forward(ids)
    torch.cuda.synchronize()
    assert a.shape == (1, 512, 4096) and a.dtype == torch.bfloat16 and torch.isfinite(a.float()).all() and torch.equal(a, b)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    t5.forward(ids)
    e1.record()
    torch.cuda.synchronize()
    print(f"T5-XXL encoder, T=512: {e0.elapsed_time(e1):.1f} ms, weights {t5.size_in_bytes() / 2**30:.1f} GiB")
    t5.close()
    clip = d.ClipTextTransformer()
    d.synth.fill_text_random_device(clip, seed=1)
    cid = torch.randint(1, 49407, (2, 77), generator=torch.Generator().manual_seed(1)).int()
    cid[:, 40] = 49407
    p = clip.forward(cid)
    torch.cuda.synchronize()
    assert p.shape == (2, 768) and p.dtype == torch.float32 and torch.isfinite(p).all()


def test_t5_quantised_linears_match_oracle_on_dequantised_weights():
    """A bitsandbytes-quantised T5 (the reference builds every T5 Linear as a QuantMethod, t5/mod.rs:132-173,258-261): nf4, fp4
    and LLM.int8 Linears fed through fmi_t5_set_linear_bnb4 / _int8 give the encoder output of the oracle run on the dequantised
    weights (BnbLinear::forward = dequantise + matmul, bitsandbytes/mod.rs:293-312), and exactly the output of the same encoder
    loaded with those dequantised weights as bf16 (the expansion is the library's bit-exact dequant)."""
    import torch
    import diffusion_rs_amd as d
    from diffusion_rs_amd import text
    from oracle import oracle as orc
    cfg = dict(vocab_size=96, d_model=128, d_kv=64, d_ff=256, num_layers=2, num_heads=2, relative_attention_num_buckets=32,
               relative_attention_max_distance=128, layer_norm_epsilon=1e-6, feed_forward_proj="gated-gelu")
    sd = d.synth.text_state_dict_numpy(d.synth.t5_tensor_shapes(cfg), seed=21)
    ids = np.random.default_rng(3).integers(0, 96, (2, 40)).astype(np.int32)
    for kind in ("nf4", "fp4", "int8"):
        gq, gd = text.T5EncoderModel(dict(cfg, quantization_config={"quant_method": "bitsandbytes"})), text.T5EncoderModel(cfg)
        dense = {}
        for name, w in sd.items():
            lin = w.ndim == 2 and ("SelfAttention." in name or "DenseReluDense." in name) and "relative_attention_bias" not in name
            if not lin:
                gq.set_tensor(name, w)
                dense[name] = w
                continue
            prefix = name[:-len(".weight")]
            if kind == "int8":
                scb = np.abs(w).max(1).astype(np.float32)
                q = np.clip(np.rint(w / scb[:, None] * 127.0), -127, 127).astype(np.int8)
                gq.set_linear_int8(prefix, torch.from_numpy(q), torch.from_numpy(scb), w.shape[0], w.shape[1])
                dense[name] = orc.dequantize_8bit(q, scb, w.shape[0], w.shape[1], "bf16").reshape(w.shape)
            else:
                packed, absmax = orc.quantize_blockwise_4bit(w.ravel(), 64, kind)
                gq.set_linear_bnb4(prefix, torch.from_numpy(packed), torch.from_numpy(absmax), 64, kind, w.shape[0], w.shape[1])
                dense[name] = orc.dequantize_blockwise(None, packed, absmax, 64, w.size, kind, "bf16").reshape(w.shape)
        assert gq.missing() == []
        gd.load_state_dict(dense)
        om = orc.T5(cfg)
        om.load(dense)
        a, b = host(gq.forward(ids, dtype=torch.float32)), host(gd.forward(ids, dtype=torch.float32))
        err = rel_l2(a, om.forward(ids))
        print(f"T5 with {kind} Linears: rel-L2 vs oracle on the dequantised weights {err:.3e}; identical to the bf16 load of them: {np.array_equal(a, b)}")
        assert np.array_equal(a, b) and err <= 1e-2
        with pytest.raises(d.FmiError):
            gq.set_linear_bnb4("encoder.final_layer_norm", torch.zeros(64, dtype=torch.uint8), torch.zeros(2), 64, "nf4", 128, 1)
        gq.close()
        gd.close()
