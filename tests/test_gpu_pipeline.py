"""End-to-end parity of the whole §8 path: FluxPipeline::forward from the embeddings onward
(pack -> 4-step Euler denoise -> unpack+affine -> VAE decode -> u8), GPU Pipeline vs the CPU oracle
pipeline on identical latents / embeddings / weights; plus the diffusers-directory loader."""
import json
import os

import numpy as np
import pytest

from tests.util import SMALL_FLUX, SMALL_VAE, bf16_round, dev, host, rel_l2

pytestmark = pytest.mark.gpu


def _oracle_pipeline(sd, vsd, lat, t5, clip, steps, guidance, sched):
    from oracle import oracle as orc
    om = orc.Flux(SMALL_FLUX)
    om.load(sd)
    ov = orc.Vae(SMALL_VAE)
    ov.load(vsd)
    B, C, h, w = lat.shape
    img, ids = orc.pack_latents(lat)
    txt_ids = np.zeros((B, t5.shape[1], 3), np.float32)
    # the schedule is computed by the ORACLE (orc.calculate_shift / orc.get_timesteps restate flux/sampling.rs:70-80 and
    # scheduler.rs:22-51), not taken from the product's scheduler object — `sched` only supplies the config values
    mu = orc.calculate_shift(img.shape[1], sched.base_image_seq_len, sched.max_image_seq_len, sched.base_shift, sched.max_shift)
    ts = orc.get_timesteps(steps, sched.use_dynamic_shifting, mu, sched.shift)
    g = np.full(B, guidance, np.float32)
    img = om.denoise(img, ids, t5, txt_ids, clip, g, ts)
    z = orc.unpack_latents(img, C, h, w) * np.float32(1.0 / SMALL_VAE["scaling_factor"]) + np.float32(SMALL_VAE["shift_factor"])
    image = ov.decode(z.astype(np.float32))
    return orc.postprocess_u8(image), image


def _write_diffusers_dir(root, sd, vsd):
    import torch
    from safetensors.torch import save_file
    os.makedirs(os.path.join(root, "transformer"))
    os.makedirs(os.path.join(root, "vae"))
    os.makedirs(os.path.join(root, "scheduler"))
    json.dump({"_class_name": "FluxPipeline"}, open(os.path.join(root, "model_index.json"), "w"))
    json.dump({"_class_name": "FlowMatchEulerDiscreteScheduler", "base_image_seq_len": 256, "base_shift": 0.5, "max_image_seq_len": 4096,
               "max_shift": 1.15, "shift": 3.0, "use_dynamic_shifting": True}, open(os.path.join(root, "scheduler", "scheduler_config.json"), "w"))
    json.dump({k: SMALL_FLUX[k] for k in ("in_channels", "pooled_projection_dim", "joint_attention_dim", "num_attention_heads", "num_layers",
                                          "num_single_layers", "guidance_embeds")}, open(os.path.join(root, "transformer", "config.json"), "w"))
    json.dump({k: SMALL_VAE[k] for k in SMALL_VAE}, open(os.path.join(root, "vae", "config.json"), "w"))
    names = list(sd)
    half = len(names) // 2  # two shards, like the real checkpoint
    save_file({k: torch.from_numpy(sd[k]).to(torch.bfloat16) for k in names[:half]}, os.path.join(root, "transformer", "diffusion_pytorch_model-00001-of-00002.safetensors"))
    save_file({k: torch.from_numpy(sd[k]).to(torch.bfloat16) for k in names[half:]}, os.path.join(root, "transformer", "diffusion_pytorch_model-00002-of-00002.safetensors"))
    save_file({k: torch.from_numpy(v) for k, v in vsd.items()}, os.path.join(root, "vae", "diffusion_pytorch_model.safetensors"))


def test_pipeline_end_to_end_matches_oracle(tmp_path):
    import torch
    import diffusion_rs_amd as d
    sd = d.synth.flux_state_dict_numpy(SMALL_FLUX, seed=0)
    vsd = d.synth.vae_state_dict_numpy(SMALL_VAE, seed=0)
    root = str(tmp_path / "tiny-flux")
    _write_diffusers_dir(root, sd, vsd)
    pipe = d.Pipeline(d.ModelSource.ModelId(root))  # Pipeline::load from a diffusers directory
    params = d.DiffusionGenerationParams(height=128, width=192, num_steps=4, guidance_scale=3.5)
    B, T = 2, 24
    rng = np.random.default_rng(3)
    t5 = bf16_round(rng.standard_normal((B, T, SMALL_FLUX["joint_attention_dim"])).astype(np.float32))
    clip = rng.standard_normal((B, SMALL_FLUX["pooled_projection_dim"])).astype(np.float32)
    lat = rng.standard_normal((B, 16, 16, 24)).astype(np.float32)  # (H/16*2, W/16*2)
    u8 = pipe.forward(["a", "b"], params, embeddings=(dev(t5, torch.bfloat16), dev(clip)), latents=dev(lat), output="tensor")
    torch.cuda.synchronize()
    assert tuple(u8.shape) == (B, 3, 128, 192) and u8.dtype == torch.uint8
    ref_u8, ref_img = _oracle_pipeline(sd, vsd, lat, t5, clip, 4, 3.5, pipe.scheduler)
    diff = np.abs(u8.cpu().numpy().astype(np.int32) - ref_u8.astype(np.int32))
    frac = float((diff <= 2).mean())
    print(f"end-to-end u8: max |d| {diff.max()}, frac<=2 {frac:.4f}, mean |d| {diff.mean():.3f}")
    assert frac >= 0.99
    # PNG front end == the pyo3 binding's return type
    pngs = pipe.forward(["a", "b"], params, embeddings=(dev(t5, torch.bfloat16), dev(clip)), latents=dev(lat))
    assert len(pngs) == 2 and all(p[:8] == b"\x89PNG\r\n\x1a\n" for p in pngs)
    # a loaded checkpoint never invents embeddings (the reference always tokenises and encodes): without text encoders it needs them passed
    with pytest.raises(d.FmiError):
        pipe.forward(["x", "y"], params, seed=5, output="tensor")
    # seeded latents are reproducible and differ per sample
    emb = (dev(t5, torch.bfloat16), dev(clip))
    a = pipe.forward(["x", "y"], params, seed=5, output="tensor", embeddings=emb).cpu().numpy()
    b = pipe.forward(["x", "y"], params, seed=5, output="tensor", embeddings=emb).cpu().numpy()
    np.testing.assert_array_equal(a, b)
    assert not np.array_equal(a[0], a[1])


def test_pipeline_rejects_non_flux_and_f32(tmp_path):
    import diffusion_rs_amd as d
    root = tmp_path / "bad"
    root.mkdir()
    json.dump({"_class_name": "StableDiffusionPipeline"}, open(root / "model_index.json", "w"))
    with pytest.raises(ValueError):
        d.Pipeline(d.ModelSource.ModelId(str(root)))
    with pytest.raises(ValueError):
        d.Pipeline(d.ModelSource.Synthetic(), dtype=d.ModelDType.F32)


def test_dduf_with_bnb_nf4_nested_absmax(tmp_path):
    """§8(f) rank 1: a DDUF (stored zip) FLUX checkpoint whose block linears are HF-bitsandbytes nf4 with
    double-quantised (nested) absmax loads through the FileLoader + bnb name detection and matches the
    oracle run on the dequantised weights (BnbLinear::dequantize_4bit, bitsandbytes/mod.rs:225-261)."""
    import io
    import zipfile
    import torch
    from safetensors.torch import save
    import diffusion_rs_amd as d
    from oracle import oracle as orc
    from tests.util import flux_inputs
    sd = d.synth.flux_state_dict_numpy(SMALL_FLUX, seed=2)
    vsd = d.synth.vae_state_dict_numpy(SMALL_VAE, seed=2)
    code256 = np.linspace(-1.0, 1.0, 256).astype(np.float32)
    nf4_map = np.array(d.synth.NF4_CODE, np.float32)
    tens, dense_for_oracle = {}, {}
    nq = 0
    for name, w in sd.items():
        if d.synth.is_block_linear(name):
            prefix = name[:-len(".weight")]
            packed, absmax = orc.quantize_blockwise_4bit(w.ravel(), 64, "nf4")
            off = float(absmax.mean())
            a = absmax - np.float32(off)
            nb = (a.size + 255) // 256
            nabs = np.array([np.abs(a[i * 256:(i + 1) * 256]).max() for i in range(nb)], np.float32)
            idx = np.array([np.abs(code256 - (a[i] / max(nabs[i // 256], 1e-30))).argmin() for i in range(a.size)], np.uint8)
            eff = (code256[idx] * nabs[np.arange(a.size) // 256]).astype(np.float32) + np.float32(off)
            state = {"blocksize": 64, "shape": list(w.shape), "dtype": "bfloat16", "nested_blocksize": 256, "nested_offset": off,
                     "nested_dtype": "float32", "quant_type": "nf4"}
            tens[name] = torch.from_numpy(packed.reshape(-1, 1))
            tens[prefix + ".weight.absmax"] = torch.from_numpy(idx)
            tens[prefix + ".weight.quant_map"] = torch.from_numpy(nf4_map.copy())
            tens[prefix + ".weight.nested_absmax"] = torch.from_numpy(nabs)
            tens[prefix + ".weight.nested_quant_map"] = torch.from_numpy(code256.copy())
            tens[prefix + ".weight.quant_state.bitsandbytes__nf4"] = torch.from_numpy(np.frombuffer(json.dumps(state).encode(), np.uint8).copy())
            dense_for_oracle[name] = orc.dequantize_blockwise(None, packed, eff, 64, w.size, "nf4", "bf16").reshape(w.shape)
            nq += 1
        else:
            tens[name] = torch.from_numpy(w).to(torch.bfloat16)
            dense_for_oracle[name] = w
    path = str(tmp_path / "tiny-flux-Q4-bnb.dduf")
    with zipfile.ZipFile(path, "w", compression=zipfile.ZIP_STORED) as z:
        z.writestr("model_index.json", json.dumps({"_class_name": "FluxPipeline"}))
        z.writestr("scheduler/scheduler_config.json", json.dumps({"_class_name": "FlowMatchEulerDiscreteScheduler", "base_image_seq_len": 256,
                   "base_shift": 0.5, "max_image_seq_len": 4096, "max_shift": 1.15, "shift": 3.0, "use_dynamic_shifting": True}))
        z.writestr("transformer/config.json", json.dumps({k: SMALL_FLUX[k] for k in ("in_channels", "pooled_projection_dim", "joint_attention_dim",
                   "num_attention_heads", "num_layers", "num_single_layers", "guidance_embeds")}))
        z.writestr("transformer/diffusion_pytorch_model.safetensors", save(tens))
        z.writestr("vae/config.json", json.dumps(dict(SMALL_VAE)))
        z.writestr("vae/diffusion_pytorch_model.safetensors", save({k: torch.from_numpy(v) for k, v in vsd.items()}))
    pipe = d.Pipeline(d.ModelSource.DdufFile(path))
    assert pipe.load_stats["bnb4"] == nq and nq > 0
    om = orc.Flux(SMALL_FLUX)
    om.load(dense_for_oracle)
    img, ids, txt, txt_ids, y = flux_inputs(SMALL_FLUX, 1, (8, 8), 32, seed=4)
    t = np.array([0.7], np.float32)
    g = np.array([3.5], np.float32)
    ref = om.forward(img, ids, txt, txt_ids, t, y, g)
    got = host(pipe.flux.forward(dev(img), dev(ids), dev(txt, torch.bfloat16), dev(txt_ids), dev(t), dev(y), dev(g)))
    err = rel_l2(got, ref)
    print(f"DDUF + bnb-nf4 (nested absmax): {nq} quantised linears, rel-L2 {err:.3e}")
    assert err <= 1e-2
    # a compressed entry is rejected (DDUF entries must be stored: tensors are sliced from the mmap)
    bad = str(tmp_path / "bad.dduf")
    with zipfile.ZipFile(bad, "w", compression=zipfile.ZIP_DEFLATED) as z:
        z.writestr("model_index.json", json.dumps({"_class_name": "FluxPipeline"}) * 50)
    with pytest.raises(ValueError):
        d.Pipeline(d.ModelSource.DdufFile(bad))


def test_prompt_to_image_with_text_encoders(tmp_path):
    """§8(f) rank 2: string prompts -> tokenizers -> T5 + CLIP on the GPU -> denoise -> VAE -> u8, from a
    diffusers directory that ships text_encoder / text_encoder_2 / tokenizer / tokenizer_2, vs the oracle
    chain (oracle T5/CLIP -> oracle FLUX -> oracle VAE) on the same token ids and latents."""
    import torch
    from safetensors.torch import save_file
    from tokenizers import Tokenizer, models, pre_tokenizers, processors
    import diffusion_rs_amd as d
    from oracle import oracle as orc
    t5_cfg = dict(vocab_size=64, d_model=SMALL_FLUX["joint_attention_dim"], d_kv=64, d_ff=256, num_layers=2, num_heads=2, relative_attention_num_buckets=32,
                  relative_attention_max_distance=128, layer_norm_epsilon=1e-6, feed_forward_proj="gated-gelu")
    clip_cfg = dict(vocab_size=40, projection_dim=SMALL_FLUX["pooled_projection_dim"], intermediate_size=128, max_position_embeddings=77, num_hidden_layers=2,
                    num_attention_heads=1)
    sd = d.synth.flux_state_dict_numpy(SMALL_FLUX, seed=4)
    vsd = d.synth.vae_state_dict_numpy(SMALL_VAE, seed=4)
    tsd = d.synth.text_state_dict_numpy(d.synth.t5_tensor_shapes(t5_cfg), seed=5)
    csd = d.synth.text_state_dict_numpy(d.synth.clip_tensor_shapes(clip_cfg), seed=6)
    root = str(tmp_path / "tiny-flux-full")
    _write_diffusers_dir(root, sd, vsd)
    for sub in ("text_encoder", "text_encoder_2", "tokenizer", "tokenizer_2"):
        os.makedirs(os.path.join(root, sub))
    json.dump(dict(clip_cfg, hidden_size=clip_cfg["projection_dim"], hidden_act="quick_gelu"), open(os.path.join(root, "text_encoder", "config.json"), "w"))
    json.dump(dict(t5_cfg, is_encoder_decoder=True), open(os.path.join(root, "text_encoder_2", "config.json"), "w"))
    save_file({k: torch.from_numpy(v).to(torch.bfloat16) for k, v in csd.items()}, os.path.join(root, "text_encoder", "model.safetensors"))
    t5_file = {k: torch.from_numpy(v).to(torch.bfloat16) for k, v in tsd.items()}
    t5_file["encoder.embed_tokens.weight"] = t5_file["shared.weight"].clone()  # tied copy real checkpoints carry
    save_file(t5_file, os.path.join(root, "text_encoder_2", "model.safetensors"))
    # CLIP tokenizer files: vocab.json + merges.txt (first line is a header the reference skips, tokenizer.rs:14-16)
    letters = "abcdefghijklmnopqrstuvwxyz "
    vocab = {ch: i for i, ch in enumerate(letters)}
    merges = [("t", "h"), ("th", "e"), ("a", "t"), ("c", "at")]
    for a, b in merges:
        vocab[a + b] = len(vocab)
    json.dump(vocab, open(os.path.join(root, "tokenizer", "vocab.json"), "w"))
    open(os.path.join(root, "tokenizer", "merges.txt"), "w").write("#version: 0.2\n" + "\n".join(f"{a} {b}" for a, b in merges) + "\n")
    # T5 tokenizer.json: word-level stand-in with an </s> post-processor (ids: pad 0, </s> 1, unk 2)
    words = ["<pad>", "</s>", "<unk>", "the", "cat", "sat", "on", "a", "mat", "dog", "ran"]
    tk = Tokenizer(models.WordLevel({w: i for i, w in enumerate(words)}, unk_token="<unk>"))
    tk.pre_tokenizer = pre_tokenizers.Whitespace()
    tk.post_processor = processors.TemplateProcessing(single="$A </s>", special_tokens=[("</s>", 1)])
    tk.save(os.path.join(root, "tokenizer_2", "tokenizer.json"))

    pipe = d.Pipeline(d.ModelSource.ModelId(root))
    assert pipe.t5 is not None and pipe.clip is not None and pipe.t5_tokenizer is not None and pipe.clip_tokenizer is not None
    prompts = ["the cat sat on a mat", "a dog ran"]
    t5_ids = np.array(d.tokenize_and_pad(prompts, pipe.t5_tokenizer), np.int32)
    clip_ids = np.array(d.tokenize_and_pad(prompts, pipe.clip_tokenizer), np.int32)
    assert t5_ids.shape == (2, 7) and t5_ids[1].tolist() == [7, 9, 10, 1, 0, 0, 0]  # </s> appended, zero padded to the longest
    assert clip_ids.shape[0] == 2 and (clip_ids[1][-3:] == 0).all()
    params = d.DiffusionGenerationParams(height=128, width=128, num_steps=3, guidance_scale=3.5)
    lat = np.random.default_rng(8).standard_normal((2, 16, 16, 16)).astype(np.float32)
    u8 = pipe.forward(prompts, params, latents=dev(lat), output="tensor")
    torch.cuda.synchronize()
    # oracle chain on the same ids
    ot, oc = orc.T5(t5_cfg), orc.Clip(clip_cfg)
    ot.load(tsd)
    oc.load(csd)
    t5_emb, clip_emb = ot.forward(t5_ids), oc.forward(clip_ids)
    g_t5, g_clip = pipe.encode_prompts(prompts)
    print(f"prompt embeddings: T5 rel-L2 {rel_l2(host(g_t5.float()), t5_emb):.3e}, CLIP rel-L2 {rel_l2(host(g_clip), clip_emb):.3e}")
    assert rel_l2(host(g_t5.float()), t5_emb) <= 1.2e-2 and rel_l2(host(g_clip), clip_emb) <= 1e-2
    ref_u8, _ = _oracle_pipeline_cfg(sd, vsd, lat, t5_emb, clip_emb, 3, 3.5, pipe.scheduler)
    diff = np.abs(u8.cpu().numpy().astype(np.int32) - ref_u8.astype(np.int32))
    print(f"prompt -> image u8: max |d| {diff.max()}, frac<=2 {float((diff <= 2).mean()):.4f}")
    assert float((diff <= 2).mean()) >= 0.99
    # token_ids= bypasses the tokenizers and gives the same image
    u8b = pipe.forward(prompts, params, latents=dev(lat), output="tensor", token_ids=(t5_ids, clip_ids))
    assert torch.equal(u8, u8b)


def _oracle_pipeline_cfg(sd, vsd, lat, t5, clip, steps, guidance, sched):
    return _oracle_pipeline(sd, vsd, lat, t5.astype(np.float32), clip.astype(np.float32), steps, guidance, sched)


def test_c1_schnell_256x256_4step_matches_oracle(tmp_path):
    """BASELINE.json configs[0] (C1) at the shapes the oracle can hold: FLUX.1-schnell semantics — no guidance
    embedder (guidance_embeds=false), 256x256 image (S = 256), T = 256 text tokens, 4 Euler steps on the
    non-dynamic shift=1.0 schedule — through Pipeline (diffusers directory) vs the oracle chain."""
    import torch
    import diffusion_rs_amd as d
    from oracle import oracle as orc
    cfg = dict(SMALL_FLUX, guidance_embeds=False)
    sd = d.synth.flux_state_dict_numpy(cfg, seed=21)
    vsd = d.synth.vae_state_dict_numpy(SMALL_VAE, seed=21)
    assert not any("guidance_embedder" in k for k in sd)
    root = str(tmp_path / "tiny-schnell")
    _write_diffusers_dir(root, sd, vsd)
    json.dump({k: cfg[k] for k in ("in_channels", "pooled_projection_dim", "joint_attention_dim", "num_attention_heads", "num_layers", "num_single_layers",
                                   "guidance_embeds")}, open(os.path.join(root, "transformer", "config.json"), "w"))
    json.dump({"_class_name": "FlowMatchEulerDiscreteScheduler", "base_image_seq_len": 256, "base_shift": 0.5, "max_image_seq_len": 4096, "max_shift": 1.15,
               "shift": 1.0, "use_dynamic_shifting": False}, open(os.path.join(root, "scheduler", "scheduler_config.json"), "w"))
    pipe = d.Pipeline(d.ModelSource.ModelId(root))
    assert not pipe.flux.is_guidance()
    B, T = 1, 256
    rng = np.random.default_rng(4)
    t5 = bf16_round(rng.standard_normal((B, T, cfg["joint_attention_dim"])).astype(np.float32))
    clip = rng.standard_normal((B, cfg["pooled_projection_dim"])).astype(np.float32)
    lat = rng.standard_normal((B, 16, 32, 32)).astype(np.float32)  # 256x256 -> 32x32 latent -> S = 256
    params = d.DiffusionGenerationParams(height=256, width=256, num_steps=4, guidance_scale=0.0)
    u8 = pipe.forward(["x"], params, embeddings=(dev(t5, torch.bfloat16), dev(clip)), latents=dev(lat), output="tensor")
    ts = pipe.scheduler.get_timesteps(4, pipe.scheduler.calculate_shift(256))
    np.testing.assert_allclose(ts, [1.0, 0.75, 0.5, 0.25, 0.0], atol=1e-12)  # linspace, shift 1.0 (scheduler.rs / schnell)
    om, ov = orc.Flux(cfg), orc.Vae(SMALL_VAE)
    om.load(sd)
    ov.load(vsd)
    img, ids = orc.pack_latents(lat)
    assert img.shape[1] == 256
    img = om.denoise(img, ids, t5, np.zeros((B, T, 3), np.float32), clip, None, ts)
    z = orc.unpack_latents(img, 16, 32, 32) * np.float32(1.0 / SMALL_VAE["scaling_factor"]) + np.float32(SMALL_VAE["shift_factor"])
    ref_u8 = orc.postprocess_u8(ov.decode(z.astype(np.float32)))
    diff = np.abs(u8.cpu().numpy().astype(np.int32) - ref_u8.astype(np.int32))
    print(f"C1 (schnell 256x256 4-step) u8: max |d| {diff.max()}, frac<=2 {float((diff <= 2).mean()):.4f}")
    assert float((diff <= 2).mean()) >= 0.99


def test_pipeline_more_prompts_than_max_batch(tmp_path):
    """The reference accepts any batch (pipelines/mod.rs:241-270); the workspace here holds MAX_BATCH = 8 samples, so
    forward() runs sub-batches — with the Philox stream of a sample fixed by its index, 11 prompts in one call must give
    exactly the images of the same prompts run one sub-batch at a time by hand."""
    import torch
    import diffusion_rs_amd as d
    sd = d.synth.flux_state_dict_numpy(SMALL_FLUX, seed=1)
    vsd = d.synth.vae_state_dict_numpy(SMALL_VAE, seed=1)
    root = str(tmp_path / "tiny-flux")
    _write_diffusers_dir(root, sd, vsd)
    pipe = d.Pipeline(d.ModelSource.ModelId(root))
    assert pipe.MAX_BATCH == 8
    params = d.DiffusionGenerationParams(height=64, width=64, num_steps=2, guidance_scale=3.5)
    B, T = 11, 8
    rng = np.random.default_rng(5)
    t5 = dev(bf16_round(rng.standard_normal((B, T, SMALL_FLUX["joint_attention_dim"])).astype(np.float32)), torch.bfloat16)
    clip = dev(rng.standard_normal((B, SMALL_FLUX["pooled_projection_dim"])).astype(np.float32))
    prompts = [f"p{i}" for i in range(B)]
    whole = pipe.forward(prompts, params, embeddings=(t5, clip), seed=7, output="tensor").cpu().numpy()
    assert whole.shape == (B, 3, 64, 64)
    first = pipe.generate_tensor(prompts[:8], params, embeddings=(t5[:8], clip[:8]), seed=7, sample_ids=list(range(8))).cpu().numpy()
    rest = pipe.generate_tensor(prompts[8:], params, embeddings=(t5[8:], clip[8:]), seed=7, sample_ids=[8, 9, 10]).cpu().numpy()
    np.testing.assert_array_equal(whole, np.concatenate([first, rest], 0))
    assert len({whole[i].tobytes() for i in range(B)}) == B  # every sample drew its own noise


def _nf4_group(orc, d, name, w, nested):
    """the tensors HF-bitsandbytes stores for one nf4 Linear (bitsandbytes/mod.rs:137-239), and the dense weight they stand for"""
    import torch
    prefix = name[:-len(".weight")]
    packed, absmax = orc.quantize_blockwise_4bit(w.ravel(), 64, "nf4")
    state = {"blocksize": 64, "shape": list(w.shape), "dtype": "bfloat16", "quant_type": "nf4"}
    out = {name: torch.from_numpy(packed.reshape(-1, 1)), prefix + ".weight.quant_map": torch.from_numpy(np.array(d.synth.NF4_CODE, np.float32))}
    eff = absmax
    if nested:
        code256 = np.linspace(-1.0, 1.0, 256).astype(np.float32)
        off = float(absmax.mean())
        a = absmax - np.float32(off)
        nb = (a.size + 255) // 256
        nabs = np.array([np.abs(a[i * 256:(i + 1) * 256]).max() for i in range(nb)], np.float32)
        idx = np.array([np.abs(code256 - (a[i] / max(nabs[i // 256], 1e-30))).argmin() for i in range(a.size)], np.uint8)
        eff = (code256[idx] * nabs[np.arange(a.size) // 256]).astype(np.float32) + np.float32(off)
        state.update({"nested_blocksize": 256, "nested_offset": off, "nested_dtype": "float32"})
        out[prefix + ".weight.absmax"] = torch.from_numpy(idx)
        out[prefix + ".weight.nested_absmax"] = torch.from_numpy(nabs)
        out[prefix + ".weight.nested_quant_map"] = torch.from_numpy(code256.copy())
    else:
        out[prefix + ".weight.absmax"] = torch.from_numpy(absmax)
    out[prefix + ".weight.quant_state.bitsandbytes__nf4"] = torch.from_numpy(np.frombuffer(json.dumps(state).encode(), np.uint8).copy())
    dense = orc.dequantize_blockwise(None, packed, eff, 64, w.size, "nf4", "bf16").reshape(w.shape)
    return out, dense


def test_q4_bnb_dduf_with_quantised_t5_prompt_to_image(tmp_path):
    """§8(f) rank 2 completed: the layout of the reference's own example checkpoint, FLUX.1-dev-Q4-bnb.dduf
    (README.md:37-41) — ONE DDUF whose transformer AND text_encoder_2 (T5) are bitsandbytes nf4 (text_encoder_2/config.json carries
    a quantization_config; the reference builds every T5 Linear through it, t5/mod.rs:132-173,258-261), bf16 CLIP, both
    tokenizers — goes prompt -> u8 through the front door, against the oracle chain on the dequantised weights."""
    import zipfile
    import torch
    from safetensors.torch import save
    from tokenizers import Tokenizer, models, pre_tokenizers, processors
    import diffusion_rs_amd as d
    from oracle import oracle as orc
    t5_cfg = dict(vocab_size=64, d_model=SMALL_FLUX["joint_attention_dim"], d_kv=64, d_ff=256, num_layers=2, num_heads=2, relative_attention_num_buckets=32,
                  relative_attention_max_distance=128, layer_norm_epsilon=1e-6, feed_forward_proj="gated-gelu")
    clip_cfg = dict(vocab_size=40, projection_dim=SMALL_FLUX["pooled_projection_dim"], intermediate_size=128, max_position_embeddings=77, num_hidden_layers=2,
                    num_attention_heads=1)
    sd = d.synth.flux_state_dict_numpy(SMALL_FLUX, seed=12)
    vsd = d.synth.vae_state_dict_numpy(SMALL_VAE, seed=12)
    tsd = d.synth.text_state_dict_numpy(d.synth.t5_tensor_shapes(t5_cfg), seed=13)
    csd = d.synth.text_state_dict_numpy(d.synth.clip_tensor_shapes(clip_cfg), seed=14)
    ft, fdense, nq = {}, {}, 0
    for name, w in sd.items():
        if d.synth.is_block_linear(name):
            grp, dense = _nf4_group(orc, d, name, w, nested=True)
            ft.update(grp)
            fdense[name] = dense
            nq += 1
        else:
            ft[name] = torch.from_numpy(w).to(torch.bfloat16)
            fdense[name] = w
    tt, tdense, nqt = {}, {}, 0
    for name, w in tsd.items():
        if w.ndim == 2 and ("SelfAttention." in name or "DenseReluDense." in name) and "relative_attention_bias" not in name:
            grp, dense = _nf4_group(orc, d, name, w, nested=(nqt % 2 == 0))  # plain and double-quantised absmax alternate
            tt.update(grp)
            tdense[name] = dense
            nqt += 1
        else:
            tt[name] = torch.from_numpy(w).to(torch.bfloat16)
            tdense[name] = w
    assert nqt == 2 * 7  # q, k, v, o, wi_0, wi_1, wo per block
    letters = "abcdefghijklmnopqrstuvwxyz "
    vocab = {ch: i for i, ch in enumerate(letters)}
    merges = [("t", "h"), ("th", "e"), ("a", "t"), ("c", "at")]
    for a, b in merges:
        vocab[a + b] = len(vocab)
    words = ["<pad>", "</s>", "<unk>", "the", "cat", "sat", "on", "a", "mat", "dog", "ran"]
    tk = Tokenizer(models.WordLevel({w: i for i, w in enumerate(words)}, unk_token="<unk>"))
    tk.pre_tokenizer = pre_tokenizers.Whitespace()
    tk.post_processor = processors.TemplateProcessing(single="$A </s>", special_tokens=[("</s>", 1)])
    qcfg = {"quant_method": "bitsandbytes", "load_in_4bit": True, "bnb_4bit_quant_type": "nf4", "bnb_4bit_use_double_quant": True}
    path = str(tmp_path / "tiny-FLUX.1-dev-Q4-bnb.dduf")
    with zipfile.ZipFile(path, "w", compression=zipfile.ZIP_STORED) as z:
        z.writestr("model_index.json", json.dumps({"_class_name": "FluxPipeline"}))
        z.writestr("scheduler/scheduler_config.json", json.dumps({"_class_name": "FlowMatchEulerDiscreteScheduler", "base_image_seq_len": 256,
                   "base_shift": 0.5, "max_image_seq_len": 4096, "max_shift": 1.15, "shift": 3.0, "use_dynamic_shifting": True}))
        z.writestr("transformer/config.json", json.dumps(dict({k: SMALL_FLUX[k] for k in ("in_channels", "pooled_projection_dim", "joint_attention_dim",
                   "num_attention_heads", "num_layers", "num_single_layers", "guidance_embeds")}, quantization_config=qcfg)))
        z.writestr("transformer/diffusion_pytorch_model.safetensors", save(ft))
        z.writestr("vae/config.json", json.dumps(dict(SMALL_VAE)))
        z.writestr("vae/diffusion_pytorch_model.safetensors", save({k: torch.from_numpy(v) for k, v in vsd.items()}))
        z.writestr("text_encoder/config.json", json.dumps(dict(clip_cfg, hidden_size=clip_cfg["projection_dim"], hidden_act="quick_gelu")))
        z.writestr("text_encoder/model.safetensors", save({k: torch.from_numpy(v).to(torch.bfloat16) for k, v in csd.items()}))
        z.writestr("text_encoder_2/config.json", json.dumps(dict(t5_cfg, is_encoder_decoder=True, quantization_config=qcfg)))
        z.writestr("text_encoder_2/model.safetensors", save(tt))
        z.writestr("tokenizer/vocab.json", json.dumps(vocab))
        z.writestr("tokenizer/merges.txt", "#version: 0.2\n" + "\n".join(f"{a} {b}" for a, b in merges) + "\n")
        z.writestr("tokenizer_2/tokenizer.json", tk.to_str())
    pipe = d.Pipeline(d.ModelSource.DdufFile(path))
    assert pipe.load_stats["bnb4"] == nq and pipe.t5 is not None and pipe.t5.load_stats["bnb4"] == nqt
    prompts = ["the cat sat on a mat", "a dog ran"]
    params = d.DiffusionGenerationParams(height=128, width=128, num_steps=3, guidance_scale=3.5)
    lat = np.random.default_rng(9).standard_normal((2, 16, 16, 16)).astype(np.float32)
    u8 = pipe.forward(prompts, params, latents=dev(lat), output="tensor")
    t5_ids = np.array(d.tokenize_and_pad(prompts, pipe.t5_tokenizer), np.int32)
    clip_ids = np.array(d.tokenize_and_pad(prompts, pipe.clip_tokenizer), np.int32)
    ot, oc = orc.T5(t5_cfg), orc.Clip(clip_cfg)
    ot.load(tdense)
    oc.load(csd)
    t5_emb, clip_emb = ot.forward(t5_ids), oc.forward(clip_ids)
    g_t5, _ = pipe.encode_prompts(prompts)
    e_t5 = rel_l2(host(g_t5.float()), t5_emb)
    ref_u8, _ = _oracle_pipeline_cfg(fdense, vsd, lat, t5_emb, clip_emb, 3, 3.5, pipe.scheduler)
    diff = np.abs(u8.cpu().numpy().astype(np.int32) - ref_u8.astype(np.int32))
    print(f"Q4-bnb DDUF, nf4 transformer ({nq} linears) + nf4 T5 ({nqt} linears): T5 embeddings rel-L2 {e_t5:.3e}; prompt -> u8 max |d| {diff.max()}, "
          f"frac<=2 {float((diff <= 2).mean()):.4f}")
    assert e_t5 <= 1.2e-2 and float((diff <= 2).mean()) >= 0.99
