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
    ts = sched.get_timesteps(steps, sched.calculate_shift(img.shape[1]))
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
    # seeded latents are reproducible and differ per sample
    a = pipe.forward(["x", "y"], params, seed=5, output="tensor").cpu().numpy()
    b = pipe.forward(["x", "y"], params, seed=5, output="tensor").cpu().numpy()
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
