"""Parity at the PRODUCTION shapes of the two models (VERDICT r2 "missing" 2 and 3).

* `Decoder::forward` (vaes/vae.rs:436-456) at the real FLUX AutoencoderKL config — block_out_channels
  [128, 256, 512, 512], 3 resnets per up level, 32 groups, mid attention — for a 64x64 latent (512^2 image: the Cin = 512
  convolutions with K = 4608, a 4096-token AttnBlock) and for the 128x128 latent of BASELINE configs[1] (1024^2: the 16384-token
  AttnBlock, GroupNorm over 4 M elements per group), against the f32 CPU oracle.  Tolerances as in tests/test_gpu_vae.py:
  rel-L2 <= 2e-2 on the decoded image, |delta u8| <= 2 on >= 99 % of the pixels.
* one Cin = 512 3x3 convolution (ResnetBlock2D conv1/conv2 of the 512-channel levels, vae.rs:157-172) and one AttnBlock
  (vae.rs:95-111) on 4096 tokens, as single ops.
* BASELINE configs[0] (C1: FLUX.1-schnell 256x256, 4 steps, batch 1 — the reference's own CPU-runnable case) IN FULL:
  D = 3072, 19 double + 38 single blocks (`Flux::forward`, model.rs:790-833), S = T = 256, the 4-step Euler loop
  (sampling.rs:25-48), every block with its own weights (12e9, generated on the GPU, handed to the oracle as f32 or — on small
  hosts — as bf16 bits widened per block).  Tolerance: latents after the loop rel-L2 <= 3e-2.
* BASELINE configs[1] (C2, the headline config) at full size for ONE model evaluation: FLUX.1-dev, S = 4096 + T = 512 tokens.
"""
import time

import numpy as np
import pytest

from tests.util import bf16_round, dev, host, rel_l2

pytestmark = pytest.mark.gpu


def _host_memory_gib():
    """memory this process may use: min(MemAvailable, cgroup limit)"""
    avail = 0.0
    try:
        for line in open("/proc/meminfo"):
            if line.startswith("MemAvailable:"):
                avail = int(line.split()[1]) / 2**20
    except OSError:
        pass
    for path in ("/sys/fs/cgroup/memory.max", "/sys/fs/cgroup/memory/memory.limit_in_bytes"):  # cgroup v2, v1
        try:
            lim = open(path).read().strip()
            if lim != "max":
                avail = min(avail, int(lim) / 2**30)
        except (OSError, ValueError):
            pass
    return avail


def _u8_agreement(d, orc, got, ref):
    import torch
    u_ref = orc.postprocess_u8(ref)
    u_got = d.postprocess_u8(torch.from_numpy(got).cuda()).cpu().numpy()
    diff = np.abs(u_ref.astype(np.int32) - u_got.astype(np.int32))
    return int(diff.max()), float((diff <= 2).mean())


@pytest.mark.parametrize("h", [64, 128])
def test_vae_decode_flux_config_matches_oracle(h):
    import diffusion_rs_amd as d
    from oracle import oracle as orc
    sd = d.synth.vae_state_dict_numpy(d.VAE_FLUX, seed=4)
    gv = d.AutoEncoderKl(d.VAE_FLUX)
    gv.load_state_dict(sd)
    ov = orc.Vae(d.VAE_FLUX)
    ov.load(sd)
    z = np.random.default_rng(h).standard_normal((1, 16, h, h)).astype(np.float32)
    t0 = time.time()
    ref = ov.decode(z)
    t_or = time.time() - t0
    got = host(gv.decode(dev(z)))
    assert got.shape == ref.shape == (1, 3, 8 * h, 8 * h)
    assert np.isfinite(got).all()
    err = rel_l2(got, ref)
    mx, frac = _u8_agreement(d, orc, got, ref)
    print(f"VAE decode at the FLUX config, latent {h}x{h} -> {8 * h}^2: rel-L2 {err:.3e}, u8 max |d| {mx}, frac<=2 {frac:.5f} "
          f"(oracle {t_or:.1f} s, ref range [{ref.min():.2f},{ref.max():.2f}])")
    # round 5: the decoder's trunk is f32 (GroupNorm reads it, the convolutions' epilogues add to it): 1.19e-2 / 99.41 % before, SURVEY 8(d)'s starting bar now
    assert err <= 1.2e-2
    assert frac >= 0.999
    gv.close()


def test_conv2d_cin512_matches_oracle():
    """3x3, Cin = Cout = 512 on a 64x64 map: the K = 4608 implicit GEMM of the 512-channel ResnetBlocks (vae.rs:157-172), plain
    and with the 2x nearest upsample folded into the gather (Upsample2D, vae.rs:225-236)."""
    import ctypes as C
    import torch
    from diffusion_rs_amd import _lib as L
    from oracle import oracle as orc
    lib = L.load()
    L.check(lib.fmi_init(0))
    rng = np.random.default_rng(7)
    Cin, Cout = 512, 512
    w = bf16_round((rng.standard_normal((Cout, Cin, 3, 3)) / np.sqrt(Cin * 9)).astype(np.float32))
    b = bf16_round((0.1 * rng.standard_normal(Cout)).astype(np.float32))
    wd = dev(w.transpose(0, 2, 3, 1).copy(), torch.bfloat16)
    bd = dev(b, torch.bfloat16)
    for (H, W, up) in ((64, 64, 0), (32, 32, 1)):
        x = bf16_round(rng.standard_normal((1, Cin, H, W)).astype(np.float32))
        xin = orc.upsample_nearest2d(x, 2 * H, 2 * W) if up else x
        ref = orc.conv2d(xin, w, b, pad=1)
        xd = dev(x.transpose(0, 2, 3, 1).copy(), torch.bfloat16)
        out = torch.full((1, ref.shape[2], ref.shape[3], Cout), float("nan"), dtype=torch.bfloat16, device="cuda")
        p = lambda t: C.c_void_p(t.data_ptr())
        L.check(lib.fmi_conv2d_nhwc(p(xd), p(wd), p(bd), None, p(out), 1, H, W, Cin, Cout, 3, up, None))
        torch.cuda.synchronize()
        got = host(out).transpose(0, 3, 1, 2)
        err = rel_l2(got, ref)
        print(f"conv2d 3x3 512->512, input {H}x{W}, upsample {up}: rel-L2 {err:.3e}")
        assert np.isfinite(got).all() and err <= 4e-3


def test_vae_attn_block_4096_tokens_matches_oracle():
    """AttnBlock::forward (vae.rs:95-111) alone on a 64x64 map of 512 channels: GroupNorm(32), q/k/v 1x1, the 4096x4096 softmax,
    to_out + residual (fmi_vae_mid_attention vs the oracle's v_attn), with the weights of the FLUX-config decoder."""
    import torch
    import diffusion_rs_amd as d
    from oracle import oracle as orc
    sd = d.synth.vae_state_dict_numpy(d.VAE_FLUX, seed=5)
    gv = d.AutoEncoderKl(d.VAE_FLUX)
    gv.load_state_dict(sd)
    ov = orc.Vae(d.VAE_FLUX)
    ov.load(sd)
    x = bf16_round((1.5 * np.random.default_rng(8).standard_normal((1, 512, 64, 64))).astype(np.float32))
    ref = ov.mid_attention(x)
    got = host(gv.mid_attention(dev(x.transpose(0, 2, 3, 1).copy(), torch.bfloat16))).transpose(0, 3, 1, 2)
    err, delta = rel_l2(got, ref), rel_l2(got - x, ref - x)
    print(f"VAE AttnBlock, 4096 tokens x 512 channels: rel-L2 {err:.3e}; of the attention branch alone (out - x) {delta:.3e}")
    assert np.isfinite(got).all() and err <= 4e-3 and delta <= 2e-2
    gv.close()


def _seeded_weight(torch, name, shape, d):
    """the synthetic checkpoint value of one tensor, seeded by its NAME: the dev and the schnell model (which differ by the guidance
    embedder only) then share every other tensor"""
    import zlib
    g = torch.Generator(device="cuda")
    g.manual_seed(zlib.crc32(name.encode()))
    if "norm_q.weight" in name or "norm_k.weight" in name or "norm_added" in name:
        return (1.0 + 0.1 * torch.randn(shape, generator=g, device="cuda")).to(torch.bfloat16)
    if name.endswith(".bias"):
        return (0.02 * torch.randn(shape, generator=g, device="cuda")).to(torch.bfloat16)
    t = torch.randn(shape, generator=g, device="cuda", dtype=torch.bfloat16)
    t.mul_(d.synth._std_for(name, 0.02, 0.01))
    return t


@pytest.fixture(scope="module")
def full_models():
    """FLUX.1 at full size — D = 3072, 19 + 38 blocks, 11.9e9 weights, every block its own — on the GPU (dev and schnell handles)
    and in the oracle (ONE object with the dev tensors: called without guidance it is the schnell model, model.rs:813-820).
    Oracle weights as f32 (48 GB) when the host has the room — the GPU box does — else as bf16 bits widened per block (24 GB;
    exact either way, the paging costs minutes)."""
    import torch
    import diffusion_rs_amd as d
    from oracle import oracle as orc
    wide = _host_memory_gib() >= 140
    t0 = time.time()
    gm_dev, gm_sch = d.FluxModel(dict(d.FLUX_DEV)), d.FluxModel(dict(d.FLUX_SCHNELL))
    om = orc.Flux(dict(d.FLUX_DEV))
    sch_names = d.synth.flux_tensor_shapes(d.FLUX_SCHNELL)
    n_w = 0
    for name, shape in d.synth.flux_tensor_shapes(d.FLUX_DEV).items():
        t = _seeded_weight(torch, name, shape, d)
        gm_dev.set_tensor(name, t)
        if name in sch_names:
            gm_sch.set_tensor(name, t)
        if wide:
            om.set_tensor(name, t.float().cpu().numpy())
        else:
            om.set_tensor_bf16(name, t.view(torch.int16).cpu().numpy().view(np.uint16))
        n_w += t.numel()
        del t
    gm_dev.assert_complete()
    gm_sch.assert_complete()
    assert gm_dev.is_guidance() and not gm_sch.is_guidance()
    print(f"full-size FLUX.1: {n_w / 1e9:.2f}e9 weights on the GPU (dev + schnell handles) and in the oracle ({'f32' if wide else 'bf16 bits, widened per block'}) "
          f"in {time.time() - t0:.0f} s")
    yield dict(torch=torch, d=d, orc=orc, gm_dev=gm_dev, gm_sch=gm_sch, om=om, wide=wide)
    gm_dev.close()
    gm_sch.close()


def test_c1_schnell_full_width_full_depth_matches_oracle(full_models):
    torch, d, orc, gm, om = (full_models[k] for k in ("torch", "d", "orc", "gm_sch", "om"))
    cfg = dict(d.FLUX_SCHNELL)
    B, T = 1, 256
    rng = np.random.default_rng(78)
    lat = rng.standard_normal((B, 16, 32, 32)).astype(np.float32)  # 256x256 image -> 32x32 latent -> S = 256
    t5 = bf16_round(rng.standard_normal((B, T, cfg["joint_attention_dim"])).astype(np.float32))
    clip = rng.standard_normal((B, cfg["pooled_projection_dim"])).astype(np.float32)
    img, ids = orc.pack_latents(lat)
    txt_ids = np.zeros((B, T, 3), np.float32)
    ts = orc.get_timesteps(4, False, 0.0, 1.0)  # schnell: no dynamic shifting, shift = 1.0 (scheduler.rs:22-51)
    got = host(gm.denoise(dev(img), dev(ids), dev(t5, torch.bfloat16), dev(txt_ids), dev(clip), None, ts))
    # With 4 steps the modulation precompute takes the f32 GEMV passes (<= 4 rows).  The path the 50-step runs take — ONE GEMM
    # over the 6.5 GB (344 D x D) modulation matrix for all steps' rows — is checked against those passes here at the full
    # model (12 steps): same weights, bf16 vs f32 silu(vec) input, so rounding-level agreement; round 3 found the GEMM reading
    # the rows beyond 4 GiB (single blocks, final layer) from a wrapped offset — an O(1) difference this comparison catches.
    from diffusion_rs_amd import _lib as L
    ts12 = orc.get_timesteps(12, False, 0.0, 1.0)
    args12 = (dev(img), dev(ids), dev(t5, torch.bfloat16), dev(txt_ids), dev(clip), None, ts12)
    L.check(gm.lib.fmi_flux_set_modulation_gemm(gm.h, 0))
    a12 = host(gm.denoise(*args12))
    L.check(gm.lib.fmi_flux_set_modulation_gemm(gm.h, 1))
    b12 = host(gm.denoise(*args12))
    e12 = rel_l2(b12, a12)
    print(f"12-step denoise at full size, modulation as one GEMM vs f32 GEMV passes: rel-L2 {e12:.3e}")
    assert e12 <= 5e-3
    t0 = time.time()
    ref = om.denoise(img, ids, t5, txt_ids, clip, None, ts)  # the dev tensors without guidance = the schnell model
    t_or = time.time() - t0
    err, moved = rel_l2(got, ref), rel_l2(ref, img)
    print(f"C1 in full (FLUX.1-schnell, D=3072, 19+38 blocks, S=T=256, 4 steps): latents rel-L2 {err:.3e} "
          f"(the loop moved them by {moved:.3f}; oracle {t_or:.0f} s)")
    assert np.isfinite(got).all() and err <= 3e-2


def test_c2_dev_1024_forward_and_two_step_trajectory_of_the_full_model_match_oracle(full_models):
    """BASELINE configs[1] — the headline config — at FULL size for one model evaluation: FLUX.1-dev, D = 3072, 19 + 38 blocks,
    S = 4096 image tokens (1024 x 1024), T = 512 text tokens, guidance 3.5: `Flux::forward` (model.rs:790-833) on the GPU against
    the f32 CPU oracle (7.4e13 FLOP: about two minutes on the box's host cores).  Every production launch shape — the
    4608 x 21504 x 3072 single-block projection with its fused q|k|v relayout, the 4608-token attention of 24 heads, the gated
    residual GEMMs at K = 15360 — is in this one comparison.  Tolerance: rel-L2 <= 2e-2 (the full-depth bar of
    tests/test_gpu_fulldepth.py).  Then two Euler steps of the real 50-step schedule with the one-GEMM modulation path forced (see below):
    latents <= 3e-2, the update itself <= 2e-2.  Needs the oracle's f32 weights in memory (48 GB): skipped on hosts without it."""
    if not full_models["wide"]:
        pytest.skip("the host cannot hold the oracle's 48 GB of f32 weights (the paged variant would take ~15 minutes for this forward)")
    torch, d, orc, gm, om = (full_models[k] for k in ("torch", "d", "orc", "gm_dev", "om"))
    cfg = dict(d.FLUX_DEV)
    B, T = 1, 512
    rng = np.random.default_rng(79)
    lat = rng.standard_normal((B, 16, 128, 128)).astype(np.float32)  # 1024x1024 image -> 128x128 latent -> S = 4096
    t5 = bf16_round(rng.standard_normal((B, T, cfg["joint_attention_dim"])).astype(np.float32))
    clip = rng.standard_normal((B, cfg["pooled_projection_dim"])).astype(np.float32)
    img, ids = orc.pack_latents(lat)
    assert img.shape[1] == 4096
    txt_ids = np.zeros((B, T, 3), np.float32)
    g = np.array([3.5], np.float32)
    # the first three timesteps of the REAL 50-step schedule of this resolution (dynamic shifting, mu from S = 4096)
    sched = d.SchedulerConfig()
    ts3 = [float(v) for v in sched.get_timesteps(50, sched.calculate_shift(4096))[:3]]
    t = np.array([ts3[0]], np.float32)
    got = host(gm.forward(dev(img), dev(ids), dev(t5, torch.bfloat16), dev(txt_ids), dev(t), dev(clip), dev(g)))
    t0 = time.time()
    ref = om.forward(img, ids, t5, txt_ids, t, clip, g)
    t_or = time.time() - t0
    err = rel_l2(got, ref)
    print(f"C2 in full (FLUX.1-dev, D=3072, 19+38 blocks, S=4096 + T=512 tokens), one Flux::forward: rel-L2 {err:.3e} (oracle {t_or:.0f} s)")
    assert np.isfinite(got).all() and err <= 2e-2
    full_models["c2"] = (img, ids, t5, txt_ids, t, clip, g, ref)  # the int8 test below measures its mode on this very forward
    # opt-in fmi_flux_set_fp8_attention(m, 2): the same forward with q and k handed to the attention as e4m3 — what the reduced operand costs at full size
    gm.set_fp8_attention(2)
    try:
        got_qk8 = host(gm.forward(dev(img), dev(ids), dev(t5, torch.bfloat16), dev(txt_ids), dev(t), dev(clip), dev(g)))
    finally:
        gm.set_fp8_attention(1)
    e_qk8 = rel_l2(got_qk8, ref)
    print(f"  the same with e4m3 q, k in front of QK^T (opt-in, bf16 block linears): rel-L2 {e_qk8:.3e}; against the bf16-operand result {rel_l2(got_qk8, got):.3e}")
    assert np.isfinite(got_qk8).all() and e_qk8 <= 2e-2 and not np.array_equal(got_qk8, got)
    # --- and the TRAJECTORY: two Euler steps of the 50-step schedule at full size against the oracle's, with the
    # modulation of both steps computed as ONE GEMM over the 6.5 GB (344 D x D) matrix — the path every 50-step run takes and the one
    # that carried the silent > 4 GiB offset wrap of round 3 — forced at 2 rows by fmi_flux_set_modulation_gemm(2); until now that
    # path was compared at full size only with the library's own GEMV passes.  Sampler::sample, pipelines/sampling.rs:25-48;
    # Modulation1/2, model.rs:211-300.  The oracle's first step is the forward above (img + pred * f32(dt), orc_flux_denoise's update).
    from diffusion_rs_amd import _lib as L
    L.check(gm.lib.fmi_flux_set_modulation_gemm(gm.h, 2))
    try:
        got2 = host(gm.denoise(dev(img), dev(ids), dev(t5, torch.bfloat16), dev(txt_ids), dev(clip), dev(g), ts3))
        mod_ms = None
        gm.set_profiling(True)
        gm.denoise(dev(img), dev(ids), dev(t5, torch.bfloat16), dev(txt_ids), dev(clip), dev(g), ts3)
        mod_ms = gm.phase_ms().get("modulation")
        gm.set_profiling(False)
    finally:
        L.check(gm.lib.fmi_flux_set_modulation_gemm(gm.h, 1))
    t0 = time.time()
    ref1 = img + ref * np.float32(ts3[1] - ts3[0])
    ref2 = om.denoise(ref1, ids, t5, txt_ids, clip, g, ts3[1:])
    t_or2 = time.time() - t0
    e2, moved = rel_l2(got2, ref2), rel_l2(ref2, img)
    print(f"C2 in full, 2 Euler steps of the 50-step schedule with the one-GEMM modulation path forced: latents rel-L2 {e2:.3e} "
          f"(the two steps moved them by {moved:.3e}; update alone: {rel_l2(got2 - img, ref2 - img):.3e}; modulation phase {mod_ms} ms; oracle {t_or2:.0f} s)")
    assert np.isfinite(got2).all() and e2 <= 3e-2
    full_models["c2_traj"] = (ts3, ref2)
    # the latents barely move in 2 of 50 steps, so the bar that means something is on the UPDATE (sum of pred * dt): the per-forward one
    assert rel_l2(got2 - img, ref2 - img) <= 2e-2


def test_c5_shape_1280x720_bf16_forward_of_the_full_model_matches_oracle(full_models):
    """BASELINE configs[4]'s SHAPE in bf16 at full size: 1280 x 720 -> latent 90 x 160 -> S = 45 * 80 = 3600 image tokens + T = 512 text tokens = 4112 — the
    ragged case of every launch: 4112 = 16 x 256 + 16 (a 17th query block of 16 rows), = 64 x 64 + 16 (a last KV tile of 16 keys, masked inside
    the attention stream), GEMM row tiles of 16 ragged rows.  One `Flux::forward` of FLUX.1-dev (19 + 38 blocks) against the f32 oracle, rel-L2 <= 2e-2
    (model.rs:790-833).  Round 4: the first full-size check of the lock-step attention's ragged paths inside the model."""
    if not full_models["wide"]:
        pytest.skip("the host cannot hold the oracle's 48 GB of f32 weights")
    torch, d, orc, gm, om = (full_models[k] for k in ("torch", "d", "orc", "gm_dev", "om"))
    cfg = dict(d.FLUX_DEV)
    B, T = 1, 512
    rng = np.random.default_rng(85)
    lat = rng.standard_normal((B, 16, 90, 160)).astype(np.float32)
    t5 = bf16_round(rng.standard_normal((B, T, cfg["joint_attention_dim"])).astype(np.float32))
    clip = rng.standard_normal((B, cfg["pooled_projection_dim"])).astype(np.float32)
    img, ids = orc.pack_latents(lat)
    assert img.shape[1] == 3600
    txt_ids = np.zeros((B, T, 3), np.float32)
    sched = d.SchedulerConfig()
    t = np.array([float(sched.get_timesteps(50, sched.calculate_shift(3600))[10])], np.float32)
    g = np.array([3.5], np.float32)
    got = host(gm.forward(dev(img), dev(ids), dev(t5, torch.bfloat16), dev(txt_ids), dev(t), dev(clip), dev(g)))
    t0 = time.time()
    ref = om.forward(img, ids, t5, txt_ids, t, clip, g)
    err = rel_l2(got, ref)
    print(f"C5 shape in bf16 (FLUX.1-dev in full, S=3600 + T=512 = 4112 tokens), one Flux::forward: rel-L2 {err:.3e} (oracle {time.time() - t0:.0f} s)")
    assert np.isfinite(got).all() and err <= 2e-2
    full_models["c5"] = (img, ids, t5, txt_ids, t, clip, g, ref)  # the int8 test measures its mode on this ragged shape too


def _decode_u8(d, orc, gv, ov, lat_gpu, lat_ref, h, w):
    """latents (1, S, 64) -> unpack + affine -> the real-config VAE -> u8, on the GPU for `lat_gpu` and in the oracle for `lat_ref`
    (pipelines/flux/mod.rs:320-332)."""
    import torch
    sf, sh = d.VAE_FLUX["scaling_factor"], d.VAE_FLUX["shift_factor"]
    z_g = d.unpack_latents(torch.from_numpy(lat_gpu).cuda(), 16, h, w, sf, sh)
    u_g = d.postprocess_u8(gv.decode(z_g)).cpu().numpy()
    z_r = orc.unpack_latents(lat_ref, 16, h, w) * np.float32(1.0 / sf) + np.float32(sh)
    u_r = orc.postprocess_u8(ov.decode(z_r.astype(np.float32)))
    diff = np.abs(u_g.astype(np.int32) - u_r.astype(np.int32))
    return int(diff.max()), float((diff <= 2).mean()), u_r


TRAJ_MARKS = (10, 25, 50)
# what the int8 mode is held to over the 50 steps (stated after measuring): the bar of the 8-bit modes on the latents at every mark
INT8_TRAJ_BAR = 3e-2
INT8_TRAJ_U8_BAR = 0.98  # (99.06 % measured with the decoder's f32 trunk; 97.5 % before it)


def test_c2_fifty_step_trajectory_of_the_full_model_matches_oracle(full_models):
    """The headline metric is a 50-STEP image: `Sampler::sample` (pipelines/sampling.rs:25-48) around the step
    closure (pipelines/flux/mod.rs:305-332) for FLUX.1-dev at its true width and depth — D = 3072, 19 + 38 blocks, every block its
    own weights — over the REAL 50-step dynamic-shift schedule, guidance 3.5, against the f32 oracle's 50 steps.  Token count chosen so
    that the oracle's 50 model evaluations fit the suite: a 192 x 256 image, S = 192 image + T = 64 text tokens (3.4e12 FLOP per step).
    The GPU runs the production path — fmi_flux_denoise, all 50 modulation rows through ONE GEMM over the 6.5 GB matrix, f32 latents —
    and is stopped after 10, 25 and 50 steps (three calls from the same start) to show how the distance to the oracle's trajectory
    grows.  Bars: latents rel-L2 <= 3e-2 at every mark; the u8 image after the real-config VAE decode (16 x 24 x 32
    latent -> 192 x 256 pixels) within 2 of the oracle's on >= 99 % of the values."""
    if not full_models["wide"]:
        pytest.skip("the host cannot hold the oracle's 48 GB of f32 weights")
    torch, d, orc, gm, om = (full_models[k] for k in ("torch", "d", "orc", "gm_dev", "om"))
    cfg = dict(d.FLUX_DEV)
    h, w, T = 24, 32, 64
    rng = np.random.default_rng(90)
    lat = rng.standard_normal((1, 16, h, w)).astype(np.float32)
    t5 = bf16_round(rng.standard_normal((1, T, cfg["joint_attention_dim"])).astype(np.float32))
    clip = rng.standard_normal((1, cfg["pooled_projection_dim"])).astype(np.float32)
    img, ids = orc.pack_latents(lat)
    S = img.shape[1]
    assert S == 192
    txt_ids = np.zeros((1, T, 3), np.float32)
    g = np.array([3.5], np.float32)
    # the schedule from the ORACLE's restatement (flux/sampling.rs:70-80, scheduler.rs:22-51); the product's must be the same numbers
    ts = [float(v) for v in orc.get_timesteps(50, True, orc.calculate_shift(S), 1.0)]
    sched = d.SchedulerConfig()
    assert len(ts) == 51 and np.abs(np.array(sched.get_timesteps(50, sched.calculate_shift(S))) - np.array(ts)).max() <= 1e-15
    args = (dev(ids), dev(t5, torch.bfloat16), dev(txt_ids), dev(clip), dev(g))
    got = {n: host(gm.denoise(dev(img), *args, ts[:n + 1])) for n in TRAJ_MARKS}
    t0 = time.time()
    ref, cur, at = {}, img, 0
    for n in TRAJ_MARKS:  # the oracle's ONE trajectory, read out at the marks
        cur = om.denoise(cur, ids, t5, txt_ids, clip, g, ts[at:n + 1])
        ref[n], at = cur.copy(), n
    t_or = time.time() - t0
    drift = {n: rel_l2(got[n], ref[n]) for n in TRAJ_MARKS}
    moved = {n: rel_l2(ref[n], img) for n in TRAJ_MARKS}
    print("C2 trajectory in full (FLUX.1-dev, D=3072, 19+38 blocks, S=192 + T=64, the real 50-step schedule), bf16: latents rel-L2 vs the f32 oracle after "
          + ", ".join(f"{n} steps {drift[n]:.3e} (moved {moved[n]:.2f})" for n in TRAJ_MARKS) + f"  (oracle {t_or:.0f} s)")
    for n in TRAJ_MARKS:
        assert np.isfinite(got[n]).all() and drift[n] <= 3e-2, (n, drift[n])
    # the image: real-config VAE decode + u8 on both sides
    vsd = d.synth.vae_state_dict_numpy(d.VAE_FLUX, seed=4)
    gv, ov = d.AutoEncoderKl(d.VAE_FLUX), orc.Vae(d.VAE_FLUX)
    gv.load_state_dict(vsd)
    ov.load(vsd)
    mx, frac, u_ref = _decode_u8(d, orc, gv, ov, got[50], ref[50], h, w)
    mx_v, frac_v, _ = _decode_u8(d, orc, gv, ov, ref[50], ref[50], h, w)  # the decoder's own share: the GPU VAE on the ORACLE's latents
    sat = float(((u_ref == 0) | (u_ref == 255)).mean())
    print(f"  u8 image after 50 steps + VAE (192 x 256): max |d| {mx}, within 2 on {frac:.4%} (the VAE alone on the oracle's latents: max {mx_v}, {frac_v:.4%}; "
          f"{sat:.1%} of the oracle's values saturated)")
    assert frac >= 0.999  # (the decoder's f32 trunk, round 5)
    full_models["traj50"] = dict(img=img, ids=ids, t5=t5, txt_ids=txt_ids, clip=clip, g=g, ts=ts, ref=ref, hw=(h, w), gv=gv, ov=ov, bf16=drift)


def test_batch_of_8_at_full_size_equals_the_samples_run_alone(full_models):
    """`Pipeline.MAX_BATCH` = 8 at the headline shape: FLUX.1-dev in full, 8 samples x (4096 + 512) tokens in ONE denoise call (36 864
    rows per launch, a 1.6 GB fused-projection buffer, 8 x 3 = 24 rows in the modulation precompute) — every row of every kernel is
    computed independently of the rows it shares a launch with, so each sample must equal the same sample run alone, bit for bit.
    (Round 3 found two size bugs that only a full-size, large-batch run exposes: 32-bit operand offsets beyond 4 GiB and an LDS
    staging limit of the embedder GEMV at B > 5.)"""
    torch, d, orc, gm = (full_models[k] for k in ("torch", "d", "orc", "gm_dev"))
    cfg = dict(d.FLUX_DEV)
    B, T = 8, 512
    rng = np.random.default_rng(82)
    lat = rng.standard_normal((B, 16, 128, 128)).astype(np.float32)
    t5 = bf16_round(rng.standard_normal((B, T, cfg["joint_attention_dim"])).astype(np.float32))
    clip = rng.standard_normal((B, cfg["pooled_projection_dim"])).astype(np.float32)
    img, ids = orc.pack_latents(lat)
    txt_ids = np.zeros((B, T, 3), np.float32)
    g = np.full((B,), 3.5, np.float32)
    sched = d.SchedulerConfig()
    ts = sched.get_timesteps(3, sched.calculate_shift(4096))
    run = lambda sl: host(gm.denoise(dev(img[sl]), dev(ids[sl]), dev(t5[sl], torch.bfloat16), dev(txt_ids[sl]), dev(clip[sl]), dev(g[sl]), ts))
    # the modulation precompute picks its path by row count (steps x batch: 24 rows -> one GEMM on bf16 silu(vec); 3 rows -> f32 GEMV
    # passes), so bit-identity is asked for with the path pinned, and the default paths are compared to rounding
    from diffusion_rs_amd import _lib as L
    L.check(gm.lib.fmi_flux_set_modulation_gemm(gm.h, 0))
    try:
        full = run(slice(0, 8))
        assert np.isfinite(full).all()
        for i in (0, 5, 7):
            one = run(slice(i, i + 1))
            nbad = int((one[0].view(np.uint32) != full[i].view(np.uint32)).sum())
            assert nbad == 0, (i, nbad)
    finally:
        L.check(gm.lib.fmi_flux_set_modulation_gemm(gm.h, 1))
    dflt = run(slice(0, 8))
    e = rel_l2(dflt, full)
    print(f"batch 8 at 1024x1024 (36 864 token rows per launch), 3 steps: samples 0, 5, 7 identical to their single-sample runs; "
          f"default modulation path (one GEMM, 24 rows) vs the GEMV passes: rel-L2 {e:.2e}")
    assert e <= 5e-3


def test_c3_nf4_full_model_forward_matches_oracle_on_dequantised_weights(full_models):
    """BASELINE configs[2] (C3: Q4-bnb) with the FULL model: every block and modulation Linear of FLUX.1-dev as bitsandbytes nf4
    (blocksize 64; packed-only policy: 7.4 GiB resident, no bf16 copy), one `Flux::forward` at 1024 image + 256 text tokens — above 383 rows, so the
    block Linears take the per-call expansion + dense GEMM and the 50-row-class launches the fused dequant-GEMM — against the
    oracle on the dequantised weights (BnbLinear::forward = dequantise + matmul, bitsandbytes/mod.rs:293-312; the dequantisation
    is the library's bit-exact kDequantizeBlockwise).  Tolerance: rel-L2 <= 2e-2."""
    if not full_models["wide"]:
        pytest.skip("needs a second 48 GB oracle weight set")
    import ctypes as C
    torch, d, orc = (full_models[k] for k in ("torch", "d", "orc"))
    from diffusion_rs_amd import _lib as L
    lib = L.load()
    cfg = dict(d.FLUX_DEV)
    gq, oq = d.FluxModel(cfg), orc.Flux(cfg)
    nq = 0
    for name, shape in d.synth.flux_tensor_shapes(cfg).items():
        t = _seeded_weight(torch, name, shape, d)
        if name.endswith(".weight") and (d.synth.is_block_linear(name) or "norm" in name and "linear" in name):
            packed, absmax = d.synth.quantize_nf4_device(t, 64)
            gq.set_linear_bnb4(name[:-len(".weight")], packed, absmax, 64, "nf4", shape[0], shape[1])
            deq = torch.empty(shape, dtype=torch.bfloat16, device="cuda")
            p = lambda x: C.c_void_p(x.data_ptr())
            for r0 in range(0, shape[0], 65536):  # (the stand-alone dequant counts elements in 32 bits: the 3.2e9-weight modulation matrix in pieces)
                rows = min(65536, shape[0] - r0)
                lib.dequantize_blockwise_bf16_nf4(None, C.c_void_p(packed.data_ptr() + r0 * shape[1] // 2), C.c_void_p(absmax.data_ptr() + r0 * shape[1] // 64 * 4),
                                                  C.c_void_p(deq.data_ptr() + r0 * shape[1] * 2), 64, rows * shape[1], None)
            torch.cuda.synchronize()
            oq.set_tensor(name, deq.float().cpu().numpy())
            nq += 1
            del packed, absmax, deq
        else:
            gq.set_tensor(name, t)
            oq.set_tensor(name, t.float().cpu().numpy())
        del t
    gq.assert_complete()
    bufs = gq.state_buffers()
    assert bufs[1][1] == 0 and bufs[2][1] == 0  # no bf16 MOD / BLOCKS arena
    rng = np.random.default_rng(80)
    lat = rng.standard_normal((1, 16, 64, 64)).astype(np.float32)  # 512 x 512 -> S = 1024
    t5 = bf16_round(rng.standard_normal((1, 256, cfg["joint_attention_dim"])).astype(np.float32))
    clip = rng.standard_normal((1, cfg["pooled_projection_dim"])).astype(np.float32)
    img, ids = orc.pack_latents(lat)
    txt_ids = np.zeros((1, 256, 3), np.float32)
    t, g = np.array([0.6], np.float32), np.array([3.5], np.float32)
    gq.set_quant_dense_cache(0)  # packed only (the opt-in since round 5): per-call expansion of the large launches
    got = host(gq.forward(dev(img), dev(ids), dev(t5, torch.bfloat16), dev(txt_ids), dev(t), dev(clip), dev(g)))
    resident = gq.size_in_bytes() / 2**30
    bufs = gq.state_buffers()
    assert bufs[1][1] == 0 and bufs[2][1] == 0  # still no bf16 arena
    gq.set_quant_dense_cache(-1)  # the default: by memory — this 288 GB part expands the block matrices of the large launches once
    got_d = host(gq.forward(dev(img), dev(ids), dev(t5, torch.bfloat16), dev(txt_ids), dev(t), dev(clip), dev(g)))
    bufs = gq.state_buffers()
    assert np.array_equal(got_d.view(np.uint32), got.view(np.uint32))  # same bits
    assert bufs[2][1] > 0 and bufs[1][1] == 0  # the block arena appeared, the 6.5 GB modulation matrix stays packed
    print(f"  default policy (by memory): {gq.size_in_bytes() / 2**30:.1f} GiB resident, the same bits as packed-only ({resident:.1f} GiB)")
    gq.close()
    t0 = time.time()
    ref = oq.forward(img, ids, t5, txt_ids, t, clip, g)
    err = rel_l2(got, ref)
    print(f"C3 with the full model ({nq} nf4 Linears, {resident:.1f} GiB resident), one Flux::forward at S=1024 + T=256: rel-L2 {err:.3e} (oracle {time.time() - t0:.0f} s)")
    assert np.isfinite(got).all() and err <= 2e-2
    del oq


def test_c2_int8_full_model_forward_is_within_the_8_bit_tolerance(full_models):
    """The int8 mode (fmi_flux_quantize_int8, default mask: every block linear but the double blocks' MLP; round 4) with the FULL model at the
    HEADLINE size — FLUX.1-dev, 19 + 38 blocks, S = 4096 + T = 512 tokens, one `Flux::forward` — against the f32 oracle: rel-L2 <= 3e-2, the bar
    VERDICT r3 set for an 8-bit mode (the e4m3 mode on the same weights: 1.0e-1; the bf16 path: 4.6e-3).  Then, at 384 + 128 tokens (where the
    oracle's second pass is affordable: it quantises 12e9 weights), against the oracle's restatement of the same recipe with the same mask:
    closer to it than the recipe is to f32.  The recipe is this library's own (parity unpinned by the reference).  A third GPU handle: the dev
    handle must stay bf16 for the fp8 test that runs last."""
    if not full_models["wide"]:
        pytest.skip("the host cannot hold the oracle's 48 GB of f32 weights (+ 48 GB of int8 codes as floats)")
    torch, d, orc, om = (full_models[k] for k in ("torch", "d", "orc", "om"))
    cfg = dict(d.FLUX_DEV)
    t0 = time.time()
    g8 = d.FluxModel(cfg)
    for name, shape in d.synth.flux_tensor_shapes(d.FLUX_DEV).items():
        g8.set_tensor(name, _seeded_weight(torch, name, shape, d))
    g8.assert_complete()
    g8.quantize_int8()
    t_load = time.time() - t0
    try:
        if "c2" in full_models:
            img, ids, t5, txt_ids, t, clip, g, ref = full_models["c2"]
        else:  # run alone: the headline forward of the oracle (80 s)
            rng = np.random.default_rng(79)
            lat = rng.standard_normal((1, 16, 128, 128)).astype(np.float32)
            t5 = bf16_round(rng.standard_normal((1, 512, cfg["joint_attention_dim"])).astype(np.float32))
            clip = rng.standard_normal((1, cfg["pooled_projection_dim"])).astype(np.float32)
            img, ids = orc.pack_latents(lat)
            txt_ids = np.zeros((1, 512, 3), np.float32)
            g = np.array([3.5], np.float32)
            sched = d.SchedulerConfig()
            t = np.array([float(sched.get_timesteps(50, sched.calculate_shift(4096))[0])], np.float32)
            ref = om.forward(img, ids, t5, txt_ids, t, clip, g)
        got = host(g8.forward(dev(img), dev(ids), dev(t5, torch.bfloat16), dev(txt_ids), dev(t), dev(clip), dev(g)))
        err = rel_l2(got, ref)
        print(f"C2 in full in int8 mode (default mask 0x{d.flux.INT8_DEFAULT_MASK:02x}), one Flux::forward at S=4096 + T=512: rel-L2 vs the f32 oracle {err:.3e} "
              f"(third handle loaded and quantised in {t_load:.0f} s)")
        assert np.isfinite(got).all() and err <= 3e-2
        if "c2_traj" in full_models:  # two Euler steps of the 50-step schedule in int8 mode against the f32 oracle's two steps (computed by the C2 test above)
            ts3, ref2 = full_models["c2_traj"]
            got2 = host(g8.denoise(dev(img), dev(ids), dev(t5, torch.bfloat16), dev(txt_ids), dev(clip), dev(g), ts3))
            e2, eu = rel_l2(got2, ref2), rel_l2(got2 - img, ref2 - img)
            print(f"  2 Euler steps in int8 mode vs the f32 oracle's: latents {e2:.3e}, update alone {eu:.3e}")
            assert np.isfinite(got2).all() and e2 <= 3e-2 and eu <= 3e-2
        if "traj50" in full_models:  # the 50-step trajectory in int8 mode against the ORACLE's (f32), not against the bf16 path's
            tj = full_models["traj50"]
            a8 = (dev(tj["ids"]), dev(tj["t5"], torch.bfloat16), dev(tj["txt_ids"]), dev(tj["clip"]), dev(tj["g"]))
            g50 = {n: host(g8.denoise(dev(tj["img"]), *a8, tj["ts"][:n + 1])) for n in TRAJ_MARKS}
            d8 = {n: rel_l2(g50[n], tj["ref"][n]) for n in TRAJ_MARKS}
            mx8, frac8, _ = _decode_u8(d, orc, tj["gv"], tj["ov"], g50[50], tj["ref"][50], *tj["hw"])
            print("  50-step trajectory (S=192 + T=64) in int8 mode vs the f32 oracle: latents after "
                  + ", ".join(f"{n} steps {d8[n]:.3e} (bf16: {tj['bf16'][n]:.3e})" for n in TRAJ_MARKS)
                  + f"; u8 image max |d| {mx8}, within 2 on {frac8:.4%}")
            for n in TRAJ_MARKS:
                assert np.isfinite(g50[n]).all() and d8[n] <= INT8_TRAJ_BAR, (n, d8[n])
            assert frac8 >= INT8_TRAJ_U8_BAR
        if "c5" in full_models:  # BASELINE configs[4]'s shape (4112 tokens: ragged GEMM row tiles, a 16-row last query block, a 16-key last KV tile) in int8 mode
            i5, d5, t55, x5, tt5, c5, g5, r5 = full_models["c5"]
            got5 = host(g8.forward(dev(i5), dev(d5), dev(t55, torch.bfloat16), dev(x5), dev(tt5), dev(c5), dev(g5)))
            e5 = rel_l2(got5, r5)
            print(f"  C5 shape (S=3600 + T=512) in int8 mode vs the f32 oracle: {e5:.3e}")
            assert np.isfinite(got5).all() and e5 <= 3e-2
        # round 6: the CALIBRATED recipe (per-channel smoothing, fmi_flux_calibrate_int8 — what Pipeline(dtype=I8) and bench.py's int8 legs run) on the same forward
        # and the C5 shape: on this Gaussian checkpoint it must be neutral (the outlier-channel checkpoint is tests/test_gpu_outlier_stats.py)
        g8s = d.FluxModel(cfg)
        try:
            for name, shape in d.synth.flux_tensor_shapes(d.FLUX_DEV).items():
                g8s.set_tensor(name, _seeded_weight(torch, name, shape, d))
            g8s.calibrate_int8(True)
            for tt in (1.0, 0.75, 0.5, 0.25):
                g8s.forward(dev(img), dev(ids), dev(t5, torch.bfloat16), dev(txt_ids), dev(np.array([tt], np.float32)), dev(clip), dev(g))
            g8s.quantize_int8()
            es = rel_l2(host(g8s.forward(dev(img), dev(ids), dev(t5, torch.bfloat16), dev(txt_ids), dev(t), dev(clip), dev(g))), ref)
            line = f"  CALIBRATED (smoothed) int8 recipe on the same forward: {es:.3e}"
            assert es <= 3e-2
            if "c5" in full_models:
                es5 = rel_l2(host(g8s.forward(dev(i5), dev(d5), dev(t55, torch.bfloat16), dev(x5), dev(tt5), dev(c5), dev(g5))), r5)
                line += f"; at the C5 shape {es5:.3e}"
                assert es5 <= 3e-2
            if "traj50" in full_models:
                gs50 = host(g8s.denoise(dev(tj["img"]), *a8, tj["ts"]))
                es50 = rel_l2(gs50, tj["ref"][50])
                line += f"; 50-step latents (S=192 + T=64; calibrated at S=4096) {es50:.3e}"
                assert es50 <= INT8_TRAJ_BAR
            print(line)
        finally:
            g8s.close()
        # the other mask worth knowing at full size (round 5): with / without the double blocks' MLP-out in 8 bits — the knapsack of DESIGN 4.3c, measured
        other = d.flux.INT8_DEFAULT_MASK ^ d.flux.Q8_DOUBLE_MLP_OUT
        g8b = d.FluxModel(cfg)
        try:
            for name, shape in d.synth.flux_tensor_shapes(d.FLUX_DEV).items():
                g8b.set_tensor(name, _seeded_weight(torch, name, shape, d))
            g8b.quantize_int8(other)
            eo = rel_l2(host(g8b.forward(dev(img), dev(ids), dev(t5, torch.bfloat16), dev(txt_ids), dev(t), dev(clip), dev(g))), ref)
            line = f"  mask 0x{other:02x} on the same forward: {eo:.3e}"
            if "c5" in full_models:
                eo5 = rel_l2(host(g8b.forward(dev(i5), dev(d5), dev(t55, torch.bfloat16), dev(x5), dev(tt5), dev(c5), dev(g5))), r5)
                line += f"; at the C5 shape {eo5:.3e}"
            if "traj50" in full_models:
                go50 = host(g8b.denoise(dev(tj["img"]), *a8, tj["ts"]))
                line += f"; 50-step latents {rel_l2(go50, tj['ref'][50]):.3e}"
            print(line)
        finally:
            g8b.close()
        # --- the GPU implements the STATED recipe: the oracle with the same mask, at a token count its weight quantisation dominates
        rng = np.random.default_rng(83)
        lat = rng.standard_normal((1, 16, 32, 48)).astype(np.float32)  # 16 x 24 = 384 tokens
        t5s = bf16_round(rng.standard_normal((1, 128, cfg["joint_attention_dim"])).astype(np.float32))
        clips = rng.standard_normal((1, cfg["pooled_projection_dim"])).astype(np.float32)
        imgs, idss = orc.pack_latents(lat)
        txts = np.zeros((1, 128, 3), np.float32)
        ts, gs = np.array([0.6], np.float32), np.array([3.5], np.float32)
        t0 = time.time()
        reff = om.forward(imgs, idss, t5s, txts, ts, clips, gs)
        om.set_int8(True, d.flux.INT8_DEFAULT_MASK, attention=True)  # 384 / 128 tokens: every block on the fused epilogue, QK^T on e4m3 operands
        try:
            ref8 = om.forward(imgs, idss, t5s, txts, ts, clips, gs)
        finally:
            om.set_int8(False)
        got8 = host(g8.forward(dev(imgs), dev(idss), dev(t5s, torch.bfloat16), dev(txts), dev(ts), dev(clips), dev(gs)))
        e8, ef, noise = rel_l2(got8, ref8), rel_l2(got8, reff), rel_l2(ref8, reff)
        print(f"the same at S=384 + T=128: vs the int8 oracle {e8:.3e}, vs the f32 oracle {ef:.3e} (recipe noise {noise:.3e}; oracle {time.time() - t0:.0f} s)")
        # (the codes are chaotic in the inputs — bf16 intermediates move activations across rounding boundaries — so, as for the fp8 mode, the bar against
        # the recipe's oracle is the recipe's own noise: measured 2.1e-2 against 2.2e-2)
        assert np.isfinite(got8).all() and ef <= 3e-2 and e8 <= 1.25 * noise
    finally:
        g8.close()


def test_c5_fp8_full_model_forward_against_the_fp8_recipe(full_models):
    """BASELINE configs[4] (C5: fp8) with the FULL model — all 19 + 38 blocks in fp8 mode (e4m3 block Linears with per-channel /
    per-token scales, fp8 QK^T in the one-wave attention stream with the power-of-two score factor) — one `Flux::forward` at 384
    image + 128 text tokens (token counts multiples of 16, so every block runs the fp8 attention; the oracle's time here is the
    quantisation of 12e9 weights, not the tokens) against the oracle's restatement of the same recipe and against the f32 oracle.  The recipe is this
    library's own (parity unpinned by the reference) and its codes are chaotic in the inputs, so the model-level bar is the
    statistical one of tests/test_gpu_fullsize.py: no further from the f32 truth than the recipe itself (+25 %), and closer to the
    recipe's oracle than the recipe's own noise.  Runs LAST: it switches the shared handles to fp8."""
    if not full_models["wide"]:
        pytest.skip("the fp8 oracle keeps a second f32 image of every weight")
    torch, d, orc, gm, om = (full_models[k] for k in ("torch", "d", "orc", "gm_dev", "om"))
    cfg = dict(d.FLUX_DEV)
    rng = np.random.default_rng(81)
    lat = rng.standard_normal((1, 16, 32, 48)).astype(np.float32)  # 384 x 256 -> 16 x 24 = 384 tokens
    t5 = bf16_round(rng.standard_normal((1, 128, cfg["joint_attention_dim"])).astype(np.float32))
    clip = rng.standard_normal((1, cfg["pooled_projection_dim"])).astype(np.float32)
    img, ids = orc.pack_latents(lat)
    assert img.shape[1] == 384
    txt_ids = np.zeros((1, 128, 3), np.float32)
    t, g = np.array([0.6], np.float32), np.array([3.5], np.float32)
    t0 = time.time()
    ref = om.forward(img, ids, t5, txt_ids, t, clip, g)
    gm.quantize_fp8()
    full_models["gm_is_fp8"] = True
    got8 = host(gm.forward(dev(img), dev(ids), dev(t5, torch.bfloat16), dev(txt_ids), dev(t), dev(clip), dev(g)))
    om.set_fp8(True, attention=True)  # (left ON: the oracle caches its e4m3 weight images while the mode is on, and the full-size fp8 test below reuses them)
    ref8 = om.forward(img, ids, t5, txt_ids, t, clip, g)
    e8, ef, noise = rel_l2(got8, ref8), rel_l2(got8, ref), rel_l2(ref8, ref)
    print(f"C5 with the full model in fp8 mode, one Flux::forward at S=384 + T=128: vs the fp8 oracle {e8:.3e}, vs the f32 oracle {ef:.3e} "
          f"(recipe noise {noise:.3e}; oracle {time.time() - t0:.0f} s)")
    assert np.isfinite(got8).all() and ef <= 1.25 * noise and e8 <= noise


def test_c5_fp8_full_model_forward_at_1280x720_batch_2(full_models):
    """BASELINE configs[4] AT ITS OWN WORKLOAD (VERDICT r5 "next" 1 / `configs_untested`): FLUX.1-dev in fp8 mode, 1280 x 720 -> S = 3600 image + T = 512
    text tokens = 4112 (every ragged path: a 17th query block of 16 rows, a last KV tile of 16 keys, GEMM row tiles of 16 rows), **B = 2** — the per-GPU
    batch of "batch = 16 on 8 x MI355X" — in ONE `Flux::forward` call (8224 rows per launch).  Checked:
      * sample 0 (the inputs of the bf16 C5 test above, whose f32 oracle result is kept) against the oracle's restatement of the fp8 recipe at this size and
        against the f32 oracle, with the statistical bars of the S = 384 test: no further from f32 than the recipe itself (+ 25 %), closer to the recipe's
        oracle than the recipe's own noise;
      * sample 1 (other latents / text / pooled vector / timestep / guidance) through a size-independent property: its rows in the B = 2 launch equal the
        same sample run alone, bit for bit (rows of a launch are independent) — and so does sample 0;
      * the fp8 attention really ran on the one-wave stream at this ragged size (no fallback to the 8-wave kernel was counted).
    The recipe is this library's own (parity unpinned by the reference: it has no fp8 compute path); its distance to f32 (~1e-1) is OUTSIDE the
    8-bit bar of 3e-2 and is reported as such — profiles/r06_e4m3_mask_study.txt shows that no e4m3 subset of the linears fits it.  Runs after the
    S = 384 fp8 test: the shared GPU handle is already in fp8 mode and the oracle's e4m3 weight images are cached."""
    import json
    if not full_models["wide"]:
        print("\n!!! NOT RUN: test_c5_fp8_full_model_forward_at_1280x720_batch_2 needs the oracle's f32 weights + their e4m3 images (96 GB) on the host !!!")
        pytest.skip("the host cannot hold the fp8 oracle's weights")
    torch, d, orc, gm, om = (full_models[k] for k in ("torch", "d", "orc", "gm_dev", "om"))
    cfg = dict(d.FLUX_DEV)
    T = 512
    if "c5" in full_models:
        i0, ids, t50, txt_ids, t0, c0, g0, ref0 = full_models["c5"]
    else:  # run alone
        rng = np.random.default_rng(85)
        lat = rng.standard_normal((1, 16, 90, 160)).astype(np.float32)
        t50 = bf16_round(rng.standard_normal((1, T, cfg["joint_attention_dim"])).astype(np.float32))
        c0 = rng.standard_normal((1, cfg["pooled_projection_dim"])).astype(np.float32)
        i0, ids = orc.pack_latents(lat)
        txt_ids = np.zeros((1, T, 3), np.float32)
        sched = d.SchedulerConfig()
        t0 = np.array([float(sched.get_timesteps(50, sched.calculate_shift(3600))[10])], np.float32)
        g0 = np.array([3.5], np.float32)
        om.set_fp8(False)
        ref0 = om.forward(i0, ids, t50, txt_ids, t0, c0, g0)
    assert i0.shape[1] == 3600
    rng = np.random.default_rng(86)
    i1 = rng.standard_normal(i0.shape).astype(np.float32)
    t51 = bf16_round(rng.standard_normal(t50.shape).astype(np.float32))
    c1 = rng.standard_normal(c0.shape).astype(np.float32)
    sched = d.SchedulerConfig()
    t1 = np.array([float(sched.get_timesteps(50, sched.calculate_shift(3600))[40])], np.float32)
    g1 = np.array([2.0], np.float32)
    cat = lambda a, b: np.concatenate([a, b], 0)
    if not full_models.get("gm_is_fp8"):
        gm.quantize_fp8()
    fb0 = json.loads(gm.lib.fmi_device_info().decode())["fp8_attention_fallbacks"]
    run = lambda im, idd, tx, tid, tt, cc, gg: host(gm.forward(dev(im), dev(idd), dev(tx, torch.bfloat16), dev(tid), dev(tt), dev(cc), dev(gg)))
    got = run(cat(i0, i1), cat(ids, ids), cat(t50, t51), cat(txt_ids, txt_ids), cat(t0, t1), cat(c0, c1), cat(g0, g1))
    assert got.shape == (2, 3600, 64) and np.isfinite(got).all()
    one0 = run(i0, ids, t50, txt_ids, t0, c0, g0)
    one1 = run(i1, ids, t51, txt_ids, t1, c1, g1)
    fb1 = json.loads(gm.lib.fmi_device_info().decode())["fp8_attention_fallbacks"]
    nb0 = int((one0[0].view(np.uint32) != got[0].view(np.uint32)).sum())
    nb1 = int((one1[0].view(np.uint32) != got[1].view(np.uint32)).sum())
    t_0 = time.time()
    om.set_fp8(True, attention=True)
    try:
        ref8 = om.forward(i0, ids, t50, txt_ids, t0, c0, g0)
    finally:
        om.set_fp8(False)
    e8, ef, noise = rel_l2(got[0], ref8[0]), rel_l2(got[0], ref0[0]), rel_l2(ref8, ref0)
    print(f"C5 at its workload in fp8 mode (FLUX.1-dev in full, S=3600 + T=512, B=2 in one call): sample 0 vs the fp8 oracle {e8:.3e}, vs the f32 oracle {ef:.3e} "
          f"(recipe noise {noise:.3e}; OUTSIDE the 8-bit bar of 3e-2, reported as such); samples in the batch vs run alone: {nb0} / {nb1} differing values; "
          f"sample 1 vs sample 0 {rel_l2(got[1], got[0]):.2f}; fp8-attention fallbacks during the three calls: {fb1 - fb0}  (fp8 oracle {time.time() - t_0:.0f} s)")
    assert nb0 == 0 and nb1 == 0
    assert fb1 == fb0
    assert rel_l2(got[1], got[0]) > 0.5  # (the two samples are different problems)
    assert ef <= 1.25 * noise and e8 <= noise
