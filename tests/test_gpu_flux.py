"""Model-level parity: Flux::forward / Sampler::sample on the GPU (bf16 MFMA, f32 accumulate)
vs the f32 CPU oracle on identical synthetic weights and inputs.

Stated tolerances (SURVEY §8d "parity tolerance"): rel-L2 <= 1e-2 on one model evaluation,
<= 3e-2 on the latents after the Euler loop.
"""
import numpy as np
import pytest

from tests.util import SMALL_FLUX, dev, flux_inputs, host, rel_l2

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def models():
    import torch
    import diffusion_rs_amd as d
    from oracle import oracle as orc
    sd = d.synth.flux_state_dict_numpy(SMALL_FLUX, seed=0)
    gm = d.FluxModel(SMALL_FLUX)
    gm.load_state_dict(sd)
    om = orc.Flux(SMALL_FLUX)
    om.load(sd)
    return dict(torch=torch, d=d, gm=gm, om=om, sd=sd)


@pytest.mark.parametrize("B,S_hw,T", [(1, (8, 12), 40), (2, (6, 6), 64), (1, (16, 16), 77)])
def test_flux_forward_matches_oracle(models, B, S_hw, T):
    torch, gm, om = models["torch"], models["gm"], models["om"]
    img, ids, txt, txt_ids, y = flux_inputs(SMALL_FLUX, B, S_hw, T)
    t = np.linspace(0.9, 0.4, B).astype(np.float32)
    g = np.full(B, 3.5, np.float32)
    ref = om.forward(img, ids, txt, txt_ids, t, y, g)
    got = gm.forward(dev(img), dev(ids), dev(txt, torch.bfloat16), dev(txt_ids), dev(t), dev(y), dev(g))
    torch.cuda.synchronize()
    got = host(got)
    assert np.isfinite(got).all()
    err = rel_l2(got, ref)
    print(f"forward B={B} S={S_hw} T={T}: rel-L2 {err:.3e}")
    assert err <= 1e-2


def test_flux_forward_rejects_missing_guidance(models):
    torch, d, gm = models["torch"], models["d"], models["gm"]
    img, ids, txt, txt_ids, y = flux_inputs(SMALL_FLUX, 1, (4, 4), 16)
    with pytest.raises(d.FmiError):
        gm.forward(dev(img), dev(ids), dev(txt, torch.bfloat16), dev(txt_ids), dev(np.ones(1, np.float32)), dev(y), None)


def test_flux_denoise_matches_oracle(models):
    torch, d, gm, om = models["torch"], models["d"], models["gm"], models["om"]
    B, S_hw, T, steps = 1, (8, 8), 32, 4
    img, ids, txt, txt_ids, y = flux_inputs(SMALL_FLUX, B, S_hw, T, seed=7)
    g = np.full(B, 3.5, np.float32)
    sched = d.SchedulerConfig()
    ts = sched.get_timesteps(steps, sched.calculate_shift(S_hw[0] * S_hw[1]))
    ref = om.denoise(img, ids, txt, txt_ids, y, g, ts)
    got = host(gm.denoise(dev(img), dev(ids), dev(txt, torch.bfloat16), dev(txt_ids), dev(y), dev(g), ts))
    err = rel_l2(got, ref)
    print(f"denoise {steps} steps: rel-L2 {err:.3e}")
    assert err <= 3e-2
    # Sampler invariant: zero steps leave the latent untouched
    same = host(gm.denoise(dev(img), dev(ids), dev(txt, torch.bfloat16), dev(txt_ids), dev(y), dev(g), ts[:1]))
    np.testing.assert_array_equal(same, img)


def test_flux_batch_independence(models):
    """Samples of a batch are independent trajectories: B=2 equals two B=1 runs."""
    torch, gm = models["torch"], models["gm"]
    img, ids, txt, txt_ids, y = flux_inputs(SMALL_FLUX, 2, (6, 8), 24, seed=3)
    t = np.array([0.7, 0.7], np.float32)
    g = np.array([3.5, 3.5], np.float32)
    both = host(gm.forward(dev(img), dev(ids), dev(txt, torch.bfloat16), dev(txt_ids), dev(t), dev(y), dev(g)))
    for b in range(2):
        one = host(gm.forward(dev(img[b:b + 1]), dev(ids[b:b + 1]), dev(txt[b:b + 1], torch.bfloat16), dev(txt_ids[b:b + 1]), dev(t[b:b + 1]),
                              dev(y[b:b + 1]), dev(g[b:b + 1])))
        np.testing.assert_array_equal(both[b:b + 1], one)


def test_flux_nf4_blocks_match_dequantised_oracle(models):
    """C3 semantics: block AND modulation linears stored nf4 and run through the fused dequant-GEMM equal the
    oracle run on the dequantised (bf16) weights — BnbLinear::forward (bitsandbytes/mod.rs:301-312)."""
    torch, d = models["torch"], models["d"]
    from oracle import oracle as orc
    sd = dict(models["sd"])
    gq = d.FluxModel(SMALL_FLUX)
    oq = orc.Flux(SMALL_FLUX)
    D = 256
    for name, w in sd.items():
        is_quant_lin = name.endswith(".weight") and w.ndim == 2 and (("transformer_blocks." in name) or name == "norm_out.linear.weight")
        if is_quant_lin:  # what a bitsandbytes checkpoint quantises: every nn.Linear of the blocks (modulation included)
            packed, absmax = orc.quantize_blockwise_4bit(w.ravel(), 64, "nf4")
            wdq = orc.dequantize_blockwise(None, packed, absmax, 64, w.size, "nf4", "bf16").reshape(w.shape)
            gq.set_linear_bnb4(name[:-len(".weight")], packed, absmax, 64, "nf4", w.shape[0], w.shape[1])
            oq.set_tensor(name, wdq)
        else:
            gq.set_tensor(name, w)
            oq.set_tensor(name, w)
    gq.assert_complete()
    img, ids, txt, txt_ids, y = flux_inputs(SMALL_FLUX, 1, (8, 8), 32, seed=11)
    t = np.array([0.8], np.float32)
    g = np.array([3.5], np.float32)
    ref = oq.forward(img, ids, txt, txt_ids, t, y, g)
    got = host(gq.forward(dev(img), dev(ids), dev(txt, torch.bfloat16), dev(txt_ids), dev(t), dev(y), dev(g)))
    err = rel_l2(got, ref)
    print(f"nf4 forward (fused dequant-GEMM): rel-L2 {err:.3e}")
    assert err <= 1e-2
    # default: packed codes only — the bf16 BLOCKS arena was never allocated
    assert gq.state_buffers()[2][1] == 0 and gq.state_buffers()[3][1] > 0
    # opt-in expanded cache: each matrix dequantised once into the bf16 arena, then the dense MFMA kernels
    from diffusion_rs_amd import _lib as L
    gq.set_quant_dense_cache(True)
    got2 = host(gq.forward(dev(img), dev(ids), dev(txt, torch.bfloat16), dev(txt_ids), dev(t), dev(y), dev(g)))
    err2 = rel_l2(got2, ref)
    print(f"nf4 forward (dequant-once + dense): rel-L2 {err2:.3e}")
    assert err2 <= 1e-2
    # both paths multiply the same bf16 weights in the same order: identical bits
    np.testing.assert_array_equal(got2, got)
    assert gq.state_buffers()[2][1] > 0
    # a second call reuses the cached copies
    got3 = host(gq.forward(dev(img), dev(ids), dev(txt, torch.bfloat16), dev(txt_ids), dev(t), dev(y), dev(g)))
    np.testing.assert_array_equal(got3, got2)
    # re-setting a quantised linear invalidates its cached copy
    name = next(n for n in sd if n.endswith("ff.net.2.weight"))
    w = sd[name]
    packed, absmax = orc.quantize_blockwise_4bit((2.0 * w).ravel(), 64, "nf4")
    gq.set_linear_bnb4(name[:-len(".weight")], packed, absmax, 64, "nf4", w.shape[0], w.shape[1])
    got5 = host(gq.forward(dev(img), dev(ids), dev(txt, torch.bfloat16), dev(txt_ids), dev(t), dev(y), dev(g)))
    assert not np.array_equal(got5, got2)


def test_flux_mixed_quantised_and_dense_parts_of_a_fused_projection(models):
    """A checkpoint may quantise only some Linears of a fused matrix (to_q nf4, to_k / to_v dense — e.g. bnb's
    skip-modules): the quantised parts are expanded at finalisation and the matrix runs dense; nothing reads
    uninitialised packed rows."""
    torch, d = models["torch"], models["d"]
    from oracle import oracle as orc
    sd = dict(models["sd"])
    gq = d.FluxModel(SMALL_FLUX)
    oq = orc.Flux(SMALL_FLUX)
    nq = 0
    for name, w in sd.items():
        if name.endswith("attn.to_q.weight") or name.endswith("attn.add_k_proj.weight") or name.endswith("proj_mlp.weight"):
            packed, absmax = orc.quantize_blockwise_4bit(w.ravel(), 64, "nf4")
            wdq = orc.dequantize_blockwise(None, packed, absmax, 64, w.size, "nf4", "bf16").reshape(w.shape)
            gq.set_linear_bnb4(name[:-len(".weight")], packed, absmax, 64, "nf4", w.shape[0], w.shape[1])
            oq.set_tensor(name, wdq)
            nq += 1
        else:
            gq.set_tensor(name, w)
            oq.set_tensor(name, w)
    assert nq == 2 + 2 + 2 + 2  # to_q of 2 double + 2 single blocks, add_k_proj of 2 double, proj_mlp of 2 single
    gq.assert_complete()
    img, ids, txt, txt_ids, y = flux_inputs(SMALL_FLUX, 1, (8, 8), 32, seed=12)
    t = np.array([0.8], np.float32)
    g = np.array([3.5], np.float32)
    ref = oq.forward(img, ids, txt, txt_ids, t, y, g)
    got = host(gq.forward(dev(img), dev(ids), dev(txt, torch.bfloat16), dev(txt_ids), dev(t), dev(y), dev(g)))
    err = rel_l2(got, ref)
    print(f"mixed nf4 / dense parts: rel-L2 {err:.3e}")
    assert err <= 1e-2
    # a later dense set_tensor on a part of a quantised matrix takes effect too (the cached state is invalidated)
    name = next(n for n in sd if n.endswith("attn.to_q.weight"))
    gq.set_tensor(name, 2.0 * sd[name])
    got2 = host(gq.forward(dev(img), dev(ids), dev(txt, torch.bfloat16), dev(txt_ids), dev(t), dev(y), dev(g)))
    assert not np.array_equal(got2, got)


def test_flux_int8_scb_blocks_match_dequantised_oracle(models, tmp_path):
    """LLM.int8 linears (BnbLinear::Int8, bitsandbytes/mod.rs:104-134,293-300): block linears stored as int8 + SCB,
    loaded through the HF-bnb naming (`<prefix>.weight` int8, `<prefix>.SCB`), equal the oracle on w*SCB/127."""
    torch, d = models["torch"], models["d"]
    from oracle import oracle as orc
    from diffusion_rs_amd import loader
    sd = dict(models["sd"])
    oq = orc.Flux(SMALL_FLUX)
    tensors = []
    nq = 0
    for name, w in sd.items():
        if d.synth.is_block_linear(name) or name == "x_embedder.weight":  # one non-block linear too (expanded at load)
            scb = np.abs(w).max(axis=1).astype(np.float32)
            w8 = np.clip(np.rint(w / scb[:, None] * 127.0), -127, 127).astype(np.int8)
            wdq = orc.dequantize_8bit(w8, scb, w.shape[0], w.shape[1], "bf16").reshape(w.shape)
            oq.set_tensor(name, wdq)
            tensors.append((name, torch.from_numpy(w8)))
            tensors.append((name[:-len(".weight")] + ".SCB", torch.from_numpy(scb)))
            nq += 1
        else:
            oq.set_tensor(name, w)
            tensors.append((name, torch.from_numpy(w).to(torch.bfloat16)))
    gq = d.FluxModel(SMALL_FLUX)
    stats = loader.load_flux(gq, iter(tensors))
    assert stats["int8"] == nq and stats["bnb4"] == 0
    img, ids, txt, txt_ids, y = flux_inputs(SMALL_FLUX, 2, (8, 8), 32, seed=13)
    t = np.array([0.8, 0.3], np.float32)
    g = np.array([3.5, 3.5], np.float32)
    ref = oq.forward(img, ids, txt, txt_ids, t, y, g)
    got = host(gq.forward(dev(img), dev(ids), dev(txt, torch.bfloat16), dev(txt_ids), dev(t), dev(y), dev(g)))
    err = rel_l2(got, ref)
    print(f"int8 (SCB) forward, {nq} quantised linears: rel-L2 {err:.3e}")
    assert err <= 1e-2
    # the default path expands the int8 tiles inside the GEMM; the dense cache expands each matrix once with the stand-alone
    # dequant kernel and runs the dense GEMM: same bits
    # a larger latent (1152 image rows: above INT8_FUSED_MAX_ROWS the int8 matrices are expanded per call into a scratch and the
    # dense kernel runs; the 64 text rows stay on the fused stage)
    img2, ids2, txt2, txt_ids2, y2 = flux_inputs(SMALL_FLUX, 2, (24, 24), 32, seed=14)
    big = host(gq.forward(dev(img2), dev(ids2), dev(txt2, torch.bfloat16), dev(txt_ids2), dev(t), dev(y2), dev(g)))
    gq.set_quant_dense_cache(True)
    got2 = host(gq.forward(dev(img), dev(ids), dev(txt, torch.bfloat16), dev(txt_ids), dev(t), dev(y), dev(g)))
    np.testing.assert_array_equal(got, got2)
    big2 = host(gq.forward(dev(img2), dev(ids2), dev(txt2, torch.bfloat16), dev(txt_ids2), dev(t), dev(y2), dev(g)))
    np.testing.assert_array_equal(big, big2)


@pytest.mark.parametrize("B,S_hw,T", [(2, (8, 8), 32), (1, (8, 12), 40), (1, (16, 16), 48)])
def test_fused_qkv_relayout_is_bit_identical(models, B, S_hw, T):
    """QkNorm + RoPE + head-major / transposed relayout fused into the QKV GEMM epilogue (default) vs the stand-alone
    kernels over the stored projection: same arithmetic, bit-identical prediction.  (8,12)/T=40: the text stream's
    offsets are not multiples of 16, so it stays on the kernels while the image stream is fused."""
    torch, gm = models["torch"], models["gm"]
    from diffusion_rs_amd import _lib as L
    lib = L.load()
    img, ids, txt, txt_ids, y = flux_inputs(SMALL_FLUX, B, S_hw, T, seed=17)
    t = np.linspace(0.9, 0.5, B).astype(np.float32)
    g = np.full(B, 3.5, np.float32)
    args = (dev(img), dev(ids), dev(txt, torch.bfloat16), dev(txt_ids), dev(t), dev(y), dev(g))
    try:
        L.check(lib.fmi_flux_set_fused_qkv_relayout(gm.h, 0))
        plain = host(gm.forward(*args))
        L.check(lib.fmi_flux_set_fused_qkv_relayout(gm.h, 1))
        fused = host(gm.forward(*args))
    finally:
        L.check(lib.fmi_flux_set_fused_qkv_relayout(gm.h, 1))
    np.testing.assert_array_equal(fused, plain)


@pytest.mark.parametrize("B,steps", [(1, 6), (2, 5)])
def test_modulation_gemm_matches_gemv_passes(models, B, steps):
    """fmi_flux_denoise precomputes all steps' modulation vectors: one MFMA GEMM (silu(vec) rounded to bf16, default)
    vs f32 GEMV passes of 4 rows.  Same weights, f32 accumulate; the only difference is the bf16 rounding of the
    GEMM's input, far inside the loop's 3e-2 bar against the oracle."""
    torch, d, gm, om = models["torch"], models["d"], models["gm"], models["om"]
    from diffusion_rs_amd import _lib as L
    lib = L.load()
    S_hw, T = (8, 8), 32
    img, ids, txt, txt_ids, y = flux_inputs(SMALL_FLUX, B, S_hw, T, seed=11)
    g = np.full(B, 3.5, np.float32)
    sched = d.SchedulerConfig()
    ts = sched.get_timesteps(steps, sched.calculate_shift(S_hw[0] * S_hw[1]))
    args = (dev(img), dev(ids), dev(txt, torch.bfloat16), dev(txt_ids), dev(y), dev(g), ts)
    try:
        L.check(lib.fmi_flux_set_modulation_gemm(gm.h, 0))
        a = host(gm.denoise(*args))
        L.check(lib.fmi_flux_set_modulation_gemm(gm.h, 1))
        b = host(gm.denoise(*args))
    finally:
        L.check(lib.fmi_flux_set_modulation_gemm(gm.h, 1))
    ref = om.denoise(img, ids, txt, txt_ids, y, g, ts)
    print(f"modulation GEMM vs GEMV passes: rel-L2 {rel_l2(b, a):.2e}; vs oracle {rel_l2(b, ref):.2e} / {rel_l2(a, ref):.2e}")
    assert rel_l2(b, a) <= 2e-3 and rel_l2(b, ref) <= 3e-2
