"""Shared helpers for the parity tests (oracle = checker, HIP library = thing under test)."""
import numpy as np

SMALL_FLUX = dict(in_channels=64, pooled_projection_dim=64, joint_attention_dim=128, num_attention_heads=2, num_layers=2,
                  num_single_layers=2, guidance_embeds=True, axes_dim=[16, 56, 56], theta=10000)
SMALL_VAE = dict(in_channels=3, out_channels=3, block_out_channels=[64, 64, 128, 128], layers_per_block=1, latent_channels=16,
                 norm_num_groups=16, mid_block_add_attention=True, use_post_quant_conv=False, scaling_factor=0.3611, shift_factor=0.1159)


def bf16_round(a):
    a = np.ascontiguousarray(a, np.float32)
    u = a.view(np.uint32).astype(np.uint64)
    r = ((u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000).astype(np.uint32)
    return r.view(np.float32).reshape(a.shape)


def rel_l2(a, b):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


def dev(a, dtype=None):
    import torch
    t = torch.from_numpy(np.ascontiguousarray(a)).cuda()
    return t.to(dtype) if dtype is not None else t


def host(t):
    return t.detach().float().cpu().numpy()


def flux_inputs(cfg, B, S_hw, T, seed=1234):
    """Synthetic inputs of SURVEY §8d: img~N(0,1), txt~N(0,1), y~N(0,1), ids as State::new builds them."""
    rng = np.random.default_rng(seed)
    h2, w2 = S_hw
    S = h2 * w2
    img = rng.standard_normal((B, S, cfg["in_channels"])).astype(np.float32)
    txt = bf16_round(rng.standard_normal((B, T, cfg["joint_attention_dim"])).astype(np.float32))
    y = rng.standard_normal((B, cfg["pooled_projection_dim"])).astype(np.float32)
    ids = np.zeros((B, S, 3), np.float32)
    ids[:, :, 1] = np.repeat(np.arange(h2), w2)[None]
    ids[:, :, 2] = np.tile(np.arange(w2), h2)[None]
    txt_ids = np.zeros((B, T, 3), np.float32)
    return img, ids, txt, txt_ids, y


def pow2_attention_scale(n):
    """An f32 `scale` with f32(scale * f32(log2 e)) == 2^n bit for bit: what fmi_sdpa_fp8qk needs to take a one-wave stream (the score
    factor rides in the MFMA's E8M0 block scale; a factor that is not exactly a power of two runs on the 8-wave kernel and is counted)."""
    log2e = np.float32(1.4426950408889634)
    target = np.float32(2.0 ** n)
    c = np.float32(target / log2e)
    for _ in range(8):
        got = np.float32(c * log2e)
        if got == target:
            return float(c)
        c = np.nextafter(c, np.float32(np.inf if got < target else -np.inf), dtype=np.float32)
    raise AssertionError(f"no f32 scale with scale * log2(e) == 2^{n}")
