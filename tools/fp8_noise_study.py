#!/usr/bin/env python3
"""How far is an 8-bit block-linear recipe from the f32 semantics, and why?  (DESIGN 4.3 "the e4m3 noise floor")

CPU only: the oracle (oracle/flux_oracle.cpp) runs FLUX at the true width (D = 3072, 24 heads, MLP 12288) with a reduced depth, once in
f32 and once per quantiser with the SAME f32 GEMM behind it, so every difference is the quantiser's.  Also a one-GEMM table (Gaussian
operands, K = 3072) that separates the per-operand noise from what the model does with it.

    python tools/fp8_noise_study.py [--double 2 --single 4] [--tokens 12x16 --txt 64]
"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

MODES = [(1, "e4m3 per row, W and A (the library's recipe)"), (2, "e4m3 + E8M0 scale per 32 k (MX), W and A"), (3, "e4m3 per row, W only"),
         (4, "e4m3 per row, A only"), (6, "MX e4m3, A only"), (5, "int8 per row, W and A")]


def rel_l2(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / np.linalg.norm(b))


def one_gemm_table():
    """rel-L2 of x W^T for Gaussian x (256, K), W (512, K): what one Linear adds, by quantiser."""
    from oracle import oracle as orc
    rng = np.random.default_rng(0)
    K = 3072
    x = rng.standard_normal((256, K)).astype(np.float32)
    w = (0.02 * rng.standard_normal((512, K))).astype(np.float32)
    ref = x.astype(np.float64) @ w.astype(np.float64).T

    def e4m3(v):  # round-to-nearest-even onto the e4m3 grid, saturating (vectorised twin of orc_f32_to_e4m3, checked below)
        a = np.minimum(np.abs(v).astype(np.float32), 448.0)
        _, e = np.frexp(a)
        quantum = np.exp2(np.maximum(e - 1, -6) - 3).astype(np.float32)
        return np.sign(v) * np.minimum(np.rint(a / quantum) * quantum, 448.0)

    probe = rng.standard_normal(2000).astype(np.float32) * 100
    assert all(float(e4m3(np.float32(p))) == orc.e4m3_to_f32(orc.f32_to_e4m3(float(p))) for p in probe)

    def e4m3_rows(a):
        am = np.abs(a).max(1, keepdims=True)
        return e4m3(a * (448.0 / am)) * (am / 448.0)

    def mx(a):
        b = a.reshape(a.shape[0], -1, 32)
        am = np.abs(b).max(2, keepdims=True)
        sc = np.exp2(np.ceil(np.log2(am / 448.0))).astype(np.float32)
        return (e4m3(b / sc) * sc).reshape(a.shape)

    def i8(a):
        am = np.abs(a).max(1, keepdims=True)
        return np.rint(a * (127.0 / am)) * (am / 127.0)

    def bf(a):
        return orc.round_bf16(a)

    rows = []
    for name, q in (("bf16", bf), ("e4m3 per row", e4m3_rows), ("e4m3 MX block 32", mx), ("int8 per row", i8)):
        xq, wq = q(x).astype(np.float64), q(w).astype(np.float64)
        rows.append((name, rel_l2(xq, x), rel_l2(xq @ wq.T, ref), rel_l2(x.astype(np.float64) @ wq.T, ref), rel_l2(xq @ w.astype(np.float64).T, ref)))
    print("one GEMM, Gaussian operands, K = 3072 (rel-L2):")
    print(f"  {'quantiser':20s} {'operand':>9s} {'W and A':>9s} {'W only':>9s} {'A only':>9s}")
    for r in rows:
        print(f"  {r[0]:20s} {r[1]:9.2e} {r[2]:9.2e} {r[3]:9.2e} {r[4]:9.2e}")


def model_table(n_double, n_single, S_hw, T):
    import diffusion_rs_amd.synth as synth
    from oracle import oracle as orc
    from tests.util import flux_inputs
    cfg = dict(in_channels=64, pooled_projection_dim=768, joint_attention_dim=4096, num_attention_heads=24, num_layers=n_double,
               num_single_layers=n_single, guidance_embeds=True, axes_dim=[16, 56, 56], theta=10000)
    sd = synth.flux_state_dict_numpy(cfg, seed=5)
    om = orc.Flux(cfg)
    om.load(sd)
    img, ids, txt, txt_ids, y = flux_inputs(cfg, 1, S_hw, T, seed=9)
    t, g = np.array([0.6], np.float32), np.array([3.5], np.float32)
    t0 = time.time()
    ref = om.forward(img, ids, txt, txt_ids, t, y, g)
    print(f"Flux::forward, D = 3072, {n_double} double + {n_single} single blocks, {S_hw[0] * S_hw[1]} + {T} tokens (f32 forward {time.time() - t0:.0f} s); rel-L2 vs f32:")
    # bf16 operands for scale: round every block-linear input?  The library's bf16 path is measured on the GPU (4.9e-3 at full depth).
    for mode, name in MODES:
        om.set_fp8(True, study_mode=mode)
        out = om.forward(img, ids, txt, txt_ids, t, y, g)
        print(f"  mode {mode}  {name:48s} {rel_l2(out, ref):.3e}", flush=True)
    om.set_fp8(False)
    return om, (img, ids, txt, txt_ids, t, y, g), ref


MASK_BITS = ["double q|k|v", "double attention out", "double MLP in", "double MLP out", "single linear1", "single linear2"]


def mask_table(om, inputs, ref, masks, mode=5, attention=False):
    """An 8-bit recipe (mode 5 = int8, 1 = e4m3 per row, ...) on a SUBSET of the block linears (orc_flux_set_q8_mask): which linears carry the error?"""
    img, ids, txt, txt_ids, t, y, g = inputs
    om.set_fp8(True, attention=attention, study_mode=mode)
    recipe = dict(MODES).get(mode, {7: "int8 per row, one scale per segment of linear2's input"}.get(mode, f"study mode {mode}"))
    print(f"{recipe} on a subset of the block linears{' + e4m3 q, k in the attention (static scales)' if attention else ''}{', e4m3 P and V too' if attention == 2 else ''}; rel-L2 vs f32:")
    for mask in masks:
        om.set_q8_mask(mask)
        out = om.forward(img, ids, txt, txt_ids, t, y, g)
        names = ", ".join(n for i, n in enumerate(MASK_BITS) if mask >> i & 1)
        print(f"  mask 0x{mask:02x}  {rel_l2(out, ref):.3e}   {names}", flush=True)
    om.set_q8_mask(0x3f)
    om.set_fp8(False)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--double", type=int, default=2)
    ap.add_argument("--single", type=int, default=4)
    ap.add_argument("--tokens", default="12x16")
    ap.add_argument("--txt", type=int, default=64)
    ap.add_argument("--skip-model", action="store_true")
    ap.add_argument("--masks", default="", help="comma-separated hex masks of block linears for the int8 subset table, e.g. 3f,15,2a")
    ap.add_argument("--only-masks", action="store_true")
    ap.add_argument("--mask-mode", type=int, default=5, help="study mode of the mask table: 5 = the int8 recipe, 7 = the same with one scale per segment of linear2's input")
    ap.add_argument("--attention", type=int, nargs="?", const=1, default=0,
                    help="mask table: also q and k of the attention on e4m3 (mask 00 = the attention operands alone); 2 = P and V on e4m3 as well (study of an unbuilt recipe)")
    a = ap.parse_args()
    if not a.only_masks:
        one_gemm_table()
    if not a.skip_model:
        h, w = (int(v) for v in a.tokens.split("x"))
        if a.only_masks:
            del MODES[:]
        om, inputs, ref = model_table(a.double, a.single, (h, w), a.txt)
        if a.masks:
            mask_table(om, inputs, ref, [int(v, 16) for v in a.masks.split(",")], mode=a.mask_mode, attention=a.attention)
