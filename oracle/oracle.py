"""ctypes binding of the CPU oracle (oracle/libflux_oracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg — never by the product package (diffusion-rs_amd/).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libflux_oracle.so")


def build(force=False):
    src = os.path.join(_HERE, "flux_oracle.cpp")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))
    return _SO


_lib = None
f32p = C.POINTER(C.c_float)
f64p = C.POINTER(C.c_double)
u8p = C.POINTER(C.c_uint8)
i8p = C.POINTER(C.c_int8)
i32p = C.POINTER(C.c_int)


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_SO)
        _lib.orc_set_threads(usable_cpus())  # (the default OpenMP team would be one thread per visible CPU, quota or not)
        _lib.orc_calculate_shift.restype = C.c_double
        _lib.orc_round_bf16.restype = C.c_float
        _lib.orc_round_bf16.argtypes = [C.c_float]
        _lib.orc_round_f16.restype = C.c_float
        _lib.orc_round_f16.argtypes = [C.c_float]
        _lib.orc_flux_create.restype = C.c_void_p
        _lib.orc_vae_create.restype = C.c_void_p
        _lib.orc_t5_create.restype = C.c_void_p
        _lib.orc_t5_create.argtypes = [C.c_int] * 8 + [C.c_float, C.c_int]
        _lib.orc_clip_create.restype = C.c_void_p
        _lib.orc_e4m3_to_f32.restype = C.c_float
        _lib.orc_e4m3_to_f32.argtypes = [C.c_uint8]
        _lib.orc_f32_to_e4m3.restype = C.c_uint8
        _lib.orc_f32_to_e4m3.argtypes = [C.c_float]
        _lib.orc_flux_set_fp8.argtypes = [C.c_void_p, C.c_int]
        _lib.orc_flux_set_fp8.restype = None
        _lib.orc_flux_set_fp8_attention.argtypes = [C.c_void_p, C.c_int]
        _lib.orc_flux_set_fp8_attention.restype = None
        _lib.orc_flux_set_q8_mask.argtypes = [C.c_void_p, C.c_int]
        _lib.orc_flux_set_q8_mask.restype = None
        _lib.orc_quantize_rows_i8.argtypes = [f32p, C.c_int, C.c_int, C.c_void_p, f32p]
        _lib.orc_quantize_rows_i8.restype = None
        _lib.orc_linear_i8.argtypes = [f32p, f32p, f32p, C.c_int, C.c_int, C.c_int, f32p]
        _lib.orc_linear_i8.restype = None
        _lib.orc_quantize_rows_i8_asym.argtypes = [f32p, C.c_int, C.c_int, C.c_int, C.c_void_p, f32p, f32p]
        _lib.orc_quantize_rows_i8_asym.restype = None
        _lib.orc_linear_i8_asym.argtypes = [f32p, f32p, f32p, C.c_int, C.c_int, C.c_int, C.c_int, f32p]
        _lib.orc_linear_i8_asym.restype = None
        _lib.orc_flux_set_q8_symmetric.argtypes = [C.c_void_p, C.c_int]
        _lib.orc_flux_set_q8_symmetric.restype = None
        _lib.orc_flux_set_calibration.argtypes = [C.c_void_p, C.c_int]
        _lib.orc_flux_set_calibration.restype = None
    return _lib


def _f(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a, a.ctypes.data_as(f32p)


def _opt(a):
    if a is None:
        return None, None
    return _f(a)


def usable_cpus():
    """CPUs this process may actually use: the scheduler affinity capped by the cgroup-v2 CPU quota (the GPU box shows 256 logical
    CPUs and caps the container at 16: an OpenMP team of 256 threads on 16 CPUs' worth of quota runs ~4x slower than one of 16)."""
    import os
    n = len(os.sched_getaffinity(0))
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(int(q) / int(per))))
    except Exception:
        pass
    return n


def set_threads(n):
    lib().orc_set_threads(int(n))


def get_threads():
    return lib().orc_get_threads()


def set_isa(isa=0):
    """GEMM paths (same bits whichever runs): 0 = widest micro-kernel the host has (AVX-512 where available) and the direct few-row
    path; bit 0 = the AVX2 micro-kernel; bit 1 = launches of <= 4 rows through the packed path too."""
    lib().orc_set_isa(int(isa))


def get_isa():
    return lib().orc_get_isa()


def linear(x, w, b=None):
    x, xp = _f(x)
    w, wp = _f(w)
    b, bp = _opt(b)
    M, K = int(np.prod(x.shape[:-1])), x.shape[-1]
    N = w.shape[0]
    assert w.shape[1] == K
    y = np.empty(x.shape[:-1] + (N,), np.float32)
    lib().orc_linear(xp, wp, bp, M, N, K, y.ctypes.data_as(f32p))
    return y


def layer_norm(x, alpha=None, beta=None, eps=1e-5):
    x, xp = _f(x)
    a, ap = _opt(alpha)
    b, bp = _opt(beta)
    out = np.empty_like(x)
    lib().orc_layer_norm(xp, ap, bp, C.c_float(eps), int(np.prod(x.shape[:-1])), x.shape[-1], out.ctypes.data_as(f32p))
    return out


def rms_norm_slow(x, alpha=None, eps=1e-5):
    x, xp = _f(x)
    a, ap = _opt(alpha)
    out = np.empty_like(x)
    lib().orc_rms_norm_slow(xp, ap, C.c_float(eps), int(np.prod(x.shape[:-1])), x.shape[-1], out.ctypes.data_as(f32p))
    return out


def softmax_last_dim(x):
    x, xp = _f(x)
    out = np.empty_like(x)
    lib().orc_softmax_last_dim(xp, int(np.prod(x.shape[:-1])), x.shape[-1], out.ctypes.data_as(f32p))
    return out


def group_norm(x, w, b, groups, eps=1e-5):
    x, xp = _f(x)
    w_, wp = _opt(w)
    b_, bp = _opt(b)
    B, Cc = x.shape[0], x.shape[1]
    HW = int(np.prod(x.shape[2:]))
    out = np.empty_like(x)
    lib().orc_group_norm(xp, wp, bp, B, Cc, HW, groups, C.c_float(eps), out.ctypes.data_as(f32p))
    return out


def conv2d(x, w, bias=None, pad=0, stride=1, dilation=1):
    x, xp = _f(x)
    w, wp = _f(w)
    b, bp = _opt(bias)
    B, Cin, H, W = x.shape
    Cout, Cin2, kh, kw = w.shape
    assert Cin == Cin2
    Ho = (H + 2 * pad - dilation * (kh - 1) - 1) // stride + 1
    Wo = (W + 2 * pad - dilation * (kw - 1) - 1) // stride + 1
    out = np.empty((B, Cout, Ho, Wo), np.float32)
    lib().orc_conv2d(xp, wp, bp, B, Cin, H, W, Cout, kh, kw, pad, stride, dilation, out.ctypes.data_as(f32p))
    return out


def upsample_nearest2d(x, dh, dw):
    x, xp = _f(x)
    B, Cc, H, W = x.shape
    out = np.empty((B, Cc, dh, dw), np.float32)
    lib().orc_upsample_nearest2d(xp, B, Cc, H, W, dh, dw, out.ctypes.data_as(f32p))
    return out


def gelu(x):
    x, xp = _f(x)
    out = np.empty_like(x)
    lib().orc_gelu(xp, C.c_int64(x.size), out.ctypes.data_as(f32p))
    return out


def silu(x):
    x, xp = _f(x)
    out = np.empty_like(x)
    lib().orc_silu(xp, C.c_int64(x.size), out.ctypes.data_as(f32p))
    return out


def sdpa(q, k, v, scale):
    q, qp = _f(q)
    k, kp = _f(k)
    v, vp = _f(v)
    B, H, Lq, d = q.shape
    Lk = k.shape[2]
    out = np.empty_like(q)
    lib().orc_sdpa(qp, kp, vp, B, H, Lq, Lk, d, C.c_float(scale), out.ctypes.data_as(f32p))
    return out


def rope_table(ids, axes_dim, theta):
    ids, ip = _f(ids)
    n = int(np.prod(ids.shape[:-1]))
    na = ids.shape[-1]
    ax = (C.c_int * na)(*axes_dim)
    half = sum(axes_dim) // 2
    pe = np.empty(ids.shape[:-1] + (half, 2, 2), np.float32)
    lib().orc_rope_table(ip, n, na, ax, theta, pe.ctypes.data_as(f32p))
    return pe


def apply_rope(x, pe):
    """x (H,L,d), pe (L,d/2,2,2)."""
    x, xp = _f(x)
    pe, pp = _f(pe)
    H, L, d = x.shape
    out = np.empty_like(x)
    lib().orc_apply_rope(xp, pp, H, L, d, out.ctypes.data_as(f32p))
    return out


def timestep_embedding(t, dim=256):
    t, tp = _f(t)
    out = np.empty((t.shape[0], dim), np.float32)
    lib().orc_timestep_embedding(tp, t.shape[0], dim, out.ctypes.data_as(f32p))
    return out


QUANT = {"int8": 0, "fp4": 1, "nf4": 2}
ODT = {"f32": 0, "f16": 1, "bf16": 2}


def dequantize_blockwise(code, A, absmax, blocksize, n, quant_type, out_dtype="f32"):
    A = np.ascontiguousarray(A, np.uint8)
    absmax, ap = _f(absmax)
    code_, cp = _opt(code)
    out = np.empty(n, np.float32)
    lib().orc_dequantize_blockwise(cp, A.ctypes.data_as(u8p), ap, out.ctypes.data_as(f32p), blocksize, n, QUANT[quant_type], ODT[out_dtype])
    return out


def dequantize_8bit(w, scb, row, col, out_dtype="f32"):
    w = np.ascontiguousarray(w, np.int8)
    scb, sp = _f(scb)
    n = row * col
    out = np.empty(n, np.float32)
    lib().orc_dequantize_8bit(w.ctypes.data_as(i8p), sp, out.ctypes.data_as(f32p), row, col, n, ODT[out_dtype])
    return out


def e4m3_to_f32(code):
    return float(lib().orc_e4m3_to_f32(int(code)))


def f32_to_e4m3(x):
    return int(lib().orc_f32_to_e4m3(float(x)))


def e4m3_table():
    """The 256 e4m3 values as f32 (NaN at 0x7f / 0xff)."""
    return np.array([e4m3_to_f32(c) for c in range(256)], np.float32)


def quantize_rows_fp8(x):
    """fp8 recipe of configs[4] (no reference counterpart): per-row e4m3 codes + f32 scale."""
    x, xp = _f(x)
    rows, K = x.shape
    out = np.empty((rows, K), np.uint8)
    scale = np.empty(rows, np.float32)
    lib().orc_quantize_rows_fp8(xp, rows, K, out.ctypes.data_as(u8p), scale.ctypes.data_as(f32p))
    return out, scale


def linear_fp8(x, w, b=None):
    """y = (q(x) q(w)^T) * sx * sw + b with both operands on the row-wise e4m3 recipe, f32 accumulate."""
    xq, xs = quantize_rows_fp8(x)
    wq, ws = quantize_rows_fp8(w)
    tab = e4m3_table()
    y = linear(tab[xq], tab[wq]) * (xs[:, None] * ws[None, :])
    return y + (0 if b is None else np.asarray(b, np.float32)[None, :])


def quantize_rows_i8(x):
    """int8 recipe (no reference counterpart; flux_oracle.cpp: orc_quantize_rows_i8): per-row symmetric codes + f32 scale."""
    x, xp = _f(x)
    rows, K = x.shape
    out = np.empty((rows, K), np.int8)
    scale = np.empty(rows, np.float32)
    lib().orc_quantize_rows_i8(xp, rows, K, out.ctypes.data_as(C.c_void_p), scale.ctypes.data_as(f32p))
    return out, scale


def quantize_rows_i8_asym(x, d0=0):
    """int8 recipe, post-GELU form (flux_oracle.cpp: orc_quantize_rows_i8_asym): columns [0, d0) symmetric, [d0, K) on 256 levels over their
    [min, max], one step per row.  Returns (codes int8, scale f32, offset f32)."""
    x, xp = _f(x)
    rows, K = x.shape
    out = np.empty((rows, K), np.int8)
    scale = np.empty(rows, np.float32)
    off = np.empty(rows, np.float32)
    lib().orc_quantize_rows_i8_asym(xp, rows, K, int(d0), out.ctypes.data_as(C.c_void_p), scale.ctypes.data_as(f32p), off.ctypes.data_as(f32p))
    return out, scale, off


def linear_i8_asym(x, w, b=None, d0=0):
    """y = float(q(x) q(w)^T) * (sx * sw) + offset * wsum + b with x on the post-GELU form and w symmetric per row (orc_linear_i8_asym)."""
    x, xp = _f(x)
    w, wp = _f(w)
    M, K = x.shape
    N = w.shape[0]
    y = np.empty((M, N), np.float32)
    b_, bp = _opt(b)
    lib().orc_linear_i8_asym(xp, wp, bp, M, N, K, int(d0), y.ctypes.data_as(f32p))
    return y


def linear_i8(x, w, b=None):
    """y = float(q(x) q(w)^T as an exact integer) * (sx * sw) + b, both operands on the row-wise int8 recipe (orc_linear_i8)."""
    x, xp = _f(x)
    w, wp = _f(w)
    M, K = x.shape
    N = w.shape[0]
    y = np.empty((M, N), np.float32)
    bp = None
    if b is not None:
        b, bp = _f(b)
    lib().orc_linear_i8(xp, wp, bp, M, N, K, y.ctypes.data_as(f32p))
    return y


def quantize_blockwise_4bit(w, blocksize, quant_type):
    w, wp = _f(w)
    n = w.size
    packed = np.zeros((n + 1) // 2, np.uint8)
    absmax = np.zeros((n + blocksize - 1) // blocksize, np.float32)
    lib().orc_quantize_blockwise_4bit(wp, C.c_int64(n), blocksize, QUANT[quant_type], packed.ctypes.data_as(u8p), absmax.ctypes.data_as(f32p))
    return packed, absmax


def round_bf16(a):
    a = np.ascontiguousarray(a, np.float32)
    u = a.view(np.uint32).astype(np.uint64)
    nan = (u & 0x7FFFFFFF) > 0x7F800000
    r = ((u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000).astype(np.uint32)
    r[nan] = (u[nan].astype(np.uint32) | 0x00400000) & 0xFFFF0000
    return r.view(np.float32).reshape(a.shape)


def calculate_shift(image_seq_len, base_seq_len=256, max_seq_len=4096, base_shift=0.5, max_shift=1.15):
    return lib().orc_calculate_shift(image_seq_len, base_seq_len, max_seq_len, C.c_double(base_shift), C.c_double(max_shift))


def get_timesteps(num_steps, use_dynamic_shifting, mu=0.0, shift=1.0):
    out = np.empty(num_steps + 1, np.float64)
    lib().orc_get_timesteps(num_steps, int(use_dynamic_shifting), C.c_double(mu), C.c_double(shift), out.ctypes.data_as(f64p))
    return out


def pack_latents(latent):
    latent, lp = _f(latent)
    B, Cc, h, w = latent.shape
    img = np.empty((B, (h // 2) * (w // 2), Cc * 4), np.float32)
    ids = np.empty((B, (h // 2) * (w // 2), 3), np.float32)
    lib().orc_pack_latents(lp, B, Cc, h, w, img.ctypes.data_as(f32p), ids.ctypes.data_as(f32p))
    return img, ids


def unpack_latents(img, Cc, h, w):
    img, ip = _f(img)
    B = img.shape[0]
    out = np.empty((B, Cc, h, w), np.float32)
    lib().orc_unpack_latents(ip, B, Cc, h, w, out.ctypes.data_as(f32p))
    return out


def postprocess_u8(x):
    x, xp = _f(x)
    out = np.empty(x.shape, np.uint8)
    lib().orc_postprocess_u8(xp, C.c_int64(x.size), out.ctypes.data_as(u8p))
    return out


# ---- seedable get_noise (row a4).  The reference's `get_noise` (pipelines/flux/sampling.rs:5-14) is
# `Tensor::randn` on an unseedable backend RNG, so the product replaces it by a counter-based
# generator whose stream is a pure function of (seed, sample, element): Philox4x32-10 (Salmon, Moraes, Dror,
# Shaw, "Parallel random numbers: as easy as 1, 2, 3", SC'11; Random123 v1.14 `philox.h`: multipliers
# 0xD2511F53 / 0xCD9E8D57, Weyl key increments 0x9E3779B9 / 0xBB67AE85, 10 rounds) followed by Box-Muller.
# Pinned by Random123's own known-answer vectors (tests/golden/philox_kat.json, tests/test_oracle_philox.py).
_PHILOX_M0, _PHILOX_M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
_PHILOX_W0, _PHILOX_W1 = 0x9E3779B9, 0xBB67AE85


def philox4x32_10(ctr, key):
    """ctr (..., 4) u32, key (..., 2) u32 -> (..., 4) u32."""
    c = np.array(ctr, dtype=np.uint32).reshape(-1, 4).astype(np.uint64)
    k = np.broadcast_to(np.array(key, dtype=np.uint32).reshape(-1, 2), (c.shape[0], 2)).astype(np.uint64)
    c0, c1, c2, c3 = c[:, 0], c[:, 1], c[:, 2], c[:, 3]
    k0, k1 = k[:, 0].copy(), k[:, 1].copy()
    mask = np.uint64(0xFFFFFFFF)
    for _ in range(10):
        p0 = _PHILOX_M0 * c0  # 32 x 32 -> 64 bit products
        p1 = _PHILOX_M1 * c2
        hi0, lo0 = p0 >> np.uint64(32), p0 & mask
        hi1, lo1 = p1 >> np.uint64(32), p1 & mask
        c0, c1, c2, c3 = hi1 ^ c1 ^ k0, lo1, hi0 ^ c3 ^ k1, lo0
        k0 = (k0 + np.uint64(_PHILOX_W0)) & mask
        k1 = (k1 + np.uint64(_PHILOX_W1)) & mask
    out = np.stack([c0, c1, c2, c3], axis=-1).astype(np.uint32)
    return out.reshape(np.array(ctr).shape)


def philox_u32(n_per_sample, B, seed, first_sample=0):
    """The raw stream behind `randn`: element e of sample b is word e % 4 of
    philox(ctr = (q lo, q hi, sample lo, sample hi), key = (seed lo, seed hi)), q = e // 4, sample = first_sample + b."""
    quads = (n_per_sample + 3) // 4
    q = np.arange(quads, dtype=np.uint64)
    out = np.empty((B, quads * 4), np.uint32)
    for b in range(B):
        smp = np.uint64(first_sample + b)
        ctr = np.stack([q & np.uint64(0xFFFFFFFF), q >> np.uint64(32), np.full(quads, smp & np.uint64(0xFFFFFFFF)),
                        np.full(quads, smp >> np.uint64(32))], axis=-1).astype(np.uint32)
        out[b] = philox4x32_10(ctr, [seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF]).reshape(-1)
    return out[:, :n_per_sample]


def philox_u32_c(n_per_sample, B, seed, first_sample=0):
    """The same stream from the C restatement (flux_oracle.cpp: orc_philox_u32; OpenMP): what the full-size fixtures draw their 12e9 words from.
    tests/test_oracle_philox.py holds it equal to the numpy statement above."""
    out = np.empty((B, n_per_sample), np.uint32)
    lib().orc_philox_u32(out.ctypes.data_as(C.c_void_p), C.c_int64(n_per_sample), C.c_int(B), C.c_uint64(seed), C.c_uint64(first_sample))
    return out


def exact_bf16(n, seed, offset, coeff, out=None):
    """bf16 bits of an "exact synthetic tensor" (diffusion-rs_amd/synth.py: exact_values_np is the definition) in one C pass; `out` = a reusable
    uint16 buffer of >= n elements."""
    if out is None:
        out = np.empty(n, np.uint16)
    lib().orc_exact_bf16(out.ctypes.data_as(C.c_void_p), C.c_int64(n), C.c_uint64(seed), C.c_float(float(offset)), C.c_float(float(coeff)))
    return out[:n]


def randn(n_per_sample, B, seed, first_sample=0):
    """N(0,1) f32 (B, n_per_sample): per counter, words (0,1) and (2,3) each give a Box-Muller pair.
    u = f32(f32(w >> 8) + 0.5f) * 2^-24 with the device's f32 roundings (for w >> 8 >= 2^23 the + 0.5 rounds to even:
    restated here in numpy float32, so u is bit-identical to the device's); z = sqrt(-2 ln u1) * (cos, sin)(2 pi u2) is
    then evaluated in f64 and rounded once — the device's f32 libm results are expected within a few ulp of it (bound
    stated in tests/test_gpu_philox.py)."""
    quads = (n_per_sample + 3) // 4
    w = philox_u32(quads * 4, B, seed, first_sample).reshape(B, quads, 4)
    u32 = ((w >> np.uint32(8)).astype(np.float32) + np.float32(0.5)) * np.float32(1.0 / 16777216.0)
    u = u32.astype(np.float64)
    out = np.empty((B, quads, 4), np.float64)
    for p in range(2):
        rad = np.sqrt(-2.0 * np.log(u[..., 2 * p]))
        ang = (np.float32(6.283185307179586) * u32[..., 2 * p + 1]).astype(np.float64)  # the device's f32 product
        out[..., 2 * p] = rad * np.cos(ang)
        out[..., 2 * p + 1] = rad * np.sin(ang)
    return out.reshape(B, quads * 4)[:, :n_per_sample].astype(np.float32)


class Flux:
    """CPU oracle of models::flux::Flux (model.rs:709-838), f32."""

    def __init__(self, cfg):
        self.cfg = dict(cfg)
        ax = (C.c_int * 3)(*cfg["axes_dim"])
        self.h = C.c_void_p(lib().orc_flux_create(cfg["in_channels"], cfg["pooled_projection_dim"], cfg["joint_attention_dim"],
                                                  cfg["num_attention_heads"], cfg["num_layers"], cfg["num_single_layers"],
                                                  int(cfg["guidance_embeds"]), ax, cfg["theta"]))
        self.D = cfg["num_attention_heads"] * sum(cfg["axes_dim"])

    def __del__(self):
        try:
            lib().orc_flux_destroy(self.h)
        except Exception:
            pass

    def set_tensor(self, name, arr):
        a, ap = _f(arr)
        lib().orc_flux_set_tensor(self.h, name.encode(), ap, C.c_int64(a.size))

    def load(self, tensors):
        for k, v in tensors.items():
            self.set_tensor(k, v)

    def set_tensor_bf16(self, name, bits):
        """The tensor as bf16 bits (uint16): held at 2 bytes per weight and widened per block (full-size checks)."""
        a = np.ascontiguousarray(bits, np.uint16)
        lib().orc_flux_set_tensor_bf16(self.h, name.encode(), a.ctypes.data_as(C.c_void_p), C.c_int64(a.size))

    def set_fp8(self, on=True, attention=False, study_mode=None):
        """Block linears on the fp8 recipe (flux_oracle.cpp: parity unpinned, no reference counterpart); attention=True
        also puts q and k of the attention on e4m3 with the static scales (what the library does when both streams of a
        block take the fused QKV epilogue: token counts / offsets multiples of 16).  study_mode 2..9 = the alternative
        quantisers of tools/fp8_noise_study.py (flux_oracle.cpp: lin_blk), never what the library is compared with; in a study the
        attention operands are quantised in every block whatever the linear mask (attention=2: P and V too)."""
        lib().orc_flux_set_fp8(self.h, int(study_mode) if (on and study_mode) else int(bool(on)))
        att = 0
        if on and attention:
            att = 1 | (2 if attention == 2 else 0) | (4 if study_mode else 0)
        lib().orc_flux_set_fp8_attention(self.h, att)

    def set_int8(self, on=True, mask=0x33, attention=False, attention_everywhere=False):
        """Block linears of `mask` on the int8 recipe (lin_blk mode 5: exact integer sums; parity unpinned, no reference counterpart),
        the others in f32.  0x33 = the library's default mask (all but the double blocks' MLP).  attention=True: q and k of the attention
        on e4m3 with the static scales in the blocks whose q|k|v linear is in the mask (bit 0 for the double blocks, bit 4 for the single
        ones) — what the library does in either 8-bit mode for 16-aligned token counts; attention_everywhere=True: in every block whatever
        the mask (the library's fmi_flux_set_fp8_attention(m, 2))."""
        lib().orc_flux_set_fp8(self.h, 5 if on else 0)
        lib().orc_flux_set_fp8_attention(self.h, (1 | (4 if attention_everywhere else 0)) if (on and attention) else 0)
        lib().orc_flux_set_q8_mask(self.h, int(mask) if on else 0x3f)

    def set_calibration(self, mode):
        """The smoothed int8 recipe's calibration (flux_oracle.cpp: orc_flux_set_calibration; the library's fmi_flux_calibrate_int8): 1 = record the
        per-channel absmax of every block linear's input during the following forwards (which run in f32), 0 = stop recording and keep the
        statistics (set_int8 then uses them: s = sqrt(amax_x / amax_W) per input channel), -1 = drop them (the unsmoothed recipe)."""
        lib().orc_flux_set_calibration(self.h, int(mode))

    def set_q8_mask(self, mask=0x3f):
        """Which block linears take the 8-bit recipe while set_fp8 is on (the others stay f32): bit 0 double q|k|v, 1 double attention
        out, 2 double MLP in, 3 double MLP out, 4 single linear1 (q, k, v, proj_mlp), 5 single linear2."""
        lib().orc_flux_set_q8_mask(self.h, int(mask))

    def forward(self, img, img_ids, txt, txt_ids, timesteps, y, guidance=None):
        img, a = _f(img)
        img_ids, b = _f(img_ids)
        txt, c = _f(txt)
        txt_ids, d = _f(txt_ids)
        timesteps, e = _f(timesteps)
        y, f = _f(y)
        g_, g = _opt(guidance)
        B, S, _ = img.shape
        T = txt.shape[1]
        pred = np.empty_like(img)
        rc = lib().orc_flux_forward(self.h, a, b, c, d, e, f, g, B, S, T, pred.ctypes.data_as(f32p))
        if rc:
            raise RuntimeError("oracle flux_forward failed (missing tensor?)")
        return pred

    def denoise(self, img, img_ids, txt, txt_ids, y, guidance, timesteps):
        img = np.array(img, dtype=np.float32, order="C", copy=True)
        img_ids, b = _f(img_ids)
        txt, c = _f(txt)
        txt_ids, d = _f(txt_ids)
        y, f = _f(y)
        g_, g = _opt(guidance)
        ts = np.ascontiguousarray(timesteps, np.float64)
        B, S, _ = img.shape
        T = txt.shape[1]
        rc = lib().orc_flux_denoise(self.h, img.ctypes.data_as(f32p), b, c, d, f, g, B, S, T, ts.ctypes.data_as(f64p), len(ts) - 1)
        if rc:
            raise RuntimeError("oracle flux_denoise failed")
        return img

    def double_block(self, idx, img, txt, vec, pe):
        img = np.array(img, dtype=np.float32, order="C", copy=True)
        txt = np.array(txt, dtype=np.float32, order="C", copy=True)
        vec, v = _f(vec)
        pe, p = _f(pe)
        B, S, _ = img.shape
        T = txt.shape[1]
        rc = lib().orc_flux_double_block(self.h, idx, img.ctypes.data_as(f32p), txt.ctypes.data_as(f32p), v, p, B, S, T)
        if rc:
            raise RuntimeError("oracle double_block failed")
        return img, txt

    def single_block(self, idx, x, vec, pe):
        x = np.array(x, dtype=np.float32, order="C", copy=True)
        vec, v = _f(vec)
        pe, p = _f(pe)
        B, L, _ = x.shape
        rc = lib().orc_flux_single_block(self.h, idx, x.ctypes.data_as(f32p), v, p, B, L)
        if rc:
            raise RuntimeError("oracle single_block failed")
        return x


class Vae:
    """CPU oracle of vaes::AutoEncoderKl::decode (autoencoder_kl.rs:112-119), f32."""

    def __init__(self, cfg):
        self.cfg = dict(cfg)
        boc = cfg["block_out_channels"]
        arr = (C.c_int * len(boc))(*boc)
        self.h = C.c_void_p(lib().orc_vae_create(arr, len(boc), cfg["layers_per_block"], cfg["latent_channels"], cfg["out_channels"],
                                                 cfg["norm_num_groups"], int(cfg["mid_block_add_attention"]), int(cfg.get("use_post_quant_conv", False))))

    def __del__(self):
        try:
            lib().orc_vae_destroy(self.h)
        except Exception:
            pass

    def load(self, tensors):
        for k, v in tensors.items():
            a, ap = _f(v)
            lib().orc_vae_set_tensor(self.h, k.encode(), ap, C.c_int64(a.size))

    def encode(self, img, noise=None, return_moments=False):
        """AutoEncoderKl::encode (autoencoder_kl.rs:103-110); noise = the DiagonalGaussian's randn (None -> mean)."""
        img, ip = _f(img)
        B, Cin, H, W = img.shape
        f = 2 ** (len(self.cfg["block_out_channels"]) - 1)
        h, w = H // f, W // f  # each Downsample: (n + 1 - 3) // 2 + 1 = floor(n / 2)
        lat = self.cfg["latent_channels"]
        mom = np.empty((B, 2 * lat, h, w), np.float32)
        z = np.empty((B, lat, h, w), np.float32)
        n_, npnt = _opt(noise)
        rc = lib().orc_vae_encode(self.h, ip, B, Cin, H, W, int(self.cfg.get("use_quant_conv", False)), npnt, mom.ctypes.data_as(f32p), z.ctypes.data_as(f32p))
        if rc:
            raise RuntimeError("oracle vae_encode failed rc=%d" % rc)
        return (z, mom) if return_moments else z

    def decode(self, z):
        z, zp = _f(z)
        B, _, h, w = z.shape
        # every level except i_level == 3 upsamples (hard-coded in the reference, vae.rs:412)
        f = 2 ** sum(1 for lvl in range(len(self.cfg["block_out_channels"])) if lvl != 3)
        out = np.empty((B, self.cfg["out_channels"], h * f, w * f), np.float32)
        rc = lib().orc_vae_decode(self.h, zp, B, h, w, out.ctypes.data_as(f32p))
        if rc:
            raise RuntimeError("oracle vae_decode failed rc=%d" % rc)
        return out

    def mid_attention(self, x):
        """AttnBlock::forward (vae.rs:95-111) of the decoder's mid block alone: x (B,C,H,W) f32 -> same shape."""
        x = np.array(x, dtype=np.float32, order="C", copy=True)
        B, _, H, W = x.shape
        if lib().orc_vae_mid_attention(self.h, x.ctypes.data_as(f32p), B, H, W):
            raise RuntimeError("oracle vae_mid_attention failed")
        return x


T5_ACT = {"relu": 0, "gated-gelu": 1, "gated-silu": 2}


class T5:
    """CPU oracle of t5::T5EncoderModel (t5/mod.rs:609-632), f32.  cfg keys = T5Config (t5/mod.rs:72-91)."""

    def __init__(self, cfg):
        self.cfg = dict(cfg)
        self.h = C.c_void_p(lib().orc_t5_create(cfg["vocab_size"], cfg["d_model"], cfg["d_kv"], cfg["d_ff"], cfg["num_layers"], cfg["num_heads"],
                                                cfg["relative_attention_num_buckets"], cfg.get("relative_attention_max_distance", 128),
                                                C.c_float(cfg["layer_norm_epsilon"]), T5_ACT[cfg.get("feed_forward_proj", "relu")]))

    def __del__(self):
        try:
            lib().orc_t5_destroy(self.h)
        except Exception:
            pass

    def load(self, tensors):
        for k, v in tensors.items():
            a, ap = _f(v)
            lib().orc_t5_set_tensor(self.h, k.encode(), ap, C.c_int64(a.size))

    def forward(self, ids):
        ids = np.ascontiguousarray(ids, dtype=np.int32)
        B, T = ids.shape
        out = np.empty((B, T, self.cfg["d_model"]), np.float32)
        rc = lib().orc_t5_forward(self.h, ids.ctypes.data_as(C.POINTER(C.c_int32)), B, T, out.ctypes.data_as(f32p))
        if rc:
            raise RuntimeError("oracle t5_forward failed rc=%d" % rc)
        return out


def t5_bucket(i, j, num_buckets=32, max_distance=128):
    return lib().orc_t5_bucket(int(i), int(j), int(num_buckets), int(max_distance))


class Clip:
    """CPU oracle of clip::text::ClipTextTransformer (clip/text.rs:243-317), f32.  cfg keys = ClipTextConfig."""

    def __init__(self, cfg):
        self.cfg = dict(cfg)
        self.h = C.c_void_p(lib().orc_clip_create(cfg["vocab_size"], cfg["projection_dim"], cfg["intermediate_size"], cfg["max_position_embeddings"],
                                                  cfg["num_hidden_layers"], cfg["num_attention_heads"]))

    def __del__(self):
        try:
            lib().orc_clip_destroy(self.h)
        except Exception:
            pass

    def load(self, tensors):
        for k, v in tensors.items():
            a, ap = _f(v)
            lib().orc_clip_set_tensor(self.h, k.encode(), ap, C.c_int64(a.size))

    def forward(self, ids, return_hidden=False):
        ids = np.ascontiguousarray(ids, dtype=np.int32)
        B, T = ids.shape
        D = self.cfg["projection_dim"]
        hid = np.empty((B, T, D), np.float32)
        pooled = np.empty((B, D), np.float32)
        rc = lib().orc_clip_forward(self.h, ids.ctypes.data_as(C.POINTER(C.c_int32)), B, T, hid.ctypes.data_as(f32p), pooled.ctypes.data_as(f32p))
        if rc:
            raise RuntimeError("oracle clip_forward failed rc=%d" % rc)
        return (pooled, hid) if return_hidden else pooled
