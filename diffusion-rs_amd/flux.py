"""Host-side mirrors of the reference's model objects over the C-ABI.

  FluxModel      <-> diffusion_rs_core::models::flux::Flux            (model.rs:709-838)
  AutoEncoderKl  <-> diffusion_rs_core::models::vaes::AutoEncoderKl   (autoencoder_kl.rs:52-128)
  FlowMatchEuler <-> pipelines::sampling::Sampler + SchedulerConfig   (sampling.rs, scheduler.rs)

torch is used only as the device-memory / stream plumbing (tensors own the buffers whose raw
pointers cross the C-ABI); every FLOP runs in libflux_mi355x.so.
"""
import ctypes as C
from dataclasses import dataclass, field
from typing import List, Optional

import numpy as np
import torch

from . import _lib as L

# fmi_flux_quantize_int8's linear mask (include/flux_mi355x.h: FMI_Q8_*)
Q8_DOUBLE_QKV, Q8_DOUBLE_OUT, Q8_DOUBLE_MLP_IN, Q8_DOUBLE_MLP_OUT, Q8_SINGLE_LINEAR1, Q8_SINGLE_LINEAR2 = 1, 2, 4, 8, 16, 32
INT8_DEFAULT_MASK = Q8_DOUBLE_QKV | Q8_DOUBLE_OUT | Q8_SINGLE_LINEAR1 | Q8_SINGLE_LINEAR2  # FMI_INT8_DEFAULT_MASK

FLUX_DEV = dict(in_channels=64, pooled_projection_dim=768, joint_attention_dim=4096, num_attention_heads=24, num_layers=19,
                num_single_layers=38, guidance_embeds=True, axes_dim=[16, 56, 56], theta=10000)
FLUX_SCHNELL = dict(FLUX_DEV, guidance_embeds=False)
VAE_FLUX = dict(in_channels=3, out_channels=3, block_out_channels=[128, 256, 512, 512], layers_per_block=2, latent_channels=16,
                norm_num_groups=32, mid_block_add_attention=True, use_post_quant_conv=False, scaling_factor=0.3611, shift_factor=0.1159)

_TORCH_DT = {torch.float32: L.F32, torch.float16: L.F16, torch.bfloat16: L.BF16, torch.uint8: L.U8, torch.int8: L.I8}
_NP_DT = {np.dtype(np.float32): L.F32, np.dtype(np.float16): L.F16}


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def _tensor_arg(t):
    """(pointer, fmi_dtype, shape, keepalive) for a torch tensor (any device) or numpy array."""
    if isinstance(t, np.ndarray):
        a = np.ascontiguousarray(t)
        if a.dtype not in _NP_DT:
            a = a.astype(np.float32)
        return C.c_void_p(a.ctypes.data), _NP_DT[a.dtype], a.shape, a
    t = t.contiguous()
    return C.c_void_p(t.data_ptr()), _TORCH_DT[t.dtype], tuple(t.shape), t


@dataclass
class SchedulerConfig:
    """pipelines/scheduler.rs:4-20 (public FLUX.1 values as defaults)."""
    base_image_seq_len: int = 256
    base_shift: float = 0.5
    max_image_seq_len: int = 4096
    max_shift: float = 1.15
    shift: float = 3.0
    use_dynamic_shifting: bool = True

    def get_timesteps(self, num_steps: int, mu: Optional[float]) -> List[float]:
        if self.use_dynamic_shifting and mu is None:
            raise ValueError("`mu` is required for dynamic shifting")  # scheduler.rs:34
        c = L.SchedulerConfigC(self.base_image_seq_len, self.base_shift, self.max_image_seq_len, self.max_shift, self.shift,
                               int(self.use_dynamic_shifting))
        out = (C.c_double * (num_steps + 1))()
        L.check(L.load().fmi_get_timesteps(C.byref(c), num_steps, C.c_double(mu or 0.0), out))
        return list(out)

    def calculate_shift(self, image_seq_len: int) -> float:
        return L.load().fmi_calculate_shift(image_seq_len, self.base_image_seq_len, self.max_image_seq_len, self.base_shift, self.max_shift)


class FluxModel:
    def __init__(self, cfg: dict, device: int = 0):
        self.lib = L.load()
        self.cfg = dict(cfg)
        L.check(self.lib.fmi_init(device), self.lib)
        c = L.FluxConfig(cfg["in_channels"], cfg["pooled_projection_dim"], cfg["joint_attention_dim"], cfg["num_attention_heads"],
                         cfg["num_layers"], cfg["num_single_layers"], int(cfg["guidance_embeds"]), (C.c_int * 3)(*cfg["axes_dim"]), cfg["theta"])
        h = C.c_void_p()
        L.check(self.lib.fmi_flux_create(C.byref(c), L.MODEL_BF16, C.byref(h)), self.lib)
        self.h = h
        self.hidden = cfg["num_attention_heads"] * 128
        self.device = torch.device("cuda", device)

    def close(self):
        if getattr(self, "h", None):
            self.lib.fmi_flux_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def is_guidance(self) -> bool:  # Flux::is_guidance, model.rs:835-837
        return bool(self.cfg["guidance_embeds"])

    def set_tensor(self, name: str, t):
        p, dt, shape, keep = _tensor_arg(t)
        sh = (C.c_int64 * len(shape))(*shape)
        L.check(self.lib.fmi_flux_set_tensor(self.h, name.encode(), p, dt, sh, len(shape)), self.lib)

    def load_state_dict(self, tensors: dict):
        for k, v in tensors.items():
            self.set_tensor(k, v)
        self.assert_complete()

    def set_linear_bnb4(self, prefix: str, packed, absmax, blocksize: int, quant_type: str, out_features: int, in_features: int):
        q = {"fp4": 1, "nf4": 2}[quant_type]
        if isinstance(packed, torch.Tensor):  # host or device tensors: the C side copies with hipMemcpyDefault
            pk, am = packed.contiguous(), absmax.to(torch.float32).contiguous()
            assert pk.dtype == torch.uint8
            pp, ap = C.c_void_p(pk.data_ptr()), C.c_void_p(am.data_ptr())
        else:
            pk = np.ascontiguousarray(packed, np.uint8)
            am = np.ascontiguousarray(absmax, np.float32)
            pp, ap = C.c_void_p(pk.ctypes.data), C.c_void_p(am.ctypes.data)
        L.check(self.lib.fmi_flux_set_linear_bnb4(self.h, prefix.encode(), pp, ap, blocksize, q, out_features, in_features), self.lib)

    def set_linear_int8(self, prefix: str, weight, scb, out_features: int, in_features: int):
        """LLM.int8 linear (BnbLinear::Int8): weight int8 (out,in), SCB f32 (out)."""
        if isinstance(weight, torch.Tensor):
            w, sc = weight.contiguous(), scb.to(torch.float32).contiguous()
            assert w.dtype == torch.int8
            wp, sp = C.c_void_p(w.data_ptr()), C.c_void_p(sc.data_ptr())
        else:
            w = np.ascontiguousarray(weight, np.int8)
            sc = np.ascontiguousarray(scb, np.float32)
            wp, sp = C.c_void_p(w.ctypes.data), C.c_void_p(sc.ctypes.data)
        L.check(self.lib.fmi_flux_set_linear_int8(self.h, prefix.encode(), wp, sp, out_features, in_features), self.lib)

    def quantize_fp8(self, stream=None):
        """Switch the DiT block linears to the fp8 (OCP e4m3) MFMA path: weights quantised once per output
        channel from the loaded bf16 values, activations per token on the fly (BASELINE configs[4])."""
        L.check(self.lib.fmi_flux_quantize_fp8(self.h, stream), self.lib)

    def quantize_int8(self, mask: int = None, stream=None):
        """Switch the DiT block linears named by `mask` (Q8_* bits; default INT8_DEFAULT_MASK) to the int8 MFMA path: symmetric per-row
        int8 codes (weights once per output channel, activations per token on the fly), exact int32 accumulation; the other block
        linears and everything else stay on the bf16 path — except the attention operands: as in the fp8 mode (set_fp8_attention, default
        on) the blocks whose q|k|v linear is in the mask hand q and k to the attention as e4m3 with static scales (QK^T on the fp8
        MFMA; P.V stays bf16).  set_fp8_attention(0) keeps bf16 operands."""
        L.check(self.lib.fmi_flux_quantize_int8(self.h, INT8_DEFAULT_MASK if mask is None else int(mask), stream), self.lib)

    def calibrate_int8(self, on: bool = True):
        """Start (or drop) the calibration of the smoothed int8 recipe: while on, every forward / denoise evaluation (bf16 mode) also records the per-channel
        absmax of each block linear's input; the next quantize_int8() folds s = sqrt(amax_x / amax_W) per channel into the weight codes and 1 / s into the
        activation quantisation (include/flux_mi355x.h: fmi_flux_calibrate_int8).  Without it quantize_int8() is the unsmoothed recipe."""
        L.check(self.lib.fmi_flux_calibrate_int8(self.h, int(bool(on))), self.lib)

    def set_fp8_attention(self, mode: int):
        """q and k of the attention as e4m3 with static scales, QK^T on the fp8 MFMA: 0 never, 1 (default) in the 8-bit modes, 2 in every
        mode — the bf16 block linears included (opt-in: a reduced-precision attention operand, not the reference's semantics)."""
        L.check(self.lib.fmi_flux_set_fp8_attention(self.h, int(mode)), self.lib)

    def missing(self) -> List[str]:
        n = self.lib.fmi_flux_missing_count(self.h)
        return [self.lib.fmi_flux_missing_name(self.h, i).decode() for i in range(n)]

    def assert_complete(self):
        m = self.missing()
        if m:
            raise L.FmiError(f"{len(m)} tensors missing, e.g. {m[:4]}")

    def size_in_bytes(self) -> int:
        return self.lib.fmi_flux_size_in_bytes(self.h)

    def set_quant_dense_cache(self, mode):
        """Quantised linears, launches above 383 (nf4 / fp4) / 256 (LLM.int8) rows — smaller ones always multiply from the packed codes:
        -1 (default) = by memory: 3 when the device has the room, else 0; 0 / False = packed only: per-call expansion into a 264 MB
        scratch + dense GEMM; 1 / True = every matrix expanded once into the bf16 arenas; 2 = always the fused kernels; 3 = the matrices
        of the large launches expanded once.  Same bits in every mode."""
        L.check(self.lib.fmi_flux_set_quant_dense_cache(self.h, int(mode)), self.lib)

    def set_split_k(self, on: bool):
        """Latency mode for launches of few rows (sequence-parallel shards): residual projections with fewer than 128 tiles are
        split along K and reduced in a fixed order, and the sequence-parallel attention of few heads walks key ranges in parallel
        (log-sum-exp merge) — deterministic, equal to the unsplit result to rounding (not bit for bit)."""
        L.check(self.lib.fmi_flux_set_split_k(self.h, int(bool(on))), self.lib)

    def set_sequence_parallel(self, rank: int, world_size: int, all_to_all=None):
        """Single-image sequence parallelism (fmi_flux_set_sequence_parallel): from now on forward / denoise take THIS rank's
        token shard (dist.sp_shard) and the joint attention trades heads for tokens through `all_to_all`
        (callable(send_ptr, recv_ptr, bytes_per_peer, stream_ptr) -> None; dist.SequenceParallel provides it on
        torch.distributed).  world_size 1 switches it off."""
        if world_size > 1 and all_to_all is None:
            raise L.FmiError("set_sequence_parallel: world_size > 1 needs an all_to_all callable")

        def _cb(_user, send, recv, nbytes, stream):
            try:
                all_to_all(send, recv, nbytes, stream)
                return 0
            except Exception as e:  # an exception must not unwind through the C frames
                import traceback
                traceback.print_exc()
                self._sp_error = e
                return 1

        self._sp_cb = L.ALL_TO_ALL_FN(_cb) if world_size > 1 else L.ALL_TO_ALL_FN()  # keep the thunk alive with the model
        L.check(self.lib.fmi_flux_set_sequence_parallel(self.h, int(rank), int(world_size), self._sp_cb, None), self.lib)

    def set_sequence_parallel_native(self, rank: int, world_size: int, comm):
        """The same with the exchange done by the library's own RCCL communicator (dist.RcclComm / fmi_comm): the callback is the C
        function fmi_comm_all_to_all itself, so an exchange is one ncclAllToAll enqueued from C on the launch stream — no Python
        between two kernels of a block."""
        fn = C.cast(self.lib.fmi_comm_all_to_all, L.ALL_TO_ALL_FN)
        self._sp_cb, self._sp_comm = fn, comm  # keep both alive with the model
        L.check(self.lib.fmi_flux_set_sequence_parallel(self.h, int(rank), int(world_size), fn, comm.h), self.lib)

    # ---- the weights as flat device buffers (multi-GPU broadcast, dist.broadcast_state)
    def state_export(self) -> bytes:
        n = C.c_size_t()
        L.check(self.lib.fmi_flux_state_export(self.h, None, 0, C.byref(n)), self.lib)
        buf = (C.c_uint8 * n.value)()
        L.check(self.lib.fmi_flux_state_export(self.h, buf, n.value, C.byref(n)), self.lib)
        return bytes(buf)

    def state_adopt(self, blob: bytes):
        buf = (C.c_uint8 * len(blob)).from_buffer_copy(blob)
        L.check(self.lib.fmi_flux_state_adopt(self.h, buf, len(blob)), self.lib)

    def state_buffers(self):
        """[(device pointer, bytes)] of the weight arenas (pointer 0 / 0 bytes for an arena not in use)."""
        out = []
        for i in range(self.lib.fmi_flux_state_buffer_count()):
            p, n = C.c_void_p(), C.c_size_t()
            L.check(self.lib.fmi_flux_state_buffer(self.h, i, C.byref(p), C.byref(n)), self.lib)
            out.append((p.value or 0, n.value))
        return out

    def state_views(self, device=None):
        """One flat uint8 torch tensor per weight arena, aliasing its device memory (None for an arena not in use): what
        dist.broadcast_state hands to the collective — the weights travel in place, without a staging copy."""
        from .dist import _DeviceBytes
        dev = torch.device(device) if device is not None else self.device
        return [torch.as_tensor(_DeviceBytes(p, n), device=dev) if n else None for p, n in self.state_buffers()]

    def _inputs(self, img, img_ids, txt, txt_ids, timesteps, y, guidance):
        B, S = int(img_ids.shape[0]), int(img_ids.shape[1])
        T = int(txt.shape[1])
        keep = [t.contiguous() if t is not None else None for t in (img, img_ids, txt, txt_ids, timesteps, y, guidance)]
        img, img_ids, txt, txt_ids, timesteps, y, guidance = keep
        for t, nm in ((img_ids, "img_ids"), (txt_ids, "txt_ids"), (timesteps, "timesteps"), (guidance, "guidance")):
            if t is not None and t.dtype != torch.float32:
                raise TypeError(f"{nm} must be float32")
        inp = L.FluxInputs(_ptr(img), _TORCH_DT[img.dtype] if img is not None else 0, _ptr(img_ids), _ptr(txt), _TORCH_DT[txt.dtype], _ptr(txt_ids),
                           _ptr(timesteps), _ptr(y), _TORCH_DT[y.dtype], _ptr(guidance), B, S, T, 0)
        return inp, keep

    def forward(self, img, img_ids, txt, txt_ids, timesteps, y, guidance=None):
        """== Flux::forward (model.rs:790-833).  Returns the velocity (B,S,C) f32."""
        inp, keep = self._inputs(img, img_ids, txt, txt_ids, timesteps, y, guidance)
        pred = torch.empty(img.shape, dtype=torch.float32, device=img.device)
        L.check(self.lib.fmi_flux_forward(self.h, C.byref(inp), _ptr(pred), _stream()), self.lib)
        return pred

    def denoise(self, img, img_ids, txt, txt_ids, y, guidance, timesteps: List[float]):
        """== Sampler::sample around Flux::forward (sampling.rs:25-48). `img` f32 (B,S,C), updated copy returned."""
        img = img.to(torch.float32).clone().contiguous()
        inp, keep = self._inputs(None, img_ids, txt, txt_ids, None, y, guidance)
        ts = (C.c_double * len(timesteps))(*timesteps)
        L.check(self.lib.fmi_flux_denoise(self.h, C.byref(inp), _ptr(img), ts, len(timesteps) - 1, _stream()), self.lib)
        return img

    def set_profiling(self, on: bool):
        L.check(self.lib.fmi_flux_set_profiling(self.h, int(on)), self.lib)

    def phase_ms(self) -> dict:
        n = self.lib.fmi_flux_phase_count()
        out = (C.c_float * n)()
        L.check(self.lib.fmi_flux_phase_ms(self.h, out), self.lib)
        return {self.lib.fmi_flux_phase_name(i).decode(): out[i] for i in range(n)}


class AutoEncoderKl:
    def __init__(self, cfg: dict, device: int = 0):
        self.lib = L.load()
        self.cfg = dict(cfg)
        L.check(self.lib.fmi_init(device), self.lib)
        boc = list(cfg["block_out_channels"])
        if len(boc) != 4:
            raise L.FmiError("block_out_channels must have 4 entries (the reference hard-codes i_level != 3, vae.rs:412)")
        c = L.VaeConfig(cfg["in_channels"], cfg["out_channels"], (C.c_int * 4)(*boc), 4, cfg["layers_per_block"], cfg["latent_channels"],
                        cfg["norm_num_groups"], int(cfg["mid_block_add_attention"]), int(cfg.get("use_post_quant_conv", False)),
                        cfg["scaling_factor"], cfg["shift_factor"], int(cfg.get("use_quant_conv", False)))
        h = C.c_void_p()
        L.check(self.lib.fmi_vae_create(C.byref(c), L.MODEL_BF16, C.byref(h)), self.lib)
        self.h = h

    def close(self):
        if getattr(self, "h", None):
            self.lib.fmi_vae_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_tensor(self, name, t):
        p, dt, shape, keep = _tensor_arg(t)
        sh = (C.c_int64 * len(shape))(*shape)
        L.check(self.lib.fmi_vae_set_tensor(self.h, name.encode(), p, dt, sh, len(shape)), self.lib)

    def missing(self):
        n = self.lib.fmi_vae_missing_count(self.h)
        return [self.lib.fmi_vae_missing_name(self.h, i).decode() for i in range(n)]

    def load_state_dict(self, tensors: dict):
        """Decoder tensors are required; encoder (`encoder.*`, `quant_conv.*`) tensors are optional
        (only `encode` needs them, and it reports what is missing)."""
        for k, v in tensors.items():
            self.set_tensor(k, v)
        m = [n for n in self.missing() if not (n.startswith("encoder.") or n.startswith("quant_conv."))]
        if m:
            raise L.FmiError(f"{len(m)} VAE tensors missing, e.g. {m[0]}")

    def encode(self, image, noise=None, seed=None, return_moments=False):
        """== VAEModel::encode (vaes/mod.rs:15-28, autoencoder_kl.rs:103-110): image (B,3,H,W) f32 ->
        z (B,16,H/8,W/8) f32 = mean + exp(0.5 logvar) * noise.  noise: explicit tensor, or Philox
        N(0,1) from `seed`, or None and seed None -> the distribution mean."""
        image = image.to(torch.float32).contiguous()
        B, _, H, W = image.shape
        lat = self.cfg["latent_channels"]
        if noise is None and seed is not None:
            noise = randn_latents(B, lat, H // 8, W // 8, seed, 0, image.device)
        if noise is not None:
            noise = noise.to(device=image.device, dtype=torch.float32).contiguous()
        z = torch.empty((B, lat, H // 8, W // 8), dtype=torch.float32, device=image.device)
        mom = torch.empty((B, 2 * lat, H // 8, W // 8), dtype=torch.float32, device=image.device) if return_moments else None
        L.check(self.lib.fmi_vae_encode(self.h, _ptr(image), B, H, W, _ptr(noise) if noise is not None else None, _ptr(z),
                                        _ptr(mom) if mom is not None else None, _stream()))
        return (z, mom) if return_moments else z

    def scale_factor(self) -> float:
        return self.lib.fmi_vae_scale_factor(self.h)

    def shift_factor(self) -> float:
        return self.lib.fmi_vae_shift_factor(self.h)

    def mid_attention(self, x_nhwc):
        """AttnBlock::forward (vae.rs:95-111) of the decoder's mid block alone: (B,H,W,C) bf16 NHWC -> same shape."""
        x = x_nhwc.contiguous()
        assert x.dtype == torch.bfloat16 and x.dim() == 4
        out = torch.empty_like(x)
        L.check(self.lib.fmi_vae_mid_attention(self.h, _ptr(x), x.shape[0], x.shape[1], x.shape[2], _ptr(out), _stream()), self.lib)
        return out

    def decode(self, z):
        """== VAEModel::decode (vaes/mod.rs:15-28): z (B,16,h,w) f32 -> (B,3,8h,8w) f32."""
        z = z.to(torch.float32).contiguous()
        B, _, h, w = z.shape
        out = torch.empty((B, self.cfg["out_channels"], 8 * h, 8 * w), dtype=torch.float32, device=z.device)
        L.check(self.lib.fmi_vae_decode(self.h, _ptr(z), B, h, w, _ptr(out), _stream()), self.lib)
        return out


# ---- tensor glue of FluxPipeline::forward (pipelines/flux/mod.rs:270-332), on device
def pack_latents(latent):
    """State::new patchify (flux/sampling.rs:131-148): (B,C,h,w) f32 -> img (B,hw/4,4C), img_ids (B,hw/4,3)."""
    latent = latent.to(torch.float32).contiguous()
    B, Cc, h, w = latent.shape
    img = torch.empty((B, (h // 2) * (w // 2), Cc * 4), dtype=torch.float32, device=latent.device)
    ids = torch.empty((B, (h // 2) * (w // 2), 3), dtype=torch.float32, device=latent.device)
    L.check(L.load().fmi_pack_latents(_ptr(latent), B, Cc, h, w, _ptr(img), _ptr(ids), _stream()))
    return img, ids


def unpack_latents(img, Cc, h, w, scale_factor, shift_factor):
    img = img.to(torch.float32).contiguous()
    B = img.shape[0]
    z = torch.empty((B, Cc, h, w), dtype=torch.float32, device=img.device)
    L.check(L.load().fmi_unpack_latents(_ptr(img), B, Cc, h, w, scale_factor, shift_factor, _ptr(z), _stream()))
    return z


def postprocess_u8(image, interleave=False):
    image = image.to(torch.float32).contiguous()
    B, Cc, H, W = image.shape
    out = torch.empty((B, H, W, Cc) if interleave else (B, Cc, H, W), dtype=torch.uint8, device=image.device)
    L.check(L.load().fmi_postprocess_u8(_ptr(image), B, Cc, H, W, int(interleave), _ptr(out), _stream()))
    return out


def randn_latents(B, Cc, h, w, seed, first_sample=0, device="cuda"):
    """Seedable get_noise (flux/sampling.rs:5-14): Philox4x32-10, sample index = first_sample + b."""
    out = torch.empty((B, Cc, h, w), dtype=torch.float32, device=device)
    L.check(L.load().fmi_randn(_ptr(out), Cc * h * w, B, seed, first_sample, _stream()))
    return out
