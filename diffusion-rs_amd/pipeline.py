"""Python mirror of the reference's front-door API (diffusion_rs_py/src/lib.rs:13-156 and
diffusion_rs_core/src/pipelines/mod.rs:24-33,110-270) over the MI355X hot path.

    ModelSource.ModelId(model_id) / ModelSource.DdufFile(file)
    DiffusionGenerationParams(height, width, num_steps, guidance_scale)
    ModelDType.{Auto,BF16,F16,F32}    Offloading.Full
    Pipeline(source, silent=False, token=None, revision=None, offloading=None, dtype=ModelDType.Auto)
    Pipeline.forward(prompts, params) -> list[bytes]   (PNG-encoded, as the pyo3 binding returns)

Scope: text encoders (when the checkpoint ships them), the denoise loop and the VAE
decode run on the GPU through the C-ABI.  Tokenisation needs the checkpoint's tokenizer files and
the `tokenizers` package; without them pass `token_ids=(t5_ids, clip_ids)` or precomputed
`embeddings=(t5_emb, clip_emb)` — a loaded checkpoint never falls back to made-up embeddings.  Only the
Synthetic source built without text encoders (the benchmark) derives deterministic placeholder
embeddings from the prompt text: only shapes matter there.  Extensions over the reference, all keyword-only:
`latents=` / `seed=` (the reference cannot be seeded), `embeddings=`, `token_ids=`, `output=`.

Multi-GPU: when torch.distributed is initialised (one process per GPU), the constructor loads the
DiT on rank 0 only and broadcasts its weight arenas over RCCL (dist.broadcast_state), and `forward` shards the
prompts — prompt i runs on rank i % world — and gathers the images to rank 0 (other ranks return None).
"""
import enum
import hashlib
import json
import os
import struct
import threading
import zlib
from dataclasses import dataclass
from typing import List, Optional, Sequence

import numpy as np
import torch

from . import flux as F
from . import synth


class ModelDType(enum.Enum):  # diffusion_rs_py/src/lib.rs:37-44
    Auto = 0
    BF16 = 1
    F16 = 2
    F32 = 3
    F8E4M3 = 4  # extension (not in the reference): DiT block linears on the e4m3 MFMA, DESIGN.md §4.3
    I8 = 5      # extension: the int8 MFMA on the linears of flux.INT8_DEFAULT_MASK (all but the double blocks' MLP), DESIGN.md §4.3c


class Offloading(enum.Enum):  # lib.rs:13-17.  Accepted and ignored: 288 GB of HBM hold everything.
    Full = 0


class ModelSource:
    """diffusion_rs_common/src/model_source.rs:18-85 (+ a Synthetic source for offline benchmarks)."""

    def __init__(self, kind, **kw):
        self.kind = kind
        self.__dict__.update(kw)

    @staticmethod
    def ModelId(model_id: str) -> "ModelSource":
        return ModelSource("model_id", model_id=model_id)

    from_model_id = ModelId

    def override_transformer_model_id(self, model_id: str) -> "ModelSource":  # model_source.rs:59-69
        if self.kind != "model_id":
            raise ValueError("Expected model ID for the model source")
        return ModelSource("model_id", model_id=self.model_id, transformer_model_id=model_id)

    @staticmethod
    def DdufFile(file: str) -> "ModelSource":
        return ModelSource("dduf", file=file)

    dduf = DdufFile

    @staticmethod
    def Synthetic(variant: str = "dev", seed: int = 0, flux_cfg: Optional[dict] = None, vae_cfg: Optional[dict] = None,
                  text_encoders: bool = False, t5_cfg: Optional[dict] = None, clip_cfg: Optional[dict] = None) -> "ModelSource":
        """Random-init weights of the named architecture generated on the GPU (no checkpoints offline).
        text_encoders=True also builds T5 / CLIP (T5-XXL: 9 GiB of bf16 weights)."""
        return ModelSource("synthetic", variant=variant, seed=seed, flux_cfg=flux_cfg, vae_cfg=vae_cfg, text_encoders=text_encoders,
                           t5_cfg=t5_cfg, clip_cfg=clip_cfg)

    def __repr__(self):
        if self.kind == "model_id":
            return f"model id: {self.model_id}"
        if self.kind == "dduf":
            return f"dduf file: {self.file}"
        return f"synthetic FLUX.1-{self.variant} (seed {self.seed})"


@dataclass
class DiffusionGenerationParams:  # pipelines/mod.rs:24-33
    height: int
    width: int
    num_steps: int
    guidance_scale: float

    def __repr__(self):
        return (f"DiffusionGenerationParams(height = {self.height}, width = {self.width}, num_steps = {self.num_steps}, "
                f"guidance_scale = {self.guidance_scale})")


def encode_png(rgb: np.ndarray) -> bytes:
    """(H,W,3) u8 -> PNG bytes (what `image.write_to(.., ImageFormat::Png)` produces in lib.rs:144-152)."""
    h, w, c = rgb.shape
    assert c == 3 and rgb.dtype == np.uint8
    raw = np.concatenate([np.zeros((h, 1), np.uint8), rgb.reshape(h, w * 3)], axis=1).tobytes()

    def chunk(tag, data):
        return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xFFFFFFFF)

    return b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 2, 0, 0, 0)) + chunk(b"IDAT", zlib.compress(raw, 6)) + chunk(b"IEND", b"")


def placeholder_embeddings(prompts: Sequence[str], T: int, joint_dim: int, pooled_dim: int, device):
    """Deterministic stand-in for T5/CLIP outputs, for the Synthetic source without text encoders only."""
    t5, clip = [], []
    for p in prompts:
        seed = int.from_bytes(hashlib.sha256(p.encode()).digest()[:8], "little")
        g = torch.Generator(device="cpu")
        g.manual_seed(seed & 0x7FFFFFFFFFFFFFFF)
        t5.append(torch.randn((T, joint_dim), generator=g))
        clip.append(torch.randn((pooled_dim,), generator=g))
    return torch.stack(t5).to(device=device, dtype=torch.bfloat16), torch.stack(clip).to(device=device, dtype=torch.float32)


class Pipeline:
    """== diffusion_rs_core::Pipeline (pipelines/mod.rs:110-270) for FluxPipeline."""

    def __init__(self, source: ModelSource, silent: bool = False, token: Optional[str] = None, revision: Optional[str] = None,
                 offloading: Optional[Offloading] = None, dtype: ModelDType = ModelDType.Auto, *, device: int = 0):
        if dtype in (ModelDType.F16, ModelDType.F32):
            raise ValueError("this build computes in bf16 on MFMA (f32 accumulate); use ModelDType.Auto or BF16")
        self.silent = silent
        self.offloading = offloading
        self.device_index = device
        self.device = torch.device("cuda", device)
        self._lock = threading.Lock()  # Arc<Mutex<dyn ModelPipeline>>, pipelines/mod.rs:110-113
        self.scheduler = F.SchedulerConfig()
        self.t5 = self.clip = self.t5_tokenizer = self.clip_tokenizer = None
        self.source_kind = source.kind
        from . import dist as D
        rank, world = D.world()
        # multi-GPU: the DiT (98 % of the bytes) is materialised on rank 0 and broadcast as flat arenas; the VAE and
        # the text encoders are loaded / generated by every rank itself (same files, same seeds)
        self.load_stats = {}
        self._dit_loader = None  # loads the DiT into self.flux on this rank (used by rank 0, and by every rank when the flat state cannot travel)
        err = None
        try:
            if source.kind == "synthetic":
                fcfg = source.flux_cfg or (F.FLUX_DEV if source.variant == "dev" else F.FLUX_SCHNELL)
                vcfg = source.vae_cfg or F.VAE_FLUX
                self.flux = F.FluxModel(fcfg, device)
                self._dit_loader = lambda: synth.fill_flux_random_device(self.flux, seed=source.seed, device=self.device)
                if rank == 0:
                    self._dit_loader()
                self.vae = F.AutoEncoderKl(vcfg, device)
                synth.fill_vae_random_device(self.vae, seed=source.seed + 1, device=self.device)
                if source.variant != "dev":
                    self.scheduler = F.SchedulerConfig(shift=1.0, use_dynamic_shifting=False)
                if getattr(source, "text_encoders", False):
                    from . import text
                    self.t5 = text.T5EncoderModel(source.t5_cfg, device)
                    synth.fill_text_random_device(self.t5, seed=source.seed + 2, device=self.device)
                    self.clip = text.ClipTextTransformer(source.clip_cfg, device)
                    synth.fill_text_random_device(self.clip, seed=source.seed + 3, device=self.device)
            elif source.kind == "model_id":
                self._load_checkpoint(source.model_id, getattr(source, "transformer_model_id", None), load_dit=(rank == 0))
            else:
                self._load_checkpoint(source.file, None, load_dit=(rank == 0))
        except Exception as e:  # a rank that fails here must not leave the others blocked in the broadcast below
            if world == 1:
                raise
            err = e
        if world > 1:
            D.agree_or_raise(err, "Pipeline load")
            try:
                self.load_stats["broadcast"] = D.broadcast_state(self.flux, self.device)
            except D.StateExportUnsupported as e:  # raised on EVERY rank (e.g. LLM.int8 matrices are not part of the flat state)
                err = None
                try:
                    if rank != 0:
                        self._dit_loader()
                except Exception as e2:
                    err = e2
                D.agree_or_raise(err, "per-rank DiT load")
                self.load_stats["broadcast"] = {"bytes": 0, "messages": 0, "seconds": 0.0, "fallback": f"every rank loaded the DiT itself ({e})"}
        if dtype == ModelDType.F8E4M3:
            self.flux.quantize_fp8()
        elif dtype == ModelDType.I8:
            # the int8 mode is CALIBRATED (round 6: per-channel smoothing, include/flux_mi355x.h fmi_flux_calibrate_int8): the statistics come from the first
            # request — four model evaluations across its schedule on its first prompt — so the weights are quantised there, not here (generate_tensor)
            self._int8_pending = True

    # Pipeline::load (pipelines/mod.rs:120-236) for a local diffusers directory or a DDUF file:
    # model_index.json -> FluxPipeline only; scheduler / transformer / vae components, plus text_encoder (CLIP),
    # text_encoder_2 (T5) and their tokenizers when the checkpoint ships them (flux/mod.rs:74-127).
    def _load_checkpoint(self, path: str, transformer_path: Optional[str], load_dit: bool = True):
        from . import loader
        fl = loader.FileLoader(path)
        if fl.read_json("model_index.json").get("_class_name") != "FluxPipeline":  # pipelines/mod.rs:146-149
            raise ValueError("Only FluxPipeline is supported")
        sc = fl.read_json("scheduler/scheduler_config.json")
        self.scheduler = F.SchedulerConfig(sc["base_image_seq_len"], sc["base_shift"], sc["max_image_seq_len"], sc["max_shift"], sc["shift"],
                                           sc["use_dynamic_shifting"])
        tl = loader.FileLoader(transformer_path) if transformer_path else fl  # ModelIdWithTransformer (model_source.rs:22-25)
        tc = tl.read_json("transformer/config.json")
        fcfg = dict(F.FLUX_DEV, **{k: tc[k] for k in ("in_channels", "pooled_projection_dim", "joint_attention_dim", "num_attention_heads", "num_layers",
                                                     "num_single_layers", "guidance_embeds") if k in tc})
        self.flux = F.FluxModel(fcfg, self.device_index)
        self._dit_loader = lambda: self.load_stats.update(loader.load_flux(self.flux, tl.tensors("transformer")))
        if load_dit:  # (multi-GPU: the other ranks receive the weight arenas, dist.broadcast_state)
            self._dit_loader()
        vc = fl.read_json("vae/config.json")
        vcfg = dict(F.VAE_FLUX, **{k: vc[k] for k in F.VAE_FLUX if k in vc})
        self.vae = F.AutoEncoderKl(vcfg, self.device_index)
        loader.load_vae(self.vae, fl.tensors("vae"))
        # text_encoder (CLIP) / text_encoder_2 (T5) + their tokenizers (flux/mod.rs:74-127), when the checkpoint ships them
        if fl.has("text_encoder/config.json") and fl.has("text_encoder_2/config.json"):
            from . import text
            cc = fl.read_json("text_encoder/config.json")
            # ClipTextConfig reads projection_dim as the hidden width (clip/text.rs:24-33); both are 768 for CLIP-L
            self.clip = text.ClipTextTransformer({k: cc[k] for k in text.CLIP_L}, self.device_index)
            loader.load_text_encoder(self.clip, fl.tensors("text_encoder"))
            tc2 = fl.read_json("text_encoder_2/config.json")
            t5cfg = {k: tc2[k] for k in text.T5_XXL if k in tc2}
            t5cfg["quantization_config"] = tc2.get("quantization_config")
            self.t5 = text.T5EncoderModel(t5cfg, self.device_index)
            loader.load_text_encoder(self.t5, fl.tensors("text_encoder_2"))
            try:
                from tokenizers import Tokenizer
                if fl.has("tokenizer/vocab.json") and fl.has("tokenizer/merges.txt"):
                    self.clip_tokenizer = text.load_bpe_tokenizer(fl.read_text("tokenizer/vocab.json"), fl.read_text("tokenizer/merges.txt"))
                if fl.has("tokenizer_2/tokenizer.json"):
                    self.t5_tokenizer = Tokenizer.from_str(fl.read_text("tokenizer_2/tokenizer.json"))
            except ImportError:
                pass  # no `tokenizers` package: forward() then needs token_ids=

    @classmethod
    def load(cls, source, silent=False, token=None, revision=None, offloading_type=None, dtype=ModelDType.Auto, **kw):
        """Rust-style constructor name (Pipeline::load, pipelines/mod.rs:120-127)."""
        return cls(source, silent, token, revision, offloading_type, dtype, **kw)

    # ------------------------------------------------------------------------------------------
    def encode_prompts(self, prompts: List[str], token_ids=None):
        """Prompt -> (t5_emb (B,T,4096) bf16, clip_emb (B,768) f32): the first half of FluxPipeline::forward
        (flux/mod.rs:237-262).  token_ids = (t5_ids, clip_ids) overrides tokenisation (no tokenizer files offline)."""
        from . import text
        if self.t5 is None or self.clip is None:
            raise F.L.FmiError("this pipeline was built without text encoders")
        if token_ids is not None:
            t5_ids, clip_ids = token_ids
        elif self.t5_tokenizer is not None and self.clip_tokenizer is not None:
            t5_ids = text.tokenize_and_pad(prompts, self.t5_tokenizer)
            clip_ids = text.tokenize_and_pad(prompts, self.clip_tokenizer)
        else:
            raise F.L.FmiError("no tokenizers loaded: pass token_ids=(t5_ids, clip_ids)")
        t5_ids = torch.as_tensor(t5_ids, dtype=torch.int32)
        if not self.flux.is_guidance():  # schnell: zero-pad the T5 ids to 256 (flux/mod.rs:243-255)
            if t5_ids.shape[1] > 256:
                raise ValueError("T5 embedding length greater than 256, please shrink the prompt or use the -dev (with guidance distillation) version.")
            t5_ids = torch.nn.functional.pad(t5_ids, (0, 256 - t5_ids.shape[1]))
        return self.t5.forward(t5_ids), self.clip.forward(torch.as_tensor(clip_ids, dtype=torch.int32))

    MAX_BATCH = 8  # samples per denoise call (the C-ABI's per-device batch limit); longer prompt lists run in chunks

    def generate_tensor(self, prompts: List[str], params: DiffusionGenerationParams, *, embeddings=None, latents=None,
                        seed: Optional[int] = None, first_sample: int = 0, token_ids=None, sample_ids: Optional[Sequence[int]] = None) -> torch.Tensor:
        """== ModelPipeline::forward for FluxPipeline (pipelines/flux/mod.rs:224-335) on THIS device.
        Returns (B,3,H,W) u8 on the device.  `sample_ids` (default first_sample + 0..B-1) name the Philox streams of
        the samples, so that a sample draws the same noise whichever rank / chunk it runs in."""
        B = len(prompts)
        ids = list(sample_ids) if sample_ids is not None else [first_sample + b for b in range(B)]
        if len(ids) != B:
            raise ValueError("sample_ids must name one stream per prompt")
        if B == 0:
            return torch.empty((0, 3, params.height, params.width), dtype=torch.uint8, device=self.device)
        sp = getattr(self, "_sp", None)
        if sp is not None and B > 1:  # sequence parallel: the ranks of the group work on ONE image at a time
            return torch.cat([self.generate_tensor(
                prompts[b:b + 1], params, embeddings=None if embeddings is None else (embeddings[0][b:b + 1], embeddings[1][b:b + 1]),
                latents=None if latents is None else latents[b:b + 1], seed=seed,
                token_ids=None if token_ids is None else (token_ids[0][b:b + 1], token_ids[1][b:b + 1]), sample_ids=ids[b:b + 1]) for b in range(B)], 0)
        if B > self.MAX_BATCH:  # the reference accepts any batch (pipelines/mod.rs:241-270)
            outs = []
            for a in range(0, B, self.MAX_BATCH):
                sl = slice(a, a + self.MAX_BATCH)
                outs.append(self.generate_tensor(
                    prompts[sl], params, embeddings=None if embeddings is None else (embeddings[0][sl], embeddings[1][sl]),
                    latents=None if latents is None else latents[sl], seed=seed,
                    token_ids=None if token_ids is None else (token_ids[0][sl], token_ids[1][sl]), sample_ids=ids[sl]))
            return torch.cat(outs, 0)
        cfg = self.flux.cfg
        dev = self.device
        with self._lock:  # the whole forward, text encoders included, like the reference's mutex (pipelines/mod.rs:247)
            if embeddings is not None:
                t5_emb, clip_emb = embeddings
                t5_emb, clip_emb = t5_emb.to(dev), clip_emb.to(dev)
            elif self.t5 is not None and self.clip is not None:
                t5_emb, clip_emb = self.encode_prompts(prompts, token_ids)
            elif self.source_kind == "synthetic":
                # schnell pads T5 ids to 256 (flux/mod.rs:243-253); dev uses the prompt length — 512 here
                T = 256 if not self.flux.is_guidance() else 512
                t5_emb, clip_emb = placeholder_embeddings(prompts, T, cfg["joint_attention_dim"], cfg["pooled_projection_dim"], dev)
            else:
                raise F.L.FmiError("this checkpoint was loaded without text encoders: pass embeddings=(t5_emb, clip_emb)")
            h = (params.height + 15) // 16 * 2  # get_noise, flux/sampling.rs:12-13
            w = (params.width + 15) // 16 * 2
            if latents is None:
                sd = seed if seed is not None else 299792458
                if ids == list(range(ids[0], ids[0] + B)):
                    latents = F.randn_latents(B, 16, h, w, sd, ids[0], dev)
                else:
                    latents = torch.cat([F.randn_latents(1, 16, h, w, sd, i, dev) for i in ids], 0)
            latents = latents.to(device=dev, dtype=torch.float32)
            img, img_ids = F.pack_latents(latents)  # State::new
            txt_ids = torch.zeros((B, t5_emb.shape[1], 3), dtype=torch.float32, device=dev)
            mu = self.scheduler.calculate_shift(img.shape[1])
            timesteps = self.scheduler.get_timesteps(params.num_steps, mu)
            guidance = torch.full((B,), float(params.guidance_scale), dtype=torch.float32, device=dev) if self.flux.is_guidance() else None
            if getattr(self, "_int8_pending", False):
                self._int8_calibrate_and_quantize(img[:1], img_ids[:1], t5_emb[:1], txt_ids[:1], clip_emb[:1], None if guidance is None else guidance[:1], timesteps)
            if sp is not None:  # every rank holds the same inputs; each denoises its token shard, then all get the latents
                img = sp.gather(self.flux.denoise(sp.shard(img), sp.shard(img_ids), sp.shard(t5_emb), sp.shard(txt_ids), clip_emb, guidance, timesteps))
            else:
                img = self.flux.denoise(img, img_ids, t5_emb, txt_ids, clip_emb, guidance, timesteps)
            z = F.unpack_latents(img, 16, h, w, self.vae.scale_factor(), self.vae.shift_factor())
            image = self.vae.decode(z)
            return F.postprocess_u8(image)

    INT8_CALIBRATION_POINTS = 4

    def _int8_calibrate_and_quantize(self, img, img_ids, t5_emb, txt_ids, clip_emb, guidance, timesteps):
        """ModelDType.I8, first request: INT8_CALIBRATION_POINTS evaluations of the bf16 model on ONE sample at timesteps spread over the request's schedule
        record the per-channel absmax of every block linear's input (the outlier channels of a DiT are the same at every step and for every prompt: they
        come from the AdaLN weights), then the block linears are quantised with the smoothing factors folded in.  ~0.25 s once per model at 1024 x 1024."""
        n = len(timesteps) - 1
        pts = sorted({min(n - 1, max(0, round(i * (n - 1) / max(1, self.INT8_CALIBRATION_POINTS - 1)))) for i in range(self.INT8_CALIBRATION_POINTS)})
        self.flux.calibrate_int8(True)
        for i in pts:
            t = torch.full((1,), float(timesteps[i]), dtype=torch.float32, device=self.device)
            self.flux.forward(img, img_ids, t5_emb, txt_ids, t, clip_emb, guidance)
        self.flux.quantize_int8()
        self._int8_pending = False

    def enable_sequence_parallel(self, group=None):
        """Single-image latency mode: the ranks of `group` (default: all of torch.distributed) denoise every
        image TOGETHER, each on 1/N of its tokens (dist.SequenceParallel; two all-to-alls per transformer block), instead of
        sharding the prompt list.  Every rank must call forward / generate_tensor with the same arguments."""
        from . import dist as D
        self._sp = D.SequenceParallel(self.device, group)
        self._sp.attach(self.flux)
        return self._sp

    def disable_sequence_parallel(self):
        if getattr(self, "_sp", None) is not None:
            self._sp.detach(self.flux)
            self._sp = None

    def forward(self, prompts: List[str], params: DiffusionGenerationParams, *, output: str = "png", **kw):
        """== Pipeline::forward (pipelines/mod.rs:241-270) + the PNG encode of the pyo3 binding.
        With torch.distributed initialised the batch is sharded (prompt i on rank i % world, no data-path collective) and
        the images are gathered to rank 0; every rank must make the same call, ranks != 0 return None."""
        from . import dist as D
        rank, world = D.world()
        if getattr(self, "_sp", None) is not None:  # all ranks produce every image together; rank 0 returns them
            u8 = self.generate_tensor(prompts, params, **kw)
            if rank != 0:
                return None
        elif world > 1:
            n = len(prompts)
            if getattr(self, "_int8_pending", False) and n > 0:
                # every rank calibrates on the SAME sample — global sample 0 of this request — so that all ranks hold the same int8 weights and an image does
                # not depend on the rank that produced it (one extra image per rank, once per model)
                sub = dict(kw)
                for key in ("embeddings", "token_ids"):
                    if sub.get(key) is not None:
                        sub[key] = tuple(t[:1] for t in sub[key])
                if sub.get("latents") is not None:
                    sub["latents"] = sub["latents"][:1]
                first = sub.pop("first_sample", 0)
                self.generate_tensor(prompts[:1], params, sample_ids=[first], **sub)

            def pick(x, ids):
                return None if x is None else x[torch.as_tensor(ids, dtype=torch.long)] if isinstance(x, torch.Tensor) else [x[i] for i in ids]

            def run_local(my_prompts, ids):
                sub = dict(kw)
                for key in ("embeddings", "token_ids"):
                    if sub.get(key) is not None:
                        sub[key] = tuple(pick(t, ids) for t in sub[key])
                if sub.get("latents") is not None:
                    sub["latents"] = pick(sub["latents"], ids)
                first = sub.pop("first_sample", 0)
                return self.generate_tensor(my_prompts, params, sample_ids=[first + i for i in ids], **sub)

            u8 = D.generate_sharded(prompts, run_local,
                                    empty=lambda: torch.empty((0, 3, params.height, params.width), dtype=torch.uint8, device=self.device))
            if rank != 0:
                return None
            assert u8.shape[0] == n
        else:
            u8 = self.generate_tensor(prompts, params, **kw)
        if output == "tensor":
            return u8
        hwc = u8.permute(0, 2, 3, 1).contiguous().cpu().numpy()
        if output == "rgb":
            return [hwc[i] for i in range(hwc.shape[0])]
        return [encode_png(hwc[i]) for i in range(hwc.shape[0])]
