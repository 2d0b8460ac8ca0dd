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
`embeddings=(t5_emb, clip_emb)`.  A pipeline built WITHOUT text encoders (the benchmark's synthetic
source) derives deterministic placeholder embeddings from the prompt text — only shapes matter there.  Extensions over the reference, all keyword-only:
`latents=` / `seed=` (the reference cannot be seeded), `embeddings=`, `output=`.
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
    """Deterministic stand-in for T5/CLIP outputs (text encoders are out of scope)."""
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
        if source.kind == "synthetic":
            fcfg = source.flux_cfg or (F.FLUX_DEV if source.variant == "dev" else F.FLUX_SCHNELL)
            vcfg = source.vae_cfg or F.VAE_FLUX
            self.flux = F.FluxModel(fcfg, device)
            synth.fill_flux_random_device(self.flux, seed=source.seed, device=self.device)
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
            self._load_checkpoint(source.model_id, getattr(source, "transformer_model_id", None))
        else:
            self._load_checkpoint(source.file, None)
        if dtype == ModelDType.F8E4M3:
            self.flux.quantize_fp8()

    # Pipeline::load (pipelines/mod.rs:120-236) for a local diffusers directory or a DDUF file:
    # model_index.json -> FluxPipeline only; scheduler / transformer / vae components
    # (text_encoder*, tokenizer* are listed by the reference loader but out of scope here).
    def _load_checkpoint(self, path: str, transformer_path: Optional[str]):
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
        self.load_stats = loader.load_flux(self.flux, tl.tensors("transformer"))
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

    def generate_tensor(self, prompts: List[str], params: DiffusionGenerationParams, *, embeddings=None, latents=None,
                        seed: Optional[int] = None, first_sample: int = 0, token_ids=None) -> torch.Tensor:
        """== ModelPipeline::forward for FluxPipeline (pipelines/flux/mod.rs:224-335) from the
        embeddings onward.  Returns (B,3,H,W) u8 on the device."""
        cfg = self.flux.cfg
        B = len(prompts)
        dev = self.device
        if embeddings is None and self.t5 is not None and (token_ids is not None or self.t5_tokenizer is not None):
            t5_emb, clip_emb = self.encode_prompts(prompts, token_ids)
        elif embeddings is None:
            # schnell pads T5 ids to 256 (flux/mod.rs:243-253); dev uses the prompt length — 512 here
            T = 256 if not self.flux.is_guidance() else 512
            t5_emb, clip_emb = placeholder_embeddings(prompts, T, cfg["joint_attention_dim"], cfg["pooled_projection_dim"], dev)
        else:
            t5_emb, clip_emb = embeddings
            t5_emb, clip_emb = t5_emb.to(dev), clip_emb.to(dev)
        h = (params.height + 15) // 16 * 2  # get_noise, flux/sampling.rs:12-13
        w = (params.width + 15) // 16 * 2
        if latents is None:
            latents = F.randn_latents(B, 16, h, w, seed if seed is not None else 299792458, first_sample, dev)
        latents = latents.to(device=dev, dtype=torch.float32)
        img, img_ids = F.pack_latents(latents)  # State::new
        txt_ids = torch.zeros((B, t5_emb.shape[1], 3), dtype=torch.float32, device=dev)
        mu = self.scheduler.calculate_shift(img.shape[1])
        timesteps = self.scheduler.get_timesteps(params.num_steps, mu)
        guidance = torch.full((B,), float(params.guidance_scale), dtype=torch.float32, device=dev) if self.flux.is_guidance() else None
        with self._lock:
            img = self.flux.denoise(img, img_ids, t5_emb, txt_ids, clip_emb, guidance, timesteps)
            z = F.unpack_latents(img, 16, h, w, self.vae.scale_factor(), self.vae.shift_factor())
            image = self.vae.decode(z)
            return F.postprocess_u8(image)

    def forward(self, prompts: List[str], params: DiffusionGenerationParams, *, output: str = "png", **kw):
        """== Pipeline::forward (pipelines/mod.rs:241-270) + the PNG encode of the pyo3 binding."""
        u8 = self.generate_tensor(prompts, params, **kw)
        if output == "tensor":
            return u8
        hwc = u8.permute(0, 2, 3, 1).contiguous().cpu().numpy()
        if output == "rgb":
            return [hwc[i] for i in range(hwc.shape[0])]
        return [encode_png(hwc[i]) for i in range(hwc.shape[0])]
