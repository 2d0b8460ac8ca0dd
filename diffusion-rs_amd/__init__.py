"""diffusion-rs_amd — MI355X-native FLUX.1 denoise path (drop-in for the hot path of
EricLBuehler/diffusion-rs).  Import as `diffusion_rs_amd` (see ../diffusion_rs_amd.py)."""
from ._lib import FmiError, LIB_PATH, load  # noqa: F401
from .flux import (AutoEncoderKl, FLUX_DEV, FLUX_SCHNELL, FluxModel, SchedulerConfig, VAE_FLUX, pack_latents, postprocess_u8,  # noqa: F401
                   randn_latents, unpack_latents)
from .pipeline import DiffusionGenerationParams, ModelDType, ModelSource, Offloading, Pipeline, encode_png  # noqa: F401
from .text import CLIP_L, T5_XXL, ClipTextTransformer, T5EncoderModel, load_bpe_tokenizer, tokenize_and_pad  # noqa: F401
from . import synth  # noqa: F401
