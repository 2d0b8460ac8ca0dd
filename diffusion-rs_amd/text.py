"""Host-side mirror of the reference's text encoders over the C-ABI:

    T5EncoderModel        <-> diffusion_rs_core::models::t5::T5EncoderModel        (t5/mod.rs:609-632)
    ClipTextTransformer   <-> diffusion_rs_core::models::clip::text::ClipTextTransformer (clip/text.rs:243-317)
    tokenize_and_pad      <-> FluxPipeline::tokenize_and_pad (pipelines/flux/mod.rs:202-221)
    load_bpe_tokenizer    <-> diffusion_rs_common::load_bpe_tokenizer (tokenizer.rs:7-23)

Same constructor configs (the JSON keys of text_encoder{,_2}/config.json), same tensor names, same
forward contracts; compute is the HIP library only (no CPU fallback).
"""
import ctypes as C
import json
from typing import List, Sequence

import torch

from . import _lib as L
from .flux import _ptr, _stream, _tensor_arg

T5_XXL = dict(vocab_size=32128, d_model=4096, d_kv=64, d_ff=10240, num_layers=24, num_heads=64, relative_attention_num_buckets=32,
              relative_attention_max_distance=128, layer_norm_epsilon=1e-6, feed_forward_proj="gated-gelu")
CLIP_L = dict(vocab_size=49408, projection_dim=768, intermediate_size=3072, max_position_embeddings=77, num_hidden_layers=12, num_attention_heads=12)
_T5_ACT = {"relu": 0, "gated-gelu": 1, "gated-silu": 2}


class _Encoder:
    _kind = ""

    def close(self):
        if getattr(self, "h", None):
            getattr(self.lib, f"fmi_{self._kind}_destroy")(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_tensor(self, name, t):
        p, dt, shape, keep = _tensor_arg(t)
        sh = (C.c_int64 * len(shape))(*shape)
        L.check(getattr(self.lib, f"fmi_{self._kind}_set_tensor")(self.h, name.encode(), p, dt, sh, len(shape)))

    def missing(self) -> List[str]:
        n = getattr(self.lib, f"fmi_{self._kind}_missing_count")(self.h)
        return [getattr(self.lib, f"fmi_{self._kind}_missing_name")(self.h, i).decode() for i in range(n)]

    def load_state_dict(self, tensors: dict, strict_extra: bool = False):
        """Feeds every tensor the model knows; unknown names are ignored unless strict_extra
        (checkpoints carry e.g. the T5 decoder-tied `encoder.embed_tokens.weight`)."""
        known = self.tensor_names()
        for k, v in tensors.items():
            if k in known:
                self.set_tensor(k, v)
            elif strict_extra:
                raise L.FmiError(f"unexpected tensor {k}")
        m = self.missing()
        if m:
            raise L.FmiError(f"{len(m)} {self._kind} tensors missing, e.g. {m[0]}")

    def size_in_bytes(self) -> int:
        return getattr(self.lib, f"fmi_{self._kind}_size_in_bytes")(self.h)

    @staticmethod
    def _ids(ids, device):
        t = torch.as_tensor(ids)
        if t.dim() != 2:
            raise L.FmiError("input_ids must be (batch, seq)")
        return t.to(device=device, dtype=torch.int32).contiguous()


class T5EncoderModel(_Encoder):
    _kind = "t5"

    def __init__(self, cfg: dict = None, device: int = 0):
        self.lib = L.load()
        L.check(self.lib.fmi_init(device), self.lib)
        self.device = torch.device("cuda", device)
        cfg = dict(T5_XXL if cfg is None else cfg)
        # `quantization_config` (bitsandbytes nf4 / fp4 / LLM.int8, t5/mod.rs:85-90): the quantised Linears arrive through
        # set_linear_bnb4 / set_linear_int8 (loader.load_text_encoder); nothing to configure up front
        self.cfg = cfg
        c = L.T5Config(cfg["vocab_size"], cfg["d_model"], cfg["d_kv"], cfg["d_ff"], cfg["num_layers"], cfg["num_heads"], cfg["relative_attention_num_buckets"],
                       cfg.get("relative_attention_max_distance", 128), cfg["layer_norm_epsilon"], _T5_ACT[cfg.get("feed_forward_proj", "relu")])
        h = C.c_void_p()
        L.check(self.lib.fmi_t5_create(C.byref(c), C.byref(h)), self.lib)
        self.h = h

    def tensor_names(self):
        from . import synth
        return synth.t5_tensor_shapes(self.cfg)

    def set_linear_bnb4(self, prefix: str, packed, absmax, blocksize: int, quant_type: str, out_features: int, in_features: int):
        """A bitsandbytes 4-bit Linear of the encoder (BnbLinear::linear_b, bitsandbytes/mod.rs:137-239; used by every T5 Linear
        when text_encoder_2/config.json carries a quantization_config, t5/mod.rs:132-173,258-261): packed codes (u8, n/2) +
        f32 absmax (n/blocksize), quant_type "nf4" | "fp4"."""
        pk = packed.to(device=self.device, dtype=torch.uint8).contiguous()
        am = absmax.to(device=self.device, dtype=torch.float32).contiguous()
        L.check(self.lib.fmi_t5_set_linear_bnb4(self.h, prefix.encode(), _ptr(pk), _ptr(am), int(blocksize), {"fp4": 1, "nf4": 2}[quant_type], int(out_features),
                                                int(in_features)))

    def set_linear_int8(self, prefix: str, weight, scb, out_features: int, in_features: int):
        """An LLM.int8 Linear (BnbLinear::Int8, bitsandbytes/mod.rs:104-134): int8 weight (out, in) + f32 SCB (out)."""
        w = weight.to(device=self.device, dtype=torch.int8).contiguous()
        sc = scb.to(device=self.device, dtype=torch.float32).contiguous()
        L.check(self.lib.fmi_t5_set_linear_int8(self.h, prefix.encode(), _ptr(w), _ptr(sc), int(out_features), int(in_features)), self.lib)

    def forward(self, input_ids, dtype=torch.bfloat16):
        """== T5EncoderModel::forward: ids (B,T) -> hidden states (B,T,d_model) in the model dtype."""
        ids = self._ids(input_ids, self.device)
        B, T = ids.shape
        out = torch.empty((B, T, self.cfg["d_model"]), dtype=dtype, device=self.device)
        L.check(self.lib.fmi_t5_forward(self.h, _ptr(ids), B, T, _ptr(out), L.BF16 if dtype == torch.bfloat16 else L.F32, _stream()), self.lib)
        return out


class ClipTextTransformer(_Encoder):
    _kind = "clip"

    def __init__(self, cfg: dict = None, device: int = 0):
        self.lib = L.load()
        L.check(self.lib.fmi_init(device), self.lib)
        self.device = torch.device("cuda", device)
        cfg = dict(CLIP_L if cfg is None else cfg)
        self.cfg = cfg
        c = L.ClipConfig(cfg["vocab_size"], cfg["projection_dim"], cfg["intermediate_size"], cfg["max_position_embeddings"], cfg["num_hidden_layers"],
                         cfg["num_attention_heads"])
        h = C.c_void_p()
        L.check(self.lib.fmi_clip_create(C.byref(c), C.byref(h)), self.lib)
        self.h = h

    def tensor_names(self):
        from . import synth
        return synth.clip_tensor_shapes(self.cfg)

    def forward(self, input_ids, return_hidden: bool = False):
        """== ClipTextTransformer::forward: ids (B,T) -> pooled (B,dim) f32 = final hidden state at argmax(id)."""
        ids = self._ids(input_ids, self.device)
        B, T = ids.shape
        D = self.cfg["projection_dim"]
        pooled = torch.empty((B, D), dtype=torch.float32, device=self.device)
        hid = torch.empty((B, T, D), dtype=torch.float32, device=self.device) if return_hidden else None
        L.check(self.lib.fmi_clip_forward(self.h, _ptr(ids), B, T, _ptr(pooled), L.F32, _ptr(hid) if hid is not None else None, _stream()), self.lib)
        return (pooled, hid) if return_hidden else pooled


# ---------------------------------------------------------------------------------- tokenisation
def load_bpe_tokenizer(vocab_json: str, merges_txt: str):
    """CLIP tokenizer exactly as the reference builds it (tokenizer.rs:7-23): a bare BPE model from
    vocab + merges (first line of merges skipped, malformed lines dropped) with NO normalizer,
    pre-tokenizer or post-processor — so no <|startoftext|>/<|endoftext|> are added (SURVEY F-notes)."""
    from tokenizers import Tokenizer
    from tokenizers.models import BPE
    vocab = json.loads(vocab_json)
    merges = [tuple(x.split(" ")) for x in merges_txt.split("\n")[1:]]
    merges = [m for m in merges if len(m) == 2]
    return Tokenizer(BPE(vocab, merges))


def tokenize_and_pad(prompts: Sequence[str], tokenizer) -> List[List[int]]:
    """FluxPipeline::tokenize_and_pad (flux/mod.rs:202-221): encode_batch(add_special_tokens=True),
    zero-pad every row to the longest one."""
    rows = [e.ids for e in tokenizer.encode_batch(list(prompts), add_special_tokens=True)]
    n = max(len(r) for r in rows)
    return [r + [0] * (n - len(r)) for r in rows]
