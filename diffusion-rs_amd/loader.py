"""Weight ingestion: diffusers directories, DDUF archives and HF-bitsandbytes
tensor naming, feeding the C-ABI's fmi_flux_set_tensor / fmi_flux_set_linear_bnb4 / fmi_vae_set_tensor.

  FileLoader      <-> diffusion_rs_common::model_source::FileLoader (model_source.rs:87-259):
                      a local directory, or a DDUF file = a zip whose entries are STORED
                      (uncompressed) so tensors can be sliced straight out of the mmap
                      (model_source.rs:225-232, varbuilder_loading.rs:111-117).
  load_flux       <-> FluxModel::new over a VarBuilder (model.rs:722-787) + the bnb detection of
                      diffusion_rs_backend::linear_b (lib.rs:197-266): a linear whose prefix has
                      `weight.absmax` / `weight.quant_state.bitsandbytes__{nf4,fp4}` is 4-bit
                      (bitsandbytes/mod.rs:137-222), nested absmax resolved as mod.rs:230-239.
There is no network here: hub model ids must already be local directories.
"""
import io
import json
import warnings
import mmap
import os
import struct
import zipfile
from typing import Dict, Iterator, List, Tuple

import numpy as np
import torch

from . import synth

_ST_DTYPES = {"F32": (torch.float32, 4), "F16": (torch.float16, 2), "BF16": (torch.bfloat16, 2), "U8": (torch.uint8, 1), "I8": (torch.int8, 1),
              "I32": (torch.int32, 4), "I64": (torch.int64, 8), "F64": (torch.float64, 8)}


def _safetensors_from_buffer(buf) -> Iterator[Tuple[str, torch.Tensor]]:
    """Zero-copy views of every tensor in a safetensors image held in `buf` (mmap / memoryview)."""
    n = struct.unpack("<Q", bytes(buf[:8]))[0]
    header = json.loads(bytes(buf[8:8 + n]))
    base = 8 + n
    for name, meta in header.items():
        if name == "__metadata__":
            continue
        dt, _ = _ST_DTYPES[meta["dtype"]]
        a, b = meta["data_offsets"]
        with warnings.catch_warnings():  # read-only mmap views are intended: tensors are only copied to the device
            warnings.simplefilter("ignore", UserWarning)
            t = torch.frombuffer(buf, dtype=dt, count=(b - a) // _ST_DTYPES[meta["dtype"]][1], offset=base + a) if b > a else torch.empty(0, dtype=dt)
        yield name, t.reshape(meta["shape"])


class FileLoader:
    """Directory or DDUF view of a diffusers checkpoint: list_files / read_json / tensors(component)."""

    def __init__(self, path: str):
        self.path = path
        self._mm = None
        if os.path.isdir(path):
            self.kind = "dir"
            self.files = sorted(os.path.relpath(os.path.join(r, f), path) for r, _, fs in os.walk(path) for f in fs)
        elif zipfile.is_zipfile(path):
            self.kind = "dduf"
            self._f = open(path, "rb")
            self._mm = mmap.mmap(self._f.fileno(), 0, access=mmap.ACCESS_READ)
            self._zip = zipfile.ZipFile(path)
            self._info = {i.filename: i for i in self._zip.infolist() if not i.is_dir()}
            for i in self._info.values():
                if i.compress_type != zipfile.ZIP_STORED:
                    raise ValueError(f"{path}: DDUF entries must be stored uncompressed ({i.filename} is compressed)")
            self.files = sorted(self._info)
        else:
            raise FileNotFoundError(f"{path}: not a directory or DDUF (zip) file; hub ids need network access, which this build does not have")

    def list_files(self) -> List[str]:
        return self.files

    def _entry_view(self, name: str) -> memoryview:
        """Slice of the mmap holding a stored zip entry (data_start()..+size, model_source.rs:225-232)."""
        i = self._info[name]
        hdr = self._mm[i.header_offset:i.header_offset + 30]
        nlen, elen = struct.unpack("<HH", hdr[26:30])
        start = i.header_offset + 30 + nlen + elen
        return memoryview(self._mm)[start:start + i.file_size]

    def read_json(self, name: str) -> dict:
        if self.kind == "dir":
            with open(os.path.join(self.path, name)) as f:
                return json.load(f)
        return json.loads(bytes(self._entry_view(name)))

    def read_text(self, name: str) -> str:
        if self.kind == "dir":
            with open(os.path.join(self.path, name), encoding="utf-8") as f:
                return f.read()
        return bytes(self._entry_view(name)).decode("utf-8")

    def has(self, name: str) -> bool:
        return name in self.files

    def tensors(self, component: str) -> Iterator[Tuple[str, torch.Tensor]]:
        """All tensors of every *.safetensors shard under `component/` (shards in name order)."""
        for fn in self.files:
            if not (fn.startswith(component + "/") and fn.endswith(".safetensors")):
                continue
            if self.kind == "dir":
                with open(os.path.join(self.path, fn), "rb") as f:
                    mm = mmap.mmap(f.fileno(), 0, access=mmap.ACCESS_READ)
                yield from _safetensors_from_buffer(memoryview(mm))
            else:
                yield from _safetensors_from_buffer(self._entry_view(fn))


def _resolve_absmax(lib, group: Dict[str, torch.Tensor], state: dict, device) -> torch.Tensor:
    """f32 absmax of a 4-bit weight; nested (double-quantised) absmax per bitsandbytes/mod.rs:230-239:
    absmax = dequantize_int8(absmax_u8, nested_quant_map, nested_absmax, nested_blocksize) + nested_offset."""
    import ctypes as C
    absmax = group["weight.absmax"]
    if "weight.nested_absmax" not in group:
        return absmax.to(torch.float32)
    a8 = absmax.to(device=device, dtype=torch.uint8).contiguous()
    code = group["weight.nested_quant_map"].to(device=device, dtype=torch.float32).contiguous()
    nabs = group["weight.nested_absmax"].to(device=device, dtype=torch.float32).contiguous()
    out = torch.empty(a8.numel(), dtype=torch.float32, device=device)
    lib.dequantize_blockwise_f32_int8(C.c_void_p(code.data_ptr()), C.c_void_p(a8.data_ptr()), C.c_void_p(nabs.data_ptr()), C.c_void_p(out.data_ptr()),
                                      int(state["nested_blocksize"]), a8.numel(), None)
    torch.cuda.synchronize()
    return out + float(state["nested_offset"])


def _feed_linears(model, tensors: Iterator[Tuple[str, torch.Tensor]], want: dict, on_extra=None) -> dict:
    """Shared by load_flux and load_text_encoder: feed `model` (set_tensor / set_linear_bnb4 / set_linear_int8, .device) from
    (name, tensor) pairs.  bitsandbytes layers arrive as groups of tensors — "<prefix>.weight" (packed u8 or int8) with
    "<prefix>.weight.absmax", ".weight.quant_map", ".weight.quant_state.bitsandbytes__nf4|fp4" (+ the nested_* tensors of double
    quantisation), or "<prefix>.SCB" for LLM.int8 — the naming BnbLinear::linear_b reads (bitsandbytes/mod.rs:111-239).
    `on_extra(name, tensor) -> bool` may claim a name the model does not list.  Returns {"dense", "bnb4", "int8", "skipped"}."""
    from . import _lib as L
    lib = L.load()
    stats = {"dense": 0, "bnb4": 0, "int8": 0, "skipped": []}
    pending: Dict[str, Dict[str, torch.Tensor]] = {}
    for name, t in tensors:
        if ".weight." in name:
            prefix, rest = name.split(".weight.", 1)
            pending.setdefault(prefix, {})["weight." + rest] = t
            continue
        if name.endswith(".SCB"):  # LLM.int8 row scales (bitsandbytes/mod.rs:113,126)
            pending.setdefault(name[:-len(".SCB")], {})["SCB"] = t
            continue
        if name.endswith(".weight_format"):
            continue
        if name.endswith(".weight") and t.dtype in (torch.uint8, torch.int8):  # packed 4-bit / int8 weight of a bnb linear
            pending.setdefault(name[:-len(".weight")], {})["weight"] = t
            continue
        if name in want:
            model.set_tensor(name, t)
            stats["dense"] += 1
        elif on_extra is not None and on_extra(name, t):
            stats["dense"] += 1
        else:
            stats["skipped"].append(name)
    for prefix, group in pending.items():
        if prefix + ".weight" not in want:
            stats["skipped"].append(prefix + ".weight")
            continue
        out_f, in_f = want[prefix + ".weight"]
        if "SCB" in group:  # BnbLinear::Int8
            if "weight" not in group or group["weight"].dtype != torch.int8:
                raise ValueError(f"`BnbLinear` int8 layer {prefix} needs an int8 `weight` next to `SCB`")
            if tuple(group["weight"].shape) != (out_f, in_f) or group["SCB"].numel() != out_f:
                raise ValueError(f"{prefix}: int8 weight {tuple(group['weight'].shape)} / SCB {tuple(group['SCB'].shape)} != expected {(out_f, in_f)}")
            model.set_linear_int8(prefix, group["weight"], group["SCB"], out_f, in_f)
            stats["int8"] += 1
            continue
        qkey = next((k for k in group if k.startswith("weight.quant_state.bitsandbytes__")), None)
        if qkey is None or "weight" not in group or "weight.absmax" not in group:
            raise ValueError(f"`BnbLinear` expects fp4/nf4 layers: incomplete tensors for {prefix}: {sorted(group)}")  # bitsandbytes/mod.rs:120
        qt = qkey.rsplit("__", 1)[1]
        state = json.loads(bytes(group[qkey].numpy().tobytes()))
        if list(state["shape"]) != [out_f, in_f]:
            raise ValueError(f"{prefix}: quant_state shape {state['shape']} != expected {(out_f, in_f)}")
        absmax = _resolve_absmax(lib, group, state, model.device)
        model.set_linear_bnb4(prefix, group["weight"].reshape(-1), absmax, int(state["blocksize"]), qt, out_f, in_f)
        stats["bnb4"] += 1
    return stats


def load_flux(flux, tensors: Iterator[Tuple[str, torch.Tensor]]) -> dict:
    """Feed a FluxModel from (name, tensor) pairs; returns {"dense": n, "bnb4": n, "int8": n, "skipped": [...]}."""
    stats = _feed_linears(flux, tensors, synth.flux_tensor_shapes(flux.cfg))
    flux.assert_complete()
    return stats


def load_vae(vae, tensors: Iterator[Tuple[str, torch.Tensor]]) -> int:
    want = synth.vae_tensor_shapes(vae.cfg, encoder=True)  # encoder tensors are loaded when the checkpoint has them
    n = 0
    for name, t in tensors:
        if name in want:
            vae.set_tensor(name, t)
            n += 1
    return n


def load_text_encoder(model, tensors: Iterator[Tuple[str, torch.Tensor]]) -> int:
    """Feed a T5EncoderModel / ClipTextTransformer from (name, tensor) pairs; names the model does not read (decoder-tied
    embeddings, position_ids buffers, CLIP's unused text_projection) are skipped.  A bitsandbytes-quantised T5 (the reference
    builds every T5 Linear through `linear_no_bias(.., &cfg.quantization_config, ..)`, t5/mod.rs:132-173,258-261 — the
    FLUX.1-dev-Q4-bnb.dduf layout of README.md:37-41) is fed through the same group logic as the DiT."""
    want = model.tensor_names()

    def extra(name, t):
        if name == "encoder.embed_tokens.weight" and "shared.weight" in model.missing():
            model.set_tensor("shared.weight", t)  # T5EncoderModel::new falls back to it (t5/mod.rs:615-621)
            return True
        return False

    stats = _feed_linears(model, tensors, want, on_extra=extra)
    m = model.missing()
    if m:
        raise ValueError(f"{len(m)} text-encoder tensors missing, e.g. {m[0]}")
    model.load_stats = stats
    return stats["dense"] + stats["bnb4"] + stats["int8"]
