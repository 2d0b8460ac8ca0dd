"""Host logic of the checkpoint loader and the tokenisation helpers (CPU only, no HIP calls):
FileLoader over a diffusers directory and over a DDUF (stored zip, zero-copy slices of the mmap,
model_source.rs:87-259), rejection of compressed DDUF entries, tensor-shape inventories, and
FluxPipeline::tokenize_and_pad / load_bpe_tokenizer (flux/mod.rs:202-221, tokenizer.rs:7-23)."""
import json
import os
import zipfile

import numpy as np
import pytest
import torch


def _st_bytes(tensors):
    from safetensors.torch import save
    return save(tensors)


def test_file_loader_directory_and_dduf_views(tmp_path):
    from diffusion_rs_amd import loader
    a = {"w": torch.arange(12, dtype=torch.float32).reshape(3, 4), "b": torch.tensor([1, 2, 3], dtype=torch.uint8)}
    c = {"z": torch.ones(5, dtype=torch.bfloat16)}
    root = tmp_path / "ckpt"
    (root / "transformer").mkdir(parents=True)
    (root / "model_index.json").write_text(json.dumps({"_class_name": "FluxPipeline"}))
    (root / "transformer" / "part-00001-of-00002.safetensors").write_bytes(_st_bytes(a))
    (root / "transformer" / "part-00002-of-00002.safetensors").write_bytes(_st_bytes(c))
    (root / "transformer" / "notes.txt").write_text("hello")
    dduf = tmp_path / "ckpt.dduf"
    with zipfile.ZipFile(dduf, "w", compression=zipfile.ZIP_STORED) as z:
        for dp, _, fs in os.walk(root):
            for f in fs:
                full = os.path.join(dp, f)
                z.write(full, os.path.relpath(full, root))
    for src in (str(root), str(dduf)):
        fl = loader.FileLoader(src)
        assert fl.kind == ("dir" if src == str(root) else "dduf")
        assert "model_index.json" in fl.list_files() and fl.has("transformer/notes.txt") and not fl.has("vae/config.json")
        assert fl.read_json("model_index.json")["_class_name"] == "FluxPipeline"
        assert fl.read_text("transformer/notes.txt") == "hello"
        got = dict(fl.tensors("transformer"))  # shards in name order, every tensor of every shard
        assert list(got) == ["w", "b", "z"] or set(got) == {"w", "b", "z"}
        assert torch.equal(got["w"], a["w"]) and torch.equal(got["b"], a["b"]) and torch.equal(got["z"], c["z"])
        assert dict(fl.tensors("vae")) == {}


def test_dduf_rejects_compressed_entries(tmp_path):
    from diffusion_rs_amd import loader
    p = tmp_path / "bad.dduf"
    with zipfile.ZipFile(p, "w", compression=zipfile.ZIP_DEFLATED) as z:
        z.writestr("model_index.json", json.dumps({"_class_name": "FluxPipeline"}) * 50)
    with pytest.raises(ValueError, match="stored uncompressed"):
        loader.FileLoader(str(p))
    with pytest.raises(FileNotFoundError):
        loader.FileLoader(str(tmp_path / "black-forest-labs/FLUX.1-dev"))  # hub ids need a network: must be local


def test_tensor_inventories():
    import diffusion_rs_amd as d
    t5 = d.synth.t5_tensor_shapes(d.T5_XXL)
    assert t5["shared.weight"] == (32128, 4096) and t5["encoder.block.23.layer.1.DenseReluDense.wo.weight"] == (4096, 10240)
    assert sum(int(np.prod(s)) for s in t5.values()) == 4762310656  # t5-v1_1-xxl encoder
    relu = d.synth.t5_tensor_shapes(dict(d.T5_XXL, feed_forward_proj="relu"))
    assert "encoder.block.0.layer.1.DenseReluDense.wi.weight" in relu and "encoder.block.0.layer.1.DenseReluDense.wi_0.weight" not in relu
    clip = d.synth.clip_tensor_shapes(d.CLIP_L)
    assert clip["text_model.embeddings.position_embedding.weight"] == (77, 768) and len(clip) == 2 + 12 * 16 + 2
    dec = d.synth.vae_tensor_shapes(d.VAE_FLUX)
    both = d.synth.vae_tensor_shapes(d.VAE_FLUX, encoder=True)
    assert list(both)[:len(dec)] == list(dec)  # decoder names first: seeded synthetic decoder weights do not depend on the flag
    assert both["encoder.conv_out.weight"] == (32, 512, 3, 3) and both["encoder.down_blocks.2.downsamplers.0.conv.weight"] == (512, 512, 3, 3)
    assert not any(k.startswith("encoder.down_blocks.3.downsamplers") for k in both)


def test_tokenize_and_pad_and_bpe_builder():
    import diffusion_rs_amd as d
    from tokenizers import Tokenizer, models, pre_tokenizers, processors
    words = ["<pad>", "</s>", "<unk>", "the", "cat", "sat"]
    tk = Tokenizer(models.WordLevel({w: i for i, w in enumerate(words)}, unk_token="<unk>"))
    tk.pre_tokenizer = pre_tokenizers.Whitespace()
    tk.post_processor = processors.TemplateProcessing(single="$A </s>", special_tokens=[("</s>", 1)])
    rows = d.tokenize_and_pad(["the cat sat", "cat", "the dog"], tk)
    assert rows == [[3, 4, 5, 1], [4, 1, 0, 0], [3, 2, 1, 0]]  # special tokens added, zero-padded to the longest row
    # CLIP tokenizer as the reference builds it: bare BPE, first merges line skipped, malformed lines dropped,
    # no normalizer / pre-tokenizer / post-processor (no start / end tokens are added)
    vocab = {ch: i for i, ch in enumerate("abct h")}
    vocab.update({"th": 10, "at": 11, "cat": 12})
    merges = "#version: 0.2\nt h\na t\nbroken-line\nc at\n"
    bpe = d.load_bpe_tokenizer(json.dumps(vocab), merges)
    ids = bpe.encode("cat th", add_special_tokens=True).ids
    assert ids == [12, vocab[" "], 10]
