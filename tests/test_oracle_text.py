"""Pins oracle/text_oracle.cpp (T5EncoderModel, ClipTextTransformer restatements) against the
HuggingFace transformers outputs committed in tests/golden/text_encoders.npz
(generator: tests/golden/gen_text_fixtures.py).  CPU only."""
import os

import numpy as np
import pytest

from tests.golden.gen_text_fixtures import CLIP_CFG, T5_CFG

GOLD = os.path.join(os.path.dirname(__file__), "golden", "text_encoders.npz")


@pytest.fixture(scope="module")
def gold():
    return np.load(GOLD)


def _weights(gold, prefix):
    return {k[len(prefix):]: gold[k] for k in gold.files if k.startswith(prefix)}


def test_t5_relative_position_bucket_matches_hf_formula():
    """The reference's bucket code (t5/mod.rs:340-376) vs the HF _relative_position_bucket closed form."""
    from oracle import oracle as orc
    import math
    nb, md = 32, 128
    half, max_exact = nb // 2, nb // 4
    for i in range(0, 300, 7):
        for j in range(0, 300, 5):
            rel = j - i
            b = half if rel > 0 else 0
            a = abs(rel)
            if a < max_exact:
                b += a
            else:
                b += min(max_exact + int(math.log(a / max_exact) / math.log(md / max_exact) * (half - max_exact)), half - 1)
            assert orc.t5_bucket(i, j, nb, md) == b, (i, j)


@pytest.mark.parametrize("tag", ["short", "long"])
def test_t5_oracle_matches_transformers(gold, tag):
    from oracle import oracle as orc
    m = orc.T5(T5_CFG)
    m.load(_weights(gold, "t5_w/"))
    out = m.forward(gold[f"t5_ids_{tag}"])
    ref = gold[f"t5_out_{tag}"]
    err = np.linalg.norm(out - ref) / np.linalg.norm(ref)
    print(f"T5 oracle vs transformers ({tag}): rel-L2 {err:.2e}, max |d| {np.abs(out - ref).max():.2e}")
    assert err <= 2e-5


def test_clip_oracle_matches_transformers(gold):
    from oracle import oracle as orc
    m = orc.Clip(CLIP_CFG)
    m.load(_weights(gold, "clip_w/"))
    pooled, hid = m.forward(gold["clip_ids"], return_hidden=True)
    for got, ref, name in ((hid, gold["clip_hidden"], "hidden"), (pooled, gold["clip_pooled"], "pooled")):
        err = np.linalg.norm(got - ref) / np.linalg.norm(ref)
        print(f"CLIP oracle vs transformers ({name}): rel-L2 {err:.2e}")
        assert err <= 2e-5
