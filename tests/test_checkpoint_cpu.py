"""Checkpoint interop (utils/train_utils.py:147-183): a checkpoint written by the REAL reference
(tests/golden/g7_reference_checkpoint.bin, made by make_golden_variants.py) loads into the HIP-backed modules with
strict=True, and a checkpoint written here loads back -- and, where the reference is present, into the reference's modules."""
import json
import os
import subprocess
import sys

import pytest
import torch

from conftest import GOLDEN, ROOT

REF = "/root/reference/scripts"


def _ckpt(pkg):
    import importlib
    return importlib.import_module(pkg.__name__ + ".checkpoint")


def test_reference_checkpoint_loads_strict(pkg):
    ck = _ckpt(pkg)
    rep = json.load(open(os.path.join(GOLDEN, "golden_report_variants.json")))["checkpoint"]
    path = os.path.join(GOLDEN, "g7_reference_checkpoint.bin")
    raw = ck.load_checkpoint(path)
    assert sorted(raw) == ["args", "dis_dict", "epoch", "gen_dict", "lang_model", "pose_dim", "speaker_model"]
    assert isinstance(raw["lang_model"], pkg.Vocab) and isinstance(raw["speaker_model"], pkg.Vocab)   # model.vocab.Vocab mapped
    assert raw["lang_model"].n_words == rep["n_words"] and raw["speaker_model"].n_words == rep["n_speakers"]
    assert raw["lang_model"].get_word_index("fox") == raw["lang_model"].word2index["fox"] > 3
    assert raw["lang_model"].get_word_index("unseen") == pkg.Vocab.UNK_token
    args, gen, loss_fn, lang, spk, pose_dim = ck.load_checkpoint_and_model(path, "cpu")
    assert isinstance(gen, pkg.PoseGenerator) and not gen.training and pose_dim == 27 and loss_fn is None
    sd = gen.state_dict()
    assert len(sd) == rep["gen_keys"] and list(sd) == list(raw["gen_dict"])           # same keys, same order
    total = float(sum(v.double().abs().sum() for v in sd.values() if v.is_floating_point()))
    assert abs(total - rep["gen_abs_sum"]) < 1e-9 * rep["gen_abs_sum"]
    # the embedding came from lang_model.word_embedding_weights at construction and was then overwritten by gen_dict
    assert torch.equal(sd["text_encoder.embedding.weight"], raw["gen_dict"]["text_encoder.embedding.weight"])
    dis = pkg.ConvDiscriminator(27)
    dis.load_state_dict(raw["dis_dict"], strict=True)
    assert len(dis.state_dict()) == rep["dis_keys"]


def test_checkpoint_round_trip(pkg, tmp_path):
    ck = _ckpt(pkg)
    raw = ck.load_checkpoint(os.path.join(GOLDEN, "g7_reference_checkpoint.bin"))
    gen, dis, _ = ck.init_model(raw["args"], raw["lang_model"], raw["speaker_model"], raw["pose_dim"], "cpu")
    gen.load_state_dict(raw["gen_dict"]); dis.load_state_dict(raw["dis_dict"])
    out = str(tmp_path / "ours.bin")
    ck.save_checkpoint({"args": raw["args"], "epoch": 8, "lang_model": raw["lang_model"], "speaker_model": raw["speaker_model"],
                        "pose_dim": 27, "gen_dict": gen.state_dict(), "dis_dict": dis.state_dict()}, out)
    assert "model.vocab" not in sys.modules and pkg.Vocab.__module__.endswith(".vocab") and "model" != pkg.Vocab.__module__.split(".")[0]
    back = ck.load_checkpoint(out)
    assert back["epoch"] == 8 and isinstance(back["lang_model"], pkg.Vocab)
    assert back["lang_model"].word2index == raw["lang_model"].word2index
    for k, v in raw["gen_dict"].items():
        assert torch.equal(back["gen_dict"][k], v), k


@pytest.mark.skipif(not os.path.isdir(REF), reason="reference sources not present on this machine")
def test_reference_reads_our_checkpoint(pkg, tmp_path):
    """A checkpoint saved by this package unpickles in a process that only knows the reference's classes and loads into the
    reference's PoseGenerator / ConvDiscriminator with strict=True (separate process: the reference's top-level module
    names must not leak into this one)."""
    ck = _ckpt(pkg)
    raw = ck.load_checkpoint(os.path.join(GOLDEN, "g7_reference_checkpoint.bin"))
    gen, dis, _ = ck.init_model(raw["args"], raw["lang_model"], raw["speaker_model"], raw["pose_dim"], "cpu")
    gen.load_state_dict(raw["gen_dict"]); dis.load_state_dict(raw["dis_dict"])
    out = str(tmp_path / "ours.bin")
    ck.save_checkpoint({"args": raw["args"], "epoch": 9, "lang_model": raw["lang_model"], "speaker_model": raw["speaker_model"],
                        "pose_dim": 27, "gen_dict": gen.state_dict(), "dis_dict": dis.state_dict()}, out)
    code = f"""
import sys, types
sys.dont_write_bytecode = True
sys.path.insert(0, {REF!r})
for name in ("fasttext", "umap"):
    sys.modules.setdefault(name, types.ModuleType(name))
import torch
import model.embedding_net
import model.multimodal_context_net as mcn
import model.vocab as vocab
c = torch.load({out!r}, map_location="cpu", weights_only=False)
assert type(c["lang_model"]) is vocab.Vocab and type(c["speaker_model"]) is vocab.Vocab, type(c["lang_model"])
a = c["args"]
G = mcn.PoseGenerator(a, pose_dim=c["pose_dim"], n_words=c["lang_model"].n_words, word_embed_size=a.wordembed_dim,
                      word_embeddings=c["lang_model"].word_embedding_weights, z_obj=c["speaker_model"])
G.load_state_dict(c["gen_dict"], strict=True)
D = mcn.ConvDiscriminator(c["pose_dim"]); D.load_state_dict(c["dis_dict"], strict=True)
print("OK", c["epoch"], len(c["gen_dict"]))
"""
    env = dict(os.environ, PYTHONDONTWRITEBYTECODE="1")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, cwd=str(tmp_path), env=env)
    assert r.returncode == 0 and "OK 9" in r.stdout, r.stderr[-2000:]
