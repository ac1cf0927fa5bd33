"""Checkpoint interop with the reference (utils/train_utils.py:147-183, train.py:36-62,153-157).

A reference checkpoint is one pickle: {'args': argparse.Namespace, 'epoch', 'lang_model': model.vocab.Vocab,
'speaker_model': model.vocab.Vocab | 1 | None, 'pose_dim', 'gen_dict', 'dis_dict'}.  The state_dict keys of the HIP-backed
modules are the reference's (modules.py), so 'gen_dict' / 'dis_dict' load with strict=True; the pickled Vocab objects name the
reference's module path `model.vocab`, which is mapped onto this package's Vocab while unpickling (same attribute names), and
mapped back when saving so the reference can read checkpoints written here.
"""
import io
import pickle
import sys
import types

import torch

from . import vocab as _vocab
from .modules import ConvDiscriminator, EmbeddingNet, PoseGenerator

_REF_VOCAB_MODULE = "model.vocab"


class _RefUnpickler(pickle.Unpickler):
    def find_class(self, module, name):
        if module == _REF_VOCAB_MODULE and name == "Vocab":
            return _vocab.Vocab
        return super().find_class(module, name)


# pickle_module for torch.load: the stdlib pickle with the Unpickler above, so model.vocab.Vocab resolves without the
# reference being importable
_ref_pickle = types.ModuleType("trimodal_ref_pickle")
_ref_pickle.__dict__.update({k: getattr(pickle, k) for k in dir(pickle) if not k.startswith("__")})
_ref_pickle.Unpickler = _RefUnpickler
_ref_pickle.load = lambda f, **kw: _RefUnpickler(f, **kw).load()
_ref_pickle.loads = lambda b, **kw: _RefUnpickler(io.BytesIO(b), **kw).load()


def init_model(args, lang_model, speaker_model, pose_dim, _device):
    """train.py:36-62 for the models on the hot path ('multimodal_context', 'gesture_autoencoder')."""
    generator = discriminator = loss_fn = None
    if args.model == "multimodal_context":
        generator = PoseGenerator(args, n_words=lang_model.n_words, word_embed_size=args.wordembed_dim,
                                  word_embeddings=lang_model.word_embedding_weights, z_obj=speaker_model,
                                  pose_dim=pose_dim).to(_device)
        discriminator = ConvDiscriminator(pose_dim).to(_device)
    elif args.model == "gesture_autoencoder":
        generator = EmbeddingNet(args, pose_dim, args.n_poses, lang_model.n_words, args.wordembed_dim,
                                 lang_model.word_embedding_weights, mode="pose").to(_device)
    else:
        raise NotImplementedError(f"model {args.model!r} is a baseline outside the hot path (SURVEY.md section 8)")
    return generator, discriminator, loss_fn


def get_speaker_model(net):
    """utils/train_utils.py:152-164: the generator's speaker Vocab (through a DataParallel-style `.module` wrapper too), or None when
    z_obj is not a Vocab (z_type 'random' keeps the integer 1 there, 'none' keeps None)."""
    try:
        speaker_model = net.module.z_obj if hasattr(net, "module") else net.z_obj
    except AttributeError:
        speaker_model = None
    if not isinstance(speaker_model, _vocab.Vocab):
        speaker_model = None
    return speaker_model


def load_checkpoint(checkpoint_path, _device="cpu"):
    """torch.load of a reference-format checkpoint (a full pickle: weights_only=False, as the reference's torch.load)."""
    return torch.load(checkpoint_path, map_location=_device, weights_only=False, pickle_module=_ref_pickle)


def load_checkpoint_and_model(checkpoint_path, _device="cpu"):
    """utils/train_utils.py:167-183: returns (args, generator, loss_fn, lang_model, speaker_model, pose_dim), generator in
    eval mode."""
    checkpoint = load_checkpoint(checkpoint_path, _device)
    args, lang_model, speaker_model = checkpoint["args"], checkpoint["lang_model"], checkpoint["speaker_model"]
    pose_dim = checkpoint["pose_dim"]
    generator, discriminator, loss_fn = init_model(args, lang_model, speaker_model, pose_dim, _device)
    generator.load_state_dict(checkpoint["gen_dict"])
    generator.train(False)
    return args, generator, loss_fn, lang_model, speaker_model, pose_dim


def save_checkpoint(state, filename):
    """utils/train_utils.py:147-149.  Vocab objects are pickled under the reference's module path so that the reference's
    own torch.load (with its `model.vocab` importable) reads the file."""
    had = sys.modules.get(_REF_VOCAB_MODULE)
    old_module = _vocab.Vocab.__module__
    if had is None:                      # temporary alias so pickle's "is it the same object" lookup succeeds
        parent = sys.modules.setdefault("model", types.ModuleType("model"))
        alias = types.ModuleType(_REF_VOCAB_MODULE)
        alias.Vocab = _vocab.Vocab
        sys.modules[_REF_VOCAB_MODULE] = alias
        parent.vocab = alias
    try:
        if had is None:
            _vocab.Vocab.__module__ = _REF_VOCAB_MODULE
        torch.save(state, filename)
    finally:
        _vocab.Vocab.__module__ = old_module
        if had is None:
            del sys.modules[_REF_VOCAB_MODULE]
            if getattr(sys.modules.get("model"), "vocab", None) is alias:
                del sys.modules["model"].vocab
            if not [k for k in vars(sys.modules["model"]) if not k.startswith("__")]:
                del sys.modules["model"]
