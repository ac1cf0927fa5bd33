#!/usr/bin/env python3
"""Golden vectors for the callers either side of the hot path, from the REAL reference (build container only):

  g8_evaluate_testset.npz   scripts/train.py:evaluate_testset (:234-329) over three synthetic batches, z_type speaker / random / none,
                            with the reference's EmbeddingSpaceEvaluator (FGD) attached; utils/data_utils.convert_dir_vec_to_pose (:77-98)
                            and convert_pose_seq_to_dir_vec (:101-121) on random input
  g9_generate_gestures.npz  scripts/synthesize.py:generate_gestures (:36-209): 1-, 2- and 4-window utterances, fade_out False / True,
                            seed poses, given / randomly drawn speaker id, z_type speaker / random / none; every window's inputs recorded
  g10_dataset.npz           data_loader/lmdb_data_loader.py: SpeechMotionDataset.__getitem__ (:107-171), default_collate_fn (:43-53),
                            utils/data_utils.make_audio_fixed_length (:68-74)
  g11_ae_eval.npz           train_eval/train_joint_embed.eval_embed (:54-62), train_feature_extractor.evaluate_testset (:26-51)

train.py / synthesize.py / lmdb_data_loader.py import libraries this image lacks (lmdb, librosa, soundfile, tensorboard, gentle, ...):
they are replaced by EMPTY stub modules before the import -- none of them is touched by the functions exercised here.  The dataset
object is created with __new__ (its __init__ only opens LMDB) and given an in-memory stand-in for the LMDB environment; the stored
value is the sample tuple itself and `pyarrow.deserialize` is replaced by the identity (pyarrow >= 2 no longer has it).

While generating, oracle/ref_model.py is checked against the reference on the full tensors; errors go to golden_report_eval.json.
Fixtures hold tensors, scalars and word strings only.

    python tests/golden/make_golden_eval.py
"""
import argparse
import json
import os
import random
import sys
import tempfile
import types
from collections import OrderedDict

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
sys.dont_write_bytecode = True

REF = "/root/reference/scripts"
MEAN_DIR_VEC = [0.0154009, -0.9690125, -0.0884354, -0.0022264, -0.8655276, 0.4342174, -0.0035145, -0.8755367, -0.4121039, -0.9236511,
                0.3061306, -0.0012415, -0.5155854, 0.8129665, 0.0871897, 0.2348464, 0.1846561, 0.8091402, 0.9271948, 0.2960011, -0.013189,
                0.5233978, 0.8092403, 0.0725451, -0.2037076, 0.1924306, 0.8196916]            # config/multimodal_context.yml:16 (data)
STUBS = ("librosa", "librosa.display", "lmdb", "soundfile", "fasttext", "umap", "configargparse", "torch.utils.tensorboard",
         "gentle", "pygame", "google", "google.cloud", "google.cloud.texttospeech")


class _Stub(types.ModuleType):
    """Empty module: any attribute is another empty module / a callable returning one (module-level `gentle.Resources()`)."""

    def __getattr__(self, n):
        if n.startswith("__"):
            raise AttributeError(n)
        m = _Stub(self.__name__ + "." + n)
        setattr(self, n, m)
        return m

    def __call__(self, *a, **k):
        return _Stub("call")


def import_reference_callers():
    sys.path[:0] = [REF, "/root/reference"]
    for name in STUBS:
        sys.modules.setdefault(name, _Stub(name))
    import model.embedding_net as embedding_net             # first: circular import
    import model.multimodal_context_net as mcn
    import model.vocab as vocab
    import train
    import synthesize
    import train_feature_extractor
    import train_eval.train_joint_embed as tje
    import data_loader.lmdb_data_loader as ldl
    import utils.data_utils as du
    from model.embedding_space_evaluator import EmbeddingSpaceEvaluator
    return dict(embedding_net=embedding_net, mcn=mcn, vocab=vocab, train=train, synthesize=synthesize, tfe=train_feature_extractor,
                tje=tje, ldl=ldl, du=du, Evaluator=EmbeddingSpaceEvaluator)


def ref_args(z_type="speaker", input_context="both"):
    return argparse.Namespace(n_pre_poses=4, n_poses=34, input_context=input_context, hidden_size=300, n_layers=4, dropout_prob=0.3,
                              freeze_wordembed=False, z_type=z_type, loss_warmup=10, loss_gan_weight=5.0, loss_regression_weight=500,
                              loss_kld_weight=0.1, loss_reg_weight=0.05, learning_rate=0.0005, discriminator_lr_weight=0.2,
                              model="multimodal_context", wordembed_dim=300, motion_resampling_framerate=15,
                              mean_dir_vec=list(MEAN_DIR_VEC))


def maxerr(a, b):
    a, b = torch.as_tensor(np.asarray(a)).double(), torch.as_tensor(np.asarray(b)).double()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


WORDS = ("so what i want to talk about today is how we move our hands when we speak and why it matters for the people who listen "
         "because gesture carries meaning that words alone do not").split()


def make_lang(vocab_cls, n_words=512):
    """A word Vocab of exactly n_words entries: real words first, then fillers (so that ids < n_words = embedding rows)."""
    lang = vocab_cls("words")
    for w in WORDS:
        lang.index_word(w)
    i = 0
    while lang.n_words < n_words:
        lang.index_word(f"filler{i}")
        i += 1
    lang.word_embedding_weights = np.zeros((lang.n_words, 300), dtype=np.float32)
    return lang


def make_speakers(vocab_cls, n):
    spk = vocab_cls("vid", insert_default_tokens=False)
    for i in range(n - 1):
        spk.index_word(f"spk{i}")
    return spk


def build_generator(R, O, z_type, V, S, g_seed, input_context="both"):
    z_mode = z_type if z_type in ("speaker", "random") else None
    gst = O.make_generator_state(g_seed, V, S, input_context=input_context, z_mode=z_mode)
    args = ref_args(z_type, input_context)
    spk = make_speakers(R["vocab"].Vocab, S)
    z_obj = spk if z_type == "speaker" else (1 if z_type == "random" else None)              # train.py:82-87
    G = R["mcn"].PoseGenerator(args, pose_dim=27, n_words=V, word_embed_size=300, word_embeddings=np.zeros((V, 300), np.float32),
                               z_obj=z_obj)
    G.load_state_dict(O.clone_state(gst), strict=True)
    return args, G, gst, spk, z_mode


class DrawRecorder:
    """Records reparameterize's eps, the 'random' z draws and python's random.choice / randrange on the reference path."""

    def __init__(self, R, seed):
        self.R, self.gen = R, torch.Generator().manual_seed(seed)
        self.eps, self.zs, self.choices, self.randranges = [], [], [], []

    def __enter__(self):
        en = self.R["embedding_net"]
        self._reparam, self._randn, self._choice, self._randrange = en.reparameterize, torch.randn, random.choice, random.randrange

        def reparameterize(mu, logvar):
            std = torch.exp(0.5 * logvar)
            eps = torch.randn(std.shape, generator=self.gen)
            self.eps.append(eps.numpy().copy())
            return mu + eps * std

        def randn(*size, **kw):
            kw.pop("device", None)
            if "generator" in kw:
                return self._randn(*size, **kw)
            z = self._randn(*size, generator=self.gen, **kw)
            self.zs.append(z.numpy().copy())
            return z

        def choice(seq):
            c = self._choice(seq)
            self.choices.append(c)
            return c

        def randrange(*a):
            c = self._randrange(*a)
            self.randranges.append(c)
            return c
        en.reparameterize, torch.randn, random.choice, random.randrange = reparameterize, randn, choice, randrange
        return self

    def __exit__(self, *exc):
        self.R["embedding_net"].reparameterize, torch.randn, random.choice, random.randrange = \
            self._reparam, self._randn, self._choice, self._randrange
        return False


def make_eval_batches(O, V, S, sizes, seed0):
    out = []
    for i, b in enumerate(sizes):
        text, audio, _vid, poses = O.make_batch(seed0 + i, b, V, S)
        out.append((text, poses, audio))
    return out


def as_loader(batches):
    """The 8-tuple the reference's DataLoader yields (default_collate_fn, lmdb_data_loader.py:43-53)."""
    return [(torch.tensor([0]), torch.tensor([0]), text, torch.zeros(text.shape[0], 34, 30), poses.clone(), audio,
             torch.zeros(text.shape[0], 1), {}) for text, poses, audio in batches]


# ====================================================================================================================
def gen_evaluate_testset(R, O, report, store):
    V, S = 512, 17
    ast = O.make_autoencoder_state(2)
    tmp = tempfile.mkdtemp()
    ae_path = os.path.join(tmp, "ae.bin")
    torch.save({"args": ref_args(), "epoch": 1, "pose_dim": 27, "gen_dict": O.clone_state(ast)}, ae_path)   # train_feature_extractor.py:155-157
    lang = make_lang(R["vocab"].Vocab, V)
    sizes = (5, 4, 3)
    store.update(n_words=V, n_speakers=S, g_seed=30, ae_seed=2, batch_seed0=500, sizes=np.array(sizes),
                 mean_dir_vec=np.array(MEAN_DIR_VEC))
    for zt in ("speaker", "random", "none"):
        args, G, gst, spk, z_mode = build_generator(R, O, zt, V, S, 30)
        evaluator = R["Evaluator"](args, ae_path, lang, torch.device("cpu"))
        batches = make_eval_batches(O, V, S, sizes, 500)
        meters = {}
        AM = R["train"].AverageMeter

        class SpyMeter(AM):
            def __init__(self, name, *a, **k):
                super().__init__(name, *a, **k)
                meters[name] = self
        R["train"].AverageMeter = SpyMeter
        outs = []
        fwd = G.forward

        def spy_forward(*a, **k):
            r = fwd(*a, **k)
            outs.append(r[0].detach().clone())
            return r
        G.forward = spy_forward
        random.seed(1234)
        G.train(True)
        with DrawRecorder(R, 777) as rec:
            ret = R["train"].evaluate_testset(as_loader(batches), G, None, evaluator, args)
        R["train"].AverageMeter = AM
        assert G.training, "the reference leaves the generator in train mode (:313)"
        accel = meters["accel"].avg
        vids = None
        if zt == "speaker":
            it = iter(rec.choices)
            vids = [torch.tensor([next(it) for _ in range(b)], dtype=torch.int64) for b in sizes]
            assert len(rec.choices) == sum(sizes)
        else:
            assert not rec.choices, "no speaker Vocab -> vid_indices = None (train_utils.py:152-164)"
        inj = {f"e{i}.eps": torch.from_numpy(e) for i, e in enumerate(rec.eps)}
        inj.update({f"e{i}.z": torch.from_numpy(z) for i, z in enumerate(rec.zs)})
        oret, oouts = O.evaluate_testset(O.clone_state(gst), batches, MEAN_DIR_VEC, O.Rand(inject=inj), ast=O.clone_state(ast),
                                         vids=vids, z_mode=z_mode)
        errs = {k: abs(oret[k] - ret[k]) / max(abs(ret[k]), 1e-12) for k in ret}
        errs["accel"] = abs(oret["accel"] - accel) / accel
        errs["out"] = max(maxerr(a, b) for a, b in zip(oouts, outs))
        report["evaluate_testset_" + zt] = errs
        store[f"{zt}/ret_keys"] = np.array(sorted(ret))
        store[f"{zt}/ret_vals"] = np.array([ret[k] for k in sorted(ret)])
        store[f"{zt}/accel"] = np.array(accel)
        store[f"{zt}/eps"] = np.concatenate(rec.eps) if rec.eps else np.zeros((0, 16), np.float32)
        store[f"{zt}/z"] = np.concatenate(rec.zs) if rec.zs else np.zeros((0, 16), np.float32)
        store[f"{zt}/vids"] = np.concatenate([v.numpy() for v in vids]) if vids else np.zeros((0,), np.int64)
        store[f"{zt}/out"] = np.concatenate([o.numpy() for o in outs])
    # ---- bone integration and its inverse on random input (utils/data_utils.py:77-121)
    r = np.random.RandomState(5)
    v4 = r.randn(3, 7, 27).astype(np.float32)
    store["dir_vec/in"] = v4
    store["dir_vec/pose_b_t"] = R["du"].convert_dir_vec_to_pose(v4)                   # (3,7,10,3) via the 4-D branch
    store["dir_vec/pose_t"] = R["du"].convert_dir_vec_to_pose(v4[0])                  # (7,10,3) via the 3-D branch
    store["dir_vec/pose_single"] = R["du"].convert_dir_vec_to_pose(v4[0, 0])          # (10,3) via the 2-D branch
    pose = r.randn(2, 6, 10, 3)
    pose[0, 0, 2] = pose[0, 0, 1]                                                        # a zero-length bone
    store["dir_vec/pose_in"] = pose
    store["dir_vec/vec_from_pose"] = R["du"].convert_pose_seq_to_dir_vec(pose)
    store["dir_vec/vec_from_pose_t"] = R["du"].convert_pose_seq_to_dir_vec(pose[1])
    report["dir_vec_to_pose"] = max(maxerr(O.dir_vec_to_pose(v4), store["dir_vec/pose_b_t"]),
                                    maxerr(O.dir_vec_to_pose(v4[0]), store["dir_vec/pose_t"]),
                                    maxerr(O.dir_vec_to_pose(v4[0, 0]), store["dir_vec/pose_single"]))
    report["pose_seq_to_dir_vec"] = max(maxerr(O.pose_seq_to_dir_vec(pose), store["dir_vec/vec_from_pose"]),
                                        maxerr(O.pose_seq_to_dir_vec(pose[1]), store["dir_vec/vec_from_pose_t"]))


def synth_words(duration, seed):
    r = np.random.RandomState(seed)
    t, words = 0.05, []
    while t < duration - 0.1:
        d = float(r.uniform(0.12, 0.5))
        words.append([WORDS[int(r.randint(len(WORDS)))] if r.rand() > 0.1 else "zzzunknown", round(t, 3), round(t + d, 3)])
        t += d + float(r.uniform(0.0, 0.6))
    return words


SYNTH_CASES = (  # name, z_type, clip seconds, fade_out, vid, seed poses, words seed
    ("w1", "speaker", 1.5, False, 5, False, 1), ("w1_fade", "speaker", 1.5, True, 5, False, 1),
    ("w2", "speaker", 3.5, False, 3, True, 2), ("w2_fade", "speaker", 3.5, True, 3, True, 2),
    ("w4", "speaker", 8.0, False, None, False, 3), ("w4_fade", "speaker", 8.0, True, None, True, 3),
    ("w2_exact", "speaker", 68266 / 16000, True, 7, False, 4),                 # no padding in the last window: fade-out appends frames
    ("w3_random", "random", 6.2, True, None, False, 5), ("w2_none", "none", 4.1, False, None, True, 6))


def gen_generate_gestures(R, O, report, store):
    V, S = 512, 17
    lang = make_lang(R["vocab"].Vocab, V)
    store.update(n_words=V, n_speakers=S, g_seed=31, cases=np.array([c[0] for c in SYNTH_CASES]), vocab_words=np.array(WORDS))
    built = {}
    for name, zt, secs, fade, vid, use_seed, wseed in SYNTH_CASES:
        if zt not in built:
            built[zt] = build_generator(R, O, zt, V, S, 31)
        args, G, gst, spk, z_mode = built[zt]
        G.train(False)
        r = np.random.RandomState(100 + wseed)
        audio = (0.1 * r.randn(int(round(secs * 16000)))).astype(np.float16).astype(np.float32)   # fp16-exact: halves the fixture
        words = synth_words(secs, wseed)
        seed_seq = (0.1 * r.randn(6, 27)).astype(np.float32) if use_seed else None
        calls = []
        fwd = G.forward

        def spy_forward(pre_seq, in_text, in_audio, vidx=None, _fwd=fwd, _calls=calls):
            _calls.append((pre_seq.detach().clone(), in_text.clone(), in_audio.clone(), None if vidx is None else vidx.clone()))
            return _fwd(pre_seq, in_text, in_audio, vidx)
        G.forward = spy_forward
        random.seed(4321 + wseed)
        devnull = open(os.devnull, "w")
        stdout, sys.stdout = sys.stdout, devnull                                # the reference prints every word
        try:
            with DrawRecorder(R, 900 + wseed) as rec:
                out = R["synthesize"].generate_gestures(args, G, lang, audio, words, vid=vid, seed_seq=seed_seq, fade_out=fade)
        finally:
            sys.stdout = stdout
            devnull.close()
            del G.forward
        used_vid = None
        if zt == "speaker":
            used_vid = int(calls[0][3][0])
            assert (vid is None) == bool(rec.randranges) and (vid is None or used_vid == vid)
        else:
            assert all(c[3] is None for c in calls) and not rec.randranges
        inj = {f"w{i}.eps": torch.from_numpy(e) for i, e in enumerate(rec.eps)}
        inj.update({f"w{i}.z": torch.from_numpy(z) for i, z in enumerate(rec.zs)})
        wins = []
        o_out = O.generate_gestures(O.clone_state(gst), audio, words, lang.get_word_index, O.Rand(inject=inj), vid=used_vid,
                                    seed_seq=seed_seq, fade_out=fade, z_mode=z_mode, windows=wins)
        assert len(wins) == len(calls) == O.num_windows(len(audio) / 16000)
        report["generate_gestures_" + name] = dict(
            out=maxerr(o_out, out), shape=list(out.shape), windows=len(calls),
            pre_seq=max(maxerr(w[0], c[0]) if float(c[0].abs().max()) > 0 else float(w[0].abs().max()) for w, c in zip(wins, calls)),
            text_equal=bool(all(torch.equal(w[1], c[1]) for w, c in zip(wins, calls))),
            audio_equal=bool(all(torch.equal(w[2], c[2]) for w, c in zip(wins, calls))))
        assert o_out.shape == out.shape, (name, o_out.shape, out.shape)
        store[f"{name}/z_type"] = np.array(zt)
        store[f"{name}/audio"] = audio.astype(np.float16)
        store[f"{name}/words"] = np.array([w[0] for w in words])
        store[f"{name}/word_times"] = np.array([[w[1], w[2]] for w in words])
        store[f"{name}/fade_out"] = np.array(fade)
        store[f"{name}/vid_arg"] = np.array(-1 if vid is None else vid)
        store[f"{name}/vid_used"] = np.array(-1 if used_vid is None else used_vid)
        store[f"{name}/seed_seq"] = seed_seq if seed_seq is not None else np.zeros((0, 27), np.float32)
        store[f"{name}/eps"] = np.concatenate(rec.eps) if rec.eps else np.zeros((0, 16), np.float32)
        store[f"{name}/z"] = np.concatenate(rec.zs) if rec.zs else np.zeros((0, 16), np.float32)
        store[f"{name}/out"] = out
        store[f"{name}/win_pre_seq"] = np.concatenate([c[0].numpy() for c in calls])
        store[f"{name}/win_text"] = np.concatenate([c[1].numpy() for c in calls])
        # each window's audio slice is 36 266 samples of the utterance: stored as its start-aligned head / tail and a checksum
        wa = np.concatenate([c[2].numpy() for c in calls])
        store[f"{name}/win_audio_head"], store[f"{name}/win_audio_tail"] = wa[:, :64].copy(), wa[:, -3000:].copy()
        store[f"{name}/win_audio_abs_sum"] = np.abs(wa.astype(np.float64)).sum(axis=1)


class _FakeTxn:
    def __init__(self, samples):
        self.samples = samples

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False

    def get(self, key):
        return self.samples[int(key.decode("ascii"))]

    def stat(self):
        return {"entries": len(self.samples)}


class _FakeEnv:
    def __init__(self, samples):
        self.samples = samples

    def begin(self, write=False):
        return _FakeTxn(self.samples)


def make_raw_samples(n=6):
    """Samples in the preprocessor's stored format [words, poses, normalized_dir_vec, audio, spectrogram, aux]
    (data_preprocessor.py:160-164); lengths and timings chosen to hit every branch of __getitem__."""
    r = np.random.RandomState(77)
    out = []
    for i in range(n):
        n_ext = 42 if i != 4 else 34                                           # int(round(34 * 1.25)); one clip without margin
        start = float(np.round(r.uniform(0, 50), 3))
        dur = n_ext / 15
        n_audio = int(dur * 16000) + (0, -300, 250, -36000, 0, 11)[i]          # longer / shorter than the 36 267 expected samples
        words, t = [], start - 0.3                                             # first onset before the clip start -> frame 0 clamp
        while t < start + dur:
            w = WORDS[int(r.randint(len(WORDS)))] if r.rand() > 0.15 else "notinvocab"
            words.append([w, float(np.round(t, 3)), float(np.round(t + 0.25, 3))])
            t += float(r.uniform(0.15, 0.9))
        aux = {"vid": f"spk{int(r.randint(16))}", "start_frame_no": 10 * i, "end_frame_no": 10 * i + n_ext, "start_time": start,
               "end_time": start + dur}
        out.append([words, r.randn(n_ext, 10, 3).astype(np.float32) * 0.2, r.randn(n_ext, 9, 3).astype(np.float32) * 0.3,
                    (0.1 * r.randn(n_audio)).astype(np.float16).astype(np.float32), r.randn(128, 90).astype(np.float16), aux])
    return out


def gen_dataset(R, O, report, store):
    ldl = R["ldl"]
    lang = make_lang(R["vocab"].Vocab, 128)
    samples = make_raw_samples()
    ldl.pyarrow.deserialize = lambda v: v                 # the stored value IS the sample tuple (see the module docstring)
    errs = {}
    store.update(n_samples=len(samples), vocab_words=np.array(WORDS), vocab_size=128)
    for i, s in enumerate(samples):
        store[f"raw{i}/words"] = np.array([w[0] for w in s[0]])
        store[f"raw{i}/word_times"] = np.array([[w[1], w[2]] for w in s[0]])
        store[f"raw{i}/pose"], store[f"raw{i}/vec"], store[f"raw{i}/audio"], store[f"raw{i}/spec"] = s[1], s[2], s[3].astype(np.float16), s[4]
        store[f"raw{i}/vid"] = np.array(s[5]["vid"])
        store[f"raw{i}/aux"] = np.array([s[5]["start_frame_no"], s[5]["end_frame_no"], s[5]["start_time"], s[5]["end_time"]])
    for rwt in (False, True):
        ds = ldl.SpeechMotionDataset.__new__(ldl.SpeechMotionDataset)
        ds.n_poses, ds.subdivision_stride, ds.skeleton_resampling_fps, ds.remove_word_timing = 34, 10, 15, rwt
        ds.expected_audio_length = int(round(34 / 15 * 16000))                                  # :62
        ds.expected_spectrogram_length = R["du"].calc_spectrogram_length_from_motion_length(34, 15)   # :63-64
        ds.lmdb_env, ds.n_samples = _FakeEnv(samples), len(samples)
        ds.set_lang_model(lang)
        items = [ds[i] for i in range(len(ds))]
        tag = "rwt" if rwt else "timed"
        for i, it in enumerate(items):
            words_t, ext, pose, vec, audio, spec, aux = it
            o = O.data_getitem(samples[i], lang.get_word_index, remove_word_timing=rwt, full=True)
            errs[f"{tag}{i}"] = dict(words=bool(np.array_equal(o[0], words_t.numpy())), ext=bool(np.array_equal(o[1], ext.numpy())),
                                     pose=bool(np.array_equal(o[2], pose.numpy())), vec=bool(np.array_equal(o[3], vec.numpy())),
                                     audio=bool(np.array_equal(o[4], audio.numpy())), spec=bool(np.array_equal(o[5], spec.numpy())))
            store[f"{tag}{i}/words"], store[f"{tag}{i}/ext"] = words_t.numpy(), ext.numpy()
            if not rwt:
                store[f"item{i}/pose"], store[f"item{i}/vec"] = pose.numpy(), vec.numpy()
                a_np = audio.numpy()                      # fixed-length audio: head, the (possibly mirrored) tail and a checksum
                store[f"item{i}/audio_len"] = np.array(a_np.shape[0])
                store[f"item{i}/audio_head"], store[f"item{i}/audio_tail"] = a_np[:64].copy(), a_np[-1500:].copy()
                store[f"item{i}/audio_abs_sum"] = np.array(np.abs(a_np.astype(np.float64)).sum())
                store[f"item{i}/spec_shape"] = np.array(spec.shape)
        if not rwt:
            col = ldl.default_collate_fn(items[:4])
            store["collate/word_seq"], store["collate/lengths"] = col[0].numpy(), col[1].numpy()
            store["collate/text"], store["collate/pose"], store["collate/vec"] = (c.numpy() for c in col[2:5])
            assert torch.equal(col[5], torch.stack([it[4] for it in items[:4]]))
            store["collate/audio_shape"] = np.array(col[5].shape)
            store["collate/spec_shape"] = np.array(col[6].shape)
            store["collate/aux_keys"] = np.array(sorted(col[7]))
            store["collate/aux_vid"] = np.array(col[7]["vid"])
            store["collate/aux_start_time"] = col[7]["start_time"].numpy()
    r = np.random.RandomState(9)
    a = r.randn(1000).astype(np.float32)
    store["fixlen/in"] = a
    store["fixlen/longer"] = R["du"].make_audio_fixed_length(a, 1500)
    store["fixlen/shorter"] = R["du"].make_audio_fixed_length(a, 700)
    store["fixlen/same"] = R["du"].make_audio_fixed_length(a, 1000)
    errs["fixlen"] = bool(np.array_equal(O.data_make_audio_fixed_length(a, 1500), store["fixlen/longer"]) and
                          np.array_equal(O.data_make_audio_fixed_length(a, 700), store["fixlen/shorter"]))
    report["dataset"] = errs
    assert all(all(v.values()) if isinstance(v, dict) else v for v in errs.values()), errs


def gen_ae_eval(R, O, report, store):
    ast = O.make_autoencoder_state(2)
    AE = R["embedding_net"].EmbeddingNet(ref_args(), 27, 34, None, None, None, mode="pose")
    AE.load_state_dict(O.clone_state(ast))
    g = torch.Generator().manual_seed(66)
    batches = [0.1 * torch.randn(b, 34, 27, generator=g) for b in (6, 6, 3)]
    AE.train(False)
    with torch.no_grad():
        loss, recon = R["tje"].eval_embed(None, None, None, batches[0], AE)
    loader = [(torch.zeros(b.shape[0], 34, 30), b) for b in batches]            # (target_poses, target_vec) of Human36M (h36m_loader.py)
    ret = R["tfe"].evaluate_testset(loader, AE)
    assert AE.training
    o_loss, o_recon = O.eval_embed(O.clone_state(ast), batches[0])
    o_ret = O.ae_evaluate_testset(O.clone_state(ast), batches)
    report["ae_eval"] = dict(eval_embed_loss=abs(o_loss - float(loss)) / float(loss), recon=maxerr(o_recon, recon),
                             evaluate_testset=abs(o_ret["loss"] - ret["loss"]) / ret["loss"])
    store.update(ae_seed=2, pose_seed=66, sizes=np.array([6, 6, 3]), eval_embed_loss=float(loss), recon=recon.numpy(),
                 evaluate_testset_loss=ret["loss"])


def main():
    from oracle import ref_model as O
    R = import_reference_callers()
    torch.set_num_threads(8)
    torch.serialization.add_safe_globals([argparse.Namespace])      # the reference's plain torch.load (evaluator :20) under torch >= 2.6
    report = OrderedDict(torch=torch.__version__, numpy=np.__version__)
    for name, fn in (("g8_evaluate_testset", gen_evaluate_testset), ("g9_generate_gestures", gen_generate_gestures),
                     ("g10_dataset", gen_dataset), ("g11_ae_eval", gen_ae_eval)):
        store = {}
        fn(R, O, report, store)
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **store)
    with open(os.path.join(HERE, "golden_report_eval.json"), "w") as f:
        json.dump(report, f, indent=1, default=float)
    print(json.dumps(report, indent=1, default=float))


if __name__ == "__main__":
    main()
