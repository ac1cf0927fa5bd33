"""Host logic either side of the hot path -- the input pipeline (data.py), window slicing / fade-out (synthesize.py), bone integration
and speaker-model lookup -- against fixtures the REAL reference produced (tests/golden/make_golden_eval.py: SpeechMotionDataset.__getitem__,
default_collate_fn, generate_gestures, convert_dir_vec_to_pose) and against the oracle pinned by the same fixtures."""
import argparse
import importlib
import os

import numpy as np
import torch

from conftest import GOLDEN
from oracle import ref_model as O
from tests.harness import check_window_audio, dataset_samples, fixture_lang, synth_case


def load(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


def _mods(pkg):
    return importlib.import_module(pkg.__name__ + ".data")


def _lang(pkg, words):
    v = pkg.Vocab("words")
    for w in words:
        v.index_word(w)
    return v


def test_extend_word_seq_hand_computed(pkg):
    D = _mods(pkg)
    lang = _lang(pkg, ["a", "b", "c", "d"])                       # ids 4, 5, 6, 7
    # 34 frames over [10.0, 10.0 + 34/15): frame duration 1/15 s
    words = [["a", 10.0, 10.1], ["b", 10.0 + 5 / 15 + 0.01, 11], ["zzz", 10.0 + 20.5 / 15, 12], ["c", 9.5, 9.9], ["d", 10.0 + 34 / 15, 13]]
    ext = D.extend_word_seq(lang, words, 10.0, 10.0 + 34 / 15, 34)
    want = np.zeros(34, dtype=np.int64)
    want[0] = 6            # "c" starts before the clip: clamped to frame 0, written after "a" (later words win, :136)
    want[5] = 5
    want[20] = 3           # unknown word -> UNK
    assert ext.dtype == np.int64 and np.array_equal(ext, want)   # "d" starts at frame 34: dropped (:135)
    # remove_word_timing (:122-131): 4 words in range -> every int(34/5) = 6 frames, in word order
    ext2 = D.extend_word_seq(lang, words, 10.0, 10.0 + 34 / 15, 34, remove_word_timing=True)
    want2 = np.zeros(34, dtype=np.int64)
    want2[6], want2[12], want2[18], want2[24] = 4, 5, 3, 6
    assert np.array_equal(ext2, want2)
    assert np.array_equal(D.words_to_tensor(lang, words[:3], end_time=10.5), np.array([1, 4, 5, 2]))


def test_make_audio_fixed_length(pkg):
    D = _mods(pkg)
    a = np.arange(5, dtype=np.float32)
    assert np.array_equal(D.make_audio_fixed_length(a, 8), np.array([0, 1, 2, 3, 4, 4, 3, 2], dtype=np.float32))   # symmetric pad
    assert np.array_equal(D.make_audio_fixed_length(a, 3), a[:3])
    assert np.array_equal(D.make_audio_fixed_length(a, 5), a)


def test_getitem_matches_oracle_and_collate(pkg):
    D = _mods(pkg)
    lang = _lang(pkg, [f"w{i}" for i in range(50)])
    spk = pkg.Vocab.speakers(9)
    ds = D.SyntheticSpeechMotionDataset(24, lang, spk, seed=3)
    items = [ds[i] for i in range(len(ds))]
    for i, it in enumerate(items):
        ext, vec, audio = O.data_getitem(ds.raw(i), lang.get_word_index)
        assert it[1].dtype == torch.int64 and np.array_equal(it[1].numpy(), ext)
        assert np.array_equal(it[3].numpy(), vec) and np.array_equal(it[4].numpy(), audio)
        assert it[3].shape == (34, 27) and it[4].shape == (36267,) and it[2].shape == (34, 30)     # expected_audio_length :62
        assert int((it[1] > 0).sum()) >= 1 and it[0][0] == 1 and it[0][-1] == 2 and len(it) == 7
    text, vec, audio, vid = D.collate(items[:8], spk)
    assert text.shape == (8, 34) and vec.shape == (8, 34, 27) and audio.shape == (8, 36267) and vid.shape == (8,)
    assert vid.dtype == torch.int64 and int(vid.min()) >= 1 and int(vid.max()) < spk.n_words
    assert D.collate(items[:2], None)[3] is None and D.collate(items[:2], 1)[3] is None          # z_type random / none


def test_dataset_items_and_collate_match_reference_golden(pkg):
    """data.SpeechMotionDataset / sample_to_tensors / default_collate_fn == the reference's (lmdb_data_loader.py:43-53,107-171), bit for
    bit, on the raw samples the reference itself was run on."""
    D = _mods(pkg)
    g = load("g10_dataset.npz")
    lang = fixture_lang(pkg.Vocab, int(g["vocab_size"]))
    samples = dataset_samples(g)
    for tag, rwt in (("timed", False), ("rwt", True)):
        ds = D.SpeechMotionDataset(samples, 34, 10, 15, remove_word_timing=rwt)
        ds.set_lang_model(lang)
        assert len(ds) == len(samples) and ds.expected_audio_length == 36267 and ds.expected_spectrogram_length == 70
        items = [ds[i] for i in range(len(ds))]
        for i, (words, ext, pose, vec, audio, spec, aux) in enumerate(items):
            assert words.dtype == torch.int64 and ext.dtype == torch.int64 and vec.dtype == torch.float32 and audio.dtype == torch.float32
            assert np.array_equal(words.numpy(), g[f"{tag}{i}/words"]) and np.array_equal(ext.numpy(), g[f"{tag}{i}/ext"]), (tag, i)
            assert np.array_equal(pose.numpy(), g[f"item{i}/pose"]) and np.array_equal(vec.numpy(), g[f"item{i}/vec"])
            a = audio.numpy()
            assert a.shape[0] == int(g[f"item{i}/audio_len"])
            assert np.array_equal(a[:64], g[f"item{i}/audio_head"]) and np.array_equal(a[-1500:], g[f"item{i}/audio_tail"])
            assert abs(float(np.abs(a.astype(np.float64)).sum()) - float(g[f"item{i}/audio_abs_sum"])) < 1e-9
            assert list(spec.shape) == list(g[f"item{i}/spec_shape"]) and aux is samples[i][5]
        if not rwt:
            col = D.default_collate_fn(items[:4])
            assert len(col) == 8 and np.array_equal(col[0].numpy(), g["collate/word_seq"]) and np.array_equal(col[1].numpy(), g["collate/lengths"])
            assert np.array_equal(col[2].numpy(), g["collate/text"]) and np.array_equal(col[3].numpy(), g["collate/pose"])
            assert np.array_equal(col[4].numpy(), g["collate/vec"]) and list(col[5].shape) == list(g["collate/audio_shape"])
            assert torch.equal(col[5], torch.stack([it[4] for it in items[:4]])) and list(col[6].shape) == list(g["collate/spec_shape"])
            assert sorted(col[7]) == [str(k) for k in g["collate/aux_keys"]] and list(col[7]["vid"]) == [str(v) for v in g["collate/aux_vid"]]
            assert np.array_equal(col[7]["start_time"].numpy(), g["collate/aux_start_time"])
    a = g["fixlen/in"]
    for n, key in ((1500, "longer"), (700, "shorter"), (1000, "same")):
        assert np.array_equal(D.make_audio_fixed_length(a, n), g["fixlen/" + key])


def test_dataset_builds_speaker_model_when_none_given(pkg):
    """lmdb_data_loader.py:92-101,173-190: speaker_model None or 0 -> the dataset builds Vocab('vid') from the data (ids from 1, one per
    video id); train.py reads train_dataset.speaker_model to build the generator.  Checked on the reference's own g10 samples: collate()
    must then return a usable vid tensor (the z_type='speaker' engine asserts on it)."""
    D = _mods(pkg)
    g = load("g10_dataset.npz")
    samples = dataset_samples(g)
    vids = []
    for smp in samples:
        if smp[5]["vid"] not in vids:
            vids.append(smp[5]["vid"])
    for arg in (None, 0):
        ds = D.SpeechMotionDataset(samples, 34, 10, 15, speaker_model=arg)
        sm = ds.speaker_model
        assert isinstance(sm, pkg.Vocab) and sm.name == "vid" and sm.n_words == len(vids) + 1          # row 0 unused
        assert [sm.word2index[v] for v in vids] == list(range(1, len(vids) + 1))
        assert sum(sm.word2count.values()) == len(samples)
        ds.set_lang_model(fixture_lang(pkg.Vocab, int(g["vocab_size"])))
        text, vec, audio, vid = D.collate([ds[i] for i in range(min(4, len(ds)))], ds.speaker_model)
        assert vid is not None and vid.dtype == torch.int64 and int(vid.min()) >= 1 and int(vid.max()) < sm.n_words
    given = pkg.Vocab.speakers(5)
    assert D.SpeechMotionDataset(samples, 34, 10, 15, speaker_model=given).speaker_model is given


def test_window_inputs_and_fade_out_match_reference_golden(pkg):
    """synthesize.window_inputs / num_windows / fade_out_to_mean == what scripts/synthesize.py:generate_gestures fed its model and
    did to its output (recorded window inputs; fade-out applied to the reference's own un-faded result)."""
    syn = importlib.import_module(pkg.__name__ + ".synthesize")
    g = load("g9_generate_gestures.npz")
    lang = fixture_lang(pkg.Vocab, int(g["n_words"]))
    args = argparse.Namespace(n_poses=34, n_pre_poses=4, motion_resampling_framerate=15)
    cases = {str(c): synth_case(g, str(c)) for c in g["cases"]}
    for name, c in cases.items():
        n = c["win_text"].shape[0]
        assert syn.num_windows(len(c["audio"]) / 16000, 34, 4, 15) == n, name
        pad = 0
        for i in range(n):
            a, ids, pad = syn.window_inputs(args, lang, c["audio"], c["words"], i)
            assert ids.dtype == np.int64 and np.array_equal(ids, c["win_text"][i]), (name, i)
            check_window_audio(c, i, a)
        twin = cases.get(name[:-5]) if name.endswith("_fade") else None
        if twin is not None and (twin["seed_seq"] is None) == (c["seed_seq"] is None):
            plain = twin["out"]                            # same utterance, seed poses and draws without fade-out
            faded = syn.fade_out_to_mean(plain.copy(), pad, args)
            assert faded.shape == c["out"].shape and float(np.abs(faded - c["out"]).max()) < 1e-6, name
    c = cases["w2_exact"]                                  # no padding in the last window: the fade-out appends 2 * n_pre frames
    assert c["win_text"].shape[0] == 2 and c["out"].shape[0] == 2 * 30 + 4 + 8 and np.all(c["out"][-4:] == 0)
    assert syn.words_in_time_range([["a", 0.0, 0.5], ["b", 0.4, 1.0], ["c", 2.0, 3.0]], 0.5, 2.0) == [["b", 0.4, 1.0]]


def test_convert_dir_vec_to_pose_and_speaker_lookup(pkg):
    M = importlib.import_module(pkg.__name__ + ".eval_metrics")
    C = importlib.import_module(pkg.__name__ + ".checkpoint")
    g = load("g8_evaluate_testset.npz")
    v = g["dir_vec/in"]
    for x, key in ((v, "pose_b_t"), (v[0], "pose_t"), (v[0, 0], "pose_single")):
        p = M.convert_dir_vec_to_pose(x)
        assert p.shape == g["dir_vec/" + key].shape and float(np.abs(p - g["dir_vec/" + key]).max()) < 1e-12
    # utils/train_utils.py:152-164: Vocab -> itself; 1 (z_type random) / None -> None; DataParallel-style .module unwrap
    spk = pkg.Vocab.speakers(5)
    holder = argparse.Namespace
    assert C.get_speaker_model(holder(z_obj=spk)) is spk and C.get_speaker_model(holder(module=holder(z_obj=spk))) is spk
    assert C.get_speaker_model(holder(z_obj=1)) is None and C.get_speaker_model(holder(z_obj=None)) is None
    assert C.get_speaker_model(holder()) is None


def _assemble_reference(rec, B, n_poses, D_, A, remove_word_timing=False):
    """numpy restatement of tg_assemble_batch (csrc/assemble.hip) over the packed records: what the device kernel must produce."""
    text = np.zeros((B, n_poses), dtype=np.int64)
    audio = np.zeros((B, A), dtype=np.float32)
    vec = np.zeros((B, n_poses, D_), dtype=np.float32)
    for b in range(B):
        a0, a1 = int(rec["audio_off"][b]), int(rec["audio_off"][b + 1])
        n = a1 - a0
        i = np.arange(A)
        j = i % (2 * n)
        audio[b] = rec["audio"][a0:a1][np.where(j < n, j, 2 * n - 1 - j)]
        v0 = int(rec["vec_off"][b])
        vec[b] = rec["vec"][v0:v0 + n_poses * D_].reshape(n_poses, D_)
        start, end = float(rec["times"][b, 0]), float(rec["times"][b, 1])
        sample_end = start + (end - start) * n_poses / int(rec["n_ext"][b])
        fd = (sample_end - start) / n_poses
        nw = int(rec["n_words"][b])
        idxs = [max(0, int(np.floor((float(rec["word_onset"][b, w]) - start) / fd))) for w in range(nw)]
        if remove_word_timing:
            cnt = sum(1 for ix in idxs if ix < n_poses)
            space = int(n_poses / (cnt + 1))
            for k in range(cnt):
                text[b, (k + 1) * space] = rec["word_idx"][b, k]
        else:
            for w, ix in enumerate(idxs):
                if ix < n_poses:
                    text[b, ix] = rec["word_idx"][b, w]
    return text, audio, vec, np.asarray(rec["vid"]).copy()


def test_raw_record_packing_reproduces_getitem_and_collate(pkg):
    """data.RecordLayout.pack: stored samples -> raw records (the form data.DeviceRecordFeeder ships to the device).  The numpy restatement
    of the assembly kernel over those records equals SpeechMotionDataset.__getitem__ + default_collate_fn (lmdb_data_loader.py:43-53,
    107-171) bit for bit -- on the reference's own g10 samples and on synthetic clips with short, exact and long audio, words before the clip
    start and past its end, shared frames, and remove_word_timing."""
    D = _mods(pkg)
    g = load("g10_dataset.npz")
    samples = dataset_samples(g)
    lang = fixture_lang(pkg.Vocab, int(g["vocab_size"]))
    ds = D.SpeechMotionDataset(samples, 34, 10, 15)
    ds.set_lang_model(lang)
    items = [ds[i] for i in range(len(samples))]
    text, vec, audio, vid = D.collate(items, ds.speaker_model)
    L = D.RecordLayout(len(samples), 34, 27, 36267)
    host = L.views(np.zeros(L.nbytes, dtype=np.uint8))
    L.pack(samples, lang, ds.speaker_model, host)
    t2, a2, v2, s2 = _assemble_reference(host, len(samples), 34, 27, 36267)
    assert np.array_equal(t2, text.numpy()) and np.array_equal(a2, audio.numpy()) and np.array_equal(v2, vec.numpy()) and np.array_equal(s2, vid.numpy())
    # synthetic clips + hand-made edge cases
    lang2 = _lang(pkg, [f"w{i}" for i in range(50)])
    spk = pkg.Vocab.speakers(9)
    syn = D.SyntheticSpeechMotionDataset(13, lang2, spk, seed=5)
    raws = [list(syn.raw(i)) for i in range(13)]
    raws[0][3] = raws[0][3][:20000]                                 # short audio: symmetric padding, more than one reflection
    raws[1][3] = raws[1][3][:36267]                                 # exactly the expected length
    raws[2][3] = raws[2][3][:9000]                                  # very short: the padding wraps around several times
    st = raws[3][5]["start_time"]
    raws[3][0] = [["w1", st - 3.0, st - 2.9], ["w2", st + 0.01, st + 0.2], ["w3", st + 0.02, st + 0.3], ["nope", st + 1.0, st + 1.2], ["w4", st + 50.0, st + 51.0]]
    raws[4][0] = []                                                 # no words at all
    for rwt in (False, True):
        ref = [D.sample_to_tensors(r, lang2, 34, 15, remove_word_timing=rwt) for r in raws]
        text, vec, audio, vid = D.collate(ref, spk)
        L = D.RecordLayout(len(raws), 34, 27, 36267, w_max=16)
        host = L.views(np.zeros(L.nbytes, dtype=np.uint8))
        L.pack(raws, lang2, spk, host)
        assert int(host["audio_off"][-1]) <= len(raws) * 36267 and int(host["n_ext"][0]) == raws[0][2].shape[0]
        t2, a2, v2, s2 = _assemble_reference(host, len(raws), 34, 27, 36267, remove_word_timing=rwt)
        assert np.array_equal(t2, text.numpy()), rwt
        assert np.array_equal(a2, audio.numpy()) and np.array_equal(v2, vec.numpy()) and np.array_equal(s2, vid.numpy())
