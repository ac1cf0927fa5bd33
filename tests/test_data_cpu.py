"""Input-pipeline stand-in (data.py) against the oracle's restatement of SpeechMotionDataset.__getitem__ and hand-computed values.
The reference's data loader cannot be imported here (needs lmdb / pyarrow 0.14): parity for this part is unpinned and rests on the
hand-computed cases below."""
import importlib

import numpy as np
import torch

from oracle import ref_model as O


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
        assert int((it[1] > 0).sum()) >= 1 and it[0][0] == 1 and it[0][-1] == 2
    text, vec, audio, vid = D.collate(items[:8], spk)
    assert text.shape == (8, 34) and vec.shape == (8, 34, 27) and audio.shape == (8, 36267) and vid.shape == (8,)
    assert vid.dtype == torch.int64 and int(vid.min()) >= 1 and int(vid.max()) < spk.n_words
    assert D.collate(items[:2], None)[3] is None and D.collate(items[:2], 1)[3] is None          # z_type random / none
