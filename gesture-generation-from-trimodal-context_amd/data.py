"""Input-pipeline stand-in: sample -> training tensors -> device batch (data_loader/lmdb_data_loader.py:107-171, :43-53;
utils/data_utils.py:68-74; train.py:169-183).

The reference reads pyarrow-0.14-serialised samples from LMDB (neither is available here); what follows that read
is restated: a *sample* is the tuple the reference deserialises, `(word_seq, pose_seq, vec_seq, audio, spectrogram, aux_info)`
with word_seq = [[word, start_s, end_s], ...] and aux_info = {'vid', 'start_time', 'end_time', ...}.  `sample_to_tensors` is
`SpeechMotionDataset.__getitem__` after the read; `collate` is `default_collate_fn` restricted to what the multimodal model
consumes; `DeviceBatchFeeder` replaces the `.to(device)` calls of the training loop with pinned staging buffers and asynchronous
copies into static device tensors on a copy stream, double-buffered, so that the host-to-device transfer of batch i+1 overlaps
iteration i of a hipGraph-replayed step (GraphedGanStep reads the same static tensors every replay).
`DeviceRecordFeeder` goes one step further (round 4): the host only PACKS the stored samples' raw arrays (audio as stored, the first
n_poses direction-vector frames, (word index, onset) pairs, times, speaker index) into one pinned buffer; the per-sample assembly of
`__getitem__` -- extend_word_seq, make_audio_fixed_length, the slicing -- and the collate run on the device in one launch
(ops.assemble_batch -> tg_assemble_batch, csrc/assemble.hip), writing the step's static inputs in place.
"""
import numpy as np
import torch


def make_audio_fixed_length(audio, expected_audio_length):
    """utils/data_utils.py:68-74: symmetric padding or truncation."""
    n_padding = expected_audio_length - len(audio)
    if n_padding > 0:
        return np.pad(audio, (0, n_padding), mode="symmetric")
    return audio[0:expected_audio_length]


def extend_word_seq(lang, words, start_time, end_time, n_frames, remove_word_timing=False):
    """lmdb_data_loader.py:115-140: one vocabulary index per pose frame at each word's onset frame, 0 (PAD) elsewhere."""
    frame_duration = (end_time - start_time) / n_frames
    out = np.zeros(n_frames, dtype=np.int64)
    if remove_word_timing:
        n_words = 0
        for word in words:
            idx = max(0, int(np.floor((word[1] - start_time) / frame_duration)))
            if idx < n_frames:
                n_words += 1
        space = int(n_frames / (n_words + 1))
        for i in range(n_words):
            out[(i + 1) * space] = lang.get_word_index(words[i][0])
    else:
        for word in words:
            idx = max(0, int(np.floor((word[1] - start_time) / frame_duration)))
            if idx < n_frames:
                out[idx] = lang.get_word_index(word[0])
    return out


def words_to_tensor(lang, words, end_time=None):
    """lmdb_data_loader.py:142-149: [SOS, w..., EOS] (used by the seq2seq baseline; kept for the collate's return arity)."""
    indexes = [lang.SOS_token]
    for word in words:
        if end_time is not None and word[1] > end_time:
            break
        indexes.append(lang.get_word_index(word[0]))
    indexes.append(lang.EOS_token)
    return np.asarray(indexes, dtype=np.int64)


def calc_spectrogram_length_from_motion_length(n_frames, fps):
    """utils/data_utils.py:44-46."""
    return int(round((n_frames / fps * 16000 - 1024) / 512 + 1))


def sample_to_tensors(sample, lang_model, n_poses, pose_resampling_fps, remove_word_timing=False):
    """SpeechMotionDataset.__getitem__ after the LMDB read (:151-171).  Returns the reference's tuple
    (word_seq (L,) int64, extended_word_seq (n_poses,) int64, pose_seq (n_poses, 30) f32, vec_seq (n_poses, 27) f32,
    audio (expected_audio_length,) f32, spectrogram (mels, expected_spectrogram_length), aux_info)."""
    word_seq, pose_seq, vec_seq, audio, spectrogram, aux_info = sample
    expected_audio_length = int(round(n_poses / pose_resampling_fps * 16000))
    duration = aux_info["end_time"] - aux_info["start_time"]
    sample_end_time = aux_info["start_time"] + duration * n_poses / vec_seq.shape[0]
    audio = make_audio_fixed_length(np.asarray(audio), expected_audio_length)
    spectrogram = np.asarray(spectrogram)[:, 0:calc_spectrogram_length_from_motion_length(n_poses, pose_resampling_fps)]
    vec_seq = np.asarray(vec_seq)[0:n_poses]
    pose_seq = np.asarray(pose_seq)[0:n_poses]
    words = words_to_tensor(lang_model, word_seq, sample_end_time)
    ext = extend_word_seq(lang_model, word_seq, aux_info["start_time"], sample_end_time, n_poses, remove_word_timing)
    return (torch.from_numpy(words), torch.from_numpy(ext), torch.from_numpy(pose_seq.reshape(pose_seq.shape[0], -1)).float(),
            torch.from_numpy(vec_seq.reshape(vec_seq.shape[0], -1)).float(), torch.from_numpy(np.ascontiguousarray(audio)).float(),
            torch.from_numpy(np.ascontiguousarray(spectrogram)), aux_info)


def collate(items, speaker_model=None):
    """default_collate_fn (:43-53) restricted to what the multimodal model consumes + the speaker lookup of train.py:178-183.
    Returns (in_text_padded (B, T) int64, target_vec (B, T, 27) f32, in_audio (B, A) f32, vid_indices (B,) int64 or None)."""
    text = torch.stack([it[1] for it in items])
    vec = torch.stack([it[3] for it in items])
    audio = torch.stack([it[4] for it in items])
    vid = None
    if speaker_model is not None and hasattr(speaker_model, "word2index"):
        vid = torch.tensor([speaker_model.word2index[it[6]["vid"]] for it in items], dtype=torch.int64)
    return text, vec, audio, vid


def default_collate_fn(data):
    """lmdb_data_loader.py:43-53 with the reference's return arity, for code written against its DataLoader (evaluate_testset):
    (word_seq placeholder, lengths placeholder, text_padded, pose_seq, vec_seq, audio, spectrogram, aux_info)."""
    from torch.utils.data.dataloader import default_collate
    _, text_padded, pose_seq, vec_seq, audio, spectrogram, aux_info = zip(*data)
    aux_info = {key: default_collate([d[key] for d in aux_info]) for key in aux_info[0]}
    return (torch.tensor([0]), torch.tensor([0]), default_collate(text_padded), default_collate(pose_seq), default_collate(vec_seq),
            default_collate(audio), default_collate(spectrogram), aux_info)


collate_reference = default_collate_fn


class SpeechMotionDataset(torch.utils.data.Dataset):
    """SpeechMotionDataset (lmdb_data_loader.py:56-171) over samples already read into memory, in the stored format
    [word_seq, pose_seq, vec_seq, audio, spectrogram, aux_info] (data_preprocessor.py:160-164).  The LMDB / pyarrow-0.14 read
    itself is out of scope (neither library exists in this image)."""

    def __init__(self, samples, n_poses, subdivision_stride, pose_resampling_fps, mean_pose=None, mean_dir_vec=None,
                 speaker_model=None, remove_word_timing=False):
        self.samples, self.n_poses, self.subdivision_stride = samples, n_poses, subdivision_stride
        self.skeleton_resampling_fps, self.mean_dir_vec, self.remove_word_timing = pose_resampling_fps, mean_dir_vec, remove_word_timing
        self.expected_audio_length = int(round(n_poses / pose_resampling_fps * 16000))
        self.expected_spectrogram_length = calc_spectrogram_length_from_motion_length(n_poses, pose_resampling_fps)
        self.lang_model = None
        self.n_samples = len(samples)
        # lmdb_data_loader.py:92-101: no speaker model given (None or 0) -> build one from the data; train.py then reads
        # train_dataset.speaker_model to size the generator's speaker embedding and passes it on to the validation set
        self.speaker_model = self._make_speaker_model(samples) if (speaker_model is None or speaker_model == 0) else speaker_model

    @staticmethod
    def _make_speaker_model(samples):
        """_make_speaker_model (lmdb_data_loader.py:173-190): Vocab('vid') without default tokens (ids start at 1), one entry per video
        id, in the order the ids are first met (the reference walks the raw videos of the LMDB; the clips here carry their video's id in
        aux_info['vid'], so first-appearance order over the clips is the same walk)."""
        from .vocab import Vocab
        model = Vocab("vid", insert_default_tokens=False)
        for smp in samples:
            model.index_word(smp[5]["vid"])
        return model

    def __len__(self):
        return self.n_samples

    def set_lang_model(self, lang_model):
        self.lang_model = lang_model

    def __getitem__(self, idx):
        return sample_to_tensors(self.samples[idx], self.lang_model, self.n_poses, self.skeleton_resampling_fps, self.remove_word_timing)


class SyntheticSpeechMotionDataset(torch.utils.data.Dataset):
    """Samples in the reference's stored format, generated deterministically: random-walk direction vectors, band-limited noise
    audio with a random (slightly wrong) length, 4-12 words with onset times inside the clip, one of `n_speakers - 1` videos."""

    def __init__(self, n_samples, lang_model, speaker_model, n_poses=34, fps=15, seed=0, mean_dir_vec=None):
        self.n, self.lang, self.spk = n_samples, lang_model, speaker_model
        self.n_poses, self.fps, self.seed = n_poses, fps, seed
        self.words = [w for w in lang_model.word2index]
        self.vids = list(speaker_model.word2index)

    def __len__(self):
        return self.n

    def raw(self, idx):
        r = np.random.RandomState(self.seed * 1000003 + idx)
        n_ext = int(round(self.n_poses * 1.25))                        # the preprocessor stores 25 % margin (:84)
        vec = np.cumsum(r.randn(n_ext, 9, 3).astype(np.float32) * 0.02, axis=0)
        vec /= np.maximum(np.linalg.norm(vec, axis=-1, keepdims=True), 1e-6)
        vec = (vec - vec.mean(0, keepdims=True)).astype(np.float32)
        pose = np.cumsum(r.randn(n_ext, 10, 3).astype(np.float32) * 0.01, axis=0)
        start = float(r.uniform(0, 100))
        dur = n_ext / self.fps
        n_audio = int(dur * 16000) + int(r.randint(-200, 200))
        audio = (0.1 * r.randn(n_audio)).astype(np.float32)
        n_words = int(r.randint(4, 13))
        onsets = np.sort(r.uniform(start, start + dur, n_words))
        words = [[self.words[int(r.randint(len(self.words)))], float(t0), float(t0 + 0.2)] for t0 in onsets]
        aux = {"vid": self.vids[int(r.randint(len(self.vids)))], "start_time": start, "end_time": start + dur,
               "start_frame_no": 0, "end_frame_no": n_ext}
        return words, pose, vec, audio, np.zeros((128, 1), np.float32), aux

    def __getitem__(self, idx):
        return sample_to_tensors(self.raw(idx), self.lang, self.n_poses, self.fps)


def packed_like(tensors, device=None, pin=False):
    """One flat byte buffer holding a copy-shaped twin of every tensor (256-byte aligned pieces) -> (flat uint8 tensor, views).  A batch kept
    this way moves host -> device or device -> device as ONE copy instead of one per tensor."""
    offs, off = [], 0
    for t in tensors:
        offs.append(off)
        off += (t.numel() * t.element_size() + 255) // 256 * 256
    flat = torch.empty(off, dtype=torch.uint8, device="cpu" if pin else (device or tensors[0].device))
    if pin:
        flat = flat.pin_memory()
    views = tuple(flat[o:o + t.numel() * t.element_size()].view(t.dtype).view(t.shape) for o, t in zip(offs, tensors))
    return flat, views


class DeviceBatchFeeder:
    """Host -> device staging for a static-shape training step (the `.to(device)` calls of train.py:172-183).

    `static` are the device tensors a GraphedGanStep was captured on (GraphedGanStep.static); with `static_flat` (GraphedGanStep.static_flat,
    the one buffer they are views of) every move below is a single copy.  put(batch) copies a collated batch into a free pinned host
    slot (two slots) and enqueues its asynchronous host-to-device copy; ready() makes the batch the step's input.  Two modes:
      * overlap=True (default when static_flat is given): the host -> device copy (19 MB at B = 128) runs on a separate copy stream into a
        device staging slot while the current iteration runs; ready() moves staging -> static with ONE device-to-device copy on the
        compute stream (microseconds), so the iteration never waits for PCIe;
      * overlap=False: the copy goes straight into the static tensors on the compute stream, stream-ordered behind the previous replay:
        +0.4 ms per iteration at B = 128 (19 MB at ~50 GiB/s)."""

    def __init__(self, static_text, static_audio, static_target, static_vid, overlap=None, static_flat=None):
        self.static = (static_text, static_audio, static_target, static_vid)
        self.dev = static_text.device
        self.flat = static_flat
        if static_flat is not None:
            lo = static_flat.data_ptr()
            assert all(lo <= t.data_ptr() < lo + static_flat.numel() for t in self.static), "static tensors must be views of static_flat"
        self.overlap = (static_flat is not None) if overlap is None else bool(overlap)
        self.h2d_done = [None, None]          # per slot: host -> device copy finished (pinned slot reusable)
        self.i = 0
        self.pending = None
        if self.flat is not None:
            offs = [t.data_ptr() - self.flat.data_ptr() for t in self.static]
            carve = lambda f: tuple(f[o:o + t.numel() * t.element_size()].view(t.dtype).view(t.shape) for o, t in zip(offs, self.static))
            self.pinned_flat = [torch.empty(self.flat.numel(), dtype=torch.uint8).pin_memory() for _ in range(2)]
            self.pinned = [carve(f) for f in self.pinned_flat]
        else:
            self.pinned = [tuple(torch.empty(t.shape, dtype=t.dtype).pin_memory() for t in self.static) for _ in range(2)]
        if self.overlap:
            if self.flat is not None:
                self.staging_flat = [torch.empty_like(self.flat) for _ in range(2)]
            else:
                self.staging = [tuple(torch.empty_like(t) for t in self.static) for _ in range(2)]
            self.stream = torch.cuda.Stream(device=self.dev)
            self.d2d_done = [None, None]      # per slot: staging -> static copy finished (staging slot reusable)

    def put(self, text, vec, audio, vid):
        slot = self.i
        self.i ^= 1
        if self.h2d_done[slot] is not None:
            self.h2d_done[slot].synchronize()                 # never overwrite pinned memory a copy may still be reading
        for dst, src in zip(self.pinned[slot], (text, audio, vec, vid)):
            # plain single-threaded memcpy: torch's multi-threaded CPU copy stalled for tens of ms now and then on a many-core host
            np.copyto(dst.numpy(), src.numpy() if isinstance(src, torch.Tensor) else np.asarray(src))
        if self.overlap:
            with torch.cuda.stream(self.stream):
                if self.d2d_done[slot] is not None:
                    self.stream.wait_event(self.d2d_done[slot])   # the staging slot has been drained into the static tensors
                if self.flat is not None:
                    self.staging_flat[slot].copy_(self.pinned_flat[slot], non_blocking=True)
                else:
                    for dst, src in zip(self.staging[slot], self.pinned[slot]):
                        dst.copy_(src, non_blocking=True)
                ev = torch.cuda.Event()
                ev.record(self.stream)
                self.h2d_done[slot] = ev
        self.pending = slot

    def ready(self):
        """Compute stream: make the newest batch the step's input (call right before replaying the step)."""
        slot = self.pending
        cur = torch.cuda.current_stream(self.dev)
        if self.overlap:
            cur.wait_event(self.h2d_done[slot])
            if self.flat is not None:
                self.flat.copy_(self.staging_flat[slot], non_blocking=True)        # one device-to-device copy
            else:
                for dst, src in zip(self.static, self.staging[slot]):
                    dst.copy_(src, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(cur)
            self.d2d_done[slot] = ev
        else:
            if self.flat is not None:
                self.flat.copy_(self.pinned_flat[slot], non_blocking=True)
            else:
                for dst, src in zip(self.static, self.pinned[slot]):
                    dst.copy_(src, non_blocking=True)              # stream-ordered behind the previous replay's reads
            ev = torch.cuda.Event()
            ev.record(cur)
            self.h2d_done[slot] = ev


# ------------------------------------------------------------------------------------------------------------------ raw records
class RecordLayout:
    """Fixed-capacity layout of one batch of RAW sample records inside a flat byte buffer (pinned host slot and its device twin share
    it): audio [B * audio_len] f32 (a clip ships min(len, audio_len) samples -- what lies behind audio_len is truncated by
    make_audio_fixed_length anyway), audio_off [B + 1] i64, vec [B * n_poses * D] f32, vec_off [B + 1] i64, word_idx / word_onset
    [B][w_max], n_words / n_ext [B] i32, times [B][2] f64, vid [B] i64."""
    FIELDS = (("audio", "float32"), ("audio_off", "int64"), ("vec", "float32"), ("vec_off", "int64"), ("word_idx", "int64"),
              ("word_onset", "float64"), ("times", "float64"), ("vid", "int64"), ("n_words", "int32"), ("n_ext", "int32"))

    def __init__(self, batch, n_poses, pose_floats, audio_len, w_max=64):
        self.B, self.n_poses, self.D, self.A, self.w_max = batch, n_poses, pose_floats, audio_len, w_max
        shapes = {"audio": (batch * audio_len,), "audio_off": (batch + 1,), "vec": (batch * n_poses * pose_floats,), "vec_off": (batch + 1,),
                  "word_idx": (batch, w_max), "word_onset": (batch, w_max), "times": (batch, 2), "vid": (batch,), "n_words": (batch,),
                  "n_ext": (batch,)}
        self.items, off = [], 0
        for name, dt in self.FIELDS:
            n = int(np.prod(shapes[name])) * np.dtype(dt).itemsize
            self.items.append((name, dt, shapes[name], off, n))
            off += (n + 255) // 256 * 256
        self.nbytes = off

    def views(self, flat):
        """name -> typed view of a flat uint8 buffer (numpy array for host buffers given as numpy, torch tensor otherwise)."""
        out = {}
        for name, dt, shape, off, n in self.items:
            piece = flat[off:off + n]
            out[name] = piece.view(dt).reshape(shape) if isinstance(flat, np.ndarray) else piece.view(getattr(torch, dt)).view(shape)
        return out

    def pack(self, samples, lang_model, speaker_model, host):
        """Raw stored samples [word_seq, pose_seq, vec_seq, audio, spectrogram, aux_info] -> the views `host` of a pinned slot.  Host work per
        clip: two memcpys and the dictionary look-ups (word -> index, video id -> speaker index); no padding, no per-frame loop, no stacking."""
        assert len(samples) == self.B
        a_off = v_off = 0
        for b, smp in enumerate(samples):
            word_seq, _, vec_seq, audio, _, aux = smp
            audio = np.asarray(audio, dtype=np.float32).reshape(-1)
            na = min(audio.shape[0], self.A)
            host["audio_off"][b] = a_off
            host["audio"][a_off:a_off + na] = audio[:na]
            a_off += na
            vec = np.asarray(vec_seq, dtype=np.float32)
            assert vec.shape[0] >= self.n_poses, "a stored sample holds at least n_poses frames (data_preprocessor.py:84)"
            nv = self.n_poses * self.D
            host["vec_off"][b] = v_off
            host["vec"][v_off:v_off + nv] = vec[:self.n_poses].reshape(-1)
            v_off += nv
            nw = len(word_seq)
            assert nw <= self.w_max, f"{nw} words in one clip: raise RecordLayout w_max"
            host["n_words"][b], host["n_ext"][b] = nw, vec.shape[0]
            for w, word in enumerate(word_seq):
                host["word_idx"][b, w] = lang_model.get_word_index(word[0])
                host["word_onset"][b, w] = word[1]
            host["times"][b, 0], host["times"][b, 1] = aux["start_time"], aux["end_time"]
            host["vid"][b] = speaker_model.word2index[aux["vid"]] if speaker_model is not None and hasattr(speaker_model, "word2index") else 0
        host["audio_off"][self.B], host["vec_off"][self.B] = a_off, v_off


class DeviceRecordFeeder:
    """DeviceBatchFeeder for RAW records: put(samples) packs a batch of stored samples into a pinned slot (RecordLayout.pack) and enqueues
    ONE host-to-device copy on a copy stream; ready() runs the assembly kernel on the compute stream, which writes the captured step's
    static inputs (in_text, in_audio, target, vid) in place -- the role of SpeechMotionDataset.__getitem__ + default_collate_fn + the
    `.to(device)` calls of train.py:169-183, with the per-sample work on the device."""

    def __init__(self, static_text, static_audio, static_target, static_vid, lang_model, speaker_model, w_max=64, remove_word_timing=False):
        self.static = (static_text, static_audio, static_target, static_vid)
        self.dev = static_text.device
        B, n_poses = static_text.shape
        self.layout = RecordLayout(B, n_poses, static_target.shape[2], static_audio.shape[1], w_max)
        self.lang, self.spk, self.rwt = lang_model, speaker_model, remove_word_timing
        self.pinned = [torch.empty(self.layout.nbytes, dtype=torch.uint8).pin_memory() for _ in range(2)]
        self.host = [self.layout.views(p.numpy()) for p in self.pinned]
        self.devbuf = [torch.empty(self.layout.nbytes, dtype=torch.uint8, device=self.dev) for _ in range(2)]
        self.rec = [self.layout.views(d) for d in self.devbuf]
        self.stream = torch.cuda.Stream(device=self.dev)
        self.h2d_done, self.used = [None, None], [None, None]
        self.i, self.pending = 0, None

    def put(self, samples):
        slot = self.i
        self.i ^= 1
        if self.h2d_done[slot] is not None:
            self.h2d_done[slot].synchronize()                 # the pinned slot is free again
        self.layout.pack(samples, self.lang, self.spk, self.host[slot])
        with torch.cuda.stream(self.stream):
            if self.used[slot] is not None:
                self.stream.wait_event(self.used[slot])       # the assembly kernel that read this device slot has run
            self.devbuf[slot].copy_(self.pinned[slot], non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(self.stream)
            self.h2d_done[slot] = ev
        self.pending = slot

    def ready(self):
        from . import ops
        slot = self.pending
        cur = torch.cuda.current_stream(self.dev)
        cur.wait_event(self.h2d_done[slot])
        ops.assemble_batch(self.rec[slot], *self.static, remove_word_timing=self.rwt)
        ev = torch.cuda.Event()
        ev.record(cur)
        self.used[slot] = ev
