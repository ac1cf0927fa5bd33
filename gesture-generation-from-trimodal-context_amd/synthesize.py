"""Long-utterance synthesis: the multimodal branch of scripts/synthesize.py:generate_gestures (:36-209).

An utterance is cut into 34-frame windows with a 30-frame stride; window i is seeded with the last 4 output frames of window
i-1 (:122-124) and its first 4 frames are cross-faded with them (:145-153).  Here the windows stay on the GPU: the seed
hand-over and the cross-fade are device ops (tg_window_blend), the host sees the result once at the end.  Many utterances run in
lock-step as one batch (the only serial dependency is window i-1 -> i of the same utterance), and with fixed shapes the
per-window forward is captured into a hipGraph (WindowDecoder).

TTS / Gentle alignment / LMDB front-ends of the reference script are network services and out of scope; `words` is the
reference's word list [[word, start_s, end_s], ...].
"""
import math
import random

import numpy as np
import torch

from . import ops, layers as L


def words_in_time_range(word_list, start_time, end_time):
    """data_loader/data_preprocessor.py:174-188."""
    out = []
    for w in word_list:
        if w[1] >= end_time:
            break
        if w[2] <= start_time:
            continue
        out.append(w)
    return out


def num_windows(clip_length, n_poses=34, n_pre_poses=4, fps=15):
    """synthesize.py:57-63."""
    unit, stride = n_poses / fps, (n_poses - n_pre_poses) / fps
    if clip_length < unit:
        return 1
    return math.ceil((clip_length - unit) / stride) + 1


def window_inputs(args, lang_model, audio, words, i, audio_sr=16000):
    """Audio slice (zero padded) and per-frame word ids of window i (synthesize.py:82-119).  Returns (audio (L,), ids (n_poses,),
    end_padding_samples)."""
    n_frames = args.n_poses
    unit_time = n_frames / args.motion_resampling_framerate
    stride_time = (n_frames - args.n_pre_poses) / args.motion_resampling_framerate
    clip_length = len(audio) / audio_sr
    audio_sample_length = int(unit_time * audio_sr)
    start_time = i * stride_time
    end_time = start_time + unit_time
    a0 = math.floor(start_time / clip_length * len(audio))
    piece = np.asarray(audio[a0:a0 + audio_sample_length], dtype=np.float32)
    pad = audio_sample_length - len(piece)
    if pad > 0:
        piece = np.pad(piece, (0, pad), "constant")
    ids = np.zeros(n_frames, dtype=np.int64)                      # 0 = PAD
    frame_duration = (end_time - start_time) / n_frames
    for w in words_in_time_range(words, start_time, end_time):
        idx = max(0, int(np.floor((w[1] - start_time) / frame_duration)))
        ids[idx] = lang_model.get_word_index(w[0])
    return piece, ids, max(pad, 0)


class WindowDecoder:
    """Batched window forward with device-side seed hand-over and cross-fade; optional hipGraph capture."""

    def __init__(self, args, pose_decoder, batch, device, graph=True, replay_draws=False):
        self.args, self.gen, self.B, self.dev = args, pose_decoder, batch, device
        # replay_draws (parity tests): the per-window eps / random z comes from self.draw (B, 16), filled by the caller before each
        # window, instead of the device RNG -- a static buffer, so a captured window replays with new contents
        self.draw = torch.zeros(batch, 16, device=device) if replay_draws else None
        self.T, self.n_pre = args.n_poses, args.n_pre_poses
        self.D = pose_decoder.pose_dim
        self.audio_len = int(self.T / args.motion_resampling_framerate * 16000)
        self.pre_seq = torch.zeros(batch, self.T, self.D + 1, device=device)
        self.text = torch.zeros(batch, self.T, dtype=torch.int64, device=device)
        self.audio = torch.zeros(batch, self.audio_len, device=device)
        self.vid = torch.zeros(batch, dtype=torch.int64, device=device)
        self.out = torch.zeros(batch, self.T, self.D, device=device)
        self.tail = torch.zeros(batch, self.n_pre, self.D, device=device)
        self.seedwin = torch.zeros(batch, self.T, self.D, device=device)
        self.use_graph, self.graph = graph, None
        # the generator's weights stand still for the lifetime of a decoder (one synthesis call): weight-only operands are formed by the first
        # window and kept (layers.FrozenWeights) -- a decoder must not outlive a change of the parameters
        self.frozen = L.FrozenWeights()
        pose_decoder.train(False)

    def seed(self, seed_seq=None):
        """pre_seq of the first window: optional seed poses with the constraint bit (synthesize.py:46-50)."""
        self.pre_seq.zero_()
        if seed_seq is not None:
            s = torch.as_tensor(seed_seq, dtype=torch.float32, device=self.dev)
            self.pre_seq[:, :self.n_pre, :-1] = s[..., :self.n_pre, :]
            self.pre_seq[:, :self.n_pre, -1] = 1

    def _forward(self, first):
        eng = self.gen.engine
        eng.rng.advance()                      # reparameterize() draws a fresh eps per window, also at inference
        inject = None if self.draw is None else {"g.eps": self.draw, "g.z": self.draw}
        vid = self.vid if eng.z_mode == "speaker" else None            # synthesize.py:67-74: no speaker input otherwise
        res = eng.forward(self.pre_seq, self.text, self.audio, vid, training=False, inject=inject)
        ops.copy2d(res["out"].view(self.B * self.T, self.D), self.out.view(self.B * self.T, self.D))
        if not first:                                            # cross-fade with the previous window's last frames
            ops.window_blend(self.tail, self.out)
        # hand-over: the last n_pre frames are kept for the next window's cross-fade (:146-147) and seed it (:122-124)
        self.tail.copy_(self.out[:, self.T - self.n_pre:, :])          # strided device copy: data movement only
        self.seedwin[:, :self.n_pre, :].copy_(self.tail)
        ops.make_pre_seq(self.seedwin, self.pre_seq, self.n_pre)          # frames < n_pre + constraint bit, zeros elsewhere

    def window(self, in_text, in_audio, vid, first, draw=None):
        """One window for the whole batch.  Inputs may be CPU or GPU tensors; returns the (B, T, D) output buffer (device,
        overwritten by the next call)."""
        self.text.copy_(in_text, non_blocking=True)
        self.audio.copy_(in_audio, non_blocking=True)
        if vid is not None:
            self.vid.copy_(vid, non_blocking=True)
        if draw is not None:
            self.draw.copy_(draw, non_blocking=True)
        with torch.no_grad(), self.frozen:
            if not self.use_graph or first:
                self._forward(first)
            else:
                if self.graph is None:
                    torch.cuda.synchronize()
                    self.graph = torch.cuda.CUDAGraph()
                    keep = [t.clone() for t in (self.pre_seq, self.tail, self.out)]
                    pg_alive = torch.distributed.is_available() and torch.distributed.is_initialized()     # RCCL helper threads: see train_gan.GraphedGanStep
                    with torch.cuda.graph(self.graph, capture_error_mode="thread_local" if pg_alive else "global"):
                        self._forward(False)
                    for t, k in zip((self.pre_seq, self.tail, self.out), keep):   # capture does not execute: restore state
                        t.copy_(k)
                self.graph.replay()
        return self.out


def generate_gestures_batch(args, pose_decoder, lang_model, audios, words_list, vids=None, seed_seqs=None, audio_sr=16000,
                            graph=True, _draws=None):
    """Lock-step synthesis of several utterances.  Returns a list of (n_i * 30 + 4, D) numpy arrays (mean-subtracted direction
    vectors, like the reference's return value without fade-out).  vids: one speaker id per utterance, or None / a falsy entry to
    draw it like the reference (synthesize.py:67-74; ignored unless args.z_type == 'speaker').  _draws (parity tests): per window,
    the (B, 16) eps / z to replay instead of the device RNG."""
    dev = next(pose_decoder.parameters()).device
    B = len(audios)
    n_win = [num_windows(len(a) / audio_sr, args.n_poses, args.n_pre_poses, args.motion_resampling_framerate) for a in audios]
    dec = WindowDecoder(args, pose_decoder, B, dev, graph=graph, replay_draws=_draws is not None)
    dec.seed(None if seed_seqs is None else np.stack([np.asarray(s)[:args.n_pre_poses] for s in seed_seqs]))
    vid = None
    if args.z_type == "speaker":                                                       # synthesize.py:67-74
        vids = [None] * B if vids is None else list(vids)
        for b in range(B):
            if not vids[b]:
                vids[b] = random.randrange(pose_decoder.z_obj.n_words)
        vid = torch.as_tensor(vids, dtype=torch.int64)
    stride = args.n_poses - args.n_pre_poses
    total = torch.zeros(B, max(n_win) * stride + args.n_pre_poses, dec.D, device=dev)
    for i in range(max(n_win)):
        a_np, t_np = [], []
        for b in range(B):
            j = min(i, n_win[b] - 1)                               # finished utterances idle on their last window
            a, ids, _ = window_inputs(args, lang_model, audios[b], words_list[b], j, audio_sr)
            a_np.append(a); t_np.append(ids)
        out = dec.window(torch.from_numpy(np.stack(t_np)), torch.from_numpy(np.stack(a_np)), vid, first=(i == 0),
                         draw=None if _draws is None else _draws[i])
        # out_list[-1][:-n_pre] + blended window == write the whole window at frame i*stride (its first n_pre frames overwrite
        # the previous window's last n_pre frames with the cross-faded values)
        total[:, i * stride:i * stride + args.n_poses, :].copy_(out)
    res = total.cpu().numpy()
    return [res[b, :n_win[b] * stride + args.n_pre_poses] for b in range(B)]


def fade_out_to_mean(out_dir_vec, end_padding_samples, args, audio_sr=16000):
    """synthesize.py:188-207: fade out to the mean pose over 2 * n_pre_poses frames starting where the real audio ended --
    frames after the fade are zeroed (mean pose), the transition is a weighted quadratic fit per dimension.  Host maths on a
    handful of frames."""
    n_smooth = args.n_pre_poses
    start_frame = len(out_dir_vec) - int(end_padding_samples / audio_sr * args.motion_resampling_framerate)
    end_frame = start_frame + n_smooth * 2
    if len(out_dir_vec) < end_frame:
        out_dir_vec = np.pad(out_dir_vec, [(0, end_frame - len(out_dir_vec)), (0, 0)], mode="constant")
    out_dir_vec[end_frame - n_smooth:] = 0
    y = out_dir_vec[start_frame:end_frame]
    x = np.arange(y.shape[0])
    w = np.ones(len(y)); w[0] = 5; w[-1] = 5
    coeffs = np.polyfit(x, y, 2, w=w)
    out_dir_vec[start_frame:end_frame] = np.stack([np.poly1d(coeffs[:, k])(x) for k in range(y.shape[1])], axis=1)
    return out_dir_vec


def generate_gestures(args, pose_decoder, lang_model, audio, words, audio_sr=16000, vid=None, seed_seq=None, fade_out=False,
                      _draws=None):
    """Single-utterance API of the reference (synthesize.py:36-209, multimodal_context model)."""
    out = generate_gestures_batch(args, pose_decoder, lang_model, [audio], [words], [vid], None if seed_seq is None else [seed_seq],
                                  audio_sr, graph=True, _draws=_draws)[0]
    if not fade_out:
        return out
    n_win = num_windows(len(audio) / audio_sr, args.n_poses, args.n_pre_poses, args.motion_resampling_framerate)
    _, _, end_padding = window_inputs(args, lang_model, audio, words, n_win - 1, audio_sr)
    return fade_out_to_mean(out, end_padding, args, audio_sr)
