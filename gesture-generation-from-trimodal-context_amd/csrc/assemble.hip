// Batch assembly on the device: what SpeechMotionDataset.__getitem__ does to a stored sample after the LMDB read
// (data_loader/lmdb_data_loader.py:107-171) plus default_collate_fn's stacking (:43-53), for a whole batch in one launch.
// The host ships RAW per-clip records in one buffer -- the audio as stored (any length), the direction vectors as stored (n_ext >= n_poses
// frames; or their first n_poses frames plus the stored frame count), the words as (vocabulary index, onset time) pairs, start / end time, the speaker index -- and this kernel writes the training
// step's input tensors in place:
//   in_text [B][n_poses]       extend_word_seq (:115-140): one index per pose frame at each word's onset frame, 0 (PAD) elsewhere; fp64
//                              arithmetic exactly as the reference's Python / numpy doubles, words applied in order (a later word on the
//                              same frame overwrites an earlier one); remove_word_timing spreads the words evenly instead
//   in_audio [B][audio_len]    utils/data_utils.py:68-74 make_audio_fixed_length: truncate, or pad with numpy's mode='symmetric'
//   target [B][n_poses][D]     vec_seq[0:n_poses] (:157), flattened
//   vid [B]                    copied
// Bit-exact against the host path (data.sample_to_tensors + collate, pinned to the reference by the g10 fixture).
#include "common.hpp"

namespace tg {

__global__ __launch_bounds__(256) void assemble_batch_kernel(
    const float* __restrict__ audio_raw, const int64_t* __restrict__ audio_off, const float* __restrict__ vec_raw,
    const int64_t* __restrict__ vec_off, const int64_t* __restrict__ word_idx, const double* __restrict__ word_onset,
    const int32_t* __restrict__ n_words, const double* __restrict__ times, const int32_t* __restrict__ n_ext_in,
    const int64_t* __restrict__ vid_in, int Wmax, int n_poses,
    int pose_floats, int audio_len, int remove_word_timing, int64_t* __restrict__ out_text, float* __restrict__ out_audio,
    float* __restrict__ out_vec, int64_t* __restrict__ out_vid) {
    const int b = blockIdx.x;
    // ---- audio: every block of the clip's row takes a 1 024-sample slice
    const long a0 = audio_off[b], n = audio_off[b + 1] - a0;
    const float* src = audio_raw + a0;
    float* dst = out_audio + (long)b * audio_len;
    for (int i = blockIdx.y * 1024 + threadIdx.x; i < min(audio_len, (int)(blockIdx.y + 1) * 1024); i += 256) {
        float v = 0.f;
        if (n > 0) {
            // numpy.pad(mode='symmetric'): the signal mirrored about its edge, edge sample repeated, with period 2 n
            const long j = i % (2 * n);
            v = src[j < n ? j : 2 * n - 1 - j];
        }
        dst[i] = v;
    }
    if (blockIdx.y != 0) return;
    // ---- direction vectors: the first n_poses frames
    const float* vs = vec_raw + vec_off[b];
    const int nv = n_poses * pose_floats;
    for (int i = threadIdx.x; i < nv; i += 256) out_vec[(long)b * nv + i] = vs[i];
    // ---- words -> one index per frame (sequential: word order decides who keeps a shared frame)
    for (int i = threadIdx.x; i < n_poses; i += 256) out_text[(long)b * n_poses + i] = 0;
    __syncthreads();
    if (threadIdx.x == 0) {
        out_vid[b] = vid_in ? vid_in[b] : 0;
        const double start = times[2 * b], end = times[2 * b + 1];
        const long n_ext = n_ext_in ? (long)n_ext_in[b] : (vec_off[b + 1] - vec_off[b]) / pose_floats;      // vec_seq.shape[0] of the stored sample
        const double duration = end - start;
        const double sample_end = start + (duration * (double)n_poses) / (double)n_ext;         // :155, evaluated left to right like the reference
        const double frame_duration = (sample_end - start) / (double)n_poses;                    // :119
        const int nw = n_words[b];
        int64_t* row = out_text + (long)b * n_poses;
        if (remove_word_timing) {
            int cnt = 0;
            for (int w = 0; w < nw; ++w) {
                const double f = floor((word_onset[(long)b * Wmax + w] - start) / frame_duration);
                const long idx = f < 0.0 ? 0 : (long)f;
                if (idx < n_poses) ++cnt;
            }
            const int space = n_poses / (cnt + 1);
            for (int i = 0; i < cnt; ++i) row[(i + 1) * space] = word_idx[(long)b * Wmax + i];
        } else {
            for (int w = 0; w < nw; ++w) {
                const double f = floor((word_onset[(long)b * Wmax + w] - start) / frame_duration);
                const long idx = f < 0.0 ? 0 : (f >= 9.0e18 ? (long)n_poses : (long)f);
                if (idx < n_poses) row[idx] = word_idx[(long)b * Wmax + w];
            }
        }
    }
}

}  // namespace tg

using namespace tg;

extern "C" int tg_assemble_batch(const float* audio_raw, const int64_t* audio_off, const float* vec_raw, const int64_t* vec_off,
                                 const int64_t* word_idx, const double* word_onset, const int32_t* n_words, const double* times,
                                 const int32_t* n_ext, const int64_t* vid_in, int32_t B, int32_t Wmax, int32_t n_poses, int32_t pose_floats, int32_t audio_len,
                                 int32_t remove_word_timing, int64_t* out_text, float* out_audio, float* out_vec, int64_t* out_vid,
                                 void* stream) {
    TG_REQUIRE(audio_raw && audio_off && vec_raw && vec_off && word_idx && word_onset && n_words && times && out_text && out_audio && out_vec &&
               out_vid, "tg_assemble_batch: null pointer");
    TG_REQUIRE(B > 0 && Wmax > 0 && n_poses > 0 && pose_floats > 0 && audio_len > 0, "tg_assemble_batch: bad sizes B=%d Wmax=%d n_poses=%d D=%d A=%d",
               B, Wmax, n_poses, pose_floats, audio_len);
    hipLaunchKernelGGL(assemble_batch_kernel, dim3(B, cdiv(audio_len, 1024)), dim3(256), 0, (hipStream_t)stream, audio_raw, audio_off, vec_raw,
                       vec_off, word_idx, word_onset, n_words, times, n_ext, vid_in, Wmax, n_poses, pose_floats, audio_len, remove_word_timing, out_text,
                       out_audio, out_vec, out_vid);
    return check_launch("tg_assemble_batch");
}
