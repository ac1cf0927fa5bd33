// Host side of the persistent, cluster-synchronised GRU recurrence for the generator (H <= 320): plan (batch tiles x cluster
// width), workspace layout and the C-ABI entry points.  The kernels are in gru_cluster_x3.hip (bf16 x 3 MFMA); the f32-MFMA
// kernels that used to live here lost every A/B since round 2 and were removed in round 4.
//
// ONE launch walks all T steps of both directions.  The recurrence of one (direction, batch tile) never needs another tile's
// data, so there is no grid-wide barrier: the CW = ceil(H / 32) workgroups that share a batch tile form a CLUSTER and hand
// h_t to each other through an exchange buffer with generation-numbered flag words (protocol: gru_cluster_x3.hip).
// Residency: 512-thread workgroups with > 128 VGPRs -> one per CU; the host launches at most 256 of them (all co-resident on
// an otherwise in-order stream).  Every spin is bounded: on a timeout the workgroup sets the timeout word, stops waiting for
// the rest of the sequence and runs to completion (results are then garbage and the host raises on the timeout word).
#include "common.hpp"

using namespace tg;

constexpr int GC_UNITS = 32;          // hidden units per workgroup
constexpr int GC_HX = 320;            // largest H: 10 workgroups per cluster
constexpr int GC_FLAG_STRIDE = 16;    // flag words per cluster (one 64-byte line)


// the kernels' launchers (gru_cluster_x3.hip)
int64_t tg_gru_x3_fwd_exchange_bytes(int b_pad, int cw);
int64_t tg_gru_x3_bwd_exchange_bytes(int b_pad, int cw);
int tg_gru_x3_fwd_launch(int mt, const float* gi, long gi_ds, const float* w0, const float* w1, const float* b0, const float* b1, float* y,
                         float* save, long save_ds, const float* drop_mask, float* y_drop, void* hx, unsigned* flags, unsigned* tmo, int B, int T,
                         int H, int n_bt, int cw, int b_pad, int save_row0, int save_rows, hipStream_t s);
int tg_gru_x3_bwd_launch(const float* dy, const float* dy_mask, const float* y, const float* save, long save_ds, const float* wt0, const float* wt1, float* dgi,
                         float* dgh, long dg_ds, void* gx, unsigned* flags, unsigned* tmo, int B, int T, int H, int n_bt, int cw, int b_pad,
                         float* gi_rmax, long rm_ds, float* gi_cmax, float* gh_cmax,
                         hipStream_t s);
// Workgroups that can be co-resident at one per CU: the device's CU count (256 on a whole MI355X; fewer under CPX/DPX partitioning
// or CU masking, where the cluster kernels must not be used: a member that can never become resident stalls its cluster until
// the spin bound).  0 when no device is usable (the caller then falls back to the per-step launches).
static int resident_cus() {
    static int cached[16] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) { (void)hipGetLastError(); return 0; }
    if (cached[dev] == 0) {
        int n = 0;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) { (void)hipGetLastError(); n = 0; }
        cached[dev] = n > 0 ? n : -1;
    }
    return cached[dev] > 0 ? cached[dev] : 0;
}

static void cluster_plan(int B, int H, int* mt, int* n_bt, int* cw) {
    *cw = cdiv(H, GC_UNITS);
    *mt = (2 * cdiv(B, 16) * *cw <= 256) ? 1 : 2;
    *n_bt = cdiv(B, 16 * *mt);
}

extern "C" int32_t tg_gru_cluster_supported(int32_t B, int32_t H) {
    if (H > GC_HX || H % 4 != 0 || B <= 0) return 0;
    int mt, n_bt, cw;
    cluster_plan(B, H, &mt, &n_bt, &cw);
    const int cus = resident_cus();
    return 2 * n_bt * cw <= (cus < 256 ? cus : 256) && cw <= GC_FLAG_STRIDE;
}

// workspace: [flag block: 2*n_bt clusters x 16 words + 16 words (timeout word first) ...] [exchange buffer 2 dirs x 2 slots]
extern "C" int64_t tg_gru_cluster_ws_bytes(int32_t B, int32_t H) {
    int mt, n_bt, cw;
    cluster_plan(B, H, &mt, &n_bt, &cw);
    const int64_t flag_words = (int64_t)(2 * n_bt + 1) * GC_FLAG_STRIDE;
    const int64_t b_pad = (int64_t)n_bt * 16 * mt;
    return flag_words * 4 + tg_gru_x3_fwd_exchange_bytes((int)b_pad, cw);
}

// (always 1 since the f32-MFMA kernels went; kept so that callers built against ABI <= 5 keep working)
extern "C" int32_t tg_gru_cluster_fused_dropout(void) { return 1; }

extern "C" int tg_gru_forward_cluster(const float* gi, int64_t gi_dir_stride, const float* w_hh_fwd, const float* w_hh_rev,
                                      const float* b_hh_fwd, const float* b_hh_rev, float* y, float* save, int64_t save_dir_stride,
                                      const float* drop_mask, float* y_drop, void* ws, int64_t ws_bytes, int32_t B, int32_t T, int32_t H,
                                      void* stream) {
    return tg_gru_forward_cluster_rows(gi, gi_dir_stride, w_hh_fwd, w_hh_rev, b_hh_fwd, b_hh_rev, y, save, save_dir_stride, drop_mask, y_drop, ws, ws_bytes,
                                       B, T, H, 0, B, stream);
}

extern "C" int tg_gru_forward_cluster_rows(const float* gi, int64_t gi_dir_stride, const float* w_hh_fwd, const float* w_hh_rev,
                                           const float* b_hh_fwd, const float* b_hh_rev, float* y, float* save, int64_t save_dir_stride,
                                           const float* drop_mask, float* y_drop, void* ws, int64_t ws_bytes, int32_t B, int32_t T, int32_t H,
                                           int32_t save_row0, int32_t save_rows, void* stream) {
    TG_REQUIRE(gi && w_hh_fwd && w_hh_rev && b_hh_fwd && b_hh_rev && y && ws, "tg_gru_forward_cluster: null pointer");
    TG_REQUIRE(save_row0 >= 0 && save_rows >= 0 && (int64_t)save_row0 + save_rows <= B, "tg_gru_forward_cluster_rows: saved rows [%d, %d) outside the batch", save_row0, save_row0 + save_rows);
    TG_REQUIRE((drop_mask == nullptr) == (y_drop == nullptr) && (drop_mask == nullptr || (aligned16(drop_mask) && aligned16(y_drop))),
               "tg_gru_forward_cluster: drop_mask / y_drop go together and must be 16-byte aligned");
    TG_REQUIRE(T > 0 && tg_gru_cluster_supported(B, H), "tg_gru_forward_cluster: unsupported shape B=%d H=%d", B, H);
    TG_REQUIRE((int64_t)B * T * 4 * H * 4 < (1LL << 31), "tg_gru_forward_cluster: B * T * 4H floats must stay below 2 GB (32-bit buffer offsets), T=%d", T);
    TG_REQUIRE(ws_bytes >= tg_gru_cluster_ws_bytes(B, H), "tg_gru_forward_cluster: workspace too small");
    TG_REQUIRE(aligned16(gi) && aligned16(w_hh_fwd) && aligned16(w_hh_rev) && aligned16(b_hh_fwd) && aligned16(b_hh_rev) && aligned16(y) &&
               aligned16(ws) && (save == nullptr || aligned16(save)) && gi_dir_stride % 4 == 0 && save_dir_stride % 4 == 0,
               "tg_gru_forward_cluster: operands must be 16-byte aligned");
    int mt, n_bt, cw;
    cluster_plan(B, H, &mt, &n_bt, &cw);
    const int64_t flag_words = (int64_t)(2 * n_bt + 1) * GC_FLAG_STRIDE;
    hipStream_t s = (hipStream_t)stream;
    // The first 16 words (the timeout marker and its diagnostics) are sticky until the host reads and clears them
    // (ops.check_async_errors), so a timeout in any launch that shares this workspace survives the launches after it.  The flag
    // words behind them are numbered by generation and never zeroed (gru_cluster_x3.hip).
    unsigned* tmo = (unsigned*)ws;
    unsigned* flags = tmo + GC_FLAG_STRIDE;
    float* hx = (float*)(tmo + flag_words);
    const int b_pad = n_bt * 16 * mt;
    return tg_gru_x3_fwd_launch(mt, gi, (long)gi_dir_stride, w_hh_fwd, w_hh_rev, b_hh_fwd, b_hh_rev, y, save, (long)save_dir_stride, drop_mask, y_drop, hx,
                                flags, tmo, B, T, H, n_bt, cw, b_pad, save_row0, save_rows, s);
}

// ---- backward
static void cluster_plan_bwd(int B, int H, int* n_bt, int* cw) {
    *cw = cdiv(H, GC_UNITS);
    *n_bt = cdiv(B, 16);
}

extern "C" int32_t tg_gru_cluster_bwd_supported(int32_t B, int32_t H) {
    if (H > GC_HX || H % 4 != 0 || B <= 0) return 0;
    int n_bt, cw;
    cluster_plan_bwd(B, H, &n_bt, &cw);
    const int cus = resident_cus();
    return 2 * n_bt * cw <= (cus < 256 ? cus : 256) && cw <= GC_FLAG_STRIDE;
}

extern "C" int64_t tg_gru_cluster_bwd_ws_bytes(int32_t B, int32_t H) {
    int n_bt, cw;
    cluster_plan_bwd(B, H, &n_bt, &cw);
    const int64_t flag_words = (int64_t)(2 * n_bt + 1) * GC_FLAG_STRIDE;
    return flag_words * 4 + tg_gru_x3_bwd_exchange_bytes(n_bt * 16, cw);
}

extern "C" int tg_gru_backward_cluster_stats(const float* dy, const float* dy_mask, const float* y, const float* save, int64_t save_dir_stride,
                                       const float* w_hh_t_fwd, const float* w_hh_t_rev, float* dgi, float* dgh, int64_t dg_dir_stride,
                                       void* ws, int64_t ws_bytes, int32_t B, int32_t T, int32_t H, float* gi_rowmax, int64_t rowmax_dir_stride,
                                       float* gi_colmax, float* gh_colmax, void* stream) {
    TG_REQUIRE((gi_rowmax != nullptr) == (gi_colmax != nullptr) && (gi_rowmax != nullptr) == (gh_colmax != nullptr) && rowmax_dir_stride >= 0,
               "tg_gru_backward_cluster_stats: gi_rowmax, gi_colmax and gh_colmax go together");
    TG_REQUIRE(dy && y && save && w_hh_t_fwd && w_hh_t_rev && dgi && dgh && ws, "tg_gru_backward_cluster_stats: null pointer");
    TG_REQUIRE(dy_mask == nullptr || aligned16(dy_mask), "tg_gru_backward_cluster_stats: dy_mask must be 16-byte aligned");
    TG_REQUIRE(T > 0 && tg_gru_cluster_bwd_supported(B, H), "tg_gru_backward_cluster_stats: unsupported shape B=%d H=%d", B, H);
    TG_REQUIRE((int64_t)B * T * 4 * H * 4 < (1LL << 31), "tg_gru_backward_cluster_stats: B * T * 4H floats must stay below 2 GB (32-bit buffer offsets), T=%d", T);
    TG_REQUIRE(ws_bytes >= tg_gru_cluster_bwd_ws_bytes(B, H), "tg_gru_backward_cluster_stats: workspace too small");
    TG_REQUIRE(aligned16(dy) && aligned16(y) && aligned16(save) && aligned16(w_hh_t_fwd) && aligned16(w_hh_t_rev) && aligned16(dgi) &&
               aligned16(dgh) && aligned16(ws) && save_dir_stride % 4 == 0 && dg_dir_stride % 4 == 0,
               "tg_gru_backward_cluster_stats: operands must be 16-byte aligned");
    int n_bt, cw;
    cluster_plan_bwd(B, H, &n_bt, &cw);
    const int64_t flag_words = (int64_t)(2 * n_bt + 1) * GC_FLAG_STRIDE;
    hipStream_t s = (hipStream_t)stream;
    unsigned* tmo = (unsigned*)ws;              // sticky timeout block: cleared by the host only (see the forward entry point)
    unsigned* flags = tmo + GC_FLAG_STRIDE;
    float* gx = (float*)(tmo + flag_words);
    return tg_gru_x3_bwd_launch(dy, dy_mask, y, save, (long)save_dir_stride, w_hh_t_fwd, w_hh_t_rev, dgi, dgh, (long)dg_dir_stride, gx, flags, tmo, B, T, H,
                                n_bt, cw, n_bt * 16, gi_rowmax, (long)rowmax_dir_stride, gi_colmax, gh_colmax, s);
}

extern "C" int tg_gru_backward_cluster(const float* dy, const float* dy_mask, const float* y, const float* save, int64_t save_dir_stride,
                                       const float* w_hh_t_fwd, const float* w_hh_t_rev, float* dgi, float* dgh, int64_t dg_dir_stride,
                                       void* ws, int64_t ws_bytes, int32_t B, int32_t T, int32_t H, void* stream) {
    return tg_gru_backward_cluster_stats(dy, dy_mask, y, save, save_dir_stride, w_hh_t_fwd, w_hh_t_rev, dgi, dgh, dg_dir_stride, ws, ws_bytes, B, T, H, nullptr, 0,
                                         nullptr, nullptr, stream);
}
