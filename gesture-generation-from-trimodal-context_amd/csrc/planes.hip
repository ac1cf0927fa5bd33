// Pre-split operand planes: an fp32 matrix [rows][cw] written ONCE as three bf16 planes (hi / mid / lo, exact: x = hi + mid + lo), each
// slab-tiled [cwp / 32][rows + 1][32] (common.hpp plane_tiled_off; cwp = cw rounded up to 32, zero columns past cw, row `rows` of every slab
// all zero).  The weights of the many-row forward products are kept this way (layers.WeightPrep refreshes them once per optimiser step) and
// go global -> LDS by DMA in the mover-wave kernel (gemm_mw.hip), which then multiplies without any split arithmetic on that operand.
// (Rounds 2-3 also had kernels that took BOTH operands pre-split -- level with splitting the activation while it is staged, DESIGN.md
// section 5; removed in round 4.)
#include "common.hpp"
#include <stdlib.h>

namespace tg {

// fp32 [rows][cw] (row stride ldx) -> planes [3][rows + 1][cwp] bf16: exact three-way split, zero columns past cw, zero row `rows`.
// One thread per 8 output columns of a row: two 16-byte loads where the source allows, three 16-byte stores.
__global__ __launch_bounds__(256) void split3_planes_kernel(const float* __restrict__ x, long ldx, int rows, int cw, int cwp, __bf16* __restrict__ planes,
                                                            long plane_stride, int vec) {
    const int c8 = cwp / 8;
    const long total = (long)(rows + 1) * c8;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long r = i / c8;
        const int c = (int)(i - r * c8) * 8;
        split3_write_piece(x, ldx, rows, cw, cwp, planes, plane_stride, r, c, vec != 0);
    }
}

// fp32 [rows][cw] -> fp16 x 2 planes [2][rows + 1][cwp] + inv[rows + 1] (common.hpp h2_write_row): one wave per row
__global__ __launch_bounds__(256) void split2h_planes_kernel(const float* __restrict__ x, long ldx, int rows, int cw, int cwp, _Float16* __restrict__ planes,
                                                             long plane_stride, float* __restrict__ inv, int vec) {
    const int lane = threadIdx.x & 63;
    for (long r = (long)blockIdx.x * 4 + (threadIdx.x >> 6); r <= rows; r += (long)gridDim.x * 4)
        h2_write_row(x, ldx, rows, cw, cwp, planes, plane_stride, inv, r, lane, vec != 0);
}

__global__ __launch_bounds__(256) void split2h_planes_tcat_kernel(const float* __restrict__ w0, const float* __restrict__ w1, int rows, int cols, int cwp,
                                                                  _Float16* __restrict__ planes, long plane_stride, float* __restrict__ inv) {
    __shared__ unsigned smax[32][H2_TCAT_ROWS];
    h2_planes_tcat_block(w0, w1, rows, cols, cwp, planes, plane_stride, inv, blockIdx.x, gridDim.x, smax);
}

// Largest magnitude of every SOURCE row of a window operand: rmax[b * rows_in + r] = max_c |ptr[b * bs + r * rs + c]|, c < cw -- what the
// fp16 x 2 kernels derive the power-of-two scale of a product's row from (the maximum over the row's taps).  One wave per row, four rows
// in flight per wave.
__global__ __launch_bounds__(256) void win_row_absmax_kernel(const float* __restrict__ x, long bs, long rs, int batches, int rows_in, int cw, float* __restrict__ rmax,
                                                             int vec) {
    const int lane = threadIdx.x & 63;
    const long total = (long)batches * rows_in;
    const long wave = (long)blockIdx.x * 4 + (threadIdx.x >> 6), n_waves = (long)gridDim.x * 4;
    for (long i0 = wave * 4; i0 < total; i0 += n_waves * 4) {
        unsigned mx[4] = {0, 0, 0, 0};
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const long i = i0 + u;
            if (i >= total) break;
            const long b = i / rows_in;
            const float* row = x + b * bs + (i - b * rows_in) * rs;
            if (vec) {
                for (int c = lane * 4; c < cw; c += 256) {
                    const f32x4 v = *reinterpret_cast<const f32x4*>(row + c);
#pragma unroll
                    for (int q = 0; q < 4; ++q) { const float f = v[q]; const unsigned bb = __float_as_uint(f) & 0x7fffffffu; mx[u] = mx[u] > bb ? mx[u] : bb; }
                }
            } else {
                for (int c = lane; c < cw; c += 64) { const unsigned bb = __float_as_uint(row[c]) & 0x7fffffffu; mx[u] = mx[u] > bb ? mx[u] : bb; }
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const unsigned m = wave_max_u32(mx[u]);
            if (lane == 0 && i0 + u < total) rmax[i0 + u] = __uint_as_float(m);
        }
    }
}

// power-of-two scale of every product row of a window (tg_h2_row_scales): 16 lanes per row, four rows per wave; with rmax != nullptr only
// lane 0 of a row's 16 does anything (taps loads)
__global__ __launch_bounds__(256) void h2_row_scales_kernel(const Win A, int M, const float* __restrict__ rmax, float* __restrict__ scale, int vec) {
    const int sub = threadIdx.x & 15;
    const int taps = A.K / A.cw;
    for (long m = ((long)blockIdx.x * 256 + threadIdx.x) >> 4; m < (((long)M + 15) & ~15l); m += ((long)gridDim.x * 256) >> 4) {
        unsigned mx = 0;
        if (m < M) {
            const int b = (int)(m / A.rows_out);
            int sr = ((int)m - b * A.rows_out) * A.step + A.shift;
            for (int t = 0; t < taps; ++t, sr += A.dil) {
                if ((unsigned)sr >= (unsigned)A.rows_in) continue;
                if (rmax) {
                    if (sub == 0) { const unsigned v = __float_as_uint(rmax[(long)b * A.rows_in + sr]) & 0x7fffffffu; mx = mx > v ? mx : v; }
                    continue;
                }
                const float* row = A.ptr + (long)b * A.bs + (long)sr * A.rs;
                if (vec) {
                    for (int c = sub * 4; c < A.cw; c += 64) {
                        const f32x4 v = *reinterpret_cast<const f32x4*>(row + c);
#pragma unroll
                        for (int q = 0; q < 4; ++q) { const float f = v[q]; const unsigned bb = __float_as_uint(f) & 0x7fffffffu; mx = mx > bb ? mx : bb; }
                    }
                } else {
                    for (int c = sub; c < A.cw; c += 16) { const unsigned bb = __float_as_uint(row[c]) & 0x7fffffffu; mx = mx > bb ? mx : bb; }
                }
            }
        }
#pragma unroll
        for (int o = 8; o >= 1; o >>= 1) { const unsigned w = (unsigned)__shfl_xor((int)mx, o, 64); mx = mx > w ? mx : w; }
        if (sub == 0 && m < M) scale[m] = h2_scale_of_exp(h2_exp_of_bits(mx));
    }
}

// row and column magnitudes of x[M][C] in one pass (tg_absmax_rows_cols): a wave walks whole rows (lane owns 4-column pieces lane, lane + 64, ...;
// at most PIECES of them: C <= 256 PIECES), keeps the row's maximum (wave reduction) and its columns' maxima in registers; the four waves of a
// workgroup combine their column maxima through LDS, one atomic per column and workgroup.  A workgroup's rows belong to ONE row group.
template <int PIECES>
__global__ __launch_bounds__(256) void absmax_rows_cols_kernel(const float* __restrict__ x, long ldx, int M, int C, int groups, int wgs_per_group,
                                                               float* __restrict__ rowmax, float* __restrict__ colmax) {
    __shared__ unsigned cm[4][256 * PIECES];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int grp = blockIdx.x / wgs_per_group, wg = blockIdx.x % wgs_per_group;
    const int rows_g = M / groups;
    const int c4n = C / 4;
    unsigned cmx[PIECES][4];
#pragma unroll
    for (int p = 0; p < PIECES; ++p)
#pragma unroll
        for (int q = 0; q < 4; ++q) cmx[p][q] = 0u;
    for (int r = wg * 4 + wave; r < rows_g; r += wgs_per_group * 4) {
        const float* row = x + ((long)grp * rows_g + r) * ldx;
        unsigned rmx = 0u;
#pragma unroll
        for (int p = 0; p < PIECES; ++p) {
            const int c4 = lane + 64 * p;
            if (c4 < c4n) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(row + 4 * c4);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float f = v[q];
                    const unsigned b = __float_as_uint(f) & 0x7fffffffu;
                    cmx[p][q] = cmx[p][q] > b ? cmx[p][q] : b;
                    rmx = rmx > b ? rmx : b;
                }
            }
        }
        if (rowmax) {
            rmx = wave_max_u32(rmx);
            if (lane == 0) rowmax[(long)grp * rows_g + r] = __uint_as_float(rmx);
        }
    }
    if (!colmax) return;
#pragma unroll
    for (int p = 0; p < PIECES; ++p)
#pragma unroll
        for (int q = 0; q < 4; ++q) cm[wave][(lane + 64 * p) * 4 + q] = cmx[p][q];
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += 256) {
        const unsigned a = cm[0][c], b = cm[1][c], d = cm[2][c], e = cm[3][c];
        const unsigned m = (a > b ? a : b) > (d > e ? d : e) ? (a > b ? a : b) : (d > e ? d : e);
        if (m) atomicMax(reinterpret_cast<unsigned*>(colmax) + (long)grp * C + c, m);
    }
}

}  // namespace tg

using namespace tg;

extern "C" int tg_absmax_rows_cols(const float* x, int64_t ldx, int32_t M, int32_t C, int32_t groups, float* rowmax, float* colmax, void* stream) {
    TG_REQUIRE(x && (rowmax || colmax) && M > 0 && C > 0 && groups > 0 && M % groups == 0 && ldx >= C, "tg_absmax_rows_cols: bad arguments");
    TG_REQUIRE(C % 4 == 0 && ldx % 4 == 0 && aligned16(x) && C <= 2048, "tg_absmax_rows_cols: C=%d (a multiple of 4, at most 2048), ldx %% 4 == 0, x 16-byte aligned", C);
    hipStream_t s = (hipStream_t)stream;
    if (colmax && zero_async(colmax, sizeof(float) * (size_t)groups * C, s)) return 1;
    const int rows_g = M / groups;
    int wpg = cdiv(rows_g, 32);                                   // ~8 rows per wave
    if (wpg * groups > 1024) wpg = 1024 / groups > 0 ? 1024 / groups : 1;
    const dim3 grid(wpg * groups);
    const int pieces = cdiv(C, 256);
#define TG_AM(P_) hipLaunchKernelGGL(absmax_rows_cols_kernel<P_>, grid, dim3(256), 0, s, x, (long)ldx, M, C, groups, wpg, rowmax, colmax)
    if (pieces <= 1) TG_AM(1); else if (pieces <= 2) TG_AM(2); else if (pieces <= 4) TG_AM(4); else TG_AM(8);
#undef TG_AM
    return check_launch("tg_absmax_rows_cols");
}

extern "C" int tg_h2_row_scales(const tg_window* A, int32_t M, const float* src_rowmax, float* row_scale, void* stream) {
    TG_REQUIRE(A && A->ptr && row_scale && M > 0 && A->rows_in > 0 && A->rows_out > 0 && A->cw > 0 && A->K % A->cw == 0, "tg_h2_row_scales: bad arguments");
    const int vec = (A->cw % 4 == 0) && (A->batch_stride % 4 == 0) && (A->row_stride % 4 == 0) && aligned16(A->ptr);
    long blocks = ((long)M + 15) / 16;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(h2_row_scales_kernel, dim3((int)blocks), dim3(256), 0, (hipStream_t)stream, to_win(A), M, src_rowmax, row_scale, vec);
    return check_launch("tg_h2_row_scales");
}

extern "C" int tg_split2h_planes(const float* x, int64_t ldx, int32_t rows, int32_t cw, void* planes, int32_t cwp, int64_t plane_stride, float* inv, void* stream) {
    TG_REQUIRE(x && planes && inv && rows > 0 && cw > 0 && ldx >= cw, "tg_split2h_planes: bad arguments");
    TG_REQUIRE(cwp >= cw && cwp % 32 == 0 && plane_stride >= (int64_t)(rows + 1) * cwp && plane_stride % 8 == 0 && aligned16(planes),
               "tg_split2h_planes: cwp=%d must be a multiple of 32 >= cw=%d, plane_stride >= (rows + 1) * cwp and a multiple of 8, planes 16-byte aligned", cwp, cw);
    const int vec = (ldx % 4 == 0) && aligned16(x);
    int blocks = cdiv(rows + 1, 4);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(split2h_planes_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, (long)ldx, rows, cw, cwp,
                       reinterpret_cast<_Float16*>(planes), (long)plane_stride, inv, vec);
    return check_launch("tg_split2h_planes");
}

extern "C" int tg_split2h_planes_tcat(const float* w0, const float* w1, int32_t rows, int32_t cols, void* planes, int32_t cwp, int64_t plane_stride, float* inv,
                                      void* stream) {
    TG_REQUIRE(w0 && w1 && planes && inv && rows > 0 && cols > 0, "tg_split2h_planes_tcat: bad arguments");
    TG_REQUIRE(cwp >= 2 * rows && cwp % 32 == 0 && plane_stride >= (int64_t)(cols + 1) * cwp && plane_stride % 8 == 0 && aligned16(planes),
               "tg_split2h_planes_tcat: cwp=%d must be a multiple of 32 >= 2 rows = %d, plane_stride >= (cols + 1) * cwp and a multiple of 8, planes 16-byte aligned", cwp, 2 * rows);
    hipLaunchKernelGGL(split2h_planes_tcat_kernel, dim3(cdiv(cols + 1, H2_TCAT_ROWS)), dim3(256), 0, (hipStream_t)stream, w0, w1, rows, cols, cwp,
                       reinterpret_cast<_Float16*>(planes), (long)plane_stride, inv);
    return check_launch("tg_split2h_planes_tcat");
}

extern "C" int tg_win_row_absmax(const tg_window* A, int32_t batches, float* rmax, void* stream) {
    TG_REQUIRE(A && A->ptr && rmax && batches > 0 && A->rows_in > 0 && A->cw > 0, "tg_win_row_absmax: bad arguments");
    const int vec = (A->cw % 4 == 0) && (A->batch_stride % 4 == 0) && (A->row_stride % 4 == 0) && aligned16(A->ptr);
    const long total = (long)batches * A->rows_in;
    long blocks = (total + 15) / 16;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(win_row_absmax_kernel, dim3((int)blocks), dim3(256), 0, (hipStream_t)stream, A->ptr, (long)A->batch_stride, (long)A->row_stride, batches,
                       A->rows_in, A->cw, rmax, vec);
    return check_launch("tg_win_row_absmax");
}

extern "C" int tg_split3_planes(const float* x, int64_t ldx, int32_t rows, int32_t cw, void* planes, int32_t cwp, int64_t plane_stride, void* stream) {
    TG_REQUIRE(x && planes && rows > 0 && cw > 0 && ldx >= cw, "tg_split3_planes: bad arguments");
    TG_REQUIRE(cwp >= cw && cwp % 32 == 0 && plane_stride >= (int64_t)(rows + 1) * cwp && plane_stride % 8 == 0 && aligned16(planes),
               "tg_split3_planes: cwp=%d must be a multiple of 32 >= cw=%d, plane_stride >= (rows + 1) * cwp and a multiple of 8, planes 16-byte aligned", cwp, cw);
    const int vec = (ldx % 4 == 0) && aligned16(x);
    const long total = (long)(rows + 1) * (cwp / 8);
    hipLaunchKernelGGL(split3_planes_kernel, dim3(ew_grid(total, 256, 2)), dim3(256), 0, (hipStream_t)stream, x, (long)ldx, rows, cw, cwp,
                       reinterpret_cast<__bf16*>(planes), (long)plane_stride, vec);
    return check_launch("tg_split3_planes");
}

