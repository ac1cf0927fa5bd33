// bf16-operand tier of the GEMM-shaped ops (math mode 1): operands are rounded to bf16 (RNE, v_cvt_pk_bf16_f32) ONCE while
// a 32-deep K slab is staged into LDS, products accumulate in fp32 on v_mfma_f32_16x16x32_bf16 (16x the f32 MFMA rate).
// Everything in HBM stays fp32 (activations, master weights, gradients); only the matrix-core feed is bf16.
//
// Tile: 128 x (32*TN) per workgroup, 4 waves as 2 x 2, each wave 64 x (16*TN) = 4 x TN MFMA tiles.  LDS rows hold
// 32 bf16 (64 B) padded to 80 B, so a lane's fragment -- A[row l&15][k = 8*(l>>4) .. +7] -- is one 16-byte ds_read.
// At this arithmetic intensity the kernel is bound by the CU's L2->L1 fill rate, not by the matrix core.
#include "common.hpp"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

namespace tg {

constexpr int BF_LD = 40;     // bf16 elements per LDS row (32 + 8 pad)

__device__ __forceinline__ bf16x4 cvt4(const f32x4 v) {
    bf16x4 r;
    r[0] = (__bf16)v[0]; r[1] = (__bf16)v[1]; r[2] = (__bf16)v[2]; r[3] = (__bf16)v[3];
    return r;
}

// C(m, n) = act(sum_k A(m,k) * Bw[n][k] + bias[n]) (+ C).  A: fp32 row window (vectorisable layout), Bw: fp32 [N][ldb].
template <int TN>
__global__ __launch_bounds__(256) void gemm_nt_bf16_kernel(Win A, const float* __restrict__ Bw, long ldb,
                                                           const float* __restrict__ bias, float* __restrict__ C, long cbs,
                                                           long crs, int cR, int M, int N, float slope, int accumulate, int n_nt) {
    constexpr int BN = 32 * TN;
    constexpr int BQ = BN / 32;                       // float4 pieces of B staged per thread per slab (BN*8/256)
    __shared__ __attribute__((aligned(16))) __bf16 as[2][128][BF_LD];
    __shared__ __attribute__((aligned(16))) __bf16 bs[2][BN][BF_LD];
    const int t = threadIdx.x;
    const int lane = t & 63, wave = t >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int r16 = lane & 15, kq = lane >> 4;
    const int lid = xcd_chunked_id(blockIdx.x, gridDim.x);
    const int m0 = (lid / n_nt) * 128, n0 = (lid % n_nt) * BN;
    const int K = A.K;

    // staging map: 8 consecutive lanes cover one 128-byte row of the slab (32 fp32), so every load instruction reads whole
    // cache lines (8 rows x 128 B); thread t owns piece (t & 7) of rows (t >> 3) + 32 q
    const int sp = 4 * (t & 7), sr0 = t >> 3;
    long a_off[4];
    int a_r[4];
    bool a_ok[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int m = m0 + sr0 + 32 * q;
        a_ok[q] = m < M;
        const int mm = a_ok[q] ? m : 0;
        const int b = mm / A.rows_out;
        a_off[q] = (long)b * A.bs;
        a_r[q] = (mm - b * A.rows_out) * A.step + A.shift;
    }
    const float* b_ptr[BQ];
    bool b_ok[BQ];
#pragma unroll
    for (int q = 0; q < BQ; ++q) {
        const int n = n0 + sr0 + 32 * q;
        b_ok[q] = n < N;
        b_ptr[q] = Bw + (long)(b_ok[q] ? n : 0) * ldb;
    }

    // two register sets: the loads of slab s+2 are issued while slab s is multiplied, so a slab's global-load latency
    // (~2 us under load) is covered by two full iterations instead of the few hundred cycles of one slab's MFMAs
    f32x4 ga[2][4], gb[2][BQ];
    auto fetch = [&](int k0, f32x4 (&ra)[4], f32x4 (&rb)[BQ]) {
        // loads are issued UNCONDITIONALLY from an always-valid address and zeroed afterwards: a predicated load makes the
        // number of outstanding loads dynamic and hipcc then drains everything (vmcnt(0)) at the next use
        const int k = k0 + sp;
        const bool inb = k < K;
        const int kc = inb ? k : 0;
        const int kk = kc / A.cw;
        const int c = kc - kk * A.cw;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int sr = a_r[q] + kk * A.dil;
            const bool ok = a_ok[q] && inb && sr >= 0 && sr < A.rows_in;
            const float* src = ok ? A.ptr + a_off[q] + (long)sr * A.rs + c : A.ptr;
            const f32x4 v = *reinterpret_cast<const f32x4*>(src);
            f32x4 z = {0.f, 0.f, 0.f, 0.f};
            ra[q] = ok ? v : z;
        }
#pragma unroll
        for (int q = 0; q < BQ; ++q) {
            const bool ok = b_ok[q] && inb;
            const f32x4 v = *reinterpret_cast<const f32x4*>(ok ? b_ptr[q] + kc : Bw);
            f32x4 z = {0.f, 0.f, 0.f, 0.f};
            rb[q] = ok ? v : z;
        }
    };

    f32x4 acc[4][TN];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    auto stage_and_multiply = [&](int buf, int k_next, f32x4 (&ra)[4], f32x4 (&rb)[BQ]) {
#pragma unroll
        for (int q = 0; q < 4; ++q) *reinterpret_cast<bf16x4*>(&as[buf][sr0 + 32 * q][sp]) = cvt4(ra[q]);
#pragma unroll
        for (int q = 0; q < BQ; ++q) *reinterpret_cast<bf16x4*>(&bs[buf][sr0 + 32 * q][sp]) = cvt4(rb[q]);
        __syncthreads();
        fetch(k_next, ra, rb);
        bf16x8 fa[4], fb[TN];
#pragma unroll
        for (int i = 0; i < 4; ++i) fa[i] = *reinterpret_cast<const bf16x8*>(&as[buf][wm * 64 + i * 16 + r16][8 * kq]);
#pragma unroll
        for (int j = 0; j < TN; ++j) fb[j] = *reinterpret_cast<const bf16x8*>(&bs[buf][wn * (16 * TN) + j * 16 + r16][8 * kq]);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
    };

    fetch(0, ga[0], gb[0]);
    fetch(32, ga[1], gb[1]);
    for (int k0 = 0; k0 < K; k0 += 64) {
        stage_and_multiply(0, k0 + 64, ga[0], gb[0]);
        if (k0 + 32 < K) stage_and_multiply(1, k0 + 96, ga[1], gb[1]);
    }

#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int row = m0 + wm * 64 + i * 16 + kq * 4 + q;
            if (row >= M) continue;
            const int cb = row / cR;
            const int cr = row - cb * cR;
            float* crow = C + (long)cb * cbs + (long)cr * crs;
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int col = n0 + wn * (16 * TN) + j * 16 + r16;
                if (col >= N) continue;
                float v = acc[i][j][q];
                if (bias) v += bias[col];
                v = act_fn(v, slope);
                if (accumulate) v += crow[col];
                crow[col] = v;
            }
        }
    }
}

// out[k][m] = bf16(A(m, k)), m < Mp (zero for m >= M): the transposed, converted copy of a (windowed) operand that the
// weight-gradient product reads with its reduction index contiguous.  32 x 32 tiles through LDS.
__global__ __launch_bounds__(256) void window_transpose_bf16_kernel(Win A, int M, int Mp, __bf16* __restrict__ out) {
    __shared__ float tile[32][33];
    const int k0 = blockIdx.x * 32, m0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;      // 32 x 8
    const int K = A.K;
#pragma unroll
    for (int r = ty; r < 32; r += 8) {
        const int m = m0 + r, k = k0 + tx;
        float v = 0.f;
        if (m < M && k < K) {
            const int b = m / A.rows_out;
            const int ri = m - b * A.rows_out;
            const int kk = k / A.cw;
            const int c = k - kk * A.cw;
            const int sr = ri * A.step + A.shift + kk * A.dil;
            if (sr >= 0 && sr < A.rows_in) v = A.ptr[(long)b * A.bs + (long)sr * A.rs + c];
        }
        tile[r][tx] = v;
    }
    __syncthreads();
#pragma unroll
    for (int r = ty; r < 32; r += 8) {
        const int k = k0 + r, m = m0 + tx;
        if (k < K && m < Mp) out[(long)k * Mp + m] = (__bf16)tile[tx][r];
    }
}

// dW[n][perm(k)] += sum_m At[n][m] * Bt[k][m]  (+ dbias[n] += sum_m At[n][m]): both operands bf16 with the reduction index
// contiguous (produced by window_transpose_bf16_kernel), fp32 atomics across the splits of the m range.
template <int TN>
__global__ __launch_bounds__(256) void gemm_wgrad_bf16_kernel(const __bf16* __restrict__ At, const __bf16* __restrict__ Bt, int Mp,
                                                              float* __restrict__ dW, long ldw, int N, int K, int cw, int out_kw,
                                                              float* __restrict__ dbias, int slabs_per_split, int n_nt, int n_kt) {
    constexpr int BN = 32 * TN;
    __shared__ __attribute__((aligned(16))) __bf16 as[2][128][BF_LD];
    __shared__ __attribute__((aligned(16))) __bf16 bs[2][BN][BF_LD];
    const int t = threadIdx.x;
    const int lane = t & 63, wave = t >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int r16 = lane & 15, kq = lane >> 4;
    const int lid = xcd_chunked_id(blockIdx.x, gridDim.x);
    const int tile_n = lid % n_nt, tile_k = (lid / n_nt) % n_kt, split = lid / (n_nt * n_kt);
    const int n0 = tile_n * 128, kc0 = tile_k * BN;
    const int m_begin = split * slabs_per_split * 32;
    const int m_end = min(Mp, m_begin + slabs_per_split * 32);

    // staging: 16-byte pieces (8 bf16).  A: 128 rows x 4 pieces = 512 -> 2 per thread; B: BN rows x 4 pieces
    const int arow = t >> 1, apc = 2 * (t & 1);
    const bool a_ok = n0 + arow < N;
    const __bf16* a_ptr = At + (long)(a_ok ? n0 + arow : 0) * Mp;
    constexpr int BP = BN * 4 / 256;                   // pieces of B per thread (1 or 2)
    const int brow = (t * BP) >> 2, bpc = (t * BP) & 3;
    const bool b_ok = kc0 + brow < K;
    const __bf16* b_ptr = Bt + (long)(b_ok ? kc0 + brow : 0) * Mp;
    bf16x8 ga[2], gb[BP];
    auto fetch = [&](int mb) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int m = mb + 8 * (apc + q);
            bf16x8 z = {};
            ga[q] = (a_ok && m < m_end) ? *reinterpret_cast<const bf16x8*>(a_ptr + m) : z;     // Mp % 8 == 0
        }
#pragma unroll
        for (int q = 0; q < BP; ++q) {
            const int m = mb + 8 * (bpc + q);
            bf16x8 z = {};
            gb[q] = (b_ok && m < m_end) ? *reinterpret_cast<const bf16x8*>(b_ptr + m) : z;
        }
    };
    f32x4 acc[4][TN];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    float bsum = 0.f;                                   // bias gradient: row sums of At (only tile_k == 0 workgroups)
    const bool want_bias = dbias != nullptr && tile_k == 0;

    if (m_begin < m_end) fetch(m_begin);
    int buf = 0;
    for (int mb = m_begin; mb < m_end; mb += 32) {
#pragma unroll
        for (int q = 0; q < 2; ++q) *reinterpret_cast<bf16x8*>(&as[buf][arow][8 * (apc + q)]) = ga[q];
#pragma unroll
        for (int q = 0; q < BP; ++q) *reinterpret_cast<bf16x8*>(&bs[buf][brow][8 * (bpc + q)]) = gb[q];
        if (want_bias) {
#pragma unroll
            for (int q = 0; q < 2; ++q)
#pragma unroll
                for (int e = 0; e < 8; ++e) bsum += (float)ga[q][e];
        }
        __syncthreads();
        if (mb + 32 < m_end) fetch(mb + 32);
        bf16x8 fa[4], fb[TN];
#pragma unroll
        for (int i = 0; i < 4; ++i) fa[i] = *reinterpret_cast<const bf16x8*>(&as[buf][wm * 64 + i * 16 + r16][8 * kq]);
#pragma unroll
        for (int j = 0; j < TN; ++j) fb[j] = *reinterpret_cast<const bf16x8*>(&bs[buf][wn * (16 * TN) + j * 16 + r16][8 * kq]);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
        buf ^= 1;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int n = n0 + wm * 64 + i * 16 + kq * 4 + q;
            if (n >= N) continue;
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int kc = kc0 + wn * (16 * TN) + j * 16 + r16;
                if (kc >= K) continue;
                const long off = out_kw > 0 ? (long)(kc % cw) * out_kw + kc / cw : (long)kc;
                atomicAdd(&dW[(long)n * ldw + off], acc[i][j][q]);
            }
        }
    if (want_bias) {
        // the two threads of a staging row hold the halves of its sum
        const float other = __shfl_xor(bsum, 1, 64);
        if ((t & 1) == 0 && a_ok) atomicAdd(&dbias[n0 + arow], bsum + other);
    }
}

static int g_math_mode = 0;

}  // namespace tg

using namespace tg;

extern "C" int tg_set_math_mode(int32_t mode) {
    TG_REQUIRE(mode >= 0 && mode <= 2, "tg_set_math_mode: 0 = fp32 matrix cores (exact fp32), 1 = bf16 operands for forward / input-gradient products, 2 = also for weight gradients");
    g_math_mode = mode;
    return 0;
}
extern "C" int tg_get_math_mode(void) { return g_math_mode; }

// Called by tg_gemm_nt (gemm.hip) when math mode 1 is on and the layout is vectorisable.
int tg_gemm_nt_bf16_launch(const Win& w, const float* Bw, long ldb, const float* bias, float* C, long cbs, long crs, int cR, int M,
                           int N, float slope, int accumulate, hipStream_t s) {
    if (N > 64) {
        const int n_nt = cdiv(N, 128);
        hipLaunchKernelGGL((gemm_nt_bf16_kernel<4>), dim3(cdiv(M, 128) * n_nt), dim3(256), 0, s, w, Bw, ldb, bias, C, cbs, crs, cR, M, N, slope, accumulate, n_nt);
    } else {
        const int n_nt = cdiv(N, 64);
        hipLaunchKernelGGL((gemm_nt_bf16_kernel<2>), dim3(cdiv(M, 128) * n_nt), dim3(256), 0, s, w, Bw, ldb, bias, C, cbs, crs, cR, M, N, slope, accumulate, n_nt);
    }
    return check_launch("tg_gemm_nt(bf16)");
}

extern "C" int64_t tg_gemm_tn_bf16_ws_bytes(int32_t M, int32_t N, int32_t K) {
    const int64_t Mp = ((int64_t)M + 7) / 8 * 8;
    return ((int64_t)N + K) * Mp * 2;
}

// Weight gradient in math mode 1: transposed bf16 copies of dY and of the (windowed) input into ws, then the split-M product.
int tg_gemm_tn_bf16_launch(const float* dY, long ldy, const Win& w, float* dW, long ldw, int M, int N, int out_kw, float* dbias,
                           void* ws, int64_t ws_bytes, hipStream_t s) {
    const int Mp = (M + 7) / 8 * 8;
    TG_REQUIRE(ws && ws_bytes >= ((int64_t)N + w.K) * Mp * 2 && aligned16(ws), "tg_gemm_tn(bf16): workspace too small or unaligned");
    __bf16* At = reinterpret_cast<__bf16*>(ws);
    __bf16* Bt = At + (long)N * Mp;
    Win yw;
    yw.ptr = dY; yw.bs = 0; yw.rs = ldy; yw.rows_in = M; yw.rows_out = M; yw.step = 1; yw.shift = 0; yw.dil = 1; yw.cw = N; yw.K = N;
    hipLaunchKernelGGL(window_transpose_bf16_kernel, dim3(cdiv(N, 32), cdiv(Mp, 32)), dim3(256), 0, s, yw, M, Mp, At);
    hipLaunchKernelGGL(window_transpose_bf16_kernel, dim3(cdiv(w.K, 32), cdiv(Mp, 32)), dim3(256), 0, s, w, M, Mp, Bt);
    const int K = w.K;
    const bool wide = K > 64;
    const int n_nt = cdiv(N, 128), n_kt = cdiv(K, wide ? 128 : 64);
    const int slabs = cdiv(Mp, 32);
    int splits = 512 / (n_nt * n_kt);
    if (splits > slabs) splits = slabs;
    if (splits < 1) splits = 1;
    const int sps = cdiv(slabs, splits);
    splits = cdiv(slabs, sps);
    if (wide)
        hipLaunchKernelGGL((gemm_wgrad_bf16_kernel<4>), dim3(n_nt * n_kt * splits), dim3(256), 0, s, At, Bt, Mp, dW, ldw, N, K, w.cw, out_kw, dbias, sps, n_nt, n_kt);
    else
        hipLaunchKernelGGL((gemm_wgrad_bf16_kernel<2>), dim3(n_nt * n_kt * splits), dim3(256), 0, s, At, Bt, Mp, dW, ldw, N, K, w.cw, out_kw, dbias, sps, n_nt, n_kt);
    return check_launch("tg_gemm_tn(bf16)");
}
