// Element-wise, data-movement, embedding, weight-norm, RNG and optimiser kernels.  All are HBM- or latency-bound:
// grid-stride loops, 16-byte accesses where the layout allows, no host synchronisation, graph-capture safe.
#include "common.hpp"

namespace tg {

#define GRID_STRIDE(i, n) for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < (n); i += (long)gridDim.x * blockDim.x)

__global__ void zero_kernel(unsigned* __restrict__ p, long n_words) {
    const long n4 = n_words >> 2;                       // 16-byte stores where the pointer allows (checked by the launcher)
    GRID_STRIDE(i, n4) reinterpret_cast<uint4*>(p)[i] = uint4{0u, 0u, 0u, 0u};
    GRID_STRIDE(i, n_words - 4 * n4) p[4 * n4 + i] = 0u;
}
__global__ void zero_kernel_unaligned(unsigned* __restrict__ p, long n_words) {
    GRID_STRIDE(i, n_words) p[i] = 0u;
}

int zero_async(void* p, size_t bytes, hipStream_t s) {
    if (bytes == 0) return 0;
    if (p == nullptr || (bytes & 3u) || (reinterpret_cast<uintptr_t>(p) & 3u)) { set_error("zero_async: bad pointer / size"); return 1; }
    const long n_words = (long)(bytes >> 2);
    long blocks = (n_words / 4 + 255) / 256;
    if (blocks < 1) blocks = 1;
    if (blocks > 4096) blocks = 4096;
    if (aligned16(p)) hipLaunchKernelGGL(zero_kernel, dim3((unsigned)blocks), dim3(256), 0, s, (unsigned*)p, n_words);
    else hipLaunchKernelGGL(zero_kernel_unaligned, dim3((unsigned)blocks), dim3(256), 0, s, (unsigned*)p, n_words);
    return check_launch("zero_async");
}

// 16-byte forms (n % 4 == 0, aligned pointers: checked by the launchers) of the residual-block glue; V = f32x4 or float
template <typename V>
__global__ void add_relu_vkernel(const V* __restrict__ a, const V* __restrict__ b, V* __restrict__ y, long n) {
    constexpr int W = sizeof(V) / 4;
    GRID_STRIDE(i, n) {
        const V av = a[i], bv = b[i];
        V o;
        for (int q = 0; q < W; ++q) { const float v = ((const float*)&av)[q] + ((const float*)&bv)[q]; ((float*)&o)[q] = v > 0.f ? v : 0.f; }
        y[i] = o;
    }
}
// the dropout scale of both gate kernels below: a stored mask, or (f32x4 form only) regenerated from the element index (dropout_scale4)
struct DropSpec {
    const uint64_t* st;
    long idx4_0;           // index0 / 4
    unsigned site;
    float p;
};
template <typename V>
__global__ void act_mask_bwd_kernel(const V* __restrict__ dy, const V* __restrict__ y, const V* __restrict__ mask, float slope, V* __restrict__ dx, long n,
                                    DropSpec drop) {
    constexpr int W = sizeof(V) / 4;
    GRID_STRIDE(i, n) {
        const V dv = dy[i], yv = y[i];
        V mv = dv;
        if (mask) mv = mask[i];
        if constexpr (W == 4) { if (drop.st) mv = dropout_scale4(drop.st, drop.site, drop.p, (unsigned long)(drop.idx4_0 + i)); }
        const bool scaled = mask || drop.st;
        V o;
        for (int q = 0; q < W; ++q) {
            float g = ((const float*)&dv)[q] * (((const float*)&yv)[q] > 0.f ? 1.f : slope);
            if (scaled) g *= ((const float*)&mv)[q];
            ((float*)&o)[q] = g;
        }
        dx[i] = o;
    }
}
// two gates of a residual block in one pass (model/tcn.py:46 backward): dsum = dy * (y > 0) -- the gradient at relu(out + x), kept for the
// residual branch -- and dc = dsum * (o > 0 ? 1 : slope) * mask, the gradient at the block's second conv
template <typename V>
__global__ void act_mask_bwd2_kernel(const V* __restrict__ dy, const V* __restrict__ y, const V* __restrict__ o, const V* __restrict__ mask, float slope,
                                     V* __restrict__ dsum, V* __restrict__ dc, long n, DropSpec drop) {
    constexpr int W = sizeof(V) / 4;
    GRID_STRIDE(i, n) {
        const V dv = dy[i], yv = y[i], ov = o[i];
        V mv = dv;
        if (mask) mv = mask[i];
        if constexpr (W == 4) { if (drop.st) mv = dropout_scale4(drop.st, drop.site, drop.p, (unsigned long)(drop.idx4_0 + i)); }
        const bool scaled = mask || drop.st;
        V s_, c_;
        for (int q = 0; q < W; ++q) {
            const float g = ((const float*)&dv)[q] * (((const float*)&yv)[q] > 0.f ? 1.f : 0.f);
            float h = g * (((const float*)&ov)[q] > 0.f ? 1.f : slope);
            if (scaled) h *= ((const float*)&mv)[q];
            ((float*)&s_)[q] = g;
            ((float*)&c_)[q] = h;
        }
        dsum[i] = s_;
        dc[i] = c_;
    }
}
template <typename V>
__global__ void mul_kernel(const V* __restrict__ x, const V* __restrict__ m, V* __restrict__ y, long n) {
    GRID_STRIDE(i, n) y[i] = x[i] * m[i];
}
__global__ void axpy_kernel(const float* __restrict__ x, float* __restrict__ y, float alpha, int acc, long n) {
    GRID_STRIDE(i, n) y[i] = acc ? y[i] + alpha * x[i] : alpha * x[i];
}
__global__ void copy2d_kernel(const float* __restrict__ src, long lds_, float* __restrict__ dst, long ldd, int rows, int cols, int acc) {
    const long n = (long)rows * cols;
    GRID_STRIDE(i, n) {
        const long r = i / cols;
        const int c = (int)(i - r * cols);
        const float v = src[r * lds_ + c];
        float* d = dst + r * ldd + c;
        *d = acc ? *d + v : v;
    }
}
__global__ void repeat_rows_kernel(const float* __restrict__ src, long lds_, float* __restrict__ dst, long ldd, int B, int T, int cols) {
    const long n = (long)B * T * cols;
    GRID_STRIDE(i, n) {
        const long bt = i / cols;
        const int c = (int)(i - bt * cols);
        dst[bt * ldd + c] = src[(bt / T) * lds_ + c];
    }
}
// eight lanes share one (b, c) output: the T strided loads of a column are independent requests in flight together instead of a
// 34-deep dependent chain per thread (40 us -> a few us for the 128 x 16 speaker-latent gradient)
__global__ void sum_rows_kernel(const float* __restrict__ src, long lds_, float* __restrict__ dst, long ldd, int B, int T, int cols, int acc) {
    const long n = (long)B * cols * 8;
    const long n_pad = (n + 63) / 64 * 64;                 // whole waves take part in the shuffles
    for (long j = (long)blockIdx.x * blockDim.x + threadIdx.x; j < n_pad; j += (long)gridDim.x * blockDim.x) {
        const long i = j >> 3;
        const int sub = (int)(j & 7);
        const bool ok = j < n;
        const long b = ok ? i / cols : 0;
        const int c = ok ? (int)(i - b * cols) : 0;
        float s = 0.f;
        if (ok)
            for (int t = sub; t < T; t += 8) s += src[(b * T + t) * lds_ + c];
        s += __shfl_xor(s, 1);
        s += __shfl_xor(s, 2);
        s += __shfl_xor(s, 4);
        if (ok && sub == 0) {
            float* d = dst + b * ldd + c;
            *d = acc ? *d + s : s;
        }
    }
}
__global__ void add_halves_kernel(const float* __restrict__ y, float* __restrict__ o, long M, int H) {
    const long n = M * H;
    GRID_STRIDE(i, n) {
        const long m = i / H;
        const int j = (int)(i - m * H);
        o[i] = y[m * 2 * H + j] + y[m * 2 * H + H + j];
    }
}
__global__ void dup_halves_kernel(const float* __restrict__ d_o, float* __restrict__ dy, long M, int H) {
    const long n = M * H;
    GRID_STRIDE(i, n) {
        const long m = i / H;
        const int j = (int)(i - m * H);
        const float v = d_o[i];
        dy[m * 2 * H + j] = v;
        dy[m * 2 * H + H + j] = v;
    }
}
__global__ void make_pre_seq_kernel(const float* __restrict__ target, float* __restrict__ pre, int B, int T, int D, int n_pre) {
    const long n = (long)B * T * (D + 1);
    GRID_STRIDE(i, n) {
        const int c = (int)(i % (D + 1));
        const long bt = i / (D + 1);
        const int t = (int)(bt % T);
        pre[i] = t < n_pre ? (c < D ? target[bt * D + c] : 1.f) : 0.f;
    }
}
__global__ void embed_gather_kernel(const float* __restrict__ table, const int64_t* __restrict__ idx, float* __restrict__ out, int n_idx, int D,
                                    int n_rows) {
    const long n = (long)n_idx * D;
    GRID_STRIDE(i, n) {
        const long r = i / D;
        const int c = (int)(i - r * D);
        const int64_t id = idx[r];
        out[i] = (id >= 0 && id < n_rows) ? table[id * D + c] : 0.f;
    }
}
// the look-up followed by F.dropout (multimodal_context_net.py:47-52) in one pass, four columns per thread (D % 4 == 0, 16-byte aligned table
// rows): out = table[idx] * mask, the mask being element i of the draw tg_dropout_mask(.., p, st, site) would write -- not stored, the
// backward regenerates it (tg_act_mask_bwd_drop)
__global__ void embed_gather_drop_kernel(const float* __restrict__ table, const int64_t* __restrict__ idx, float* __restrict__ out, int n_idx, int D,
                                         int n_rows, float p, const uint64_t* __restrict__ st, uint32_t site) {
    const int d4 = D / 4;
    const long n4 = (long)n_idx * d4;
    GRID_STRIDE(i, n4) {
        const long r = i / d4;
        const int c4 = (int)(i - r * d4);
        const int64_t id = idx[r];
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (id >= 0 && id < n_rows) v = reinterpret_cast<const f32x4*>(table + id * D)[c4];
        reinterpret_cast<f32x4*>(out)[i] = v * dropout_scale4(st, site, p, (unsigned long)i);
    }
}
// Dense embedding gradient.  The padded text input is mostly index 0: consecutive look-ups with the same
// index are summed in registers and flushed with ONE atomic per run, so the hot PAD row sees n_idx/CHUNK atomics per
// column instead of thousands.
// The chunk's indices and gradient rows are loaded up front (16 + 16 independent requests per thread), the run combine then works on
// registers: the former 64-long loop issued one index load and one row load per iteration, each waiting for the last (49 us for the 4352
// word rows on 68 workgroups, 34 us for 128 speaker rows on 2).
constexpr int SCATTER_CHUNK = 16;
__global__ __launch_bounds__(256) void embed_scatter_kernel(const float* __restrict__ dout, const int64_t* __restrict__ idx,
                                                            float* __restrict__ dtable, int n_idx, int D, int n_rows) {
    const int i0 = blockIdx.x * SCATTER_CHUNK;
    int64_t ids[SCATTER_CHUNK];
#pragma unroll
    for (int q = 0; q < SCATTER_CHUNK; ++q) ids[q] = i0 + q < n_idx ? idx[i0 + q] : -1;      // -1: never flushed, never equal to a valid id
    for (int c = threadIdx.x; c < D; c += blockDim.x) {
        float v[SCATTER_CHUNK];
#pragma unroll
        for (int q = 0; q < SCATTER_CHUNK; ++q) v[q] = dout[(long)min(i0 + q, n_idx - 1) * D + c];
        int64_t cur = ids[0];
        float acc = 0.f;
#pragma unroll
        for (int q = 0; q < SCATTER_CHUNK; ++q) {
            if (ids[q] != cur) {
                if (cur >= 0 && cur < n_rows) atomicAdd(&dtable[cur * D + c], acc);
                acc = 0.f;
                cur = ids[q];
            }
            acc += v[q];
        }
        if (cur >= 0 && cur < n_rows) atomicAdd(&dtable[cur * D + c], acc);
    }
}
// Deterministic form (tg_set_deterministic): workgroup i (16 waves) owns index i.  If an earlier position holds the same id it does nothing;
// otherwise it is the id's FIRST occurrence: wave 0 lists the positions of that id from i on, in order (64 positions per ballot, 1 024 per LDS
// list -- the padding id of a word batch occurs thousands of times), wave w adds the rows of list entries w, w + 16, .. in that order, and the
// sixteen partial rows are added in wave order into the table row: no atomics, one writer per row, a FIXED summation order (a tree of fixed
// shape, not position order; round 5 -- the one-wave form walked the padding id's ~3 000 rows alone, 350 us per launch).  D <= 512.
constexpr int SCATTER_DET_LIST = 1024;
constexpr int SCATTER_DET_WAVES = 16;
__global__ __launch_bounds__(64 * SCATTER_DET_WAVES) void embed_scatter_det_kernel(const float* __restrict__ dout, const int64_t* __restrict__ idx,
                                                                                   float* __restrict__ dtable, int n_idx, int D, int n_rows) {
    __shared__ int pos[SCATTER_DET_LIST];
    __shared__ float part[SCATTER_DET_WAVES][512];
    __shared__ int s_n, s_next, s_skip;
    const int i = blockIdx.x, lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int64_t id = idx[i];
    if (id < 0 || id >= n_rows) return;                   // (workgroup-uniform)
    if (threadIdx.x == 0) s_skip = 0;
    __syncthreads();
    for (int j0 = 64 * w; j0 < i; j0 += 64 * SCATTER_DET_WAVES) {          // an earlier occurrence owns the row
        const int j = j0 + lane;
        if (__any(j < i && idx[j] == id)) { if (lane == 0) s_skip = 1; break; }
    }
    __syncthreads();
    if (s_skip) return;
    float acc[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) acc[q] = 0.f;
    int j0 = i;
    while (j0 < n_idx) {
        if (w == 0) {                                     // the next (up to) 1 024 occurrences, in order
            int n = 0, j = j0;
            for (; j < n_idx && n + 64 <= SCATTER_DET_LIST; j += 64) {
                const int jj = j + lane;
                const bool hit = jj < n_idx && idx[jj] == id;
                const unsigned long long m = __ballot(hit);
                if (hit) pos[n + __popcll(m & ((1ull << lane) - 1ull))] = jj;
                n += __popcll(m);
            }
            if (lane == 0) { s_n = n; s_next = j; }
        }
        __syncthreads();
        const int n = s_n;
        j0 = s_next;
        for (int q2 = w; q2 < n; q2 += 2 * SCATTER_DET_WAVES) {           // two independent row loads in flight per column; added in list order
            const long r0 = pos[q2], r1 = pos[min(q2 + SCATTER_DET_WAVES, n - 1)];
            const bool has1 = q2 + SCATTER_DET_WAVES < n;
            float v0[8], v1[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const bool c = lane + 64 * q < D;
                v0[q] = c ? dout[r0 * D + lane + 64 * q] : 0.f;
                v1[q] = (c && has1) ? dout[r1 * D + lane + 64 * q] : 0.f;
            }
#pragma unroll
            for (int q = 0; q < 8; ++q) { acc[q] += v0[q]; acc[q] += v1[q]; }
        }
        __syncthreads();
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) part[w][lane + 64 * q] = acc[q];
    __syncthreads();
    for (int c = threadIdx.x; c < D; c += 64 * SCATTER_DET_WAVES) {
        float t = 0.f;
#pragma unroll
        for (int q = 0; q < SCATTER_DET_WAVES; ++q) t += part[q][c];
        dtable[id * D + c] += t;
    }
}

__global__ void permute3_kernel(const float* __restrict__ in, float* __restrict__ out, int d0, int d1, int d2, int p0, int p1, int p2) {
    const int d[3] = {d0, d1, d2};
    const int o0 = d[p0], o1 = d[p1], o2 = d[p2];
    const long n = (long)o0 * o1 * o2;
    GRID_STRIDE(i, n) {
        const int i2 = (int)(i % o2);
        const long t = i / o2;
        const int i1 = (int)(t % o1);
        const int i0 = (int)(t / o1);
        int x[3];
        x[p0] = i0; x[p1] = i1; x[p2] = i2;
        out[i] = in[((long)x[0] * d1 + x[1]) * d2 + x[2]];
    }
}

// Batched permute3: one launch for a table of independent (src, dst, dims, perm) jobs -- all weight transposes / conv packs of a
// network after an optimiser step (layers.WeightPrep).  Table entry = 10 int64: src, dst, d0, d1, d2, p0, p1, p2, first workgroup,
// workgroup count; a workgroup finds its job by scanning the (short) table, then grid-strides inside the job.
__global__ __launch_bounds__(256) void permute3_batch_kernel(const long* __restrict__ desc, int n_jobs) {
    __shared__ float tile[32][33];
    // job = the last table entry whose first workgroup is <= blockIdx.x.  Every wave counts them with one ballot per 64 entries (lane j
    // loads entry j's first workgroup: independent loads) -- the serial scan this replaces walked the table one dependent global load at a
    // time, ~0.5 us per entry: 20-28 us for the 30-40 jobs of a network, more than the transposes themselves.
    int job = -1;
    for (int j0 = 0; j0 < n_jobs; j0 += 64) {
        const int j = j0 + (int)(threadIdx.x & 63);
        const bool mine = j < n_jobs && (long)blockIdx.x >= desc[10 * (j < n_jobs ? j : 0) + 8];
        job += __popcll(__ballot(mine));
    }
    if (job < 0) job = 0;
    const long* e = desc + 10 * job;
    const float* in = reinterpret_cast<const float*>(e[0]);
    float* out = reinterpret_cast<float*>(e[1]);
    const int d[3] = {(int)e[2], (int)e[3], (int)e[4]};
    const int p0 = (int)e[5], p1 = (int)e[6], p2 = (int)e[7];
    const long wg0 = e[8], nwg = e[9];
    if (p0 == 0 && p1 == 2 && p2 == 1) {
        // the common job -- swap the two inner dimensions (weight transposes, conv packs): 32 x 32 tiles through LDS, reads
        // coalesced along d2 and writes coalesced along d1 (the element-wise gather below read one cache line per lane: 75 us for
        // the generator's 6.5 M transposed weights)
        const int t1 = (d[1] + 31) / 32, t2 = (d[2] + 31) / 32;
        const long n_tiles = (long)d[0] * t1 * t2;
        const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;            // 32 x 8
        for (long tl = (long)blockIdx.x - wg0; tl < n_tiles; tl += nwg) {   // workgroup-uniform loop
            const int c2 = (int)(tl % t2);
            const long r = tl / t2;
            const int c1 = (int)(r % t1);
            const int b = (int)(r / t1);
            const float* src = in + (long)b * d[1] * d[2];
            float* dst = out + (long)b * d[1] * d[2];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int i1 = c1 * 32 + ty + 8 * q, i2 = c2 * 32 + tx;
                if (i1 < d[1] && i2 < d[2]) tile[ty + 8 * q][tx] = src[(long)i1 * d[2] + i2];
            }
            __syncthreads();
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int i2 = c2 * 32 + ty + 8 * q, i1 = c1 * 32 + tx;
                if (i1 < d[1] && i2 < d[2]) dst[(long)i2 * d[1] + i1] = tile[tx][ty + 8 * q];
            }
            __syncthreads();
        }
        return;
    }
    if (p0 == 8) {
        // bf16 x 3 planes of a weight matrix (gemm_planes.hip): src [d0 rows][d1 = cw] fp32 contiguous -> [3][d0 + 1][d2 = cwp] bf16
        const int rows = d[0], cw = d[1], cwp = d[2], c8 = cwp / 8;
        const long total = (long)(rows + 1) * c8, plane = (long)(rows + 1) * cwp;
        const bool vec = (cw % 4 == 0) && ((reinterpret_cast<uintptr_t>(in) & 15u) == 0);
        for (long i = ((long)blockIdx.x - wg0) * 256 + threadIdx.x; i < total; i += nwg * 256) {
            const long r = i / c8;
            split3_write_piece(in, (long)cw, rows, cw, cwp, reinterpret_cast<__bf16*>(out), plane, r, (int)(i - r * c8) * 8, vec);
        }
        return;
    }
    if (p0 == 10) {
        // fp16 x 2 planes of a weight matrix (planes.hip tg_split2h_planes): src [d0 rows][d1 = cw] fp32 contiguous -> [2][d0 + 1][d2 = cwp] fp16 followed
        // by the rows' inverse scales (d0 + 1 floats); one wave per row
        const int rows = d[0], cw = d[1], cwp = d[2];
        const long plane = (long)(rows + 1) * cwp;
        _Float16* const planes = reinterpret_cast<_Float16*>(out);
        float* const inv = reinterpret_cast<float*>(planes + 2 * plane);
        const bool vec = (cw % 4 == 0) && ((reinterpret_cast<uintptr_t>(in) & 15u) == 0);
        for (long r = ((long)blockIdx.x - wg0) * 4 + (threadIdx.x >> 6); r <= rows; r += nwg * 4)
            h2_write_row(in, (long)cw, rows, cw, cwp, planes, plane, inv, r, (int)(threadIdx.x & 63), vec);
        return;
    }
    if (p0 == 11) {
        // fp16 x 2 planes of the K-concatenated transpose of two [d0 rows][d1 cols] matrices (src and the pointer in p1): common.hpp h2_planes_tcat_block
        const int rows = d[0], cols = d[1], cwp = d[2];
        const long plane = (long)(cols + 1) * cwp;
        _Float16* const planes = reinterpret_cast<_Float16*>(out);
        __shared__ unsigned smax[32][H2_TCAT_ROWS];
        h2_planes_tcat_block(in, reinterpret_cast<const float*>(e[6]), rows, cols, cwp, planes, plane, reinterpret_cast<float*>(planes + 2 * plane),
                             (int)((long)blockIdx.x - wg0), (int)nwg, smax);
        return;
    }
    if (p0 == 9) {
        // conv input-gradient pack (conv_dgrad_pack_kernel): src (Co, Ci, kw) -> [stride][Ci][J * Co], stride = p1, zero taps past kw
        const int Co = d[0], Ci = d[1], kw = d[2], st = p1, J = (kw + st - 1) / st;
        const int n = st * Ci * J * Co;
        for (int i = (int)((long)blockIdx.x - wg0) * 256 + threadIdx.x; i < n; i += (int)nwg * 256) {
            const int co = i % Co;
            int t = i / Co;
            const int j = t % J; t /= J;
            const int ci = t % Ci;
            const int r = t / Ci;
            const int k = r + st * j;
            out[i] = k < kw ? in[(co * Ci + ci) * kw + k] : 0.f;
        }
        return;
    }
    const int o1 = d[p1], o2 = d[p2];
    const long n = (long)d[0] * d[1] * d[2];
    for (long i = ((long)blockIdx.x - wg0) * 256 + threadIdx.x; i < n; i += nwg * 256) {
        const int i2 = (int)(i % o2);
        const long t = i / o2;
        const int i1 = (int)(t % o1);
        const int i0 = (int)(t / o1);
        int x[3];
        x[p0] = i0; x[p1] = i1; x[p2] = i2;
        out[i] = in[((long)x[0] * d[1] + x[1]) * d[2] + x[2]];
    }
}

__global__ void conv_dgrad_pack_kernel(const float* __restrict__ w, float* __restrict__ out, int Co, int Ci, int kw, int s, int J) {
    const long n = (long)s * Ci * J * Co;
    GRID_STRIDE(i, n) {
        const int co = (int)(i % Co);
        long t = i / Co;
        const int j = (int)(t % J); t /= J;
        const int ci = (int)(t % Ci);
        const int r = (int)(t / Ci);
        const int k = r + s * j;
        out[i] = k < kw ? w[((long)co * Ci + ci) * kw + k] : 0.f;
    }
}

// ---- weight norm -------------------------------------------------------------------------------------------------
__device__ __forceinline__ float block_sum(float v, float* sh) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    if (l == 0) sh[w] = v;
    __syncthreads();
    float t = 0.f;
    for (int q = 0; q < (int)(blockDim.x >> 6); ++q) t += sh[q];
    __syncthreads();
    return t;
}
__global__ __launch_bounds__(256) void weight_norm_fwd_kernel(const float* __restrict__ v, const float* __restrict__ g, float* __restrict__ w,
                                                              int Ci, int kw) {
    __shared__ float sh[4];
    const int co = blockIdx.x, n = Ci * kw;
    const float* vr = v + (long)co * n;
    float s = 0.f;
    for (int i = threadIdx.x; i < n; i += blockDim.x) s += vr[i] * vr[i];
    const float nrm = sqrtf(block_sum(s, sh));
    const float sc = g[co] / nrm;
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const int ci = i / kw, kk = i - ci * kw;
        w[(long)co * n + (long)kk * Ci + ci] = vr[i] * sc;
    }
}
struct WnBatch { const float* v[8]; const float* g[8]; float* w[8]; float* wt[8]; };
__global__ __launch_bounds__(256) void weight_norm_fwd_batch_kernel(const WnBatch b, int Co, int Ci, int kw) {
    __shared__ float sh[4];
    const int co = blockIdx.x, j = blockIdx.y, n = Ci * kw;
    const float* vr = b.v[j] + (long)co * n;
    float* w = b.w[j];
    float* wt = b.wt[j];
    float s = 0.f;
    for (int i = threadIdx.x; i < n; i += blockDim.x) s += vr[i] * vr[i];
    const float nrm = sqrtf(block_sum(s, sh));
    const float sc = b.g[j][co] / nrm;
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const int ci = i / kw, kk = i - ci * kw;
        const float val = vr[i] * sc;
        w[(long)co * n + (long)kk * Ci + ci] = val;
        if (wt) wt[(long)ci * (kw * Co) + (long)kk * Co + co] = val;
    }
}
__global__ __launch_bounds__(256) void weight_norm_bwd_kernel(const float* __restrict__ dw, const float* __restrict__ v, const float* __restrict__ g,
                                                              float* __restrict__ dg, float* __restrict__ dv, int Ci, int kw) {
    __shared__ float sh[4];
    const int co = blockIdx.x, n = Ci * kw;
    const float* vr = v + (long)co * n;
    float s = 0.f, d = 0.f;
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const int ci = i / kw, kk = i - ci * kw;
        s += vr[i] * vr[i];
        d += dw[(long)co * n + (long)kk * Ci + ci] * vr[i];
    }
    const float n2 = block_sum(s, sh);
    const float dot = block_sum(d, sh);
    const float nrm = sqrtf(n2);
    if (threadIdx.x == 0) dg[co] += dot / nrm;
    const float a = g[co] / nrm, bcoef = dot / n2;
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const int ci = i / kw, kk = i - ci * kw;
        dv[(long)co * n + i] += a * (dw[(long)co * n + (long)kk * Ci + ci] - bcoef * vr[i]);
    }
}

// every weight-normed conv of a network in one launch (blockIdx.y = conv): dw[j] = packed weight gradient of conv j
struct WnBwdBatch { const float* dw[8]; const float* v[8]; const float* g[8]; float* dg[8]; float* dv[8]; };
__global__ __launch_bounds__(256) void weight_norm_bwd_batch_kernel(const WnBwdBatch b, int Ci, int kw) {
    __shared__ float sh[4];
    const int co = blockIdx.x, j = blockIdx.y, n = Ci * kw;
    const float* vr = b.v[j] + (long)co * n;
    const float* dw = b.dw[j] + (long)co * n;
    float s = 0.f, d = 0.f;
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const int ci = i / kw, kk = i - ci * kw;
        s += vr[i] * vr[i];
        d += dw[(long)kk * Ci + ci] * vr[i];
    }
    const float n2 = block_sum(s, sh);
    const float dot = block_sum(d, sh);
    const float nrm = sqrtf(n2);
    if (threadIdx.x == 0) b.dg[j][co] += dot / nrm;
    const float a = b.g[j][co] / nrm, bcoef = dot / n2;
    float* dv = b.dv[j] + (long)co * n;
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const int ci = i / kw, kk = i - ci * kw;
        dv[i] += a * (dw[(long)kk * Ci + ci] - bcoef * vr[i]);
    }
}

// ---- RNG -----------------------------------------------------------------------------------------------------------
__global__ void rng_advance_kernel(uint64_t* st) { st[1] += 1; }
// start of a training iteration: both networks' RNG step counters and the Adam step counters of the optimisers that will step in it, in ONE
// launch (they were four single-thread launches of 4.7 us each inside the captured iteration)
__global__ void iter_begin_kernel(uint64_t* rng_a, uint64_t* rng_b, int32_t* cnt_a, int32_t* cnt_b) {
    if (rng_a) rng_a[1] += 1;
    if (rng_b) rng_b[1] += 1;
    if (cnt_a) *cnt_a += 1;
    if (cnt_b) *cnt_b += 1;
}
// The whole head of a GAN iteration in ONE launch (it was seven: counters, make_pre_seq, randperm, gather, and three ATen copies that
// stacked the inputs of the generator calls -- 4.7 us each inside the captured iteration).  Workgroup 0: the counters of tg_iter_begin, then
// (speaker mode) the permutation of the diversity term drawn from rng_a at its NEW step (or the injected one), and the stacked speaker ids
// [vid] * (copies - 1) + [vid[perm]] (or vid in every copy when nothing is permuted).  Every other workgroup: the stacked seed poses
// (tg_make_pre_seq, `copies` times) and the stacked word ids.
__global__ __launch_bounds__(1024) void iter_head_kernel(uint64_t* rng_a, uint64_t* rng_b, int32_t* cnt_a, int32_t* cnt_b, const float* __restrict__ target,
                                                         float* __restrict__ pre, long pre_ld, int B, int T, int D, int n_pre, int copies,
                                                         const int64_t* __restrict__ text, int64_t* __restrict__ text_s,
                                                         const int64_t* __restrict__ vid, int64_t* __restrict__ vid_s, int permute_last,
                                                         const int64_t* __restrict__ perm_in, uint32_t perm_site, int64_t* __restrict__ perm_out,
                                                         float* __restrict__ target_copy) {
    if (blockIdx.x == 0) {
        __shared__ uint64_t keys[1024];
        __shared__ int perm_sh[1024];
        __shared__ uint64_t s_rng[2];
        const int i = threadIdx.x;
        if (i == 0) {
            // rng_a's new step reaches the other threads of this workgroup through LDS (a __threadfence() here wrote back the XCD's dirty L2
            // lines -- the previous iteration's optimiser step -- at the head of the iteration's critical path)
            if (rng_a) { s_rng[0] = rng_a[0]; s_rng[1] = rng_a[1] + 1; rng_a[1] = s_rng[1]; }
            if (rng_b) rng_b[1] += 1;
            if (cnt_a) *cnt_a += 1;
            if (cnt_b) *cnt_b += 1;
        }
        __syncthreads();
        if (!vid_s) return;
        if (permute_last) {
            if (perm_in) {
                if (i < B) perm_sh[i] = (int)perm_in[i];
            } else {
                if (i < B) {
                    uint32_t r[4];
                    const uint64_t seed = s_rng[0], step = s_rng[1];          // the step thread 0 just wrote
                    philox4x32(seed, (uint64_t)i, perm_site, (uint32_t)step, r);
                    keys[i] = ((uint64_t)r[0] << 32) | r[1];
                }
                __syncthreads();
                if (i < B) {                               // exactly randperm_kernel
                    const uint64_t k = keys[i];
                    int rank = 0;
                    for (int j = 0; j < B; ++j) rank += (keys[j] < k) || (keys[j] == k && j < i);
                    perm_sh[rank] = i;
                }
            }
            __syncthreads();
            if (i < B && perm_out) perm_out[i] = perm_sh[i];
        }
        if (i < B) {
            const int64_t v = vid[i];
            for (int c = 0; c < copies; ++c) {
                int64_t o = v;
                if (permute_last && c == copies - 1) {
                    const int p = perm_sh[i];
                    o = (p >= 0 && p < B) ? vid[p] : 0;        // as gather_i64_kernel
                }
                vid_s[(long)c * B + i] = o;
            }
        }
        return;
    }
    const long per = (long)B * T * (D + 1), n = per * copies;
    const long tid = (long)(blockIdx.x - 1) * blockDim.x + threadIdx.x, nth = (long)(gridDim.x - 1) * blockDim.x;
    for (long i = tid; i < n; i += nth) {
        const long e = i % per;
        const int c = (int)(e % (D + 1));
        const long bt = e / (D + 1);
        const int t = (int)(bt % T);
        pre[(i / (D + 1)) * pre_ld + c] = t < n_pre ? (c < D ? target[bt * D + c] : 1.f) : 0.f;      // row (copy, b, t) of the stacked output
    }
    if (text_s) {
        const long pt = (long)B * T, nt = pt * copies;
        for (long i = tid; i < nt; i += nth) text_s[i] = text[i % pt];
    }
    if (target_copy) {
        const long nc = (long)B * T * D;
        for (long i = tid; i < nc; i += nth) target_copy[i] = target[i];
    }
}
// VEC: n % 4 == 0 and 16-byte aligned pointers (checked by the launcher): one 16-byte access per array and Philox draw instead of four
// scalar ones (the GRU inter-layer dropout moved 93 MB at 2.7 TB/s through the scalar form)
template <bool VEC>
__global__ void dropout_mask_kernel(float* __restrict__ mask, long n, float p, const uint64_t* __restrict__ st, uint32_t site) {
    const uint64_t seed = st[0];
    const uint32_t step = (uint32_t)st[1];
    const float keep = 1.f / (1.f - p);
    const long n4 = (n + 3) / 4;
    GRID_STRIDE(i, n4) {
        uint32_t r[4];
        philox4x32(seed, (uint64_t)i, site, step, r);
        if (VEC) {
            f32x4 m;
#pragma unroll
            for (int q = 0; q < 4; ++q) m[q] = u01(r[q]) >= p ? keep : 0.f;
            reinterpret_cast<f32x4*>(mask)[i] = m;
        } else {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const long e = i * 4 + q;
                if (e < n) mask[e] = u01(r[q]) >= p ? keep : 0.f;
            }
        }
    }
}
// draw the mask and apply it in one pass: y = x * mask (same draws as dropout_mask_kernel for the same state / site)
template <bool VEC>
__global__ void dropout_apply_kernel(const float* __restrict__ x, float* __restrict__ y, float* __restrict__ mask, long n, float p,
                                     const uint64_t* __restrict__ st, uint32_t site) {
    const uint64_t seed = st[0];
    const uint32_t step = (uint32_t)st[1];
    const float keep = 1.f / (1.f - p);
    const long n4 = (n + 3) / 4;
    GRID_STRIDE(i, n4) {
        uint32_t r[4];
        if (VEC) {
            const f32x4 xv = reinterpret_cast<const f32x4*>(x)[i];       // in flight while the ten Philox rounds run
            philox4x32(seed, (uint64_t)i, site, step, r);
            f32x4 m;
#pragma unroll
            for (int q = 0; q < 4; ++q) m[q] = u01(r[q]) >= p ? keep : 0.f;
            if (mask) reinterpret_cast<f32x4*>(mask)[i] = m;       // (NULL: the consumers regenerate it, ops.Drop)
            reinterpret_cast<f32x4*>(y)[i] = xv * m;
        } else {
            philox4x32(seed, (uint64_t)i, site, step, r);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const long e = i * 4 + q;
                if (e < n) {
                    const float m = u01(r[q]) >= p ? keep : 0.f;
                    if (mask) mask[e] = m;
                    y[e] = x[e] * m;
                }
            }
        }
    }
}
__global__ void normal_kernel(float* __restrict__ out, long n, const uint64_t* __restrict__ st, uint32_t site) {
    const uint64_t seed = st[0];
    const uint32_t step = (uint32_t)st[1];
    const long n4 = (n + 3) / 4;
    GRID_STRIDE(i, n4) {
        uint32_t r[4];
        philox4x32(seed, (uint64_t)i, site, step, r);
        float z[4];
#pragma unroll
        for (int q = 0; q < 2; ++q) {   // Box-Muller
            const float rad = sqrtf(-2.f * logf(u01(r[2 * q])));
            const float ang = 6.283185307179586f * u01(r[2 * q + 1]);
            z[2 * q] = rad * cosf(ang);
            z[2 * q + 1] = rad * sinf(ang);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const long e = i * 4 + q;
            if (e < n) out[e] = z[q];
        }
    }
}
__global__ __launch_bounds__(1024) void randperm_kernel(int64_t* __restrict__ out, int n, const uint64_t* __restrict__ st, uint32_t site) {
    __shared__ uint64_t keys[1024];
    const int i = threadIdx.x;
    if (i < n) {
        uint32_t r[4];
        philox4x32(st[0], (uint64_t)i, site, (uint32_t)st[1], r);
        keys[i] = ((uint64_t)r[0] << 32) | r[1];
    }
    __syncthreads();
    if (i < n) {
        const uint64_t k = keys[i];
        int rank = 0;
        for (int j = 0; j < n; ++j) rank += (keys[j] < k) || (keys[j] == k && j < i);
        out[rank] = i;
    }
}
__global__ void gather_i64_kernel(const int64_t* __restrict__ src, const int64_t* __restrict__ perm, int64_t* __restrict__ out, int n) {
    GRID_STRIDE(i, n) {
        const int64_t p = perm[i];
        out[i] = (p >= 0 && p < n) ? src[p] : 0;
    }
}

// ---- speaker path ----------------------------------------------------------------------------------------------------
__global__ void reparam_fwd_kernel(const float* __restrict__ mu, const float* __restrict__ lv, const float* __restrict__ eps, float* __restrict__ z, long n) {
    GRID_STRIDE(i, n) z[i] = mu[i] + eps[i] * expf(0.5f * lv[i]);
}
__global__ void reparam_bwd_kernel(const float* __restrict__ dz, const float* __restrict__ lv, const float* __restrict__ eps, float* __restrict__ dmu,
                                   float* __restrict__ dlv, long n) {
    GRID_STRIDE(i, n) {
        dmu[i] += dz[i];
        dlv[i] += dz[i] * eps[i] * 0.5f * expf(0.5f * lv[i]);
    }
}
__global__ void sigmoid_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ y, float* __restrict__ dx, long n) {
    GRID_STRIDE(i, n) dx[i] = dy[i] * y[i] * (1.f - y[i]);
}
__global__ void sigmoid_kernel(const float* __restrict__ x, float* __restrict__ y, long n) { GRID_STRIDE(i, n) y[i] = sigmoidf_(x[i]); }

// ---- Adam ------------------------------------------------------------------------------------------------------------
__global__ void counter_inc_kernel(int32_t* c) { *c += 1; }
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                                                   long n, float lr, float b1, float b2, float eps, const int32_t* __restrict__ step_dev) {
    __shared__ float coef[2];
    if (threadIdx.x == 0) {   // bias corrections in fp64, like the Python scalars torch.optim.Adam uses
        const double t = (double)*step_dev;
        const double bc1 = 1.0 - pow((double)b1, t), bc2 = 1.0 - pow((double)b2, t);
        coef[0] = (float)((double)lr / bc1);
        coef[1] = (float)sqrt(bc2);
    }
    __syncthreads();
    const float step_size = coef[0], bc2s = coef[1];
    const long n4 = n / 4;
    GRID_STRIDE(i, n4) {
        f32x4 pv = reinterpret_cast<f32x4*>(p)[i], mv = reinterpret_cast<f32x4*>(m)[i], vv = reinterpret_cast<f32x4*>(v)[i];
        const f32x4 gv = reinterpret_cast<const f32x4*>(g)[i];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            mv[q] = mv[q] + (gv[q] - mv[q]) * (1.f - b1);
            vv[q] = vv[q] * b2 + gv[q] * gv[q] * (1.f - b2);
            pv[q] -= step_size * mv[q] / (sqrtf(vv[q]) / bc2s + eps);
        }
        reinterpret_cast<f32x4*>(p)[i] = pv; reinterpret_cast<f32x4*>(m)[i] = mv; reinterpret_cast<f32x4*>(v)[i] = vv;
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {   // tail
        const long i = n4 * 4 + threadIdx.x;
        m[i] = m[i] + (g[i] - m[i]) * (1.f - b1);
        v[i] = v[i] * b2 + g[i] * g[i] * (1.f - b2);
        p[i] -= step_size * m[i] / (sqrtf(v[i]) / bc2s + eps);
    }
}

}  // namespace tg

using namespace tg;
#define ST ((hipStream_t)stream)
#define EW(kernel, n, ...)                                                                      \
    do {                                                                                        \
        if ((n) > 0) hipLaunchKernelGGL(kernel, dim3(ew_grid((n))), dim3(256), 0, ST, __VA_ARGS__); \
    } while (0)

extern "C" {

int tg_add_relu(const float* a, const float* b, float* y, int64_t n, void* stream) {
    TG_REQUIRE(a && b && y && n >= 0, "tg_add_relu: bad arguments");
    if (n % 4 == 0 && aligned16(a) && aligned16(b) && aligned16(y)) EW(add_relu_vkernel<f32x4>, n / 4, (const f32x4*)a, (const f32x4*)b, (f32x4*)y, (long)n / 4);
    else EW(add_relu_vkernel<float>, n, a, b, y, (long)n);
    return check_launch("tg_add_relu");
}
int tg_act_mask_bwd(const float* dy, const float* y, const float* mask, float slope, float* dx, int64_t n, void* stream) {
    TG_REQUIRE(dy && y && dx && n >= 0, "tg_act_mask_bwd: bad arguments");
    const DropSpec none = {nullptr, 0, 0u, 0.f};
    if (n % 4 == 0 && aligned16(dy) && aligned16(y) && aligned16(dx) && (!mask || aligned16(mask)))
        EW(act_mask_bwd_kernel<f32x4>, n / 4, (const f32x4*)dy, (const f32x4*)y, (const f32x4*)mask, slope, (f32x4*)dx, (long)n / 4, none);
    else EW(act_mask_bwd_kernel<float>, n, dy, y, mask, slope, dx, (long)n, none);
    return check_launch("tg_act_mask_bwd");
}
int tg_act_mask_bwd_drop(const float* dy, const float* y, float p, const uint64_t* rng_state, uint32_t site, int64_t index0, float slope, float* dx,
                         int64_t n, void* stream) {
    TG_REQUIRE(dy && y && dx && rng_state && n >= 0 && n % 4 == 0 && index0 >= 0 && index0 % 4 == 0 && p >= 0.f && p < 1.f && aligned16(dy) &&
                   aligned16(y) && aligned16(dx), "tg_act_mask_bwd_drop: bad arguments (n, index0 multiples of 4, 16-byte aligned pointers, 0 <= p < 1)");
    const DropSpec drop = {rng_state, (long)(index0 / 4), site, p};
    EW(act_mask_bwd_kernel<f32x4>, n / 4, (const f32x4*)dy, (const f32x4*)y, (const f32x4*)nullptr, slope, (f32x4*)dx, (long)n / 4, drop);
    return check_launch("tg_act_mask_bwd_drop");
}
int tg_act_mask_bwd2_drop(const float* dy, const float* y, const float* o, float p, const uint64_t* rng_state, uint32_t site, int64_t index0, float slope,
                          float* dsum, float* dc, int64_t n, void* stream) {
    TG_REQUIRE(dy && y && o && dsum && dc && rng_state && n >= 0 && n % 4 == 0 && index0 >= 0 && index0 % 4 == 0 && p >= 0.f && p < 1.f &&
                   aligned16(dy) && aligned16(y) && aligned16(o) && aligned16(dsum) && aligned16(dc),
               "tg_act_mask_bwd2_drop: bad arguments (n, index0 multiples of 4, 16-byte aligned pointers, 0 <= p < 1)");
    const DropSpec drop = {rng_state, (long)(index0 / 4), site, p};
    EW(act_mask_bwd2_kernel<f32x4>, n / 4, (const f32x4*)dy, (const f32x4*)y, (const f32x4*)o, (const f32x4*)nullptr, slope, (f32x4*)dsum, (f32x4*)dc,
       (long)n / 4, drop);
    return check_launch("tg_act_mask_bwd2_drop");
}
int tg_act_mask_bwd2(const float* dy, const float* y, const float* o, const float* mask, float slope, float* dsum, float* dc, int64_t n, void* stream) {
    TG_REQUIRE(dy && y && o && dsum && dc && n >= 0, "tg_act_mask_bwd2: bad arguments");
    if (n % 4 == 0 && aligned16(dy) && aligned16(y) && aligned16(o) && aligned16(dsum) && aligned16(dc) && (!mask || aligned16(mask)))
        EW(act_mask_bwd2_kernel<f32x4>, n / 4, (const f32x4*)dy, (const f32x4*)y, (const f32x4*)o, (const f32x4*)mask, slope, (f32x4*)dsum, (f32x4*)dc, (long)n / 4,
           DropSpec{nullptr, 0, 0u, 0.f});
    else EW(act_mask_bwd2_kernel<float>, n, dy, y, o, mask, slope, dsum, dc, (long)n, DropSpec{nullptr, 0, 0u, 0.f});
    return check_launch("tg_act_mask_bwd2");
}
int tg_zero(void* p, int64_t bytes, void* stream) {
    TG_REQUIRE(p != nullptr && bytes >= 0 && bytes % 4 == 0, "tg_zero: null pointer or size %ld not a non-negative multiple of 4", (long)bytes);
    return zero_async(p, (size_t)bytes, (hipStream_t)stream);
}
int tg_mul(const float* x, const float* mask, float* y, int64_t n, void* stream) {
    TG_REQUIRE(x && mask && y && n >= 0, "tg_mul: bad arguments");
    if (n % 4 == 0 && aligned16(x) && aligned16(mask) && aligned16(y)) EW(mul_kernel<f32x4>, n / 4, (const f32x4*)x, (const f32x4*)mask, (f32x4*)y, (long)n / 4);
    else EW(mul_kernel<float>, n, x, mask, y, (long)n);
    return check_launch("tg_mul");
}
int tg_axpy(const float* x, float* y, float alpha, int32_t accumulate, int64_t n, void* stream) {
    TG_REQUIRE(x && y && n >= 0, "tg_axpy: bad arguments");
    EW(axpy_kernel, n, x, y, alpha, accumulate, (long)n);
    return check_launch("tg_axpy");
}
int tg_copy2d(const float* src, int64_t lds, float* dst, int64_t ldd, int32_t rows, int32_t cols, int32_t accumulate, void* stream) {
    TG_REQUIRE(src && dst && rows > 0 && cols > 0 && lds >= cols && ldd >= cols, "tg_copy2d: bad arguments");
    EW(copy2d_kernel, (long)rows * cols, src, (long)lds, dst, (long)ldd, rows, cols, accumulate);
    return check_launch("tg_copy2d");
}
int tg_repeat_rows(const float* src, int64_t lds, float* dst, int64_t ldd, int32_t B, int32_t T, int32_t cols, void* stream) {
    TG_REQUIRE(src && dst && B > 0 && T > 0 && cols > 0 && lds >= cols && ldd >= cols, "tg_repeat_rows: bad arguments");
    EW(repeat_rows_kernel, (long)B * T * cols, src, (long)lds, dst, (long)ldd, B, T, cols);
    return check_launch("tg_repeat_rows");
}
int tg_sum_rows(const float* src, int64_t lds, float* dst, int64_t ldd, int32_t B, int32_t T, int32_t cols, int32_t accumulate, void* stream) {
    TG_REQUIRE(src && dst && B > 0 && T > 0 && cols > 0 && lds >= cols && ldd >= cols, "tg_sum_rows: bad arguments");
    EW(sum_rows_kernel, (long)B * cols * 8, src, (long)lds, dst, (long)ldd, B, T, cols, accumulate);
    return check_launch("tg_sum_rows");
}
// out [M][8] = a0 [M][K] . w0 [K][8] + a1 [M][K] . w1 [K][8]: the input gradient of a GRU layer with 8 input channels (the discriminator's first
// layer: dx = dgi_fwd W_ih_fwd + dgi_rev W_ih_rev, W_ih stored [3H][8]).  The generic narrow kernel took 13 us per direction on this shape (N = 8
// leaves its 32-wide tiles mostly empty); this is a bandwidth-sized pass: a workgroup takes 32 rows, eight threads per row (one per output
// channel), 16-byte loads of the row shared by the eight, both weight matrices in LDS.
__global__ __launch_bounds__(256) void narrow8_pair_kernel(const float* __restrict__ a0, const float* __restrict__ a1, const float* __restrict__ w0,
                                                           const float* __restrict__ w1, float* __restrict__ out, int M, int K) {
    extern __shared__ __attribute__((aligned(16))) float nsw[];   // [2][K][8]
    for (int i = threadIdx.x; i < K * 8; i += 256) { nsw[i] = w0[i]; nsw[K * 8 + i] = w1[i]; }
    __syncthreads();
    // eight threads per row: thread j takes the 16-byte pieces j, j + 8, .. of the row (the eight together read 128 contiguous bytes per
    // step), forms its partial of all eight outputs, and the eight partials are summed over the lanes (xor 1, 2, 4).  (One output per thread
    // over the whole row issued every load eight times and read the weights as 768 scalar LDS reads: 11.4 us.)
    const int j = threadIdx.x & 7, r = threadIdx.x >> 3;
    long row = (long)blockIdx.x * 32 + r;
    const bool live = row < M;
    if (!live) row = M - 1;
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int d = 0; d < 2; ++d) {
        const f32x4* ap = reinterpret_cast<const f32x4*>((d ? a1 : a0) + row * K);
        const float* w = nsw + d * K * 8;
#pragma unroll 4
        for (int k4 = j; k4 < K / 4; k4 += 8) {
            const f32x4 v = ap[k4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const f32x4 wa = *reinterpret_cast<const f32x4*>(w + (4 * k4 + e) * 8), wb = *reinterpret_cast<const f32x4*>(w + (4 * k4 + e) * 8 + 4);
#pragma unroll
                for (int c = 0; c < 4; ++c) { acc[c] = __builtin_fmaf(v[e], wa[c], acc[c]); acc[4 + c] = __builtin_fmaf(v[e], wb[c], acc[4 + c]); }
            }
        }
    }
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        acc[c] += __shfl_xor(acc[c], 1);
        acc[c] += __shfl_xor(acc[c], 2);
        acc[c] += __shfl_xor(acc[c], 4);
    }
    if (live) {
        float v = acc[0];
#pragma unroll
        for (int c = 1; c < 8; ++c) v = j == c ? acc[c] : v;
        out[row * 8 + j] = v;
    }
}
int tg_narrow8_pair(const float* a0, const float* a1, const float* w0, const float* w1, float* out, int32_t M, int32_t K, void* stream) {
    TG_REQUIRE(a0 && a1 && w0 && w1 && out && M > 0 && K > 0 && K % 4 == 0 && K <= 2048 && aligned16(a0) && aligned16(a1),
               "tg_narrow8_pair: bad arguments (K a multiple of 4, <= 2048; 16-byte aligned a0 / a1)");
    hipLaunchKernelGGL(narrow8_pair_kernel, dim3((M + 31) / 32), dim3(256), (size_t)2 * K * 8 * sizeof(float), ST, a0, a1, w0, w1, out, M, K);
    return check_launch("tg_narrow8_pair");
}
// out = p[0] + p[1] + .. + p[parts - 1], `parts` arrays of n floats `stride` floats apart (the K-split partial products of a many-slab input
// gradient, combined in a fixed order)
__global__ void sum_parts_kernel(const float* __restrict__ p, long stride, int parts, float* __restrict__ out, long n4) {
    GRID_STRIDE(i, n4) {
        f32x4 acc = reinterpret_cast<const f32x4*>(p)[i];
        for (int q = 1; q < parts; ++q) acc += reinterpret_cast<const f32x4*>(p + q * stride)[i];
        reinterpret_cast<f32x4*>(out)[i] = acc;
    }
}
int tg_sum_parts(const float* parts, int64_t stride, int32_t n_parts, float* out, int64_t n, void* stream) {
    TG_REQUIRE(parts && out && n_parts >= 1 && n > 0 && n % 4 == 0 && stride % 4 == 0 && stride >= n && aligned16(parts) && aligned16(out),
               "tg_sum_parts: bad arguments (n, stride multiples of 4, 16-byte aligned)");
    EW(sum_parts_kernel, n / 4, parts, (long)stride, n_parts, out, (long)n / 4);
    return check_launch("tg_sum_parts");
}
int tg_add_halves(const float* y, float* o, int32_t M, int32_t H, void* stream) {
    TG_REQUIRE(y && o && M > 0 && H > 0, "tg_add_halves: bad arguments");
    EW(add_halves_kernel, (long)M * H, y, o, (long)M, H);
    return check_launch("tg_add_halves");
}
int tg_dup_halves(const float* d_o, float* dy, int32_t M, int32_t H, void* stream) {
    TG_REQUIRE(d_o && dy && M > 0 && H > 0, "tg_dup_halves: bad arguments");
    EW(dup_halves_kernel, (long)M * H, d_o, dy, (long)M, H);
    return check_launch("tg_dup_halves");
}
int tg_make_pre_seq(const float* target, float* pre, int32_t B, int32_t T, int32_t D, int32_t n_pre, void* stream) {
    TG_REQUIRE(target && pre && B > 0 && T > 0 && D > 0 && n_pre >= 0 && n_pre <= T, "tg_make_pre_seq: bad arguments");
    EW(make_pre_seq_kernel, (long)B * T * (D + 1), target, pre, B, T, D, n_pre);
    return check_launch("tg_make_pre_seq");
}
int tg_embed_gather(const float* table, const int64_t* idx, float* out, int32_t n_idx, int32_t D, int32_t n_rows, void* stream) {
    TG_REQUIRE(table && idx && out && n_idx > 0 && D > 0 && n_rows > 0, "tg_embed_gather: bad arguments");
    EW(embed_gather_kernel, (long)n_idx * D, table, idx, out, n_idx, D, n_rows);
    return check_launch("tg_embed_gather");
}
int tg_embed_gather_drop(const float* table, const int64_t* idx, float* out, int32_t n_idx, int32_t D, int32_t n_rows, float p, const uint64_t* rng_state,
                         uint32_t site, void* stream) {
    TG_REQUIRE(table && idx && out && rng_state && n_idx > 0 && D > 0 && D % 4 == 0 && n_rows > 0 && p >= 0.f && p < 1.f && aligned16(table) && aligned16(out),
               "tg_embed_gather_drop: bad arguments (D %% 4 == 0, 16-byte aligned table / out, 0 <= p < 1)");
    EW(embed_gather_drop_kernel, (long)n_idx * (D / 4), table, idx, out, n_idx, D, n_rows, p, rng_state, site);
    return check_launch("tg_embed_gather_drop");
}
int tg_embed_scatter_add(const float* dout, const int64_t* idx, float* dtable, int32_t n_idx, int32_t D, int32_t n_rows, void* stream) {
    TG_REQUIRE(dout && idx && dtable && n_idx > 0 && D > 0 && n_rows > 0, "tg_embed_scatter_add: bad arguments");
    if (deterministic()) {
        TG_REQUIRE(D <= 512, "tg_embed_scatter_add (deterministic): D = %d > 512", D);
        hipLaunchKernelGGL(embed_scatter_det_kernel, dim3(n_idx), dim3(64 * SCATTER_DET_WAVES), 0, ST, dout, idx, dtable, n_idx, D, n_rows);
        return check_launch("tg_embed_scatter_add(deterministic)");
    }
    hipLaunchKernelGGL(embed_scatter_kernel, dim3(cdiv(n_idx, SCATTER_CHUNK)), dim3(256), 0, ST, dout, idx, dtable, n_idx, D, n_rows);
    return check_launch("tg_embed_scatter_add");
}
int tg_permute3(const float* in, float* out, int32_t d0, int32_t d1, int32_t d2, int32_t p0, int32_t p1, int32_t p2, void* stream) {
    TG_REQUIRE(in && out && d0 > 0 && d1 > 0 && d2 > 0, "tg_permute3: bad sizes");
    TG_REQUIRE(p0 >= 0 && p0 < 3 && p1 >= 0 && p1 < 3 && p2 >= 0 && p2 < 3 && p0 != p1 && p0 != p2 && p1 != p2, "tg_permute3: bad permutation");
    EW(permute3_kernel, (long)d0 * d1 * d2, in, out, d0, d1, d2, p0, p1, p2);
    return check_launch("tg_permute3");
}
int tg_permute3_batch(const int64_t* desc, int32_t n_jobs, int32_t total_workgroups, void* stream) {
    TG_REQUIRE(desc && n_jobs > 0 && total_workgroups > 0, "tg_permute3_batch: bad arguments");
    hipLaunchKernelGGL(permute3_batch_kernel, dim3(total_workgroups), dim3(256), 0, ST, reinterpret_cast<const long*>(desc), n_jobs);
    return check_launch("tg_permute3_batch");
}
int tg_conv_dgrad_pack(const float* w, float* out, int32_t Co, int32_t Ci, int32_t kw, int32_t s, void* stream) {
    TG_REQUIRE(w && out && Co > 0 && Ci > 0 && kw > 0 && s > 0, "tg_conv_dgrad_pack: bad arguments");
    const int J = (kw + s - 1) / s;
    EW(conv_dgrad_pack_kernel, (long)s * Ci * J * Co, w, out, Co, Ci, kw, s, J);
    return check_launch("tg_conv_dgrad_pack");
}
int tg_weight_norm_fwd(const float* v, const float* g, float* w_packed, int32_t Co, int32_t Ci, int32_t kw, void* stream) {
    TG_REQUIRE(v && g && w_packed && Co > 0 && Ci > 0 && kw > 0, "tg_weight_norm_fwd: bad arguments");
    hipLaunchKernelGGL(weight_norm_fwd_kernel, dim3(Co), dim3(256), 0, ST, v, g, w_packed, Ci, kw);
    return check_launch("tg_weight_norm_fwd");
}
int tg_weight_norm_fwd_batch(int32_t n, const float* const* v, const float* const* g, float* const* w_packed, float* const* w_t,
                             int32_t Co, int32_t Ci, int32_t kw, void* stream) {
    TG_REQUIRE(n >= 1 && n <= 8 && v && g && w_packed && Co > 0 && Ci > 0 && kw > 0, "tg_weight_norm_fwd_batch: 1..8 convs, non-null tables");
    WnBatch b;
    for (int i = 0; i < 8; ++i) {
        const int j = i < n ? i : 0;
        TG_REQUIRE(v[j] && g[j] && w_packed[j], "tg_weight_norm_fwd_batch: null entry %d", j);
        b.v[i] = v[j]; b.g[i] = g[j]; b.w[i] = w_packed[j]; b.wt[i] = w_t ? w_t[j] : nullptr;
    }
    hipLaunchKernelGGL(weight_norm_fwd_batch_kernel, dim3(Co, n), dim3(256), 0, ST, b, Co, Ci, kw);
    return check_launch("tg_weight_norm_fwd_batch");
}
int tg_weight_norm_bwd(const float* dw_packed, const float* v, const float* g, float* dg, float* dv, int32_t Co, int32_t Ci, int32_t kw, void* stream) {
    TG_REQUIRE(dw_packed && v && g && dg && dv && Co > 0 && Ci > 0 && kw > 0, "tg_weight_norm_bwd: bad arguments");
    hipLaunchKernelGGL(weight_norm_bwd_kernel, dim3(Co), dim3(256), 0, ST, dw_packed, v, g, dg, dv, Ci, kw);
    return check_launch("tg_weight_norm_bwd");
}
int tg_weight_norm_bwd_batch(int32_t n, const float* const* dw_packed, const float* const* v, const float* const* g, float* const* dg,
                             float* const* dv, int32_t Co, int32_t Ci, int32_t kw, void* stream) {
    TG_REQUIRE(n >= 1 && n <= 8 && dw_packed && v && g && dg && dv && Co > 0 && Ci > 0 && kw > 0, "tg_weight_norm_bwd_batch: 1..8 convs, non-null tables");
    WnBwdBatch b;
    for (int i = 0; i < 8; ++i) {
        const int j = i < n ? i : 0;
        TG_REQUIRE(dw_packed[j] && v[j] && g[j] && dg[j] && dv[j], "tg_weight_norm_bwd_batch: null entry %d", j);
        b.dw[i] = dw_packed[j]; b.v[i] = v[j]; b.g[i] = g[j]; b.dg[i] = dg[j]; b.dv[i] = dv[j];
    }
    hipLaunchKernelGGL(weight_norm_bwd_batch_kernel, dim3(Co, n), dim3(256), 0, ST, b, Ci, kw);
    return check_launch("tg_weight_norm_bwd_batch");
}
int tg_iter_begin(uint64_t* rng_a, uint64_t* rng_b, int32_t* adam_step_a, int32_t* adam_step_b, void* stream) {
    TG_REQUIRE(rng_a || rng_b || adam_step_a || adam_step_b, "tg_iter_begin: nothing to advance");
    hipLaunchKernelGGL(iter_begin_kernel, dim3(1), dim3(1), 0, ST, rng_a, rng_b, adam_step_a, adam_step_b);
    return check_launch("tg_iter_begin");
}
int tg_iter_head(uint64_t* rng_a, uint64_t* rng_b, int32_t* adam_step_a, int32_t* adam_step_b, const float* target, float* pre_stacked, int64_t pre_ld,
                 int32_t B, int32_t T, int32_t D, int32_t n_pre, int32_t copies, const int64_t* text, int64_t* text_stacked, const int64_t* vid,
                 int64_t* vid_stacked, int32_t permute_last, const int64_t* perm_in, uint32_t perm_site, int64_t* perm_out, float* target_copy,
                 void* stream) {
    TG_REQUIRE(target && pre_stacked && B > 0 && T > 0 && D > 0 && n_pre >= 0 && copies >= 1 && pre_ld >= D + 1, "tg_iter_head: bad arguments");
    TG_REQUIRE((text == nullptr) == (text_stacked == nullptr) && (vid == nullptr) == (vid_stacked == nullptr), "tg_iter_head: text / vid and their stacked outputs go together");
    TG_REQUIRE(!vid || B <= 1024, "tg_iter_head: B=%d speaker ids must fit one workgroup (<= 1024)", B);
    TG_REQUIRE(!permute_last || (vid && (perm_in || rng_a)), "tg_iter_head: a drawn permutation needs vid and rng_a");
    const long n = (long)B * T * (D + 1) * copies;
    const int blocks = 1 + (int)((n + 4095) / 4096 < 255 ? (n + 4095) / 4096 : 255);
    hipLaunchKernelGGL(iter_head_kernel, dim3(blocks < 2 ? 2 : blocks), dim3(1024), 0, ST, rng_a, rng_b, adam_step_a, adam_step_b, target, pre_stacked, (long)pre_ld, B, T, D,
                       n_pre, copies, text, text_stacked, vid, vid_stacked, permute_last, perm_in, perm_site, perm_out, target_copy);
    return check_launch("tg_iter_head");
}
int tg_rng_advance(uint64_t* rng_state, void* stream) {
    TG_REQUIRE(rng_state, "tg_rng_advance: null");
    hipLaunchKernelGGL(rng_advance_kernel, dim3(1), dim3(1), 0, ST, rng_state);
    return check_launch("tg_rng_advance");
}
int tg_dropout_mask(float* mask, int64_t n, float p, const uint64_t* rng_state, uint32_t site, void* stream) {
    TG_REQUIRE(mask && rng_state && n >= 0 && p >= 0.f && p < 1.f, "tg_dropout_mask: bad arguments");
    if (n % 4 == 0 && aligned16(mask)) EW(dropout_mask_kernel<true>, n / 4, mask, (long)n, p, rng_state, site);
    else EW(dropout_mask_kernel<false>, (n + 3) / 4, mask, (long)n, p, rng_state, site);
    return check_launch("tg_dropout_mask");
}
int tg_dropout_apply(const float* x, float* y, float* mask, int64_t n, float p, const uint64_t* rng_state, uint32_t site, void* stream) {
    TG_REQUIRE(x && y && rng_state && n >= 0 && p >= 0.f && p < 1.f, "tg_dropout_apply: bad arguments");
    if (n % 4 == 0 && aligned16(x) && aligned16(y) && aligned16(mask)) EW(dropout_apply_kernel<true>, n / 4, x, y, mask, (long)n, p, rng_state, site);
    else EW(dropout_apply_kernel<false>, (n + 3) / 4, x, y, mask, (long)n, p, rng_state, site);
    return check_launch("tg_dropout_apply");
}
int tg_normal(float* out, int64_t n, const uint64_t* rng_state, uint32_t site, void* stream) {
    TG_REQUIRE(out && rng_state && n >= 0, "tg_normal: bad arguments");
    EW(normal_kernel, (n + 3) / 4, out, (long)n, rng_state, site);
    return check_launch("tg_normal");
}
int tg_randperm(int64_t* out, int32_t n, const uint64_t* rng_state, uint32_t site, void* stream) {
    TG_REQUIRE(out && rng_state && n > 0 && n <= 1024, "tg_randperm: n=%d must be in [1,1024]", n);
    hipLaunchKernelGGL(randperm_kernel, dim3(1), dim3(1024), 0, ST, out, n, rng_state, site);
    return check_launch("tg_randperm");
}
int tg_gather_i64(const int64_t* src, const int64_t* perm, int64_t* out, int32_t n, void* stream) {
    TG_REQUIRE(src && perm && out && n > 0, "tg_gather_i64: bad arguments");
    EW(gather_i64_kernel, (long)n, src, perm, out, n);
    return check_launch("tg_gather_i64");
}
int tg_reparam_fwd(const float* mu, const float* logvar, const float* eps, float* z, int64_t n, void* stream) {
    TG_REQUIRE(mu && logvar && eps && z && n >= 0, "tg_reparam_fwd: bad arguments");
    EW(reparam_fwd_kernel, n, mu, logvar, eps, z, (long)n);
    return check_launch("tg_reparam_fwd");
}
int tg_reparam_bwd(const float* dz, const float* logvar, const float* eps, float* dmu, float* dlogvar, int64_t n, void* stream) {
    TG_REQUIRE(dz && logvar && eps && dmu && dlogvar && n >= 0, "tg_reparam_bwd: bad arguments");
    EW(reparam_bwd_kernel, n, dz, logvar, eps, dmu, dlogvar, (long)n);
    return check_launch("tg_reparam_bwd");
}
int tg_sigmoid(const float* x, float* y, int64_t n, void* stream) {
    TG_REQUIRE(x && y && n >= 0, "tg_sigmoid: bad arguments");
    EW(sigmoid_kernel, n, x, y, (long)n);
    return check_launch("tg_sigmoid");
}
int tg_sigmoid_bwd(const float* dy, const float* y, float* dx, int64_t n, void* stream) {
    TG_REQUIRE(dy && y && dx && n >= 0, "tg_sigmoid_bwd: bad arguments");
    EW(sigmoid_bwd_kernel, n, dy, y, dx, (long)n);
    return check_launch("tg_sigmoid_bwd");
}
int tg_counter_inc(int32_t* counter, void* stream) {
    TG_REQUIRE(counter, "tg_counter_inc: null");
    hipLaunchKernelGGL(counter_inc_kernel, dim3(1), dim3(1), 0, ST, counter);
    return check_launch("tg_counter_inc");
}
int tg_adam_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2, float eps,
                 const int32_t* step_dev, void* stream) {
    TG_REQUIRE(p && g && m && v && step_dev && n > 0, "tg_adam_step: bad arguments");
    TG_REQUIRE(aligned16(p) && aligned16(g) && aligned16(m) && aligned16(v), "tg_adam_step: slabs must be 16-byte aligned");
    hipLaunchKernelGGL(adam_kernel, dim3(ew_grid(n / 4 + 1, 256, 4)), dim3(256), 0, ST, p, g, m, v, (long)n, lr, beta1, beta2, eps, step_dev);
    return check_launch("tg_adam_step");
}

}  // extern "C"
