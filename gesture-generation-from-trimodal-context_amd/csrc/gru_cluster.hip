// Persistent, cluster-synchronised GRU recurrence for the generator (H <= 320): ONE launch walks all T steps of both
// directions, replacing T per-step launches (gru.hip) whose ~14 us were mostly dispatch, cold-miss and drain latency.
//
// The recurrence of one (direction, batch tile) never needs another tile's data, so there is no grid-wide barrier: the
// CW = ceil(H / 32) workgroups that share a batch tile form a CLUSTER.  Workgroup m of a cluster owns hidden units
// [32m, 32m + 32) of all three gates and keeps its 96 rows of W_hh in REGISTERS as MFMA B-fragments for the whole
// sequence (8 waves = 2 unit tiles x 4 K-slices, 15 float4 per lane).  Per step it
//   1. waits until every member of its cluster has published h_{t-1}                      (one wave polls CW flag words),
//   2. loads the h_{t-1} tile [16*MT rows][H] as MFMA A-fragments straight from the exchange buffer (sc1 loads),
//   3. 60*MT MFMAs per wave, K-slice partials meet in LDS,
//   4. fused gate epilogue on 4 consecutive hidden units per thread, h_t -> exchange buffer with 16-byte write-through
//      (sc1) stores, every wave drains its stores, barrier, ONE lane publishes flag = step + 1,
//   5. y / saved gates go out as plain stores behind the publish (off the critical path).
// Hand-off protocol: write-through payload + drained flag, sc1 loads on the consumer (MI355X_MICROARCH.md, inter-workgroup
// visibility, table row 1; cdna_hip_programming.md Guideline 16 R1).  The exchange buffer is double-buffered by step parity and
// laid out so that every 128-byte line is written whole by one store instruction of one wave.
// Residency: 512-thread workgroups with > 128 VGPRs -> one per CU; the host launches at most 256 of them (all co-resident on
// an otherwise in-order stream).  Every spin is bounded: on a timeout the workgroup sets the timeout word, stops waiting for
// the rest of the sequence and runs to completion (results are then garbage and the host raises on the timeout word).
#include <stdlib.h>

#include "common.hpp"

namespace tg {

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) unsigned gu32;

constexpr int GC_UNITS = 32;          // hidden units per workgroup
constexpr int GC_HX = 320;            // exchange-buffer row stride (floats): 10 x 128-byte lines
constexpr int GC_FLAG_STRIDE = 16;    // flag words per cluster (one 64-byte line)
constexpr int GC_KS = 4;
constexpr int GC_PF = 5;              // K chunks (of 16) per wave: covers H <= 320
// Bound of every spin: 2^26 polls (>= 20 ns each: 1.3 s or more).  A wait is normally < 10 us; it gets long only when another
// queue's kernels (e.g. a blit copy on a copy stream) keep some cluster members from becoming resident for a while -- that is a
// delay, not a dead-lock, and must not invalidate the results (a 2^19-poll bound, ~8 ms, did trip beside a slow host-to-device
// copy); only a member that can never run trips this one.
constexpr unsigned GC_SPIN_LIMIT = 1u << 26;
constexpr unsigned GC_RSRC3 = 0x00020000u;   // gfx9 raw buffer descriptor word 3 (32-bit data format)

__device__ __forceinline__ f32x4 as_f32x4(u32x4 v) { return __builtin_bit_cast(f32x4, v); }
__device__ __forceinline__ u32x4 as_u32x4(f32x4 v) { return __builtin_bit_cast(u32x4, v); }

// flags[cluster * 16 + member] = number of steps that member has published; word 0 of `tmo` = timeout marker
template <int MT>
__global__ __launch_bounds__(512) void gru_seq_fwd_cluster_kernel(
    const float* __restrict__ gi, long gi_ds, const float* __restrict__ whh0, const float* __restrict__ whh1,
    const float* __restrict__ bhh0, const float* __restrict__ bhh1, float* __restrict__ Y, float* __restrict__ save, long save_ds,
    float* hx, unsigned* flags, unsigned* tmo, int B, int T, int H, int n_bt, int CW, int b_pad) {
    __shared__ __attribute__((aligned(16))) float red[GC_KS][2][MT][3][4][64];
    const int n_cl = 2 * n_bt;
    int cl, m;
    if (n_cl % 8 == 0) {        // members of one cluster on block ids of one residue mod 8: same XCD as observed (speed only)
        cl = (blockIdx.x % 8) + 8 * ((blockIdx.x / 8) / CW);
        m = (blockIdx.x / 8) % CW;
    } else {
        cl = blockIdx.x / CW;
        m = blockIdx.x % CW;
    }
    const int dir = cl / n_bt, bt = cl % n_bt;
    const float* whh = dir ? whh1 : whh0;
    const float* bhh = dir ? bhh1 : bhh0;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int ut = wave & 1, ks = wave >> 1;
    const int r16 = lane & 15, kq = lane >> 4;
    const int b0 = bt * (16 * MT);

    // ---- W_hh slice -> registers (B operand: lane (r16, kq) holds W[g*H + j][16c + 4kq + v])
    f32x4 w[3][GC_PF];
    {
        const int j = m * GC_UNITS + ut * 16 + r16;
#pragma unroll
        for (int p = 0; p < GC_PF; ++p) {
            const int k = 16 * (ks + GC_KS * p) + 4 * kq;
#pragma unroll
            for (int g = 0; g < 3; ++g) {
                f32x4 z = {0.f, 0.f, 0.f, 0.f};
                w[g][p] = (j < H && k < H) ? *reinterpret_cast<const f32x4*>(whh + (long)(g * H + j) * H + k) : z;
            }
        }
    }
    // ---- epilogue role: thread e < 128*MT finalises row (e / 8) of the tile, hidden units 4*(e % 8) .. +3 of the slice
    const int e = threadIdx.x;
    const bool epi = e < 128 * MT;
    const int row_l = e >> 3, ug = e & 7;
    const int e_mt = (row_l >> 4) % MT, e_lane = ((row_l & 15) >> 2) * 16 + 4 * (ug & 3), e_i = row_l & 3, e_ut = ug >> 2;
    const int row = b0 + row_l;
    const int unit0 = m * GC_UNITS + 4 * ug;
    const bool e_ok = epi && row < B && unit0 < H;           // H % 4 == 0: the four units are valid together
    f32x4 bh[3];
#pragma unroll
    for (int g = 0; g < 3; ++g) {
        f32x4 z = {0.f, 0.f, 0.f, 0.f};
        bh[g] = (epi && unit0 < H) ? *reinterpret_cast<const f32x4*>(bhh + g * H + unit0) : z;
    }
    f32x4 hp = {0.f, 0.f, 0.f, 0.f};

    const long slot_floats = (long)b_pad * GC_HX;
    __amdgpu_buffer_rsrc_t hx_rsrc = __builtin_amdgcn_make_buffer_rsrc(hx, 0, (int)(4 * slot_floats * 4), GC_RSRC3);
    gu32* my_flag = (gu32*)(flags + cl * GC_FLAG_STRIDE + m);
    gu32* cl_flags = (gu32*)(flags + cl * GC_FLAG_STRIDE);
    bool aborted = false;

    for (int step = 0; step < T; ++step) {
        const int tau = dir ? T - 1 - step : step;
        // input-side pre-activations do not depend on the recurrence: issue their loads before the wait
        f32x4 gv[3];
#pragma unroll
        for (int g = 0; g < 3; ++g) {
            f32x4 z = {0.f, 0.f, 0.f, 0.f};
            gv[g] = e_ok ? *reinterpret_cast<const f32x4*>(gi + dir * gi_ds + ((long)row * T + tau) * (3 * H) + g * H + unit0) : z;
        }
        f32x4 acc[MT][3];
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int g = 0; g < 3; ++g) acc[i][g] = f32x4{0.f, 0.f, 0.f, 0.f};

        if (step > 0) {
            if (wave == 0 && !aborted) {           // ONE wave polls the cluster's flag words, relaxed, bounded
                unsigned spins = 0;
                for (;;) {
                    const unsigned v = lane < CW ? __hip_atomic_load(cl_flags + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0xffffffffu;
                    if (__all(v >= (unsigned)step)) break;
                    __builtin_amdgcn_s_sleep(1);
                    if (++spins > GC_SPIN_LIMIT) {                    // wave-uniform
                        if (lane == 0) {      // who / when (words 1, 2) and the flag words it saw (4..): diagnostics for the host
                            __hip_atomic_store((gu32*)tmo + 1, (unsigned)step, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            __hip_atomic_store((gu32*)tmo + 2, (unsigned)blockIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        }
                        if (lane < CW) __hip_atomic_store((gu32*)tmo + 4 + lane, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        if (lane == 0) __hip_atomic_store((gu32*)tmo, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        aborted = true;
                        break;
                    }
                }
            }
            __syncthreads();                        // the other waves load only behind the polling wave's barrier
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");     // compiler ordering only; every load below is sc1
            const int rslot = (step - 1) & 1;
            const int off0 = (int)(((long)(dir * 2 + rslot) * slot_floats) * 4);
            f32x4 a[MT][GC_PF];
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int p = 0; p < GC_PF; ++p) {
                    const int k = 16 * (ks + GC_KS * p) + 4 * kq;          // < 320 = GC_HX always
                    const int off = off0 + ((b0 + i * 16 + r16) * GC_HX + k) * 4;
                    // columns >= 32 * CW are written by nobody (H <= 288): never feed scratch bit patterns (NaN * 0) to the MFMA
                    const f32x4 ld = as_f32x4(__builtin_amdgcn_raw_buffer_load_b128(hx_rsrc, off, 0, 16));   // aux 16 = sc1
                    a[i][p] = (k < GC_UNITS * CW) ? ld : f32x4{0.f, 0.f, 0.f, 0.f};
                }
#pragma unroll
            for (int p = 0; p < GC_PF; ++p)
#pragma unroll
                for (int v = 0; v < 4; ++v)
#pragma unroll
                    for (int i = 0; i < MT; ++i)
#pragma unroll
                        for (int g = 0; g < 3; ++g)
                            acc[i][g] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i][p][v], w[g][p][v], acc[i][g], 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int g = 0; g < 3; ++g)
#pragma unroll
                for (int q = 0; q < 4; ++q) red[ks][ut][i][g][q][lane] = acc[i][g][q];
        __syncthreads();

        f32x4 h = {0.f, 0.f, 0.f, 0.f}, r4, z4, n4, hn4;
        if (epi) {
            f32x4 gh[3];
#pragma unroll
            for (int g = 0; g < 3; ++g) {
                f32x4 s = *reinterpret_cast<const f32x4*>(&red[0][e_ut][e_mt][g][e_i][e_lane]);
#pragma unroll
                for (int q = 1; q < GC_KS; ++q) s += *reinterpret_cast<const f32x4*>(&red[q][e_ut][e_mt][g][e_i][e_lane]);
                gh[g] = s;
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float hn = gh[2][q] + bh[2][q];
                const float r = gate_sigmoid(gv[0][q] + gh[0][q] + bh[0][q]);
                const float z = gate_sigmoid(gv[1][q] + gh[1][q] + bh[1][q]);
                const float n = gate_tanh(gv[2][q] + r * hn);
                h[q] = (1.f - z) * n + z * hp[q];
                r4[q] = r; z4[q] = z; n4[q] = n; hn4[q] = hn;
            }
            hp = h;
            // publish h_t: 16-byte write-through store; 8 consecutive lanes write one whole 128-byte line
            const int woff = (int)((((long)(dir * 2 + (step & 1)) * slot_floats) + (long)row * GC_HX + unit0) * 4);
            __builtin_amdgcn_raw_buffer_store_b128(as_u32x4(h), hx_rsrc, woff, 0, 16);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // EVERY wave drains its stores before the flag
        __syncthreads();                                       // (also: `red` is free again)
        if (threadIdx.x == 0) __hip_atomic_store(my_flag, (unsigned)(step + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (e_ok) {                                            // outputs for later kernels: plain stores, off the critical path
            *reinterpret_cast<f32x4*>(Y + ((long)row * T + tau) * (2 * H) + dir * H + unit0) = h;
            if (save) {
                float* sp = save + dir * save_ds + ((long)row * T + tau) * (4 * H) + unit0;
                *reinterpret_cast<f32x4*>(sp) = r4;
                *reinterpret_cast<f32x4*>(sp + H) = z4;
                *reinterpret_cast<f32x4*>(sp + 2 * H) = n4;
                *reinterpret_cast<f32x4*>(sp + 3 * H) = hn4;
            }
        }
    }
}


// ------------------------------------------------------------------------------------------------------ backward
// Reverse-time walk with the same cluster structure.  Workgroup m owns hidden units [32m, 32m + 32):
//     dh_tau = dy_tau + dh_next * z_next + dgh_next @ W_hh            (contraction over the 3H gate rows)
//     dn = dh (1-z)(1-n^2),  dz = dh (h_prev - n) z (1-z),  dr = dn * hn * r (1-r);  dgi = [dr, dz, dn], dgh = [dr, dz, dn r]
// W_hh^T slice [32 units][3H] lives in registers (15 float4 per lane: 8 waves = 2 unit tiles x 4 K-slices of the 57 chunks);
// the hand-off payload is the step's dgh tile, exchanged gate-major ([row][gate][320]) so that every 128-byte line is
// written by one workgroup.  dh_next and z_next belong to the thread that produced them and stay in registers.
constexpr int GC_PFB = 15;            // K chunks per wave in the backward product: 4 x 15 x 16 >= 3 x 320

template <int MT>
__global__ __launch_bounds__(512) void gru_seq_bwd_cluster_kernel(
    const float* __restrict__ dY, const float* __restrict__ Y, const float* __restrict__ save, long save_ds,
    const float* __restrict__ wt0, const float* __restrict__ wt1, float* __restrict__ dgi, float* __restrict__ dgh, long dg_ds,
    float* gx, unsigned* flags, unsigned* tmo, int B, int T, int H, int n_bt, int CW, int b_pad) {
    __shared__ __attribute__((aligned(16))) float red[GC_KS][2][MT][4][64];
    const int n_cl = 2 * n_bt;
    int cl, m;
    if (n_cl % 8 == 0) {
        cl = (blockIdx.x % 8) + 8 * ((blockIdx.x / 8) / CW);
        m = (blockIdx.x / 8) % CW;
    } else {
        cl = blockIdx.x / CW;
        m = blockIdx.x % CW;
    }
    const int dir = cl / n_bt, bt = cl % n_bt;
    const float* wt = dir ? wt1 : wt0;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int ut = wave & 1, ks = wave >> 1;
    const int r16 = lane & 15, kq = lane >> 4;
    const int b0 = bt * (16 * MT);
    const int H3 = 3 * H;

    // ---- W_hh^T slice -> registers (B operand: lane (r16, kq) holds W_hh[k][j] = wt[j][k], k = 16c + 4kq + v), and the
    // exchange-buffer offset of the same k (gate-major, padded rows)
    f32x4 w[GC_PFB];
    int koff[GC_PFB];
    {
        const int j = m * GC_UNITS + ut * 16 + r16;
#pragma unroll
        for (int p = 0; p < GC_PFB; ++p) {
            const int k = 16 * (ks + GC_KS * p) + 4 * kq;
            f32x4 z = {0.f, 0.f, 0.f, 0.f};
            w[p] = (j < H && k < H3) ? *reinterpret_cast<const f32x4*>(wt + (long)j * H3 + k) : z;
            const int kc = k < H3 ? k : 0;                 // beyond 3H the weight fragment is zero: any finite operand will do
            koff[p] = (kc / H) * GC_HX + (kc % H);
        }
    }
    const int e = threadIdx.x;
    const bool epi = e < 128 * MT;
    const int row_l = e >> 3, ug = e & 7;
    const int e_mt = (row_l >> 4) % MT, e_lane = ((row_l & 15) >> 2) * 16 + 4 * (ug & 3), e_i = row_l & 3, e_ut = ug >> 2;
    const int row = b0 + row_l;
    const int unit0 = m * GC_UNITS + 4 * ug;
    const bool e_ok = epi && row < B && unit0 < H;
    f32x4 dh_c = {0.f, 0.f, 0.f, 0.f}, z_c = {0.f, 0.f, 0.f, 0.f};

    const long slot_floats = (long)b_pad * 3 * GC_HX;
    __amdgpu_buffer_rsrc_t gx_rsrc = __builtin_amdgcn_make_buffer_rsrc(gx, 0, (int)(4 * slot_floats * 4), GC_RSRC3);
    gu32* my_flag = (gu32*)(flags + cl * GC_FLAG_STRIDE + m);
    gu32* cl_flags = (gu32*)(flags + cl * GC_FLAG_STRIDE);
    bool aborted = false;

    for (int step = 0; step < T; ++step) {
        const int tau = dir ? step : T - 1 - step;
        const int tau_prev = dir ? tau + 1 : tau - 1;           // producer of h_prev for this cell
        const bool has_prev = dir ? (tau < T - 1) : (tau > 0);
        // operands of the gate gradients do not depend on the recurrence: issue their loads before the wait
        f32x4 zero = {0.f, 0.f, 0.f, 0.f};
        f32x4 dy = zero, r = zero, z = zero, n = zero, hn = zero, hp = zero;
        if (e_ok) {
            dy = *reinterpret_cast<const f32x4*>(dY + ((long)row * T + tau) * (2 * H) + dir * H + unit0);
            const float* sp = save + dir * save_ds + ((long)row * T + tau) * (4 * H) + unit0;
            r = *reinterpret_cast<const f32x4*>(sp);
            z = *reinterpret_cast<const f32x4*>(sp + H);
            n = *reinterpret_cast<const f32x4*>(sp + 2 * H);
            hn = *reinterpret_cast<const f32x4*>(sp + 3 * H);
            if (has_prev) hp = *reinterpret_cast<const f32x4*>(Y + ((long)row * T + tau_prev) * (2 * H) + dir * H + unit0);
        }
        f32x4 acc[MT][2];
#pragma unroll
        for (int i = 0; i < MT; ++i) acc[i][0] = acc[i][1] = zero;

        if (step > 0) {
            if (wave == 0 && !aborted) {
                unsigned spins = 0;
                for (;;) {
                    const unsigned v = lane < CW ? __hip_atomic_load(cl_flags + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0xffffffffu;
                    if (__all(v >= (unsigned)step)) break;
                    __builtin_amdgcn_s_sleep(1);
                    if (++spins > GC_SPIN_LIMIT) {                    // wave-uniform
                        if (lane == 0) {      // who / when (words 1, 2) and the flag words it saw (4..): diagnostics for the host
                            __hip_atomic_store((gu32*)tmo + 1, (unsigned)step, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            __hip_atomic_store((gu32*)tmo + 2, (unsigned)blockIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        }
                        if (lane < CW) __hip_atomic_store((gu32*)tmo + 4 + lane, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        if (lane == 0) __hip_atomic_store((gu32*)tmo, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        aborted = true;
                        break;
                    }
                }
            }
            __syncthreads();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            const int rslot = (step - 1) & 1;
            const int off0 = (int)(((long)(dir * 2 + rslot) * slot_floats) * 4);
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                f32x4 a[GC_PFB];
                const int rbase = off0 + (b0 + i * 16 + r16) * (3 * GC_HX) * 4;
#pragma unroll
                for (int p = 0; p < GC_PFB; ++p)
                    a[p] = as_f32x4(__builtin_amdgcn_raw_buffer_load_b128(gx_rsrc, rbase + koff[p] * 4, 0, 16));
                // two accumulators: the dependent-accumulator latency of v_mfma_f32_16x16x4_f32 (40 cycles) exceeds its issue
                // interval (32)
#pragma unroll
                for (int p = 0; p < GC_PFB; ++p) {
                    acc[i][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[p][0], w[p][0], acc[i][0], 0, 0, 0);
                    acc[i][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[p][1], w[p][1], acc[i][1], 0, 0, 0);
                    acc[i][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[p][2], w[p][2], acc[i][0], 0, 0, 0);
                    acc[i][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[p][3], w[p][3], acc[i][1], 0, 0, 0);
                }
            }
        }
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int q = 0; q < 4; ++q) red[ks][ut][i][q][lane] = acc[i][0][q] + acc[i][1][q];
        __syncthreads();

        f32x4 g_r = zero, g_z = zero, g_n = zero, g_nr = zero;
        if (epi) {
            f32x4 s = *reinterpret_cast<const f32x4*>(&red[0][e_ut][e_mt][e_i][e_lane]);
#pragma unroll
            for (int q = 1; q < GC_KS; ++q) s += *reinterpret_cast<const f32x4*>(&red[q][e_ut][e_mt][e_i][e_lane]);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float dh = dy[q] + (step > 0 ? s[q] + dh_c[q] * z_c[q] : 0.f);
                const float dn = dh * (1.f - z[q]) * (1.f - n[q] * n[q]);
                const float dz = dh * (hp[q] - n[q]) * z[q] * (1.f - z[q]);
                const float dr = dn * hn[q] * r[q] * (1.f - r[q]);
                dh_c[q] = dh; z_c[q] = z[q];
                g_r[q] = dr; g_z[q] = dz; g_n[q] = dn; g_nr[q] = dn * r[q];
            }
            // publish this step's dgh tile (gate-major rows of the exchange buffer), 16-byte write-through stores
            const int woff = (int)((((long)(dir * 2 + (step & 1)) * slot_floats) + (long)row * (3 * GC_HX) + unit0) * 4);
            __builtin_amdgcn_raw_buffer_store_b128(as_u32x4(g_r), gx_rsrc, woff, 0, 16);
            __builtin_amdgcn_raw_buffer_store_b128(as_u32x4(g_z), gx_rsrc, woff + GC_HX * 4, 0, 16);
            __builtin_amdgcn_raw_buffer_store_b128(as_u32x4(g_nr), gx_rsrc, woff + 2 * GC_HX * 4, 0, 16);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0) __hip_atomic_store(my_flag, (unsigned)(step + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (e_ok) {
            float* gi_o = dgi + dir * dg_ds + ((long)row * T + tau) * H3 + unit0;
            float* gh_o = dgh + dir * dg_ds + ((long)row * T + tau) * H3 + unit0;
            *reinterpret_cast<f32x4*>(gi_o) = g_r; *reinterpret_cast<f32x4*>(gi_o + H) = g_z; *reinterpret_cast<f32x4*>(gi_o + 2 * H) = g_n;
            *reinterpret_cast<f32x4*>(gh_o) = g_r; *reinterpret_cast<f32x4*>(gh_o + H) = g_z; *reinterpret_cast<f32x4*>(gh_o + 2 * H) = g_nr;
        }
    }
}

}  // namespace tg

using namespace tg;

// bf16 x 3 versions of the two kernels (gru_cluster_x3.hip): default; TG_GRU_X3=0 in the environment keeps the f32-MFMA kernels above
int64_t tg_gru_x3_fwd_exchange_bytes(int b_pad, int cw);
int64_t tg_gru_x3_bwd_exchange_bytes(int b_pad, int cw);
int tg_gru_x3_fwd_launch(int mt, const float* gi, long gi_ds, const float* w0, const float* w1, const float* b0, const float* b1, float* y,
                         float* save, long save_ds, const float* drop_mask, float* y_drop, void* hx, unsigned* flags, unsigned* tmo, int B, int T,
                         int H, int n_bt, int cw, int b_pad, int save_row0, int save_rows, hipStream_t s);
int tg_gru_x3_bwd_launch(const float* dy, const float* dy_mask, const float* y, const float* save, long save_ds, const float* wt0, const float* wt1, float* dgi,
                         float* dgh, long dg_ds, void* gx, unsigned* flags, unsigned* tmo, int B, int T, int H, int n_bt, int cw, int b_pad,
                         hipStream_t s);
static bool use_gru_x3() {
    static int x3 = -1;
    if (x3 < 0) {
        const char* e = getenv("TG_GRU_X3");
        x3 = (e && e[0] == '0') ? 0 : 1;
    }
    return x3 == 1;
}

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
    const int64_t f32_bytes = 4 * b_pad * GC_HX * 4, x3_bytes = tg_gru_x3_fwd_exchange_bytes((int)b_pad, cw);
    return flag_words * 4 + (f32_bytes > x3_bytes ? f32_bytes : x3_bytes);
}

extern "C" int32_t tg_gru_cluster_fused_dropout(void) { return use_gru_x3() ? 1 : 0; }

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
    TG_REQUIRE((drop_mask == nullptr) == (y_drop == nullptr) && (drop_mask == nullptr || (use_gru_x3() && aligned16(drop_mask) && aligned16(y_drop))),
               "tg_gru_forward_cluster: drop_mask / y_drop go together, 16-byte aligned, and need tg_gru_cluster_fused_dropout() != 0");
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
    // every polled word is zeroed on the stream before every launch (a kernel node under graph capture).  The first 16 words (the
    // timeout marker and its diagnostics) are NOT touched here: they are sticky until the host reads and clears them
    // (ops.check_async_errors), so a timeout in any launch that shares this workspace survives the launches after it.
    unsigned* tmo = (unsigned*)ws;
    unsigned* flags = tmo + GC_FLAG_STRIDE;
    // (the bf16 x 3 kernels number their flags by generation and need no zeroing: gru_cluster_x3.hip)
    if (!use_gru_x3() && zero_async(flags, (size_t)(flag_words - GC_FLAG_STRIDE) * 4, s)) return 1;
    float* hx = (float*)(tmo + flag_words);
    const int b_pad = n_bt * 16 * mt;
    if (use_gru_x3())
        return tg_gru_x3_fwd_launch(mt, gi, (long)gi_dir_stride, w_hh_fwd, w_hh_rev, b_hh_fwd, b_hh_rev, y, save, (long)save_dir_stride, drop_mask,
                                    y_drop, hx, flags, tmo, B, T, H, n_bt, cw, b_pad, save_row0, save_rows, s);     // (the f32-MFMA fallback below saves every row)
    dim3 grid(2 * n_bt * cw);
    if (mt == 1)
        hipLaunchKernelGGL(gru_seq_fwd_cluster_kernel<1>, grid, dim3(512), 0, s, gi, (long)gi_dir_stride, w_hh_fwd, w_hh_rev, b_hh_fwd,
                           b_hh_rev, y, save, (long)save_dir_stride, hx, flags, tmo, B, T, H, n_bt, cw, b_pad);
    else
        hipLaunchKernelGGL(gru_seq_fwd_cluster_kernel<2>, grid, dim3(512), 0, s, gi, (long)gi_dir_stride, w_hh_fwd, w_hh_rev, b_hh_fwd,
                           b_hh_rev, y, save, (long)save_dir_stride, hx, flags, tmo, B, T, H, n_bt, cw, b_pad);
    return check_launch("tg_gru_forward_cluster");
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
    const int64_t f32_bytes = 4 * (int64_t)n_bt * 16 * 3 * GC_HX * 4, x3_bytes = tg_gru_x3_bwd_exchange_bytes(n_bt * 16, cw);
    return flag_words * 4 + (f32_bytes > x3_bytes ? f32_bytes : x3_bytes);
}

extern "C" int tg_gru_backward_cluster(const float* dy, const float* dy_mask, const float* y, const float* save, int64_t save_dir_stride,
                                       const float* w_hh_t_fwd, const float* w_hh_t_rev, float* dgi, float* dgh, int64_t dg_dir_stride,
                                       void* ws, int64_t ws_bytes, int32_t B, int32_t T, int32_t H, void* stream) {
    TG_REQUIRE(dy && y && save && w_hh_t_fwd && w_hh_t_rev && dgi && dgh && ws, "tg_gru_backward_cluster: null pointer");
    TG_REQUIRE(dy_mask == nullptr || (use_gru_x3() && aligned16(dy_mask)), "tg_gru_backward_cluster: dy_mask needs tg_gru_cluster_fused_dropout() != 0");
    TG_REQUIRE(T > 0 && tg_gru_cluster_bwd_supported(B, H), "tg_gru_backward_cluster: unsupported shape B=%d H=%d", B, H);
    TG_REQUIRE((int64_t)B * T * 4 * H * 4 < (1LL << 31), "tg_gru_backward_cluster: B * T * 4H floats must stay below 2 GB (32-bit buffer offsets), T=%d", T);
    TG_REQUIRE(ws_bytes >= tg_gru_cluster_bwd_ws_bytes(B, H), "tg_gru_backward_cluster: workspace too small");
    TG_REQUIRE(aligned16(dy) && aligned16(y) && aligned16(save) && aligned16(w_hh_t_fwd) && aligned16(w_hh_t_rev) && aligned16(dgi) &&
               aligned16(dgh) && aligned16(ws) && save_dir_stride % 4 == 0 && dg_dir_stride % 4 == 0,
               "tg_gru_backward_cluster: operands must be 16-byte aligned");
    int n_bt, cw;
    cluster_plan_bwd(B, H, &n_bt, &cw);
    const int64_t flag_words = (int64_t)(2 * n_bt + 1) * GC_FLAG_STRIDE;
    hipStream_t s = (hipStream_t)stream;
    unsigned* tmo = (unsigned*)ws;              // sticky timeout block: cleared by the host only (see the forward entry point)
    unsigned* flags = tmo + GC_FLAG_STRIDE;
    if (!use_gru_x3() && zero_async(flags, (size_t)(flag_words - GC_FLAG_STRIDE) * 4, s)) return 1;
    float* gx = (float*)(tmo + flag_words);
    if (use_gru_x3())
        return tg_gru_x3_bwd_launch(dy, dy_mask, y, save, (long)save_dir_stride, w_hh_t_fwd, w_hh_t_rev, dgi, dgh, (long)dg_dir_stride, gx, flags, tmo, B, T,
                                    H, n_bt, cw, n_bt * 16, s);
    hipLaunchKernelGGL(gru_seq_bwd_cluster_kernel<1>, dim3(2 * n_bt * cw), dim3(512), 0, s, dy, y, save, (long)save_dir_stride, w_hh_t_fwd,
                       w_hh_t_rev, dgi, dgh, (long)dg_dir_stride, gx, flags, tmo, B, T, H, n_bt, cw, n_bt * 16);
    return check_launch("tg_gru_backward_cluster");
}
