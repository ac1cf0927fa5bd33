// LDS image of a 32-row operand slab whose fragments are read TRANSPOSED (ds_read_b64_tr_b16): the weight-gradient kernels' operands are
// row (m) major in memory while the MFMA wants eight consecutive m of one column per lane.  Shared by gemm_split.hip and gemm_tn_mw.hip.
#pragma once
#include "common.hpp"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

namespace tg {

typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
// LDS image of one plane of a slab: [32 rows][W columns] bf16.  W = 128 (256-byte rows): unpadded, the 16-byte chunk index XORed with
// ((row & 3) << 2) | ((row >> 2) & 3) -- layout (b) of cdna_hip_programming.md T10: the transposed reads of a 32-lane half (two blocks
// 8 rows apart, same columns) then land on 64 distinct banks, and 16 consecutive staging stores still cover one contiguous half row.
// Other widths: plain rows padded by 8 bf16 (2-way conflicted transposed reads: rows 8 apart share banks whatever the padding).
template <int W> struct TrImage {
    static constexpr bool SWZ = W == 128;
    static constexpr int LD = SWZ ? W : (W == 64 ? W + 4 : W + 8);      // 64-wide: 136-byte rows -> 52 KB per workgroup, three per CU
    static __device__ __forceinline__ int swz(int row) { return ((row & 3) << 2) | ((row >> 2) & 3); }
    // element offset of column `col` (a multiple of 4) of row `row`
    static __device__ __forceinline__ int at(int row, int col) {
        if constexpr (SWZ) return row * LD + 8 * ((col >> 3) ^ swz(row)) + (col & 7);
        else return row * LD + col;
    }
    // fragment of lane (r16, kq) for the 16-column tile starting at `col0` (a multiple of 16): rows 8 kq .. 8 kq + 7, column r16.
    // Lane 4 q + p of a 16-lane group supplies the address of the block's row q, columns 4 p .. 4 p + 3.
    static __device__ __forceinline__ bf16x8 frag(const __bf16* img, int col0, int r16, int kq) {
        const int row = 8 * kq + (r16 >> 2), col = col0 + 4 * (r16 & 3);
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(img + at(row, col)));
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(img + at(row + 4, col)));
        typedef short s16x8 __attribute__((ext_vector_type(8)));
        const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        return __builtin_bit_cast(bf16x8, v);
    }
};


}  // namespace tg
