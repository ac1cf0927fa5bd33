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

}  // namespace tg

using namespace tg;

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

