// Row-gather primitives shared by the decode and aux kernels: one wavefront owns one batch row and
// keeps a full activation-width vector in registers (NC chunks of 8 elements per lane).
#pragma once
#include "common.hpp"

namespace mtc {

template <int DT, int NC>
struct RowCtx {
    // chunk i of this lane covers elements [(lane + 64 i) * 8, +8)
    int lane, nch;
    __device__ __forceinline__ bool valid(int i) const { return lane + 64 * i < nch; }
    __device__ __forceinline__ size_t off(int i) const { return (size_t)(lane + 64 * i) * 8; }
};

// acc += a * W_dec[j] for `n` entries starting at index i0 of the lane-resident batch (cols/vals)
template <int DT, int NC>
__device__ __forceinline__ void accumulate_run(const RowCtx<DT, NC>& rc, const void* __restrict__ W_dec, int A_t,
                                               int my_col, float my_val, int i0, int n, float (&acc)[NC][8]) {
    for (int i = i0; i < i0 + n; ++i) {
        const int j = __builtin_amdgcn_readlane(my_col, i);
        const float a = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(my_val), i));
        const size_t rowoff = (size_t)j * A_t;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            if (rc.valid(c)) {
                float w[8];
                load8<DT>(W_dec, rowoff + rc.off(c), w);
#pragma unroll
                for (int k = 0; k < 8; ++k) acc[c][k] = fmaf(a, w[k], acc[c][k]);
            }
        }
    }
}

// d_vals[e] = g . W_dec[j] + l1_term for `n` entries
template <int DT, int NC>
__device__ __forceinline__ void dot_run(const RowCtx<DT, NC>& rc, const void* __restrict__ W_dec, int A_t, int my_col,
                                        int n, const float (&g)[NC][8], float l1_term, float* __restrict__ d_vals_out) {
    float mine = 0.f;
    for (int i = 0; i < n; ++i) {
        const int j = __builtin_amdgcn_readlane(my_col, i);
        const size_t rowoff = (size_t)j * A_t;
        float p = 0.f;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            if (rc.valid(c)) {
                float w[8];
                load8<DT>(W_dec, rowoff + rc.off(c), w);
#pragma unroll
                for (int k = 0; k < 8; ++k) p = fmaf(g[c][k], w[k], p);
            }
        }
        p = wave_sum(p);
        if (rc.lane == i) mine = p + l1_term;
    }
    if (rc.lane < n) d_vals_out[rc.lane] = mine;
}

static inline int nc_for(int A) { return ((A >> 3) + 63) / 64; }

}  // namespace mtc

// NC = 16-byte-chunks-of-8-elements per lane; instantiate the sizes the supported widths need
#define DISPATCH_NC(dt, nc, KERNEL, grid, block, stream, ...)                                                  \
    do {                                                                                                        \
        const int nc_ = (nc);                                                                                   \
        if ((dt) == MTC_BF16) {                                                                                 \
            if (nc_ <= 1) hipLaunchKernelGGL((KERNEL<MTC_BF16, 1>), grid, block, 0, stream, __VA_ARGS__);       \
            else if (nc_ <= 2) hipLaunchKernelGGL((KERNEL<MTC_BF16, 2>), grid, block, 0, stream, __VA_ARGS__);  \
            else if (nc_ <= 3) hipLaunchKernelGGL((KERNEL<MTC_BF16, 3>), grid, block, 0, stream, __VA_ARGS__);  \
            else if (nc_ <= 5) hipLaunchKernelGGL((KERNEL<MTC_BF16, 5>), grid, block, 0, stream, __VA_ARGS__);  \
            else if (nc_ <= 7) hipLaunchKernelGGL((KERNEL<MTC_BF16, 7>), grid, block, 0, stream, __VA_ARGS__);  \
            else if (nc_ <= 9) hipLaunchKernelGGL((KERNEL<MTC_BF16, 9>), grid, block, 0, stream, __VA_ARGS__);  \
            else hipLaunchKernelGGL((KERNEL<MTC_BF16, 16>), grid, block, 0, stream, __VA_ARGS__);               \
        } else {                                                                                                \
            if (nc_ <= 1) hipLaunchKernelGGL((KERNEL<MTC_F32, 1>), grid, block, 0, stream, __VA_ARGS__);        \
            else if (nc_ <= 2) hipLaunchKernelGGL((KERNEL<MTC_F32, 2>), grid, block, 0, stream, __VA_ARGS__);   \
            else if (nc_ <= 3) hipLaunchKernelGGL((KERNEL<MTC_F32, 3>), grid, block, 0, stream, __VA_ARGS__);   \
            else if (nc_ <= 5) hipLaunchKernelGGL((KERNEL<MTC_F32, 5>), grid, block, 0, stream, __VA_ARGS__);   \
            else if (nc_ <= 7) hipLaunchKernelGGL((KERNEL<MTC_F32, 7>), grid, block, 0, stream, __VA_ARGS__);   \
            else if (nc_ <= 9) hipLaunchKernelGGL((KERNEL<MTC_F32, 9>), grid, block, 0, stream, __VA_ARGS__);   \
            else hipLaunchKernelGGL((KERNEL<MTC_F32, 16>), grid, block, 0, stream, __VA_ARGS__);                \
        }                                                                                                       \
    } while (0)

