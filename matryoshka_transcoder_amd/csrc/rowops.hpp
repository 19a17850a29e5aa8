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

// 8 consecutive elements as loaded (16 B for bf16, 32 B for fp32); converted only when consumed, so
// several rows can be in flight per lane without holding them as floats.
template <int DT> struct Raw;
template <> struct Raw<MTC_BF16> { u32x4 v; };
template <> struct Raw<MTC_F32> { f32x4 a, b; };

template <int DT> __device__ __forceinline__ Raw<DT> load_raw(const void* base, size_t elem_off);
template <> __device__ __forceinline__ Raw<MTC_BF16> load_raw<MTC_BF16>(const void* base, size_t elem_off) {
    Raw<MTC_BF16> r;
    r.v = *reinterpret_cast<const u32x4*>(reinterpret_cast<const u16*>(base) + elem_off);
    return r;
}
template <> __device__ __forceinline__ Raw<MTC_F32> load_raw<MTC_F32>(const void* base, size_t elem_off) {
    Raw<MTC_F32> r;
    const f32x4* p = reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(base) + elem_off);
    r.a = p[0]; r.b = p[1];
    return r;
}
__device__ __forceinline__ void unpack(const Raw<MTC_BF16>& r, float (&w)[8]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) { w[2 * i] = __uint_as_float(r.v[i] << 16); w[2 * i + 1] = __uint_as_float(r.v[i] & 0xffff0000u); }
}
__device__ __forceinline__ void unpack(const Raw<MTC_F32>& r, float (&w)[8]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) { w[i] = r.a[i]; w[4 + i] = r.b[i]; }
}

// rows kept in flight per lane: about 8-10 16-byte loads, bounded so that acc (8 NC) + raw rows
// (4 U NC) stay near 128 VGPRs (4 waves per SIMD)
#ifndef MTC_GATHER_U5
#define MTC_GATHER_U5 2
#endif
__host__ __device__ constexpr int gather_u(int nc) { return nc <= 1 ? 8 : nc <= 2 ? 4 : nc <= 5 ? MTC_GATHER_U5 : 1; }

// acc += a * W[j] for `n` entries starting at index i0 of the lane-resident batch (cols/vals);
// U rows are requested before any is consumed (U * NC 16-byte loads in flight per lane).
template <int DT, int NC, int U>
__device__ __forceinline__ void accumulate_run_u(const RowCtx<DT, NC>& rc, const void* __restrict__ W, int A,
                                                 int my_col, float my_val, int i0, int n, float (&acc)[NC][8]) {
    int i = i0;
    const int end = i0 + n;
    for (; i + U <= end; i += U) {
        Raw<DT> r[U][NC];
        float a[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int j = __builtin_amdgcn_readlane(my_col, i + u);
            a[u] = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(my_val), i + u));
            const size_t rowoff = (size_t)j * A;
#pragma unroll
            for (int c = 0; c < NC; ++c)
                if (rc.valid(c)) r[u][c] = load_raw<DT>(W, rowoff + rc.off(c));
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int c = 0; c < NC; ++c)
                if (rc.valid(c)) {
                    float w[8];
                    unpack(r[u][c], w);
#pragma unroll
                    for (int k = 0; k < 8; ++k) acc[c][k] = fmaf(a[u], w[k], acc[c][k]);
                }
    }
    for (; i < end; ++i) {
        const int j = __builtin_amdgcn_readlane(my_col, i);
        const float a = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(my_val), i));
        const size_t rowoff = (size_t)j * A;
#pragma unroll
        for (int c = 0; c < NC; ++c)
            if (rc.valid(c)) {
                float w[8];
                unpack(load_raw<DT>(W, rowoff + rc.off(c)), w);
#pragma unroll
                for (int k = 0; k < 8; ++k) acc[c][k] = fmaf(a, w[k], acc[c][k]);
            }
    }
}
template <int DT, int NC>
__device__ __forceinline__ void accumulate_run(const RowCtx<DT, NC>& rc, const void* __restrict__ W, int A, int my_col,
                                               float my_val, int i0, int n, float (&acc)[NC][8]) {
    accumulate_run_u<DT, NC, gather_u(NC)>(rc, W, A, my_col, my_val, i0, n, acc);
}

// d_vals_out[i] = g . W[j_i] + l1_term for the `n` (<= 64) entries whose columns sit in my_col.
// Four rows at a time; the four per-lane partials are folded with 7 cross-lane exchanges
// (halve the value set while halving the lane set) instead of 4 x 6.
template <int DT, int NC>
__device__ __forceinline__ void dot_run(const RowCtx<DT, NC>& rc, const void* __restrict__ W, int A, int my_col,
                                        int n, const float (&g)[NC][8], float l1_term, float* __restrict__ d_vals_out) {
    const int lane = rc.lane;
    int i = 0;
    constexpr int UD = gather_u(NC) >= 4 ? 4 : gather_u(NC);
    for (; i + 4 <= n; i += 4) {
        float p[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int h = 0; h < 4; h += UD) {
            Raw<DT> r[UD][NC];
#pragma unroll
            for (int u = 0; u < UD; ++u) {
                const int j = __builtin_amdgcn_readlane(my_col, i + h + u);
                const size_t rowoff = (size_t)j * A;
#pragma unroll
                for (int c = 0; c < NC; ++c)
                    if (rc.valid(c)) r[u][c] = load_raw<DT>(W, rowoff + rc.off(c));
            }
#pragma unroll
            for (int u = 0; u < UD; ++u)
#pragma unroll
                for (int c = 0; c < NC; ++c)
                    if (rc.valid(c)) {
                        float w[8];
                        unpack(r[u][c], w);
#pragma unroll
                        for (int k = 0; k < 8; ++k) p[h + u] = fmaf(g[c][k], w[k], p[h + u]);
                    }
        }
        // lanes < 32 keep entries {0,1}, lanes >= 32 keep {2,3}
        const bool hi = lane >= 32;
        const float s0 = hi ? p[0] : p[2], s1 = hi ? p[1] : p[3];
        float k0 = (hi ? p[2] : p[0]) + __shfl_xor(s0, 32, 64);
        float k1 = (hi ? p[3] : p[1]) + __shfl_xor(s1, 32, 64);
        // within each half: lanes with bit 4 clear keep the first of the pair, set keep the second
        const bool odd = (lane & 16) != 0;
        float v = (odd ? k1 : k0) + __shfl_xor(odd ? k0 : k1, 16, 64);
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
        // entry index held by this lane's 16-lane group: 2*(lane>>5) + ((lane>>4)&1)
        if ((lane & 15) == 0) d_vals_out[i + 2 * (lane >> 5) + ((lane >> 4) & 1)] = v + l1_term;
    }
    for (; i < n; ++i) {
        const int j = __builtin_amdgcn_readlane(my_col, i);
        const size_t rowoff = (size_t)j * A;
        float p = 0.f;
#pragma unroll
        for (int c = 0; c < NC; ++c)
            if (rc.valid(c)) {
                float w[8];
                unpack(load_raw<DT>(W, rowoff + rc.off(c)), w);
#pragma unroll
                for (int k = 0; k < 8; ++k) p = fmaf(g[c][k], w[k], p);
            }
        p = wave_sum(p);
        if (lane == 0) d_vals_out[i] = p + l1_term;
    }
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

