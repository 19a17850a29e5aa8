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
// streaming (non-temporal) read of data a kernel touches once: keeps it from displacing the lines other accesses reuse
template <int DT> __device__ __forceinline__ Raw<DT> load_raw_nt(const void* base, size_t elem_off);
template <> __device__ __forceinline__ Raw<MTC_BF16> load_raw_nt<MTC_BF16>(const void* base, size_t elem_off) {
    Raw<MTC_BF16> r;
    r.v = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(reinterpret_cast<const u16*>(base) + elem_off));
    return r;
}
template <> __device__ __forceinline__ Raw<MTC_F32> load_raw_nt<MTC_F32>(const void* base, size_t elem_off) {
    Raw<MTC_F32> r;
    const f32x4* p = reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(base) + elem_off);
    r.a = __builtin_nontemporal_load(p); r.b = __builtin_nontemporal_load(p + 1);
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

// inverse of unpack: 8 floats -> the storage form (bf16: rounded to nearest even, 4 registers)
template <int DT> __device__ __forceinline__ Raw<DT> pack_raw(const float (&w)[8]);
template <> __device__ __forceinline__ Raw<MTC_BF16> pack_raw<MTC_BF16>(const float (&w)[8]) {
    Raw<MTC_BF16> r;
#pragma unroll
    for (int i = 0; i < 4; ++i) r.v[i] = pack2_bf16(w[2 * i], w[2 * i + 1]);
    return r;
}
template <> __device__ __forceinline__ Raw<MTC_F32> pack_raw<MTC_F32>(const float (&w)[8]) {
    Raw<MTC_F32> r;
#pragma unroll
    for (int i = 0; i < 4; ++i) { r.a[i] = w[i]; r.b[i] = w[4 + i]; }
    return r;
}
template <int DT> __device__ __forceinline__ void store_raw(void* base, size_t elem_off, const Raw<DT>& r);
template <> __device__ __forceinline__ void store_raw<MTC_BF16>(void* base, size_t elem_off, const Raw<MTC_BF16>& r) {
    *reinterpret_cast<u32x4*>(reinterpret_cast<u16*>(base) + elem_off) = r.v;
}
template <> __device__ __forceinline__ void store_raw<MTC_F32>(void* base, size_t elem_off, const Raw<MTC_F32>& r) {
    f32x4* p = reinterpret_cast<f32x4*>(reinterpret_cast<float*>(base) + elem_off);
    p[0] = r.a; p[1] = r.b;
}

// p += a . b over one chunk of 8 elements in storage form.  bf16: v_dot2c_f32_bf16 on the packed
// pairs (no unpacking; 4 instructions per chunk); fp32: an fma chain.
__device__ __forceinline__ float dot_chunk(const Raw<MTC_BF16>& a, const Raw<MTC_BF16>& b, float p) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const uint32_t ai = a.v[i], bi = b.v[i];     // (bit_cast of a vector ELEMENT lvalue reads element 0: copy first)
        p = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, ai), __builtin_bit_cast(bf16x2, bi), p, false);
    }
    return p;
}
__device__ __forceinline__ float dot_chunk(const Raw<MTC_F32>& a, const Raw<MTC_F32>& b, float p) {
#pragma unroll
    for (int i = 0; i < 4; ++i) p = fmaf(a.a[i], b.a[i], p);
#pragma unroll
    for (int i = 0; i < 4; ++i) p = fmaf(a.b[i], b.b[i], p);
    return p;
}
template <int DT> __device__ __forceinline__ Raw<DT> zero_raw();
template <> __device__ __forceinline__ Raw<MTC_BF16> zero_raw<MTC_BF16>() { Raw<MTC_BF16> r; r.v = u32x4{0, 0, 0, 0}; return r; }
template <> __device__ __forceinline__ Raw<MTC_F32> zero_raw<MTC_F32>() { Raw<MTC_F32> r; r.a = f32x4{0, 0, 0, 0}; r.b = r.a; return r; }

// rows kept in flight per lane: about 8-10 16-byte loads, bounded so that acc (8 NC) + raw rows
// (4 U NC) stay near 128 VGPRs (4 waves per SIMD)
#ifndef MTC_GATHER_U5
#define MTC_GATHER_U5 2
#endif
__host__ __device__ constexpr int gather_u(int nc) { return nc <= 1 ? 8 : nc <= 2 ? 4 : nc <= 5 ? MTC_GATHER_U5 : 1; }

// Gather loads go through buffer instructions: address = descriptor base + SGPR row offset + a
// per-lane 32-bit VGPR chunk offset that is fixed for the whole kernel, so a load costs no address
// arithmetic and no 64-bit address registers.  Lanes of a chunk beyond the row carry an offset
// past the descriptor's range: the hardware returns zeros without touching memory, which keeps every
// load unconditional (straight-line code, exact vmcnt bookkeeping).  Matrices must be < 2 GiB
// (checked on the host: MTC_GATHER_MAX_BYTES).
#define MTC_GATHER_MAX_BYTES 0x7fffffffull
typedef __amdgpu_buffer_rsrc_t rsrc_t;
__device__ __forceinline__ rsrc_t make_rsrc(const void* base) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, 0x7fffffff, 0x00020000);
}
template <int DT, int NC>
__device__ __forceinline__ void chunk_byte_offsets(const RowCtx<DT, NC>& rc, uint32_t (&co)[NC]) {
#pragma unroll
    for (int c = 0; c < NC; ++c)
        co[c] = rc.valid(c) ? (uint32_t)(rc.lane + 64 * c) * (8u * (uint32_t)sizeof(typename DType<DT>::T)) : 0x80000000u;
}
template <int DT> __device__ __forceinline__ Raw<DT> load_raw_at(rsrc_t rs, uint32_t row_off, uint32_t byte_off);
template <> __device__ __forceinline__ Raw<MTC_BF16> load_raw_at<MTC_BF16>(rsrc_t rs, uint32_t row_off, uint32_t byte_off) {
    Raw<MTC_BF16> r;
    r.v = __builtin_amdgcn_raw_buffer_load_b128(rs, byte_off, row_off, 0);
    return r;
}
template <> __device__ __forceinline__ Raw<MTC_F32> load_raw_at<MTC_F32>(rsrc_t rs, uint32_t row_off, uint32_t byte_off) {
    Raw<MTC_F32> r;
    r.a = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, byte_off, row_off, 0));
    r.b = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, byte_off + 16u, row_off, 0));
    return r;
}

// acc += a * W[j] for `n` entries starting at index i0 of the lane-resident batch (cols/vals).
// Rotating pipeline of P rows: inside the main loop row i+P is requested as soon as row i has been
// consumed, unconditionally, so P * NC 16-byte loads per lane stay in flight and the compiler's
// vmcnt bookkeeping stays exact across the back edge (conditional refills made it drain the queue at
// every stage).
template <int DT, int NC>
__device__ __forceinline__ void fma_row(const Raw<DT> (&r)[NC], float a, float (&acc)[NC][8]) {
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        float w[8];
        unpack(r[c], w);
#pragma unroll
        for (int k = 0; k < 8; ++k) acc[c][k] = fmaf(a, w[k], acc[c][k]);
    }
}
template <int DT, int NC>
__device__ __forceinline__ void load_row(rsrc_t rs, uint32_t row_bytes, int j, const uint32_t (&co)[NC], Raw<DT> (&r)[NC]) {
    const uint32_t row_off = (uint32_t)j * row_bytes;
#pragma unroll
    for (int c = 0; c < NC; ++c) r[c] = load_raw_at<DT>(rs, row_off, co[c]);
}
// same, with every lane pushed out of range when `oob` = 0x80000000: a request that is not needed (pipeline
// refills past the end of a run) then returns zeros without touching memory
template <int DT, int NC>
__device__ __forceinline__ void load_row_masked(rsrc_t rs, uint32_t row_bytes, int j, const uint32_t (&co)[NC], uint32_t oob,
                                                Raw<DT> (&r)[NC]) {
    const uint32_t row_off = (uint32_t)j * row_bytes;
#pragma unroll
    for (int c = 0; c < NC; ++c) r[c] = load_raw_at<DT>(rs, row_off, co[c] | oob);
}
template <int DT, int NC, int P>
__device__ __forceinline__ void accumulate_run_u(const RowCtx<DT, NC>& rc, const void* __restrict__ W, int A,
                                                 int my_col, float my_val, int i0_, int n_, float (&acc)[NC][8]) {
    const int i0 = __builtin_amdgcn_readfirstlane(i0_), n = __builtin_amdgcn_readfirstlane(n_);
    if (n <= 0) return;
    const uint32_t row_bytes = (uint32_t)A * (uint32_t)sizeof(typename DType<DT>::T);
    const int end = i0 + n, last = end - 1;
    uint32_t co[NC];
    chunk_byte_offsets(rc, co);
    const rsrc_t Wb = make_rsrc(W);
    Raw<DT> r[P][NC];
#pragma unroll
    for (int u = 0; u < P; ++u)      // a run shorter than P: the surplus requests are pushed out of range (zeros, no traffic)
        load_row_masked<DT, NC>(Wb, row_bytes, __builtin_amdgcn_readlane(my_col, min(i0 + u, last)), co,
                                i0 + u <= last ? 0u : 0x80000000u, r[u]);
    for (int i = i0; i < end; i += P) {
#pragma unroll
        for (int u = 0; u < P; ++u) {
            const int cur = i + u;
            const float a = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(my_val), min(cur, last)));
            fma_row<DT, NC>(r[u], cur < end ? a : 0.f, acc);
            // refills past the end of the run are pushed out of range: unconditional instruction, no memory request
            load_row_masked<DT, NC>(Wb, row_bytes, __builtin_amdgcn_readlane(my_col, min(cur + P, last)), co,
                                    cur + P <= last ? 0u : 0x80000000u, r[u]);
        }
    }
}
template <int DT, int NC>
__device__ __forceinline__ void accumulate_run(const RowCtx<DT, NC>& rc, const void* __restrict__ W, int A, int my_col,
                                               float my_val, int i0, int n, float (&acc)[NC][8]) {
    accumulate_run_u<DT, NC, gather_u(NC)>(rc, W, A, my_col, my_val, i0, n, acc);
}

// d loss / d a_bj as the reference's autograd holds it: the OUTPUT of a matmul in the model dtype (g @ W_dec^T, fp32
// accumulation, one rounding), the l1 term accumulated onto it in that dtype -- no-ops for fp32 models.  With bf16 this
// makes the reference's bf16 encoder gradients reproducible bit for bit (tests/golden/bf16_plain).
template <int DT>
__device__ __forceinline__ float dvals_round(float dot, float l1_term) {
    float v = DType<DT>::rnd(dot);
    if (l1_term != 0.f) v = DType<DT>::rnd(v + DType<DT>::rnd(l1_term));
    return v;
}

// dL/dx_hat of the l2 term as the reference's autograd forms it on model-dtype tensors (sae.py:236, training.py:172):
//   d = round(x_post - x_t);  g_post = round(inv_n * (2 d))  with inv_n = 1 / (B A_t) ROUNDED to the model dtype (mean's
//   backward divides a model-dtype tensor by N: for N not a power of two a bf16 run carries that 0.2 % scale);
//   through the de-normalisation x_post = x_rec * std + mean:  g_rec = round(g_post * std).
// Verified against torch's bf16 autograd: 100 % of the entries equal (the one-rounding form 2/N * d * std: ~60 %).
template <int DT>
__device__ __forceinline__ float g_post_of(float x_post, float x_t, float inv_n) {
    const float d = DType<DT>::rnd(x_post - x_t);
    return DType<DT>::rnd(inv_n * (2.f * d));
}

// d_vals_out[i] = round(g . W[j_i]) (+ l1_term) for the `n` (<= 64) entries whose columns sit in my_col; g is
// the row in storage form (model dtype, zero in chunks beyond the row).  Four rows at a time; the four
// per-lane partials are folded with 7 cross-lane exchanges (halve the value set while halving the lane
// set) instead of 4 x 6.
template <int DT, int NC>
__device__ __forceinline__ void dot_run(const RowCtx<DT, NC>& rc, const void* __restrict__ W, int A, int my_col,
                                        int n, const Raw<DT> (&g)[NC], float l1_term, float* __restrict__ d_vals_out) {
    const uint32_t row_bytes = (uint32_t)A * (uint32_t)sizeof(typename DType<DT>::T);
    const int lane = rc.lane;
    uint32_t co[NC];
    chunk_byte_offsets(rc, co);
    const rsrc_t Wb = make_rsrc(W);
    int i = 0;
    constexpr int UD = gather_u(NC) >= 4 ? 4 : gather_u(NC);
    for (; i + 4 <= n; i += 4) {
        float p[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int h = 0; h < 4; h += UD) {
            Raw<DT> r[UD][NC];
#pragma unroll
            for (int u = 0; u < UD; ++u)
                load_row<DT, NC>(Wb, row_bytes, __builtin_amdgcn_readlane(my_col, i + h + u), co, r[u]);
#pragma unroll
            for (int u = 0; u < UD; ++u)
#pragma unroll
                for (int c = 0; c < NC; ++c) p[h + u] = dot_chunk(g[c], r[u][c], p[h + u]);
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
        if ((lane & 15) == 0) d_vals_out[i + 2 * (lane >> 5) + ((lane >> 4) & 1)] = dvals_round<DT>(v, l1_term);
    }
    for (; i < n; ++i) {
        Raw<DT> r1[NC];
        load_row<DT, NC>(Wb, row_bytes, __builtin_amdgcn_readlane(my_col, i), co, r1);
        float p = 0.f;
#pragma unroll
        for (int c = 0; c < NC; ++c) p = dot_chunk(g[c], r1[c], p);
        p = wave_sum(p);
        if (lane == 0) d_vals_out[i] = dvals_round<DT>(p, l1_term);
    }
}

// acc += a * g_row and returns this lane's share of g_row . w (w in storage form)
template <int DT, int NC>
__device__ __forceinline__ float fma_dot_row(const Raw<DT> (&r)[NC], float a, const Raw<DT> (&w)[NC], float (&acc)[NC][8]) {
    float pp = 0.f;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        pp = dot_chunk(r[c], w[c], pp);
        float gv[8];
        unpack(r[c], gv);
#pragma unroll
        for (int k = 0; k < 8; ++k) acc[c][k] = fmaf(a, gv[k], acc[c][k]);
    }
    return pp;
}

// Decoder-side backward of one feature j over its `n` (<= 64) lane-resident entries (rows / coefs /
// CSR positions): ONE gather of g_b serves both  acc += a_bj g_b  (the W_dec[j] gradient) and
// d a_bj = g_b . W_dec[j] + l1_term  (written to d_vals_out[pos]; the per-row second sweep of the
// decode kernel re-gathered W_dec for it).  `w` is the feature's W_dec row in storage form.  Same per-lane FMA order
// and the same reduction tree as dot_run, so the values are bit-identical to it.
template <int DT, int NC>
__device__ __forceinline__ void accumulate_dot_run(const RowCtx<DT, NC>& rc, const void* __restrict__ src, int A,
                                                   int my_row, float my_c, int my_pos, int n_, const Raw<DT> (&w)[NC],
                                                   float l1_term, float* __restrict__ d_vals_out, float (&acc)[NC][8]) {
    const int n = __builtin_amdgcn_readfirstlane(n_);
    if (n <= 0) return;
    const int lane = rc.lane, last = n - 1;
    const uint32_t row_bytes = (uint32_t)A * (uint32_t)sizeof(typename DType<DT>::T);
    uint32_t co[NC];
    chunk_byte_offsets(rc, co);
    const rsrc_t rs = make_rsrc(src);
#pragma unroll 1
    for (int i = 0; i < n; i += 4) {
        float p[4];
#pragma unroll
        for (int h = 0; h < 4; h += 2) {
            Raw<DT> r[2][NC];
#pragma unroll
            for (int u = 0; u < 2; ++u)      // entries past the end: out-of-range request (zeros, no traffic; coefficient 0, result unused)
                load_row_masked<DT, NC>(rs, row_bytes, __builtin_amdgcn_readlane(my_row, min(i + h + u, last)), co,
                                        i + h + u <= last ? 0u : 0x80000000u, r[u]);
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int cur = i + h + u;
                float a = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(my_c), min(cur, last)));
                a = cur < n ? a : 0.f;
                p[h + u] = fma_dot_row<DT, NC>(r[u], a, w, acc);
            }
        }
        const bool hi = lane >= 32;
        const float s0 = hi ? p[0] : p[2], s1 = hi ? p[1] : p[3];
        const float k0 = (hi ? p[2] : p[0]) + __shfl_xor(s0, 32, 64);
        const float k1 = (hi ? p[3] : p[1]) + __shfl_xor(s1, 32, 64);
        const bool odd = (lane & 16) != 0;
        float v = (odd ? k1 : k0) + __shfl_xor(odd ? k0 : k1, 16, 64);
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
        const int e = i + 2 * (lane >> 5) + ((lane >> 4) & 1);
        const int pos = __shfl(my_pos, min(e, last), 64);
        if ((lane & 15) == 0 && e < n) d_vals_out[pos] = dvals_round<DT>(v, l1_term);
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

