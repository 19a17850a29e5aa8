// K5: decoder unit-norm projection (sae.py:155-163) fused with Adam (training.py:60,178:
// torch.optim.Adam, eps 1e-8, no weight decay, no amsgrad).
//
// Pure HBM streaming: per parameter the step reads p, grad, m, v and writes p, m, v
// (2s + s_g + 4 s_m bytes: 28 B fp32, 14 B bf16).  W_dec rows are handled one wavefront per row,
// the whole row held in registers, so norm, projection and Adam cost one read of the row.
// Arithmetic is evaluated op by op with a rounding to the parameter dtype after each op, the
// way PyTorch's single-tensor Adam evaluates it on same-dtype tensors:
//   m   <- m + (1-b1) (g - m)                                   [lerp_]
//   v   <- v*b2 ; v <- v + (1-b2) g g                            [mul_, addcmul_]
//   den <- sqrt(v) ; den <- den / sqrt(bc2) ; den <- den + eps   [sqrt, div, add_]
//   p   <- p - (lr/bc1) * (m / den)                              [addcdiv_]
#include "common.hpp"
#include "rowops.hpp"
#include <math.h>
#include <stdlib.h>

namespace mtc {

struct AdamK {
    float w1;            // 1 - beta1
    float beta2;
    float w2;            // 1 - beta2
    float inv_bc2_sqrt;  // 1 / sqrt(1 - beta2^step): tensor / python-scalar is a multiply by the reciprocal on the GPU
    float eps;
    float neg_step;      // -(lr / (1 - beta1^step))
};

// fp32 parameters: IEEE sqrt and division.  bf16 parameters: every intermediate is rounded to 8
// mantissa bits straight away, so the 1-ulp v_sqrt_f32 / v_rcp_f32 results round to the same bf16
// value except when the exact result sits within 2^-16 relative of a rounding boundary; the
// correctly rounded sequences cost ~25 extra VALU ops per element and made the step ALU-bound.
template <int DT> __device__ __forceinline__ float sqrt_dt(float x) { return sqrtf(x); }
template <> __device__ __forceinline__ float sqrt_dt<MTC_BF16>(float x) { return __builtin_amdgcn_sqrtf(x); }
template <int DT> __device__ __forceinline__ float div_dt(float a, float b) { return a / b; }
template <> __device__ __forceinline__ float div_dt<MTC_BF16>(float a, float b) { return a * __builtin_amdgcn_rcpf(b); }

template <int DT>
__device__ __forceinline__ void adam_elem(float& p, float g, float& m, float& v, const AdamK& k) {
    typedef DType<DT> D_;
    m = D_::rnd(fmaf(k.w1, g - m, m));
    v = D_::rnd(v * k.beta2);
    v = D_::rnd(fmaf(k.w2 * g, g, v));
    float den = D_::rnd(sqrt_dt<DT>(v));
    den = D_::rnd(den * k.inv_bc2_sqrt);
    den = D_::rnd(den + k.eps);
    p = D_::rnd(fmaf(k.neg_step, div_dt<DT>(m, den), p));
}

template <int DT>
__global__ __launch_bounds__(256) void adam_flat_kernel(size_t n_chunks, void* __restrict__ p, const void* __restrict__ g,
                                                        void* __restrict__ m, void* __restrict__ v, AdamK k) {
    for (size_t c = (size_t)blockIdx.x * 256 + threadIdx.x; c < n_chunks; c += (size_t)gridDim.x * 256) {
        float pp[8], gg[8], mm[8], vv[8];
        load8<DT>(p, c * 8, pp); load8<DT>(g, c * 8, gg); load8<DT>(m, c * 8, mm); load8<DT>(v, c * 8, vv);
#pragma unroll
        for (int i = 0; i < 8; ++i) adam_elem<DT>(pp[i], gg[i], mm[i], vv[i], k);
        store8<DT>(p, c * 8, pp); store8<DT>(m, c * 8, mm); store8<DT>(v, c * 8, vv);
    }
}

// wave per decoder row.  MODE 0: projection only (weights normalised, grad projected, written back)
//                        MODE 1: projection + Adam (grad optionally written back)
template <int DT, int NC, int MODE>
__global__ __launch_bounds__(256) void unitnorm_rows_kernel(int D, int A, void* __restrict__ W, void* __restrict__ grad,
                                                            void* __restrict__ m, void* __restrict__ v, AdamK k,
                                                            int write_grad) {
    typedef DType<DT> D_;
    const int lane = threadIdx.x & 63;
    const int j = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
    if (j >= D) return;
    const int nch = A >> 3;
    float w[NC][8], g[NC][8];
    float ss = 0.f;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        if (lane + 64 * c < nch) {
            const size_t off = (size_t)j * A + (size_t)(lane + 64 * c) * 8;
            load8<DT>(W, off, w[c]);
            load8<DT>(grad, off, g[c]);
#pragma unroll
            for (int i = 0; i < 8; ++i) ss = fmaf(w[c][i], w[c][i], ss);
        }
    }
    const float norm = D_::rnd(sqrtf(wave_sum(ss)));            // W_dec.norm(dim=-1) in the param dtype
    float dot = 0.f;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        if (lane + 64 * c < nch) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                w[c][i] = D_::rnd(div_dt<DT>(w[c][i], norm));       // W_dec_normed
                dot += D_::rnd(g[c][i] * w[c][i]);                // (grad * W_normed).sum(-1)
            }
        }
    }
    dot = D_::rnd(wave_sum(dot));
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        if (lane + 64 * c < nch) {
            const size_t off = (size_t)j * A + (size_t)(lane + 64 * c) * 8;
#pragma unroll
            for (int i = 0; i < 8; ++i) g[c][i] = D_::rnd(g[c][i] - D_::rnd(dot * w[c][i]));
            if (MODE == 0 || write_grad) store8<DT>(grad, off, g[c]);
            if (MODE == 1) {
                float mm[8], vv[8];
                load8<DT>(m, off, mm); load8<DT>(v, off, vv);
#pragma unroll
                for (int i = 0; i < 8; ++i) adam_elem<DT>(w[c][i], g[c][i], mm[i], vv[i], k);
                store8<DT>(m, off, mm); store8<DT>(v, off, vv);
            }
            store8<DT>(W, off, w[c]);
        }
    }
}

// K4 + K5 fused: one wave per feature builds the gradient row in registers (same gather as
// grad_rows_kernel), rounds it to the parameter dtype exactly as a materialised .grad would be, and
// applies [unit-norm projection +] Adam to the parameter row straight away -- the dense gradient is
// never written to or re-read from HBM (saves 2 * 2 * D * A * s bytes per step).
//   PROJECT = 1: decoder rows (src = g, coef = vals)        PROJECT = 0: encoder rows (src = x_norm,
//   coef = d_vals) plus the encoder bias element of the feature.
#ifndef MTC_FUSED_WAVES
#define MTC_FUSED_WAVES 3
#endif
#ifndef MTC_FUSED_WAVES_WIDE
#define MTC_FUSED_WAVES_WIDE 1          // rows of 8+ chunks per lane (A = 4608, the 27B shapes): at 2 waves per SIMD the
                                        // decoder form spills 67 registers inside its gather loop (7.70 vs 6.53 ms per step)
#endif
//   WITH_DOT = 1 (decoder rows only): the same gather of g also produces d a_bj = g_b . W_dec[j] + l1_term
//   for the feature's entries (d_vals_out[csc_pos]), which replaces the decode kernel's second sweep.
//   EXTRA = 1: a second, dense gradient contribution for some rows (the auxiliary branch's GEMM results over the dead
//   features): extra_row[j] >= 0 names the row of `extra` [.][A] (and the element of `extra_bias`) that is added to the
//   rounded sparse gradient of feature j, with the rounding a dense `grad += ...` would make.  EXTRA = 0 compiles to
//   the kernel without these operands.
template <int DT, int COEF_DT, int NC, int PROJECT, int WITH_DOT, int EXTRA>
__global__ __launch_bounds__(256, (DT == MTC_BF16 ? (NC <= 5 ? MTC_FUSED_WAVES : NC <= 7 ? 2 : MTC_FUSED_WAVES_WIDE) : (NC <= 2 ? 4 : NC <= 5 ? 2 : 1))) void fused_grad_adam_kernel(
    int D, int A, const int32_t* __restrict__ col_ptr, const int32_t* __restrict__ csc_row,
    const int32_t* __restrict__ csc_pos, const void* __restrict__ coef, const void* __restrict__ src,
    const float* __restrict__ scale_p, void* __restrict__ W, void* __restrict__ m, void* __restrict__ v,
    void* __restrict__ bias, void* __restrict__ m_b, void* __restrict__ v_b, AdamK k, float* __restrict__ d_vals_out,
    float l1_term, const int32_t* __restrict__ extra_row, const void* __restrict__ extra,
    const void* __restrict__ extra_bias, int csc_order /* d a_bj indexed by CSC entry instead of CSR position */) {
    typedef DType<DT> D_;
    typedef typename D_::T T;
    RowCtx<DT, NC> rc;
    rc.lane = threadIdx.x & 63;
    rc.nch = A >> 3;
    const int lane = rc.lane;
    const int j = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
    if (j >= D) return;
    const float scale = scale_p ? *scale_p : 1.f;
    float acc[NC][8];
#pragma unroll
    for (int c = 0; c < NC; ++c)
#pragma unroll
        for (int q = 0; q < 8; ++q) acc[c][q] = 0.f;
    float csum = 0.f;
    Raw<DT> gr[NC], pw[NC], pm[NC], pv[NC];
    if constexpr (WITH_DOT != 0) {                     // the (pre-update) parameter row takes part in the gather loop
        uint32_t co[NC];
        chunk_byte_offsets(rc, co);          // chunks beyond the row read as zeros
        load_row<DT, NC>(make_rsrc(W), (uint32_t)A * (uint32_t)sizeof(T), j, co, pw);
    }
    const int s0 = col_ptr[j], e_end = col_ptr[j + 1];
    // (requested with the column pointers: asked for after the gather loop it cost every wave one more round trip
    // before its moment rows could be requested, +80 us on the launch)
    int xr = -1;
    if constexpr (EXTRA != 0) xr = __builtin_amdgcn_readfirstlane(extra_row[j]);
    for (int e0 = s0; e0 < e_end; e0 += 64) {
        const int nb = min(64, e_end - e0);
        int my_row = 0, my_pos = 0; float my_c = 0.f;
        if (lane < nb) {
            my_row = csc_row[e0 + lane];
            my_pos = csc_pos[e0 + lane];
            // encoder rows: the coefficients d a_bj come in the order the decoder launch wrote them
            const int at = (!PROJECT && csc_order) ? e0 + lane : my_pos;
            my_c = DType<COEF_DT>::ld(reinterpret_cast<const typename DType<COEF_DT>::T*>(coef) + at);
        }
        csum += my_c;
        if constexpr (WITH_DOT != 0)
            accumulate_dot_run<DT, NC>(rc, src, A, my_row, my_c, csc_order ? e0 + lane : my_pos, nb, pw, l1_term, d_vals_out, acc);
        else accumulate_run<DT, NC>(rc, src, A, my_row, my_c, 0, nb, acc);
    }
    // the .grad a dense backward would hold: rounded to the parameter dtype and kept in storage
    // form (bf16: 4 registers per chunk instead of 8), which frees the registers for the parameter
    // and moment rows -- all of them are requested before any is consumed
    if constexpr (EXTRA != 0) {
        // on the accumulators, before they are packed (packing first and unpacking per chunk made the compiler spill 68
        // registers, and a launch with a scratch segment starts its waves more slowly: +80 us)
        if (xr >= 0) {
#pragma unroll
            for (int c = 0; c < NC; ++c)
                if (rc.valid(c)) {
                    float b[8];
                    load8<DT>(extra, (size_t)xr * A + rc.off(c), b);
#pragma unroll
                    for (int q = 0; q < 8; ++q) acc[c][q] = D_::rnd(D_::rnd(acc[c][q] * scale) + b[q]);
                }
        } else {
#pragma unroll
            for (int c = 0; c < NC; ++c)
#pragma unroll
                for (int q = 0; q < 8; ++q) acc[c][q] *= scale;
        }
#pragma unroll
        for (int c = 0; c < NC; ++c)
            if (rc.valid(c)) gr[c] = pack_raw<DT>(acc[c]);
    } else {
#pragma unroll
        for (int c = 0; c < NC; ++c)
            if (rc.valid(c)) {
#pragma unroll
                for (int q = 0; q < 8; ++q) acc[c][q] *= scale;
                gr[c] = pack_raw<DT>(acc[c]);
            }
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int c = 0; c < NC; ++c)
        if (rc.valid(c)) {
            const size_t off = (size_t)j * A + rc.off(c);
            if (!WITH_DOT) pw[c] = load_raw<DT>(W, off);
            // decoder moments stream past the caches (read once, written once per step); the encoder-row kernel is the
            // last of the step and keeps plain accesses: with streaming stores there the next encoder launch measured
            // 17 us slower, which cancels the gain
            pm[c] = PROJECT ? load_raw_nt<DT>(m, off) : load_raw<DT>(m, off);
            pv[c] = PROJECT ? load_raw_nt<DT>(v, off) : load_raw<DT>(v, off);
        }
    float norm = 1.f, dot = 0.f;
    if (PROJECT) {
        float ss = 0.f;
#pragma unroll
        for (int c = 0; c < NC; ++c)
            if (rc.valid(c)) {
                float w[8];
                unpack(pw[c], w);
#pragma unroll
                for (int q = 0; q < 8; ++q) ss = fmaf(w[q], w[q], ss);
            }
        norm = D_::rnd(sqrtf(wave_sum(ss)));
#pragma unroll
        for (int c = 0; c < NC; ++c)
            if (rc.valid(c)) {
                float w[8], g[8];
                unpack(pw[c], w); unpack(gr[c], g);
#pragma unroll
                for (int q = 0; q < 8; ++q) dot += D_::rnd(g[q] * D_::rnd(div_dt<DT>(w[q], norm)));
            }
        dot = D_::rnd(wave_sum(dot));
    }
#pragma unroll
    for (int c = 0; c < NC; ++c)
        if (rc.valid(c)) {
            const size_t off = (size_t)j * A + rc.off(c);
            float w[8], g[8], mm[8], vv[8];
            unpack(pw[c], w); unpack(gr[c], g); unpack(pm[c], mm); unpack(pv[c], vv);
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                if (PROJECT) {
                    w[q] = D_::rnd(div_dt<DT>(w[q], norm));
                    g[q] = D_::rnd(g[q] - D_::rnd(dot * w[q]));
                }
                adam_elem<DT>(w[q], g[q], mm[q], vv[q], k);
            }
            store8<DT>(W, off, w);
            if (PROJECT) { store8_nt<DT>(m, off, mm); store8_nt<DT>(v, off, vv); }
            else { store8<DT>(m, off, mm); store8<DT>(v, off, vv); }
        }
    if (bias) {
        csum = wave_sum(csum);
        if (lane == 0) {
            T* bp = reinterpret_cast<T*>(bias) + j; T* mp = reinterpret_cast<T*>(m_b) + j; T* vp = reinterpret_cast<T*>(v_b) + j;
            float b = D_::ld(bp), mb = D_::ld(mp), vb = D_::ld(vp);
            float gb = D_::rnd(scale * csum);
            if constexpr (EXTRA != 0)
                if (xr >= 0 && extra_bias) gb = D_::rnd(gb + D_::ld(reinterpret_cast<const T*>(extra_bias) + xr));
            adam_elem<DT>(b, gb, mb, vb, k);
            D_::st(bp, b); D_::st(mp, mb); D_::st(vp, vb);
        }
    }
}


// Column-sliced form of the encoder-row kernel (fused_grad_adam_kernel<PROJECT = 0>): the same per-feature gradient +
// Adam, but one launch walks the parameter matrix slice by slice -- LPF lanes (LPF * 8 columns) per feature, 64 / LPF
// features per wave -- in an order that keeps every XCD on ONE column slice of the gather table x [B][A] for a long
// stretch: that slice (B * LPF * 16 bytes, 2 MiB at B = 4096 / LPF = 32) stays in the XCD's 4 MiB L2, so the k-fold row
// gathers stop crossing the fabric (the row kernel re-fetches the 18.9 MB table ~96 times per step through the Infinity
// Cache: 1.8 GB of fabric reads next to 1 GB of compulsory parameter / moment stream).  The encoder rows need no
// row-wide reduction (no projection), so slices are independent; the bias element rides with slice 0.
// Coefficients d a_bj are read in CSC order (coef[e], written that way by the decoder launch).  Same arithmetic per
// element as the row kernel (entries accumulated in ascending row order, one rounding of the gradient, op-by-op Adam).
#ifndef MTC_SLICED_WAVES
#define MTC_SLICED_WAVES 8               // waves per SIMD the register budget is set for
#endif
#ifndef MTC_SLICED_U
#define MTC_SLICED_U 2
#endif
//   PASS_A = 1 (measurement only, mtc_diag_sliced_decoder_bound): what a column-sliced DECODER-row kernel would have
//   to run before its update pass -- gather, gradient slice, and the slice's share of the two row-wide sums the unit-norm
//   projection needs (|w|^2 and grad . w), written to scratch [D][slices][2]; no moments, nothing updated.
template <int DT, int LPF, int EXTRA, int PASS_A = 0>
__global__ __launch_bounds__(256, MTC_SLICED_WAVES) void fused_enc_sliced_kernel(
    int D, int A, int units_per_slice, int total_units, const int32_t* __restrict__ col_ptr,
    const int32_t* __restrict__ csc_row, const float* __restrict__ coef, const void* __restrict__ src,
    const float* __restrict__ scale_p, void* __restrict__ W, void* __restrict__ m, void* __restrict__ v,
    void* __restrict__ bias, void* __restrict__ m_b, void* __restrict__ v_b, AdamK k,
    const int32_t* __restrict__ extra_row, const void* __restrict__ extra, const void* __restrict__ extra_bias,
    float* __restrict__ pass_a_out = nullptr) {
    typedef DType<DT> D_;
    typedef typename D_::T T;
    constexpr int F = 64 / LPF;                    // features per wave
    // A wave spends most of its life waiting on a chain of dependent reads (column pointers -> entry list -> rows ->
    // parameter pieces): the launch is latency x occupancy bound, so registers are kept to 64 (8 waves per SIMD) and the
    // pipeline stages small.  (Measured and dropped: walking several units per wave with the next unit's pointers and
    // entry list requested behind the current unit's reads -- same time, more code.)
    constexpr int U = MTC_SLICED_U;                // gathered rows in flight per lane and pipeline stage
    constexpr uint32_t OOB = 0x80000000u;          // buffer offset beyond every descriptor: the request returns zeros, no traffic
    // workgroup b runs on XCD b % 8: unit = position in that XCD's contiguous range of the slice-major unit list
    // (unit = one block of 4 * F features in one column slice)
    const int per_xcd = (total_units + 7) >> 3;
    const int u = (int)(blockIdx.x & 7) * per_xcd + (int)(blockIdx.x >> 3);
    if (u >= total_units) return;
    const int slice = u / units_per_slice, fb = u - slice * units_per_slice;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int sub = lane & (LPF - 1);
    const int grp = lane & ~(LPF - 1);             // first lane of this feature's lane group
    const int j = (fb * 4 + wave) * F + lane / LPF;
    const bool jv = j < D;
    const int col0 = (slice * LPF + sub) * 8;
    const size_t off = (size_t)(jv ? j : 0) * A + col0;
    const int s0 = jv ? col_ptr[j] : 0;
    const int n = jv ? col_ptr[j + 1] - s0 : 0;
    int xr = -1;
    if constexpr (EXTRA != 0) xr = jv ? extra_row[j] : -1;    // with the column pointers: long back when it is needed
    int nmax = n;
#pragma unroll
    for (int o = LPF; o < 64; o <<= 1) nmax = max(nmax, __shfl_xor(nmax, o, 64));
    nmax = __builtin_amdgcn_readfirstlane(nmax);
    const rsrc_t rs = make_rsrc(src), rs_row = make_rsrc(csc_row), rs_coef = make_rsrc(coef);
    const uint32_t row_bytes = (uint32_t)A * (uint32_t)sizeof(T), lane_off = (uint32_t)col0 * (uint32_t)sizeof(T);
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    float csum = 0.f;
    // The entry lists are not read per entry: the LPF lanes of a feature fetch LPF of its (row, coefficient) pairs with
    // one request each and hand them round through the LDS crossbar (ds_bpermute), which leaves ONE memory request per
    // gathered row.  Groups of U rows are software-pipelined (group g + 1 in flight while group g is accumulated); every
    // request is unconditional (entries past the feature's end: OOB) so the wait counts are exact.
    // B, A < 2^24 (checked on the host).
    for (int base = 0; base < nmax; base += LPF) {
        const uint32_t eoff = base + sub < n ? (uint32_t)(s0 + base + sub) * 4u : OOB;
        const int my_r = (int)__builtin_amdgcn_raw_buffer_load_b32(rs_row, eoff, 0, 0);
        const float my_c = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs_coef, eoff, 0, 0));
        const int cnt = min(LPF, nmax - base);     // uniform
        const int left = n - base;                 // entries of THIS feature from `base` on (may be <= 0)
        Raw<DT> dr[2][U];
        float c[2][U];
        auto issue = [&](int i, Raw<DT> (&d_)[U], float (&c_)[U]) {
#pragma unroll
            for (int q = 0; q < U; ++q) {
                const int src_lane = grp + ((i + q) & (LPF - 1));
                const int rq = __shfl(my_r, src_lane, 64);
                const float cq = __shfl(my_c, src_lane, 64);
                const bool ok = i + q < left && i + q < LPF;
                c_[q] = ok ? cq : 0.f;
                d_[q] = load_raw_at<DT>(rs, 0u, ok ? __umul24((uint32_t)rq, row_bytes) + lane_off : OOB);
            }
        };
        auto consume = [&](const Raw<DT> (&d_)[U], const float (&c_)[U]) {
#pragma unroll
            for (int q = 0; q < U; ++q) {
                // (Tried: the FMAs on the matrix pipe -- v_mfma_f32_4x4x4_16b_bf16 with A = c * identity gives every lane
                // acc[i] += c * x[i] on its own packed values, 2 instructions instead of 12.  Bit-identical on small
                // problems, but at 3.4e8 elements x 41 entries a few accumulators differ from fmaf in the last fp32 bit
                // (the matrix pipe's adder is not a correctly rounded fused multiply-add), for a 2 % gain: dropped.)
                float x[8];
                unpack(d_[q], x);
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[e] = fmaf(c_[q], x[e], acc[e]);
                csum += c_[q];
            }
        };
        issue(0, dr[0], c[0]);
        for (int i = 0; i < cnt; i += 2 * U) {
            issue(i + U, dr[1], c[1]);
            consume(dr[0], c[0]);
            issue(i + 2 * U, dr[0], c[0]);
            if (i + U < cnt) consume(dr[1], c[1]);
        }
    }
    if constexpr (PASS_A != 0) {
        const Raw<DT> pwa = load_raw_nt<DT>(W, jv ? off : 0);
        float wv[8], ss = 0.f, dt = 0.f;
        unpack(pwa, wv);
#pragma unroll
        for (int e = 0; e < 8; ++e) { ss = fmaf(wv[e], wv[e], ss); dt = fmaf(D_::rnd(acc[e]), wv[e], dt); }
#pragma unroll
        for (int o = 1; o < LPF; o <<= 1) { ss += __shfl_xor(ss, o, 64); dt += __shfl_xor(dt, o, 64); }
        if (jv && sub == 0) {
            const int n_slices = A / (LPF * 8);
            pass_a_out[((size_t)j * n_slices + slice) * 2] = ss;
            pass_a_out[((size_t)j * n_slices + slice) * 2 + 1] = dt;
        }
        return;
    }
    if (!jv) return;
    // parameter / moment pieces AFTER the gather loop: a wave's memory reads return in order, so requesting them first
    // would hold every gathered row (an L2 hit) behind three HBM round trips; other waves hide this latency instead
    const Raw<DT> pw = load_raw_nt<DT>(W, off), pm = load_raw_nt<DT>(m, off), pv = load_raw_nt<DT>(v, off);
    const float scale = scale_p ? *scale_p : 1.f;
    float g[8], w[8], mm[8], vv[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) g[e] = D_::rnd(acc[e] * scale);          // the .grad a dense backward would hold
    if constexpr (EXTRA != 0) {
        if (xr >= 0) {
            float b[8];
            load8<DT>(extra, (size_t)xr * A + col0, b);
#pragma unroll
            for (int e = 0; e < 8; ++e) g[e] = D_::rnd(g[e] + b[e]);
        }
    }
    unpack(pw, w); unpack(pm, mm); unpack(pv, vv);
#pragma unroll
    for (int e = 0; e < 8; ++e) adam_elem<DT>(w[e], g[e], mm[e], vv[e], k);
    store8<DT>(W, off, w);
    store8_nt<DT>(m, off, mm);
    store8_nt<DT>(v, off, vv);
    if (bias && slice == 0 && sub == 0) {
        T* bp = reinterpret_cast<T*>(bias) + j; T* mp = reinterpret_cast<T*>(m_b) + j; T* vp = reinterpret_cast<T*>(v_b) + j;
        float b = D_::ld(bp), mb = D_::ld(mp), vb = D_::ld(vp);
        float gb = D_::rnd(scale * csum);
        if constexpr (EXTRA != 0)
            if (xr >= 0 && extra_bias) gb = D_::rnd(gb + D_::ld(reinterpret_cast<const T*>(extra_bias) + xr));
        adam_elem<DT>(b, gb, mb, vb, k);
        D_::st(bp, b); D_::st(mp, mb); D_::st(vp, vb);
    }
}

static int g_sliced_mode = -1;     // 0 automatic, 1 never, 2 whenever the shape allows (tests)
static int sliced_lpf(const mtc_desc* d, int A) {
    if (g_sliced_mode < 0) {
        const char* e = getenv("MTC_FUSED_SLICED");
        g_sliced_mode = e ? atoi(e) : 0;
        if (g_sliced_mode < 0 || g_sliced_mode > 2) g_sliced_mode = 0;
    }
    if (g_sliced_mode == 1 || d->dtype != MTC_BF16 || d->B >= (1 << 24)) return 0;     // (24-bit row-offset multiply)
    // widest slice whose share of the gather table (B rows x LPF x 16 bytes) leaves room in a 4 MiB L2
    for (int lpf = 32; lpf >= 8; lpf >>= 1) {
        if (A % (lpf * 8)) continue;
        const size_t table = (size_t)d->B * lpf * 16;
        if (table <= (size_t)5 * 512 * 1024) {
            // small problems (table already L2-sized as a whole) gain nothing: keep the row kernel
            if (g_sliced_mode == 2 || (size_t)d->B * A * 2 > (size_t)6 * 1024 * 1024) return lpf;
            return 0;
        }
    }
    return 0;
}

static AdamK make_adam(const mtc_adam_hyper* h) {
    AdamK k;
    const double bc1 = 1.0 - pow(h->beta1, (double)h->step);
    const double bc2 = 1.0 - pow(h->beta2, (double)h->step);
    k.w1 = (float)(1.0 - h->beta1);
    k.beta2 = (float)h->beta2;
    k.w2 = (float)(1.0 - h->beta2);
    k.inv_bc2_sqrt = (float)(1.0 / sqrt(bc2));
    k.eps = (float)h->eps;
    k.neg_step = (float)(-(h->lr / bc1));
    return k;
}

}  // namespace mtc

using namespace mtc;

#define LAUNCH_ROWS(DTV, MODE)                                                                                               \
    do {                                                                                                                     \
        const int nc_ = nc_for(d->A_t);                                                                                      \
        const dim3 grid((d->D + 3) / 4), block(256);                                                                         \
        if (nc_ <= 1) hipLaunchKernelGGL((unitnorm_rows_kernel<DTV, 1, MODE>), grid, block, 0, s, d->D, d->A_t, W_dec, grad, m, v, k, write_grad); \
        else if (nc_ <= 2) hipLaunchKernelGGL((unitnorm_rows_kernel<DTV, 2, MODE>), grid, block, 0, s, d->D, d->A_t, W_dec, grad, m, v, k, write_grad); \
        else if (nc_ <= 3) hipLaunchKernelGGL((unitnorm_rows_kernel<DTV, 3, MODE>), grid, block, 0, s, d->D, d->A_t, W_dec, grad, m, v, k, write_grad); \
        else if (nc_ <= 5) hipLaunchKernelGGL((unitnorm_rows_kernel<DTV, 5, MODE>), grid, block, 0, s, d->D, d->A_t, W_dec, grad, m, v, k, write_grad); \
        else if (nc_ <= 7) hipLaunchKernelGGL((unitnorm_rows_kernel<DTV, 7, MODE>), grid, block, 0, s, d->D, d->A_t, W_dec, grad, m, v, k, write_grad); \
        else if (nc_ <= 9) hipLaunchKernelGGL((unitnorm_rows_kernel<DTV, 9, MODE>), grid, block, 0, s, d->D, d->A_t, W_dec, grad, m, v, k, write_grad); \
        else hipLaunchKernelGGL((unitnorm_rows_kernel<DTV, 16, MODE>), grid, block, 0, s, d->D, d->A_t, W_dec, grad, m, v, k, write_grad); \
    } while (0)

extern "C" int mtc_unitnorm_adam_rows(const mtc_desc* d, void* W_dec, void* grad, void* m, void* v,
                                      const mtc_adam_hyper* h_host, int write_grad, void* stream) {
    if (check_desc(d, "mtc_unitnorm_adam_rows")) return -1;
    MTC_REQUIRE(W_dec && grad && m && v && h_host, "mtc_unitnorm_adam_rows: null pointer");
    MTC_REQUIRE(h_host->step >= 1, "mtc_unitnorm_adam_rows: step is 1-based");
    const AdamK k = make_adam(h_host);
    hipStream_t s = as_stream(stream);
    if (d->dtype == MTC_BF16) LAUNCH_ROWS(MTC_BF16, 1);
    else LAUNCH_ROWS(MTC_F32, 1);
    return check_launch("mtc_unitnorm_adam_rows");
}

extern "C" int mtc_unitnorm_project(const mtc_desc* d, void* W_dec, void* grad, void* stream) {
    if (check_desc(d, "mtc_unitnorm_project")) return -1;
    MTC_REQUIRE(W_dec && grad, "mtc_unitnorm_project: null pointer");
    AdamK k = {};
    void* m = nullptr; void* v = nullptr;
    const int write_grad = 1;
    hipStream_t s = as_stream(stream);
    if (d->dtype == MTC_BF16) LAUNCH_ROWS(MTC_BF16, 0);
    else LAUNCH_ROWS(MTC_F32, 0);
    return check_launch("mtc_unitnorm_project");
}

// Measurement aid (no counterpart in the reference, not on the product path): lower bound of a column-sliced form of the
// DECODER-row gradient + Adam launch.  The unit-norm projection needs two row-wide sums before any element of the row can
// be updated, so a sliced form has two passes over the slices: pass 0 = gather + gradient slice + partial sums (this
// kernel with PASS_A: no moments, nothing written but 8 bytes per feature and slice; the d a_bj dot products a real
// pass A also owes are NOT included), pass 1 = gather again + update = the encoder-row sliced kernel run on the decoder's
// operands (no projection arithmetic).  tools/fused_ab.py times both next to the row kernel.
extern "C" int mtc_diag_sliced_decoder_bound(const mtc_desc* d, const int32_t* col_ptr, const int32_t* csc_row,
                                             const float* coef_csc, const void* g, void* W_dec, void* exp_avg, void* exp_avg_sq,
                                             float* scratch, int32_t pass, void* stream) {
    if (check_desc(d, "mtc_diag_sliced_decoder_bound")) return -1;
    MTC_REQUIRE(col_ptr && csc_row && coef_csc && g && W_dec && exp_avg && exp_avg_sq && scratch, "mtc_diag_sliced_decoder_bound: null pointer");
    MTC_REQUIRE(d->dtype == MTC_BF16 && d->A_t % 256 == 0, "mtc_diag_sliced_decoder_bound: bf16, A_t a multiple of 256");
    const int A = d->A_t, lpf = 32;
    const int feats_per_block = 4 * (64 / lpf);
    const int units_per_slice = (d->D + feats_per_block - 1) / feats_per_block;
    const int total_units = units_per_slice * (A / (lpf * 8));
    const dim3 grid(8 * ((total_units + 7) / 8)), block(256);
    mtc_adam_hyper h = {3e-4, 0.9, 0.99, 1e-8, 1, 0};
    const AdamK k = make_adam(&h);
    hipStream_t s = as_stream(stream);
    if (pass == 0)
        hipLaunchKernelGGL((fused_enc_sliced_kernel<MTC_BF16, 32, 0, 1>), grid, block, 0, s, d->D, A, units_per_slice, total_units, col_ptr,
                           csc_row, coef_csc, g, (const float*)nullptr, W_dec, exp_avg, exp_avg_sq, (void*)nullptr, (void*)nullptr,
                           (void*)nullptr, k, (const int32_t*)nullptr, (const void*)nullptr, (const void*)nullptr, scratch);
    else
        hipLaunchKernelGGL((fused_enc_sliced_kernel<MTC_BF16, 32, 0, 0>), grid, block, 0, s, d->D, A, units_per_slice, total_units, col_ptr,
                           csc_row, coef_csc, g, (const float*)nullptr, W_dec, exp_avg, exp_avg_sq, (void*)nullptr, (void*)nullptr,
                           (void*)nullptr, k, (const int32_t*)nullptr, (const void*)nullptr, (const void*)nullptr, (float*)nullptr);
    return check_launch("mtc_diag_sliced_decoder_bound");
}

extern "C" int mtc_fused_set_sliced(int mode) {
    if (mode < 0 || mode > 2) return -1;
    const int prev = g_sliced_mode < 0 ? 0 : g_sliced_mode;
    g_sliced_mode = mode;
    return prev;
}

extern "C" int mtc_adam_flat(int32_t dtype, size_t n, void* p, const void* grad, void* m, void* v,
                             const mtc_adam_hyper* h_host, void* stream) {
    MTC_REQUIRE(p && grad && m && v && h_host, "mtc_adam_flat: null pointer");
    MTC_REQUIRE((n & 7) == 0, "mtc_adam_flat: element count must be a multiple of 8 (got %zu)", n);
    MTC_REQUIRE(dtype == MTC_F32 || dtype == MTC_BF16, "mtc_adam_flat: bad dtype");
    MTC_REQUIRE(h_host->step >= 1, "mtc_adam_flat: step is 1-based");
    if (n == 0) return 0;
    const AdamK k = make_adam(h_host);
    const size_t n_chunks = n / 8;
    const int grid = (int)std::min<size_t>((n_chunks + 255) / 256, 4096);
    if (dtype == MTC_BF16)
        hipLaunchKernelGGL(adam_flat_kernel<MTC_BF16>, dim3(grid), dim3(256), 0, as_stream(stream), n_chunks, p, grad, m, v, k);
    else
        hipLaunchKernelGGL(adam_flat_kernel<MTC_F32>, dim3(grid), dim3(256), 0, as_stream(stream), n_chunks, p, grad, m, v, k);
    return check_launch("mtc_adam_flat");
}

#define LAUNCH_FUSED_NC(DTV, COEF, PROJ, DOT, EX, NCV)                                                                          \
    hipLaunchKernelGGL((fused_grad_adam_kernel<DTV, COEF, NCV, PROJ, DOT, EX>), grid, block, 0, s, d->D, A, col_ptr, csc_row,      \
                       csc_pos, coef, src, scale, W, m, v, bias, m_b, v_b, k, dv_out, l1_term, extra_row, extra, extra_bias,          \
                       csc_order)
#define LAUNCH_FUSED(DTV, COEF, PROJ, DOT, EX)                                                                                    \
    do {                                                                                                                     \
        const int nc_ = nc_for(A);                                                                                           \
        const dim3 grid((d->D + 3) / 4), block(256);                                                                         \
        if (nc_ <= 1) LAUNCH_FUSED_NC(DTV, COEF, PROJ, DOT, EX, 1);                                                          \
        else if (nc_ <= 2) LAUNCH_FUSED_NC(DTV, COEF, PROJ, DOT, EX, 2);                                                     \
        else if (nc_ <= 3) LAUNCH_FUSED_NC(DTV, COEF, PROJ, DOT, EX, 3);                                                     \
        else if (nc_ <= 5) LAUNCH_FUSED_NC(DTV, COEF, PROJ, DOT, EX, 5);                                                     \
        else if (nc_ <= 7) LAUNCH_FUSED_NC(DTV, COEF, PROJ, DOT, EX, 7);                                                     \
        else if (nc_ <= 9) LAUNCH_FUSED_NC(DTV, COEF, PROJ, DOT, EX, 9);                                                     \
        else LAUNCH_FUSED_NC(DTV, COEF, PROJ, DOT, EX, 16);                                                                  \
    } while (0)

extern "C" int mtc_fused_backward_adam_dec(const mtc_desc* d, const int32_t* col_ptr, const int32_t* csc_row,
                                           const int32_t* csc_pos, const void* vals, const void* g, const float* scale,
                                           void* W_dec, void* exp_avg, void* exp_avg_sq, const mtc_adam_hyper* h_host,
                                           float* d_vals_out, float l1_term, void* stream) {
    return mtc_fused_backward_adam_dec_extra(d, col_ptr, csc_row, csc_pos, vals, g, scale, W_dec, exp_avg, exp_avg_sq, h_host,
                                             d_vals_out, l1_term, nullptr, nullptr, 0, stream);
}

extern "C" int mtc_fused_backward_adam_dec_extra(const mtc_desc* d, const int32_t* col_ptr, const int32_t* csc_row,
                                                 const int32_t* csc_pos, const void* vals, const void* g, const float* scale,
                                                 void* W_dec, void* exp_avg, void* exp_avg_sq, const mtc_adam_hyper* h_host,
                                                 float* d_vals_out, float l1_term, const int32_t* extra_row,
                                                 const void* extra_rows, int32_t d_vals_csc_order, void* stream) {
    if (check_desc(d, "mtc_fused_backward_adam_dec")) return -1;
    const int csc_order = d_vals_csc_order ? 1 : 0;
    MTC_REQUIRE((extra_row == nullptr) == (extra_rows == nullptr), "mtc_fused_backward_adam_dec: extra_row and extra_rows go together");
    MTC_REQUIRE(col_ptr && csc_row && csc_pos && vals && g && W_dec && exp_avg && exp_avg_sq && h_host,
                "mtc_fused_backward_adam_dec: null pointer");
    MTC_REQUIRE(h_host->step >= 1, "mtc_fused_backward_adam_dec: step is 1-based");
    const AdamK k = make_adam(h_host);
    hipStream_t s = as_stream(stream);
    const int A = d->A_t;
    const void* coef = vals; const void* src = g;
    void* W = W_dec; void* m = exp_avg; void* v = exp_avg_sq; void* bias = nullptr; void* m_b = nullptr; void* v_b = nullptr;
    float* dv_out = d_vals_out;
    const void* extra = extra_rows; const void* extra_bias = nullptr;
    if (extra_row) {
        MTC_REQUIRE(dv_out, "mtc_fused_backward_adam_dec: the extra-row form also produces d_vals (pass d_vals_out)");
        if (d->dtype == MTC_BF16) LAUNCH_FUSED(MTC_BF16, MTC_BF16, 1, 1, 1);
        else LAUNCH_FUSED(MTC_F32, MTC_F32, 1, 1, 1);
    } else if (dv_out) {
        if (d->dtype == MTC_BF16) LAUNCH_FUSED(MTC_BF16, MTC_BF16, 1, 1, 0);
        else LAUNCH_FUSED(MTC_F32, MTC_F32, 1, 1, 0);
    } else {
        if (d->dtype == MTC_BF16) LAUNCH_FUSED(MTC_BF16, MTC_BF16, 1, 0, 0);
        else LAUNCH_FUSED(MTC_F32, MTC_F32, 1, 0, 0);
    }
    return check_launch("mtc_fused_backward_adam_dec");
}

extern "C" int mtc_fused_backward_adam_enc(const mtc_desc* d, const int32_t* col_ptr, const int32_t* csc_row,
                                           const int32_t* csc_pos, const float* d_vals, const void* x_norm, const float* scale,
                                           void* W_encT, void* exp_avg, void* exp_avg_sq, void* b_enc, void* b_exp_avg,
                                           void* b_exp_avg_sq, const mtc_adam_hyper* h_host, void* stream) {
    return mtc_fused_backward_adam_enc_extra(d, col_ptr, csc_row, csc_pos, d_vals, x_norm, scale, W_encT, exp_avg, exp_avg_sq,
                                             b_enc, b_exp_avg, b_exp_avg_sq, h_host, nullptr, nullptr, nullptr, 0, stream);
}

extern "C" int mtc_fused_backward_adam_enc_extra(const mtc_desc* d, const int32_t* col_ptr, const int32_t* csc_row,
                                                 const int32_t* csc_pos, const float* d_vals, const void* x_norm,
                                                 const float* scale, void* W_encT, void* exp_avg, void* exp_avg_sq, void* b_enc,
                                                 void* b_exp_avg, void* b_exp_avg_sq, const mtc_adam_hyper* h_host,
                                                 const int32_t* extra_row, const void* extra_rows, const void* extra_bias_in,
                                                 int32_t d_vals_csc_order, void* stream) {
    if (check_desc(d, "mtc_fused_backward_adam_enc")) return -1;
    const int csc_order = d_vals_csc_order ? 1 : 0;
    MTC_REQUIRE((extra_row == nullptr) == (extra_rows == nullptr), "mtc_fused_backward_adam_enc: extra_row and extra_rows go together");
    MTC_REQUIRE(col_ptr && csc_row && csc_pos && d_vals && x_norm && W_encT && exp_avg && exp_avg_sq && b_enc && b_exp_avg &&
                b_exp_avg_sq && h_host, "mtc_fused_backward_adam_enc: null pointer");
    MTC_REQUIRE(h_host->step >= 1, "mtc_fused_backward_adam_enc: step is 1-based");
    const AdamK k = make_adam(h_host);
    hipStream_t s = as_stream(stream);
    const int A = d->A_s;
    const void* coef = d_vals; const void* src = x_norm;
    void* W = W_encT; void* m = exp_avg; void* v = exp_avg_sq; void* bias = b_enc; void* m_b = b_exp_avg; void* v_b = b_exp_avg_sq;
    float* dv_out = nullptr; const float l1_term = 0.f;
    const void* extra = extra_rows; const void* extra_bias = extra_bias_in;
    const int lpf = csc_order ? sliced_lpf(d, A) : 0;
    if (lpf) {
        // column-sliced walk: units = (slice, block of 4 waves x 64 / lpf features), slice-major, an eighth of the list per XCD
        const int feats_per_block = 4 * (64 / lpf);
        const int units_per_slice = (d->D + feats_per_block - 1) / feats_per_block;
        const int total_units = units_per_slice * (A / (lpf * 8));
        const dim3 grid(8 * ((total_units + 7) / 8)), block(256);
#define LAUNCH_SLICED(LPFV, EX)                                                                                                  \
        hipLaunchKernelGGL((fused_enc_sliced_kernel<MTC_BF16, LPFV, EX, 0>), grid, block, 0, s, d->D, A, units_per_slice, total_units, \
                           col_ptr, csc_row, d_vals, src, scale, W, m, v, bias, m_b, v_b, k, extra_row, extra, extra_bias, (float*)nullptr)
        if (extra_row) { if (lpf == 32) LAUNCH_SLICED(32, 1); else if (lpf == 16) LAUNCH_SLICED(16, 1); else LAUNCH_SLICED(8, 1); }
        else { if (lpf == 32) LAUNCH_SLICED(32, 0); else if (lpf == 16) LAUNCH_SLICED(16, 0); else LAUNCH_SLICED(8, 0); }
#undef LAUNCH_SLICED
        return check_launch("mtc_fused_backward_adam_enc");
    }
    if (extra_row) {
        if (d->dtype == MTC_BF16) LAUNCH_FUSED(MTC_BF16, MTC_F32, 0, 0, 1);
        else LAUNCH_FUSED(MTC_F32, MTC_F32, 0, 0, 1);
    } else {
        if (d->dtype == MTC_BF16) LAUNCH_FUSED(MTC_BF16, MTC_F32, 0, 0, 0);
        else LAUNCH_FUSED(MTC_F32, MTC_F32, 0, 0, 0);
    }
    return check_launch("mtc_fused_backward_adam_enc");
}
