// K3: nested-prefix decode in sparse form (sae.py:217-230), per-prefix squared error for the loss
// and FVU (sae.py:236,248-255; config.py:205-234), output gradient g, and d(acts) for the selected
// entries -- one WAVEFRONT per batch row, accumulators in registers.
//
// A row's selected entries arrive sorted by feature index, i.e. already in prefix-group order, so
// the G nested reconstructions are ONE running sum sampled at the group boundaries.  Each entry
// gathers one decoder row W_dec[j] (A_t contiguous elements, 16 B per lane per load: the whole
// wave reads 1 KiB contiguous segments), so the kernel is a pure HBM / Infinity-Cache gather:
// algorithmic bytes = nnz * A_t * s (+ 2 * B * A_t * s for target in / reconstruction out).
// When `d_vals` is requested the same wave sweeps its entries a second time with g in registers
// (the decoder rows it needs were just touched, so the second sweep is served by L2 / MALL).
#include "common.hpp"
#include "rowops.hpp"

#ifndef MTC_DECODE_WAVES
#define MTC_DECODE_WAVES 4     // waves per SIMD the row-gather kernels are register-bounded for (NC <= 5)
#endif

namespace mtc {

struct DecodeParams {
    int B, A_t, G;
    int prefix[MTC_MAX_PREFIX];
    int unit_norm;
    float inv_n;           // 1 / (B * A_t) rounded to the model dtype (see g_post_of)
    float l1_term;         // l1_coeff / (B * D_total)
};

// Epilogue at a prefix boundary: reconstruction in the model dtype, squared error against target.
// LAST: also x_hat, g (kept in acc on return when KEEP_G), target moments.
template <int DT, int NC>
__device__ __forceinline__ float boundary_sse(const RowCtx<DT, NC>& rc, const DecodeParams& p, int row,
                                              const float (&acc)[NC][8], const void* __restrict__ b_dec,
                                              const void* __restrict__ x_target, float mean, float stdv,
                                              void* __restrict__ recon_out /* row base or null */) {
    typedef DType<DT> D_;
    float sse = 0.f;
    if (!x_target && !recon_out) return 0.f;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        if (rc.valid(c)) {
            float b[8], t[8] = {0, 0, 0, 0, 0, 0, 0, 0}, r[8];
            load8<DT>(b_dec, rc.off(c), b);
            if (x_target) load8<DT>(x_target, (size_t)row * p.A_t + rc.off(c), t);
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                float v = D_::rnd(acc[c][k] + b[k]);
                if (p.unit_norm) v = D_::rnd(D_::rnd(v * stdv) + mean);
                r[k] = v;
                const float dlt = t[k] - v;
                sse = fmaf(dlt, dlt, sse);
            }
            if (recon_out) store8<DT>(recon_out, rc.off(c), r);
        }
    }
    return wave_sum(sse);
}

template <int DT, int NC>
__global__ __launch_bounds__(256, (NC <= 5 ? MTC_DECODE_WAVES : 2)) void decode_fwd_kernel(DecodeParams p, const int32_t* __restrict__ row_ptr,
                                                         const int32_t* __restrict__ col_idx, const void* __restrict__ vals,
                                                         const void* __restrict__ W_dec, const void* __restrict__ b_dec,
                                                         const void* __restrict__ x_target, const void* __restrict__ mean_p,
                                                         const void* __restrict__ std_p, void* __restrict__ x_hat,
                                                         void* __restrict__ g_out, float* __restrict__ row_stats,
                                                         void* __restrict__ recons, float* __restrict__ d_vals) {
    typedef DType<DT> D_;
    typedef typename D_::T T;
    RowCtx<DT, NC> rc;
    rc.lane = threadIdx.x & 63;
    rc.nch = p.A_t >> 3;
    const int row = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
    if (row >= p.B) return;
    const int e_begin = row_ptr[row], e_end = row_ptr[row + 1];
    float mean = 0.f, stdv = 1.f;
    if (p.unit_norm) {
        mean = D_::ld(reinterpret_cast<const T*>(mean_p) + row);
        stdv = D_::ld(reinterpret_cast<const T*>(std_p) + row);
    }
    float acc[NC][8];
#pragma unroll
    for (int c = 0; c < NC; ++c)
#pragma unroll
        for (int k = 0; k < 8; ++k) acc[c][k] = 0.f;

    int gi = 0;
    float abs_sum = 0.f;
    const T* vp = reinterpret_cast<const T*>(vals);
    for (int e0 = e_begin; e0 < e_end; e0 += 64) {
        const int nb = min(64, e_end - e0);
        const int my_col = rc.lane < nb ? col_idx[e0 + rc.lane] : 0x7fffffff;
        const float my_val = rc.lane < nb ? D_::ld(vp + e0 + rc.lane) : 0.f;
        abs_sum += fabsf(my_val);
        int i = 0;
        while (i < nb) {
            // entries of the batch that still belong to group gi (columns ascend, so a lane prefix)
            const int upto = __popcll(__ballot(my_col < p.prefix[gi]));
            if (upto <= i) {
                const float sse = boundary_sse<DT, NC>(rc, p, row, acc, b_dec, x_target, mean, stdv,
                    recons ? reinterpret_cast<T*>(recons) + ((size_t)gi * p.B + row) * p.A_t : nullptr);
                if (rc.lane == 0 && row_stats) row_stats[(size_t)gi * p.B + row] = sse;
                ++gi;
                continue;
            }
            accumulate_run<DT, NC>(rc, W_dec, p.A_t, my_col, my_val, i, upto - i, acc);
            i = upto;
        }
    }
    // remaining boundaries (all but the last, which also produces x_hat / g)
    for (; gi < p.G - 1; ++gi) {
        const float sse = boundary_sse<DT, NC>(rc, p, row, acc, b_dec, x_target, mean, stdv,
            recons ? reinterpret_cast<T*>(recons) + ((size_t)gi * p.B + row) * p.A_t : nullptr);
        if (rc.lane == 0 && row_stats) row_stats[(size_t)gi * p.B + row] = sse;
    }
    // ---- last prefix: x_hat, sse, target moments, g
    float sse = 0.f, sx = 0.f, sxx = 0.f;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        if (rc.valid(c)) {
            float b[8], t[8] = {0, 0, 0, 0, 0, 0, 0, 0}, r[8], gg[8];
            load8<DT>(b_dec, rc.off(c), b);
            if (x_target) load8<DT>(x_target, (size_t)row * p.A_t + rc.off(c), t);
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                float v = D_::rnd(acc[c][k] + b[k]);
                if (p.unit_norm) v = D_::rnd(D_::rnd(v * stdv) + mean);
                r[k] = v;
                const float dlt = v - t[k];
                sse = fmaf(dlt, dlt, sse);
                sx += t[k];
                sxx = fmaf(t[k], t[k], sxx);
                gg[k] = g_post_of<DT>(v, t[k], p.inv_n);
                if (p.unit_norm) gg[k] = D_::rnd(gg[k] * stdv);
                acc[c][k] = gg[k];                    // acc now holds g (as stored, in the model dtype) for the second sweep
            }
            store8<DT>(x_hat, (size_t)row * p.A_t + rc.off(c), r);
            if (recons) store8<DT>(reinterpret_cast<T*>(recons) + ((size_t)(p.G - 1) * p.B + row) * p.A_t, rc.off(c), r);
            if (g_out) store8<DT>(g_out, (size_t)row * p.A_t + rc.off(c), gg);
        }
    }
    sse = wave_sum(sse); sx = wave_sum(sx); sxx = wave_sum(sxx); abs_sum = wave_sum(abs_sum);
    if (rc.lane == 0 && row_stats) {
        row_stats[(size_t)(p.G - 1) * p.B + row] = sse;
        row_stats[(size_t)(p.G + 0) * p.B + row] = sx;
        row_stats[(size_t)(p.G + 1) * p.B + row] = sxx;
        row_stats[(size_t)(p.G + 2) * p.B + row] = abs_sum;
        row_stats[(size_t)(p.G + 3) * p.B + row] = 0.f;
    }
    // ---- second sweep: d a_bj = g_b . W_dec[j] (+ l1 term)
    if (d_vals) {
        Raw<DT> gp[NC];
#pragma unroll
        for (int c = 0; c < NC; ++c) gp[c] = rc.valid(c) ? pack_raw<DT>(acc[c]) : zero_raw<DT>();
        for (int e0 = e_begin; e0 < e_end; e0 += 64) {
            const int nb = min(64, e_end - e0);
            const int my_col = rc.lane < nb ? col_idx[e0 + rc.lane] : 0;
            dot_run<DT, NC>(rc, W_dec, p.A_t, my_col, nb, gp, p.l1_term, d_vals + e0);
        }
    }
}

// standalone d_vals (aux-active and feature-sharded paths: g comes from memory)
template <int DT, int NC>
__global__ __launch_bounds__(256, (NC <= 5 ? MTC_DECODE_WAVES : 2)) void decode_bwd_vals_kernel(int B, int A_t, const int32_t* __restrict__ row_ptr,
                                                              const int32_t* __restrict__ col_idx,
                                                              const void* __restrict__ W_dec, const void* __restrict__ g,
                                                              float l1_term, float* __restrict__ d_vals) {
    RowCtx<DT, NC> rc;
    rc.lane = threadIdx.x & 63;
    rc.nch = A_t >> 3;
    const int row = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
    if (row >= B) return;
    Raw<DT> gr[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) gr[c] = rc.valid(c) ? load_raw<DT>(g, (size_t)row * A_t + rc.off(c)) : zero_raw<DT>();
    const int e_begin = row_ptr[row], e_end = row_ptr[row + 1];
    for (int e0 = e_begin; e0 < e_end; e0 += 64) {
        const int nb = min(64, e_end - e0);
        const int my_col = rc.lane < nb ? col_idx[e0 + rc.lane] : 0;
        dot_run<DT, NC>(rc, W_dec, A_t, my_col, nb, gr, l1_term, d_vals + e0);
    }
}

// feature-sharded: bias-free cumulative partial reconstructions, fp32, partial[G][B][A_t]
template <int DT, int NC>
__global__ __launch_bounds__(256) void decode_partial_kernel(DecodeParams p, const int32_t* __restrict__ row_ptr,
                                                             const int32_t* __restrict__ col_idx,
                                                             const void* __restrict__ vals, const void* __restrict__ W_dec,
                                                             int rps, long long pre_shard_stride, float* __restrict__ partial_last,
                                                             void* __restrict__ partial_pre, float* __restrict__ abs_rows) {
    typedef DType<DT> D_;
    typedef typename D_::T T;
    RowCtx<DT, NC> rc;
    rc.lane = threadIdx.x & 63;
    rc.nch = p.A_t >> 3;
    const int row = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
    if (row >= p.B) return;
    const int e_begin = row_ptr[row], e_end = row_ptr[row + 1];
    float acc[NC][8];
#pragma unroll
    for (int c = 0; c < NC; ++c)
#pragma unroll
        for (int k = 0; k < 8; ++k) acc[c][k] = 0.f;
    // rows are grouped by the rank that will finish them (rps rows each) so that a reduce-scatter hands
    // every rank one contiguous block: pre[shard * pre_shard_stride + (gi * rps + r) * A_t], last[row] (already
    // shard-major); pre_shard_stride = (G-1) * rps * A_t for one call over the whole batch, larger when the batch is
    // decoded in row blocks that share one exchange buffer [shard][block][G-1][rps][A_t]
    const int shard = row / rps, r_in = row - shard * rps;
    auto flush = [&](int gi) {
#pragma unroll
        for (int c = 0; c < NC; ++c)
            if (rc.valid(c)) {
                if (gi == p.G - 1) store8<MTC_F32>(partial_last, (size_t)row * p.A_t + rc.off(c), acc[c]);
                else store8<DT>(partial_pre, (size_t)shard * (size_t)pre_shard_stride + ((size_t)gi * rps + r_in) * p.A_t + rc.off(c), acc[c]);
            }
    };
    int gi = 0;
    float abs_sum = 0.f;
    const T* vp = reinterpret_cast<const T*>(vals);
    for (int e0 = e_begin; e0 < e_end; e0 += 64) {
        const int nb = min(64, e_end - e0);
        const int my_col = rc.lane < nb ? col_idx[e0 + rc.lane] : 0x7fffffff;
        const float my_val = rc.lane < nb ? D_::ld(vp + e0 + rc.lane) : 0.f;
        abs_sum += fabsf(my_val);
        int i = 0;
        while (i < nb) {
            const int upto = __popcll(__ballot(my_col < p.prefix[gi]));
            if (upto <= i) { flush(gi); ++gi; continue; }
            accumulate_run<DT, NC>(rc, W_dec, p.A_t, my_col, my_val, i, upto - i, acc);
            i = upto;
        }
    }
    for (; gi < p.G; ++gi) flush(gi);
    abs_sum = wave_sum(abs_sum);
    if (rc.lane == 0 && abs_rows) abs_rows[row] = abs_sum;
}

// epilogue of the sharded path on the reduced partials of this rank's rows (one wave per row, no gather);
// p.B = local rows, p.inv_n carries the GLOBAL batch
template <int DT>
__global__ __launch_bounds__(256) void decode_finish_kernel(DecodeParams p, const float* __restrict__ partial_last,
                                                            const void* __restrict__ partial_pre,
                                                            const void* __restrict__ b_dec, const void* __restrict__ x_target,
                                                            const void* __restrict__ mean_p, const void* __restrict__ std_p,
                                                            void* __restrict__ x_hat, void* __restrict__ g_out,
                                                            float* __restrict__ row_stats, void* __restrict__ recons) {
    typedef DType<DT> D_;
    typedef typename D_::T T;
    const int lane = threadIdx.x & 63;
    const int row = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
    if (row >= p.B) return;
    float mean = 0.f, stdv = 1.f;
    if (p.unit_norm) {
        mean = D_::ld(reinterpret_cast<const T*>(mean_p) + row);
        stdv = D_::ld(reinterpret_cast<const T*>(std_p) + row);
    }
    const int nch = p.A_t >> 3;
    float sx = 0.f, sxx = 0.f;
    for (int gi = 0; gi < p.G; ++gi) {
        const bool last = gi == p.G - 1;
        if (last ? partial_last == nullptr : partial_pre == nullptr) continue;     // split calls: last prefix now, the others later
        float sse = 0.f;
        for (int c = lane; c < nch; c += 64) {
            float a[8], b[8], t[8], r[8], gg[8];
            const size_t off = (size_t)c * 8;
            if (last) load8<MTC_F32>(partial_last, (size_t)row * p.A_t + off, a);
            else load8<DT>(partial_pre, ((size_t)gi * p.B + row) * p.A_t + off, a);
            load8<DT>(b_dec, off, b);
            load8<DT>(x_target, (size_t)row * p.A_t + off, t);
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                float v = D_::rnd(a[k] + b[k]);
                if (p.unit_norm) v = D_::rnd(D_::rnd(v * stdv) + mean);
                r[k] = v;
                const float dlt = v - t[k];
                sse = fmaf(dlt, dlt, sse);
                gg[k] = g_post_of<DT>(v, t[k], p.inv_n);
                if (p.unit_norm) gg[k] = D_::rnd(gg[k] * stdv);
                if (last) { sx += t[k]; sxx = fmaf(t[k], t[k], sxx); }
            }
            if (recons) store8<DT>(reinterpret_cast<T*>(recons) + ((size_t)gi * p.B + row) * p.A_t, off, r);
            if (last) {
                store8<DT>(x_hat, (size_t)row * p.A_t + off, r);
                if (g_out) store8<DT>(g_out, (size_t)row * p.A_t + off, gg);
            }
        }
        sse = wave_sum(sse);
        if (lane == 0) row_stats[(size_t)gi * p.B + row] = sse;
    }
    sx = wave_sum(sx); sxx = wave_sum(sxx);
    if (lane == 0 && partial_last) {
        row_stats[(size_t)(p.G + 0) * p.B + row] = sx;
        row_stats[(size_t)(p.G + 1) * p.B + row] = sxx;
    }
}

// one workgroup: row_stats -> the reference's output scalars
template <int DT>
__global__ __launch_bounds__(1024) void reduce_scalars_kernel(int B, int A_t, int G, const float* __restrict__ row_stats,
                                                              const float* __restrict__ aux_rows,
                                                              const mtc_select_state* __restrict__ st,
                                                              const float* __restrict__ nnz_abs_override, double n_feat_total,
                                                              float l1_coeff, float aux_penalty, float* __restrict__ scalars) {
    typedef DType<DT> D_;
    __shared__ double sm[16];
    __shared__ double tot[MTC_MAX_PREFIX + 4];
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const int nq = G + 4;                    // G sse rows, sum x, sum x^2, sum |a|, aux
    for (int q = 0; q < nq; ++q) {
        const float* src = q < G + 3 ? row_stats + (size_t)q * B : aux_rows;
        double s = 0.0;
        if (src) for (int r = t; r < B; r += 1024) s += (double)src[r];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
        if (lane == 0) sm[w] = s;
        __syncthreads();
        if (t == 0) { double a = 0; for (int i = 0; i < 16; ++i) a += sm[i]; tot[q] = a; }
        __syncthreads();
    }
    if (t != 0) return;
    const double n = (double)B * (double)A_t;
    const double sx = tot[G], sxx = tot[G + 1];
    double abs_sum = tot[G + 2];
    double nnz = (double)st->nnz;
    if (nnz_abs_override) { nnz = (double)nnz_abs_override[0]; abs_sum = (double)nnz_abs_override[1]; }
    const float var_t = (float)((sxx - sx * sx / n) / (n - 1.0));          // unbiased, all elements (config.py:228)
    float fmin = 3.4e38f, fmax = -3.4e38f, fsum = 0.f, flast = 0.f;
    for (int i = 0; i < G; ++i) {
        const float f = (float)(tot[i] / n) / (var_t + 1e-8f);
        fmin = fminf(fmin, f); fmax = fmaxf(fmax, f); fsum += f; flast = f;
    }
    const float l2 = D_::rnd((float)(tot[G - 1] / n));
    const float l1 = D_::rnd((float)(abs_sum / ((double)B * n_feat_total)));
    const float l0 = (float)(nnz / (double)B);
    const float l1n = D_::rnd((float)(abs_sum / (double)B));
    float aux = 0.f, loss;
    if (aux_rows) {
        aux = aux_penalty * (float)(tot[G + 3] / n);
        loss = l2 + l1_coeff * l1 + aux;                                   // promoted to fp32 (sae.py:245,342)
    } else {
        loss = D_::rnd(D_::rnd(l2 + D_::rnd(l1_coeff * l1)) + 0.f);
    }
    scalars[0] = l2; scalars[1] = l1; scalars[2] = l0; scalars[3] = l1n; scalars[4] = aux; scalars[5] = loss;
    scalars[6] = flast; scalars[7] = fmin; scalars[8] = fmax; scalars[9] = fsum / (float)G;
    scalars[10] = var_t; scalars[11] = (float)nnz;
    // 12..15: for bf16 models, the model-dtype values a caller hands out (l2, l1, l1_norm, aux = 0, and the loss as the
    // reference forms it while no feature is dead: every op rounded, sae.py:245,331) packed as 8 bf16 -- the host
    // wrapper views them instead of launching a conversion of this vector every step
    if constexpr (DT == MTC_BF16) {
        const float loss_dt = D_::rnd(D_::rnd(l2 + D_::rnd(l1_coeff * l1)) + 0.f);
        uint32_t* pk = reinterpret_cast<uint32_t*>(scalars + 12);
        pk[0] = f32_to_bf16_bits(l2) | (f32_to_bf16_bits(l1) << 16);
        pk[1] = f32_to_bf16_bits(l1n);                     // (aux: +0 in the high half)
        pk[2] = f32_to_bf16_bits(loss_dt);
        pk[3] = 0u;
    } else {
        scalars[12] = scalars[13] = scalars[14] = scalars[15] = 0.f;
    }
}

static DecodeParams make_params(const mtc_desc* d, float l1_term) {
    DecodeParams p;
    p.B = d->B; p.A_t = d->A_t; p.G = d->G;
    for (int i = 0; i < MTC_MAX_PREFIX; ++i) p.prefix[i] = i < d->G ? d->prefix[i] : 0x7fffffff;
    p.prefix[d->G - 1] = 0x7fffffff;          // last group takes everything that is left
    p.unit_norm = d->unit_norm;
    p.inv_n = inv_n_in_dtype(d->dtype, (double)d->B * (double)d->A_t);
    p.l1_term = l1_term;
    return p;
}

}  // namespace mtc

using namespace mtc;

extern "C" int mtc_decode_fwd(const mtc_desc* d, const int32_t* row_ptr, const int32_t* col_idx, const void* vals,
                              const void* W_dec, const void* b_dec, const void* x_target, const void* mean,
                              const void* std, void* x_hat, void* g, float* row_stats, void* recons, float* d_vals,
                              float l1_term, void* stream) {
    if (check_desc(d, "mtc_decode_fwd")) return -1;
    MTC_REQUIRE(row_ptr && col_idx && vals && W_dec && b_dec && x_hat, "mtc_decode_fwd: null pointer");
    MTC_REQUIRE(x_target || (!g && !d_vals && !row_stats),
                "mtc_decode_fwd: g / d_vals / row_stats need x_target (pure decode passes all four as NULL)");
    MTC_REQUIRE(!x_target || row_stats, "mtc_decode_fwd: row_stats is required with x_target");
    MTC_REQUIRE(!d->unit_norm || (mean && std), "mtc_decode_fwd: unit_norm needs mean/std");
    const DecodeParams p = make_params(d, l1_term);
    DISPATCH_NC(d->dtype, nc_for(d->A_t), decode_fwd_kernel, dim3((d->B + 3) / 4), dim3(256), as_stream(stream), p,
                row_ptr, col_idx, vals, W_dec, b_dec, x_target, mean, std, x_hat, g, row_stats, recons, d_vals);
    return check_launch("mtc_decode_fwd");
}

extern "C" int mtc_decode_bwd_vals(const mtc_desc* d, const int32_t* row_ptr, const int32_t* col_idx, const void* W_dec,
                                   const void* g, float l1_term, float* d_vals, void* stream) {
    if (check_desc(d, "mtc_decode_bwd_vals")) return -1;
    MTC_REQUIRE(row_ptr && col_idx && W_dec && g && d_vals, "mtc_decode_bwd_vals: null pointer");
    DISPATCH_NC(d->dtype, nc_for(d->A_t), decode_bwd_vals_kernel, dim3((d->B + 3) / 4), dim3(256), as_stream(stream),
                d->B, d->A_t, row_ptr, col_idx, W_dec, g, l1_term, d_vals);
    return check_launch("mtc_decode_bwd_vals");
}

extern "C" int mtc_decode_partial(const mtc_desc* d, const int32_t* row_ptr, const int32_t* col_idx, const void* vals,
                                  const void* W_dec, int32_t rows_per_shard, float* partial_last, void* partial_pre,
                                  float* abs_rows, void* stream) {
    return mtc_decode_partial_strided(d, row_ptr, col_idx, vals, W_dec, rows_per_shard, 0, partial_last, partial_pre, abs_rows, stream);
}

extern "C" int mtc_decode_partial_strided(const mtc_desc* d, const int32_t* row_ptr, const int32_t* col_idx, const void* vals,
                                          const void* W_dec, int32_t rows_per_shard, int64_t pre_shard_stride,
                                          float* partial_last, void* partial_pre, float* abs_rows, void* stream) {
    if (check_desc(d, "mtc_decode_partial")) return -1;
    const int64_t dense_stride = (int64_t)(d->G - 1) * rows_per_shard * d->A_t;
    if (pre_shard_stride == 0) pre_shard_stride = dense_stride;
    MTC_REQUIRE(pre_shard_stride >= dense_stride, "mtc_decode_partial: pre_shard_stride (%lld) < (G-1) * rows_per_shard * A_t (%lld)",
                (long long)pre_shard_stride, (long long)dense_stride);
    MTC_REQUIRE(row_ptr && col_idx && vals && W_dec && partial_last && (partial_pre || d->G == 1),
                "mtc_decode_partial: null pointer");
    MTC_REQUIRE(rows_per_shard > 0 && d->B % rows_per_shard == 0,
                "mtc_decode_partial: rows_per_shard (%d) must divide B (%d)", rows_per_shard, d->B);
    const DecodeParams p = make_params(d, 0.f);
    DISPATCH_NC(d->dtype, nc_for(d->A_t), decode_partial_kernel, dim3((d->B + 3) / 4), dim3(256), as_stream(stream), p,
                row_ptr, col_idx, vals, W_dec, rows_per_shard, (long long)pre_shard_stride, partial_last, partial_pre, abs_rows);
    return check_launch("mtc_decode_partial");
}

extern "C" int mtc_decode_finish(const mtc_desc* d, int32_t B_total, const float* partial_last, const void* partial_pre,
                                 const void* b_dec, const void* x_target, const void* mean, const void* std, void* x_hat,
                                 void* g, float* row_stats, void* recons, void* stream) {
    if (check_desc(d, "mtc_decode_finish")) return -1;
    MTC_REQUIRE((partial_last || partial_pre) && b_dec && x_target && row_stats && (x_hat || !partial_last),
                "mtc_decode_finish: null pointer");
    MTC_REQUIRE(!d->unit_norm || (mean && std), "mtc_decode_finish: unit_norm needs mean/std");
    MTC_REQUIRE(B_total >= d->B, "mtc_decode_finish: B_total (%d) < local rows (%d)", B_total, d->B);
    DecodeParams p = make_params(d, 0.f);
    p.inv_n = inv_n_in_dtype(d->dtype, (double)B_total * (double)d->A_t);
    if (d->dtype == MTC_BF16)
        hipLaunchKernelGGL(decode_finish_kernel<MTC_BF16>, dim3((d->B + 3) / 4), dim3(256), 0, as_stream(stream), p,
                           partial_last, partial_pre, b_dec, x_target, mean, std, x_hat, g, row_stats, recons);
    else
        hipLaunchKernelGGL(decode_finish_kernel<MTC_F32>, dim3((d->B + 3) / 4), dim3(256), 0, as_stream(stream), p,
                           partial_last, partial_pre, b_dec, x_target, mean, std, x_hat, g, row_stats, recons);
    return check_launch("mtc_decode_finish");
}

extern "C" int mtc_reduce_scalars(const mtc_desc* d, const float* row_stats, const float* aux_rows,
                                  const mtc_select_state* st, const float* nnz_abs_override, int32_t n_feat_total,
                                  float l1_coeff, float aux_penalty, float* scalars, void* stream) {
    if (check_desc(d, "mtc_reduce_scalars")) return -1;
    MTC_REQUIRE(row_stats && st && scalars, "mtc_reduce_scalars: null pointer");
    if (d->dtype == MTC_BF16)
        hipLaunchKernelGGL(reduce_scalars_kernel<MTC_BF16>, dim3(1), dim3(1024), 0, as_stream(stream), d->B, d->A_t, d->G,
                           row_stats, aux_rows, st, nnz_abs_override, (double)n_feat_total, l1_coeff, aux_penalty, scalars);
    else
        hipLaunchKernelGGL(reduce_scalars_kernel<MTC_F32>, dim3(1), dim3(1024), 0, as_stream(stream), d->B, d->A_t, d->G,
                           row_stats, aux_rows, st, nnz_abs_override, (double)n_feat_total, l1_coeff, aux_penalty, scalars);
    return check_launch("mtc_reduce_scalars");
}
