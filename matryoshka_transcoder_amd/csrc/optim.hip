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
#include <math.h>

namespace mtc {

struct AdamK {
    float w1;          // 1 - beta1
    float beta2;
    float w2;          // 1 - beta2
    float bc2_sqrt;    // sqrt(1 - beta2^step)
    float eps;
    float neg_step;    // -(lr / (1 - beta1^step))
};

template <int DT>
__device__ __forceinline__ void adam_elem(float& p, float g, float& m, float& v, const AdamK& k) {
    typedef DType<DT> D_;
    m = D_::rnd(fmaf(k.w1, g - m, m));
    v = D_::rnd(v * k.beta2);
    v = D_::rnd(fmaf(k.w2 * g, g, v));
    float den = D_::rnd(sqrtf(v));
    den = D_::rnd(den / k.bc2_sqrt);
    den = D_::rnd(den + k.eps);
    p = D_::rnd(fmaf(k.neg_step, m / den, p));
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
    const int j = blockIdx.x * 4 + (threadIdx.x >> 6);
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
                w[c][i] = D_::rnd(w[c][i] / norm);                // W_dec_normed
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

static AdamK make_adam(const mtc_adam_hyper* h) {
    AdamK k;
    const double bc1 = 1.0 - pow(h->beta1, (double)h->step);
    const double bc2 = 1.0 - pow(h->beta2, (double)h->step);
    k.w1 = (float)(1.0 - h->beta1);
    k.beta2 = (float)h->beta2;
    k.w2 = (float)(1.0 - h->beta2);
    k.bc2_sqrt = (float)sqrt(bc2);
    k.eps = (float)h->eps;
    k.neg_step = (float)(-(h->lr / bc1));
    return k;
}

static int nc_for(int A) { return ((A >> 3) + 63) / 64; }

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
