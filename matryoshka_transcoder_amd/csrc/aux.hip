// K3b: auxiliary dead-feature loss (sae.py:318-347) in sparse form.
//
//   dead     = num_batches_not_active >= n_batches_to_dead           -> compact ascending index list
//   per row  : top-min(k_aux, n_dead) of acts[b, dead]               -> per-row radix select in LDS
//   x_aux_b  = sum_j a_bj W_dec[j]          (no b_dec)               -> same row gather as the decode
//   e_b      = round(x_aux_b) - (x_t - x_hat)_b   (fp32; the residual is NOT detached)
//   aux      = aux_penalty * mean(e^2)
//   g_aux    = 2 aux_penalty / (B A_t) * e ;  g += g_aux * std_row ;  d a_bj = g_aux_b . W_dec[j]
// Zeros are never selected (a zero activation contributes nothing forward or backward); ties at a
// row's threshold are taken in ascending feature order.  n_dead stays on the device.
#include "common.hpp"
#include "rowops.hpp"

namespace mtc {

// ---------------------------------------------------------------- dead list (one workgroup)
// Chunks of 1024 consecutive features (coalesced counter reads); pass 1 leaves one count per (chunk, wave) in LDS, wave 0
// turns them into exclusive offsets, pass 2 re-reads the counters (L2) and writes the ordered list, the feature -> slot
// map and the gatherable tail.
__global__ __launch_bounds__(1024) void dead_list_kernel(int D, const float* __restrict__ cnt, float n_to_dead,
                                                         int32_t* __restrict__ dead_idx, int32_t* __restrict__ n_dead,
                                                         int32_t* __restrict__ row_map /* optional [D]: slot in the list or -1 */,
                                                         int32_t* __restrict__ next_bound /* optional: #features with cnt >= n_to_dead - 1 */) {
    extern __shared__ uint32_t wc[];             // [chunks * 16]
    __shared__ uint32_t sm_near[16];
    __shared__ uint32_t s_total;
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const int chunks = (D + 1023) / 1024;
    uint32_t near = 0;
    for (int c0 = 0; c0 < chunks; c0 += 8) {          // 8 chunks' reads in flight (one workgroup: nothing else hides the latency)
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) { const int j = (c0 + u) * 1024 + t; v[u] = j < D ? cnt[j] : -1.f; }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (c0 + u >= chunks) break;
            near += v[u] >= n_to_dead - 1.f ? 1u : 0u;  // a counter grows by at most 1 per forward: the next list's length bound
            const uint64_t b = __ballot(v[u] >= n_to_dead);
            if (lane == 0) wc[(c0 + u) * 16 + w] = (uint32_t)__popcll(b);
        }
    }
    near = wave_sum_u32(near);
    if (lane == 0) sm_near[w] = near;
    __syncthreads();
    if (w == 0) {
        uint32_t running = 0;
        const int n = chunks * 16;
        for (int base = 0; base < n; base += 64) {
            const uint32_t v = base + lane < n ? wc[base + lane] : 0u;
            uint32_t inc = v;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) { const uint32_t u = __shfl_up(inc, o, 64); if (lane >= o) inc += u; }
            if (base + lane < n) wc[base + lane] = running + inc - v;
            running += __shfl(inc, 63, 64);
        }
        if (lane == 0) s_total = running;
    }
    __syncthreads();
    const uint32_t total = s_total;
    for (int c0 = 0; c0 < chunks; c0 += 8) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) { const int j = (c0 + u) * 1024 + t; v[u] = j < D ? cnt[j] : -1.f; }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int c = c0 + u;
            if (c >= chunks) break;
            const int j = c * 1024 + t;
            const bool dead = v[u] >= n_to_dead;       // (j >= D reads as -1: never dead)
            const uint64_t b = __ballot(dead);
            const uint32_t pos = wc[c * 16 + w] + (uint32_t)__popcll(b & ((1ull << lane) - 1ull));
            if (j < D) {
                if (row_map) row_map[j] = dead ? (int32_t)pos : -1;
                if (dead) dead_idx[pos] = j;
            }
        }
    }
    // the tail of the list holds a valid feature (0): callers that size their launches from an upper bound of the
    // count gather rows through it (those rows only ever meet zero activations / masked gradients)
    for (int i = (int)total + t; i < D; i += 1024) dead_idx[i] = 0;
    if (t == 0) {
        uint32_t near_total = 0;
        for (int i = 0; i < 16; ++i) near_total += sm_near[i];
        *n_dead = (int32_t)total;
        if (next_bound) *next_bound = (int32_t)near_total;
    }
}

// ---------------------------------------------------------------- per-row top-k over the dead columns
// one workgroup (256 threads) per row; 4096-bin LDS histogram per radix level
template <int DT>
__global__ __launch_bounds__(256) void aux_select_kernel(const void* __restrict__ acts, int D, const int32_t* __restrict__ dead_idx,
                                                         const int32_t* __restrict__ n_dead_p, int k_aux_max,
                                                         uint32_t* __restrict__ row_cnt, uint32_t* __restrict__ row_tau,
                                                         uint32_t* __restrict__ row_quota) {
    __shared__ uint32_t hist[4096];
    __shared__ uint32_t tsum[256];
    __shared__ uint32_t s_bin, s_above, s_total;
    const int t = threadIdx.x;
    const int row = blockIdx.x;
    const int n_dead = *n_dead_p;
    const uint32_t k_aux = (uint32_t)min(k_aux_max, n_dead);
    if (n_dead == 0 || k_aux == 0) {
        if (t == 0) { row_cnt[row] = 0; row_tau[row] = 0; row_quota[row] = 0; }
        return;
    }
    const size_t base = (size_t)row * D;
    const int levels = DT == MTC_BF16 ? 2 : 3;
    uint32_t prefix = 0, k_rem = k_aux, n_gt = 0;
    bool all_pos = false;
    for (int level = 0; level < levels && !all_pos; ++level) {
        const int sh = level == 0 ? 19 : (level == 1 ? 7 : 0);
        const uint32_t fm = level == 2 ? 0x7fu : 0xfffu;
        const uint32_t pm = level == 0 ? 0u : (level == 1 ? 0xfff80000u : 0xffffff80u);
        for (int i = t; i < 4096; i += 256) hist[i] = 0;
        __syncthreads();
        for (int i = t; i < n_dead; i += 256) {
            const uint32_t k = key_of<DT>(acts, base + dead_idx[i]);
            if (k != 0 && k <= 0x7f800000u && ((k ^ prefix) & pm) == 0) atomicAdd(&hist[(k >> sh) & fm], 1u);
        }
        __syncthreads();
        // thread t owns bins [16t, 16t+15]; suffix scan over threads
        uint32_t mine = 0;
#pragma unroll
        for (int i = 0; i < 16; ++i) mine += hist[16 * t + i];
        tsum[t] = mine;
        if (t == 0) { s_bin = 0xffffffffu; s_above = 0; }
        __syncthreads();
        for (int off = 1; off < 256; off <<= 1) {
            const uint32_t add = t + off < 256 ? tsum[t + off] : 0u;
            __syncthreads();
            tsum[t] += add;
            __syncthreads();
        }
        uint32_t above = t + 1 < 256 ? tsum[t + 1] : 0u;
        for (int i = 15; i >= 0; --i) {
            const uint32_t h = hist[16 * t + i];
            if (above < k_rem && k_rem <= above + h) { s_bin = 16 * t + i; s_above = above; }
            above += h;
        }
        if (t == 0) s_total = tsum[0];
        __syncthreads();
        if (s_bin == 0xffffffffu) {            // fewer positives than k_aux: take them all
            all_pos = true; n_gt = s_total;
        } else {
            n_gt += s_above; k_rem -= s_above; prefix |= s_bin << sh;
        }
        __syncthreads();
    }
    if (t == 0) {
        row_cnt[row] = all_pos ? n_gt : k_aux;
        row_tau[row] = all_pos ? 0u : prefix;
        row_quota[row] = all_pos ? 0u : k_rem;
    }
}

// wave per row: ordered compaction of the row's aux entries
template <int DT>
__global__ __launch_bounds__(256) void aux_fill_kernel(const void* __restrict__ acts, int B, int D,
                                                       const int32_t* __restrict__ dead_idx, const int32_t* __restrict__ n_dead_p,
                                                       const uint32_t* __restrict__ row_tau, const uint32_t* __restrict__ row_quota,
                                                       const int32_t* __restrict__ row_ptr, int32_t* __restrict__ col_idx,
                                                       void* __restrict__ vals) {
    typedef typename DType<DT>::T T;
    const int lane = threadIdx.x & 63;
    const int row = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
    if (row >= B) return;
    const int n_dead = *n_dead_p;
    const uint32_t tau = row_tau[row], quota = row_quota[row];
    const size_t base = (size_t)row * D;
    uint32_t out = (uint32_t)row_ptr[row], ties_seen = 0;
    T* vout = reinterpret_cast<T*>(vals);
    for (int i0 = 0; i0 < n_dead; i0 += 64) {
        const int i = i0 + lane;
        int col = 0; uint32_t k = 0;
        if (i < n_dead) { col = dead_idx[i]; k = key_of<DT>(acts, base + col); }
        const bool ok = k != 0 && k <= 0x7f800000u;
        const bool gt = ok && k > tau;
        const bool eq = ok && quota && k == tau;
        bool sel = gt;
        const uint64_t eqb = __ballot(eq);
        if (eqb) {
            const uint32_t rank = ties_seen + (uint32_t)__popcll(eqb & ((1ull << lane) - 1ull));
            if (eq && rank < quota) sel = true;
            ties_seen += (uint32_t)__popcll(eqb);
        }
        const uint64_t sb = __ballot(sel);
        if (sel) {
            const uint32_t pos = out + (uint32_t)__popcll(sb & ((1ull << lane) - 1ull));
            col_idx[pos] = col;
            vout[pos] = reinterpret_cast<const T*>(acts)[base + col];
        }
        out += (uint32_t)__popcll(sb);
    }
}

// wave per row: aux reconstruction, residual error, loss partial, gradient updates, d(aux acts)
template <int DT, int NC>
__global__ __launch_bounds__(256) void aux_fwd_kernel(int B, int A_t, int unit_norm, float coef_aux /* fp32 aux_penalty/(B*A_t) */,
                                                      float inv_n /* 1/(B*A_t) in the model dtype */,
                                                      const int32_t* __restrict__ row_ptr, const int32_t* __restrict__ col_idx,
                                                      const void* __restrict__ vals, const void* __restrict__ W_dec,
                                                      const void* __restrict__ x_target, const void* __restrict__ x_hat,
                                                      const void* __restrict__ std_p, void* __restrict__ g,
                                                      void* __restrict__ g_aux, float* __restrict__ aux_rows,
                                                      float* __restrict__ d_aux_vals,
                                                      const void* __restrict__ x_hat_aux /* optional: reconstruction made elsewhere (dense GEMM path); then no gather */,
                                                      const int32_t* __restrict__ active /* optional: *active == 0 -> the branch is off (sae.py:331) */) {
    typedef DType<DT> D_;
    typedef typename D_::T T;
    RowCtx<DT, NC> rc;
    rc.lane = threadIdx.x & 63;
    rc.nch = A_t >> 3;
    const int row = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
    if (row >= B) return;
    if (active && *active == 0) {                 // no dead feature after all: zero loss, g untouched, zero g_aux
        if (rc.lane == 0) aux_rows[row] = 0.f;
        if (g_aux) {
            const float z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
            for (int c = 0; c < NC; ++c)
                if (rc.valid(c)) store8<DT>(g_aux, (size_t)row * A_t + rc.off(c), z);
        }
        return;
    }
    const int e_begin = x_hat_aux ? 0 : row_ptr[row], e_end = x_hat_aux ? 0 : row_ptr[row + 1];
    float acc[NC][8];
#pragma unroll
    for (int c = 0; c < NC; ++c)
#pragma unroll
        for (int k = 0; k < 8; ++k) acc[c][k] = 0.f;
    const T* vp = reinterpret_cast<const T*>(vals);
    for (int e0 = e_begin; e0 < e_end; e0 += 64) {
        const int nb = min(64, e_end - e0);
        const int my_col = rc.lane < nb ? col_idx[e0 + rc.lane] : 0;
        const float my_val = rc.lane < nb ? D_::ld(vp + e0 + rc.lane) : 0.f;
        accumulate_run<DT, NC>(rc, W_dec, A_t, my_col, my_val, 0, nb, acc);
    }
    const float stdv = unit_norm ? D_::ld(reinterpret_cast<const T*>(std_p) + row) : 1.f;
    float sse = 0.f;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        if (rc.valid(c)) {
            const size_t off = (size_t)row * A_t + rc.off(c);
            float t[8], h[8], gg[8] = {0, 0, 0, 0, 0, 0, 0, 0}, ga[8];
            load8<DT>(x_target, off, t);
            load8<DT>(x_hat, off, h);
            if (g) load8<DT>(g, off, gg);
            if (x_hat_aux) load8<DT>(x_hat_aux, off, acc[c]);
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const float e = D_::rnd(acc[c][k]) - (t[k] - h[k]);
                sse = fmaf(e, e, sse);
                // fp32 branch of the loss (sae.py:336-341): d aux / d recon_aux = fl32(penalty / N) * 2 e; it reaches the
                // model-dtype tensors rounded (the .float() casts' backward), and x_hat's gradient is the model-dtype SUM
                // of its two consumers' gradients: round(g_post(l2) + round(ga)), then through the de-normalisation
                ga[k] = D_::rnd(coef_aux * (2.f * e));
                acc[c][k] = ga[k];                       // acc now holds g_aux for the second sweep
                float gp = D_::rnd(g_post_of<DT>(h[k], t[k], inv_n) + ga[k]);
                gg[k] = unit_norm ? D_::rnd(gp * stdv) : gp;
            }
            if (g) store8<DT>(g, off, gg);
            if (g_aux) store8<DT>(g_aux, off, ga);
        }
    }
    sse = wave_sum(sse);
    if (rc.lane == 0) aux_rows[row] = sse;
    if (d_aux_vals) {
        Raw<DT> gp[NC];                       // g_aux as stored (model dtype), zero beyond the row
#pragma unroll
        for (int c = 0; c < NC; ++c) gp[c] = rc.valid(c) ? pack_raw<DT>(acc[c]) : zero_raw<DT>();
        for (int e0 = e_begin; e0 < e_end; e0 += 64) {
            const int nb = min(64, e_end - e0);
            const int my_col = rc.lane < nb ? col_idx[e0 + rc.lane] : 0;
            dot_run<DT, NC>(rc, W_dec, A_t, my_col, nb, gp, 0.f, d_aux_vals + e0);
        }
    }
}

// Dense form of the aux selection for the GEMM path: out[row][i] = acts[row][dead_idx[i]] if that entry is among the
// row's top k_aux dead activations (same threshold / tie quota as aux_fill_kernel), else 0; columns i >= n_dead
// (padding up to n_pad) are 0.  Wave per row.
template <int DT>
__global__ __launch_bounds__(256) void aux_dense_fill_kernel(const void* __restrict__ acts, int B, int D, int n_pad,
                                                             const int32_t* __restrict__ dead_idx,
                                                             const int32_t* __restrict__ n_dead_p,
                                                             const uint32_t* __restrict__ row_tau,
                                                             const uint32_t* __restrict__ row_quota, void* __restrict__ out) {
    typedef typename DType<DT>::T T;
    const int lane = threadIdx.x & 63;
    const int row = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
    if (row >= B) return;
    const int n_dead = min(*n_dead_p, n_pad);
    const uint32_t tau = row_tau[row], quota = row_quota[row];
    const size_t base = (size_t)row * D;
    T* o = reinterpret_cast<T*>(out) + (size_t)row * n_pad;
    uint32_t ties_seen = 0;
    // four 64-column groups per trip: their index and activation reads are requested together (the two dependent reads
    // of a group would otherwise cost a full round trip per 64 columns); the groups are then examined in column order
    for (int i0 = 0; i0 < n_pad; i0 += 256) {
        int col[4];
        T raw[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) { const int i = i0 + 64 * u + lane; col[u] = i < n_dead ? dead_idx[i] : -1; }
#pragma unroll
        for (int u = 0; u < 4; ++u) raw[u] = col[u] >= 0 ? reinterpret_cast<const T*>(acts)[base + col[u]] : (T)0;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = i0 + 64 * u + lane;
            if (i0 + 64 * u >= n_pad) break;             // uniform
            uint32_t k = 0;
            if (col[u] >= 0) {
                if constexpr (DT == MTC_BF16) k = (uint32_t)raw[u] << 16;
                else k = __float_as_uint(raw[u]);
            }
            const bool ok = k != 0 && k <= 0x7f800000u;
            const bool eq = ok && quota && k == tau;
            bool sel = ok && k > tau;
            const uint64_t eqb = __ballot(eq);
            if (eqb) {
                const uint32_t rank = ties_seen + (uint32_t)__popcll(eqb & ((1ull << lane) - 1ull));
                if (eq && rank < quota) sel = true;
                ties_seen += (uint32_t)__popcll(eqb);
            }
            if (i < n_pad) o[i] = sel ? raw[u] : (T)0;
        }
    }
}

// aux_select + aux_dense_fill in ONE pass over the activations: one workgroup per batch row gathers the row's dead
// columns ONCE into LDS (raw bits; the two-launch form reads them three times: two radix levels + the fill), runs the
// radix levels on the LDS copy, resolves the tie quota in dead-list order and writes the dense row with 8-byte stores.
// Same threshold / quota / selection as the two kernels above (tests compare them bit for bit).
//   LDS: hist 16 KB + keys n_pad * sizeof(T) (dynamic)
#ifndef MTC_AUX_GU
#define MTC_AUX_GU 4
#endif
template <int DT>
__global__ __launch_bounds__(256) void aux_select_dense_kernel(const void* __restrict__ acts, int D, int n_pad,
                                                               const int32_t* __restrict__ dead_idx,
                                                               const int32_t* __restrict__ n_dead_p, int k_aux_max,
                                                               uint32_t* __restrict__ row_cnt, uint32_t* __restrict__ row_tau,
                                                               uint32_t* __restrict__ row_quota, void* __restrict__ out) {
    typedef typename DType<DT>::T T;
    constexpr int EPD = 4 / (int)sizeof(T);                  // elements per dword
    extern __shared__ __align__(16) uint32_t kw[];           // [n_pad / EPD] raw bits of the row's dead columns (0 beyond n_dead)
    // 4096 bins as packed 16-bit pairs (a row holds <= 20 480 entries here: the one-pass form needs the row in 40 KB of LDS),
    // which brings the workgroup under 20 KB: 8 workgroups per CU instead of 5 hide each other's selection phases
    __shared__ uint32_t hist[2048];
    __shared__ uint32_t wtot[4];
    __shared__ uint32_t s_bin, s_above, s_h, s_cut;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int row = blockIdx.x;
    const int n_dead = min(*n_dead_p, n_pad);
    const uint32_t k_aux = (uint32_t)min(k_aux_max, n_dead);
    const int nw = n_pad / EPD;                               // host: n_pad % (2 EPD) == 0
    uint2* o2 = reinterpret_cast<uint2*>(reinterpret_cast<T*>(out) + (size_t)row * n_pad);
    if (n_dead == 0 || k_aux == 0) {
        for (int w = 2 * t; w < nw; w += 512) o2[w >> 1] = make_uint2(0u, 0u);
        if (t == 0) { row_cnt[row] = 0; row_tau[row] = 0; row_quota[row] = 0; }
        return;
    }
    const T* a = reinterpret_cast<const T*>(acts) + (size_t)row * D;
    auto key_bits = [](uint32_t word, int e) -> uint32_t {   // radix key of element e of a dword
        if constexpr (DT == MTC_BF16) return ((word >> (16 * e)) & 0xffffu) << 16;
        else return word;
    };
    for (int i = t; i < 2048; i += 256) hist[i] = 0;
    if (t == 0) { s_bin = 0xffffffffu; s_above = 0; s_h = 0; s_cut = 0xffffffffu; }
    __syncthreads();
    constexpr int GU = MTC_AUX_GU;
    // ---- gather the row's dead columns (MTC_AUX_GU dwords per thread in flight: the pass is a chain of index -> value round trips), level-0 histogram on the way
    for (int w0 = t; w0 < nw; w0 += 256 * GU) {
        int col[GU][EPD];
#pragma unroll
        for (int u = 0; u < GU; ++u) {
            const int w = w0 + 256 * u;
#pragma unroll
            for (int e = 0; e < EPD; ++e) col[u][e] = -1;
            if (w < nw) {
                if constexpr (EPD == 2) {
                    const int2 c2 = reinterpret_cast<const int2*>(dead_idx)[w];   // the list's tail is gatherable (feature 0)
                    col[u][0] = 2 * w < n_dead ? c2.x : -1;
                    col[u][1] = 2 * w + 1 < n_dead ? c2.y : -1;
                } else {
                    col[u][0] = w < n_dead ? dead_idx[w] : -1;
                }
            }
        }
        uint32_t word[GU];
#pragma unroll
        for (int u = 0; u < GU; ++u) {
            if constexpr (EPD == 2) {
                const uint32_t lo = col[u][0] >= 0 ? (uint32_t)a[col[u][0]] : 0u;
                const uint32_t hi = col[u][1] >= 0 ? (uint32_t)a[col[u][1]] : 0u;
                word[u] = lo | (hi << 16);
            } else {
                word[u] = col[u][0] >= 0 ? __float_as_uint(a[col[u][0]]) : 0u;
            }
        }
#pragma unroll
        for (int u = 0; u < GU; ++u) {
            const int w = w0 + 256 * u;
            if (w < nw) {
                kw[w] = word[u];
#pragma unroll
                for (int e = 0; e < EPD; ++e) {
                    const uint32_t k = key_bits(word[u], e);
                    if (k != 0 && k <= 0x7f800000u) { const uint32_t b = (k >> 19) & 0xfffu; atomicAdd(&hist[b >> 1], (b & 1u) ? 0x10000u : 1u); }
                }
            }
        }
    }
    __syncthreads();
    const int nwd = (n_dead + EPD - 1) / EPD;                 // dwords that hold list entries
    const int levels = DT == MTC_BF16 ? 2 : 3;
    uint32_t prefix = 0, k_rem = k_aux, n_gt = 0;
    bool all_pos = false;
    for (int level = 0; level < levels && !all_pos; ++level) {
        const int sh = level == 0 ? 19 : (level == 1 ? 7 : 0);
        const uint32_t fm = level == 2 ? 0x7fu : 0xfffu;
        const uint32_t pm = level == 0 ? 0u : (level == 1 ? 0xfff80000u : 0xffffff80u);
        if (level > 0) {
            for (int i = t; i < 2048; i += 256) hist[i] = 0;
            if (t == 0) s_bin = 0xffffffffu;
            __syncthreads();
            for (int w = t; w < nwd; w += 256) {
                const uint32_t word = kw[w];
#pragma unroll
                for (int e = 0; e < EPD; ++e) {
                    const uint32_t k = key_bits(word, e);
                    if (k != 0 && k <= 0x7f800000u && ((k ^ prefix) & pm) == 0) {
                        const uint32_t b = (k >> sh) & fm;
                        atomicAdd(&hist[b >> 1], (b & 1u) ? 0x10000u : 1u);
                    }
                }
            }
            __syncthreads();
        }
        // thread t owns bins [16t, 16t+15]; suffix sums over the threads: in the wave by shuffles, across waves through LDS
        uint32_t h[16], mine = 0;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const uint32_t pair = hist[8 * t + (i >> 1)];
            h[i] = (i & 1) ? pair >> 16 : pair & 0xffffu;
            mine += h[i];
        }
        uint32_t suf = mine;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t o = __shfl_down(suf, off);
            if (lane + off < 64) suf += o;
        }
        if (lane == 0) wtot[wave] = suf;
        __syncthreads();
        uint32_t above = suf - mine, total = 0;
#pragma unroll
        for (int w2 = 0; w2 < 4; ++w2) { total += wtot[w2]; if (w2 > wave) above += wtot[w2]; }
#pragma unroll
        for (int i = 15; i >= 0; --i) {
            if (above < k_rem && k_rem <= above + h[i]) { s_bin = 16 * t + i; s_above = above; s_h = h[i]; }
            above += h[i];
        }
        __syncthreads();
        if (s_bin == 0xffffffffu) {            // fewer positives than k_aux: take them all
            all_pos = true; n_gt = total;
        } else {
            n_gt += s_above; k_rem -= s_above; prefix |= s_bin << sh;
        }
        __syncthreads();
    }
    const uint32_t tau = all_pos ? 0u : prefix, quota = all_pos ? 0u : k_rem;
    if (t == 0) {
        row_cnt[row] = all_pos ? n_gt : k_aux;
        row_tau[row] = tau;
        row_quota[row] = quota;
    }
    // ---- ties at tau beyond the quota: the first `quota` of them in list order are taken.  Only then is the order needed:
    // every thread counts the ties of one contiguous piece of the list (odd dword pitch: conflict-free), an exclusive
    // scan over the threads finds the piece holding the quota-th tie, whose owner finds its position
    if (!all_pos && quota < s_h) {
        const int lw = ((nwd + 255) / 256) | 1;
        const int wb = t * lw, we = min(wb + lw, nwd);
        uint32_t c = 0;
        for (int w = wb; w < we; ++w) {
            const uint32_t word = kw[w];
#pragma unroll
            for (int e = 0; e < EPD; ++e) c += key_bits(word, e) == tau ? 1u : 0u;
        }
        uint32_t inc = c;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t o = __shfl_up(inc, off);
            if (lane >= off) inc += o;
        }
        if (lane == 63) wtot[wave] = inc;
        __syncthreads();
        uint32_t excl = inc - c;
#pragma unroll
        for (int w2 = 0; w2 < 4; ++w2) if (w2 < wave) excl += wtot[w2];
        if (excl < quota && quota <= excl + c) {
            uint32_t need = quota - excl;
            for (int w = wb; w < we && need; ++w) {
                const uint32_t word = kw[w];
#pragma unroll
                for (int e = 0; e < EPD; ++e)
                    if (need && key_bits(word, e) == tau && --need == 0) s_cut = (uint32_t)(w * EPD + e);
            }
        }
        __syncthreads();
    }
    const uint32_t cut = s_cut;               // index of the last tie taken (0xffffffff: all of them)
    for (int w = 2 * t; w < nw; w += 512) {
        const uint2 in = *reinterpret_cast<const uint2*>(&kw[w]);
        uint32_t wd[2] = {in.x, in.y};
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            uint32_t keep = 0;
#pragma unroll
            for (int e = 0; e < EPD; ++e) {
                const uint32_t k = key_bits(wd[q], e);
                const uint32_t i = (uint32_t)((w + q) * EPD + e);
                const bool ok = k != 0 && k <= 0x7f800000u;
                const bool sel = ok && (k > tau || (quota && k == tau && i <= cut));
                if (sel) keep |= EPD == 2 ? (0xffffu << (16 * e)) : 0xffffffffu;
            }
            wd[q] &= keep;
        }
        o2[w >> 1] = make_uint2(wd[0], wd[1]);
    }
}

// Feature-sharded aux selection, step 1: the values of a row's local top-k_aux dead activations (threshold / tie quota
// from aux_select on the LOCAL dead columns), left-packed into cand[row][0..k_pad) and zero-padded -- what each rank
// contributes to the all-gather from which the global per-row threshold is found.  Wave per row.
template <int DT>
__global__ __launch_bounds__(256) void aux_row_candidates_kernel(const void* __restrict__ acts, int B, int D, int k_pad,
                                                                 const int32_t* __restrict__ dead_idx,
                                                                 const int32_t* __restrict__ n_dead_p,
                                                                 const uint32_t* __restrict__ row_tau,
                                                                 const uint32_t* __restrict__ row_quota, void* __restrict__ cand) {
    typedef typename DType<DT>::T T;
    const int lane = threadIdx.x & 63;
    const int row = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
    if (row >= B) return;
    const int n_dead = *n_dead_p;
    const uint32_t tau = row_tau[row], quota = row_quota[row];
    const size_t base = (size_t)row * D;
    T* o = reinterpret_cast<T*>(cand) + (size_t)row * k_pad;
    uint32_t out = 0, ties_seen = 0;
    for (int i0 = 0; i0 < n_dead; i0 += 64) {
        const int i = i0 + lane;
        uint32_t k = 0; T raw = 0;
        if (i < n_dead) { const int col = dead_idx[i]; raw = reinterpret_cast<const T*>(acts)[base + col]; k = key_of<DT>(acts, base + col); }
        const bool ok = k != 0 && k <= 0x7f800000u;
        const bool eq = ok && quota && k == tau;
        bool sel = ok && k > tau;
        const uint64_t eqb = __ballot(eq);
        if (eqb) {
            const uint32_t rank = ties_seen + (uint32_t)__popcll(eqb & ((1ull << lane) - 1ull));
            if (eq && rank < quota) sel = true;
            ties_seen += (uint32_t)__popcll(eqb);
        }
        const uint64_t sb = __ballot(sel);
        if (sel) {
            const uint32_t pos = out + (uint32_t)__popcll(sb & ((1ull << lane) - 1ull));
            if (pos < (uint32_t)k_pad) o[pos] = raw;
        }
        out += (uint32_t)__popcll(sb);
    }
    for (uint32_t i = min(out, (uint32_t)k_pad) + lane; i < (uint32_t)k_pad; i += 64) o[i] = (T)0;
}

// step 2: per row, how many local dead activations lie above / exactly at the GLOBAL threshold (bits; 0 = take every
// positive one).  Wave per row.
template <int DT>
__global__ __launch_bounds__(256) void aux_count_at_kernel(const void* __restrict__ acts, int B, int D,
                                                           const int32_t* __restrict__ dead_idx,
                                                           const int32_t* __restrict__ n_dead_p,
                                                           const uint32_t* __restrict__ row_tau, int32_t* __restrict__ row_gt,
                                                           int32_t* __restrict__ row_eq) {
    const int lane = threadIdx.x & 63;
    const int row = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
    if (row >= B) return;
    const int n_dead = *n_dead_p;
    const uint32_t tau = row_tau[row];
    const size_t base = (size_t)row * D;
    uint32_t gt = 0, eq = 0;
    for (int i = lane; i < n_dead; i += 64) {
        const uint32_t k = key_of<DT>(acts, base + dead_idx[i]);
        const bool ok = k != 0 && k <= 0x7f800000u;
        gt += (ok && k > tau) ? 1u : 0u;
        eq += (ok && tau != 0 && k == tau) ? 1u : 0u;
    }
    gt = wave_sum_u32(gt); eq = wave_sum_u32(eq);
    if (lane == 0) { row_gt[row] = (int32_t)gt; row_eq[row] = (int32_t)eq; }
}

// the same with the ties counted per prefix group (row_eq [B][G]): a feature-sharded caller grants a row's ties in
// ascending GLOBAL feature order = (group, shard, local feature), for which it needs every shard's count per group
struct AuxGroupBounds { int G; int prefix[MTC_MAX_PREFIX]; };
template <int DT>
__global__ __launch_bounds__(256) void aux_count_at_groups_kernel(const void* __restrict__ acts, int B, int D,
                                                                  const int32_t* __restrict__ dead_idx,
                                                                  const int32_t* __restrict__ n_dead_p,
                                                                  const uint32_t* __restrict__ row_tau, AuxGroupBounds gb,
                                                                  int32_t* __restrict__ row_gt, int32_t* __restrict__ row_eq) {
    const int lane = threadIdx.x & 63;
    const int row = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
    if (row >= B) return;
    const int n_dead = *n_dead_p;
    const uint32_t tau = row_tau[row];
    const size_t base = (size_t)row * D;
    uint32_t gt = 0;
    uint32_t eq[MTC_MAX_PREFIX];
#pragma unroll
    for (int q = 0; q < MTC_MAX_PREFIX; ++q) eq[q] = 0;
    for (int i = lane; i < n_dead; i += 64) {
        const int col = dead_idx[i];
        const uint32_t k = key_of<DT>(acts, base + col);
        const bool ok = k != 0 && k <= 0x7f800000u;
        gt += (ok && k > tau) ? 1u : 0u;
        if (ok && tau != 0 && k == tau) {
            int g = 0;
            for (int q = 0; q + 1 < gb.G; ++q) g += col >= gb.prefix[q] ? 1 : 0;
#pragma unroll
            for (int q = 0; q < MTC_MAX_PREFIX; ++q) eq[q] += (g == q) ? 1u : 0u;
        }
    }
    gt = wave_sum_u32(gt);
    for (int q = 0; q < gb.G; ++q) {
        uint32_t e = 0;
#pragma unroll
        for (int z = 0; z < MTC_MAX_PREFIX; ++z) e = z == q ? eq[z] : e;
        e = wave_sum_u32(e);
        if (lane == 0) row_eq[(size_t)row * gb.G + q] = (int32_t)e;
    }
    if (lane == 0) row_gt[row] = (int32_t)gt;
}

// d[i] = 0 where the aux activation is not selected (ReLU / TopK mask of the aux branch's backward)
template <int DT>
__global__ __launch_bounds__(256) void aux_mask_kernel(size_t n, const void* __restrict__ acts_aux, void* __restrict__ dvals) {
    typedef typename DType<DT>::T T;
    const T* a = reinterpret_cast<const T*>(acts_aux);
    T* dv = reinterpret_cast<T*>(dvals);
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
        if (DType<DT>::ld(a + i) <= 0.f) dv[i] = (T)0;
}

// dst[c][r] = src[row(r)][c] for r < R, c < C (row(r) = src_rows[r] when given: gather + transpose in one pass);
// columns R <= r < R_pad of dst are zero-filled (GEMM operands padded to the k-tile).  128 x 128 (bf16) / 64 x 64 (fp32)
// tiles through LDS: both the reads (rows of src) and the writes (rows of dst) move 256-byte row pieces, 4 elements per lane.
template <typename T>
__global__ __launch_bounds__(256) void transpose_rows_kernel(int R, int C, int R_pad, const T* __restrict__ src, long long ld_src,
                                                             const int32_t* __restrict__ src_rows, T* __restrict__ dst,
                                                             long long ld_dst) {
    struct alignas(4 * sizeof(T)) Vec4 { T e[4]; };
    constexpr int TS = sizeof(T) == 2 ? 128 : 64;               // 33 KB / 17 KB of LDS
    __shared__ T tile[TS][TS + (sizeof(T) == 2 ? 2 : 1)];      // row pitch = 65 / 65 dwords (odd): column reads spread over the banks
    const int r0 = blockIdx.x * TS, c0 = blockIdx.y * TS;
    const int t = threadIdx.x;
    // load: thread covers 4 consecutive columns of one row per pass; TS / 4 threads span a row piece
    constexpr int TPR = TS / 4, RPP = 256 / TPR;               // threads per row piece, rows per pass
    {
        const int cq = (t % TPR) * 4, rr = t / TPR;
        for (int i = rr; i < TS; i += RPP) {
            const int r = r0 + i;
            T v[4] = {(T)0, (T)0, (T)0, (T)0};
            if (r < R) {
                const long long row = src_rows ? (long long)src_rows[r] : (long long)r;
                const T* p = src + (size_t)row * ld_src + c0 + cq;
                if (c0 + cq + 3 < C && ((uintptr_t)p & (4 * sizeof(T) - 1)) == 0) {      // one 8- / 16-byte request
                    const Vec4 w = *reinterpret_cast<const Vec4*>(p);
#pragma unroll
                    for (int q = 0; q < 4; ++q) v[q] = w.e[q];
                } else {
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        if (c0 + cq + q < C) v[q] = p[q];
                }
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) tile[i][cq + q] = v[q];
        }
    }
    __syncthreads();
    // store: thread covers 4 consecutive r of one dst row (= source column) per pass
    {
        const int rq = (t % TPR) * 4, cc = t / TPR;
        for (int i = cc; i < TS; i += RPP) {
            const int c = c0 + i;
            if (c >= C) continue;
            T* p = dst + (size_t)c * ld_dst + r0 + rq;
            if (r0 + rq + 3 < R_pad && ((uintptr_t)p & (4 * sizeof(T) - 1)) == 0) {
                Vec4 w;
#pragma unroll
                for (int q = 0; q < 4; ++q) w.e[q] = tile[rq + q][i];
                *reinterpret_cast<Vec4*>(p) = w;
            } else {
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    if (r0 + rq + q < R_pad) p[q] = tile[rq + q][i];
            }
        }
    }
}

// Backward mask of the aux branch on the transposed layout: d[j][b] = 0 where the aux activation a[j][b] was not
// selected (ReLU / TopK mask), and e_b[j] = sum_b d[j][b] (the encoder-bias gradient element of dead feature j), rounded
// once to the model dtype.  One wave per dead feature.
template <int DT>
__global__ __launch_bounds__(256) void aux_mask_rowsum_kernel(int n, int B, const void* __restrict__ a_, long long ld_a,
                                                              void* __restrict__ d_, long long ld_d, void* __restrict__ e_b,
                                                              int vec8) {
    typedef DType<DT> D_;
    typedef typename D_::T T;
    const int lane = threadIdx.x & 63;
    const int j = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
    if (j >= n) return;
    const T* a = reinterpret_cast<const T*>(a_) + (size_t)j * ld_a;
    T* d = reinterpret_cast<T*>(d_) + (size_t)j * ld_d;
    float s = 0.f;
    if (vec8) {                       // rows in 8-element pieces (host: B, both strides and both bases allow it)
        for (int b = lane * 8; b < B; b += 512) {
            float av[8], dv[8];
            load8<DT>(a, (size_t)b, av);
            load8<DT>(d, (size_t)b, dv);
#pragma unroll
            for (int i = 0; i < 8; ++i) { dv[i] = av[i] > 0.f ? dv[i] : 0.f; s += dv[i]; }
            store8<DT>(d, (size_t)b, dv);
        }
    } else {
        for (int b = lane; b < B; b += 64) {
            if (D_::ld(a + b) > 0.f) s += D_::ld(d + b);
            else d[b] = (T)0;
        }
    }
    s = wave_sum(s);
    if (lane == 0) D_::st(reinterpret_cast<T*>(e_b) + j, s);
}

// dst[idx[j]][:] = round(dst[idx[j]][:] + src[j][:]) for j < n: what `grad[dead] += rows` does on a dense gradient
// (unique indices: no atomics).  A = 1 handles the bias vector.
template <int DT>
__global__ __launch_bounds__(256) void rows_scatter_add_kernel(int n, int A, const int32_t* __restrict__ idx, const void* __restrict__ src_,
                                                               long long ld_src, void* __restrict__ dst_, long long ld_dst,
                                                               const int32_t* __restrict__ n_dev, int dst_rows) {
    typedef DType<DT> D_;
    typedef typename D_::T T;
    const int j = blockIdx.x;
    if (j >= n || (n_dev && j >= *n_dev)) return;
    if (idx[j] < 0 || idx[j] >= dst_rows) return;          // never dereference an index outside the destination
    const T* src = reinterpret_cast<const T*>(src_) + (size_t)j * ld_src;
    T* dst = reinterpret_cast<T*>(dst_) + (size_t)idx[j] * ld_dst;
    for (int c = threadIdx.x; c < A; c += 256) D_::st(dst + c, D_::ld(dst + c) + D_::ld(src + c));
}

}  // namespace mtc

using namespace mtc;

extern "C" int mtc_transpose_rows(int32_t dtype, int32_t R, int32_t C, int32_t R_pad, const void* src, int64_t ld_src,
                                  const int32_t* src_rows, void* dst, int64_t ld_dst, void* stream) {
    MTC_REQUIRE(dtype == MTC_F32 || dtype == MTC_BF16, "mtc_transpose_rows: bad dtype %d", dtype);
    MTC_REQUIRE(R >= 0 && C >= 0 && R_pad >= R && ld_src >= C && ld_dst >= R_pad, "mtc_transpose_rows: bad sizes");
    if (C == 0 || R_pad == 0) return 0;
    MTC_REQUIRE(src && dst, "mtc_transpose_rows: null pointer");
    const int ts = dtype == MTC_BF16 ? 128 : 64;
    const dim3 grid((R_pad + ts - 1) / ts, (C + ts - 1) / ts), block(256);
    if (dtype == MTC_BF16)
        hipLaunchKernelGGL(transpose_rows_kernel<u16>, grid, block, 0, as_stream(stream), R, C, R_pad, (const u16*)src,
                           (long long)ld_src, src_rows, (u16*)dst, (long long)ld_dst);
    else
        hipLaunchKernelGGL(transpose_rows_kernel<float>, grid, block, 0, as_stream(stream), R, C, R_pad, (const float*)src,
                           (long long)ld_src, src_rows, (float*)dst, (long long)ld_dst);
    return check_launch("mtc_transpose_rows");
}

extern "C" int mtc_aux_mask_rowsum(int32_t dtype, int32_t n, int32_t B, const void* acts_auxT, int64_t ld_a, void* d_actsT,
                                   int64_t ld_d, void* e_b, void* stream) {
    MTC_REQUIRE(dtype == MTC_F32 || dtype == MTC_BF16, "mtc_aux_mask_rowsum: bad dtype %d", dtype);
    if (n <= 0) return 0;
    MTC_REQUIRE(acts_auxT && d_actsT && e_b && B > 0 && ld_a >= B && ld_d >= B, "mtc_aux_mask_rowsum: bad arguments");
    const dim3 grid((n + 3) / 4), block(256);
    const int vec8 = B % 8 == 0 && ld_a % 8 == 0 && ld_d % 8 == 0 && (((uintptr_t)acts_auxT | (uintptr_t)d_actsT) & 15) == 0;
    if (dtype == MTC_BF16)
        hipLaunchKernelGGL(aux_mask_rowsum_kernel<MTC_BF16>, grid, block, 0, as_stream(stream), n, B, acts_auxT, (long long)ld_a,
                           d_actsT, (long long)ld_d, e_b, vec8);
    else
        hipLaunchKernelGGL(aux_mask_rowsum_kernel<MTC_F32>, grid, block, 0, as_stream(stream), n, B, acts_auxT, (long long)ld_a,
                           d_actsT, (long long)ld_d, e_b, vec8);
    return check_launch("mtc_aux_mask_rowsum");
}

extern "C" int mtc_rows_scatter_add(int32_t dtype, int32_t n, int32_t A, const int32_t* idx, const void* src, int64_t ld_src,
                                    void* dst, int64_t ld_dst, int32_t dst_rows, const int32_t* n_dev, void* stream) {
    MTC_REQUIRE(dtype == MTC_F32 || dtype == MTC_BF16, "mtc_rows_scatter_add: bad dtype %d", dtype);
    if (n <= 0 || A <= 0) return 0;
    MTC_REQUIRE(idx && src && dst && ld_src >= A && ld_dst >= A && dst_rows > 0, "mtc_rows_scatter_add: bad arguments");
    if (dtype == MTC_BF16)
        hipLaunchKernelGGL(rows_scatter_add_kernel<MTC_BF16>, dim3(n), dim3(256), 0, as_stream(stream), n, A, idx, src,
                           (long long)ld_src, dst, (long long)ld_dst, n_dev, dst_rows);
    else
        hipLaunchKernelGGL(rows_scatter_add_kernel<MTC_F32>, dim3(n), dim3(256), 0, as_stream(stream), n, A, idx, src,
                           (long long)ld_src, dst, (long long)ld_dst, n_dev, dst_rows);
    return check_launch("mtc_rows_scatter_add");
}

extern "C" int mtc_aux_dead_list(const mtc_desc* d, const float* cnt, float n_batches_to_dead, int32_t* dead_idx,
                                 int32_t* n_dead_dev, int32_t* row_map, int32_t* next_bound_dev, void* stream) {
    if (check_desc(d, "mtc_aux_dead_list")) return -1;
    MTC_REQUIRE(cnt && dead_idx && n_dead_dev, "mtc_aux_dead_list: null pointer");
    const size_t lds = (size_t)((d->D + 1023) / 1024) * 16 * sizeof(uint32_t);
    MTC_REQUIRE(lds <= 48 * 1024, "mtc_aux_dead_list: D = %d is beyond the 786432 features one launch lists", d->D);
    hipLaunchKernelGGL(dead_list_kernel, dim3(1), dim3(1024), lds, as_stream(stream), d->D, cnt, n_batches_to_dead, dead_idx,
                       n_dead_dev, row_map, next_bound_dev);
    return check_launch("mtc_aux_dead_list");
}

extern "C" int mtc_aux_select(const mtc_desc* d, const void* acts, const int32_t* dead_idx, const int32_t* n_dead_dev,
                              int32_t top_k_aux, uint32_t* row_cnt, uint32_t* row_tau_bits, uint32_t* row_tie_quota,
                              void* stream) {
    if (check_desc(d, "mtc_aux_select")) return -1;
    MTC_REQUIRE(acts && dead_idx && n_dead_dev && row_cnt && row_tau_bits && row_tie_quota, "mtc_aux_select: null pointer");
    MTC_REQUIRE(top_k_aux >= 0, "mtc_aux_select: top_k_aux must be >= 0");
    if (d->dtype == MTC_BF16)
        hipLaunchKernelGGL(aux_select_kernel<MTC_BF16>, dim3(d->B), dim3(256), 0, as_stream(stream), acts, d->D, dead_idx,
                           n_dead_dev, top_k_aux, row_cnt, row_tau_bits, row_tie_quota);
    else
        hipLaunchKernelGGL(aux_select_kernel<MTC_F32>, dim3(d->B), dim3(256), 0, as_stream(stream), acts, d->D, dead_idx,
                           n_dead_dev, top_k_aux, row_cnt, row_tau_bits, row_tie_quota);
    return check_launch("mtc_aux_select");
}

extern "C" int mtc_aux_fill(const mtc_desc* d, const void* acts, const int32_t* dead_idx, const int32_t* n_dead_dev,
                            const uint32_t* row_tau_bits, const uint32_t* row_tie_quota, const int32_t* row_ptr,
                            int32_t* col_idx, void* vals, void* stream) {
    if (check_desc(d, "mtc_aux_fill")) return -1;
    MTC_REQUIRE(acts && dead_idx && n_dead_dev && row_tau_bits && row_tie_quota && row_ptr && col_idx && vals,
                "mtc_aux_fill: null pointer");
    const dim3 grid((d->B + 3) / 4), block(256);
    if (d->dtype == MTC_BF16)
        hipLaunchKernelGGL(aux_fill_kernel<MTC_BF16>, grid, block, 0, as_stream(stream), acts, d->B, d->D, dead_idx, n_dead_dev,
                           row_tau_bits, row_tie_quota, row_ptr, col_idx, vals);
    else
        hipLaunchKernelGGL(aux_fill_kernel<MTC_F32>, grid, block, 0, as_stream(stream), acts, d->B, d->D, dead_idx, n_dead_dev,
                           row_tau_bits, row_tie_quota, row_ptr, col_idx, vals);
    return check_launch("mtc_aux_fill");
}

extern "C" int mtc_aux_fwd(const mtc_desc* d, const int32_t* row_ptr, const int32_t* col_idx, const void* vals,
                           const void* W_dec, const void* x_target, const void* x_hat, const void* std, float aux_penalty,
                           void* g, void* g_aux, float* aux_rows, float* d_aux_vals, void* stream) {
    if (check_desc(d, "mtc_aux_fwd")) return -1;
    MTC_REQUIRE(row_ptr && col_idx && vals && W_dec && x_target && x_hat && aux_rows, "mtc_aux_fwd: null pointer");
    MTC_REQUIRE(!d->unit_norm || std, "mtc_aux_fwd: unit_norm needs std");
    MTC_REQUIRE(!d_aux_vals || g_aux, "mtc_aux_fwd: d_aux_vals needs g_aux");
    const float coef_aux = (float)((double)aux_penalty / ((double)d->B * (double)d->A_t));
    const float inv_n = inv_n_in_dtype(d->dtype, (double)d->B * (double)d->A_t);
    DISPATCH_NC(d->dtype, nc_for(d->A_t), aux_fwd_kernel, dim3((d->B + 3) / 4), dim3(256), as_stream(stream), d->B, d->A_t,
                d->unit_norm, coef_aux, inv_n, row_ptr, col_idx, vals, W_dec, x_target, x_hat, std, g, g_aux, aux_rows, d_aux_vals,
                (const void*)nullptr, (const int32_t*)nullptr);
    return check_launch("mtc_aux_fwd");
}

extern "C" int mtc_aux_dense_fill(const mtc_desc* d, const void* acts, const int32_t* dead_idx, const int32_t* n_dead_dev,
                                  const uint32_t* row_tau_bits, const uint32_t* row_tie_quota, int32_t n_pad,
                                  void* acts_aux, void* stream) {
    if (check_desc(d, "mtc_aux_dense_fill")) return -1;
    MTC_REQUIRE(acts && dead_idx && n_dead_dev && row_tau_bits && row_tie_quota && acts_aux, "mtc_aux_dense_fill: null pointer");
    MTC_REQUIRE(n_pad > 0 && n_pad <= d->D, "mtc_aux_dense_fill: n_pad must be in 1..D");
    const dim3 grid((d->B + 3) / 4), block(256);
    if (d->dtype == MTC_BF16)
        hipLaunchKernelGGL(aux_dense_fill_kernel<MTC_BF16>, grid, block, 0, as_stream(stream), acts, d->B, d->D, n_pad, dead_idx,
                           n_dead_dev, row_tau_bits, row_tie_quota, acts_aux);
    else
        hipLaunchKernelGGL(aux_dense_fill_kernel<MTC_F32>, grid, block, 0, as_stream(stream), acts, d->B, d->D, n_pad, dead_idx,
                           n_dead_dev, row_tau_bits, row_tie_quota, acts_aux);
    return check_launch("mtc_aux_dense_fill");
}

extern "C" int mtc_aux_select_dense(const mtc_desc* d, const void* acts, const int32_t* dead_idx, const int32_t* n_dead_dev,
                                    int32_t top_k_aux, int32_t n_pad, uint32_t* row_cnt, uint32_t* row_tau_bits,
                                    uint32_t* row_tie_quota, void* acts_aux, void* stream) {
    if (check_desc(d, "mtc_aux_select_dense")) return -1;
    MTC_REQUIRE(acts && dead_idx && n_dead_dev && row_cnt && row_tau_bits && row_tie_quota && acts_aux,
                "mtc_aux_select_dense: null pointer");
    MTC_REQUIRE(top_k_aux >= 0, "mtc_aux_select_dense: top_k_aux must be >= 0");
    MTC_REQUIRE(n_pad > 0 && n_pad <= d->D, "mtc_aux_select_dense: n_pad must be in 1..D");
    const size_t es = d->dtype == MTC_BF16 ? 2 : 4;
    const size_t lds = (size_t)n_pad * es;
    // one pass with the row's dead columns held in LDS when they fit beside the histogram (and rows of acts_aux take
    // 8-byte stores); the two-launch form otherwise -- same result either way
    if (lds <= 40 * 1024 && n_pad % (int)(8 / es) == 0 && ((uintptr_t)acts_aux & 7) == 0 && ((uintptr_t)dead_idx & 7) == 0) {
        if (d->dtype == MTC_BF16)
            hipLaunchKernelGGL(aux_select_dense_kernel<MTC_BF16>, dim3(d->B), dim3(256), lds, as_stream(stream), acts, d->D, n_pad,
                               dead_idx, n_dead_dev, top_k_aux, row_cnt, row_tau_bits, row_tie_quota, acts_aux);
        else
            hipLaunchKernelGGL(aux_select_dense_kernel<MTC_F32>, dim3(d->B), dim3(256), lds, as_stream(stream), acts, d->D, n_pad,
                               dead_idx, n_dead_dev, top_k_aux, row_cnt, row_tau_bits, row_tie_quota, acts_aux);
        return check_launch("mtc_aux_select_dense");
    }
    if (mtc_aux_select(d, acts, dead_idx, n_dead_dev, top_k_aux, row_cnt, row_tau_bits, row_tie_quota, stream)) return -1;
    return mtc_aux_dense_fill(d, acts, dead_idx, n_dead_dev, row_tau_bits, row_tie_quota, n_pad, acts_aux, stream);
}

extern "C" int mtc_aux_row_candidates(const mtc_desc* d, const void* acts, const int32_t* dead_idx, const int32_t* n_dead_dev,
                                      const uint32_t* row_tau_bits, const uint32_t* row_tie_quota, int32_t k_pad, void* cand,
                                      void* stream) {
    if (check_desc(d, "mtc_aux_row_candidates")) return -1;
    MTC_REQUIRE(acts && dead_idx && n_dead_dev && row_tau_bits && row_tie_quota && cand, "mtc_aux_row_candidates: null pointer");
    MTC_REQUIRE(k_pad > 0, "mtc_aux_row_candidates: k_pad must be positive");
    const dim3 grid((d->B + 3) / 4), block(256);
    if (d->dtype == MTC_BF16)
        hipLaunchKernelGGL(aux_row_candidates_kernel<MTC_BF16>, grid, block, 0, as_stream(stream), acts, d->B, d->D, k_pad, dead_idx,
                           n_dead_dev, row_tau_bits, row_tie_quota, cand);
    else
        hipLaunchKernelGGL(aux_row_candidates_kernel<MTC_F32>, grid, block, 0, as_stream(stream), acts, d->B, d->D, k_pad, dead_idx,
                           n_dead_dev, row_tau_bits, row_tie_quota, cand);
    return check_launch("mtc_aux_row_candidates");
}

extern "C" int mtc_aux_count_at(const mtc_desc* d, const void* acts, const int32_t* dead_idx, const int32_t* n_dead_dev,
                                const uint32_t* row_tau_bits, int32_t* row_gt, int32_t* row_eq, void* stream) {
    if (check_desc(d, "mtc_aux_count_at")) return -1;
    MTC_REQUIRE(acts && dead_idx && n_dead_dev && row_tau_bits && row_gt && row_eq, "mtc_aux_count_at: null pointer");
    const dim3 grid((d->B + 3) / 4), block(256);
    if (d->dtype == MTC_BF16)
        hipLaunchKernelGGL(aux_count_at_kernel<MTC_BF16>, grid, block, 0, as_stream(stream), acts, d->B, d->D, dead_idx, n_dead_dev,
                           row_tau_bits, row_gt, row_eq);
    else
        hipLaunchKernelGGL(aux_count_at_kernel<MTC_F32>, grid, block, 0, as_stream(stream), acts, d->B, d->D, dead_idx, n_dead_dev,
                           row_tau_bits, row_gt, row_eq);
    return check_launch("mtc_aux_count_at");
}

extern "C" int mtc_aux_count_at_groups(const mtc_desc* d, const void* acts, const int32_t* dead_idx, const int32_t* n_dead_dev,
                                       const uint32_t* row_tau_bits, int32_t* row_gt, int32_t* row_eq_groups, void* stream) {
    if (check_desc(d, "mtc_aux_count_at_groups")) return -1;
    MTC_REQUIRE(acts && dead_idx && n_dead_dev && row_tau_bits && row_gt && row_eq_groups, "mtc_aux_count_at_groups: null pointer");
    AuxGroupBounds gb;
    gb.G = d->G;
    for (int i = 0; i < MTC_MAX_PREFIX; ++i) gb.prefix[i] = i < d->G ? d->prefix[i] : 0x7fffffff;
    const dim3 grid((d->B + 3) / 4), block(256);
    if (d->dtype == MTC_BF16)
        hipLaunchKernelGGL(aux_count_at_groups_kernel<MTC_BF16>, grid, block, 0, as_stream(stream), acts, d->B, d->D, dead_idx,
                           n_dead_dev, row_tau_bits, gb, row_gt, row_eq_groups);
    else
        hipLaunchKernelGGL(aux_count_at_groups_kernel<MTC_F32>, grid, block, 0, as_stream(stream), acts, d->B, d->D, dead_idx,
                           n_dead_dev, row_tau_bits, gb, row_gt, row_eq_groups);
    return check_launch("mtc_aux_count_at_groups");
}

extern "C" int mtc_aux_residual(const mtc_desc* d, const void* x_hat_aux, const void* x_target, const void* x_hat,
                                const void* std, float aux_penalty, void* g, void* g_aux, float* aux_rows,
                                const int32_t* active_dev, void* stream) {
    if (check_desc(d, "mtc_aux_residual")) return -1;
    MTC_REQUIRE(x_hat_aux && x_target && x_hat && aux_rows, "mtc_aux_residual: null pointer");
    MTC_REQUIRE(!d->unit_norm || std, "mtc_aux_residual: unit_norm needs std");
    const float coef_aux = (float)((double)aux_penalty / ((double)d->B * (double)d->A_t));
    const float inv_n = inv_n_in_dtype(d->dtype, (double)d->B * (double)d->A_t);
    DISPATCH_NC(d->dtype, nc_for(d->A_t), aux_fwd_kernel, dim3((d->B + 3) / 4), dim3(256), as_stream(stream), d->B, d->A_t,
                d->unit_norm, coef_aux, inv_n, (const int32_t*)nullptr, (const int32_t*)nullptr, (const void*)nullptr,
                (const void*)nullptr, x_target, x_hat, std, g, g_aux, aux_rows, (float*)nullptr, x_hat_aux, active_dev);
    return check_launch("mtc_aux_residual");
}

extern "C" int mtc_aux_mask(int32_t dtype, uint64_t n, const void* acts_aux, void* d_acts_aux, void* stream) {
    MTC_REQUIRE(dtype == MTC_F32 || dtype == MTC_BF16, "mtc_aux_mask: bad dtype %d", dtype);
    MTC_REQUIRE(acts_aux && d_acts_aux, "mtc_aux_mask: null pointer");
    if (n == 0) return 0;
    const unsigned blocks = (unsigned)((n + 255) / 256 < 8192 ? (n + 255) / 256 : 8192);
    if (dtype == MTC_BF16)
        hipLaunchKernelGGL(aux_mask_kernel<MTC_BF16>, dim3(blocks), dim3(256), 0, as_stream(stream), (size_t)n, acts_aux, d_acts_aux);
    else
        hipLaunchKernelGGL(aux_mask_kernel<MTC_F32>, dim3(blocks), dim3(256), 0, as_stream(stream), (size_t)n, acts_aux, d_acts_aux);
    return check_launch("mtc_aux_mask");
}
