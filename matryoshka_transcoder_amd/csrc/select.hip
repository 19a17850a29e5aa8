// K2: global batch-TopK (sae.py:176-187) as an exact k-th-largest threshold select, the eval-mode
// threshold select (sae.py:189-191) and the threshold EMA (sae.py:349-361).
//
// ReLU outputs are >= 0, and the float bit pattern of a positive float is order preserving, so the
// k-th largest value is found by a most-significant-bits-first radix select over the bit pattern:
//   level 0: bits [30:19] (12 bits)   level 1: bits [18:7]   level 2: bits [6:0]   (bf16: 2 levels)
// Each level is one streaming pass that builds a 4096-bin histogram (LDS atomics per workgroup,
// then global atomics for the non-empty bins) followed by a one-workgroup suffix scan.  The
// histograms are additive, so a feature-sharded caller all-reduces them between the two kernels.
// With tau known, one wavefront per batch row counts (> tau, == tau), a one-workgroup scan turns
// the counts into CSR offsets and a per-row tie quota (ties are taken in ascending flat-index
// order until k are selected), and one wavefront per row compacts its selected entries in
// ascending column order (= prefix-group order).  Everything stays on the device.
#include "common.hpp"
#include "rowops.hpp"

namespace mtc {

constexpr int HIST_BINS = 4096;

// min over all selected values, one candidate per wave.  4096 same-address atomics serialise in L2
// (~40 us), so a wave first looks at the current value: it only ever decreases, hence a stale read
// can cost a redundant atomic but never skips a needed one.
__device__ __forceinline__ void note_min_selected(mtc_select_state* st, uint32_t mn) {
    if (mn == 0xffffffffu) return;
    const uint32_t cur = __hip_atomic_load(&st->min_sel_bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (mn < cur) atomicMin(&st->min_sel_bits, mn);
}

__device__ __forceinline__ int level_shift(int level) { return level == 0 ? 19 : (level == 1 ? 7 : 0); }
__device__ __forceinline__ uint32_t level_mask(int level) { return level == 2 ? 0x7fu : 0xfffu; }
// bits above the level's field (must match the prefix fixed so far)
__device__ __forceinline__ uint32_t level_prefix_mask(int level) {
    return level == 0 ? 0u : (level == 1 ? 0xfff80000u : 0xffffff80u);
}

__global__ void select_init_kernel(mtc_select_state* st, uint32_t k_total) {
    if (threadIdx.x == 0) {
        st->key_prefix = 0; st->k_rem = k_total; st->all_pos = 0; st->n_gt = 0;
        st->tau_bits = 0; st->n_tie_take = 0; st->min_sel_bits = 0xffffffffu; st->nnz = 0;
        for (int i = 0; i < 8; ++i) st->pad[i] = 0;
    }
}

template <int DT>
__global__ __launch_bounds__(256) void select_hist_kernel(const void* __restrict__ acts_in, size_t n_chunks, int level,
                                                          const mtc_select_state* __restrict__ st,
                                                          uint32_t* __restrict__ hist,
                                                          const uint32_t* __restrict__ pred, uint32_t want,
                                                          const void* __restrict__ alt_acts, size_t alt_chunks,
                                                          uint32_t grid_main, uint32_t grid_alt) {
    __shared__ uint32_t lh[HIST_BINS];
    // pred == null: plain pass.  pred, no alt: runs only if *pred == want.  pred + alt (ONE launch for the compact / full
    // pair of the carried-bound select): *pred == want streams `acts_in`, otherwise `alt_acts`; the grid is sized for the
    // larger of the two and the surplus workgroups leave at once
    const void* acts = acts_in;
    uint32_t nblk = gridDim.x;
    if (pred) {
        const bool m = *pred == want;                  // uniform
        if (alt_acts) {
            if (!m) { acts = alt_acts; n_chunks = alt_chunks; }
            nblk = m ? grid_main : grid_alt;
            if (blockIdx.x >= nblk) return;
        } else if (!m) return;
    }
    for (int i = threadIdx.x; i < HIST_BINS; i += 256) lh[i] = 0;
    __syncthreads();
    const uint32_t all_pos = st->all_pos;
    const uint32_t prefix = st->key_prefix;
    const int sh = level_shift(level);
    const uint32_t fm = level_mask(level), pm = level_prefix_mask(level);
    if (!(all_pos && level > 0)) {
        // 4 independent 16-byte loads per lane per iteration (memory-level parallelism)
        const size_t stride = (size_t)nblk * 256;
        for (size_t c0 = (size_t)blockIdx.x * 256 + threadIdx.x; c0 < n_chunks; c0 += 4 * stride) {
            Raw<DT> r[4];
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (c0 + u * stride < n_chunks) r[u] = load_raw<DT>(acts, (c0 + u * stride) * 8);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (c0 + u * stride >= n_chunks) break;
                float v[8];
                unpack(r[u], v);
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const uint32_t k = __float_as_uint(v[i]);
                    // positive, finite-or-inf, not NaN, and (levels > 0) inside the bin chosen so far
                    if (k != 0 && k <= 0x7f800000u && ((k ^ prefix) & pm) == 0) atomicAdd(&lh[(k >> sh) & fm], 1u);
                }
            }
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < HIST_BINS; i += 256) {
        const uint32_t v = lh[i];
        if (v) atomicAdd(&hist[i], v);
    }
}

// one workgroup of 1024 threads: suffix scan over 4096 bins, pick the bin holding the k_rem-th largest
__global__ __launch_bounds__(1024) void select_scan_kernel(const uint32_t* __restrict__ hist, int level, int last_level,
                                                           mtc_select_state* st, const uint32_t* __restrict__ pred = nullptr,
                                                           uint32_t want = 0) {
    __shared__ uint32_t part[1024];
    __shared__ uint32_t found_bin, found_above;
    const int t = threadIdx.x;
    if (pred && *pred != want) return;                 // uniform (carried-bound select: the window tail served this call)
    if (st->all_pos) return;                           // uniform: whole block leaves
    // thread t owns bins [4t, 4t+3]; "above" = higher bin index
    const uint32_t h0 = hist[4 * t], h1 = hist[4 * t + 1], h2 = hist[4 * t + 2], h3 = hist[4 * t + 3];
    part[t] = h0 + h1 + h2 + h3;
    if (t == 0) { found_bin = 0xffffffffu; found_above = 0; }
    __syncthreads();
    // inclusive suffix sum over threads (Hillis-Steele on 1024 entries)
    for (int off = 1; off < 1024; off <<= 1) {
        const uint32_t add = (t + off < 1024) ? part[t + off] : 0u;
        __syncthreads();
        part[t] += add;
        __syncthreads();
    }
    const uint32_t k_rem = st->k_rem;
    const uint32_t above_thread = (t + 1 < 1024) ? part[t + 1] : 0u;   // count in bins above this thread's 4
    // walk my 4 bins from the top
    uint32_t above = above_thread;
    const uint32_t hs[4] = {h0, h1, h2, h3};
#pragma unroll
    for (int i = 3; i >= 0; --i) {
        if (above < k_rem && k_rem <= above + hs[i]) { found_bin = 4 * t + i; found_above = above; }
        above += hs[i];
    }
    __syncthreads();
    if (t == 0) {
        const uint32_t total = part[0];
        if (found_bin == 0xffffffffu) {
            // fewer than k_rem candidates: only possible at level 0 (fewer positives than k)
            st->all_pos = 1; st->n_gt = total; st->tau_bits = 0; st->n_tie_take = 0; st->k_rem = 0;
        } else {
            st->n_gt += found_above;
            st->k_rem = k_rem - found_above;
            st->key_prefix |= found_bin << level_shift(level);
            if (level == last_level) { st->tau_bits = st->key_prefix; st->n_tie_take = st->k_rem; }
        }
    }
}

// wave per row: counts of (> tau) and (== tau); threshold mode: (> thr) only
template <int DT>
__global__ __launch_bounds__(256) void select_count_kernel(const void* __restrict__ acts_in, int B, int D,
                                                           mtc_select_state* __restrict__ st,
                                                           const void* __restrict__ threshold,
                                                           uint32_t* __restrict__ row_gt, uint32_t* __restrict__ row_eq,
                                                           const uint32_t* __restrict__ pred, uint32_t want,
                                                           const void* __restrict__ alt_acts, int alt_D) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= B) return;
    // pred + alt: ONE launch serves the compact / full pair (*pred == want streams `acts_in`, otherwise the alt matrix)
    bool use_alt = false;
    if (pred && *pred != want) {                        // uniform
        if (!alt_acts) return;
        use_alt = true;
    }
    const void* __restrict__ acts = use_alt ? alt_acts : acts_in;
    if (use_alt) { D = alt_D; }
    float tau; bool ties = false, nz = false;
    if (threshold) {
        tau = DType<DT>::ld(reinterpret_cast<const typename DType<DT>::T*>(threshold));
        nz = tau != tau;                      // NaN threshold = "every non-zero entry" (dense -> CSR)
        tau = nz ? 0.f : fmaxf(tau, 0.f);
    } else if (st->all_pos) tau = 0.f;
    else { tau = __uint_as_float(st->tau_bits); ties = true; }
    const size_t base = (size_t)row * D;
    uint32_t gt = 0, eq = 0, mn = 0xffffffffu;
    const int nch = D >> 3;
    for (int c0 = lane; c0 < nch; c0 += 256) {
        Raw<DT> r[4];
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (c0 + 64 * u < nch) r[u] = load_raw<DT>(acts, base + (size_t)(c0 + 64 * u) * 8);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (c0 + 64 * u >= nch) break;
            float v[8];
            unpack(r[u], v);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (nz ? (v[i] != 0.f) : (v[i] > tau)) { ++gt; mn = min(mn, __float_as_uint(v[i])); }
                else if (ties && v[i] == tau) ++eq;
            }
        }
    }
    gt = wave_sum_u32(gt); eq = wave_sum_u32(eq);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mn = min(mn, (uint32_t)__shfl_xor((int)mn, o, 64));
    if (lane == 0) {
        row_gt[row] = gt; row_eq[row] = eq;
        // with a tau the smallest selected value IS tau (the k-th largest entry is selected; the offsets kernel notes it):
        // 4096 waves arriving together would all pass the pruning load and queue up on one address (~20 us)
        if (!ties) note_min_selected(st, mn);
    }
}

// bf16 fast path: after level 0 only 3 key bits remain, so one pass per row can count, for every
// row, the entries above the level-0 bin and the 8 sub-bin populations inside it; tau, the tie
// quota and the per-row (> tau, == tau) counts then follow without another pass over acts.
__global__ __launch_bounds__(256) void select_rowsub_bf16_kernel(const void* __restrict__ acts_in, int B, int D,
                                                                 mtc_select_state* __restrict__ st,
                                                                 uint32_t* __restrict__ row_above,
                                                                 uint32_t* __restrict__ row_sub /* [B][8] */,
                                                                 const uint32_t* __restrict__ pred, uint32_t want,
                                                                 const void* __restrict__ alt_acts, int alt_D) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= B) return;
    // pred + alt: ONE launch serves the compact / full pair (*pred == want streams `acts_in`, otherwise the alt matrix)
    bool use_alt = false;
    if (pred && *pred != want) {                        // uniform
        if (!alt_acts) return;
        use_alt = true;
    }
    const void* __restrict__ acts = use_alt ? alt_acts : acts_in;
    if (use_alt) { D = alt_D; }
    const bool all_pos = st->all_pos != 0;
    const uint32_t bin = st->key_prefix >> 19;            // level-0 bin of tau
    const size_t base = (size_t)row * D;
    const int nch = D >> 3;
    uint32_t above = 0, mn = 0xffffffffu;
    uint32_t sub[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int c0 = lane; c0 < nch; c0 += 512) {
        Raw<MTC_BF16> r[8];
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (c0 + 64 * u < nch) r[u] = load_raw<MTC_BF16>(acts, base + (size_t)(c0 + 64 * u) * 8);
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (c0 + 64 * u >= nch) break;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const uint32_t k = h ? (r[u].v[i] & 0xffff0000u) : (r[u].v[i] << 16);
                    if (k == 0 || k > 0x7f800000u) continue;
                    const uint32_t kb = k >> 19;
                    if (all_pos || kb > bin) { ++above; mn = min(mn, k); }
                    else if (kb == bin) {
                        const uint32_t sidx = (k >> 16) & 7u;
#pragma unroll
                        for (int q = 0; q < 8; ++q) sub[q] += (sidx == (uint32_t)q) ? 1u : 0u;
                    }
                }
            }
        }
    }
    above = wave_sum_u32(above);
#pragma unroll
    for (int q = 0; q < 8; ++q) sub[q] = wave_sum_u32(sub[q]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mn = min(mn, (uint32_t)__shfl_xor((int)mn, o, 64));
    if (lane == 0) {
        row_above[row] = above;
#pragma unroll
        for (int q = 0; q < 8; ++q) row_sub[(size_t)row * 8 + q] = sub[q];
        if (all_pos) note_min_selected(st, mn);         // otherwise min selected = tau, noted by the offsets kernel (see select_count_kernel)
    }
}

// one workgroup: global sub-bin totals -> tau / quota -> per-row (> tau, == tau)
__device__ __forceinline__ void select_finish_body(int B, mtc_select_state* st, const uint32_t* __restrict__ row_above,
                                                   const uint32_t* __restrict__ row_sub,
                                                   const uint32_t* __restrict__ global_sub /* [8] or null */,
                                                   uint32_t* __restrict__ row_gt, uint32_t* __restrict__ row_eq) {
    __shared__ uint32_t sm[16][8];
    __shared__ uint32_t s_star;
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const bool all_pos = st->all_pos != 0;
    uint32_t tot[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int r = t; r < B; r += 1024)
#pragma unroll
        for (int q = 0; q < 8; ++q) tot[q] += row_sub[(size_t)r * 8 + q];
#pragma unroll
    for (int q = 0; q < 8; ++q) { tot[q] = wave_sum_u32(tot[q]); if (lane == 0) sm[w][q] = tot[q]; }
    __syncthreads();
    if (t == 0) {
        uint32_t star = 0;
        if (!all_pos) {
            const uint32_t k_rem = st->k_rem;
            uint32_t above = 0;
            for (int q = 7; q >= 0; --q) {
                uint32_t h = 0;
                if (global_sub) h = global_sub[q];
                else for (int i = 0; i < 16; ++i) h += sm[i][q];
                if (above < k_rem && k_rem <= above + h) {
                    star = (uint32_t)q;
                    st->n_gt += above;
                    st->k_rem = k_rem - above;
                    st->key_prefix |= (uint32_t)q << 16;
                    st->tau_bits = st->key_prefix;
                    st->n_tie_take = k_rem - above;
                    break;
                }
                above += h;
            }
        }
        s_star = star;
    }
    __syncthreads();
    const uint32_t star = s_star;
    for (int r = t; r < B; r += 1024) {
        uint32_t gt = row_above[r], eq = 0;
        if (!all_pos) {
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const uint32_t c = row_sub[(size_t)r * 8 + q];
                if ((uint32_t)q > star) gt += c;
                else if ((uint32_t)q == star) eq = c;
            }
        }
        row_gt[r] = gt; row_eq[r] = eq;
    }
}
__global__ __launch_bounds__(1024) void select_finish_bf16_kernel(int B, mtc_select_state* st,
                                                                  const uint32_t* __restrict__ row_above,
                                                                  const uint32_t* __restrict__ row_sub,
                                                                  const uint32_t* __restrict__ global_sub,
                                                                  uint32_t* __restrict__ row_gt, uint32_t* __restrict__ row_eq) {
    select_finish_body(B, st, row_above, row_sub, global_sub, row_gt, row_eq);
}

// block-wide exclusive scan helper (1024 threads); returns exclusive prefix of v, total in *total
__device__ __forceinline__ uint32_t block_excl_scan_1024(uint32_t v, uint32_t* sm /*[16]*/, uint32_t* total) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    uint32_t inc = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t n = __shfl_up(inc, o, 64);
        if (lane >= o) inc += n;
    }
    if (lane == 63) sm[w] = inc;
    __syncthreads();
    uint32_t wbase = 0, tot = 0;
    for (int i = 0; i < 16; ++i) { const uint32_t s = sm[i]; if (i < w) wbase += s; tot += s; }
    __syncthreads();
    *total = tot;
    return wbase + inc - v;
}

__device__ __forceinline__ void select_offsets_body(int B, const uint32_t* __restrict__ row_gt,
                                                    const uint32_t* __restrict__ row_eq,
                                                    const uint32_t* __restrict__ tie_before_in, mtc_select_state* st,
                                                    int32_t* __restrict__ row_ptr, uint32_t* __restrict__ tie_take,
                                                    uint32_t* carry) {
    __shared__ uint32_t sm[16];
    const int t = threadIdx.x;
    const int per = (B + 1023) / 1024;
    const int r0 = t * per, r1 = min(B, r0 + per);
    const uint32_t quota = st->n_tie_take;
    // pass 1: ties preceding each row
    uint32_t my_eq = 0;
    for (int r = r0; r < r1; ++r) my_eq += row_eq[r];
    uint32_t tot;
    uint32_t eq_base = block_excl_scan_1024(my_eq, sm, &tot);
    // pass 2: quota per row, nnz per row
    uint32_t my_nnz = 0;
    uint32_t run = eq_base;
    for (int r = r0; r < r1; ++r) {
        const uint32_t e = row_eq[r];
        const uint32_t before = tie_before_in ? tie_before_in[r] : run;
        const uint32_t take = before >= quota ? 0u : min(e, quota - before);
        tie_take[r] = take;
        my_nnz += row_gt[r] + take;
        run += e;
    }
    uint32_t nnz_total;
    uint32_t base = block_excl_scan_1024(my_nnz, sm, &nnz_total);
    for (int r = r0; r < r1; ++r) {
        row_ptr[r] = (int32_t)base;
        base += row_gt[r] + tie_take[r];
    }
    if (t == 0) {
        row_ptr[B] = (int32_t)nnz_total;
        st->nnz = nnz_total;
        if (quota > 0) st->min_sel_bits = min(st->min_sel_bits, st->tau_bits);
        if (carry) {
            // next step's lower bound = tau lowered by 8 bf16 ulps (about 6 %); none if tau is undefined
            const uint32_t margin = 8u << 16;
            uint32_t g = 0;
            if (!st->all_pos && nnz_total != 0 && st->tau_bits > margin + 0x00800000u && st->tau_bits <= 0x7f800000u)
                g = st->tau_bits - margin;
            carry[0] = g;
        }
    }
}
__global__ __launch_bounds__(1024) void select_offsets_kernel(int B, const uint32_t* __restrict__ row_gt,
                                                              const uint32_t* __restrict__ row_eq,
                                                              const uint32_t* __restrict__ tie_before_in,
                                                              mtc_select_state* st, int32_t* __restrict__ row_ptr,
                                                              uint32_t* __restrict__ tie_take, uint32_t* carry) {
    select_offsets_body(B, row_gt, row_eq, tie_before_in, st, row_ptr, tie_take, carry);
}

// wave per row: ordered compaction of the selected entries
template <int DT>
__global__ __launch_bounds__(256) void select_fill_kernel(const void* __restrict__ acts_in, int B, int D,
                                                          const mtc_select_state* __restrict__ st,
                                                          const void* __restrict__ threshold,
                                                          const int32_t* __restrict__ row_ptr,
                                                          const uint32_t* __restrict__ tie_take,
                                                          int32_t* __restrict__ col_idx, void* __restrict__ vals,
                                                          const int32_t* __restrict__ col_map_in /* [B][D] or null */,
                                                          const uint32_t* __restrict__ pred, uint32_t want,
                                                          const void* __restrict__ alt_acts, int alt_D) {
    typedef typename DType<DT>::T T;
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= B) return;
    const int32_t* col_map = col_map_in;
    // pred + alt: ONE launch serves the compact / full pair (*pred == want streams `acts_in`, otherwise the alt matrix)
    bool use_alt = false;
    if (pred && *pred != want) {                        // uniform
        if (!alt_acts) return;
        use_alt = true;
    }
    const void* __restrict__ acts = use_alt ? alt_acts : acts_in;
    if (use_alt) { D = alt_D; col_map = nullptr; }     // the alt (full) matrix is in natural column order
    float tau; uint32_t quota = 0; bool nz = false;
    if (threshold) {
        tau = DType<DT>::ld(reinterpret_cast<const T*>(threshold));
        nz = tau != tau;
        tau = nz ? 0.f : fmaxf(tau, 0.f);
    } else if (st->all_pos) tau = 0.f;
    else { tau = __uint_as_float(st->tau_bits); quota = tie_take[row]; }
    const size_t base = (size_t)row * D;
    uint32_t out = (uint32_t)row_ptr[row];
    uint32_t ties_seen = 0;                             // wave-uniform running count of == tau in this row
    T* vout = reinterpret_cast<T*>(vals);
    const int nch = D >> 3;
    for (int cb = 0; cb < nch; cb += 256) {
        Raw<DT> r[4];
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (cb + 64 * u + lane < nch) r[u] = load_raw<DT>(acts, base + (size_t)(cb + 64 * u + lane) * 8);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int c = cb + 64 * u + lane;
            if (cb + 64 * u >= nch) break;                   // wave-uniform
            float v[8];
            uint32_t gtm = 0, eqm = 0;
            if (c < nch) {
                unpack(r[u], v);
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    if (nz ? (v[i] != 0.f) : (v[i] > tau)) gtm |= 1u << i;
                    else if (quota && v[i] == tau) eqm |= 1u << i;
                }
            }
            const uint64_t any = __ballot((gtm | eqm) != 0);
            if (any == 0) continue;                          // wave-uniform
            uint32_t selm = gtm;
            if (quota && __ballot(eqm != 0)) {
                // tie rank = ties in earlier lanes + ties seen in earlier blocks
                const uint32_t ne = __popc(eqm);
                uint32_t inc = ne;
#pragma unroll
                for (int o = 1; o < 64; o <<= 1) { const uint32_t n = __shfl_up(inc, o, 64); if (lane >= o) inc += n; }
                uint32_t rank = ties_seen + inc - ne;
#pragma unroll
                for (int i = 0; i < 8; ++i)
                    if (eqm & (1u << i)) { if (rank < quota) selm |= 1u << i; ++rank; }
                ties_seen += __shfl(inc, 63, 64);
            }
            const uint32_t ns = __popc(selm);
            uint32_t inc = ns;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) { const uint32_t n = __shfl_up(inc, o, 64); if (lane >= o) inc += n; }
            uint32_t pos = out + inc - ns;
#pragma unroll
            for (int i = 0; i < 8; ++i)
                if (selm & (1u << i)) {
                    col_idx[pos] = col_map ? col_map[base + (size_t)c * 8 + i] : c * 8 + i;
                    DType<DT>::st(vout + pos, v[i]);
                    ++pos;
                }
            out += __shfl(inc, 63, 64);
        }
    }
}

// ---------------------------------------------------------------- carried lower bound (training loop)
// tau moves slowly from one training step to the next.  With a lower bound `guess` (the previous
// step's tau minus a margin) ONE pass over acts copies every entry >= guess into a compact
// [B][C] matrix (+ its column map), in row / ascending-column order; the exact select above then runs
// on that matrix (a few MB) instead of three more passes over B*D activations.  The result is
// identical: every entry >= tau is a candidate and order is preserved, so tau, the tie quota and
// the tie order come out the same.  If the bound was wrong (fewer than k candidates, or a row with
// more than C) the flag stays 0 and the full-size passes run instead -- both sets of launches are
// enqueued, each predicated on the device-side flag, so nothing is read back.
//   carry[0] guess bits (0 = none)  [1] flag: 1 = compact path valid  [2] candidates  [3] overflow
//   carry[4] calls served by the compact path  [5] calls served by the full path
__device__ __forceinline__ uint32_t mbcnt64(uint64_t m) {
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
}

// cheap superset test: does this 8-element chunk hold ANY bit pattern >= the bound?  (NaNs and negatives, which cannot
// come out of the encoder's ReLU anyway, also say yes; the exact test below rejects them.)
typedef __attribute__((ext_vector_type(2))) unsigned short us2;
template <int DT> __device__ __forceinline__ bool chunk_may_hold(const Raw<DT>& r, uint32_t guess);
template <> __device__ __forceinline__ bool chunk_may_hold<MTC_BF16>(const Raw<MTC_BF16>& r, uint32_t guess) {
    const uint32_t x0 = r.v[0], x1 = r.v[1], x2 = r.v[2], x3 = r.v[3];
    const us2 a = __builtin_elementwise_max(__builtin_bit_cast(us2, x0), __builtin_bit_cast(us2, x1));
    const us2 b = __builtin_elementwise_max(__builtin_bit_cast(us2, x2), __builtin_bit_cast(us2, x3));
    const us2 c = __builtin_elementwise_max(a, b);
    const uint32_t mx = max((uint32_t)c.x, (uint32_t)c.y);
    return mx >= (guess >> 16);
}
template <> __device__ __forceinline__ bool chunk_may_hold<MTC_F32>(const Raw<MTC_F32>& r, uint32_t guess) {
    uint32_t mx = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float fa = r.a[i], fb = r.b[i];
        mx = max(mx, max(__float_as_uint(fa), __float_as_uint(fb)));
    }
    return mx >= guess;
}

// Wave per row.  The row streams through registers (U chunks of 8 elements per lane in flight, the next block
// requested before this one is looked at); a chunk that MAY hold a candidate (one packed-max test, ~0.5 % of the
// elements are candidates) is appended, in column order, to a per-wave queue in LDS.  Whenever the queue holds more
// than a wave's worth, 64 queued chunks are examined exactly, one per lane, and their candidates are written out in
// order.  The exact test / ordered compaction therefore runs once per ~25 loads instead of once per load.
constexpr int CAND_QCAP = 128;
constexpr int CAND_WIN = 32;            // window of the bf16 fast path: bins 0 .. 30 = ulps above the bound, bin 31 = 31 or more
// marks (optional; written by the encoder's epilogue, see encode_bf16_mfma256p_kernel<.., MARK>): uint8 [B][ceil(D / 128)],
// bit i of a byte = "features [16 i, 16 i + 16) of that 128-feature block may hold an entry >= the bound", followed by
// the bound they were made for.  When that bound is the one carried now, only the flagged 32-byte granules are read
// (the queue then holds chunk numbers and the data is fetched when a batch of 64 is examined); otherwise -- no marks,
// another bound -- the row is streamed.  Same candidates, same order either way.
template <int DT>
__global__ __launch_bounds__(256) void select_candidates_kernel(const void* __restrict__ acts, int B, int D, int C,
                                                                const uint32_t* __restrict__ carry, void* __restrict__ cand_vals,
                                                                int32_t* __restrict__ cand_cols, uint32_t* __restrict__ cand_cnt,
                                                                const unsigned char* __restrict__ marks,
                                                                unsigned short* __restrict__ row_win = nullptr) {
    typedef typename DType<DT>::T T;
    constexpr int U = DT == MTC_BF16 ? 8 : 4;
    __shared__ Raw<DT> q_raw[4][CAND_QCAP];
    __shared__ int q_chunk[4][CAND_QCAP];
    __shared__ uint32_t s_win[4][CAND_WIN];
    const int lane = threadIdx.x & 63;
    const int w = threadIdx.x >> 6;
    const int row = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + w);
    if (row >= B) return;
    const uint32_t guess = carry[0];
    if (guess == 0) return;
    // bf16: the row's candidates counted by distance from the bound in bf16 ulps (see select_window_tail_kernel); the
    // counters are private to the wave, whose LDS operations complete in order
    const bool win = DT == MTC_BF16 && row_win != nullptr;
    if (win && lane < CAND_WIN) s_win[w][lane] = 0;
    const size_t base = (size_t)row * D;
    T* vout = reinterpret_cast<T*>(cand_vals) + (size_t)row * C;
    int32_t* cout = cand_cols + (size_t)row * C;
    const int nch = D >> 3;
    uint32_t out = 0;
    int qn = 0;                                          // queued chunks (wave-uniform)
    const int nblk = (D + 127) >> 7;
    const bool marked = marks != nullptr &&
                        *reinterpret_cast<const uint32_t*>(marks + ((((size_t)B * nblk) + 3) & ~(size_t)3)) == guess;   // uniform

    // exact pass over queued chunks [0, qn): one chunk per lane, candidates leave in (chunk, element) order
    auto drain = [&]() {
        for (int q0 = 0; q0 < qn; q0 += 64) {
            const int e = q0 + lane;
            float v[8];
            uint32_t m = 0;
            int c = 0;
            if (e < qn) c = q_chunk[w][e];
            if (e < qn && c >= 0) {
                const Raw<DT> r = marked ? load_raw_nt<DT>(acts, base + (size_t)c * 8) : q_raw[w][e];
                unpack(r, v);
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const uint32_t k = __float_as_uint(v[i]);
                    if (k >= guess && k <= 0x7f800000u) m |= 1u << i;
                }
            }
            if (__ballot(m != 0) == 0) continue;         // wave-uniform
            // exclusive prefix of popc(m) over the lanes from ballots of its bit planes
            const uint32_t ns = __popc(m);
            const uint64_t b0 = __ballot((ns & 1u) != 0), b1 = __ballot((ns & 2u) != 0);
            const uint64_t b2 = __ballot((ns & 4u) != 0), b3 = __ballot((ns & 8u) != 0);
            uint32_t pos = out + mbcnt64(b0);
            uint32_t tot = (uint32_t)__popcll(b0);
            if (b1) { pos += 2u * mbcnt64(b1); tot += 2u * (uint32_t)__popcll(b1); }
            if (b2) { pos += 4u * mbcnt64(b2); tot += 4u * (uint32_t)__popcll(b2); }
            if (b3) { pos += 8u * mbcnt64(b3); tot += 8u * (uint32_t)__popcll(b3); }
            uint32_t mm = m;
            while (__ballot(mm != 0)) {
                if (mm) {
                    const int i = __ffs(mm) - 1;
                    float val = v[0];
#pragma unroll
                    for (int q = 1; q < 8; ++q) val = i == q ? v[q] : val;
                    if (pos < (uint32_t)C) { DType<DT>::st(vout + pos, val); cout[pos] = c * 8 + i; }
                    if (win) atomicAdd(&s_win[w][min((__float_as_uint(val) - guess) >> 16, (uint32_t)CAND_WIN - 1u)], 1u);
                    ++pos;
                    mm &= mm - 1;
                }
            }
            out += tot;
        }
        qn = 0;
    };

    if (marked) {
        // one mark byte per lane and round (128 features = 16 chunks); flagged granules queue up as two chunk numbers
        // each, in (block, bit) order = column order
        const unsigned char* mrow = marks + (size_t)row * nblk;
        for (int b0 = 0; b0 < nblk; b0 += 64) {
            const int blk = b0 + lane;
            uint32_t byte = blk < nblk ? (uint32_t)mrow[blk] : 0u;
            const uint32_t n2 = 2u * __popc(byte);                         // chunks this lane queues
            // exclusive prefix over the lanes (n2 <= 16: five bit planes)
            uint32_t at = 0, tot = 0;
#pragma unroll
            for (int p = 1; p < 5; ++p) {
                const uint64_t bp = __ballot((n2 >> p) & 1u);
                at += (mbcnt64(bp) << p);
                tot += ((uint32_t)__popcll(bp) << p);
            }
            if (tot == 0) continue;                                        // wave-uniform
            // a round can queue up to 1024 chunks: hand them over in slices the queue has room for
            uint32_t done = 0;                                             // chunks of this round already queued (uniform)
            while (done < tot) {
                const uint32_t room = (uint32_t)(CAND_QCAP - qn);
                const uint32_t upto = min(tot, done + room);               // queue chunks [done, upto) of the round
                uint32_t k = at;                                           // running index of this lane's chunks
                uint32_t bits = byte;
                while (bits) {
                    const int bit = __ffs(bits) - 1;
                    bits &= bits - 1;
#pragma unroll
                    for (int h = 0; h < 2; ++h, ++k)
                        if (k >= done && k < upto) {
                            const int c = blk * 16 + bit * 2 + h;
                            q_chunk[w][qn + (int)(k - done)] = c < nch ? c : -1;      // (a granule's second half may lie beyond D)
                        }
                }
                qn += (int)(upto - done);
                done = upto;
                if (qn > CAND_QCAP - 64 || done < tot) drain();
            }
        }
        drain();
    } else {
    Raw<DT> nxt[U];
#pragma unroll
    for (int u = 0; u < U; ++u)
        if (64 * u + lane < nch) nxt[u] = load_raw_nt<DT>(acts, base + (size_t)(64 * u + lane) * 8);
    for (int cb = 0; cb < nch; cb += 64 * U) {
        Raw<DT> cur[U];
#pragma unroll
        for (int u = 0; u < U; ++u) cur[u] = nxt[u];
        const int nb = cb + 64 * U;                    // next block is requested before this one is scanned
#pragma unroll
        for (int u = 0; u < U; ++u)
            if (nb + 64 * u + lane < nch) nxt[u] = load_raw_nt<DT>(acts, base + (size_t)(nb + 64 * u + lane) * 8);
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (cb + 64 * u >= nch) break;                   // wave-uniform
            const int c = cb + 64 * u + lane;
            const bool may = c < nch && chunk_may_hold<DT>(cur[u], guess);
            const uint64_t bal = __ballot(may);
            if (bal == 0) continue;                          // wave-uniform
            if (may) {
                const int e = qn + (int)mbcnt64(bal);
                q_raw[w][e] = cur[u];
                q_chunk[w][e] = c;
            }
            qn += (int)__popcll(bal);
            if (qn > CAND_QCAP - 64) drain();
        }
    }
    drain();
    }
    // zero the unused tail of the compact row (the select passes treat 0 as "not a candidate")
    const uint32_t n_kept = min(out, (uint32_t)C);
    const uint32_t al = (n_kept + 7u) & ~7u;
    if (n_kept + lane < al) DType<DT>::st(vout + n_kept + lane, 0.f);
    const float z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (uint32_t ch = (al >> 3) + lane; ch < (uint32_t)(C >> 3); ch += 64) store8<DT>(vout, (size_t)ch * 8, z);
    if (lane == 0) cand_cnt[row] = out;
    if (win) {
        // INCLUSIVE prefix over the bins: row_win[row][b] = candidates of the row at most b ulps above the bound (bin 31:
        // all of them), so the tail needs two entries per row once it knows tau's bin
        uint32_t p = lane < CAND_WIN ? s_win[w][lane] : 0u;
#pragma unroll
        for (int o = 1; o < CAND_WIN; o <<= 1) { const uint32_t n = __shfl_up(p, o, 64); if (lane >= o) p += n; }
        if (lane < CAND_WIN) row_win[(size_t)row * CAND_WIN + lane] = (unsigned short)min(p, 65535u);
    }
}

// one workgroup, after the candidate pass: total / overflow from the per-row counts -> which path is valid
// (also what mtc_select_init and the histogram memset do, to save two launches)
__global__ __launch_bounds__(1024) void select_carry_decide_kernel(uint32_t* carry, const uint32_t* __restrict__ cand_cnt, int B, int C,
                                                                   uint32_t k_total, mtc_select_state* st,
                                                                   uint32_t* __restrict__ hist) {
    __shared__ uint32_t s_tot[16], s_over[16];
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    for (int i = t; i < 3 * HIST_BINS; i += 1024) hist[i] = 0;
    if (t == 0) {
        st->key_prefix = 0; st->k_rem = k_total; st->all_pos = 0; st->n_gt = 0;
        st->tau_bits = 0; st->n_tie_take = 0; st->min_sel_bits = 0xffffffffu; st->nnz = 0;
        for (int i = 0; i < 8; ++i) st->pad[i] = 0;
    }
    uint32_t tot = 0, over = 0;
    if (carry[0] != 0)
        for (int r = t; r < B; r += 1024) { const uint32_t n = cand_cnt[r]; tot += n; over |= n > (uint32_t)C ? 1u : 0u; }
    tot = wave_sum_u32(tot); over = wave_sum_u32(over);
    if (lane == 0) { s_tot[w] = tot; s_over[w] = over; }
    __syncthreads();
    if (t == 0) {
        tot = 0; over = 0;
        for (int i = 0; i < 16; ++i) { tot += s_tot[i]; over += s_over[i]; }
        const uint32_t ok = (carry[0] != 0 && tot >= k_total && over == 0) ? 1u : 0u;
        carry[1] = ok; carry[2] = tot; carry[3] = over;
        carry[4] += ok; carry[5] += 1u - ok;
    }
}

// ---------------------------------------------------------------- helpers of the feature-sharded caller
// offsets with the tie order of a sharded dictionary: ties are granted in (row, shard, feature) order, so the
// ties that precede this shard's row r are all ties of earlier rows (every shard) plus those of earlier shards in
// row r.  all_eq [world][B] = the all-gathered per-row tie counts; the quota in st is the global one.
// Row stride of all_eq is `stride` (>= B); with stride > B, all_eq[w][B] carries shard w's smallest selected value
// (bits) and minmax_out receives {global min bits, -(anything selected anywhere)} for mtc_threshold_ema_global.
__global__ __launch_bounds__(1024) void select_offsets_sharded_kernel(int B, const uint32_t* __restrict__ row_gt,
                                                                      const uint32_t* __restrict__ all_eq_, int stride, int world,
                                                                      int rank, mtc_select_state* st, int32_t* __restrict__ row_ptr,
                                                                      uint32_t* __restrict__ tie_take,
                                                                      long long* __restrict__ minmax_out) {
    const uint32_t* all_eq = all_eq_;
#define B_STRIDE stride
    __shared__ uint32_t sm[16];
    const int t = threadIdx.x;
    const int per = (B + 1023) / 1024;
    const int r0 = t * per, r1 = min(B, r0 + per);
    const uint32_t quota = st->n_tie_take;
    uint32_t my_eq = 0;
    for (int r = r0; r < r1; ++r)
        for (int w = 0; w < world; ++w) my_eq += all_eq[(size_t)w * B_STRIDE + r];
    uint32_t tot;
    uint32_t run = block_excl_scan_1024(my_eq, sm, &tot);
    uint32_t my_nnz = 0;
    for (int r = r0; r < r1; ++r) {
        uint32_t row_tot = 0, earlier = 0;
        for (int w = 0; w < world; ++w) {
            const uint32_t e = all_eq[(size_t)w * B_STRIDE + r];
            row_tot += e;
            if (w < rank) earlier += e;
        }
        const uint32_t mine = all_eq[(size_t)rank * B_STRIDE + r];
        const uint32_t before = run + earlier;
        const uint32_t take = before >= quota ? 0u : min(mine, quota - before);
        tie_take[r] = take;
        my_nnz += row_gt[r] + take;
        run += row_tot;
    }
    uint32_t nnz_total;
    uint32_t base = block_excl_scan_1024(my_nnz, sm, &nnz_total);
    for (int r = r0; r < r1; ++r) {
        row_ptr[r] = (int32_t)base;
        base += row_gt[r] + tie_take[r];
    }
    if (t == 0) {
        row_ptr[B] = (int32_t)nnz_total;
        st->nnz = nnz_total;
        if (quota > 0) st->min_sel_bits = min(st->min_sel_bits, st->tau_bits);
        if (minmax_out && stride > B) {
            uint32_t gmin = 0xffffffffu;
            for (int w = 0; w < world; ++w) gmin = min(gmin, all_eq[(size_t)w * B_STRIDE + B]);
            if (quota > 0) gmin = min(gmin, st->tau_bits);
            // the histogram was all-reduced, so n_gt / the quota in st are global counts
            const bool any = st->all_pos ? st->n_gt > 0 : (st->n_gt + st->n_tie_take) > 0;
            minmax_out[0] = (long long)gmin;
            minmax_out[1] = any ? -1ll : 0ll;
        }
    }
#undef B_STRIDE
}

// {min selected bits, -nnz} as int64 for ONE all-reduce(MIN) over the shards, and the EMA from its result
__global__ void select_pack_minmax_kernel(const mtc_select_state* __restrict__ st, long long* __restrict__ out) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    out[0] = (long long)st->min_sel_bits;
    out[1] = -(long long)st->nnz;
}
template <int DT>
__global__ void threshold_ema_global_kernel(const long long* __restrict__ glob, void* threshold, float lr) {
    typedef DType<DT> D_;
    typedef typename D_::T T;
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    if (glob[1] == 0 || glob[0] >= 0xffffffffll) return;               // nothing selected anywhere
    T* p = reinterpret_cast<T*>(threshold);
    const float thr = D_::ld(p);
    const float mn = __uint_as_float((uint32_t)glob[0]);
    const float a = D_::rnd((float)(1.0 - (double)lr) * thr);
    const float b = D_::rnd(lr * mn);
    D_::st(p, D_::rnd(a + b));
}
// out[0] = nnz of this shard, out[1] = sum |a| of its selected entries (one workgroup; feeds one all-reduce(SUM))
__global__ __launch_bounds__(1024) void nnz_abs_kernel(int B, const float* __restrict__ abs_rows, const mtc_select_state* __restrict__ st,
                                                       float* __restrict__ out) {
    __shared__ float sm[16];
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    float s = 0.f;
    for (int r = t; r < B; r += 1024) s += abs_rows[r];
    s = wave_sum(s);
    if (lane == 0) sm[w] = s;
    __syncthreads();
    if (t == 0) {
        float tot = 0.f;
        for (int i = 0; i < 16; ++i) tot += sm[i];
        out[0] = (float)st->nnz;
        out[1] = tot;
    }
}

template <int DT>
__device__ __forceinline__ void threshold_ema_body(const mtc_select_state* st, void* threshold, float lr) {
    typedef DType<DT> D_;
    typedef typename D_::T T;
    if (st->nnz == 0 || st->min_sel_bits == 0xffffffffu) return;     // "if positive_mask.any()"
    T* p = reinterpret_cast<T*>(threshold);
    const float thr = D_::ld(p);
    const float mn = __uint_as_float(st->min_sel_bits);
    const float a = D_::rnd((float)(1.0 - (double)lr) * thr);
    const float b = D_::rnd(lr * mn);
    D_::st(p, D_::rnd(a + b));
}

// The single-workgroup tail of the carried-bound select as ONE launch: [bf16: sub-bin totals -> tau / quota -> per-row
// counts] -> CSR offsets, tie quota, next step's bound -> [threshold EMA, sae.py:349-361, when the buffer is given].
// The pieces run back to back in the same 1024 threads; each piece's global results are read by the next after a
// barrier (one workgroup: its own writes are visible to it).
template <int DT>
__global__ __launch_bounds__(1024) void select_tail_kernel(int B, mtc_select_state* st, int with_finish,
                                                           const uint32_t* __restrict__ row_above,
                                                           const uint32_t* __restrict__ row_sub, uint32_t* row_gt,
                                                           uint32_t* row_eq, int32_t* __restrict__ row_ptr,
                                                           uint32_t* __restrict__ tie_take, uint32_t* carry, void* threshold,
                                                           float lr, const uint32_t* __restrict__ pred = nullptr,
                                                           uint32_t want = 0) {
    if (pred && *pred != want) return;                 // uniform (the window tail served this call)
    if (with_finish) {
        select_finish_body(B, st, row_above, row_sub, nullptr, row_gt, row_eq);
        __threadfence_block();
        __syncthreads();
    }
    select_offsets_body(B, row_gt, row_eq, nullptr, st, row_ptr, tie_take, carry);
    if (threshold && threadIdx.x == 0) threshold_ema_body<DT>(st, threshold, lr);     // thread 0 wrote st->nnz / min_sel_bits itself
}

// bf16 fast path of the carried-bound select: ONE workgroup turns the candidate pass's per-row window counts into the whole
// answer.  A bf16 key has 16 bits and every candidate is >= the bound, so "key16 - bound16" is an EXACT histogram bin.
// row_win [B][32] holds, per row, the inclusive prefix counts over those bins (bin 31 = "31 ulps or more": every candidate of
// the row).  Their column sums S[b] are the global counts "at most b ulps above the bound": tot - S[b] entries lie above bin
// b, so the scan from the top gives tau, the count above it and the tie quota (what select_hist / select_scan /
// select_rowsub / the finish piece produce in four launches), and two entries per row give its (> tau, == tau) counts.  Then
// CSR offsets with the tie quota in ascending flat index, the next bound and the threshold EMA (select_offsets_body /
// threshold_ema_body restated on registers).  Valid when the bound held (enough candidates, no row overflow) AND tau lies
// inside the window (the bound sits 8 ulps under the previous tau, so tau may rise 22 ulps in one step); otherwise
// carry[1] = 0, the level histograms are zeroed and the full-size launches behind this one run instead (they leave at once
// when carry[1] == 1).  Also initialises st (what mtc_select_init does).
// A single workgroup pays ~2 us per dependent round trip to L2 and one CU moves ~one cache line per clock, so every global
// value is requested once, in batches, with consecutive lanes on consecutive addresses: the column sums read row_win as a
// flat array (16-byte piece u = i * 1024 + t: a thread always sees the same quarter of the bins), rows t, t + 1024, ... belong
// to thread t for the per-row part (B <= 4096; REGS), whose four row segments are scanned side by side.  Larger batches
// (REGS = false) loop over a contiguous row range per thread.
__device__ __forceinline__ void block_excl_scan4_1024(const uint32_t (&v)[4], uint32_t (*sm)[4], uint32_t (&excl)[4],
                                                      uint32_t (&total)[4]) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    uint32_t inc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) inc[j] = v[j];
#pragma unroll
    for (int o = 1; o < 64; o <<= 1)
#pragma unroll
        for (int j = 0; j < 4; ++j) { const uint32_t n = __shfl_up(inc[j], o, 64); if (lane >= o) inc[j] += n; }
    if (lane == 63)
#pragma unroll
        for (int j = 0; j < 4; ++j) sm[w][j] = inc[j];
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        uint32_t wbase = 0, tot = 0;
        for (int i = 0; i < 16; ++i) { const uint32_t x = sm[i][j]; if (i < w) wbase += x; tot += x; }
        excl[j] = wbase + inc[j] - v[j];
        total[j] = tot;
    }
    __syncthreads();
}

template <int DT, bool REGS>
__global__ __launch_bounds__(1024) void select_window_tail_kernel(int B, int C, uint32_t k_total, mtc_select_state* st,
                                                                  uint32_t* carry, const uint32_t* __restrict__ cand_cnt,
                                                                  const unsigned short* __restrict__ row_win,
                                                                  uint32_t* __restrict__ hist, uint32_t* row_gt, uint32_t* row_eq,
                                                                  int32_t* __restrict__ row_ptr, uint32_t* __restrict__ tie_take,
                                                                  void* threshold, float lr) {
    static_assert(CAND_WIN == 32, "the column sums below are laid out for 32 bins = 4 16-byte pieces per row");
    typedef DType<DT> D_;
    __shared__ uint32_t s_x[8][1024];
    __shared__ uint32_t s_y[64][CAND_WIN];
    __shared__ uint32_t s_part[16][2];
    __shared__ uint32_t s_col[CAND_WIN + 2];
    __shared__ uint32_t s_ok, s_bin, s_quota;
    __shared__ uint32_t sm[16];
    __shared__ uint32_t sm4[16][4];
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const uint32_t guess = carry[0], served_fast = carry[4], served_full = carry[5];
    float thr_old = 0.f;
    if (threshold && t == 0) thr_old = D_::ld(reinterpret_cast<const typename D_::T*>(threshold));
    const int per = (B + 1023) / 1024;
    const int r0 = t * per, r1 = min(B, r0 + per);                   // REGS = false: this thread's rows
    uint32_t cnt[4] = {0u, 0u, 0u, 0u};                              // REGS: candidates of rows t + 1024 j
    uint32_t tot = 0, over = 0;
    if (REGS) {
        const int n16 = B * 4;                                       // 16-byte pieces of row_win
        uint4 piece[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int u = i * 1024 + t;
            piece[i] = (u < n16 && guess != 0) ? reinterpret_cast<const uint4*>(row_win)[u] : uint4{0u, 0u, 0u, 0u};
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int r = t + 1024 * j;
            cnt[j] = (r < B && guess != 0) ? cand_cnt[r] : 0u;
            tot += cnt[j]; over |= cnt[j] > (uint32_t)C ? 1u : 0u;
        }
        // packed 16-bit adds over the thread's 16 pieces: exact whenever no row overflowed the compact matrix (a prefix
        // count is then <= C = 1024, the sum <= 16384); with an overflowing row the verdict is "failed" from `over` alone
        uint32_t ps[4] = {0u, 0u, 0u, 0u};
#pragma unroll
        for (int i = 0; i < 16; ++i) { ps[0] += piece[i].x; ps[1] += piece[i].y; ps[2] += piece[i].z; ps[3] += piece[i].w; }
#pragma unroll
        for (int k = 0; k < 4; ++k) s_x[k][t] = ps[k];
        tot = wave_sum_u32(tot); over = wave_sum_u32(over);
        if (lane == 0) { s_part[w][0] = tot; s_part[w][1] = over; }
        __syncthreads();
        // wave k adds word k of the threads l, l + 64, ... (same quarter q = l & 3 of the bins): bins 2 (4 q + k), + 1
        if (w < 4) {
            uint32_t lo = 0, hi = 0;
#pragma unroll
            for (int j = 0; j < 16; ++j) { const uint32_t v = s_x[w][lane + 64 * j]; lo += v & 0xffffu; hi += v >> 16; }
            s_y[lane][2 * w] = lo;
            s_y[lane][2 * w + 1] = hi;
        }
        __syncthreads();
        if (t < CAND_WIN) {
            const int word = t >> 1, q = word >> 2, k = word & 3, h = t & 1;
            uint32_t v = 0;
#pragma unroll
            for (int l = 0; l < 16; ++l) v += s_y[4 * l + q][2 * k + h];
            s_col[t] = v;
        }
    } else {
        uint32_t acc[CAND_WIN];
#pragma unroll
        for (int b = 0; b < CAND_WIN; ++b) acc[b] = 0;
        if (guess != 0)
            for (int r = r0; r < r1; ++r) {
                const uint32_t n = cand_cnt[r];
                tot += n; over |= n > (uint32_t)C ? 1u : 0u;
                const uint4* p = reinterpret_cast<const uint4*>(row_win + (size_t)r * CAND_WIN);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const uint4 v = p[q];
                    const uint32_t x[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                    for (int i = 0; i < 4; ++i) { acc[q * 8 + 2 * i] += x[i] & 0xffffu; acc[q * 8 + 2 * i + 1] += x[i] >> 16; }
                }
            }
        tot = wave_sum_u32(tot); over = wave_sum_u32(over);
        if (lane == 0) { s_part[w][0] = tot; s_part[w][1] = over; }
        // wave c adds word c of the threads l, l + 64, ... (bank = lane), eight bins at a time
#pragma unroll
        for (int part = 0; part < 4; ++part) {
            if (part) __syncthreads();
#pragma unroll
            for (int c = 0; c < 8; ++c) s_x[c][t] = acc[part * 8 + c];
            __syncthreads();
            if (w < 8) {
                uint32_t v = 0;
#pragma unroll
                for (int k = 0; k < 16; ++k) v += s_x[w][lane + 64 * k];
                s_y[lane][part * 8 + w] = v;
            }
        }
        __syncthreads();
        if (t < CAND_WIN) {
            uint32_t v = 0;
#pragma unroll
            for (int l = 0; l < 64; ++l) v += s_y[l][t];
            s_col[t] = v;
        }
    }
    if (t >= CAND_WIN && t < CAND_WIN + 2) {
        uint32_t v = 0;
#pragma unroll
        for (int i = 0; i < 16; ++i) v += s_part[i][t - CAND_WIN];
        s_col[t] = v;
    }
    __syncthreads();
    if (t == 0) {
        tot = s_col[CAND_WIN]; over = s_col[CAND_WIN + 1];
        uint32_t found = 0, bin = 0, above = 0;
        if (guess != 0 && over == 0 && tot >= k_total) {
            // entries above bin b: tot - S[b]; tau's bin is the highest one with at least k_total entries at or above it
            for (int b = CAND_WIN - 2; b >= 0; --b) {
                const uint32_t ab = tot - s_col[b];                // above bin b
                if (ab >= k_total) break;                            // (b = 30: tau lies further up than the window resolves)
                const uint32_t at = tot - (b ? s_col[b - 1] : 0u);   // at or above bin b
                if (at >= k_total) { found = 1; bin = (uint32_t)b; above = ab; break; }
            }
        }
        st->all_pos = 0;
        for (int i = 0; i < 8; ++i) st->pad[i] = 0;
        if (found) {
            st->n_gt = above; st->k_rem = k_total - above; st->n_tie_take = k_total - above;
            st->key_prefix = st->tau_bits = guess + (bin << 16);
        } else {
            st->key_prefix = 0; st->k_rem = k_total; st->n_gt = 0; st->tau_bits = 0; st->n_tie_take = 0;
            st->min_sel_bits = 0xffffffffu; st->nnz = 0;
        }
        carry[1] = found; carry[2] = tot; carry[3] = over;
        carry[4] = served_fast + found; carry[5] = served_full + 1u - found;
        s_ok = found; s_bin = bin; s_quota = k_total - above;
    }
    __syncthreads();
    if (!s_ok) {                                                     // uniform: the full-size launches take over
        for (int i = t; i < 3 * HIST_BINS; i += 1024) hist[i] = 0;
        return;
    }
    const uint32_t bin = s_bin, quota = s_quota;
    uint32_t nnz_total;
    if (REGS) {
        // rows t + 1024 j: four row segments scanned side by side; ties are granted in ascending flat index, so a row takes
        // what the quota has left after the rows before it (earlier segments first)
        unsigned short at[4], below[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int r = min(t + 1024 * j, B - 1);
            at[j] = row_win[(size_t)r * CAND_WIN + bin];
            below[j] = bin ? row_win[(size_t)r * CAND_WIN + bin - 1] : (unsigned short)0;
        }
        uint32_t gt[4], eq[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const bool in = t + 1024 * j < B;
            gt[j] = in ? cnt[j] - at[j] : 0u;
            eq[j] = in ? (uint32_t)at[j] - below[j] : 0u;
        }
        uint32_t ex[4], totals[4];
        block_excl_scan4_1024(eq, sm4, ex, totals);
        uint32_t nn[4], take[4], seg = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const uint32_t before = seg + ex[j];
            take[j] = before >= quota ? 0u : min(eq[j], quota - before);
            nn[j] = gt[j] + take[j];
            seg += totals[j];
        }
        block_excl_scan4_1024(nn, sm4, ex, totals);
        seg = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int r = t + 1024 * j;
            if (r < B) { row_ptr[r] = (int32_t)(seg + ex[j]); tie_take[r] = take[j]; }
            seg += totals[j];
        }
        nnz_total = seg;
    } else {
        uint32_t dummy, my_eq = 0;
        for (int r = r0; r < r1; ++r) {
            const uint32_t at = row_win[(size_t)r * CAND_WIN + bin], below = bin ? row_win[(size_t)r * CAND_WIN + bin - 1] : 0u;
            row_gt[r] = cand_cnt[r] - at; row_eq[r] = at - below;
            my_eq += at - below;
        }
        uint32_t run = block_excl_scan_1024(my_eq, sm, &dummy);
        uint32_t my_nnz = 0;
        for (int r = r0; r < r1; ++r) {
            const uint32_t e = row_eq[r];
            const uint32_t take = run >= quota ? 0u : min(e, quota - run);
            tie_take[r] = take;
            my_nnz += row_gt[r] + take;
            run += e;
        }
        uint32_t base = block_excl_scan_1024(my_nnz, sm, &nnz_total);
        for (int r = r0; r < r1; ++r) {
            row_ptr[r] = (int32_t)base;
            base += row_gt[r] + tie_take[r];
        }
    }
    if (t == 0) {
        row_ptr[B] = (int32_t)nnz_total;
        st->nnz = nnz_total;
        const uint32_t tau_bits = guess + (bin << 16);
        st->min_sel_bits = tau_bits;                                 // with a tau the smallest selected value is tau (quota >= 1)
        // next step's lower bound = tau lowered by 8 bf16 ulps (select_offsets_body)
        const uint32_t margin = 8u << 16;
        carry[0] = (nnz_total != 0 && tau_bits > margin + 0x00800000u && tau_bits <= 0x7f800000u) ? tau_bits - margin : 0u;
        if (threshold && nnz_total != 0) {
            // threshold_ema_body on the values at hand: (1 - lr) * threshold + lr * min_positive, each op rounded to the buffer dtype
            const float a = D_::rnd((float)(1.0 - (double)lr) * thr_old);
            const float b = D_::rnd(lr * __uint_as_float(tau_bits));
            D_::st(reinterpret_cast<typename D_::T*>(threshold), D_::rnd(a + b));
        }
    }
}

// Fallback of the bf16 carried-bound select (carry[1] == 0: no bound yet, the bound failed, or tau lay beyond the window)
// behind the full-size level-0 histogram: scan + per-row pass + tail as ONE launch, so a step served by the window pays for
// two empty launches instead of four.  Every workgroup repeats the scan of the 4096 bins for itself (a pure function of
// the histogram: tau's level-0 bin and the count above it), 16 rows per workgroup are counted as in
// select_rowsub_bf16_kernel, and the workgroup whose ticket comes last -- every other one has then released its rows --
// applies the scan to st and runs the single-workgroup tail (select_tail_kernel's pieces).  *ticket must be 0 at launch
// (the window tail zeroes it together with the histograms).
__global__ __launch_bounds__(1024) void select_fallback_rows_kernel(const void* __restrict__ acts, int B, int D,
                                                                    mtc_select_state* st, const uint32_t* __restrict__ hist,
                                                                    uint32_t* row_above, uint32_t* row_sub, uint32_t* row_gt,
                                                                    uint32_t* row_eq, int32_t* __restrict__ row_ptr,
                                                                    uint32_t* __restrict__ tie_take, uint32_t* carry,
                                                                    void* threshold, float lr, const uint32_t* __restrict__ pred,
                                                                    uint32_t* ticket) {
    __shared__ uint32_t part[1024];
    __shared__ uint32_t found_bin, found_above, s_last;
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    if (*pred != 0) return;                            // uniform: the window tail served this call
    // ---- scan (select_scan_kernel, level 0), results in LDS only
    const uint32_t h0 = hist[4 * t], h1 = hist[4 * t + 1], h2 = hist[4 * t + 2], h3 = hist[4 * t + 3];
    part[t] = h0 + h1 + h2 + h3;
    if (t == 0) { found_bin = 0xffffffffu; found_above = 0; }
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
        const uint32_t add = (t + off < 1024) ? part[t + off] : 0u;
        __syncthreads();
        part[t] += add;
        __syncthreads();
    }
    const uint32_t k_rem = st->k_rem;                  // = k_total: st was initialised by the launch before the histogram
    {
        uint32_t above = (t + 1 < 1024) ? part[t + 1] : 0u;
        const uint32_t hs[4] = {h0, h1, h2, h3};
#pragma unroll
        for (int i = 3; i >= 0; --i) {
            if (above < k_rem && k_rem <= above + hs[i]) { found_bin = 4 * t + i; found_above = above; }
            above += hs[i];
        }
    }
    __syncthreads();
    const uint32_t total = part[0];
    const bool all_pos = found_bin == 0xffffffffu;     // fewer positives than k: everything positive is selected
    const uint32_t bin = found_bin;
    // ---- per-row pass (select_rowsub_bf16_kernel), wave per row
    const int row = blockIdx.x * 16 + w;
    if (row < B) {
        const size_t base = (size_t)row * D;
        const int nch = D >> 3;
        uint32_t above = 0, mn = 0xffffffffu;
        uint32_t sub[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int c0 = lane; c0 < nch; c0 += 512) {
            Raw<MTC_BF16> r[8];
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (c0 + 64 * u < nch) r[u] = load_raw<MTC_BF16>(acts, base + (size_t)(c0 + 64 * u) * 8);
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                if (c0 + 64 * u >= nch) break;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const uint32_t k = h ? (r[u].v[i] & 0xffff0000u) : (r[u].v[i] << 16);
                        if (k == 0 || k > 0x7f800000u) continue;
                        const uint32_t kb = k >> 19;
                        if (all_pos || kb > bin) { ++above; mn = min(mn, k); }
                        else if (kb == bin) {
                            const uint32_t sidx = (k >> 16) & 7u;
#pragma unroll
                            for (int q = 0; q < 8; ++q) sub[q] += (sidx == (uint32_t)q) ? 1u : 0u;
                        }
                    }
                }
            }
        }
        above = wave_sum_u32(above);
#pragma unroll
        for (int q = 0; q < 8; ++q) sub[q] = wave_sum_u32(sub[q]);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) mn = min(mn, (uint32_t)__shfl_xor((int)mn, o, 64));
        if (lane == 0) {
            row_above[row] = above;
#pragma unroll
            for (int q = 0; q < 8; ++q) row_sub[(size_t)row * 8 + q] = sub[q];
            if (all_pos) note_min_selected(st, mn);
        }
    }
    // ---- hand-over: every wave's stores have left, one lane releases them and takes a ticket
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (t == 0) {
        __threadfence();
        s_last = atomicAdd(ticket, 1u) == gridDim.x - 1 ? 1u : 0u;
    }
    __syncthreads();
    if (!s_last) return;                               // uniform
    __threadfence();                                   // acquire: the other workgroups' rows, not this CU's cached lines
    __syncthreads();
    if (t == 0) {
        if (all_pos) {
            st->all_pos = 1; st->n_gt = total; st->tau_bits = 0; st->n_tie_take = 0; st->k_rem = 0;
        } else {
            st->n_gt += found_above;
            st->k_rem = k_rem - found_above;
            st->key_prefix |= bin << 19;
        }
    }
    __threadfence_block();
    __syncthreads();
    select_finish_body(B, st, row_above, row_sub, nullptr, row_gt, row_eq);
    __threadfence_block();
    __syncthreads();
    select_offsets_body(B, row_gt, row_eq, nullptr, st, row_ptr, tie_take, carry);
    if (threshold && t == 0) threshold_ema_body<MTC_BF16>(st, threshold, lr);
}

template <int DT>
__global__ void threshold_ema_kernel(const mtc_select_state* __restrict__ st, void* threshold, float lr) {
    typedef DType<DT> D_;
    typedef typename D_::T T;
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    if (st->nnz == 0 || st->min_sel_bits == 0xffffffffu) return;     // "if positive_mask.any()"
    T* p = reinterpret_cast<T*>(threshold);
    const float thr = D_::ld(p);
    const float mn = __uint_as_float(st->min_sel_bits);
    // (1 - lr) * threshold + lr * min_positive, each op rounded to the buffer dtype
    const float a = D_::rnd((float)(1.0 - (double)lr) * thr);
    const float b = D_::rnd(lr * mn);
    D_::st(p, D_::rnd(a + b));
}

// ---------------------------------------------------------------- feature-sharded select with the carried bound
// One process per GPU holds D / world features of every prefix group.  The exact global select needs three facts from
// the other shards, each ONE collective of fixed shape (nothing is read back, no collective is conditional):
//   X1  all-reduce of xbuf[4104]: a 4096-bin WINDOW histogram of this shard's candidates (entries >= the carried bound;
//       bf16 keys have 15 bits, so bin = key16 - bound16 is EXACT: the scan of the reduced window yields tau itself, the
//       count above it and the tie quota) + {candidates, rows that overflowed the compact matrix};
//   X2  all-reduce of hist_full[32768]: the whole bf16 key space, filled ONLY when X1 says the bound failed somewhere
//       (too few candidates globally, an overflowing row on any shard, tau in the clamped top bin) -- all zeros otherwise;
//   X3  all-gather of pack[B*G + 1]: per (row, prefix group) counts of entries equal to tau (+ the shard's smallest
//       positive value when fewer than k positives exist), from which every shard grants the ties in ascending GLOBAL flat
//       index (row, then feature: group, shard, local feature) -- the order of the single-GPU select, so the selection
//       does not depend on the number of shards.
// Granting in that order is a prefix of every shard's local (row, column) order of ties, so one number per row
// (tie_take) still describes it and the ordered fill is the single-GPU kernel.
constexpr int WIN_BINS = 4096;
constexpr int FULL_BINS = 32768;
constexpr int XBUF_WORDS = 3 * HIST_BINS;          // window + 2 counters (bf16), or the three level histograms (fp32)

struct GroupBounds { int G; int prefix[MTC_MAX_PREFIX]; };

__global__ __launch_bounds__(1024) void shard_select_init_kernel(mtc_select_state* st, uint32_t k_total, uint32_t* xbuf,
                                                                 uint32_t* hist_full, uint32_t* pack_min) {
    const int t = threadIdx.x;
    for (int i = t; i < XBUF_WORDS; i += 1024) xbuf[i] = 0;
    if (hist_full) for (int i = t; i < FULL_BINS; i += 1024) hist_full[i] = 0;
    if (t == 0) {
        st->key_prefix = 0; st->k_rem = k_total; st->all_pos = 0; st->n_gt = 0;
        st->tau_bits = 0; st->n_tie_take = 0; st->min_sel_bits = 0xffffffffu; st->nnz = 0;
        for (int i = 0; i < 8; ++i) st->pad[i] = 0;
        *pack_min = 0xffffffffu;
    }
}

// window histogram of the compact candidate matrix (bf16): a workgroup walks rows blockIdx.x, + gridDim.x, ...; only the
// filled part of a row is read (cand_cnt).  Block 0 also leaves {candidates, overflowing rows} behind the window.
__global__ __launch_bounds__(256) void shard_window_hist_kernel(const u16* __restrict__ cand_vals, const uint32_t* __restrict__ cand_cnt,
                                                                int B, int C, const uint32_t* __restrict__ carry,
                                                                uint32_t* __restrict__ xbuf) {
    __shared__ uint32_t lh[WIN_BINS];
    __shared__ uint32_t s_tot[4], s_over[4];
    const uint32_t guess = carry[0];
    if (guess == 0) return;                             // no bound: the candidate pass did not run (uniform)
    const uint32_t g16 = guess >> 16;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < WIN_BINS; i += 256) lh[i] = 0;
    __syncthreads();
    for (int row = blockIdx.x * 4 + w; row < B; row += gridDim.x * 4) {
        const int n = min((int)cand_cnt[row], C);
        const u16* src = cand_vals + (size_t)row * C;
        for (int c = lane; c * 8 < n; c += 64) {
            const u32x4 r = *reinterpret_cast<const u32x4*>(src + (size_t)c * 8);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const uint32_t k16 = h ? (r[i] >> 16) : (r[i] & 0xffffu);
                    if (k16 >= g16 && k16 <= 0x7f80u) atomicAdd(&lh[min(k16 - g16, (uint32_t)WIN_BINS - 1)], 1u);   // (0 = padding)
                }
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < WIN_BINS; i += 256) {
        const uint32_t v = lh[i];
        if (v) atomicAdd(&xbuf[i], v);
    }
    if (blockIdx.x == 0) {
        uint32_t tot = 0, over = 0;
        for (int r = threadIdx.x; r < B; r += 256) { const uint32_t n = cand_cnt[r]; tot += n; over += n > (uint32_t)C ? 1u : 0u; }
        tot = wave_sum_u32(tot); over = wave_sum_u32(over);
        if (lane == 0) { s_tot[w] = tot; s_over[w] = over; }
        __syncthreads();
        if (threadIdx.x == 0) {
            xbuf[WIN_BINS] = s_tot[0] + s_tot[1] + s_tot[2] + s_tot[3];
            xbuf[WIN_BINS + 1] = s_over[0] + s_over[1] + s_over[2] + s_over[3];
        }
    }
}

// suffix scan of `nbins` reduced bins by one workgroup of 1024 threads (nbins a multiple of 1024): the bin that holds the
// k-th largest entry, and the count above it.  Returns through shared memory; found_bin = ~0 when the bins hold fewer than k.
template <int PER>
__device__ __forceinline__ void suffix_find(const uint32_t* __restrict__ hist, uint32_t k, uint32_t* part /*[1024]*/,
                                            uint32_t* found /*[3]: bin, above, total*/) {
    const int t = threadIdx.x;
    uint32_t hs[PER], mine = 0;
#pragma unroll
    for (int i = 0; i < PER; ++i) { hs[i] = hist[PER * t + i]; mine += hs[i]; }
    part[t] = mine;
    if (t == 0) { found[0] = 0xffffffffu; found[1] = 0; }
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
        const uint32_t add = (t + off < 1024) ? part[t + off] : 0u;
        __syncthreads();
        part[t] += add;
        __syncthreads();
    }
    uint32_t above = (t + 1 < 1024) ? part[t + 1] : 0u;
#pragma unroll
    for (int i = PER - 1; i >= 0; --i) {
        if (above < k && k <= above + hs[i]) { found[0] = PER * t + i; found[1] = above; }
        above += hs[i];
    }
    if (t == 0) found[2] = part[0];
    __syncthreads();
}

// after X1: is the carried bound valid everywhere?  If so tau / n_gt / the tie quota follow from the window.
__global__ __launch_bounds__(1024) void shard_window_scan_kernel(const uint32_t* __restrict__ xbuf, uint32_t k_total,
                                                                 uint32_t* carry, mtc_select_state* st) {
    __shared__ uint32_t part[1024];
    __shared__ uint32_t found[3];
    suffix_find<WIN_BINS / 1024>(xbuf, k_total, part, found);
    if (threadIdx.x != 0) return;
    const uint32_t guess = carry[0], tot = xbuf[WIN_BINS], over = xbuf[WIN_BINS + 1];
    // (the clamped top bin mixes values: a tau in it cannot be told from the window)
    const uint32_t ok = (guess != 0 && over == 0 && tot >= k_total && k_total > 0 && found[0] != 0xffffffffu &&
                         found[0] < (uint32_t)WIN_BINS - 1 && found[2] == tot) ? 1u : 0u;
    if (ok) {
        const uint32_t tau = ((guess >> 16) + found[0]) << 16;
        st->key_prefix = tau; st->tau_bits = tau; st->n_gt = found[1];
        st->k_rem = k_total - found[1]; st->n_tie_take = k_total - found[1]; st->all_pos = 0;
    }
    carry[1] = ok; carry[2] = tot; carry[3] = over;
    carry[4] += ok; carry[5] += 1u - ok;
}

// fallback (flag == 0 only): the whole bf16 key space of this shard's activations, 128 KiB of LDS bins per workgroup
__global__ __launch_bounds__(256) void shard_full_hist_bf16_kernel(const u16* __restrict__ acts, size_t n_chunks,
                                                                   const uint32_t* __restrict__ flag,
                                                                   uint32_t* __restrict__ hist_full) {
    extern __shared__ uint32_t lhf[];
    if (*flag != 0) return;                             // uniform: the window served this call
    for (int i = threadIdx.x; i < FULL_BINS; i += 256) lhf[i] = 0;
    __syncthreads();
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t c = (size_t)blockIdx.x * 256 + threadIdx.x; c < n_chunks; c += stride) {
        const u32x4 r = *reinterpret_cast<const u32x4*>(acts + c * 8);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const uint32_t k16 = h ? (r[i] >> 16) : (r[i] & 0xffffu);
                if (k16 != 0 && k16 <= 0x7f80u) atomicAdd(&lhf[k16], 1u);
            }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < FULL_BINS; i += 256) {
        const uint32_t v = lhf[i];
        if (v) atomicAdd(&hist_full[i], v);
    }
}

// after X2 (flag == 0 only): tau from the reduced full histogram, or "every positive is selected"
__global__ __launch_bounds__(1024) void shard_full_scan_kernel(const uint32_t* __restrict__ hist_full, uint32_t k_total,
                                                               const uint32_t* __restrict__ flag, mtc_select_state* st) {
    __shared__ uint32_t part[1024];
    __shared__ uint32_t found[3];
    if (*flag != 0) return;                             // uniform
    suffix_find<FULL_BINS / 1024>(hist_full, k_total, part, found);
    if (threadIdx.x != 0) return;
    if (found[0] == 0xffffffffu) {
        st->all_pos = 1; st->n_gt = found[2]; st->tau_bits = 0; st->n_tie_take = 0; st->k_rem = 0; st->key_prefix = 0;
    } else {
        const uint32_t tau = found[0] << 16;
        st->key_prefix = tau; st->tau_bits = tau; st->n_gt = found[1];
        st->k_rem = k_total - found[1]; st->n_tie_take = k_total - found[1]; st->all_pos = 0;
    }
}

// wave per row: entries > tau (row_gt) and entries == tau per prefix group (pack[row * G + g]); reads the compact
// candidate matrix (+ its column map) when *flag == 1, the activations otherwise (flag == null: always the activations).
template <int DT>
__global__ __launch_bounds__(256) void shard_group_counts_kernel(const void* __restrict__ cand_vals, const int32_t* __restrict__ cand_cols,
                                                                 int C, const void* __restrict__ acts_full, int B, int D, GroupBounds gb,
                                                                 const mtc_select_state* __restrict__ st,
                                                                 const uint32_t* __restrict__ flag, uint32_t* __restrict__ row_gt,
                                                                 uint32_t* __restrict__ pack) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= B) return;
    const bool compact = flag && *flag == 1;            // uniform
    const void* __restrict__ src = compact ? cand_vals : acts_full;
    const int cols = compact ? C : D;
    const bool all_pos = st->all_pos != 0;
    const float tau = all_pos ? 0.f : __uint_as_float(st->tau_bits);
    const size_t base = (size_t)row * cols;
    uint32_t gt = 0, mn = 0xffffffffu;
    uint32_t eq[MTC_MAX_PREFIX];
#pragma unroll
    for (int q = 0; q < MTC_MAX_PREFIX; ++q) eq[q] = 0;
    const int nch = cols >> 3;
    for (int c0 = lane; c0 < nch; c0 += 256) {
        Raw<DT> r[4];
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (c0 + 64 * u < nch) r[u] = load_raw<DT>(src, base + (size_t)(c0 + 64 * u) * 8);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (c0 + 64 * u >= nch) break;
            float v[8];
            unpack(r[u], v);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (v[i] > tau) { ++gt; mn = min(mn, __float_as_uint(v[i])); }
                else if (!all_pos && v[i] == tau) {
                    const size_t at = base + (size_t)(c0 + 64 * u) * 8 + i;
                    const int col = compact ? cand_cols[at] : (c0 + 64 * u) * 8 + i;
                    int g = 0;
                    for (int q = 0; q + 1 < gb.G; ++q) g += col >= gb.prefix[q] ? 1 : 0;
#pragma unroll
                    for (int q = 0; q < MTC_MAX_PREFIX; ++q) eq[q] += (g == q) ? 1u : 0u;
                }
            }
        }
    }
    gt = wave_sum_u32(gt);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mn = min(mn, (uint32_t)__shfl_xor((int)mn, o, 64));
    for (int q = 0; q < gb.G; ++q) {
        uint32_t e = 0;
#pragma unroll
        for (int z = 0; z < MTC_MAX_PREFIX; ++z) e = z == q ? eq[z] : e;
        e = wave_sum_u32(e);
        if (lane == 0) pack[(size_t)row * gb.G + q] = e;
    }
    if (lane == 0) {
        row_gt[row] = gt;
        // only "every positive is selected" needs the smallest positive (with a tau the smallest selected value is tau)
        if (all_pos && mn != 0xffffffffu) atomicMin(&pack[(size_t)B * gb.G], mn);
    }
}

// one workgroup, after X3: CSR offsets + per-row tie quota in ascending GLOBAL flat index, the threshold EMA from the
// global smallest selected value, and the next call's bound.  all_pack [world][B * G + 1].
template <int DT>
__global__ __launch_bounds__(1024) void shard_offsets_groups_kernel(int B, int G, const uint32_t* __restrict__ row_gt,
                                                                    const uint32_t* __restrict__ all_pack, int world, int rank,
                                                                    mtc_select_state* st, int32_t* __restrict__ row_ptr,
                                                                    uint32_t* __restrict__ tie_take, uint32_t* carry,
                                                                    void* threshold, float lr) {
    __shared__ uint32_t sm[16];
    const int t = threadIdx.x;
    const int per = (B + 1023) / 1024;
    const int r0 = min(B, t * per), r1 = min(B, r0 + per);
    const size_t stride = (size_t)B * G + 1;
    const uint32_t quota = st->all_pos ? 0u : st->n_tie_take;
    uint32_t my_eq = 0;
    for (int r = r0; r < r1; ++r)
        for (int w = 0; w < world; ++w)
            for (int g = 0; g < G; ++g) my_eq += all_pack[(size_t)w * stride + (size_t)r * G + g];
    uint32_t tot;
    uint32_t run = block_excl_scan_1024(my_eq, sm, &tot);      // ties in earlier rows (every shard, every group)
    uint32_t my_nnz = 0;
    for (int r = r0; r < r1; ++r) {
        uint32_t take = 0;
        for (int g = 0; g < G; ++g) {
            uint32_t grp_tot = 0, earlier = 0;
            for (int w = 0; w < world; ++w) {
                const uint32_t e = all_pack[(size_t)w * stride + (size_t)r * G + g];
                grp_tot += e;
                if (w < rank) earlier += e;
            }
            const uint32_t mine = all_pack[(size_t)rank * stride + (size_t)r * G + g];
            const uint32_t before = run + earlier;             // global flat order: row, group, shard, local feature
            take += before >= quota ? 0u : min(mine, quota - before);
            run += grp_tot;
        }
        tie_take[r] = take;
        my_nnz += row_gt[r] + take;
    }
    uint32_t nnz_total;
    uint32_t base = block_excl_scan_1024(my_nnz, sm, &nnz_total);
    for (int r = r0; r < r1; ++r) {
        row_ptr[r] = (int32_t)base;
        base += row_gt[r] + tie_take[r];
    }
    if (t != 0) return;
    row_ptr[B] = (int32_t)nnz_total;
    st->nnz = nnz_total;
    // global smallest selected value: tau when one exists (the k-th largest entry is selected somewhere), else the
    // smallest positive over the shards
    uint32_t gmin = 0xffffffffu;
    bool any;
    if (st->all_pos) {
        for (int w = 0; w < world; ++w) gmin = min(gmin, all_pack[(size_t)w * stride + (size_t)B * G]);
        any = st->n_gt > 0;
    } else {
        gmin = st->tau_bits;
        any = st->n_gt + st->n_tie_take > 0;
    }
    st->min_sel_bits = gmin;
    if (threshold && any && gmin != 0xffffffffu) {
        typedef DType<DT> D_;
        typedef typename D_::T T;
        T* p = reinterpret_cast<T*>(threshold);
        const float thr = D_::ld(p);
        const float a = D_::rnd((float)(1.0 - (double)lr) * thr);
        const float b = D_::rnd(lr * __uint_as_float(gmin));
        D_::st(p, D_::rnd(a + b));
    }
    if (carry) {
        // the same on every shard: derived from global quantities only
        const uint32_t margin = 8u << 16;
        uint32_t g = 0;
        if (!st->all_pos && any && st->tau_bits > margin + 0x00800000u && st->tau_bits <= 0x7f800000u) g = st->tau_bits - margin;
        carry[0] = g;
    }
}

template <int DT>
__global__ __launch_bounds__(256) void densify_kernel(const int32_t* __restrict__ row_ptr, const int32_t* __restrict__ col_idx,
                                                      const void* __restrict__ vals, int B, int D, int max_col,
                                                      void* __restrict__ dense) {
    typedef typename DType<DT>::T T;
    const int row = blockIdx.x;
    T* out = reinterpret_cast<T*>(dense) + (size_t)row * D;
    for (int c = threadIdx.x; c < (D >> 3); c += 256) {
        const float z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        store8<DT>(out, (size_t)c * 8, z);
    }
    __syncthreads();
    const T* v = reinterpret_cast<const T*>(vals);
    for (int e = row_ptr[row] + threadIdx.x; e < row_ptr[row + 1]; e += 256) {
        const int c = col_idx[e];
        if (c < max_col) out[c] = v[e];
    }
}

}  // namespace mtc

using namespace mtc;

#define DISPATCH_DT(dt, KERNEL, grid, block, shmem, stream, ...)                                      \
    do {                                                                                              \
        if ((dt) == MTC_BF16) hipLaunchKernelGGL(KERNEL<MTC_BF16>, grid, block, shmem, stream, __VA_ARGS__); \
        else hipLaunchKernelGGL(KERNEL<MTC_F32>, grid, block, shmem, stream, __VA_ARGS__);            \
    } while (0)

extern "C" int mtc_select_levels(const mtc_desc* d) { return d && d->dtype == MTC_BF16 ? 2 : 3; }

extern "C" int mtc_select_init(mtc_select_state* st, uint32_t k_total, void* stream) {
    MTC_REQUIRE(st, "mtc_select_init: null state");
    hipLaunchKernelGGL(select_init_kernel, dim3(1), dim3(64), 0, as_stream(stream), st, k_total);
    return check_launch("mtc_select_init");
}

// `cols` = row length of the matrix the pass streams (D, or C for the compact candidate matrix)
// every workgroup zeroes and flushes a 4096-bin LDS histogram: give each at least 16 chunks per thread
static uint32_t hist_grid(size_t n_chunks) {
    return (uint32_t)std::max<size_t>(1, std::min<size_t>((n_chunks + 256 * 16 - 1) / (256 * 16), 2048));
}
// alt_acts != null (needs pred): one launch for both matrices, see the kernel
static int hist_impl(const mtc_desc* d, int cols, const void* acts, int level, const mtc_select_state* st, uint32_t* hist,
                     const uint32_t* pred, uint32_t want, void* stream, const void* alt_acts = nullptr, int alt_cols = 0) {
    const size_t n_chunks = (size_t)d->B * cols / 8, alt_chunks = (size_t)d->B * alt_cols / 8;
    const uint32_t g_main = hist_grid(n_chunks), g_alt = alt_acts ? hist_grid(alt_chunks) : 0;
    DISPATCH_DT(d->dtype, select_hist_kernel, dim3(std::max(g_main, g_alt)), dim3(256), 0, as_stream(stream), acts, n_chunks,
                level, st, hist, pred, want, alt_acts, alt_chunks, g_main, g_alt);
    return check_launch("mtc_select_hist");
}
extern "C" int mtc_select_hist(const mtc_desc* d, const void* acts, int level, const mtc_select_state* st,
                               uint32_t* hist, void* stream) {
    if (check_desc(d, "mtc_select_hist")) return -1;
    MTC_REQUIRE(acts && st && hist, "mtc_select_hist: null pointer");
    MTC_REQUIRE(level >= 0 && level < mtc_select_levels(d), "mtc_select_hist: bad level %d", level);
    return hist_impl(d, d->D, acts, level, st, hist, nullptr, 0, stream);
}

extern "C" int mtc_select_scan(const mtc_desc* d, int level, const uint32_t* hist, mtc_select_state* st, void* stream) {
    if (check_desc(d, "mtc_select_scan")) return -1;
    MTC_REQUIRE(hist && st, "mtc_select_scan: null pointer");
    hipLaunchKernelGGL(select_scan_kernel, dim3(1), dim3(1024), 0, as_stream(stream), hist, level,
                       mtc_select_levels(d) - 1, st);
    return check_launch("mtc_select_scan");
}

static int count_impl(const mtc_desc* d, int cols, const void* acts, const mtc_select_state* st, const void* threshold,
                      uint32_t* row_gt, uint32_t* row_eq, const uint32_t* pred, uint32_t want, void* stream,
                      const void* alt_acts = nullptr, int alt_cols = 0) {
    DISPATCH_DT(d->dtype, select_count_kernel, dim3((d->B + 3) / 4), dim3(256), 0, as_stream(stream), acts, d->B, cols,
                const_cast<mtc_select_state*>(st), threshold, row_gt, row_eq, pred, want, alt_acts, alt_cols);
    return check_launch("mtc_select_count");
}
extern "C" int mtc_select_count(const mtc_desc* d, const void* acts, const mtc_select_state* st, const void* threshold,
                                uint32_t* row_gt, uint32_t* row_eq, void* stream) {
    if (check_desc(d, "mtc_select_count")) return -1;
    MTC_REQUIRE(acts && st && row_gt && row_eq, "mtc_select_count: null pointer");
    return count_impl(d, d->D, acts, st, threshold, row_gt, row_eq, nullptr, 0, stream);
}

static int offsets_impl(const mtc_desc* d, const uint32_t* row_gt, const uint32_t* row_eq, const uint32_t* tie_before,
                        mtc_select_state* st, int32_t* row_ptr, uint32_t* tie_take, uint32_t* carry, void* stream) {
    hipLaunchKernelGGL(select_offsets_kernel, dim3(1), dim3(1024), 0, as_stream(stream), d->B, row_gt, row_eq,
                       tie_before, st, row_ptr, tie_take, carry);
    return check_launch("mtc_select_offsets");
}
extern "C" int mtc_select_offsets(const mtc_desc* d, const uint32_t* row_gt, const uint32_t* row_eq,
                                  const uint32_t* tie_before, mtc_select_state* st, int32_t* row_ptr,
                                  uint32_t* tie_take, void* stream) {
    if (check_desc(d, "mtc_select_offsets")) return -1;
    MTC_REQUIRE(row_gt && row_eq && st && row_ptr && tie_take, "mtc_select_offsets: null pointer");
    return offsets_impl(d, row_gt, row_eq, tie_before, st, row_ptr, tie_take, nullptr, stream);
}

static int fill_impl(const mtc_desc* d, int cols, const void* acts, const mtc_select_state* st, const void* threshold,
                     const int32_t* row_ptr, const uint32_t* tie_take, int32_t* col_idx, void* vals, const int32_t* col_map,
                     const uint32_t* pred, uint32_t want, void* stream, const void* alt_acts = nullptr, int alt_cols = 0) {
    DISPATCH_DT(d->dtype, select_fill_kernel, dim3((d->B + 3) / 4), dim3(256), 0, as_stream(stream), acts, d->B, cols,
                st, threshold, row_ptr, tie_take, col_idx, vals, col_map, pred, want, alt_acts, alt_cols);
    return check_launch("mtc_select_fill");
}
extern "C" int mtc_select_fill(const mtc_desc* d, const void* acts, const mtc_select_state* st, const void* threshold,
                               const int32_t* row_ptr, const uint32_t* tie_take, int32_t* col_idx, void* vals,
                               void* stream) {
    if (check_desc(d, "mtc_select_fill")) return -1;
    MTC_REQUIRE(acts && st && row_ptr && tie_take && col_idx && vals, "mtc_select_fill: null pointer");
    return fill_impl(d, d->D, acts, st, threshold, row_ptr, tie_take, col_idx, vals, nullptr, nullptr, 0, stream);
}

static int rowsub_impl(const mtc_desc* d, int cols, const void* acts, mtc_select_state* st, uint32_t* row_above,
                       uint32_t* row_sub, const uint32_t* pred, uint32_t want, void* stream, const void* alt_acts = nullptr,
                       int alt_cols = 0) {
    hipLaunchKernelGGL(select_rowsub_bf16_kernel, dim3((d->B + 3) / 4), dim3(256), 0, as_stream(stream), acts, d->B, cols, st,
                       row_above, row_sub, pred, want, alt_acts, alt_cols);
    return check_launch("mtc_select_rowsub");
}
extern "C" int mtc_select_rowsub(const mtc_desc* d, const void* acts, mtc_select_state* st, uint32_t* row_above,
                                 uint32_t* row_sub, void* stream) {
    if (check_desc(d, "mtc_select_rowsub")) return -1;
    MTC_REQUIRE(d->dtype == MTC_BF16, "mtc_select_rowsub: bf16 only (fp32 keeps 19 key bits after level 0)");
    MTC_REQUIRE(acts && st && row_above && row_sub, "mtc_select_rowsub: null pointer");
    return rowsub_impl(d, d->D, acts, st, row_above, row_sub, nullptr, 0, stream);
}

extern "C" int mtc_select_finish_rows(const mtc_desc* d, mtc_select_state* st, const uint32_t* row_above,
                                      const uint32_t* row_sub, const uint32_t* global_sub, uint32_t* row_gt,
                                      uint32_t* row_eq, void* stream) {
    if (check_desc(d, "mtc_select_finish_rows")) return -1;
    MTC_REQUIRE(st && row_above && row_sub && row_gt && row_eq, "mtc_select_finish_rows: null pointer");
    hipLaunchKernelGGL(select_finish_bf16_kernel, dim3(1), dim3(1024), 0, as_stream(stream), d->B, st, row_above, row_sub,
                       global_sub, row_gt, row_eq);
    return check_launch("mtc_select_finish_rows");
}

extern "C" size_t mtc_select_workspace_bytes(const mtc_desc* d) {
    if (!d) return 0;
    return 3 * HIST_BINS * sizeof(uint32_t) + 12 * (size_t)((d->B + 63) & ~63) * sizeof(uint32_t);
}

extern "C" int mtc_select_global_topk(const mtc_desc* d, const void* acts, uint32_t k_total, const void* threshold,
                                      void* workspace, size_t workspace_bytes, mtc_select_state* st, int32_t* row_ptr,
                                      int32_t* col_idx, void* vals, void* stream) {
    if (check_desc(d, "mtc_select_global_topk")) return -1;
    MTC_REQUIRE(workspace && workspace_bytes >= mtc_select_workspace_bytes(d),
                "mtc_select_global_topk: workspace too small (%zu < %zu)", workspace_bytes, mtc_select_workspace_bytes(d));
    uint32_t* hist = reinterpret_cast<uint32_t*>(workspace);
    const size_t bpad = (size_t)((d->B + 63) & ~63);
    uint32_t* row_gt = hist + 3 * HIST_BINS;
    uint32_t* row_eq = row_gt + bpad;
    uint32_t* tie_take = row_eq + bpad;
    int rc;
    if ((rc = mtc_select_init(st, k_total, stream))) return rc;
    if (!threshold && d->dtype == MTC_BF16) {
        // level-0 histogram, then one per-row pass that resolves the remaining 3 key bits
        uint32_t* row_above = tie_take + bpad;
        uint32_t* row_sub = row_above + bpad;
        if (hipMemsetAsync(hist, 0, HIST_BINS * sizeof(uint32_t), as_stream(stream)) != hipSuccess) {
            set_error("mtc_select_global_topk: memset failed");
            return -2;
        }
        if ((rc = mtc_select_hist(d, acts, 0, st, hist, stream))) return rc;
        if ((rc = mtc_select_scan(d, 0, hist, st, stream))) return rc;
        if ((rc = mtc_select_rowsub(d, acts, st, row_above, row_sub, stream))) return rc;
        if ((rc = mtc_select_finish_rows(d, st, row_above, row_sub, nullptr, row_gt, row_eq, stream))) return rc;
        if ((rc = mtc_select_offsets(d, row_gt, row_eq, nullptr, st, row_ptr, tie_take, stream))) return rc;
        return mtc_select_fill(d, acts, st, threshold, row_ptr, tie_take, col_idx, vals, stream);
    }
    if (!threshold) {
        const int levels = mtc_select_levels(d);
        if (hipMemsetAsync(hist, 0, 3 * HIST_BINS * sizeof(uint32_t), as_stream(stream)) != hipSuccess) {
            set_error("mtc_select_global_topk: memset failed");
            return -2;
        }
        for (int l = 0; l < levels; ++l) {
            if ((rc = mtc_select_hist(d, acts, l, st, hist + l * HIST_BINS, stream))) return rc;
            if ((rc = mtc_select_scan(d, l, hist + l * HIST_BINS, st, stream))) return rc;
        }
    }
    if ((rc = mtc_select_count(d, acts, st, threshold, row_gt, row_eq, stream))) return rc;
    if ((rc = mtc_select_offsets(d, row_gt, row_eq, nullptr, st, row_ptr, tie_take, stream))) return rc;
    return mtc_select_fill(d, acts, st, threshold, row_ptr, tie_take, col_idx, vals, stream);
}

static int carry_cols(const mtc_desc* d) { return d->D < 1024 ? d->D : 1024; }

extern "C" size_t mtc_select_carry_workspace_bytes(const mtc_desc* d) {
    if (!d) return 0;
    const size_t es = d->dtype == MTC_BF16 ? 2 : 4;
    const size_t cells = (size_t)d->B * carry_cols(d);
    return ((mtc_select_workspace_bytes(d) + 255) & ~(size_t)255) + ((cells * es + 255) & ~(size_t)255) +
           ((cells * sizeof(int32_t) + 255) & ~(size_t)255) + (((size_t)d->B * sizeof(uint32_t) + 255) & ~(size_t)255) +
           (size_t)d->B * CAND_WIN * sizeof(unsigned short);
}

extern "C" int mtc_select_global_topk_carry(const mtc_desc* d, const void* acts, uint32_t k_total, void* workspace,
                                            size_t workspace_bytes, uint32_t* carry, mtc_select_state* st,
                                            int32_t* row_ptr, int32_t* col_idx, void* vals, void* stream) {
    return mtc_select_global_topk_carry_ema(d, acts, k_total, workspace, workspace_bytes, carry, st, row_ptr, col_idx, vals,
                                            nullptr, 0.f, stream);
}

extern "C" int mtc_select_global_topk_carry_ema(const mtc_desc* d, const void* acts, uint32_t k_total, void* workspace,
                                                size_t workspace_bytes, uint32_t* carry, mtc_select_state* st,
                                                int32_t* row_ptr, int32_t* col_idx, void* vals, void* threshold,
                                                float ema_lr, void* stream) {
    return mtc_select_global_topk_carry_marked(d, acts, k_total, workspace, workspace_bytes, carry, st, row_ptr, col_idx, vals,
                                               threshold, ema_lr, nullptr, stream);
}

extern "C" int mtc_select_global_topk_carry_marked(const mtc_desc* d, const void* acts, uint32_t k_total, void* workspace,
                                                   size_t workspace_bytes, uint32_t* carry, mtc_select_state* st,
                                                   int32_t* row_ptr, int32_t* col_idx, void* vals, void* threshold,
                                                   float ema_lr, const void* marks, void* stream) {
    if (check_desc(d, "mtc_select_global_topk_carry")) return -1;
    MTC_REQUIRE(acts && workspace && carry && st && row_ptr && col_idx && vals, "mtc_select_global_topk_carry: null pointer");
    MTC_REQUIRE(workspace_bytes >= mtc_select_carry_workspace_bytes(d),
                "mtc_select_global_topk_carry: workspace too small (%zu < %zu)", workspace_bytes,
                mtc_select_carry_workspace_bytes(d));
    hipStream_t s = as_stream(stream);
    const int C = carry_cols(d);
    const size_t es = d->dtype == MTC_BF16 ? 2 : 4;
    char* ws = reinterpret_cast<char*>(workspace);
    uint32_t* hist = reinterpret_cast<uint32_t*>(ws);
    const size_t bpad = (size_t)((d->B + 63) & ~63);
    uint32_t* row_gt = hist + 3 * HIST_BINS;
    uint32_t* row_eq = row_gt + bpad;
    uint32_t* tie_take = row_eq + bpad;
    uint32_t* row_above = tie_take + bpad;
    uint32_t* row_sub = row_above + bpad;
    char* cand_vals = ws + ((mtc_select_workspace_bytes(d) + 255) & ~(size_t)255);
    int32_t* cand_cols = reinterpret_cast<int32_t*>(cand_vals + (((size_t)d->B * C * es + 255) & ~(size_t)255));
    uint32_t* cand_cnt = reinterpret_cast<uint32_t*>(reinterpret_cast<char*>(cand_cols) +
                                                     (((size_t)d->B * C * sizeof(int32_t) + 255) & ~(size_t)255));
    unsigned short* row_win = reinterpret_cast<unsigned short*>(reinterpret_cast<char*>(cand_cnt) +
                                                                 (((size_t)d->B * sizeof(uint32_t) + 255) & ~(size_t)255));
    const uint32_t* flag = carry + 1;
    int rc;
    if (d->dtype == MTC_BF16) {
        // candidate pass (+ per-row window counts) -> ONE single-workgroup launch that finds tau, the quota, the CSR offsets,
        // the next bound and the threshold EMA from those counts -> ordered fill from the compact matrix.  The two
        // launches in between are the fallback (bound failed / tau beyond the window): full-size passes that leave at once
        // when carry[1] == 1; the fill reads the compact matrix or acts accordingly.  Nothing is read back.
        hipLaunchKernelGGL(select_candidates_kernel<MTC_BF16>, dim3((d->B + 3) / 4), dim3(256), 0, s, acts, d->B, d->D, C,
                           (const uint32_t*)carry, (void*)cand_vals, cand_cols, cand_cnt,
                           reinterpret_cast<const unsigned char*>(marks), row_win);
        if (d->B <= 4096)
            hipLaunchKernelGGL((select_window_tail_kernel<MTC_BF16, true>), dim3(1), dim3(1024), 0, s, d->B, C, k_total, st,
                               carry, (const uint32_t*)cand_cnt, (const unsigned short*)row_win, hist, row_gt, row_eq, row_ptr,
                               tie_take, threshold, ema_lr);
        else
            hipLaunchKernelGGL((select_window_tail_kernel<MTC_BF16, false>), dim3(1), dim3(1024), 0, s, d->B, C, k_total, st,
                               carry, (const uint32_t*)cand_cnt, (const unsigned short*)row_win, hist, row_gt, row_eq, row_ptr,
                               tie_take, threshold, ema_lr);
        if ((rc = check_launch("mtc_select_global_topk_carry"))) return rc;
        if ((rc = hist_impl(d, d->D, acts, 0, st, hist, flag, 0, stream))) return rc;
        hipLaunchKernelGGL(select_fallback_rows_kernel, dim3((d->B + 15) / 16), dim3(1024), 0, s, acts, d->B, d->D, st,
                           (const uint32_t*)hist, row_above, row_sub, row_gt, row_eq, row_ptr, tie_take, carry, threshold, ema_lr,
                           flag, hist + HIST_BINS /* ticket: a word of the unused level-1 histogram, zeroed with it */);
        if ((rc = check_launch("mtc_select_global_topk_carry"))) return rc;
        return fill_impl(d, C, cand_vals, st, nullptr, row_ptr, tie_take, col_idx, vals, cand_cols, flag, 1, stream, acts, d->D);
    }
    // one pass over acts with the carried bound, then the verdict (flag) for the launches below
    hipLaunchKernelGGL(select_candidates_kernel<MTC_F32>, dim3((d->B + 3) / 4), dim3(256), 0, s, acts, d->B, d->D, C,
                       (const uint32_t*)carry, (void*)cand_vals, cand_cols, cand_cnt,
                       reinterpret_cast<const unsigned char*>(marks), (unsigned short*)nullptr);
    hipLaunchKernelGGL(select_carry_decide_kernel, dim3(1), dim3(1024), 0, s, carry, cand_cnt, d->B, C, k_total, st, hist);
    if ((rc = check_launch("mtc_select_global_topk_carry"))) return rc;
    // each streaming pass is ONE launch that reads the compact matrix if flag == 1 and acts if flag == 0 (grid sized for
    // the larger); the scans between them work on the shared histogram / per-row arrays whichever matrix filled them
    // (fp32: 32-bit keys, three histogram levels on whichever matrix the flag names)
    const int levels = mtc_select_levels(d);
    for (int l = 0; l < levels; ++l) {
        if ((rc = hist_impl(d, C, cand_vals, l, st, hist + l * HIST_BINS, flag, 1, stream, acts, d->D))) return rc;
        if ((rc = mtc_select_scan(d, l, hist + l * HIST_BINS, st, stream))) return rc;
    }
    if ((rc = count_impl(d, C, cand_vals, st, nullptr, row_gt, row_eq, flag, 1, stream, acts, d->D))) return rc;
    // one launch: offsets + next bound + [threshold EMA].  With a tau the smallest selected value is tau,
    // noted by the offsets piece; without one (fewer positives than k) the count pass noted it -- either way
    // st->min_sel_bits is final before the fill, which does not touch it in TopK mode
    hipLaunchKernelGGL(select_tail_kernel<MTC_F32>, dim3(1), dim3(1024), 0, s, d->B, st, 0,
                (const uint32_t*)row_above, (const uint32_t*)row_sub, row_gt, row_eq, row_ptr, tie_take, carry, threshold, ema_lr,
                (const uint32_t*)nullptr, 0u);
    if ((rc = check_launch("mtc_select_global_topk_carry"))) return rc;
    return fill_impl(d, C, cand_vals, st, nullptr, row_ptr, tie_take, col_idx, vals, cand_cols, flag, 1, stream, acts, d->D);
}

extern "C" int mtc_select_offsets_sharded(const mtc_desc* d, const uint32_t* row_gt, const uint32_t* all_eq, int32_t eq_stride,
                                          int32_t world, int32_t rank, mtc_select_state* st, int32_t* row_ptr,
                                          uint32_t* tie_take, int64_t* minmax_out, void* stream) {
    if (check_desc(d, "mtc_select_offsets_sharded")) return -1;
    MTC_REQUIRE(row_gt && all_eq && st && row_ptr && tie_take, "mtc_select_offsets_sharded: null pointer");
    MTC_REQUIRE(world >= 1 && rank >= 0 && rank < world, "mtc_select_offsets_sharded: bad rank %d of %d", rank, world);
    MTC_REQUIRE(eq_stride >= d->B, "mtc_select_offsets_sharded: eq_stride %d < B %d", eq_stride, d->B);
    hipLaunchKernelGGL(select_offsets_sharded_kernel, dim3(1), dim3(1024), 0, as_stream(stream), d->B, row_gt, all_eq, eq_stride,
                       world, rank, st, row_ptr, tie_take, reinterpret_cast<long long*>(minmax_out));
    return check_launch("mtc_select_offsets_sharded");
}

// ---- feature-sharded select in four phases (collectives X1 / X2 / X3 of the caller between them; see the kernels)
namespace {
struct ShardWs {
    uint32_t *row_gt, *tie_take, *cand_cnt;
    char* cand_vals;
    int32_t* cand_cols;
    int C;
};
size_t shard_ws_layout(const mtc_desc* d, char* base, ShardWs* out) {
    const size_t es = d->dtype == MTC_BF16 ? 2 : 4;
    const size_t bpad = ((size_t)d->B + 63) & ~(size_t)63;
    const int C = carry_cols(d);
    size_t off = 0;
    auto take = [&](size_t bytes) { const size_t at = off; off += (bytes + 255) & ~(size_t)255; return at; };
    const size_t o_gt = take(bpad * 4), o_tt = take(bpad * 4), o_cnt = take(bpad * 4);
    const size_t o_cv = take((size_t)d->B * C * es), o_cc = take((size_t)d->B * C * 4);
    if (out) {
        out->row_gt = reinterpret_cast<uint32_t*>(base + o_gt);
        out->tie_take = reinterpret_cast<uint32_t*>(base + o_tt);
        out->cand_cnt = reinterpret_cast<uint32_t*>(base + o_cnt);
        out->cand_vals = base + o_cv;
        out->cand_cols = reinterpret_cast<int32_t*>(base + o_cc);
        out->C = C;
    }
    return off;
}
GroupBounds group_bounds(const mtc_desc* d) {
    GroupBounds gb;
    gb.G = d->G;
    for (int i = 0; i < MTC_MAX_PREFIX; ++i) gb.prefix[i] = i < d->G ? d->prefix[i] : 0x7fffffff;
    return gb;
}
}  // namespace

extern "C" size_t mtc_select_sharded_workspace_bytes(const mtc_desc* d) { return d ? shard_ws_layout(d, nullptr, nullptr) : 0; }
extern "C" size_t mtc_select_sharded_xbuf_words(void) { return XBUF_WORDS; }
extern "C" size_t mtc_select_sharded_full_bins(void) { return FULL_BINS; }

extern "C" int mtc_select_sharded_begin(const mtc_desc* d, const void* acts, uint32_t k_total, void* workspace,
                                        size_t workspace_bytes, uint32_t* carry, mtc_select_state* st, uint32_t* xbuf,
                                        uint32_t* hist_full, uint32_t* pack, const void* marks, void* stream) {
    if (check_desc(d, "mtc_select_sharded_begin")) return -1;
    MTC_REQUIRE(acts && workspace && st && xbuf && pack, "mtc_select_sharded_begin: null pointer");
    MTC_REQUIRE(workspace_bytes >= mtc_select_sharded_workspace_bytes(d), "mtc_select_sharded_begin: workspace too small (%zu < %zu)",
                workspace_bytes, mtc_select_sharded_workspace_bytes(d));
    MTC_REQUIRE(d->dtype != MTC_BF16 || (carry && hist_full), "mtc_select_sharded_begin: bf16 needs carry and hist_full");
    hipStream_t s = as_stream(stream);
    ShardWs w;
    shard_ws_layout(d, reinterpret_cast<char*>(workspace), &w);
    hipLaunchKernelGGL(shard_select_init_kernel, dim3(1), dim3(1024), 0, s, st, k_total, xbuf,
                       d->dtype == MTC_BF16 ? hist_full : nullptr, pack + (size_t)d->B * d->G);
    if (d->dtype == MTC_BF16) {
        // one pass over the activations with the carried bound, then the window histogram of what it kept
        hipLaunchKernelGGL(select_candidates_kernel<MTC_BF16>, dim3((d->B + 3) / 4), dim3(256), 0, s, acts, d->B, d->D, w.C,
                           (const uint32_t*)carry, (void*)w.cand_vals, w.cand_cols, w.cand_cnt,
                           reinterpret_cast<const unsigned char*>(marks));
        hipLaunchKernelGGL(shard_window_hist_kernel, dim3(std::min(256, (d->B + 3) / 4)), dim3(256), 0, s,
                           (const u16*)w.cand_vals, (const uint32_t*)w.cand_cnt, d->B, w.C, (const uint32_t*)carry, xbuf);
    }
    return check_launch("mtc_select_sharded_begin");
}

extern "C" int mtc_select_sharded_window(const mtc_desc* d, const void* acts, uint32_t k_total, uint32_t* carry,
                                         mtc_select_state* st, const uint32_t* xbuf, uint32_t* hist_full, void* stream) {
    if (check_desc(d, "mtc_select_sharded_window")) return -1;
    MTC_REQUIRE(d->dtype == MTC_BF16, "mtc_select_sharded_window: bf16 only (fp32 runs the three histogram levels)");
    MTC_REQUIRE(acts && carry && st && xbuf && hist_full, "mtc_select_sharded_window: null pointer");
    hipStream_t s = as_stream(stream);
    hipLaunchKernelGGL(shard_window_scan_kernel, dim3(1), dim3(1024), 0, s, xbuf, k_total, carry, st);
    static bool attr_done[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = -1;
    if (dev < 0 || !attr_done[dev]) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(shard_full_hist_bf16_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                FULL_BINS * 4) == hipSuccess && dev >= 0) attr_done[dev] = true;
    }
    const size_t n_chunks = (size_t)d->B * d->D / 8;
    const int grid = (int)std::max<size_t>(1, std::min<size_t>((n_chunks + 256 * 64 - 1) / (256 * 64), 256));
    hipLaunchKernelGGL(shard_full_hist_bf16_kernel, dim3(grid), dim3(256), FULL_BINS * 4, s, (const u16*)acts, n_chunks,
                       (const uint32_t*)(carry + 1), hist_full);
    return check_launch("mtc_select_sharded_window");
}

extern "C" int mtc_select_sharded_counts(const mtc_desc* d, const void* acts, uint32_t k_total, void* workspace,
                                         const uint32_t* carry, mtc_select_state* st, const uint32_t* hist_full,
                                         uint32_t* pack, void* stream) {
    if (check_desc(d, "mtc_select_sharded_counts")) return -1;
    MTC_REQUIRE(acts && workspace && st && pack, "mtc_select_sharded_counts: null pointer");
    MTC_REQUIRE(d->dtype != MTC_BF16 || (carry && hist_full), "mtc_select_sharded_counts: bf16 needs carry and hist_full");
    hipStream_t s = as_stream(stream);
    ShardWs w;
    shard_ws_layout(d, reinterpret_cast<char*>(workspace), &w);
    const uint32_t* flag = d->dtype == MTC_BF16 ? carry + 1 : nullptr;
    if (flag) hipLaunchKernelGGL(shard_full_scan_kernel, dim3(1), dim3(1024), 0, s, hist_full, k_total, flag, st);
    DISPATCH_DT(d->dtype, shard_group_counts_kernel, dim3((d->B + 3) / 4), dim3(256), 0, s, (const void*)w.cand_vals,
                (const int32_t*)w.cand_cols, w.C, acts, d->B, d->D, group_bounds(d), (const mtc_select_state*)st, flag, w.row_gt, pack);
    return check_launch("mtc_select_sharded_counts");
}

extern "C" int mtc_select_sharded_finish(const mtc_desc* d, const void* acts, void* workspace, uint32_t* carry,
                                         mtc_select_state* st, const uint32_t* all_pack, int32_t world, int32_t rank,
                                         int32_t* row_ptr, int32_t* col_idx, void* vals, void* threshold, float ema_lr,
                                         void* stream) {
    if (check_desc(d, "mtc_select_sharded_finish")) return -1;
    MTC_REQUIRE(acts && workspace && st && all_pack && row_ptr && col_idx && vals, "mtc_select_sharded_finish: null pointer");
    MTC_REQUIRE(world >= 1 && rank >= 0 && rank < world, "mtc_select_sharded_finish: bad rank %d of %d", rank, world);
    MTC_REQUIRE(d->dtype != MTC_BF16 || carry, "mtc_select_sharded_finish: bf16 needs carry");
    hipStream_t s = as_stream(stream);
    ShardWs w;
    shard_ws_layout(d, reinterpret_cast<char*>(workspace), &w);
    uint32_t* cr = d->dtype == MTC_BF16 ? carry : nullptr;
    DISPATCH_DT(d->dtype, shard_offsets_groups_kernel, dim3(1), dim3(1024), 0, s, d->B, d->G, (const uint32_t*)w.row_gt, all_pack,
                world, rank, st, row_ptr, w.tie_take, cr, threshold, ema_lr);
    int rc;
    if ((rc = check_launch("mtc_select_sharded_finish"))) return rc;
    if (cr) return fill_impl(d, w.C, w.cand_vals, st, nullptr, row_ptr, w.tie_take, col_idx, vals, w.cand_cols, cr + 1, 1, stream, acts, d->D);
    return fill_impl(d, d->D, acts, st, nullptr, row_ptr, w.tie_take, col_idx, vals, nullptr, nullptr, 0, stream);
}

extern "C" int mtc_select_pack_minmax(const mtc_select_state* st, int64_t* out2, void* stream) {
    MTC_REQUIRE(st && out2, "mtc_select_pack_minmax: null pointer");
    hipLaunchKernelGGL(select_pack_minmax_kernel, dim3(1), dim3(64), 0, as_stream(stream), st, reinterpret_cast<long long*>(out2));
    return check_launch("mtc_select_pack_minmax");
}

extern "C" int mtc_threshold_ema_global(const mtc_desc* d, const int64_t* minmax2, void* threshold, float lr, void* stream) {
    if (check_desc(d, "mtc_threshold_ema_global")) return -1;
    MTC_REQUIRE(minmax2 && threshold, "mtc_threshold_ema_global: null pointer");
    DISPATCH_DT(d->dtype, threshold_ema_global_kernel, dim3(1), dim3(64), 0, as_stream(stream),
                reinterpret_cast<const long long*>(minmax2), threshold, lr);
    return check_launch("mtc_threshold_ema_global");
}

extern "C" int mtc_shard_nnz_abs(const mtc_desc* d, const float* abs_rows, const mtc_select_state* st, float* out2, void* stream) {
    if (check_desc(d, "mtc_shard_nnz_abs")) return -1;
    MTC_REQUIRE(abs_rows && st && out2, "mtc_shard_nnz_abs: null pointer");
    hipLaunchKernelGGL(nnz_abs_kernel, dim3(1), dim3(1024), 0, as_stream(stream), d->B, abs_rows, st, out2);
    return check_launch("mtc_shard_nnz_abs");
}

extern "C" int mtc_threshold_ema(const mtc_desc* d, const mtc_select_state* st, void* threshold, float lr, void* stream) {
    if (check_desc(d, "mtc_threshold_ema")) return -1;
    MTC_REQUIRE(st && threshold, "mtc_threshold_ema: null pointer");
    DISPATCH_DT(d->dtype, threshold_ema_kernel, dim3(1), dim3(64), 0, as_stream(stream), st, threshold, lr);
    return check_launch("mtc_threshold_ema");
}

extern "C" int mtc_densify(const mtc_desc* d, const int32_t* row_ptr, const int32_t* col_idx, const void* vals,
                           int32_t max_col, void* dense, void* stream) {
    if (check_desc(d, "mtc_densify")) return -1;
    MTC_REQUIRE(row_ptr && col_idx && vals && dense, "mtc_densify: null pointer");
    DISPATCH_DT(d->dtype, densify_kernel, dim3(d->B), dim3(256), 0, as_stream(stream), row_ptr, col_idx, vals, d->B, d->D,
                max_col, dense);
    return check_launch("mtc_densify");
}
