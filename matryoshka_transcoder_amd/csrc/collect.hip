// Top-activating-sample tables on the device (reference: src/utils/activation_samples.py:83-183).
//
// The reference keeps one Python heap per feature and pushes every activation above a threshold into it, one
// Python object per activation, after copying the dense [batch, seq, D] tensor to the host.  Here each feature
// owns a fixed table of M slots in HBM (values sorted descending + slot ids; per slot the sequence position, the
// token context window and the token before it).  One wave per feature merges the feature's entries of the batch
// (read through the CSC view of the sparse activations, i.e. in (feature, row) order = the reference's arrival
// order) into its table: rank-by-counting over table + up to 64 new entries in LDS, survivors keep or take a
// slot, and only the survivors' token windows are written.  Ties: an older sample beats a newer one of the same
// value (the heap replaces only on strictly greater, activation_samples.py:181-183).
#include <math.h>
#include "common.hpp"

namespace mtc {

constexpr int COLLECT_MAX_M = 256;                     // table entries per feature
constexpr int COLLECT_CHUNK = 64;                      // new entries merged per round (one per lane)
constexpr int COLLECT_PER_LANE = (COLLECT_MAX_M + COLLECT_CHUNK) / 64;

template <int DT>
__global__ __launch_bounds__(64) void collect_top_samples_kernel(
    int D, int seq_len, const int32_t* __restrict__ col_ptr, const int32_t* __restrict__ csc_row,
    const int32_t* __restrict__ csc_pos, const void* __restrict__ vals, const int64_t* __restrict__ tokens,
    const uint8_t* __restrict__ track, float thr, int M, int half, int CW, float* __restrict__ top_val,
    int32_t* __restrict__ top_slot, int32_t* __restrict__ slot_meta, int32_t* __restrict__ slot_ctx,
    unsigned long long* __restrict__ feat_count, float* __restrict__ feat_max) {
    typedef DType<DT> D_;
    __shared__ float s_val[COLLECT_MAX_M + COLLECT_CHUNK];
    __shared__ int s_slot[COLLECT_MAX_M + COLLECT_CHUNK];     // >= 0: owns that slot; -1: new, no slot yet
    __shared__ int s_src[COLLECT_MAX_M + COLLECT_CHUNK];      // >= 0: flat row whose tokens must still be written
    __shared__ int s_free[COLLECT_CHUNK];
    __shared__ int s_nfree, s_nnew;

    const int f = blockIdx.x;
    const int lane = threadIdx.x;
    if (f >= D || (track && !track[f])) return;
    const int c0 = col_ptr[f], c1 = col_ptr[f + 1];
    if (c0 == c1) return;

    for (int i = lane; i < M; i += 64) {
        s_val[i] = top_val[(size_t)f * M + i];
        s_slot[i] = top_slot[(size_t)f * M + i];
        s_src[i] = -1;
    }
    unsigned long long seen = 0;
    float mx = -INFINITY;
    bool changed = false;

    for (int base = c0; base < c1; base += COLLECT_CHUNK) {
        const int e = base + lane;
        float v = -INFINITY;
        int n = 0;
        if (e < c1) {
            v = D_::ld(reinterpret_cast<const typename D_::T*>(vals) + csc_pos[e]);
            n = csc_row[e];
        }
        const bool pass = e < c1 && v > thr;
        const unsigned long long mask = __ballot(pass);
        const int np = __popcll(mask);
        seen += (unsigned long long)np;
        mx = fmaxf(mx, pass ? v : -INFINITY);
        if (np == 0) continue;                                  // wave-uniform
        if (lane == 0) { s_nfree = 0; s_nnew = 0; }
        if (pass) {
            const int k = M + __popcll(mask & ((1ull << lane) - 1ull));
            s_val[k] = v;
            s_slot[k] = -1;
            s_src[k] = n;
        }
        __syncthreads();
        const int total = M + np;
        // rank = number of entries that come before this one: larger value, or equal value and earlier arrival
        float e_val[COLLECT_PER_LANE];
        int e_slot[COLLECT_PER_LANE], e_src[COLLECT_PER_LANE], e_rank[COLLECT_PER_LANE];
#pragma unroll
        for (int q = 0; q < COLLECT_PER_LANE; ++q) {
            const int i = lane + 64 * q;
            e_rank[q] = total;                                   // "not an entry"
            if (i < total) {
                const float vi = s_val[i];
                int r = 0;
                for (int j = 0; j < total; ++j) {
                    const float vj = s_val[j];
                    r += (vj > vi || (vj == vi && j < i)) ? 1 : 0;
                }
                e_val[q] = vi; e_slot[q] = s_slot[i]; e_src[q] = s_src[i]; e_rank[q] = r;
            }
        }
        __syncthreads();
        // entries pushed out release their slots; new survivors queue up for one (the two counts are equal:
        // exactly M slots exist and exactly M entries survive)
        int want[COLLECT_PER_LANE];
#pragma unroll
        for (int q = 0; q < COLLECT_PER_LANE; ++q) {
            want[q] = -1;
            if (lane + 64 * q < total) {
                if (e_rank[q] >= M && e_slot[q] >= 0) s_free[atomicAdd(&s_nfree, 1)] = e_slot[q];
                if (e_rank[q] < M && e_slot[q] < 0) want[q] = atomicAdd(&s_nnew, 1);
            }
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < COLLECT_PER_LANE; ++q) {
            if (lane + 64 * q < total && e_rank[q] < M) {
                if (want[q] >= 0) { e_slot[q] = s_free[want[q]]; changed = true; }
                s_val[e_rank[q]] = e_val[q];
                s_slot[e_rank[q]] = e_slot[q];
                s_src[e_rank[q]] = e_src[q];
            }
        }
        __syncthreads();
    }

    // statistics (activation_samples.py:128-134)
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    if (lane == 0 && seen) {
        feat_count[f] += seen;
        feat_max[f] = fmaxf(feat_max[f], mx);
    }
    if (!__any(changed)) return;
    for (int i = lane; i < M; i += 64) {
        top_val[(size_t)f * M + i] = s_val[i];
        top_slot[(size_t)f * M + i] = s_slot[i];
        const int n = s_src[i];
        if (n < 0) continue;
        // token window of a new survivor (activation_samples.py:143-147)
        const int b = n / seq_len, s = n - b * seq_len;
        const int start = max(0, s - half), end = min(seq_len, s + half + 1);
        const size_t slot = (size_t)f * M + s_slot[i];
        const int64_t* row = tokens + (size_t)b * seq_len;
        slot_meta[slot * 4 + 0] = s;
        slot_meta[slot * 4 + 1] = end - start;
        slot_meta[slot * 4 + 2] = s > 0 ? (int32_t)row[s - 1] : -1;
        slot_meta[slot * 4 + 3] = n;
        for (int t = 0; t < end - start; ++t) slot_ctx[slot * CW + t] = (int32_t)row[start + t];
    }
}

}  // namespace mtc

using namespace mtc;

extern "C" int mtc_collect_top_samples(int32_t dtype, int32_t n_rows, int32_t D, int32_t seq_len, const int32_t* col_ptr,
                                       const int32_t* csc_row, const int32_t* csc_pos, const void* vals,
                                       const int64_t* tokens, const uint8_t* track, float storage_threshold,
                                       int32_t max_samples, int32_t context_size, float* top_val, int32_t* top_slot,
                                       int32_t* slot_meta, int32_t* slot_ctx, uint64_t* feat_count, float* feat_max,
                                       void* stream) {
    MTC_REQUIRE(dtype == MTC_F32 || dtype == MTC_BF16, "mtc_collect_top_samples: bad dtype %d", dtype);
    MTC_REQUIRE(n_rows > 0 && D > 0 && seq_len > 0, "mtc_collect_top_samples: n_rows, D, seq_len must be positive");
    MTC_REQUIRE(n_rows % seq_len == 0, "mtc_collect_top_samples: %d activation rows are not a whole number of "
                "sequences of %d tokens", n_rows, seq_len);
    MTC_REQUIRE(max_samples >= 1 && max_samples <= COLLECT_MAX_M,
                "mtc_collect_top_samples: max_samples_per_feature must be in 1..%d (got %d)", COLLECT_MAX_M, max_samples);
    MTC_REQUIRE(context_size >= 0 && context_size <= 4096, "mtc_collect_top_samples: context_size must be in 0..4096");
    MTC_REQUIRE(col_ptr && csc_row && csc_pos && vals && tokens && top_val && top_slot && slot_meta && slot_ctx &&
                feat_count && feat_max, "mtc_collect_top_samples: null pointer");
    const int half = context_size / 2, CW = 2 * half + 1;
    hipStream_t s = as_stream(stream);
    unsigned long long* cnt = reinterpret_cast<unsigned long long*>(feat_count);
    if (dtype == MTC_BF16)
        collect_top_samples_kernel<MTC_BF16><<<D, 64, 0, s>>>(D, seq_len, col_ptr, csc_row, csc_pos, vals, tokens, track,
                                                              storage_threshold, max_samples, half, CW, top_val, top_slot,
                                                              slot_meta, slot_ctx, cnt, feat_max);
    else
        collect_top_samples_kernel<MTC_F32><<<D, 64, 0, s>>>(D, seq_len, col_ptr, csc_row, csc_pos, vals, tokens, track,
                                                             storage_threshold, max_samples, half, CW, top_val, top_slot,
                                                             slot_meta, slot_ctx, cnt, feat_max);
    return check_launch("mtc_collect_top_samples");
}
