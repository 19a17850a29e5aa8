// K4: k-sparse backward (autograd of sae.py:173-245 as driven by training.py:172) and the
// feature-major index it needs; dead-feature counters (sae.py:165-169).
//
// The selected entries are re-indexed by FEATURE (CSC).  One wavefront per feature j then builds
//     dW_dec[j]  = sum_b a_bj  * g_b        (gathers rows of g,      [B][A_t] model dtype)
//     dW_encT[j] = sum_b da_bj * x_b        (gathers rows of x_norm, [B][A_s] model dtype)
// in registers and writes each gradient row exactly once -- no float atomics, rows of silent
// features are written as zeros, so the dense gradient the optimizer expects costs no extra pass.
// The gathered operands (g: B*A_t*s B, x: B*A_s*s B) are small enough to live in L2 / Infinity
// Cache; algorithmic bytes = nnz*(A_t + A_s)*s gathered + D*(A_t + A_s)*s written.
#include "common.hpp"
#include "rowops.hpp"

namespace mtc {

// ---------------------------------------------------------------- CSC build
// wave per row: count entries per feature
__global__ __launch_bounds__(256) void csc_count_kernel(int B, const int32_t* __restrict__ row_ptr,
                                                        const int32_t* __restrict__ col_idx, uint32_t* __restrict__ cnt) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= B) return;
    for (int e = row_ptr[row] + lane; e < row_ptr[row + 1]; e += 64) atomicAdd(&cnt[col_idx[e]], 1u);
}

// exclusive scan over the D per-feature counts in two small launches (tiles of 4096 features):
//   A: per-tile totals;  B: each tile adds the totals of the tiles before it, scans itself, writes
//   col_ptr / cursor, updates the dead counters and lists the features with long segments.
constexpr int SCAN_TILE = 4096;
__global__ __launch_bounds__(1024) void csc_tile_sum_kernel(int D, const uint32_t* __restrict__ cnt,
                                                            uint32_t* __restrict__ tile_sum) {
    __shared__ uint32_t sm[16];
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const int j0 = blockIdx.x * SCAN_TILE + t * 4;
    uint32_t v = 0;
    if (j0 + 3 < D) { const u32x4 q = *reinterpret_cast<const u32x4*>(cnt + j0); v = q[0] + q[1] + q[2] + q[3]; }
    else for (int i = 0; i < 4; ++i) if (j0 + i < D) v += cnt[j0 + i];
    v = wave_sum_u32(v);
    if (lane == 0) sm[w] = v;
    __syncthreads();
    if (t == 0) { uint32_t a = 0; for (int i = 0; i < 16; ++i) a += sm[i]; tile_sum[blockIdx.x] = a; }
}

__global__ __launch_bounds__(1024) void csc_scan_kernel(int D, int n_tiles, const uint32_t* __restrict__ cnt,
                                                        const uint32_t* __restrict__ tile_sum, int32_t* __restrict__ col_ptr,
                                                        uint32_t* __restrict__ cursor, float* __restrict__ dead_cnt,
                                                        int32_t* __restrict__ long_list, uint32_t* __restrict__ n_long) {
    __shared__ uint32_t sm[16];
    __shared__ uint32_t s_base;
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    if (w == 0) {
        uint32_t a = 0;
        for (int i = lane; i < (int)blockIdx.x; i += 64) a += tile_sum[i];
        a = wave_sum_u32(a);
        if (lane == 0) s_base = a;
    }
    const int j0 = blockIdx.x * SCAN_TILE + t * 4;
    uint32_t c[4] = {0, 0, 0, 0};
    if (j0 + 3 < D) { const u32x4 q = *reinterpret_cast<const u32x4*>(cnt + j0); c[0] = q[0]; c[1] = q[1]; c[2] = q[2]; c[3] = q[3]; }
    else for (int i = 0; i < 4; ++i) if (j0 + i < D) c[i] = cnt[j0 + i];
    const uint32_t mine = c[0] + c[1] + c[2] + c[3];
    uint32_t inc = mine;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const uint32_t n = __shfl_up(inc, o, 64); if (lane >= o) inc += n; }
    if (lane == 63) sm[w] = inc;
    __syncthreads();
    uint32_t base = s_base + inc - mine;
    for (int i = 0; i < w; ++i) base += sm[i];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int j = j0 + i;
        if (j < D) {
            col_ptr[j] = (int32_t)base;
            cursor[j] = base;
            if (dead_cnt) dead_cnt[j] = c[i] ? 0.f : dead_cnt[j] + 1.f;
            if (c[i] > 64) long_list[atomicAdd(n_long, 1u)] = j;
            base += c[i];
        }
    }
    if (blockIdx.x == n_tiles - 1 && t == 1023) {
        uint32_t total = s_base;
        for (int i = 0; i < 16; ++i) total += sm[i];
        col_ptr[D] = (int32_t)total;
    }
}

__global__ __launch_bounds__(256) void csc_fill_kernel(int B, const int32_t* __restrict__ row_ptr,
                                                       const int32_t* __restrict__ col_idx, uint32_t* __restrict__ cursor,
                                                       int32_t* __restrict__ csc_row, int32_t* __restrict__ csc_pos) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= B) return;
    for (int e = row_ptr[row] + lane; e < row_ptr[row + 1]; e += 64) {
        const uint32_t pos = atomicAdd(&cursor[col_idx[e]], 1u);
        csc_row[pos] = row;
        csc_pos[pos] = e;
    }
}

// Entries of one feature arrive in atomic order; sort each segment by batch row so that the
// fp32 accumulation order (and therefore every gradient bit) is reproducible run to run.
// One wave per feature; segments <= 64 are sorted by rank counting in registers.
__global__ __launch_bounds__(256) void csc_sort_kernel(int D, const int32_t* __restrict__ col_ptr,
                                                       int32_t* __restrict__ csc_row, int32_t* __restrict__ csc_pos) {
    const int lane = threadIdx.x & 63;
    const int j = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
    if (j >= D) return;
    const int s = col_ptr[j], n = col_ptr[j + 1] - s;
    if (n <= 1) return;
    if (n <= 64) {
        const int r = lane < n ? csc_row[s + lane] : 0x7fffffff;
        const int p = lane < n ? csc_pos[s + lane] : 0;
        int rank = 0;
        for (int i = 0; i < n; ++i) rank += (__builtin_amdgcn_readlane(r, i) < r) ? 1 : 0;   // rows are unique per feature
        if (lane < n) { csc_row[s + rank] = r; csc_pos[s + rank] = p; }
        return;
    }
    // longer segments: csc_sort_long_kernel
}

// Long segments (> 64 entries, listed by the scan): one workgroup per such feature scatters its entries into a
// per-row slot array (slot[r] = csc_pos for row r, or -1) and sweeps it in row order.  B <= 65536.
__global__ __launch_bounds__(256) void csc_sort_long_kernel(int B, const int32_t* __restrict__ col_ptr,
                                                            const int32_t* __restrict__ long_list,
                                                            const uint32_t* __restrict__ n_long,
                                                            int32_t* __restrict__ csc_row, int32_t* __restrict__ csc_pos,
                                                            int32_t* __restrict__ slot_ws /* gridDim.x * B ints */) {
    __shared__ uint32_t wsum[4];
    int32_t* slot = slot_ws + (size_t)blockIdx.x * B;
    const int nl = (int)*n_long;
    for (int li = blockIdx.x; li < nl; li += gridDim.x) {
        const int j = long_list[li];
        const int s = col_ptr[j], n = col_ptr[j + 1] - s;
        for (int r = threadIdx.x; r < B; r += 256) slot[r] = -1;
        __syncthreads();
        for (int i = threadIdx.x; i < n; i += 256) slot[csc_row[s + i]] = csc_pos[s + i];
        __syncthreads();
        // ordered compaction of the occupied rows
        uint32_t out = 0;
        for (int r0 = 0; r0 < B; r0 += 256) {
            const int r = r0 + threadIdx.x;
            const int v = r < B ? slot[r] : -1;
            const uint64_t bal = __ballot(v >= 0);
            const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
            if (lane == 0) wsum[w] = (uint32_t)__popcll(bal);
            __syncthreads();
            uint32_t pre = 0, tot = 0;
            for (int i = 0; i < 4; ++i) { if (i < w) pre += wsum[i]; tot += wsum[i]; }
            if (v >= 0) {
                const uint32_t pos = out + pre + (uint32_t)__popcll(bal & ((1ull << lane) - 1ull));
                csc_row[s + pos] = r;
                csc_pos[s + pos] = v;
            }
            out += tot;
            __syncthreads();
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------- gradient rows
// out[j][:] = scale * sum_{e in feature j} coef[csc_pos[e]] * src[csc_row[e]][:]
// DT: model dtype of src and out; COEF_DT: dtype of coef (vals are model dtype, d_vals fp32).
// bias_out (optional) = scale * sum coef.
template <int DT, int COEF_DT, int NC>
__global__ __launch_bounds__(256, (NC <= 5 ? 4 : 2)) void grad_rows_kernel(
    int D, int A, const int32_t* __restrict__ col_ptr, const int32_t* __restrict__ csc_row,
    const int32_t* __restrict__ csc_pos, const void* __restrict__ coef, const void* __restrict__ src,
    const float* __restrict__ scale_p, void* __restrict__ out, void* __restrict__ bias_out, int accumulate) {
    typedef DType<DT> O_;
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
        for (int k = 0; k < 8; ++k) acc[c][k] = 0.f;
    float csum = 0.f;
    const int s = col_ptr[j], e_end = col_ptr[j + 1];
    for (int e0 = s; e0 < e_end; e0 += 64) {
        const int nb = min(64, e_end - e0);
        int my_row = 0; float my_c = 0.f;
        if (lane < nb) {
            my_row = csc_row[e0 + lane];
            const int pos = csc_pos[e0 + lane];
            my_c = DType<COEF_DT>::ld(reinterpret_cast<const typename DType<COEF_DT>::T*>(coef) + pos);
        }
        csum += my_c;
        accumulate_run<DT, NC>(rc, src, A, my_row, my_c, 0, nb, acc);
    }
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        if (rc.valid(c)) {
            const size_t off = (size_t)j * A + rc.off(c);
            float o[8];
            if (accumulate) {
                load8<DT>(out, off, o);
#pragma unroll
                for (int k = 0; k < 8; ++k) o[k] = O_::rnd(o[k] + scale * acc[c][k]);
            } else {
#pragma unroll
                for (int k = 0; k < 8; ++k) o[k] = scale * acc[c][k];
            }
            store8<DT>(out, off, o);
        }
    }
    if (bias_out) {
        csum = wave_sum(csum);
        if (lane == 0) {
            typename O_::T* bp = reinterpret_cast<typename O_::T*>(bias_out) + j;
            O_::st(bp, accumulate ? O_::ld(bp) + scale * csum : scale * csum);
        }
    }
}

// ---------------------------------------------------------------- db_dec = sum_b g_b
// block = 4 waves x 64 lanes: lane owns one 8-column chunk (16-B loads), the waves split the slab's
// rows (4 rows in flight per lane), LDS folds the 4 waves; partial[slab][A] in fp32
constexpr int COLSUM_SLABS = 64;
template <int DT>
__global__ __launch_bounds__(256) void colsum_partial_kernel(int B, int A, const void* __restrict__ g, int rows_per_slab,
                                                             float* __restrict__ partial) {
    __shared__ float sm[4][64][8];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int ch = blockIdx.x * 64 + lane;                 // chunk of 8 columns
    const int slab = blockIdx.y;
    const bool ok = ch < (A >> 3);
    const int r0 = slab * rows_per_slab, r1 = min(B, r0 + rows_per_slab);
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (ok) {
        for (int r = r0 + w; r < r1; r += 16) {
            Raw<DT> q[4];
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (r + 4 * u < r1) q[u] = load_raw<DT>(g, (size_t)(r + 4 * u) * A + (size_t)ch * 8);
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (r + 4 * u < r1) {
                    float v[8];
                    unpack(q[u], v);
#pragma unroll
                    for (int k = 0; k < 8; ++k) acc[k] += v[k];
                }
        }
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) sm[w][lane][k] = acc[k];
    __syncthreads();
    if (w == 0 && ok) {
        float o[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) o[k] = (sm[0][lane][k] + sm[1][lane][k]) + (sm[2][lane][k] + sm[3][lane][k]);
        store8<MTC_F32>(partial, (size_t)slab * A + (size_t)ch * 8, o);
    }
}
template <int DT>
__global__ __launch_bounds__(256) void colsum_finish_kernel(int A, int slabs, const float* __restrict__ partial,
                                                            const float* __restrict__ scale_p, void* __restrict__ out,
                                                            int accumulate) {
    typedef DType<DT> D_;
    const int col = blockIdx.x * 256 + threadIdx.x;
    if (col >= A) return;
    float s = 0.f;
    for (int i = 0; i < slabs; ++i) s += partial[(size_t)i * A + col];
    s *= scale_p ? *scale_p : 1.f;
    typename D_::T* p = reinterpret_cast<typename D_::T*>(out) + col;
    D_::st(p, accumulate ? D_::ld(p) + s : s);
}

// ---------------------------------------------------------------- scratch of mtc_colsum_static
// The one entry point that allocates nothing (include/mtc.h): its per-slab partial sums live in this file-scope
// array instead of a caller workspace.  Every GPU the library runs on holds its own copy (2 MiB) until the process
// ends.  In host code the array's NAME is only a stand-in for the symbol, not a device address: the address is
// looked up once per device with hipGetSymbolAddress and only that pointer is ever handed to a kernel.
constexpr int COLSUM_STATIC_MAX_A = 8192;
static __device__ __align__(256) float colsum_scratch[COLSUM_SLABS * COLSUM_STATIC_MAX_A];

}  // namespace mtc

using namespace mtc;

// workgroups of the long-segment sort: each needs a B-int slot array in the workspace; up to 64 MiB of them
// (1024 groups at B = 4096).  With the auxiliary loss active every dead feature is a long segment
// (B * k_aux / n_dead entries), thousands per step.
static int long_sort_blocks(const mtc_desc* d) {
    const size_t fit = ((size_t)64 << 20) / ((size_t)d->B * sizeof(int32_t));
    return (int)(fit < 64 ? 64 : fit > 1024 ? 1024 : fit);
}

extern "C" size_t mtc_csc_workspace_bytes(const mtc_desc* d) {
    if (!d) return 0;
    // counts[D] + cursor[D] + long_list[D] + tile sums / counter (pad) + long-segment row slots
    return 3 * (size_t)d->D * sizeof(uint32_t) + 4096 + (size_t)long_sort_blocks(d) * d->B * sizeof(int32_t);
}

// part 1: counts -> column pointers, cursors, dead counters (everything the dead-feature list needs); part 2: fill + per-column
// order.  mtc_build_csc runs both; a caller that wants the dead list early enqueues it between the two (same workspace).
extern "C" int mtc_build_csc_part(const mtc_desc* d, const int32_t* row_ptr, const int32_t* col_idx, void* workspace,
                                  size_t workspace_bytes, int32_t* col_ptr, int32_t* csc_row, int32_t* csc_pos, float* cnt,
                                  int32_t part, void* stream) {
    if (check_desc(d, "mtc_build_csc")) return -1;
    MTC_REQUIRE(part == 1 || part == 2, "mtc_build_csc_part: part must be 1 or 2");
    MTC_REQUIRE(row_ptr && col_idx && workspace && col_ptr && csc_row && csc_pos, "mtc_build_csc: null pointer");
    MTC_REQUIRE(workspace_bytes >= mtc_csc_workspace_bytes(d), "mtc_build_csc: workspace too small");
    MTC_REQUIRE(d->B <= 65536, "mtc_build_csc: B > 65536 unsupported");
    const int n_tiles = (d->D + SCAN_TILE - 1) / SCAN_TILE;
    MTC_REQUIRE(n_tiles <= 1000, "mtc_build_csc: D too large");
    // layout: [n_long (1) | tile_sum (<=1000) | pad to 1024 words] counts[D] cursor[D] long_list[D] slots
    uint32_t* n_long = reinterpret_cast<uint32_t*>(workspace);
    uint32_t* tile_sum = n_long + 1;
    uint32_t* counts = n_long + 1024;
    uint32_t* cursor = counts + d->D;
    int32_t* long_list = reinterpret_cast<int32_t*>(cursor + d->D);
    int32_t* slots = long_list + d->D;
    hipStream_t s = as_stream(stream);
    const dim3 rgrid((d->B + 3) / 4), block(256);
    if (part == 1) {
        if (hipMemsetAsync(workspace, 0, (1024 + (size_t)d->D) * sizeof(uint32_t), s) != hipSuccess) {
            set_error("mtc_build_csc: memset failed");
            return -2;
        }
        hipLaunchKernelGGL(csc_count_kernel, rgrid, block, 0, s, d->B, row_ptr, col_idx, counts);
        hipLaunchKernelGGL(csc_tile_sum_kernel, dim3(n_tiles), dim3(1024), 0, s, d->D, counts, tile_sum);
        hipLaunchKernelGGL(csc_scan_kernel, dim3(n_tiles), dim3(1024), 0, s, d->D, n_tiles, counts, tile_sum, col_ptr, cursor, cnt,
                           long_list, n_long);
    } else {
        hipLaunchKernelGGL(csc_fill_kernel, rgrid, block, 0, s, d->B, row_ptr, col_idx, cursor, csc_row, csc_pos);
        hipLaunchKernelGGL(csc_sort_kernel, dim3((d->D + 3) / 4), block, 0, s, d->D, col_ptr, csc_row, csc_pos);
        hipLaunchKernelGGL(csc_sort_long_kernel, dim3(long_sort_blocks(d)), block, 0, s, d->B, col_ptr, long_list, n_long, csc_row,
                           csc_pos, slots);
    }
    return check_launch("mtc_build_csc");
}

extern "C" int mtc_build_csc(const mtc_desc* d, const int32_t* row_ptr, const int32_t* col_idx, void* workspace,
                             size_t workspace_bytes, int32_t* col_ptr, int32_t* csc_row, int32_t* csc_pos, float* cnt,
                             void* stream) {
    const int rc = mtc_build_csc_part(d, row_ptr, col_idx, workspace, workspace_bytes, col_ptr, csc_row, csc_pos, cnt, 1, stream);
    return rc ? rc : mtc_build_csc_part(d, row_ptr, col_idx, workspace, workspace_bytes, col_ptr, csc_row, csc_pos, cnt, 2, stream);
}

#define LAUNCH_GRAD(DTV, COEF, ncv)                                                                                  \
    do {                                                                                                             \
        const int nc_ = (ncv);                                                                                       \
        if (nc_ <= 1) hipLaunchKernelGGL((grad_rows_kernel<DTV, COEF, 1>), grid, block, 0, s, D, A, col_ptr, csc_row, csc_pos, coef, src, scale, out, bias_out, accumulate); \
        else if (nc_ <= 2) hipLaunchKernelGGL((grad_rows_kernel<DTV, COEF, 2>), grid, block, 0, s, D, A, col_ptr, csc_row, csc_pos, coef, src, scale, out, bias_out, accumulate); \
        else if (nc_ <= 3) hipLaunchKernelGGL((grad_rows_kernel<DTV, COEF, 3>), grid, block, 0, s, D, A, col_ptr, csc_row, csc_pos, coef, src, scale, out, bias_out, accumulate); \
        else if (nc_ <= 5) hipLaunchKernelGGL((grad_rows_kernel<DTV, COEF, 5>), grid, block, 0, s, D, A, col_ptr, csc_row, csc_pos, coef, src, scale, out, bias_out, accumulate); \
        else if (nc_ <= 7) hipLaunchKernelGGL((grad_rows_kernel<DTV, COEF, 7>), grid, block, 0, s, D, A, col_ptr, csc_row, csc_pos, coef, src, scale, out, bias_out, accumulate); \
        else if (nc_ <= 9) hipLaunchKernelGGL((grad_rows_kernel<DTV, COEF, 9>), grid, block, 0, s, D, A, col_ptr, csc_row, csc_pos, coef, src, scale, out, bias_out, accumulate); \
        else hipLaunchKernelGGL((grad_rows_kernel<DTV, COEF, 16>), grid, block, 0, s, D, A, col_ptr, csc_row, csc_pos, coef, src, scale, out, bias_out, accumulate); \
    } while (0)

static void launch_grad_rows(int dtype, bool coef_is_vals, int D, int A, const int32_t* col_ptr, const int32_t* csc_row,
                             const int32_t* csc_pos, const void* coef, const void* src, const float* scale, void* out,
                             void* bias_out, int accumulate, hipStream_t s) {
    const dim3 grid((D + 3) / 4), block(256);
    const int nc = nc_for(A);
    if (dtype == MTC_BF16) {
        if (coef_is_vals) LAUNCH_GRAD(MTC_BF16, MTC_BF16, nc);     // dW_dec: src = g, coef = vals (model dtype)
        else LAUNCH_GRAD(MTC_BF16, MTC_F32, nc);                   // dW_encT: src = x_norm, coef = d_vals (fp32)
    } else {
        LAUNCH_GRAD(MTC_F32, MTC_F32, nc);
    }
}

extern "C" int mtc_backward_sparse(const mtc_desc* d, const int32_t* col_ptr, const int32_t* csc_row,
                                   const int32_t* csc_pos, const void* vals, const float* d_vals, const void* g,
                                   const void* x_norm, const float* scale, void* dW_dec, void* dW_encT, void* db_enc,
                                   int accumulate, void* stream) {
    if (check_desc(d, "mtc_backward_sparse")) return -1;
    MTC_REQUIRE(col_ptr && csc_row && csc_pos && vals && d_vals && g && x_norm && dW_dec && dW_encT && db_enc,
                "mtc_backward_sparse: null pointer");
    hipStream_t s = as_stream(stream);
    launch_grad_rows(d->dtype, true, d->D, d->A_t, col_ptr, csc_row, csc_pos, vals, g, scale, dW_dec, nullptr, accumulate, s);
    launch_grad_rows(d->dtype, false, d->D, d->A_s, col_ptr, csc_row, csc_pos, d_vals, x_norm, scale, dW_encT, db_enc,
                     accumulate, s);
    return check_launch("mtc_backward_sparse");
}

extern "C" size_t mtc_colsum_workspace_bytes(const mtc_desc* d) { return d ? COLSUM_SLABS * (size_t)d->A_t * sizeof(float) : 0; }

static void colsum_launch(const mtc_desc* d, const void* g, const float* scale, float* workspace, void* db_dec, int accumulate,
                          hipStream_t s);

extern "C" int mtc_colsum(const mtc_desc* d, const void* g, const float* scale, void* workspace, size_t workspace_bytes,
                          void* db_dec, int accumulate, void* stream) {
    if (check_desc(d, "mtc_colsum")) return -1;
    MTC_REQUIRE(g && workspace && db_dec, "mtc_colsum: null pointer");
    MTC_REQUIRE(workspace_bytes >= mtc_colsum_workspace_bytes(d), "mtc_colsum: workspace too small");
    colsum_launch(d, g, scale, reinterpret_cast<float*>(workspace), db_dec, accumulate, as_stream(stream));
    return check_launch("mtc_colsum");
}

// Same kernels, scratch in the library's static device array: no hipMalloc / hipFree / allocating call on this path.
// The per-device address of the array is looked up on the first call on each device (with that device current).
// Every partial the finish kernel reads is written by the partial kernel of the same call, so nothing has to be
// cleared between calls; calls that overlap on one GPU (two streams) share the array and race.
constexpr int STATIC_MAX_DEV = 64;
static float* g_colsum_scratch_ptr[STATIC_MAX_DEV];
extern "C" int mtc_colsum_static(const mtc_desc* d, const void* g, const float* scale, void* db_dec, int accumulate,
                                 void* stream) {
    if (check_desc(d, "mtc_colsum_static")) return -1;
    MTC_REQUIRE(g && db_dec, "mtc_colsum_static: null pointer");
    MTC_REQUIRE(d->A_t <= COLSUM_STATIC_MAX_A,
                "mtc_colsum_static: A_t = %d exceeds the static scratch (at most %d columns); use mtc_colsum with a workspace",
                d->A_t, COLSUM_STATIC_MAX_A);
    int dev = -1;
    MTC_REQUIRE(hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < STATIC_MAX_DEV, "mtc_colsum_static: no current device");
    if (!g_colsum_scratch_ptr[dev]) {
        float* ptr = nullptr;
        const hipError_t e = hipGetSymbolAddress(reinterpret_cast<void**>(&ptr), HIP_SYMBOL(colsum_scratch));
        MTC_REQUIRE(e == hipSuccess && ptr, "mtc_colsum_static: hipGetSymbolAddress failed (%s)", hipGetErrorString(e));
        g_colsum_scratch_ptr[dev] = ptr;
    }
    colsum_launch(d, g, scale, g_colsum_scratch_ptr[dev], db_dec, accumulate, as_stream(stream));
    return check_launch("mtc_colsum_static");
}

static void colsum_launch(const mtc_desc* d, const void* g, const float* scale, float* workspace, void* db_dec, int accumulate,
                          hipStream_t s) {
    const int slabs = COLSUM_SLABS, rows_per = (d->B + slabs - 1) / slabs;
    if (d->dtype == MTC_BF16)
        hipLaunchKernelGGL(colsum_partial_kernel<MTC_BF16>, dim3((d->A_t / 8 + 63) / 64, slabs), dim3(256), 0, s, d->B, d->A_t, g,
                           rows_per, workspace);
    else
        hipLaunchKernelGGL(colsum_partial_kernel<MTC_F32>, dim3((d->A_t / 8 + 63) / 64, slabs), dim3(256), 0, s, d->B, d->A_t, g,
                           rows_per, workspace);
    if (d->dtype == MTC_BF16)
        hipLaunchKernelGGL(colsum_finish_kernel<MTC_BF16>, dim3((d->A_t + 255) / 256), dim3(256), 0, s, d->A_t, slabs,
                           reinterpret_cast<const float*>(workspace), scale, db_dec, accumulate);
    else
        hipLaunchKernelGGL(colsum_finish_kernel<MTC_F32>, dim3((d->A_t + 255) / 256), dim3(256), 0, s, d->A_t, slabs,
                           reinterpret_cast<const float*>(workspace), scale, db_dec, accumulate);
}
