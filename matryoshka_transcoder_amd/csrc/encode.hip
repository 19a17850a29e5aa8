// K0 row normalisation (sae.py:139-147) and K1 encoder contraction + bias + ReLU (sae.py:173-174).
//
// K1 (bf16): acts[b][f] = relu(rnd(rnd(sum_k x[b][k] * W_encT[f][k]) + b_enc[f])).
// Both operands are K-contiguous, so fragments are plain 16-byte LDS reads.  The MFMA A operand
// is the WEIGHT tile and the B operand the BATCH tile: the 16x16 accumulator then holds, per
// lane, 4 consecutive features of one batch row, which pack into one 8-byte store.
//   tile 128 (batch) x 128 (features) x 64 (k), 4 waves (2x2), each wave 64x64 = 4x4 MFMA tiles
//   of v_mfma_f32_16x16x32_bf16; operands staged global -> LDS with 16-byte LDS-DMA
//   (global_load_lds_dwordx4), double buffered; LDS rows are 128 B with the 16-byte chunk index
//   XOR-ed by ((row >> 1) & 7) -- applied on the per-lane SOURCE address (the LDS image of an
//   LDS-DMA is lane-linear) and again on the ds_read_b128 address, which makes every 16-lane
//   read group hit 16 distinct 16-byte slots (conflict free).
//   Block ids are remapped so that each XCD walks a contiguous range of 8x8 super-tiles
//   (private 4 MiB L2 per XCD).
#include <stdlib.h>
#include "common.hpp"

namespace mtc {

// ------------------------------------------------------------------------------------------- K0
template <int DT>
__global__ __launch_bounds__(256) void row_normalize_kernel(const void* __restrict__ x, void* __restrict__ xn,
                                                            void* __restrict__ mean_out, void* __restrict__ std_out,
                                                            int B, int A) {
    typedef DType<DT> D_;
    typedef typename D_::T T;
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= B) return;
    const T* xr = reinterpret_cast<const T*>(x) + (size_t)row * A;
    T* out = reinterpret_cast<T*>(xn) + (size_t)row * A;
    const int nchunk = A >> 3;
    float s = 0.f;
    for (int c = lane; c < nchunk; c += 64) {
        float v[8];
        load8<DT>(xr, (size_t)c * 8, v);
#pragma unroll
        for (int i = 0; i < 8; ++i) s += v[i];
    }
    const float mean = D_::rnd(wave_sum(s) / (float)A);
    // centred values are rounded to the model dtype before the std, as the eager reference does
    float sc = 0.f;
    for (int c = lane; c < nchunk; c += 64) {
        float v[8];
        load8<DT>(xr, (size_t)c * 8, v);
#pragma unroll
        for (int i = 0; i < 8; ++i) sc += D_::rnd(v[i] - mean);
    }
    const float mean_c = wave_sum(sc) / (float)A;
    float ss = 0.f;
    for (int c = lane; c < nchunk; c += 64) {
        float v[8];
        load8<DT>(xr, (size_t)c * 8, v);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float d = D_::rnd(v[i] - mean) - mean_c;
            ss += d * d;
        }
    }
    const float stdv = D_::rnd(sqrtf(wave_sum(ss) / (float)(A - 1)));
    const float denom = D_::rnd(stdv + 1e-5f);
    for (int c = lane; c < nchunk; c += 64) {
        float v[8], o[8];
        load8<DT>(xr, (size_t)c * 8, v);
#pragma unroll
        for (int i = 0; i < 8; ++i) o[i] = D_::rnd(D_::rnd(v[i] - mean) / denom);
        store8<DT>(out, (size_t)c * 8, o);
    }
    if (lane == 0) {
        D_::st(reinterpret_cast<T*>(mean_out) + row, mean);
        D_::st(reinterpret_cast<T*>(std_out) + row, stdv);
    }
}

// ------------------------------------------------------------------------------------------- K1 bf16 MFMA
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;

constexpr int BM = 128;   // batch rows per block
constexpr int BN = 128;   // features per block
constexpr int BK = 64;    // k per stage (128-byte LDS rows)
constexpr int TILE_BYTES = BM * BK * 2;   // 16 KiB per operand per stage

__device__ __forceinline__ void xcd_tile_coords(int bid, int nblk, int mt, int nt, int& tm, int& tn) {
    // blocks b, b+8, ... share an XCD: give each XCD a contiguous run of logical ids (bijective form)
    const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7;
    const int lid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    // logical ids walk GROUP-wide groups of batch tiles under each feature tile
    constexpr int GROUP = 4;     // measured at config #2: 4 and 2 beat 8 and 16 by 3-5 % (smaller per-XCD working set per K step)
    const int per_group = GROUP * nt;
    const int g = lid / per_group;
    const int first_m = g * GROUP;
    const int gsz = min(mt - first_m, GROUP);
    const int in = lid - g * per_group;
    tm = first_m + in % gsz;
    tn = in / gsz;
}

// Epilogue floor: 0 for the training encoder (plain ReLU); max(threshold, 0) when the eval-mode threshold
// (sae.py:189-191: where(relu(pre) > threshold, ., 0)) is fused in.  A NaN threshold keeps nothing, like the
// reference's comparison.
template <int DT>
__device__ __forceinline__ float epilogue_floor(const void* thr) {
    if (!thr) return 0.f;
    const float t = DType<DT>::ld(reinterpret_cast<const typename DType<DT>::T*>(thr));
    return t != t ? t : fmaxf(t, 0.f);
}

// ReLU / threshold that never emits -0.0 (the select reads bit patterns), packed to 4 bf16
__device__ __forceinline__ uint2 relu_pack4(float p0, float p1, float p2, float p3, float fl) {
    p0 = p0 > fl ? p0 : 0.f; p1 = p1 > fl ? p1 : 0.f; p2 = p2 > fl ? p2 : 0.f; p3 = p3 > fl ? p3 : 0.f;
    uint2 o;
    o.x = (__float_as_uint(p0) >> 16) | (__float_as_uint(p1) & 0xffff0000u);
    o.y = (__float_as_uint(p2) >> 16) | (__float_as_uint(p3) & 0xffff0000u);
    return o;
}

__device__ __forceinline__ void glds16(const void* gsrc, void* lds_dst) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_dst, 16, 0, 0);
}
// the same with a cache-policy operand (2 = non-temporal): experiments on the persistent kernel's operand streams
template <int AUX>
__device__ __forceinline__ void glds16_aux(const void* gsrc, void* lds_dst) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_dst, 16, 0, AUX);
}
#ifndef MTC_ENC_GROUP
#define MTC_ENC_GROUP 4
#endif
#ifndef MTC_ENC_W_AUX
#define MTC_ENC_W_AUX 0
#endif
#ifndef MTC_ENC_X_AUX
#define MTC_ENC_X_AUX 0
#endif
#ifndef MTC_ENC_P0_ORDER
#define MTC_ENC_P0_ORDER 0               // experiment: 1 = phase 0 issues its W staging before its 12 LDS reads, 2 = between the k-halves
#endif

// EPI 0: the encoder (round, + bias, round, ReLU / threshold; lda = ldb = K, ldc = N).  EPI 1 / 2: plain C = A . Bt^T
// (mtc_gemm_nt: the auxiliary branch's products) rounded once to bf16 / kept in fp32, with row strides lda / ldb / ldc
// and an optional row gather on A (a_rows: A's row m is A[a_rows[m]]).
template <int EPI>
__global__ __launch_bounds__(256, 2) void encode_bf16_mfma_kernel(const u16* __restrict__ X, const u16* __restrict__ Wt,
                                                                  const u16* __restrict__ bias, void* __restrict__ acts_,
                                                                  int M, int N, int K, int mt, int nt, const void* __restrict__ thr,
                                                                  long long lda, long long ldb, long long ldc,
                                                                  const int32_t* __restrict__ a_rows) {
    u16* acts = reinterpret_cast<u16*>(acts_);
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // layout: [stage][operand(0 = X, 1 = W)][128 rows][128 B]
    int tm, tn;
    xcd_tile_coords(blockIdx.x, mt * nt, mt, nt, tm, tn);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wb = wid & 1, wf = wid >> 1;          // wave's 64-row batch half / 64-feature half
    const int row0 = tm * BM, col0 = tn * BN;

    // ---- staging addresses: wave `wid` moves pieces wid*4 .. wid*4+3 (8 rows x 128 B each) of both tiles
    const int srow = lane >> 3;                      // row inside a piece
    const u16* xsrc[4];
    const u16* wsrc[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const int r = (wid * 4 + p) * 8 + srow;      // row inside the tile
        const int chunk = (lane & 7) ^ ((r >> 1) & 7);
        int xr = min(row0 + r, M - 1);                                    // clamp: tails read valid rows
        const int wr = min(col0 + r, N - 1);
        if (EPI != 0 && a_rows) xr = a_rows[xr];
        xsrc[p] = X + (size_t)xr * (EPI == 0 ? (long long)K : lda) + chunk * 8;
        wsrc[p] = Wt + (size_t)wr * (EPI == 0 ? (long long)K : ldb) + chunk * 8;
    }
    auto stage = [&](int buf, int kt) {
        char* base = smem + buf * 2 * TILE_BYTES;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            glds16(xsrc[p] + kt * BK, base + (wid * 4 + p) * 1024);
            glds16(wsrc[p] + kt * BK, base + TILE_BYTES + (wid * 4 + p) * 1024);
        }
    };

    // ---- fragment read offsets (bytes inside an operand tile)
    const int fr = lane & 15, fq = lane >> 4;
    const int swz = (fr >> 1) & 7;
    int aoff[2];                                     // per k-step: chunk (ks*4 + fq) ^ swz
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) aoff[ks] = fr * 128 + (((ks * 4 + fq) ^ swz) << 4);

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int ktiles = K / BK;
    stage(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int kt = 0; kt < ktiles; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < ktiles) stage(cur ^ 1, kt + 1);
        const char* xs = smem + cur * 2 * TILE_BYTES + wb * 64 * 128;
        const char* ws = smem + cur * 2 * TILE_BYTES + TILE_BYTES + wf * 64 * 128;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 wfr[4], xfr[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                wfr[i] = *reinterpret_cast<const bf16x8*>(ws + i * 16 * 128 + aoff[ks]);
                xfr[i] = *reinterpret_cast<const bf16x8*>(xs + i * 16 * 128 + aoff[ks]);
            }
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wfr[i], xfr[j], acc[i][j], 0, 0, 0);
            __builtin_amdgcn_s_setprio(0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }

    // ---- epilogue: lane holds features f = fbase + i*16 + fq*4 + r (r = 0..3) of batch row b = bbase + j*16 + fr
    const int fbase = col0 + wf * 64 + fq * 4;
    const int bbase = row0 + wb * 64 + fr;
    if (EPI != 0) {                               // plain product (N % 4 == 0, ldc % 4 == 0: checked on the host)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int f = fbase + i * 16;
            if (f >= N) continue;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int b = bbase + j * 16;
                if (b >= M) continue;
                const f32x4 v = acc[i][j];
                if (EPI == 2) {
                    *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(acts_) + (size_t)b * ldc + f) = v;
                } else {
                    uint2 o;
                    o.x = pack2_bf16(v[0], v[1]); o.y = pack2_bf16(v[2], v[3]);
                    *reinterpret_cast<uint2*>(acts + (size_t)b * ldc + f) = o;
                }
            }
        }
        return;
    }
    const float fl = epilogue_floor<MTC_BF16>(thr);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int f = fbase + i * 16;
        if (f >= N) continue;
        const uint2 bb = *reinterpret_cast<const uint2*>(bias + f);     // N % 8 == 0 -> in bounds, aligned
        const float b0 = bf16_bits_to_f32(bb.x & 0xffffu), b1 = bf16_bits_to_f32(bb.x >> 16);
        const float b2 = bf16_bits_to_f32(bb.y & 0xffffu), b3 = bf16_bits_to_f32(bb.y >> 16);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int b = bbase + j * 16;
            if (b >= M) continue;
            const f32x4 v = acc[i][j];
            *reinterpret_cast<uint2*>(acts + (size_t)b * N + f) =
                relu_pack4(round_bf16(round_bf16(v[0]) + b0), round_bf16(round_bf16(v[1]) + b1),
                           round_bf16(round_bf16(v[2]) + b2), round_bf16(round_bf16(v[3]) + b3), fl);
        }
    }
}

// ------------------------------------------------------------------------------------------- K1 fp32 MFMA
// fp32 model dtype (the reference's default, config.py:72): the same 128 x 128 tile, LDS image and
// staging as the bf16 kernel above with 32 floats per 128-byte LDS row.  v_mfma_f32_16x16x4_f32 is an
// exact fp32 fma chain.  One ds_read_b128 gives a lane 4 consecutive k of its row; MFMA number c of a
// group consumes component c, so the four lane groups (lane >> 4) feed k = 16 g + 4 q + c -- a
// permutation of k that is the same for both operands, which is all a dot product needs.
constexpr int BK32 = 32;
template <int EPI>
__global__ __launch_bounds__(256, 2) void encode_f32_mfma_kernel(const float* __restrict__ X, const float* __restrict__ Wt,
                                                                 const float* __restrict__ bias, float* __restrict__ acts,
                                                                 int M, int N, int K, int mt, int nt, const void* __restrict__ thr,
                                                                 long long lda, long long ldb, long long ldc,
                                                                 const int32_t* __restrict__ a_rows) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int tm, tn;
    xcd_tile_coords(blockIdx.x, mt * nt, mt, nt, tm, tn);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wb = wid & 1, wf = wid >> 1;
    const int row0 = tm * BM, col0 = tn * BN;
    const int srow = lane >> 3;
    const float* xsrc[4];
    const float* wsrc[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const int r = (wid * 4 + p) * 8 + srow;
        const int chunk = (lane & 7) ^ ((r >> 1) & 7);
        int xr = min(row0 + r, M - 1);
        const int wr = min(col0 + r, N - 1);
        if (EPI != 0 && a_rows) xr = a_rows[xr];
        xsrc[p] = X + (size_t)xr * (EPI == 0 ? (long long)K : lda) + chunk * 4;
        wsrc[p] = Wt + (size_t)wr * (EPI == 0 ? (long long)K : ldb) + chunk * 4;
    }
    auto stage = [&](int buf, int kt) {
        char* base = smem + buf * 2 * TILE_BYTES;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            glds16(xsrc[p] + kt * BK32, base + (wid * 4 + p) * 1024);
            glds16(wsrc[p] + kt * BK32, base + TILE_BYTES + (wid * 4 + p) * 1024);
        }
    };
    const int fr = lane & 15, fq = lane >> 4;
    const int swz = (fr >> 1) & 7;
    int aoff[2];
#pragma unroll
    for (int g = 0; g < 2; ++g) aoff[g] = fr * 128 + (((g * 4 + fq) ^ swz) << 4);

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int ktiles = K / BK32;
    stage(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int kt = 0; kt < ktiles; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < ktiles) stage(cur ^ 1, kt + 1);
        const char* xs = smem + cur * 2 * TILE_BYTES + wb * 64 * 128;
        const char* ws = smem + cur * 2 * TILE_BYTES + TILE_BYTES + wf * 64 * 128;
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            f32x4 wfr[4], xfr[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                wfr[i] = *reinterpret_cast<const f32x4*>(ws + i * 16 * 128 + aoff[g]);
                xfr[i] = *reinterpret_cast<const f32x4*>(xs + i * 16 * 128 + aoff[g]);
            }
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(wfr[i][c], xfr[j][c], acc[i][j], 0, 0, 0);
            __builtin_amdgcn_s_setprio(0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
    // ---- epilogue: lane holds features f = fbase + i*16 + r (r = 0..3) of batch row b = bbase + j*16
    const int fbase = col0 + wf * 64 + fq * 4;
    const int bbase = row0 + wb * 64 + fr;
    const float fl = EPI == 0 ? epilogue_floor<MTC_F32>(thr) : 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int f = fbase + i * 16;
        if (f >= N) continue;
        f32x4 bb = f32x4{0.f, 0.f, 0.f, 0.f};
        if (EPI == 0) bb = *reinterpret_cast<const f32x4*>(bias + f);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int b = bbase + j * 16;
            if (b >= M) continue;
            f32x4 v = acc[i][j];
            if (EPI == 0) {
#pragma unroll
                for (int r = 0; r < 4; ++r) { const float pre = v[r] + bb[r]; v[r] = pre > fl ? pre : 0.f; }
                *reinterpret_cast<f32x4*>(acts + (size_t)b * N + f) = v;
            } else {
                *reinterpret_cast<f32x4*>(acts + (size_t)b * ldc + f) = v;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------- K1 bf16 MFMA, 256x256 tile
// Large problems (>= 2 tiles per CU): 256 (batch) x 256 (features) x 64 tiles double the FLOP per
// operand byte of the 128^2 kernel (whose 64 FLOP/B needs ~37 TB/s of L2 at MFMA peak).  One
// workgroup of 8 waves per CU (128 KiB LDS: 2 buffers x {W half 0, W half 1, X half 0, X half 1} of
// 128 rows x 128 B, same XOR swizzle as above); wave (wr, wc) owns 128 features x 64 batch rows =
// 8 x 4 accumulator tiles (128 VGPRs).  The two wave groups wr = 0 / 1 share each SIMD pairwise and
// run HALF A PHASE APART (group 1 passes one extra barrier up front): while one group issues its
// LDS fragment reads and the LDS-DMA loads of the next K tile, the other runs a cluster of 16
// MFMAs (one 64x32 quadrant x K = 64) at raised priority; every barrier swaps the roles.
//   phase 0: read W(m0) [8 x b128] + X(n0) [4]; stage W half 0 of tile kt+1;           MFMA (m0, n0)
//   phase 1: read X(n1) [4];                    stage W half 1 of tile kt+1;           MFMA (m0, n1)
//   phase 2: read W(m1) [8];                                                           MFMA (m1, n1)
//   phase 3: stage both X halves of tile kt+2, then vmcnt(4) (= all of tile kt+1 in);  MFMA (m1, n0)
// LDS-DMA loads therefore get 4-10 barrier intervals to land and the wait is counted, never a drain.
// A staged buffer is first read one barrier after every wave's wait, and restaged only after the
// barrier that follows the lgkmcnt(0) of its last reader.
constexpr int B2M = 256, B2N = 256;
constexpr int HALF_BYTES = 128 * 128;          // 16 KiB: 128 rows x 64 bf16

// MFMA intrinsics are pure at the IR level, so LLVM may sink / hoist them across s_barrier; pinning the
// fragments after the barrier and the accumulators before the next one (empty asm, "+v") keeps each
// 16-MFMA cluster between its two barriers.
template <int I0, int J0>
__device__ __forceinline__ void mfma_quadrant(bf16x8 (&wf)[2][4], bf16x8 (&xf)[2][2], f32x4 (&acc)[8][4]) {
#ifdef MTC_ENC_SPLIT_WAIT
    // fragments are pinned k-half by k-half: the MFMAs of the first half start when ITS reads have landed (counted lgkmcnt),
    // the second half's reads are still in flight underneath them
#ifndef MTC_ABLATE_NOPRIO
    __builtin_amdgcn_s_setprio(1);
#endif
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
        for (int i = 0; i < 4; ++i) asm volatile("" : "+v"(wf[ks][i]));
#pragma unroll
        for (int j = 0; j < 2; ++j) asm volatile("" : "+v"(xf[ks][j]));
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
                acc[I0 + i][J0 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[ks][i], xf[ks][j], acc[I0 + i][J0 + j], 0, 0, 0);
    }
#ifndef MTC_ABLATE_NOPRIO
    __builtin_amdgcn_s_setprio(0);
#endif
#else
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
        for (int i = 0; i < 4; ++i) asm volatile("" : "+v"(wf[ks][i]));
#pragma unroll
        for (int j = 0; j < 2; ++j) asm volatile("" : "+v"(xf[ks][j]));
    }
#ifndef MTC_ABLATE_NOPRIO
    __builtin_amdgcn_s_setprio(1);
#endif
#ifndef MTC_ABLATE_NOMFMA
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
                acc[I0 + i][J0 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[ks][i], xf[ks][j], acc[I0 + i][J0 + j], 0, 0, 0);
#endif
#ifndef MTC_ABLATE_NOPRIO
    __builtin_amdgcn_s_setprio(0);
#endif
#endif
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) asm volatile("" : "+v"(acc[I0 + i][J0 + j]));
}

#define PHASE_FENCE() do { __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0); } while (0)

__global__ __launch_bounds__(512, 1) void encode_bf16_mfma256_kernel(const u16* __restrict__ X, const u16* __restrict__ Wt,
                                                                     const u16* __restrict__ bias, u16* __restrict__ acts,
                                                                     int M, int N, int K, int mt, int nt, const void* __restrict__ thr) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int tm, tn;
    xcd_tile_coords(blockIdx.x, mt * nt, mt, nt, tm, tn);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wid >> 2, wc = wid & 3;
    const int row0 = tm * B2M, col0 = tn * B2N;

    // ---- staging: half-tile h (0,1 = W halves; 2,3 = X halves) has 16 pieces of 8 rows; wave wid moves 2 of them
    const u16* src[4][2];
#pragma unroll
    for (int h = 0; h < 4; ++h)
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int r = (wid * 2 + q) * 8 + (lane >> 3);                  // row inside the half-tile
            const int chunk = (lane & 7) ^ ((r >> 1) & 7);
            if (h < 2) src[h][q] = Wt + (size_t)min(col0 + h * 128 + r, N - 1) * K + chunk * 8;
            else src[h][q] = X + (size_t)min(row0 + (h - 2) * 128 + r, M - 1) * K + chunk * 8;
        }
    auto stage_half = [&](int buf, int h, int kt) {
        char* base = smem + buf * 4 * HALF_BYTES + h * HALF_BYTES + wid * 2 * 1024;
        glds16(src[h][0] + kt * BK, base);
        glds16(src[h][1] + kt * BK, base + 1024);
    };

    const int fr = lane & 15, fq = lane >> 4;
    const int swz = (fr >> 1) & 7;
    int off[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) off[ks] = fr * 128 + (((ks * 4 + fq) ^ swz) << 4);
    const int w_base = wr * HALF_BYTES;                                          // this wave's W half-tile
    const int x_base = (2 + (wc >> 1)) * HALF_BYTES + (wc & 1) * 64 * 128;     // its 64 batch rows

    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    bf16x8 wf[2][4], xa[2][2], xb[2][2];

    const int ktiles = K / BK;
    // prologue: tile 0 entirely, plus the X halves of tile 1 (the staging schedule below runs that far ahead)
#pragma unroll
    for (int h = 0; h < 4; ++h) stage_half(0, h, 0);
    if (ktiles > 1) {
        stage_half(1, 2, 1); stage_half(1, 3, 1);
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    PHASE_FENCE();
    if (wr == 1) PHASE_FENCE();                                                  // stagger the second wave group

    // Staging schedule (buffer = tile parity).  A half-tile may be overwritten once the barrier after the
    // lgkmcnt(0) of its last reader has passed: X halves are last read in phase 1, W halves in phase 2.
    //   tile kt, phase 0: W half 0 of tile kt+1      phase 1: W half 1 of tile kt+1
    //            phase 3: X halves of tile kt+2, then vmcnt(4): everything older (all of tile kt+1) has landed
    for (int kt = 0; kt < ktiles; ++kt) {
        const char* cur = smem + (kt & 1) * 4 * HALF_BYTES;
        const int nb = (kt + 1) & 1;
#ifdef MTC_ABLATE_NOGLDS
        const bool more = false, more2 = false;
#else
        const bool more = kt + 1 < ktiles, more2 = kt + 2 < ktiles;
#endif
        // ---------------- phase 0
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
            for (int j = 0; j < 2; ++j) xa[ks][j] = *reinterpret_cast<const bf16x8*>(cur + x_base + j * 16 * 128 + off[ks]);
#pragma unroll
            for (int i = 0; i < 4; ++i) wf[ks][i] = *reinterpret_cast<const bf16x8*>(cur + w_base + i * 16 * 128 + off[ks]);
        }
        if (more) stage_half(nb, 0, kt + 1);
        PHASE_FENCE();
        mfma_quadrant<0, 0>(wf, xa, acc);
        PHASE_FENCE();
        // ---------------- phase 1
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int j = 0; j < 2; ++j) xb[ks][j] = *reinterpret_cast<const bf16x8*>(cur + x_base + (2 + j) * 16 * 128 + off[ks]);
        if (more) stage_half(nb, 1, kt + 1);
        PHASE_FENCE();
        mfma_quadrant<0, 2>(wf, xb, acc);
        PHASE_FENCE();
        // ---------------- phase 2
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int i = 0; i < 4; ++i) wf[ks][i] = *reinterpret_cast<const bf16x8*>(cur + w_base + (4 + i) * 16 * 128 + off[ks]);
        PHASE_FENCE();
        mfma_quadrant<4, 2>(wf, xb, acc);
        PHASE_FENCE();
        // ---------------- phase 3
        if (more2) {
            stage_half(kt & 1, 2, kt + 2); stage_half(kt & 1, 3, kt + 2);
            asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        PHASE_FENCE();
        mfma_quadrant<4, 0>(wf, xa, acc);
        PHASE_FENCE();
    }
    if (wr == 0) PHASE_FENCE();                                                  // pair group 1's last barrier

    // ---- epilogue: lane holds features f = fbase + i*16 + r (r = 0..3) of batch row b = bbase + j*16
    const int fbase = col0 + wr * 128 + fq * 4;
    const int bbase = row0 + wc * 64 + fr;
    const float fl = epilogue_floor<MTC_BF16>(thr);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int f = fbase + i * 16;
        if (f >= N) continue;
        const uint2 bb = *reinterpret_cast<const uint2*>(bias + f);
        const float b0 = bf16_bits_to_f32(bb.x & 0xffffu), b1 = bf16_bits_to_f32(bb.x >> 16);
        const float b2 = bf16_bits_to_f32(bb.y & 0xffffu), b3 = bf16_bits_to_f32(bb.y >> 16);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int b = bbase + j * 16;
            if (b >= M) continue;
            const f32x4 v = acc[i][j];
            *reinterpret_cast<uint2*>(acts + (size_t)b * N + f) =
                relu_pack4(round_bf16(round_bf16(v[0]) + b0), round_bf16(round_bf16(v[1]) + b1),
                           round_bf16(round_bf16(v[2]) + b2), round_bf16(round_bf16(v[3]) + b3), fl);
        }
    }
}

// ------------------------------------------------------------------------------------------- K1 bf16 MFMA, 256x256 tile, persistent
// Same tile, LDS image, phase schedule and MFMA order as encode_bf16_mfma256_kernel (bit-identical activations),
// restructured around what the time-over-K measurement showed: at config #2 the k-loop of that kernel accounts for
// 0.44 ms of its 0.58-0.65 ms; the rest is per-tile fixed cost (a cold prologue, an epilogue of 32 8-byte stores per
// lane, ~1000 VALU instructions of rounding) paid 9 times per CU with nothing running beside it.  Here
//   * ONE workgroup per CU walks its tiles (XCD-contiguous logical ids, stride = workgroups per XCD);
//   * the LDS-DMA stream never stops: W / X stages of the NEXT tile's first k-tiles are issued in the last two
//     k-steps of the current one (the stream is indexed by a running step counter, buffer = step parity), so only
//     the first tile of a workgroup has a prologue;
//   * the epilogue rounds / biases / ReLUs on packed pairs (v_cvt_pk_bf16_f32, v_pk_max_i16: ~3.5 VALU per value
//     instead of ~9), transposes each wave's 64 x 128 block through a private 4 KiB LDS region (XOR-swizzled, no
//     barrier) and writes it as 16-byte stores, four full 256-byte row segments per wave-instruction (the 8-byte
//     stores wrote 32-byte partial lines: 1.6x the bytes at the fabric); the stores drain under the next tile's MFMAs.
typedef __attribute__((ext_vector_type(2))) short short2v;

// round(round(acc) + bias), ReLU, for a pair of values, packed as two bf16 (sae.py:173-174)
template <bool THR>
__device__ __forceinline__ uint32_t bias_relu_pack2(float v0, float v1, float b0, float b1, float fl) {
    const uint32_t p = pack2_bf16(v0, v1);                                         // first rounding: the GEMM result
    const uint32_t q = pack2_bf16(__uint_as_float(p << 16) + b0, __uint_as_float(p & 0xffff0000u) + b1);   // second: + b_enc
    if (!THR) {
        // ReLU on the packed pair: a signed 16-bit max with 0 clears negative values and -0.0 (the select reads bit patterns)
        const short2v z = {0, 0};
        return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(short2v, q), z));
    }
    const float y0 = __uint_as_float(q << 16), y1 = __uint_as_float(q & 0xffff0000u);
    return (y0 > fl ? (q & 0xffffu) : 0u) | (y1 > fl ? (q & 0xffff0000u) : 0u);
}

__device__ __forceinline__ void lid_to_tile(int lid, int mt, int nt, int& tm, int& tn) {
    constexpr int GROUP = MTC_ENC_GROUP;     // logical ids walk GROUP-wide groups of batch tiles under each feature tile (see xcd_tile_coords)
    const int per_group = GROUP * nt;
    const int g = lid / per_group;
    const int first_m = g * GROUP;
    const int gsz = min(mt - first_m, GROUP);
    const int in = lid - g * per_group;
    tm = first_m + in % gsz;
    tn = in / gsz;
}

#ifdef MTC_ENCODE_STAMPS
// Diagnostic build only (tools/encode_stamps.sh builds a separate libmtc_diag.so): per-wave cycle accounting of the
// persistent encoder.  Stamp values go to a buffer of their own; no output value depends on them.
__device__ unsigned long long g_enc_stamps[3 * 256 * 8 * 8];
#define STAMP() __builtin_amdgcn_s_memtime()
#endif

constexpr int STG_BYTES = 4096;                 // per-wave epilogue staging: 16 batch rows x 128 features x 2 B
constexpr int P256_LDS = 2 * 4 * HALF_BYTES + 8 * STG_BYTES;     // 160 KiB: the whole LDS of a CU

#ifdef MTC_ENCODE_STAMPS
#define PFENCE(i) do { PHASE_FENCE(); { const unsigned long long t_ = STAMP(); t_iv[i] += t_ - t_prev; t_prev = t_; } } while (0)
#else
#define PFENCE(i) PHASE_FENCE()
#endif

// MARK = 1 (training loop): the epilogue also leaves, per batch row and per 128-feature block, one byte whose bit i says
// "features [16 i, 16 i + 16) of this block may hold an activation >= the select's carried bound" (carry[0], the
// lower bound of this step's TopK threshold; select.hip).  The select's candidate pass then reads only the flagged
// 32-byte granules of the activations (~8 % of them) instead of all B * D: the first pass over the activations happens
// here, on the packed values the epilogue holds anyway (one packed max + a compare per 4 values, two cross-lane ORs and
// one byte store per 16 rows x 128 features).  marks: uint8 [M][ceil(N / 128)] + one uint32 behind it (4-byte aligned)
// that records the bound the marks were made for -- a select running with another bound ignores them.
//
// EPI 1 / 2 (mtc_gemm_nt on large shapes: the auxiliary branch's products): plain C = A . Bt^T rounded once to bf16 / kept
// in fp32, row strides lda / ldb / ldc, optional row gather on A (a_rows) -- the same tile walk, staging and schedule.
template <bool THR, bool MARK = false, int EPI = 0>
__global__ __launch_bounds__(512, 1) void encode_bf16_mfma256p_kernel(const u16* __restrict__ X, const u16* __restrict__ Wt,
                                                                      const u16* __restrict__ bias, u16* __restrict__ acts,
                                                                      int M, int N, int K, int mt, int nt, const void* __restrict__ thr,
                                                                      const uint32_t* __restrict__ carry = nullptr,
                                                                      unsigned char* __restrict__ marks = nullptr,
                                                                      long long lda = 0, long long ldb = 0, long long ldc = 0,
                                                                      const int32_t* __restrict__ a_rows = nullptr) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    uint32_t g16 = 0x10000u;                                                     // "nothing is a candidate"
    const int nblk = (N + 127) >> 7;
    if constexpr (MARK) {
        const uint32_t guess = carry[0];
        if (guess != 0) g16 = guess >> 16;
        if (blockIdx.x == 0 && tid == 0) *reinterpret_cast<uint32_t*>(marks + ((((size_t)M * nblk) + 3) & ~(size_t)3)) = guess;
    }
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wid >> 2, wc = wid & 3;
    // ---- this workgroup's tiles: logical ids first, first + stride, ... (count of them)
    const int tiles = mt * nt, G = gridDim.x, bid = blockIdx.x;
    int first, stride, count;
    if ((G & 7) == 0) {
        // blocks b, b+8, ... share an XCD (speed only): each XCD walks a contiguous range of logical ids
        const int q = tiles >> 3, r = tiles & 7, xcd = bid & 7, j = bid >> 3, gx = G >> 3;
        const int start = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
        const int cnt = q + (xcd < r ? 1 : 0);
        first = start + j; stride = gx;
        count = j < cnt ? (cnt - j + gx - 1) / gx : 0;
    } else {
        first = bid; stride = G;
        count = bid < tiles ? (tiles - bid + G - 1) / G : 0;
    }
    if (count == 0) return;                                                      // whole workgroup: no barrier entered yet
    const int ktiles = K / BK;
    const int S = count * ktiles;                                                // k-steps of this workgroup, all tiles
#ifdef MTC_ENCODE_STAMPS
    const unsigned long long t_start = STAMP(), rt_start = __builtin_amdgcn_s_memrealtime();
    unsigned long long t_epi = 0, t_first_wait = 0, t_other_wait = 0, t_loop0 = 0, t_first_bar = 0;
    bool after_epi = false;
    unsigned long long t_iv[8] = {0, 0, 0, 0, 0, 0, 0, 0}, t_prev = 0;
    unsigned long long t_p0_issue = 0, t_p3_issue = 0, t_p3_wait = 0, t_m0 = 0;
#endif

    // ---- staging: half-tile h has 16 pieces of 8 rows x 128 B; wave wid moves pieces 2 wid, 2 wid + 1 of every half
    const int prow0 = (wid * 2) * 8 + (lane >> 3), prow1 = prow0 + 8;            // row inside a half-tile
    const int pch0 = ((lane & 7) ^ ((prow0 >> 1) & 7)) * 8, pch1 = ((lane & 7) ^ ((prow1 >> 1) & 7)) * 8;
    const u16* wsrc[2][2];
    const u16* xsrc[2][2];
    auto set_wh = [&](int lid, int h) {
        int tm, tn;
        lid_to_tile(lid, mt, nt, tm, tn);
        const long long ld = EPI == 0 ? (long long)K : ldb;
        wsrc[h][0] = Wt + (size_t)min(tn * B2N + h * 128 + prow0, N - 1) * ld + pch0;    // clamp: tails read valid rows
        wsrc[h][1] = Wt + (size_t)min(tn * B2N + h * 128 + prow1, N - 1) * ld + pch1;
    };
    auto set_w = [&](int lid) { set_wh(lid, 0); set_wh(lid, 1); };
    auto set_x = [&](int lid) {
        int tm, tn;
        lid_to_tile(lid, mt, nt, tm, tn);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const long long ld = EPI == 0 ? (long long)K : lda;
            int r0 = min(tm * B2M + h * 128 + prow0, M - 1), r1 = min(tm * B2M + h * 128 + prow1, M - 1);
            if constexpr (EPI != 0) {
                // (a plain load between the LDS-DMA requests: its wait drains them once per tile, which only makes the
                // counted waits behind it stronger)
                if (a_rows) { r0 = a_rows[r0]; r1 = a_rows[r1]; }
            }
            xsrc[h][0] = X + (size_t)r0 * ld + pch0;
            xsrc[h][1] = X + (size_t)r1 * ld + pch1;
        }
    };
    auto stage_w = [&](int buf, int h, int kt) {
        char* base = smem + buf * 4 * HALF_BYTES + h * HALF_BYTES + wid * 2 * 1024;
        glds16_aux<MTC_ENC_W_AUX>(wsrc[h][0] + kt * BK, base);
        glds16_aux<MTC_ENC_W_AUX>(wsrc[h][1] + kt * BK, base + 1024);
    };
    auto stage_x = [&](int buf, int h, int kt) {
        char* base = smem + buf * 4 * HALF_BYTES + (2 + h) * HALF_BYTES + wid * 2 * 1024;
        glds16_aux<MTC_ENC_X_AUX>(xsrc[h][0] + kt * BK, base);
        glds16_aux<MTC_ENC_X_AUX>(xsrc[h][1] + kt * BK, base + 1024);
    };
    // the two operand streams run ahead of the MFMAs by one (W) / two (X) k-steps and cross tile boundaries
    int wk = 0, wm = 0, xk = 0, xm = 0;                                           // next stage to issue: k-tile, tile ordinal
#ifdef MTC_ENC_W0_AHEAD
    // W half 0 runs two k-steps ahead (staged in phase 3 beside X), W half 1 one k-step ahead (phase 1): phase 0 issues no LDS-DMA
    int w0k = 0, w0m = 0;
    auto adv_w0 = [&]() { if (++w0k == ktiles) { w0k = 0; if (++w0m < count) set_wh(first + w0m * stride, 0); } };
    auto adv_w = [&]() { if (++wk == ktiles) { wk = 0; if (++wm < count) set_wh(first + wm * stride, 1); } };
#else
    auto adv_w = [&]() { if (++wk == ktiles) { wk = 0; if (++wm < count) set_w(first + wm * stride); } };
#endif
    auto adv_x = [&]() { if (++xk == ktiles) { xk = 0; if (++xm < count) set_x(first + xm * stride); } };

    const int fr = lane & 15, fq = lane >> 4;
    const int swz = (fr >> 1) & 7;
    int off[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) off[ks] = fr * 128 + (((ks * 4 + fq) ^ swz) << 4);
    const int w_base = wr * HALF_BYTES;                                          // this wave's W half-tile
    const int x_base = (2 + (wc >> 1)) * HALF_BYTES + (wc & 1) * 64 * 128;     // its 64 batch rows

    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    bf16x8 wf[2][4], xa[2][2], xb[2][2];

    // prologue (first tile only): step 0 entirely, plus the X halves of step 1
    set_w(first); set_x(first);
    stage_w(0, 0, 0); stage_w(0, 1, 0); adv_w();
#ifdef MTC_ENC_W0_AHEAD
    adv_w0();
#endif
    stage_x(0, 0, 0); stage_x(0, 1, 0); adv_x();
    if (S > 1) {
        stage_x(1, 0, xk); stage_x(1, 1, xk); adv_x();
#ifdef MTC_ENC_W0_AHEAD
        stage_w(1, 0, w0k); adv_w0();
        asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
#else
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
#endif
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    PHASE_FENCE();
    if (wr == 1) PHASE_FENCE();                                                  // stagger the second wave group

    const float fl = epilogue_floor<MTC_BF16>(thr);
    char* stg = smem + 2 * 4 * HALF_BYTES + wid * STG_BYTES;
    int kt = 0, cm = 0;
#ifdef MTC_ENCODE_STAMPS
    t_loop0 = STAMP();
    t_prev = t_loop0;
#endif
    for (int it = 0; it < S; ++it) {
        const char* cur = smem + (it & 1) * 4 * HALF_BYTES;
        const int nb = (it + 1) & 1;
        const bool more = it + 1 < S, more2 = it + 2 < S;
        // ---------------- phase 0
#if MTC_ENC_P0_ORDER == 1
        if (more) stage_w(nb, 0, wk);
        __builtin_amdgcn_sched_barrier(0);
#endif
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
#if MTC_ENC_P0_ORDER == 2
            if (ks == 1) {
                __builtin_amdgcn_sched_barrier(0);
                if (more) stage_w(nb, 0, wk);
                __builtin_amdgcn_sched_barrier(0);
            }
#endif
#ifndef MTC_ABLATE_P0_XA
#pragma unroll
            for (int j = 0; j < 2; ++j) xa[ks][j] = *reinterpret_cast<const bf16x8*>(cur + x_base + j * 16 * 128 + off[ks]);
#endif
#pragma unroll
            for (int i = 0; i < 4; ++i) wf[ks][i] = *reinterpret_cast<const bf16x8*>(cur + w_base + i * 16 * 128 + off[ks]);
        }
#if !defined(MTC_ENC_STAGE_W_P1) && MTC_ENC_P0_ORDER == 0 && !defined(MTC_ENC_W0_AHEAD)
        if (more) stage_w(nb, 0, wk);
#endif
#ifdef MTC_ENCODE_STAMPS
        { const unsigned long long ts_ = STAMP(); t_p0_issue += ts_ - t_prev; }
#endif
        PFENCE(0);
        mfma_quadrant<0, 0>(wf, xa, acc);
#ifdef MTC_ENCODE_STAMPS
        { const unsigned long long ts_ = STAMP(); t_m0 += ts_ - t_prev; }
#endif
        PFENCE(1);
        // ---------------- phase 1
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int j = 0; j < 2; ++j) xb[ks][j] = *reinterpret_cast<const bf16x8*>(cur + x_base + (2 + j) * 16 * 128 + off[ks]);
#ifdef MTC_ENC_STAGE_W_P1
        if (more) stage_w(nb, 0, wk);
#endif
        if (more) { stage_w(nb, 1, wk); adv_w(); }
        PFENCE(2);
        mfma_quadrant<0, 2>(wf, xb, acc);
        PFENCE(3);
        // ---------------- phase 2
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int i = 0; i < 4; ++i) wf[ks][i] = *reinterpret_cast<const bf16x8*>(cur + w_base + (4 + i) * 16 * 128 + off[ks]);
        PFENCE(4);
        mfma_quadrant<4, 2>(wf, xb, acc);
        PFENCE(5);
        // ---------------- phase 3
#ifdef MTC_ENCODE_STAMPS
        const unsigned long long tw0 = STAMP();
#endif
        if (more2) {
            stage_x(it & 1, 0, xk); stage_x(it & 1, 1, xk); adv_x();
#ifdef MTC_ENC_W0_AHEAD
            stage_w(it & 1, 0, w0k); adv_w0();
#endif
#ifdef MTC_ENCODE_STAMPS
            __builtin_amdgcn_sched_barrier(0);
            { const unsigned long long ts_ = STAMP(); t_p3_issue += ts_ - tw0; t_p3_wait -= ts_; }
#endif
#ifdef MTC_ENC_W0_AHEAD
            asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
#else
            asm volatile("s_waitcnt vmcnt(4)" ::: "memory");                     // everything older (all of step it+1, earlier stores) is done
#endif
#ifdef MTC_ENCODE_STAMPS
            t_p3_wait += STAMP();
#endif
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
#ifdef MTC_ENCODE_STAMPS
        const unsigned long long tw1 = STAMP();
#endif
        PFENCE(6);
#ifdef MTC_ENCODE_STAMPS
        const unsigned long long tw2 = STAMP();
        if (after_epi) { t_first_wait += tw1 - tw0; t_first_bar += tw2 - tw1; } else t_other_wait += tw2 - tw0;
        after_epi = false;
#endif
        mfma_quadrant<4, 0>(wf, xa, acc);
        PFENCE(7);
        if (++kt < ktiles) continue;
#ifdef MTC_ENCODE_STAMPS
        const unsigned long long te0 = STAMP();
#endif
        // ---------------- tile done: epilogue (no barrier; the next tile's first stages are already in flight)
        kt = 0;
        int tm, tn;
        lid_to_tile(first + cm * stride, mt, nt, tm, tn);
        ++cm;
        const int fcol = tn * B2N + wr * 128;                                    // first feature of this wave's block
        const int brow = tm * B2M + wc * 64;                                     // first batch row of it
        if constexpr (EPI == 2) {
            // fp32 result straight from the accumulators: 16 bytes per lane, 64 contiguous bytes per four lanes
            float* C32 = reinterpret_cast<float*>(acts);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int b = brow + j * 16 + fr;
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int f = fcol + i * 16 + fq * 4;
                    if (b < M && f < N) *reinterpret_cast<f32x4*>(C32 + (size_t)b * ldc + f) = acc[i][j];
                    acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
                }
            }
            continue;
        }
        uint2 bb[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int f = fcol + i * 16 + fq * 4;
            bb[i] = (EPI == 0 && f < N) ? *reinterpret_cast<const uint2*>(bias + f) : uint2{0u, 0u};   // N % 8 == 0 -> in bounds, aligned
        }
        const long long ldo = EPI == 0 ? (long long)N : ldc;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            // lane holds features i*16 + fq*4 + (0..3) of batch row j*16 + fr: 8 bytes into row fr of the staging block,
            // 16-byte chunk (2 i + fq/2) XOR-ed with the row (conflict-free ds_write_b64 / ds_read_b128)
            uint32_t mark = 0;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const f32x4 v = acc[i][j];
                uint2 o;
                if constexpr (EPI == 0) {
                    o.x = bias_relu_pack2<THR>(v[0], v[1], bf16_bits_to_f32(bb[i].x & 0xffffu), bf16_bits_to_f32(bb[i].x >> 16), fl);
                    o.y = bias_relu_pack2<THR>(v[2], v[3], bf16_bits_to_f32(bb[i].y & 0xffffu), bf16_bits_to_f32(bb[i].y >> 16), fl);
                } else {
                    o.x = pack2_bf16(v[0], v[1]); o.y = pack2_bf16(v[2], v[3]);                    // one rounding
                }
                *reinterpret_cast<uint2*>(stg + fr * 256 + ((((i << 1) | (fq >> 1)) ^ fr) << 4) + ((fq & 1) << 3)) = o;
                acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
                if constexpr (MARK) {
                    typedef __attribute__((ext_vector_type(2))) unsigned short us2v;
                    const uint32_t ox = o.x, oy = o.y;
                    const us2v m2 = __builtin_elementwise_max(__builtin_bit_cast(us2v, ox), __builtin_bit_cast(us2v, oy));
                    const uint32_t mx = max((uint32_t)m2.x, (uint32_t)m2.y);
                    // (columns beyond N hold products of clamped weight rows: never candidates)
                    mark |= (mx >= g16 && fcol + i * 16 + fq * 4 < N) ? (1u << i) : 0u;
                }
            }
            if constexpr (MARK) {
                // the 16 features of bit i sit in the four lanes fr, fr + 16, fr + 32, fr + 48
                mark |= (uint32_t)__shfl_xor((int)mark, 16, 64);
                mark |= (uint32_t)__shfl_xor((int)mark, 32, 64);
                const int b = brow + j * 16 + fr;
                if (fq == 0 && b < M && fcol < N) marks[(size_t)b * nblk + (fcol >> 7)] = (unsigned char)mark;
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                   // wave-private region: no barrier needed
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const int R = (lane >> 4) + 4 * s, c = lane & 15;
                const u32x4 val = *reinterpret_cast<const u32x4*>(stg + R * 256 + ((c ^ R) << 4));
                const int b = brow + j * 16 + R, f = fcol + c * 8;
                if (b < M && f < N) *reinterpret_cast<u32x4*>(acts + (size_t)b * ldo + f) = val;
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                   // reads done before the next round overwrites
        }
#ifdef MTC_ENCODE_STAMPS
        t_epi += STAMP() - te0;
        t_prev = STAMP();
        after_epi = true;
#endif
    }
    if (wr == 0) PHASE_FENCE();                                                  // pair group 1's last barrier
#ifdef MTC_ENCODE_STAMPS
    if (lane == 0) {
        unsigned long long* o = g_enc_stamps + ((size_t)(bid & 255) * 8 + wid) * 8;
        const unsigned long long t_end = STAMP();
        o[0] = t_end - t_start; o[1] = t_loop0 - t_start; o[2] = t_epi; o[3] = t_first_wait; o[4] = t_other_wait;
        o[5] = __builtin_amdgcn_s_memrealtime() - rt_start; o[6] = t_first_bar; o[7] = (unsigned long long)count;
        unsigned long long* o2 = g_enc_stamps + 256 * 8 * 8 + ((size_t)(bid & 255) * 8 + wid) * 8;
        for (int i = 0; i < 8; ++i) o2[i] = t_iv[i];
        unsigned long long* o3 = g_enc_stamps + 2 * 256 * 8 * 8 + ((size_t)(bid & 255) * 8 + wid) * 8;
        o3[0] = t_p0_issue; o3[1] = t_p3_issue; o3[2] = t_p3_wait; o3[3] = t_m0; o3[4] = 0; o3[5] = 0; o3[6] = 0; o3[7] = 0;
    }
#endif
}

// ------------------------------------------------------------------------------------------- K1 generic
// Any shape / dtype (fp32 model dtype, and bf16 shapes whose K is not a multiple of 64).
// 64x64 tile, 16-deep k slices, 4x4 outputs per thread, fp32 FMA.
template <int DT, int EPI>
__global__ __launch_bounds__(256) void encode_generic_kernel(const void* __restrict__ X, const void* __restrict__ Wt,
                                                             const void* __restrict__ bias, void* __restrict__ acts,
                                                             int M, int N, int K, const void* __restrict__ thr,
                                                             long long lda, long long ldb, long long ldc,
                                                             const int32_t* __restrict__ a_rows) {
    typedef DType<DT> D_;
    typedef typename D_::T T;
    __shared__ float xs[16][64 + 4];
    __shared__ float ws[16][64 + 4];
    const int tid = threadIdx.x;
    const int row0 = blockIdx.y * 64, col0 = blockIdx.x * 64;
    const int tx = tid & 15, ty = tid >> 4;
    float acc[4][4] = {};
    const T* Xp = reinterpret_cast<const T*>(X);
    const T* Wp = reinterpret_cast<const T*>(Wt);
    for (int k0 = 0; k0 < K; k0 += 16) {
        for (int i = tid; i < 64 * 16; i += 256) {
            const int r = i >> 4, k = i & 15;
            const bool kin = (k0 + k) < K;
            if (EPI == 0) {
                xs[k][r] = (kin && row0 + r < M) ? D_::ld(Xp + (size_t)(row0 + r) * K + k0 + k) : 0.f;
                ws[k][r] = (kin && col0 + r < N) ? D_::ld(Wp + (size_t)(col0 + r) * K + k0 + k) : 0.f;
            } else {
                const bool xin = kin && row0 + r < M;
                const long long xr = xin ? (a_rows ? (long long)a_rows[row0 + r] : (long long)(row0 + r)) : 0;
                xs[k][r] = xin ? D_::ld(Xp + (size_t)xr * lda + k0 + k) : 0.f;
                ws[k][r] = (kin && col0 + r < N) ? D_::ld(Wp + (size_t)(col0 + r) * ldb + k0 + k) : 0.f;
            }
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            float a[4], b[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) { a[i] = xs[k][ty * 4 + i]; b[i] = ws[k][tx * 4 + i]; }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(a[i], b[j], acc[i][j]);
        }
        __syncthreads();
    }
    T* out = reinterpret_cast<T*>(acts);
    const T* bp = reinterpret_cast<const T*>(bias);
    const float fl = EPI == 0 ? epilogue_floor<DT>(thr) : 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = row0 + ty * 4 + i;
        if (r >= M) continue;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int c = col0 + tx * 4 + j;
            if (c >= N) continue;
            if (EPI == 0) {
                const float pre = D_::rnd(D_::rnd(acc[i][j]) + D_::ld(bp + c));
                D_::st(out + (size_t)r * N + c, pre > fl ? pre : 0.f);
            } else if (EPI == 2) {
                reinterpret_cast<float*>(acts)[(size_t)r * ldc + c] = acc[i][j];
            } else {
                D_::st(out + (size_t)r * ldc + c, acc[i][j]);
            }
        }
    }
}

}  // namespace mtc

using namespace mtc;

extern "C" int mtc_row_normalize(const mtc_desc* d, const void* x, void* x_norm, void* mean, void* std,
                                 void* stream) {
    if (check_desc(d, "mtc_row_normalize")) return -1;
    MTC_REQUIRE(x && x_norm && mean && std, "mtc_row_normalize: null pointer");
    MTC_REQUIRE(d->A_s >= 8, "mtc_row_normalize: A_s must be >= 8");
    const dim3 grid((d->B + 3) / 4), block(256);
    if (d->dtype == MTC_BF16)
        hipLaunchKernelGGL(row_normalize_kernel<MTC_BF16>, grid, block, 0, as_stream(stream), x, x_norm, mean, std, d->B, d->A_s);
    else
        hipLaunchKernelGGL(row_normalize_kernel<MTC_F32>, grid, block, 0, as_stream(stream), x, x_norm, mean, std, d->B, d->A_s);
    return check_launch("mtc_row_normalize");
}

// Tile choice and one-time kernel attributes.  The environment is read ONCE (first launch), the LDS opt-in of
// each kernel is set once per device; mtc_encode_set_tile() overrides the choice at run time (tests force the
// 256 x 256 kernel at small shapes through it).
static int g_tile = -1;               // -1: not initialised; 0 auto, 1 = 128 x 128, 2 = 256 x 256
static int g_f32_generic = -1;
static void encode_options_once() {
    if (g_tile < 0) {
        const char* force = getenv("MTC_ENCODE_TILE");
        g_tile = (force && force[0] >= '1' && force[0] <= '3') ? force[0] - '0' : 0;
    }
    if (g_f32_generic < 0) g_f32_generic = getenv("MTC_ENCODE_F32_GENERIC") ? 1 : 0;
}
constexpr int MAX_DEV = 64;
static bool g_attr_done[12][MAX_DEV];
static int g_cus[MAX_DEV];
static void lds_opt_in(int which, const void* fn, int lds) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) dev = -1;
    if (dev >= 0 && dev < MAX_DEV && g_attr_done[which][dev]) return;
    if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) return;   // the launch reports it
    if (dev >= 0 && dev < MAX_DEV) g_attr_done[which][dev] = true;
}

static int cu_count() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAX_DEV) return 256;
    if (g_cus[dev] == 0) {
        int n = 0;
        g_cus[dev] = (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0) ? n : 256;
    }
    return g_cus[dev];
}

#ifdef MTC_ENCODE_STAMPS
extern "C" int mtc_diag_read_encode_stamps(unsigned long long* host, size_t n) {
    return hipMemcpyFromSymbol(host, HIP_SYMBOL(g_enc_stamps), n * sizeof(unsigned long long)) == hipSuccess ? 0 : -1;
}
#endif

extern "C" int mtc_encode_set_tile(int tile) {
    encode_options_once();
    const int prev = g_tile;
    MTC_REQUIRE(tile >= 0 && tile <= 3, "mtc_encode_set_tile: tile must be 0 (auto), 1 (128 x 128), 2 (256 x 256) or 3 (256 x 256, one tile per workgroup)");
    g_tile = tile;
    return prev;
}

// does this shape run the persistent 256 x 256 kernel (the one that can mark candidate granules)?
static bool uses_persistent(const mtc_desc* d) {
    encode_options_once();
    const int M = d->B, N = d->D, K = d->A_s;
    const int big_tiles = ((M + B2M - 1) / B2M) * ((N + B2N - 1) / B2N);
    return d->dtype == MTC_BF16 && (K % BK) == 0 && g_tile != 3 && ((big_tiles >= 512 && g_tile != 1) || g_tile == 2);
}

static int encode_impl(const mtc_desc* d, const void* x, const void* W_encT, const void* b_enc, const void* thr,
                       void* acts, void* stream, const uint32_t* carry = nullptr, unsigned char* marks = nullptr) {
    const int M = d->B, N = d->D, K = d->A_s;
    encode_options_once();
    if (d->dtype == MTC_F32 && (K % BK32) == 0 && !g_f32_generic) {
        const int lds = 2 * 2 * TILE_BYTES;
        lds_opt_in(0, reinterpret_cast<const void*>(encode_f32_mfma_kernel<0>), lds);
        const int mt = (M + BM - 1) / BM, nt = (N + BN - 1) / BN;
        hipLaunchKernelGGL(encode_f32_mfma_kernel<0>, dim3(mt * nt), dim3(256), lds, as_stream(stream),
                           reinterpret_cast<const float*>(x), reinterpret_cast<const float*>(W_encT),
                           reinterpret_cast<const float*>(b_enc), reinterpret_cast<float*>(acts), M, N, K, mt, nt, thr,
                           0ll, 0ll, 0ll, (const int32_t*)nullptr);
        return check_launch("mtc_encode_fwd");
    }
    const int big_tiles = ((M + B2M - 1) / B2M) * ((N + B2N - 1) / B2N);
    if (d->dtype == MTC_BF16 && (K % BK) == 0 && g_tile == 3) {
        // the round-1 one-tile-per-workgroup kernel, kept for A/B timing against the persistent one
        const int lds = 2 * 4 * HALF_BYTES;
        lds_opt_in(1, reinterpret_cast<const void*>(encode_bf16_mfma256_kernel), lds);
        const int mt = (M + B2M - 1) / B2M, nt = (N + B2N - 1) / B2N;
        hipLaunchKernelGGL(encode_bf16_mfma256_kernel, dim3(mt * nt), dim3(512), lds, as_stream(stream),
                           (const u16*)x, (const u16*)W_encT, (const u16*)b_enc, (u16*)acts, M, N, K, mt, nt, thr);
    } else if (d->dtype == MTC_BF16 && (K % BK) == 0 && ((big_tiles >= 512 && g_tile != 1) || g_tile == 2)) {
        const int mt = (M + B2M - 1) / B2M, nt = (N + B2N - 1) / B2N;
        const int grid = min(mt * nt, cu_count());                               // one persistent workgroup per CU
        if (thr) {
            lds_opt_in(3, reinterpret_cast<const void*>(encode_bf16_mfma256p_kernel<true, false>), P256_LDS);
            hipLaunchKernelGGL((encode_bf16_mfma256p_kernel<true, false>), dim3(grid), dim3(512), P256_LDS, as_stream(stream),
                               (const u16*)x, (const u16*)W_encT, (const u16*)b_enc, (u16*)acts, M, N, K, mt, nt, thr,
                               (const uint32_t*)nullptr, (unsigned char*)nullptr);
        } else if (carry && marks) {
            lds_opt_in(9, reinterpret_cast<const void*>(encode_bf16_mfma256p_kernel<false, true>), P256_LDS);
            hipLaunchKernelGGL((encode_bf16_mfma256p_kernel<false, true>), dim3(grid), dim3(512), P256_LDS, as_stream(stream),
                               (const u16*)x, (const u16*)W_encT, (const u16*)b_enc, (u16*)acts, M, N, K, mt, nt, thr, carry, marks);
        } else {
            lds_opt_in(4, reinterpret_cast<const void*>(encode_bf16_mfma256p_kernel<false, false>), P256_LDS);
            hipLaunchKernelGGL((encode_bf16_mfma256p_kernel<false, false>), dim3(grid), dim3(512), P256_LDS, as_stream(stream),
                               (const u16*)x, (const u16*)W_encT, (const u16*)b_enc, (u16*)acts, M, N, K, mt, nt, thr,
                               (const uint32_t*)nullptr, (unsigned char*)nullptr);
        }
    } else if (d->dtype == MTC_BF16 && (K % BK) == 0) {
        const int lds = 2 * 2 * TILE_BYTES;
        lds_opt_in(2, reinterpret_cast<const void*>(encode_bf16_mfma_kernel<0>), lds);
        const int mt = (M + BM - 1) / BM, nt = (N + BN - 1) / BN;
        hipLaunchKernelGGL(encode_bf16_mfma_kernel<0>, dim3(mt * nt), dim3(256), lds, as_stream(stream),
                           (const u16*)x, (const u16*)W_encT, (const u16*)b_enc, acts, M, N, K, mt, nt, thr,
                           0ll, 0ll, 0ll, (const int32_t*)nullptr);
    } else {
        const dim3 grid((N + 63) / 64, (M + 63) / 64), block(256);
        if (d->dtype == MTC_BF16)
            hipLaunchKernelGGL((encode_generic_kernel<MTC_BF16, 0>), grid, block, 0, as_stream(stream), x, W_encT, b_enc, acts, M, N, K, thr,
                               0ll, 0ll, 0ll, (const int32_t*)nullptr);
        else
            hipLaunchKernelGGL((encode_generic_kernel<MTC_F32, 0>), grid, block, 0, as_stream(stream), x, W_encT, b_enc, acts, M, N, K, thr,
                               0ll, 0ll, 0ll, (const int32_t*)nullptr);
    }
    return check_launch("mtc_encode_fwd");
}

extern "C" int mtc_encode_fwd(const mtc_desc* d, const void* x, const void* W_encT, const void* b_enc, void* acts,
                              void* stream) {
    if (check_desc(d, "mtc_encode_fwd")) return -1;
    MTC_REQUIRE(x && W_encT && b_enc && acts, "mtc_encode_fwd: null pointer");
    return encode_impl(d, x, W_encT, b_enc, nullptr, acts, stream);
}

extern "C" size_t mtc_encode_marks_bytes(const mtc_desc* d) {
    if (!d || !uses_persistent(d)) return 0;
    return ((((size_t)d->B * ((d->D + 127) >> 7)) + 3) & ~(size_t)3) + 4;
}

extern "C" int mtc_encode_fwd_marked(const mtc_desc* d, const void* x, const void* W_encT, const void* b_enc, void* acts,
                                     const uint32_t* carry, void* marks, void* stream) {
    if (check_desc(d, "mtc_encode_fwd_marked")) return -1;
    MTC_REQUIRE(x && W_encT && b_enc && acts && carry && marks, "mtc_encode_fwd_marked: null pointer");
    MTC_REQUIRE(uses_persistent(d), "mtc_encode_fwd_marked: this shape does not run the marking kernel (mtc_encode_marks_bytes() = 0)");
    return encode_impl(d, x, W_encT, b_enc, nullptr, acts, stream, carry, reinterpret_cast<unsigned char*>(marks));
}

extern "C" int mtc_encode_threshold_infer(const mtc_desc* d, const void* x, const void* W_encT, const void* b_enc,
                                          const void* threshold, void* feats, void* stream) {
    if (check_desc(d, "mtc_encode_threshold_infer")) return -1;
    MTC_REQUIRE(x && W_encT && b_enc && threshold && feats, "mtc_encode_threshold_infer: null pointer");
    return encode_impl(d, x, W_encT, b_enc, threshold, feats, stream);
}

// C[m][n] = sum_k A[row(m)][k] * Bt[n][k]: the products of the auxiliary branch (sae.py:336-341 and their autograd) on
// the encoder's own tiles.  fp32 accumulation, ONE rounding to the model dtype (what a matmul of model-dtype tensors
// gives), or fp32 output (partial sums that are all-reduced over dictionary shards).
extern "C" int mtc_gemm_nt(int32_t dtype, int32_t M, int32_t N, int32_t K, const void* A, int64_t lda, const int32_t* a_rows,
                           const void* Bt, int64_t ldb, void* C, int64_t ldc, int32_t out_f32, void* stream) {
    MTC_REQUIRE(dtype == MTC_F32 || dtype == MTC_BF16, "mtc_gemm_nt: bad dtype %d", dtype);
    MTC_REQUIRE(M >= 0 && N >= 0 && K > 0, "mtc_gemm_nt: bad sizes M %d N %d K %d", M, N, K);
    if (M == 0 || N == 0) return 0;
    MTC_REQUIRE(A && Bt && C, "mtc_gemm_nt: null pointer");
    MTC_REQUIRE(lda >= K && ldb >= K && ldc >= N, "mtc_gemm_nt: row strides shorter than the rows");
    encode_options_once();
    hipStream_t s = as_stream(stream);
    const int es = dtype == MTC_BF16 ? 2 : 4;
    const bool aligned = (lda * es) % 16 == 0 && (ldb * es) % 16 == 0 && (N % 4) == 0 && (ldc % 4) == 0 &&
                         ((uintptr_t)A % 16) == 0 && ((uintptr_t)Bt % 16) == 0 && ((uintptr_t)C % 16) == 0;
    const int mt = (M + BM - 1) / BM, nt = (N + BN - 1) / BN;
    const int lds = 2 * 2 * TILE_BYTES;
    // large products run the persistent 256 x 256 kernel (measured at the auxiliary branch's shapes, 144 - 352 big tiles:
    // 0.121 / 0.109 / 0.099 ms against 0.149 / 0.128 / 0.140 on the 128 x 128 tile, tools/aux_gemm_ab.py); MTC_ENCODE_TILE /
    // mtc_encode_set_tile() force either form like they do for the encoder
    const int big_tiles = ((M + B2M - 1) / B2M) * ((N + B2N - 1) / B2N);
    const bool wide_ok = (N % 8) == 0 && (ldc % 8) == 0;                          // 16-byte bf16 stores of the epilogue
    if (dtype == MTC_BF16 && aligned && (K % BK) == 0 && (out_f32 || wide_ok) && g_tile != 1 && g_tile != 3 &&
        (g_tile == 2 || 2 * big_tiles >= cu_count())) {
        const int mt2 = (M + B2M - 1) / B2M, nt2 = (N + B2N - 1) / B2N;
        const int grid = min(mt2 * nt2, cu_count());
        if (out_f32) {
            lds_opt_in(8, reinterpret_cast<const void*>(encode_bf16_mfma256p_kernel<false, false, 2>), P256_LDS);
            hipLaunchKernelGGL((encode_bf16_mfma256p_kernel<false, false, 2>), dim3(grid), dim3(512), P256_LDS, s, (const u16*)A,
                               (const u16*)Bt, (const u16*)nullptr, (u16*)C, M, N, K, mt2, nt2, (const void*)nullptr,
                               (const uint32_t*)nullptr, (unsigned char*)nullptr, (long long)lda, (long long)ldb, (long long)ldc, a_rows);
        } else {
            lds_opt_in(10, reinterpret_cast<const void*>(encode_bf16_mfma256p_kernel<false, false, 1>), P256_LDS);
            hipLaunchKernelGGL((encode_bf16_mfma256p_kernel<false, false, 1>), dim3(grid), dim3(512), P256_LDS, s, (const u16*)A,
                               (const u16*)Bt, (const u16*)nullptr, (u16*)C, M, N, K, mt2, nt2, (const void*)nullptr,
                               (const uint32_t*)nullptr, (unsigned char*)nullptr, (long long)lda, (long long)ldb, (long long)ldc, a_rows);
        }
    } else if (dtype == MTC_BF16 && aligned && (K % BK) == 0) {
        if (out_f32) {
            lds_opt_in(5, reinterpret_cast<const void*>(encode_bf16_mfma_kernel<2>), lds);
            hipLaunchKernelGGL(encode_bf16_mfma_kernel<2>, dim3(mt * nt), dim3(256), lds, s, (const u16*)A, (const u16*)Bt,
                               (const u16*)nullptr, C, M, N, K, mt, nt, (const void*)nullptr, (long long)lda, (long long)ldb,
                               (long long)ldc, a_rows);
        } else {
            lds_opt_in(6, reinterpret_cast<const void*>(encode_bf16_mfma_kernel<1>), lds);
            hipLaunchKernelGGL(encode_bf16_mfma_kernel<1>, dim3(mt * nt), dim3(256), lds, s, (const u16*)A, (const u16*)Bt,
                               (const u16*)nullptr, C, M, N, K, mt, nt, (const void*)nullptr, (long long)lda, (long long)ldb,
                               (long long)ldc, a_rows);
        }
    } else if (dtype == MTC_F32 && aligned && (K % BK32) == 0) {
        lds_opt_in(7, reinterpret_cast<const void*>(encode_f32_mfma_kernel<1>), lds);
        hipLaunchKernelGGL(encode_f32_mfma_kernel<1>, dim3(mt * nt), dim3(256), lds, s, (const float*)A, (const float*)Bt,
                           (const float*)nullptr, (float*)C, M, N, K, mt, nt, (const void*)nullptr, (long long)lda, (long long)ldb,
                           (long long)ldc, a_rows);
    } else {
        const dim3 grid((N + 63) / 64, (M + 63) / 64), block(256);
        if (dtype == MTC_BF16 && out_f32)
            hipLaunchKernelGGL((encode_generic_kernel<MTC_BF16, 2>), grid, block, 0, s, A, Bt, (const void*)nullptr, C, M, N, K,
                               (const void*)nullptr, (long long)lda, (long long)ldb, (long long)ldc, a_rows);
        else if (dtype == MTC_BF16)
            hipLaunchKernelGGL((encode_generic_kernel<MTC_BF16, 1>), grid, block, 0, s, A, Bt, (const void*)nullptr, C, M, N, K,
                               (const void*)nullptr, (long long)lda, (long long)ldb, (long long)ldc, a_rows);
        else
            hipLaunchKernelGGL((encode_generic_kernel<MTC_F32, 1>), grid, block, 0, s, A, Bt, (const void*)nullptr, C, M, N, K,
                               (const void*)nullptr, (long long)lda, (long long)ldb, (long long)ldc, a_rows);
    }
    return check_launch("mtc_gemm_nt");
}
