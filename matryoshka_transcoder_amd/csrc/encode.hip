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
    // logical ids walk 8-wide groups of batch tiles under each feature tile
    constexpr int GROUP = 8;
    const int per_group = GROUP * nt;
    const int g = lid / per_group;
    const int first_m = g * GROUP;
    const int gsz = min(mt - first_m, GROUP);
    const int in = lid - g * per_group;
    tm = first_m + in % gsz;
    tn = in / gsz;
}

__device__ __forceinline__ void glds16(const void* gsrc, void* lds_dst) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_dst, 16, 0, 0);
}

__global__ __launch_bounds__(256, 2) void encode_bf16_mfma_kernel(const u16* __restrict__ X, const u16* __restrict__ Wt,
                                                                  const u16* __restrict__ bias, u16* __restrict__ acts,
                                                                  int M, int N, int K, int mt, int nt) {
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
        const int xr = min(row0 + r, M - 1), wr = min(col0 + r, N - 1);   // clamp: tails read valid rows
        xsrc[p] = X + (size_t)xr * K + chunk * 8;
        wsrc[p] = Wt + (size_t)wr * K + chunk * 8;
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
            const float p0 = fmaxf(round_bf16(round_bf16(v[0]) + b0), 0.f);
            const float p1 = fmaxf(round_bf16(round_bf16(v[1]) + b1), 0.f);
            const float p2 = fmaxf(round_bf16(round_bf16(v[2]) + b2), 0.f);
            const float p3 = fmaxf(round_bf16(round_bf16(v[3]) + b3), 0.f);
            uint2 o;
            o.x = (__float_as_uint(p0) >> 16) | (__float_as_uint(p1) & 0xffff0000u);
            o.y = (__float_as_uint(p2) >> 16) | (__float_as_uint(p3) & 0xffff0000u);
            *reinterpret_cast<uint2*>(acts + (size_t)b * N + f) = o;
        }
    }
}

// ------------------------------------------------------------------------------------------- K1 generic
// Any shape / dtype (fp32 model dtype, and bf16 shapes whose K is not a multiple of 64).
// 64x64 tile, 16-deep k slices, 4x4 outputs per thread, fp32 FMA.
template <int DT>
__global__ __launch_bounds__(256) void encode_generic_kernel(const void* __restrict__ X, const void* __restrict__ Wt,
                                                             const void* __restrict__ bias, void* __restrict__ acts,
                                                             int M, int N, int K) {
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
            xs[k][r] = (kin && row0 + r < M) ? D_::ld(Xp + (size_t)(row0 + r) * K + k0 + k) : 0.f;
            ws[k][r] = (kin && col0 + r < N) ? D_::ld(Wp + (size_t)(col0 + r) * K + k0 + k) : 0.f;
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
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = row0 + ty * 4 + i;
        if (r >= M) continue;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int c = col0 + tx * 4 + j;
            if (c >= N) continue;
            const float pre = D_::rnd(D_::rnd(acc[i][j]) + D_::ld(bp + c));
            D_::st(out + (size_t)r * N + c, fmaxf(pre, 0.f));
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

extern "C" int mtc_encode_fwd(const mtc_desc* d, const void* x, const void* W_encT, const void* b_enc, void* acts,
                              void* stream) {
    if (check_desc(d, "mtc_encode_fwd")) return -1;
    MTC_REQUIRE(x && W_encT && b_enc && acts, "mtc_encode_fwd: null pointer");
    const int M = d->B, N = d->D, K = d->A_s;
    if (d->dtype == MTC_BF16 && (K % BK) == 0) {
        const int lds = 2 * 2 * TILE_BYTES;
        hipFuncSetAttribute(reinterpret_cast<const void*>(encode_bf16_mfma_kernel),
                            hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        const int mt = (M + BM - 1) / BM, nt = (N + BN - 1) / BN;
        hipLaunchKernelGGL(encode_bf16_mfma_kernel, dim3(mt * nt), dim3(256), lds, as_stream(stream),
                           (const u16*)x, (const u16*)W_encT, (const u16*)b_enc, (u16*)acts, M, N, K, mt, nt);
    } else {
        const dim3 grid((N + 63) / 64, (M + 63) / 64), block(256);
        if (d->dtype == MTC_BF16)
            hipLaunchKernelGGL(encode_generic_kernel<MTC_BF16>, grid, block, 0, as_stream(stream), x, W_encT, b_enc, acts, M, N, K);
        else
            hipLaunchKernelGGL(encode_generic_kernel<MTC_F32>, grid, block, 0, as_stream(stream), x, W_encT, b_enc, acts, M, N, K);
    }
    return check_launch("mtc_encode_fwd");
}
