// Activation-store batch assembly (transcoder_activation_store.py:157-207).  The reference wraps the device
// buffers in a TensorDataset + DataLoader(shuffle=True): per-sample host indexing and a host-side collate of B
// device rows per batch.  Here a batch is ONE launch: a wave per output row copies row perm[i] of the source
// buffer and of the target buffer (16-byte accesses; the rows of a pair travel together, so pairs cannot come
// apart).  HBM-bound: reads B (A_s + A_t) s bytes of random rows, writes the same.
#include <algorithm>
#include "common.hpp"

namespace mtc {

__global__ __launch_bounds__(256) void gather_row_pairs_kernel(int n, int row_bytes_s, int row_bytes_t, long long buf_rows,
                                                               const char* __restrict__ src_buf, const char* __restrict__ tgt_buf,
                                                               const long long* __restrict__ perm, char* __restrict__ out_s,
                                                               char* __restrict__ out_t, int* __restrict__ bad) {
    const int lane = threadIdx.x & 63;
    const int i = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
    if (i >= n) return;
    const long long r = perm[i];
    if (r < 0 || r >= buf_rows) {                      // never read outside the buffers: flag it, write zeros
        if (lane == 0 && bad) atomicAdd(bad, 1);
        for (int b = lane * 16; b < row_bytes_s; b += 1024) *reinterpret_cast<u32x4*>(out_s + (size_t)i * row_bytes_s + b) = u32x4{0, 0, 0, 0};
        for (int b = lane * 16; b < row_bytes_t; b += 1024) *reinterpret_cast<u32x4*>(out_t + (size_t)i * row_bytes_t + b) = u32x4{0, 0, 0, 0};
        return;
    }
    const char* ps = src_buf + (size_t)r * row_bytes_s;
    const char* pt = tgt_buf + (size_t)r * row_bytes_t;
    for (int b = lane * 16; b < row_bytes_s; b += 1024)
        *reinterpret_cast<u32x4*>(out_s + (size_t)i * row_bytes_s + b) = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(ps + b));
    for (int b = lane * 16; b < row_bytes_t; b += 1024)
        *reinterpret_cast<u32x4*>(out_t + (size_t)i * row_bytes_t + b) = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(pt + b));
}

// Streaming copy, 16 bytes per lane: the HBM calibration the roofline of the gather / stream kernels is read against
// (SURVEY 8d "confirm with a copy-kernel calibration run"), and the contiguous refill copy of the activation store.
// ONE 16-byte element per thread and as many workgroups as that takes: measured on a 1 GiB copy (tools/copy_exp.py,
// round 3), workgroups that walk the buffer with a grid-wide stride reach 4.2-5.1 TB/s (read + write) at 1 K-64 K
// workgroups, whatever the unroll or cache policy, while consecutive workgroups on consecutive 4 KiB reach 6.2 (plain) -
// 6.7 TB/s (non-temporal loads and stores): neighbouring workgroups then share DRAM pages.
__global__ __launch_bounds__(256) void copy16_kernel(const u32x4* __restrict__ src, u32x4* __restrict__ dst, size_t n16) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n16) __builtin_nontemporal_store(__builtin_nontemporal_load(src + i), dst + i);
}

}  // namespace mtc

using namespace mtc;

extern "C" int mtc_copy16(void* dst, const void* src, size_t bytes, void* stream) {
    MTC_REQUIRE(dst && src, "mtc_copy16: null pointer");
    MTC_REQUIRE(bytes % 16 == 0 && ((uintptr_t)dst % 16) == 0 && ((uintptr_t)src % 16) == 0,
                "mtc_copy16: size and pointers must be multiples of 16 bytes");
    if (bytes == 0) return 0;
    const size_t n16 = bytes / 16;
    MTC_REQUIRE(n16 / 256 < 0x7fffffffull, "mtc_copy16: at most 2^31 workgroups of 4 KiB");
    const unsigned grid = (unsigned)((n16 + 255) / 256);
    hipLaunchKernelGGL(copy16_kernel, dim3(grid), dim3(256), 0, as_stream(stream), reinterpret_cast<const u32x4*>(src),
                       reinterpret_cast<u32x4*>(dst), n16);
    return check_launch("mtc_copy16");
}

extern "C" int mtc_gather_row_pairs(int32_t dtype, int32_t n_rows, int32_t A_s, int32_t A_t, int64_t buffer_rows,
                                    const void* source_buffer, const void* target_buffer, const int64_t* perm,
                                    void* out_source, void* out_target, int32_t* bad_index_count, void* stream) {
    MTC_REQUIRE(dtype == MTC_F32 || dtype == MTC_BF16, "mtc_gather_row_pairs: bad dtype");
    MTC_REQUIRE(n_rows >= 0 && A_s > 0 && A_t > 0 && buffer_rows > 0, "mtc_gather_row_pairs: bad sizes");
    if (n_rows == 0) return 0;
    MTC_REQUIRE(source_buffer && target_buffer && perm && out_source && out_target, "mtc_gather_row_pairs: null pointer");
    const int s = dtype == MTC_BF16 ? 2 : 4;
    MTC_REQUIRE((A_s * s) % 16 == 0 && (A_t * s) % 16 == 0,
                "mtc_gather_row_pairs: rows must be a multiple of 16 bytes (A_s = %d, A_t = %d)", A_s, A_t);
    hipLaunchKernelGGL(gather_row_pairs_kernel, dim3((n_rows + 3) / 4), dim3(256), 0, as_stream(stream), n_rows, A_s * s, A_t * s,
                       (long long)buffer_rows, reinterpret_cast<const char*>(source_buffer),
                       reinterpret_cast<const char*>(target_buffer), reinterpret_cast<const long long*>(perm),
                       reinterpret_cast<char*>(out_source), reinterpret_cast<char*>(out_target), bad_index_count);
    return check_launch("mtc_gather_row_pairs");
}
