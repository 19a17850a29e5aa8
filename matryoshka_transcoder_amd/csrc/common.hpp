// Shared device/host helpers for libmtc (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "../../include/mtc.h"

namespace mtc {

void set_error(const char* fmt, ...);
int check_desc(const mtc_desc* d, const char* who);
int check_launch(const char* who);

#define MTC_REQUIRE(cond, ...)                       \
    do {                                             \
        if (!(cond)) {                               \
            mtc::set_error(__VA_ARGS__);             \
            return -1;                               \
        }                                            \
    } while (0)

typedef __bf16 bf16;
typedef uint16_t u16;

// 16-byte vector views
typedef __attribute__((ext_vector_type(8))) short short8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;

__device__ __forceinline__ float bf16_bits_to_f32(uint32_t b) { return __uint_as_float(b << 16); }

// round-to-nearest-even f32 -> bf16 through the gfx950 converter (v_cvt_pk_bf16_f32: one instruction
// per PAIR of values; the hand-rolled add-and-shift costs ~6 VALU ops per value and made the Adam
// and decode epilogues ALU-bound)
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(2))) float f32x2;
__device__ __forceinline__ uint32_t pack2_bf16(float lo, float hi) {
    const f32x2 v = {lo, hi};
    const bf16x2 b = __builtin_convertvector(v, bf16x2);
    return __builtin_bit_cast(uint32_t, b);
}
__device__ __forceinline__ uint32_t f32_to_bf16_bits(float f) {
    return (uint32_t)__builtin_bit_cast(uint16_t, (__bf16)f);
}
__device__ __forceinline__ float round_bf16(float f) { return bf16_bits_to_f32(f32_to_bf16_bits(f)); }

// Model-dtype traits: T is the storage type, everything is computed in float.
template <int DT> struct DType;
template <> struct DType<MTC_F32> {
    typedef float T;
    static constexpr int kVec = 4;          // elements per 16 B
    __device__ static __forceinline__ float rnd(float x) { return x; }
    __device__ static __forceinline__ float ld(const T* p) { return *p; }
    __device__ static __forceinline__ void st(T* p, float v) { *p = v; }
};
template <> struct DType<MTC_BF16> {
    typedef u16 T;
    static constexpr int kVec = 8;
    __device__ static __forceinline__ float rnd(float x) { return round_bf16(x); }
    __device__ static __forceinline__ float ld(const T* p) { return bf16_bits_to_f32(*p); }
    __device__ static __forceinline__ void st(T* p, float v) { *p = (u16)f32_to_bf16_bits(v); }
};

// load/store 8 consecutive elements (a "chunk") as floats
template <int DT> __device__ __forceinline__ void load8(const void* base, size_t elem_off, float (&v)[8]);
template <> __device__ __forceinline__ void load8<MTC_BF16>(const void* base, size_t elem_off, float (&v)[8]) {
    const u32x4 r = *reinterpret_cast<const u32x4*>(reinterpret_cast<const u16*>(base) + elem_off);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        v[2 * i] = __uint_as_float(r[i] << 16);
        v[2 * i + 1] = __uint_as_float(r[i] & 0xffff0000u);
    }
}
template <> __device__ __forceinline__ void load8<MTC_F32>(const void* base, size_t elem_off, float (&v)[8]) {
    const f32x4* p = reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(base) + elem_off);
    const f32x4 a = p[0], b = p[1];
#pragma unroll
    for (int i = 0; i < 4; ++i) { v[i] = a[i]; v[4 + i] = b[i]; }
}
template <int DT> __device__ __forceinline__ void store8(void* base, size_t elem_off, const float (&v)[8]);
template <> __device__ __forceinline__ void store8<MTC_BF16>(void* base, size_t elem_off, const float (&v)[8]) {
    u32x4 r;
#pragma unroll
    for (int i = 0; i < 4; ++i) r[i] = pack2_bf16(v[2 * i], v[2 * i + 1]);
    *reinterpret_cast<u32x4*>(reinterpret_cast<u16*>(base) + elem_off) = r;
}
template <> __device__ __forceinline__ void store8<MTC_F32>(void* base, size_t elem_off, const float (&v)[8]) {
    f32x4* p = reinterpret_cast<f32x4*>(reinterpret_cast<float*>(base) + elem_off);
    f32x4 a, b;
#pragma unroll
    for (int i = 0; i < 4; ++i) { a[i] = v[i]; b[i] = v[4 + i]; }
    p[0] = a; p[1] = b;
}

// streaming (non-temporal) form of store8: for rows nothing reads again soon (Adam moments)
template <int DT> __device__ __forceinline__ void store8_nt(void* base, size_t elem_off, const float (&v)[8]);
template <> __device__ __forceinline__ void store8_nt<MTC_BF16>(void* base, size_t elem_off, const float (&v)[8]) {
    u32x4 r;
#pragma unroll
    for (int i = 0; i < 4; ++i) r[i] = pack2_bf16(v[2 * i], v[2 * i + 1]);
    __builtin_nontemporal_store(r, reinterpret_cast<u32x4*>(reinterpret_cast<u16*>(base) + elem_off));
}
template <> __device__ __forceinline__ void store8_nt<MTC_F32>(void* base, size_t elem_off, const float (&v)[8]) {
    f32x4* p = reinterpret_cast<f32x4*>(reinterpret_cast<float*>(base) + elem_off);
    f32x4 a, b;
#pragma unroll
    for (int i = 0; i < 4; ++i) { a[i] = v[i]; b[i] = v[4 + i]; }
    __builtin_nontemporal_store(a, p); __builtin_nontemporal_store(b, p + 1);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ uint32_t wave_sum_u32(uint32_t v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// float -> order-preserving key for POSITIVE floats is the bit pattern itself.
template <int DT> __device__ __forceinline__ uint32_t key_of(const void* acts, size_t i);
template <> __device__ __forceinline__ uint32_t key_of<MTC_BF16>(const void* acts, size_t i) {
    return (uint32_t)reinterpret_cast<const u16*>(acts)[i] << 16;
}
template <> __device__ __forceinline__ uint32_t key_of<MTC_F32>(const void* acts, size_t i) {
    return reinterpret_cast<const uint32_t*>(acts)[i];
}

// 1 / N as a scalar of the model dtype (host side): fp32, or bf16 with round-to-nearest-even
inline float inv_n_in_dtype(int dtype, double n) {
    const float f = (float)(1.0 / n);
    if (dtype != MTC_BF16) return f;
    uint32_t b;
    __builtin_memcpy(&b, &f, 4);
    b += 0x7fffu + ((b >> 16) & 1u);
    b &= 0xffff0000u;
    float r;
    __builtin_memcpy(&r, &b, 4);
    return r;
}

inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

}  // namespace mtc
