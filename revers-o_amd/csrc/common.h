// Shared device/host helpers for librevo (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef uint16_t bf16_t;  // raw bf16 bits
typedef __attribute__((ext_vector_type(8))) short s16x8;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

#define REVO_WAVE 64

// ---------------------------------------------------------------- bf16 -----
__device__ __forceinline__ float bf16_to_f32(bf16_t v) {
    return __uint_as_float(((uint32_t)v) << 16);
}
// round-to-nearest-even; the plain cast lowers to v_cvt_pk_bf16_f32 and keeps NaN a NaN
__device__ __forceinline__ bf16_t f32_to_bf16(float f) {
    __bf16 b = (__bf16)f;
    return __builtin_bit_cast(bf16_t, b);
}
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
// one v_cvt_pk_bf16_f32 (two scalar casts + shift + or cost four VALU instructions instead)
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
    const f32x2 v = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2));
}

// ------------------------------------------------------- wave reductions ---
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
// N sums at once (N a power of two <= 32): lane l returns the wave's total of value idx -- with the addition tree of wave_sum,
// so the same bits -- where idx = bit 5 of l + 2 x bit 4 + 4 x bit 3 ... (one index bit per halving); lanes that differ only in
// the bits below the last halving hold the same value.  N - 1 + log2(64 / N) shuffles instead of 6 N.
template <int N>
__device__ __forceinline__ float wave_sum_transposed(float (&v)[N], int lane, int& idx) {
    static_assert(N >= 2 && N <= 32 && (N & (N - 1)) == 0, "a power of two, at most 32");
    idx = 0;
    int bit = 32, mul = 1;
#pragma unroll
    for (int n = N; n > 1; n >>= 1) {
        const bool up = (lane & bit) != 0;
#pragma unroll
        for (int i = 0; i < n / 2; ++i) {
            const float keep = up ? v[2 * i + 1] : v[2 * i];
            const float send = up ? v[2 * i] : v[2 * i + 1];
            v[i] = keep + __shfl_xor(send, bit, 64);
        }
        idx += up ? mul : 0;
        mul <<= 1;
        bit >>= 1;
    }
#pragma unroll
    for (; bit > 0; bit >>= 1) v[0] += __shfl_xor(v[0], bit, 64);
    return v[0];
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// --------------------------------------------- order-preserving score key --
// 64-bit key whose unsigned order is (score desc... larger key = better):
// high word = order-preserving map of the fp32 score, low word = ~index so that
// among equal scores the SMALLER index gives the LARGER key (score desc, index asc).
__device__ __forceinline__ uint32_t f32_orderable(float f) {
    uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float orderable_f32(uint32_t o) {
    uint32_t u = (o & 0x80000000u) ? (o & 0x7fffffffu) : ~o;
    return __uint_as_float(u);
}
__device__ __forceinline__ uint64_t make_key(float score, uint32_t idx) {
    return ((uint64_t)f32_orderable(score) << 32) | (uint64_t)(~idx);
}
__device__ __forceinline__ float key_score(uint64_t k) { return orderable_f32((uint32_t)(k >> 32)); }
__device__ __forceinline__ uint32_t key_index(uint64_t k) { return ~(uint32_t)k; }
// key 0 == "empty slot": worse than any real (score, idx) pair (a real pair has
// a non-zero high word unless score is -NaN-all-ones, which never occurs).

// ------------------------------------------------------------- host side ---
#include <string>
void revo_set_error(const std::string& msg);
#define REVO_HIP_CHECK(expr)                                                         \
    do {                                                                             \
        hipError_t _e = (expr);                                                      \
        if (_e != hipSuccess) {                                                      \
            revo_set_error(std::string(#expr) + ": " + hipGetErrorString(_e));       \
            return -1;                                                               \
        }                                                                            \
    } while (0)
#define REVO_REQUIRE(cond, msg)                                                      \
    do {                                                                             \
        if (!(cond)) {                                                               \
            revo_set_error(std::string("requirement failed: ") + msg);               \
            return -2;                                                               \
        }                                                                            \
    } while (0)

// Opt-in to more than 64 KiB of dynamic LDS for a kernel.  The attribute is per DEVICE, so the
// "already done" flag is kept per device (a process may hold handles on several GPUs).
#define REVO_MAX_DEVICES 64
#define REVO_FUNC_LDS(fn, bytes)                                                                   \
    do {                                                                                           \
        static bool done_[REVO_MAX_DEVICES] = {};                                                  \
        int dev_ = 0;                                                                              \
        REVO_HIP_CHECK(hipGetDevice(&dev_));                                                       \
        if (dev_ < 0 || dev_ >= REVO_MAX_DEVICES || !done_[dev_]) {                                \
            REVO_HIP_CHECK(hipFuncSetAttribute((const void*)(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (bytes))); \
            if (dev_ >= 0 && dev_ < REVO_MAX_DEVICES) done_[dev_] = true;                          \
        }                                                                                          \
    } while (0)
