// Wave-level helpers shared by the search kernels (topk.hip, topk_exact.hip): 64-bit shuffles, bitonic sorts of one
// key per lane, and the exact fp32 re-score of candidate rows (one fixed summation order everywhere a score that is
// RETURNED is computed, so the fast path, the collect path and the brute-force path give the same bits).
#pragma once
#include "common.h"

namespace revo {

// --------------------------------------------------- wave-level sorting ----
__device__ __forceinline__ uint64_t shfl_xor_u64(uint64_t v, int m) {
    const uint32_t lo = __shfl_xor((uint32_t)v, m, 64), hi = __shfl_xor((uint32_t)(v >> 32), m, 64);
    return ((uint64_t)hi << 32) | lo;
}
__device__ __forceinline__ uint64_t shfl_up1_u64(uint64_t v) {
    const uint32_t lo = __shfl_up((uint32_t)v, 1, 64), hi = __shfl_up((uint32_t)(v >> 32), 1, 64);
    return ((uint64_t)hi << 32) | lo;
}
__device__ __forceinline__ uint64_t readlane_u64(uint64_t v, int l) {
    const uint32_t lo = __builtin_amdgcn_readlane((uint32_t)v, l), hi = __builtin_amdgcn_readlane((uint32_t)(v >> 32), l);
    return ((uint64_t)hi << 32) | lo;
}
__device__ __forceinline__ float readlane_f32(float v, int l) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l));
}
// sort the 64 lane values, best (largest key) in lane 0
__device__ __forceinline__ uint64_t wave_sort_desc(uint64_t v, int lane) {
#pragma unroll
    for (int k2 = 2; k2 <= 64; k2 <<= 1) {
#pragma unroll
        for (int j = k2 >> 1; j > 0; j >>= 1) {
            const uint64_t o = shfl_xor_u64(v, j);
            const bool desc = (lane & k2) == 0;
            const bool lower = (lane & j) == 0;
            const bool take_max = (lower == desc);
            v = take_max ? (v > o ? v : o) : (v < o ? v : o);
        }
    }
    return v;
}
// v is bitonic across the wave -> sorted, best in lane 0
__device__ __forceinline__ uint64_t wave_bitonic_merge_desc(uint64_t v, int lane) {
#pragma unroll
    for (int j = 32; j > 0; j >>= 1) {
        const uint64_t o = shfl_xor_u64(v, j);
        v = ((lane & j) == 0) ? (v > o ? v : o) : (v < o ? v : o);
    }
    return v;
}

// the 64 lane values sorted, largest in lane 0 (32-bit)
__device__ __forceinline__ uint32_t wave_sort_desc_u32(uint32_t v, int lane) {
#pragma unroll
    for (int k2 = 2; k2 <= 64; k2 <<= 1) {
#pragma unroll
        for (int j = k2 >> 1; j > 0; j >>= 1) {
            const uint32_t o = __shfl_xor(v, j, 64);
            const bool take_max = (((lane & j) == 0) == ((lane & k2) == 0));
            v = take_max ? (v > o ? v : o) : (v < o ? v : o);
        }
    }
    return v;
}

// run: the best 64 keys so far, best first (0 = empty); v: 64 more keys in any order.  Returns the best 64 of both.
__device__ __forceinline__ uint64_t wave_fold_best64(uint64_t run, uint64_t v, int lane) {
    v = wave_sort_desc(v, lane);
    const uint64_t rev = shfl_xor_u64(v, 63);            // worst first
    const uint64_t m2 = run > rev ? run : rev;           // element-wise maximum of a descending and an ascending list: the best 64, bitonic
    return wave_bitonic_merge_desc(m2, lane);
}

// Exact fp32 scores of up to four gallery rows against one query row: per-lane fma chain over the elements
// lane*4 + 256*i (+0..3, in that order), then the xor butterfly.  The four rows' loads are independent (their HBM
// round trips overlap); every row keeps its own chain, so a row's score does not depend on what it is batched with.
__device__ __forceinline__ void exact_dot4(const float* __restrict__ qr, const float* const (&gr)[4], int D, int lane,
                                           float (&out)[4]) {
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    for (int c = lane * 4; c < D; c += 256) {
        const f32x4 a = *(const f32x4*)(qr + c);
        f32x4 b[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) b[u] = *(const f32x4*)(gr[u] + c);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            acc[u] = fmaf(a[0], b[u][0], acc[u]);
            acc[u] = fmaf(a[1], b[u][1], acc[u]);
            acc[u] = fmaf(a[2], b[u][2], acc[u]);
            acc[u] = fmaf(a[3], b[u][3], acc[u]);
        }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) out[u] = wave_sum(acc[u]);
}

}  // namespace revo
