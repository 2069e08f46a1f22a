// Wave-level helpers shared by the search kernels (topk.hip, topk_exact.hip): 64-bit shuffles, bitonic sorts of one
// key per lane, and the exact fp32 re-score of candidate rows (one fixed summation order everywhere a score that is
// RETURNED is computed, so the fast path, the collect path and the brute-force path give the same bits).
#pragma once
#include "common.h"
#include "kernels.h"

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

// ------------------------------------- uncertified queries answered by the scan's own segments (CertArgs, kernels.h) ----
// Can the segments of query q hold every row whose scan score reaches lb?  The scan admitted every row scoring at least
// max(base, fl(t - marg)) for bounds t <= U (U = the query's ksel-th best scan score; fl(t - marg) is monotone in t), so
// every row at or above max(base, fl(U - marg)) that is not a pre-pass row is in the segments, and every pre-pass row
// scoring above `base` is in the pre-pass list -- unless a drain or a recomputed tile touched the query (dropflag).
__device__ __forceinline__ bool cert_segments_cover(const CertArgs& cert, int q, float lb, float U) {
    if (cert.nsegs <= 0 || !(U > -INFINITY) || !(lb > -INFINITY)) return false;
    const float basef = orderable_f32(cert.tau_base[q]);
    return lb > basef && lb >= U - cert.marg[q] && cert.dropflag[q] == 0;
}
// One wave: entry j's list := every key of query q's pre-pass list and segments whose scan score reaches lb.
__device__ __forceinline__ void cert_fill_from_segments(const CertArgs& cert, int q, float lb, int j, int lane) {
    const SegSrc sg = (cert.nsegs > 1 && q >= cert.segs[1].q0) ? cert.segs[1] : cert.segs[0];
    const int SEG = 2 * cert.seg_ksel;
    const long ql = q - sg.q0;
    const int* cq = sg.cnt + ql * sg.splits;
    const uint64_t* sq = sg.seg + ql * (long)sg.splits * SEG;
    uint64_t* col = cert.ws.col + (long)j * EXACT_COL_CAP;
    const uint32_t lbk = f32_orderable(lb);
    const unsigned long long below = (1ull << lane) - 1ull;
    int n = 0;                                 // wave-uniform
    {
        const uint64_t pk = lane < cert.seg_ksel ? cert.prelist[(long)q * cert.seg_ksel + lane] : 0ull;
        const bool take = pk != 0ull && (uint32_t)(pk >> 32) >= lbk;
        const unsigned long long m = __ballot(take);
        if (take) col[__popcll(m & below)] = pk;
        n = __popcll(m);
    }
    for (int sb = 0; sb < sg.splits; sb += 64) {
        const int my = sb + lane;
        int c = my < sg.splits ? cq[my] : 0;
        c = c < SEG ? c : SEG;
        int mc = c;                            // the longest of these 64 segments
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { const int t = __shfl_xor(mc, o, 64); mc = mc > t ? mc : t; }
        for (int e = 0; e < mc; ++e) {
            const uint64_t key = e < c ? sq[(long)my * SEG + e] : 0ull;
            const bool take = key != 0ull && (uint32_t)(key >> 32) >= lbk;
            const unsigned long long m = __ballot(take);
            if (m == 0ull) continue;
            const int pos = n + __popcll(m & below);
            if (take && pos < EXACT_COL_CAP) col[pos] = key;
            n += __popcll(m);
        }
    }
    if (lane == 0) cert.ws.col_cnt[j] = n;     // more than the list holds: the exact finish passes the entry on (brute force)
}

}  // namespace revo
