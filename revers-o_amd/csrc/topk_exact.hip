// The fallback behind the search's exactness certificate (kernels.h): what makes
// "the top-k of an exhaustive fp32 scoring of the gallery" (the reference's brute-force  G @ q ; argsort,
// core_system.py:659-664 -> qdrant local mode) a guarantee and not a likelihood of the bf16 scan.
//
//   finish (topk.hip)     re-scores the scan's ksel candidates in fp32 and checks the certificate; a query that fails
//                         it becomes entry j of the fallback workspace (query index, collect bound, bf16 query row)
//   collect (topk256.hip) one more MFMA pass over the gallery for those entries: every row whose bf16 score reaches
//                         the bound is appended to the entry's list
//   exact_finish          fp32 re-score of every collected row, running best-64 list, the entry's k results; an entry
//                         whose list overflowed is passed on
//   bruteforce            for those: the fp32 score of EVERY row of the gallery (the same fma chain as every other
//                         re-score), per-wave best-64 lists, merged per slice; bruteforce_final merges the slices
//
// Every launch is sized for the worst case and reads the entry count from device memory: workgroups past it exit, so a
// search in which every query is certified pays four empty launches and no host round trip.
#include "kernels.h"
#include "topk_util.h"

namespace revo {

// write one entry's results: `run` = its best keys (fp32 score, row), best first, 0 = empty
__device__ __forceinline__ void exact_write(uint64_t run, int lane, long orow, int k, int has_thr, float thr,
                                            long idx_offset, float* __restrict__ out_scores,
                                            long long* __restrict__ out_idx, int* __restrict__ out_counts) {
    const bool ok = run != 0ull && lane < k && (!has_thr || key_score(run) >= thr);
    const int cnt = __popcll(__ballot(ok));
    if (lane < k) {
        out_scores[orow * k + lane] = ok ? key_score(run) : -INFINITY;
        out_idx[orow * k + lane] = ok ? (long long)key_index(run) + idx_offset : -1ll;
    }
    if (lane == 0) out_counts[orow] = cnt;
}

// ------------------------------------------------------------ exact finish ----
// One workgroup of four waves per entry.  Wave w takes the 64-key chunks w, w + 4, ... of the entry's collect
// list, re-scores them (exact_dot4: four rows in flight) and folds them into its best-64 list; the four lists meet in
// LDS.  The collect list holds every row that can be in the result, each exactly once.
constexpr int XF_WAVES = 4;
__global__ __launch_bounds__(XF_WAVES * 64) void topk_exact_finish_kernel(ExactWs ws, const float* __restrict__ Qf, long ldqf,
                                                                          const float* __restrict__ Gf, long ldgf, int D,
                                                                          int k, int has_thr, float thr, long idx_offset,
                                                                          int force_bruteforce, int out_compact,
                                                                          float* __restrict__ out_scores,
                                                                          long long* __restrict__ out_idx,
                                                                          int* __restrict__ out_counts) {
    __shared__ uint64_t partial[XF_WAVES][64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    // entries 0 .. ctr[0] - 1 were filled by the collect pass, cap - 1 .. cap - ctr[5] by the finish step from the scan's segments
    const int nfront = ws.ctr[0], entries = nfront + ws.ctr[5];
    for (int jj = blockIdx.x; jj < entries; jj += gridDim.x) {
    const int j = jj < nfront ? jj : ws.cap - 1 - (jj - nfront);
    const int n = ws.col_cnt[j];
    if (n > EXACT_COL_CAP || force_bruteforce) {
        if (threadIdx.x == 0) ws.over_j[atomicAdd(ws.ctr + 1, 1)] = j;
        continue;
    }
    if (threadIdx.x == 0) atomicAdd(ws.ctr + 3, n);
    const int q = ws.unc_q[j];
    const float* qr = Qf + (long)q * ldqf;
    const uint64_t* col = ws.col + (long)j * EXACT_COL_CAP;
    uint64_t run = 0ull;
    for (int c0 = w * 64; c0 < n; c0 += XF_WAVES * 64) {
        const int m = (n - c0) < 64 ? (n - c0) : 64;
        const uint64_t key = lane < m ? col[c0 + lane] : 0ull;
        const uint32_t idx = key_index(key);
        float score = -INFINITY;
        for (int e0 = 0; e0 < m; e0 += 4) {
            const float* gr[4];
            float t[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int cn = e0 + u < m ? e0 + u : m - 1;
                gr[u] = Gf + (long)__shfl(idx, cn, 64) * ldgf;
            }
            exact_dot4(qr, gr, D, lane, t);
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (lane == e0 + u && e0 + u < m) score = t[u];
        }
        run = wave_fold_best64(run, lane < m ? make_key(score, idx) : 0ull, lane);
    }
    partial[w][lane] = run;
    __syncthreads();
    if (w == 0) {
#pragma unroll 1
        for (int o = 1; o < XF_WAVES; ++o) {
            const uint64_t rev = partial[o][63 - lane];
            run = wave_bitonic_merge_desc(run > rev ? run : rev, lane);
        }
        exact_write(run, lane, out_compact ? (long)ws.orow[j] : (long)q, k, has_thr, thr, idx_offset, out_scores, out_idx, out_counts);
    }
    __syncthreads();                           // `partial` is reused by the next entry
    }
}
int launch_topk_exact_finish(const ExactWs& ws, int max_entries, const float* Qf, long ldqf, const float* Gf, long ldgf,
                             int D, int k, int has_thr, float thr, long idx_offset, int force_bruteforce, int out_compact,
                             float* out_scores, long long* out_idx, int* out_counts, hipStream_t st) {
    if (max_entries <= 0) return 0;
    REVO_REQUIRE(Gf && k >= 1 && k <= 64, "exact finish: needs the fp32 master rows and 1 <= k <= 64");
    // a bounded launch whatever the search's size (workgroups stride over the entries): an idle pass costs microseconds
    hipLaunchKernelGGL(topk_exact_finish_kernel, dim3((unsigned)(max_entries < 2048 ? max_entries : 2048)), dim3(XF_WAVES * 64), 0, st, ws, Qf, ldqf, Gf,
                       ldgf, D, k, has_thr, thr, idx_offset, force_bruteforce, out_compact, out_scores, out_idx, out_counts);
    REVO_HIP_CHECK(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------- brute force ----
// grid (EXACT_L3_SLICES, up to 64); workgroup (s, y) scores slice s of the gallery for overflowed entries y, y + 64, ...
// 16 waves per workgroup, four rows per wave in flight (16 KB per wave: the pass runs at the rate a CU can pull HBM);
// each wave keeps its best 64 (score, row) keys sorted in one register per lane and inserts a row only if it beats
// the 64th.  The slice's 16 lists meet in LDS; its best 64 go to slot s of the entry's (now useless) collect buffer.
constexpr int BF_WAVES = 16;
__global__ __launch_bounds__(BF_WAVES * 64) void topk_exact_bruteforce_kernel(ExactWs ws, const float* __restrict__ Qf,
                                                                              long ldqf, const float* __restrict__ Gf,
                                                                              long ldgf, long N, int D) {
    __shared__ uint64_t partial[BF_WAVES][64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int over = ws.ctr[1];
    for (int i = blockIdx.y; i < over; i += gridDim.y) {
    const int j = ws.over_j[i];
    const float* qr = Qf + (long)ws.unc_q[j] * ldqf;
    const long per = (N + EXACT_L3_SLICES - 1) / EXACT_L3_SLICES;
    const long r0 = (long)blockIdx.x * per;
    const long r1 = r0 + per < N ? r0 + per : N;
    uint64_t run = 0ull;                       // lane l: the wave's (l+1)-th best key so far
    for (long r = r0 + (long)w * 4; r < r1; r += BF_WAVES * 4) {
        const float* gr[4];
        float t[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) gr[u] = Gf + (r + u < r1 ? r + u : r1 - 1) * ldgf;
        exact_dot4(qr, gr, D, lane, t);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (r + u >= r1) break;                                        // wave-uniform
            const uint64_t key = make_key(t[u], (uint32_t)(r + u));       // wave-uniform value
            const uint64_t worst = readlane_u64(run, 63);
            if (key <= worst) continue;
            const int pos = __popcll(__ballot(run > key));                 // entries that stay ahead of it
            const uint64_t prev = shfl_up1_u64(run);
            run = lane < pos ? run : (lane == pos ? key : prev);
        }
    }
    partial[w][lane] = run;
    __syncthreads();
    if (w == 0) {
#pragma unroll 1
        for (int o = 1; o < BF_WAVES; ++o) {
            const uint64_t rev = partial[o][63 - lane];
            run = wave_bitonic_merge_desc(run > rev ? run : rev, lane);
        }
        ws.col[(long)j * EXACT_COL_CAP + (long)blockIdx.x * 64 + lane] = run;
    }
    __syncthreads();
    }
}
// one wave per overflowed entry: merge the slices' lists, write the results
__global__ __launch_bounds__(256) void topk_exact_bruteforce_final_kernel(ExactWs ws, int k, int has_thr, float thr,
                                                                          long idx_offset, int out_compact,
                                                                          float* __restrict__ out_scores,
                                                                          long long* __restrict__ out_idx,
                                                                          int* __restrict__ out_counts) {
    const int lane = threadIdx.x & 63;
    const int over = ws.ctr[1];
    for (int i = blockIdx.x * 4 + (threadIdx.x >> 6); i < over; i += gridDim.x * 4) {
    const int j = ws.over_j[i];
    const uint64_t* lists = ws.col + (long)j * EXACT_COL_CAP;
    uint64_t run = lists[lane];
#pragma unroll 1
    for (int s = 1; s < EXACT_L3_SLICES; ++s) {
        const uint64_t rev = lists[s * 64 + 63 - lane];
        run = wave_bitonic_merge_desc(run > rev ? run : rev, lane);
    }
    exact_write(run, lane, out_compact ? (long)ws.orow[j] : (long)ws.unc_q[j], k, has_thr, thr, idx_offset, out_scores, out_idx,
                out_counts);
    }
}
int launch_topk_exact_bruteforce(const ExactWs& ws, int max_entries, const float* Qf, long ldqf, const float* Gf, long ldgf,
                                 long N, int D, int k, int has_thr, float thr, long idx_offset, int out_compact,
                                 float* out_scores, long long* out_idx, int* out_counts, hipStream_t st) {
    if (max_entries <= 0 || N <= 0) return 0;
    REVO_REQUIRE(Gf && k >= 1 && k <= 64 && N < (1ll << 32), "exact brute force: needs the fp32 master rows, 1 <= k <= 64, N < 2^32");
    const int ny = max_entries < 64 ? max_entries : 64;
    hipLaunchKernelGGL(topk_exact_bruteforce_kernel, dim3(EXACT_L3_SLICES, (unsigned)ny), dim3(BF_WAVES * 64), 0, st,
                       ws, Qf, ldqf, Gf, ldgf, N, D);
    hipLaunchKernelGGL(topk_exact_bruteforce_final_kernel, dim3((unsigned)((ny + 3) / 4)), dim3(256), 0, st, ws, k,
                       has_thr, thr, idx_offset, out_compact, out_scores, out_idx, out_counts);
    REVO_HIP_CHECK(hipGetLastError());
    return 0;
}

// ------------------------------------------------- entries from an explicit list ----
__global__ __launch_bounds__(256) void topk_exact_prepare_kernel(ExactWs ws, const int* __restrict__ q_idx,
                                                                 const float* __restrict__ need, int n, CertArgs cert, int D,
                                                                 const uint64_t* __restrict__ cand, long cand_stride, int ksel) {
    const int lane = threadIdx.x & 63;
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);        // place in the caller's list = output row
    if (i >= n) return;
    const int q = q_idx[i];
    const float G = __uint_as_float(cert.gstat[0]), Eg = __uint_as_float(cert.gstat[1]);
    const float eps = cert_eps(cert.qstat[(long)q * 2], cert.qstat[(long)q * 2 + 1], G, Eg, D);
    const float nd = need[i];
    float lb = nd - eps;
    lb -= fabsf(lb) * 2.4e-7f;
    if (nd == -INFINITY) lb = -INFINITY;
    // the scan's segments answer if they hold every row at or above lb (topk_util.h); U = this shard's ksel-th best scan score
    bool from_seg = false;
    if (cert.mode == 0 && cert.nsegs > 0 && cand) {
        const uint64_t last = cand[(long)q * cand_stride + ksel - 1];
        from_seg = last != 0ull && cert_segments_cover(cert, q, lb, key_score(last));
    }
    int j = 0;
    if (lane == 0) j = atomicAdd(ws.ctr + (from_seg ? 5 : 0), 1);
    j = __builtin_amdgcn_readfirstlane(j);
    if (from_seg) j = ws.cap - 1 - j;
    if (lane == 0) {
        ws.unc_q[j] = q;
        ws.unc_lb[j] = lb;
        ws.orow[j] = i;
        if (!from_seg) ws.col_cnt[j] = 0;
    }
    if (from_seg) { cert_fill_from_segments(cert, q, lb, j, lane); return; }
    const bf16_t* qs = cert.Qb + (long)q * cert.ldq;
    bf16_t* qd = ws.qb_u + (long)j * ws.ldqb;
    for (int c = lane * 8; c < D; c += 512) *(uint4*)(qd + c) = *(const uint4*)(qs + c);
}
int launch_topk_exact_prepare(const ExactWs& ws, const int* q_idx, const float* need, int n, const CertArgs& cert, int D,
                              const uint64_t* cand, long cand_stride, int ksel, hipStream_t st) {
    if (n <= 0) return 0;
    hipLaunchKernelGGL(topk_exact_prepare_kernel, dim3((unsigned)(n + 3) / 4), dim3(256), 0, st, ws, q_idx, need, n,
                       cert, D, cand, cand_stride, ksel);
    REVO_HIP_CHECK(hipGetLastError());
    return 0;
}

}  // namespace revo
