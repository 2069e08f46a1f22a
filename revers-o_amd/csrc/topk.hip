// Gallery cosine similarity + top-k (K12/K13/K14 of SURVEY.md §2b): replaces
// qdrant-client local mode's  scores = G @ q ; argsort ; threshold walk  behind
// vector_db.search (reference call site core_system.py:659-664).
//
//   scan    bf16 MFMA  Q x G^T with the per-query candidate selection fused into
//           the epilogue: the Q x N score matrix never reaches memory.  Every
//           wave owns 32 query rows of the 128-row tile and keeps their current
//           best KSEL candidates as sorted 64-bit keys in LDS; a score is looked
//           at again only if it reaches the row's running KSEL-th best.
//   reduce  merges the per-split candidate lists of a query (bitonic merge).
//   finish  re-scores the KSEL survivors exactly in fp32 against the fp32 master
//           rows, orders them (score desc, index asc), applies the threshold and
//           writes k results.
//   merge   combines per-shard results (multi-GPU, after the RCCL all-gather).
#include "gemm_core.h"
#include "kernels.h"
#include "topk_util.h"

namespace revo {

// ------------------------------------------------------------- the scan ----
// Insert cand into the sorted (best-first) list of one row; returns the row's
// new admission score (score of the KSEL-th entry, or -inf while the list is
// not full).  All 64 lanes execute; lanes >= KSEL are passive.
template <int KSEL>
__device__ __forceinline__ float list_insert(uint64_t* list, uint64_t cand, int lane, float tau_old) {
    const bool in = lane < KSEL;
    const uint64_t e = in ? list[lane] : 0ull;
    const int pos = __popcll(__ballot(in && e > cand));
    if (pos >= KSEL) return tau_old;
    const uint64_t prev = shfl_up1_u64(e);
    const uint64_t ne = lane < pos ? e : (lane == pos ? cand : prev);
    if (in) list[lane] = ne;
    const uint64_t last = readlane_u64(ne, KSEL - 1);
    return last ? key_score(last) : -INFINITY;
}

constexpr int SCAN_BM = 128, SCAN_BN = 128;
constexpr int SCAN_GEMM_LDS = 2 * (SCAN_BM + SCAN_BN) * 128;

template <int KSEL>
__global__ __launch_bounds__(GEMM_THREADS) void topk_scan_kernel(ScanArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int MF = 2, NF = 8;   // wave tile 32 x 128
    uint64_t* lists = (uint64_t*)(smem + SCAN_GEMM_LDS);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const int lr = lane & 15, lq = lane >> 4;
    const int q0 = blockIdx.x * SCAN_BM;
    const int sp = blockIdx.y;

    uint64_t* mylists = lists + (wave * 32) * KSEL;
    for (int i = lane; i < 32 * KSEL; i += 64) mylists[i] = 0ull;

    float tau[MF];
#pragma unroll
    for (int m = 0; m < MF; ++m) tau[m] = (q0 + wave * 32 + m * 16 + lr) < p.Q ? -INFINITY : INFINITY;

    const long tiles = (p.N + SCAN_BN - 1) / SCAN_BN;
    const long per = (tiles + p.splits - 1) / p.splits;
    const long t0 = sp * per;
    const long t1 = (t0 + per) < tiles ? (t0 + per) : tiles;

    TileLoader<SCAN_BM> la;
    la.init(p.Qb, p.ldq, q0, p.Q, wave, lane);

    for (long t = t0; t < t1; ++t) {
        const long n0 = t * SCAN_BN;
        TileLoader<SCAN_BN> lb;
        {
            // gallery rows are addressed with a 64-bit base and a 32-bit in-tile row
            const long left = p.N - n0;
            lb.init(p.Gb + n0 * p.ldg, p.ldg, 0, left < SCAN_BN ? (int)left : SCAN_BN, wave, lane);
        }
        f32x4 acc[MF][NF];
#pragma unroll
        for (int m = 0; m < MF; ++m)
#pragma unroll
            for (int n = 0; n < NF; ++n) acc[m][n] = (f32x4){0.f, 0.f, 0.f, 0.f};
        gemm_mainloop<SCAN_BM, SCAN_BN, MF, NF>(la, lb, smem, p.D, wave, lane, wave * 32, 0, acc);

        const int nvalid = (p.N - n0) < SCAN_BN ? (int)(p.N - n0) : SCAN_BN;
#pragma unroll
        for (int m = 0; m < MF; ++m) {
            float mx = -INFINITY;
#pragma unroll
            for (int n = 0; n < NF; ++n)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int col = n * 16 + lq * 4 + j;
                    if (nvalid < SCAN_BN && col >= nvalid) acc[m][n][j] = -INFINITY;
                    mx = fmaxf(mx, acc[m][n][j]);
                }
            if (__ballot(mx >= tau[m]) == 0ull) continue;
#pragma unroll
            for (int n = 0; n < NF; ++n)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float v = acc[m][n][j];
                    unsigned long long mask = __ballot(v >= tau[m] && v > -INFINITY);
                    while (mask) {
                        const int l = __builtin_ctzll(mask);
                        mask &= mask - 1;
                        const float sv = readlane_f32(v, l);
                        const uint32_t gidx = (uint32_t)(n0 + n * 16 + (l >> 4) * 4 + j);
                        const int rloc = m * 16 + (l & 15);
                        const float cur = readlane_f32(tau[m], l);
                        const float nt = list_insert<KSEL>(mylists + rloc * KSEL, make_key(sv, gidx), lane, cur);
                        if (lr == (l & 15)) tau[m] = nt;
                    }
                }
        }
    }
    // publish this split's candidates: part[q][split][KSEL]
    for (int rloc = 0; rloc < 32; ++rloc) {
        const int q = q0 + wave * 32 + rloc;
        if (q >= p.Q) break;
        if (lane < KSEL) p.part[((long)q * p.splits + sp) * KSEL + lane] = mylists[rloc * KSEL + lane];
    }
}

// Splits chosen so that (query tiles x splits) fills the 256 CUs about once
// or twice (one 96-128 KB workgroup per CU), while each split keeps enough
// tiles to amortise the early, insertion-heavy part of the scan.
int topk_scan_workspace_splits(int Q, long N) {
    const int qtiles = (Q + SCAN_BM - 1) / SCAN_BM;
    const long tiles = (N + SCAN_BN - 1) / SCAN_BN;
    if (qtiles <= 0 || tiles <= 0) return 1;
    const int cus = 256;
    int best = 1;
    double best_eff = -1.0;
    for (int s = 1; s <= 256; ++s) {
        if (s > tiles) break;
        const long per = (tiles + s - 1) / s;
        if (s > 1 && per < 16) break;                 // keep the insertion-heavy start amortised
        const long wgs = (long)qtiles * s;
        const long rounds = (wgs + cus - 1) / cus;
        const double eff = (double)wgs / (double)(rounds * cus);
        if (eff > best_eff + 1e-9) { best_eff = eff; best = s; }
    }
    return best;
}

int launch_topk_scan(const ScanArgs& a, hipStream_t st) {
    REVO_REQUIRE(a.D % GEMM_BK == 0, "search: D must be a multiple of 64");
    REVO_REQUIRE(a.ksel == 32 || a.ksel == 64, "search: ksel must be 32 or 64");
    REVO_REQUIRE(a.N < (1ll << 32), "search: a shard holds at most 2^32 rows");
    REVO_REQUIRE(a.ldq % 8 == 0 && a.ldg % 8 == 0, "search: row strides must be multiples of 8");
    if (a.Q <= 0 || a.N <= 0) return 0;
    dim3 grid((a.Q + SCAN_BM - 1) / SCAN_BM, a.splits), block(GEMM_THREADS);
    if (a.ksel == 32) {
        constexpr int LDS = SCAN_GEMM_LDS + SCAN_BM * 32 * 8;
        REVO_FUNC_LDS(topk_scan_kernel<32>, LDS);
        hipLaunchKernelGGL((topk_scan_kernel<32>), grid, block, LDS, st, a);
    } else {
        constexpr int LDS = SCAN_GEMM_LDS + SCAN_BM * 64 * 8;
        REVO_FUNC_LDS(topk_scan_kernel<64>, LDS);
        hipLaunchKernelGGL((topk_scan_kernel<64>), grid, block, LDS, st, a);
    }
    REVO_HIP_CHECK(hipGetLastError());
    return 0;
}

// ----------------------------------------------------------- the reduce ----
// One workgroup of RW waves per query: wave w merges lists w, w + RW, ... (independent chains,
// so their load latencies overlap), the partial winners meet in LDS and wave 0 merges them.
// (One wave per query walked all lists serially: 142 us for 64 queries x 250 slices.)
constexpr int RW = 8;
template <int KSEL>
__device__ __forceinline__ uint64_t topk_merge_lists(uint64_t run, uint64_t other_reversed, int lane) {
    uint64_t v;
    if (KSEL == 32) v = lane < 32 ? run : other_reversed;                       // best-first then worst-first: bitonic
    else v = run > other_reversed ? run : other_reversed;                       // top 64 of both, bitonic
    v = wave_bitonic_merge_desc(v, lane);
    return lane < KSEL ? v : 0ull;
}
template <int KSEL>
__global__ __launch_bounds__(RW * 64) void topk_reduce_kernel(uint64_t* part, int Q, int splits) {
    __shared__ uint64_t partial[RW][64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int q = blockIdx.x;
    uint64_t* base = part + (long)q * splits * KSEL;
    uint64_t run = 0ull;
    const int rl = 63 - lane < KSEL ? 63 - lane : 0;       // reversed entry read by this lane (lanes < 64 - KSEL: unused)
    if (w < splits) {
        run = lane < KSEL ? base[(long)w * KSEL + lane] : 0ull;
        uint64_t nxt = w + RW < splits ? base[(long)(w + RW) * KSEL + rl] : 0ull;
        for (int s = w + RW; s < splits; s += RW) {
            const uint64_t cur = nxt;
            if (s + RW < splits) nxt = base[(long)(s + RW) * KSEL + rl];   // in flight during the merge
            run = topk_merge_lists<KSEL>(run, cur, lane);
        }
    }
    partial[w][lane] = run;
    __syncthreads();
    if (w == 0) {
#pragma unroll 1
        for (int o = 1; o < RW; ++o) run = topk_merge_lists<KSEL>(run, partial[o][63 - lane], lane);
        if (lane < KSEL) base[lane] = run;
    }
}
int launch_topk_reduce(uint64_t* part, int Q, int splits, int ksel, hipStream_t st) {
    if (Q <= 0 || splits <= 1) return 0;
    dim3 grid(Q), block(RW * 64);
    if (ksel == 32) hipLaunchKernelGGL((topk_reduce_kernel<32>), grid, block, 0, st, part, Q, splits);
    else hipLaunchKernelGGL((topk_reduce_kernel<64>), grid, block, 0, st, part, Q, splits);
    REVO_HIP_CHECK(hipGetLastError());
    return 0;
}

// ----------------------------------------------------------- the finish ----
// WIDE (few queries): one workgroup per query, its four waves share the fp32 re-scores -- a wave re-scores four candidates
// per HBM round trip, 32 candidates are eight round trips in a row for one wave (28 us of a one-query search) and two for four.
// Every candidate keeps its own fma chain: the scores are the same bits.
template <bool WIDE>
__global__ __launch_bounds__(256) void topk_finish_kernel(const uint64_t* __restrict__ part, long part_stride, int ksel,
                                                          const float* __restrict__ Qf, long ldqf,
                                                          const float* __restrict__ Gf, long ldgf, int D, int Q, int k,
                                                          int has_thr, float thr, long idx_offset,
                                                          const uint32_t* __restrict__ all_bounds, int parts, int top_m,
                                                          float* __restrict__ out_scores,
                                                          long long* __restrict__ out_idx, int* __restrict__ out_counts,
                                                          int has_cert, CertArgs cert) {
    __shared__ float wide_sc[WIDE ? 64 : 1];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int q = WIDE ? (int)blockIdx.x : (int)blockIdx.x * 4 + wv;
    if (q >= Q) return;
    const uint64_t key = lane < ksel ? part[(long)q * part_stride + lane] : 0ull;
    // Row-sharded gallery: every shard has published the (bf16-scan) scores of its best top_m candidates for this
    // query.  They belong to distinct gallery rows, so the j-th largest of all of them is a lower bound of the
    // j-th best scan score over the WHOLE gallery: a candidate of this shard below it is not among the global
    // best j.  j = min(64, 2 ksel): twice what the unsharded search re-scores -- the shards together hold that many
    // candidates anyway, the fp32 row gathers still shrink by parts / 2, and with the k-th-to-64th score gap (1.6 x
    // the k-th-to-32nd) the certificate below all but never fails, which saves the whole second round of the
    // protocol (an exact pass over the shard + a third all-gather) that ~5 of 10 000 queries otherwise trigger.
    uint32_t bound = 0u;
    if (all_bounds) {
        uint32_t run = 0u;                                       // best 64 so far, descending
        const int total = parts * top_m;
        for (int base = 0; base < total; base += 64) {
            const int e = base + lane;
            uint32_t v = 0u;
            if (e < total) v = all_bounds[((long)(e / top_m) * Q + q) * top_m + (e % top_m)];
            v = wave_sort_desc_u32(v, lane);
            const uint32_t rev = __shfl_xor(v, 63, 64);
            uint32_t mx = run > rev ? run : rev;                 // best 64 of both, bitonic
#pragma unroll
            for (int j = 32; j > 0; j >>= 1) {
                const uint32_t o = __shfl_xor(mx, j, 64);
                mx = ((lane & j) == 0) ? (mx > o ? mx : o) : (mx < o ? mx : o);
            }
            run = mx;
        }
        const int brank = 2 * ksel < 64 ? 2 * ksel : 64;
        bound = (uint32_t)__shfl((int)run, brank - 1, 64);     // 0 while fewer than that were published
    }
    // (a scan that started from an estimated admission score, CertArgs::estimated: entries below the estimate -- rows of the
    //  pre-pass list that the estimate overtook -- are not worth a re-score: the certificate counts the estimate as the
    //  score an un-re-scored row may have anyway.  With the estimate in place a shard's list is about as short as the
    //  exchanged bound would make it, so the sharded search can do without that exchange: sharded.py)
    if (has_cert && cert.estimated) {
        const uint32_t eo = cert.tau_base[q];
        bound = eo > bound ? eo : bound;
    }
    const bool valid = key != 0ull && (uint32_t)(key >> 32) >= bound;
    const uint32_t idx = key_index(key);
    float score = valid ? key_score(key) : -INFINITY;
    const unsigned long long vmask = __ballot(valid);
    const int nv = __popcll(vmask);   // valid entries are a prefix (lists are best-first)
    if (Gf) {
        // exact fp32 dot of the normalised fp32 query and gallery rows; fixed summation
        // order: per-lane fma chain over elements lane*4 + 256*i, then a butterfly (exact_dot4).
        const float* qr = Qf + (long)q * ldqf;
        // four candidates at a time: their row reads are independent, so the HBM round trips overlap
        // (each candidate keeps its own fma chain in the same order: the scores do not change)
        for (int cnd0 = WIDE ? wv * 4 : 0; cnd0 < nv; cnd0 += WIDE ? 16 : 4) {
            const float* gr[4];
            float t[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int cn = cnd0 + u < nv ? cnd0 + u : nv - 1;
                gr[u] = Gf + (long)__shfl(idx, cn, 64) * ldgf;
            }
            exact_dot4(qr, gr, D, lane, t);
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (lane == cnd0 + u && cnd0 + u < nv) {
                    if (WIDE) wide_sc[lane] = t[u];
                    else score = t[u];
                }
        }
        if (WIDE) {
            __syncthreads();
            if (wv != 0) return;                   // wave 0 carries on with all the scores
            if (lane < nv) score = wide_sc[lane];
        }
    } else if (WIDE && wv != 0) return;
    uint64_t k2 = valid ? make_key(score, idx) : 0ull;
    k2 = wave_sort_desc(k2, lane);
    const bool ok = k2 != 0ull && lane < k && (!has_thr || key_score(k2) >= thr);
    const int cnt = __popcll(__ballot(ok));   // passing entries are a prefix of the sorted order
    if (lane < k) {
        out_scores[(long)q * k + lane] = ok ? key_score(k2) : -INFINITY;
        out_idx[(long)q * k + lane] = ok ? (long long)key_index(k2) + idx_offset : -1ll;
    }
    if (lane == 0) out_counts[q] = cnt;
    if (!has_cert) return;
    if (!Gf) {                                 // no fp32 master rows: the returned scores are the scan's, nothing to certify
        if (cert.cert_out && lane == 0) cert.cert_out[q] = -INFINITY;
        return;
    }

    // ---- exactness certificate.  U = the largest bf16-scan score a row that was NOT re-scored in fp32 can have:
    // the first candidate dropped by the shard bound if there is one, else the worst kept candidate when the list is
    // full (every row outside it scored no better in the scan), else nothing (the whole gallery was re-scored).
    // need = the fp32 score a row must reach to change the result: the k-th re-scored score, or the threshold when
    // that is higher.  If need > U + eps, with eps >= |scan score - fp32 score| for every row, no such row exists.
    const int ne = __popcll(__ballot(key != 0ull));
    float U = -INFINITY;
    if (nv < ne) U = key_score(readlane_u64(key, nv));
    else if (ne == ksel) U = key_score(readlane_u64(key, ksel - 1));
    // the scan started from an ESTIMATED admission score (topk_select_rows_kernel, est_z): rows below it were dropped
    // unseen, whatever the list holds (a full list whose last entry reaches the estimate proves it was a true bound)
    if (cert.estimated) U = fmaxf(U, orderable_f32(cert.tau_base[q]));
    const float G = __uint_as_float(cert.gstat[0]), Eg = __uint_as_float(cert.gstat[1]);
    const float eps = cert_eps(cert.qstat[(long)q * 2], cert.qstat[(long)q * 2 + 1], G, Eg, D);
    if (cert.cert_out) {                       // row-sharded search: the merge step decides, over all shards
        if (lane == 0) cert.cert_out[q] = U + eps;
        return;
    }
    const uint64_t kth = readlane_u64(k2, k - 1);
    const float sk = kth ? key_score(kth) : -INFINITY;
    const float need = has_thr ? fmaxf(sk, thr) : sk;
    const bool certified = (U == -INFINITY || need > U + eps) && cert.mode != 1 && cert.mode != 2;
    if (lane == 0) atomicAdd(cert.ws.ctr + 2, 1);
    if (certified) return;
    float lb = need - eps;                     // rows that can enter the result have a bf16 score of at least this
    lb -= fabsf(lb) * 2.4e-7f;                 // (the subtraction's own rounding)
    // Can the scan's own segments answer?  It admitted every row scoring at least max(base, t - marg) for bounds
    // t <= U (fl(t - marg) is monotone in t): every row with scan score >= max(base, fl(U - marg)) that is not a
    // pre-pass row is in the query's segments, and every pre-pass row scoring above `base` is in its pre-pass list.
    // If `lb` reaches that level, the rows the collect pass would find are all there: no pass over the gallery.
    const bool from_seg = cert.mode == 0 && ne == ksel && cert_segments_cover(cert, q, lb, U);
    int j = 0;
    if (lane == 0) j = atomicAdd(cert.ws.ctr + (cert.mode == 3 ? 4 : (from_seg ? 5 : 0)), 1);
    if (cert.mode == 3) return;                // counted only
    j = __builtin_amdgcn_readfirstlane(j);
    if (from_seg) j = cert.ws.cap - 1 - j;     // numbered from the back: the collect pass takes the front entries
    if (lane == 0) {
        cert.ws.unc_q[j] = q;
        cert.ws.unc_lb[j] = need == -INFINITY ? -INFINITY : lb;
        if (!from_seg) cert.ws.col_cnt[j] = 0;
    }
    if (!from_seg) {
        const bf16_t* qs = cert.Qb + (long)q * cert.ldq;
        bf16_t* qd = cert.ws.qb_u + (long)j * cert.ws.ldqb;
        for (int c = lane * 8; c < D; c += 512) *(uint4*)(qd + c) = *(const uint4*)(qs + c);
        return;
    }
    cert_fill_from_segments(cert, q, lb, j, lane);
}
__global__ __launch_bounds__(256) void cert_margin_kernel(const float* __restrict__ qstat, const uint32_t* __restrict__ gstat,
                                                          int D, int Q, float* __restrict__ marg, int* __restrict__ dropflag) {
    const int q = blockIdx.x * 256 + threadIdx.x;
    if (q >= Q) return;
    const float eps = cert_eps(qstat[(long)q * 2], qstat[(long)q * 2 + 1], __uint_as_float(gstat[0]), __uint_as_float(gstat[1]), D);
    // 2 eps: the k-th re-scored score is at least U - eps (each of the ksel candidates scores at least U in the scan), so
    // the collect bound need - eps is at least U - 2 eps; a little more for the roundings on the way
    marg[q] = 2.0f * eps * 1.0001f + 1e-7f;
    dropflag[q] = 0;
}
int launch_cert_margin(const float* qstat, const uint32_t* gstat, int D, int Q, float* marg, int* dropflag, hipStream_t st) {
    if (Q <= 0) return 0;
    hipLaunchKernelGGL(cert_margin_kernel, dim3((unsigned)((Q + 255) / 256)), dim3(256), 0, st, qstat, gstat, D, Q, marg, dropflag);
    REVO_HIP_CHECK(hipGetLastError());
    return 0;
}
int launch_topk_finish(const uint64_t* part, long part_stride, int ksel, const float* Qf, long ldqf, const float* Gf,
                       long ldgf, int D, int Q, int k, int has_thr, float thr, long idx_offset, const uint32_t* all_bounds,
                       int parts, int top_m, float* out_scores, long long* out_idx, int* out_counts, const CertArgs* cert,
                       hipStream_t st) {
    REVO_REQUIRE(k >= 1 && k <= ksel && ksel <= 64, "search: need 1 <= k <= ksel <= 64");
    REVO_REQUIRE(!Gf || (D % 4 == 0 && ldqf % 4 == 0 && ldgf % 4 == 0), "search: fp32 rows must be 16-byte aligned");
    if (Q <= 0) return 0;
    CertArgs c{};
    const int has_cert = cert ? 1 : 0;
    if (has_cert) c = *cert;
    if (Q <= 256 && Gf)
        hipLaunchKernelGGL(topk_finish_kernel<true>, dim3(Q), dim3(256), 0, st, part, part_stride, ksel, Qf, ldqf, Gf,
                           ldgf, D, Q, k, has_thr, thr, idx_offset, all_bounds, parts, top_m, out_scores, out_idx, out_counts,
                           has_cert, c);
    else
        hipLaunchKernelGGL(topk_finish_kernel<false>, dim3((Q + 3) / 4), dim3(256), 0, st, part, part_stride, ksel, Qf, ldqf, Gf,
                           ldgf, D, Q, k, has_thr, thr, idx_offset, all_bounds, parts, top_m, out_scores, out_idx, out_counts,
                           has_cert, c);
    REVO_HIP_CHECK(hipGetLastError());
    return 0;
}

// bounds[q][j] = order-preserving u32 score of candidate j of query q (lists are sorted best first; 0 = empty slot)
__global__ __launch_bounds__(256) void topk_publish_kernel(const uint64_t* __restrict__ part, long part_stride, int Q, int top_m,
                                                           uint32_t* __restrict__ bounds) {
    const long t = (long)blockIdx.x * 256 + threadIdx.x;
    if (t >= (long)Q * top_m) return;
    const long q = t / top_m;
    const int j = (int)(t - q * top_m);
    bounds[t] = (uint32_t)(part[q * part_stride + j] >> 32);
}
int launch_topk_publish(const uint64_t* part, long part_stride, int Q, int top_m, uint32_t* bounds, hipStream_t st) {
    if (Q <= 0) return 0;
    const long n = (long)Q * top_m;
    hipLaunchKernelGGL(topk_publish_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, part, part_stride, Q, top_m, bounds);
    REVO_HIP_CHECK(hipGetLastError());
    return 0;
}

// ------------------------------------------- pre-pass: top-KSEL of score rows ----
// scores [Q][n] fp32 (a plain GEMM of the queries against the first n gallery rows) ->
// the KSEL best (score, column) keys of every row, best first, written to
// part[q][slot][KSEL], and tau0[q] = the KSEL-th score (-inf if fewer than KSEL columns).
//
// One workgroup per query, one wave per strip of 4096 columns, the strip held in registers (16 x 16-byte
// loads per lane, all in flight at once; 8192-column strips took 174 VGPRs, two waves per SIMD, and were
// 10 % slower).  Pass 1: every lane takes the maximum of its 64 scores; those 64
// maxima belong to 64 different columns, so their KSEL-th largest is a lower bound T of the strip's KSEL-th
// best score -- one 64-value sort instead of a running list.  Pass 2: the few scores >= T (about 45 of 4096 for
// KSEL = 32) are compacted into LDS and sorted once.  (The previous form walked 64-column chunks against a
// running bound and sorted whenever 64 survivors had gathered, 16 waves per query with a serial 15-step
// merge at the end: 0.59 ms for 10 000 x 8192 scores; this one is bound by reading the scores.)
// sort the wave's compacted survivors, fold them into the running list, raise the bound (a real call: inlined
// 256 times into the unrolled walk over the registers it made the compiler keep the strip in scratch memory)
template <int KSEL>
__device__ __noinline__ void sel_flush(uint64_t& run, float& tau, int& cnt, const uint64_t* cand, int lane) {
    uint64_t x = lane < cnt ? cand[lane] : 0ull;
    x = wave_sort_desc(x, lane);
    const uint64_t rev = shfl_xor_u64(x, 63);
    const uint64_t m2 = run > rev ? run : rev;
    uint64_t r = wave_bitonic_merge_desc(m2, lane);
    if (lane >= KSEL) r = 0ull;
    const uint64_t last = readlane_u64(r, KSEL - 1);
    if (last) tau = fmaxf(tau, key_score(last));
    run = r;
    cnt = 0;
}
constexpr int SEL_STRIP = 4096;     // columns per wave
constexpr int SEL_MAXW = 16;        // waves per query: n <= 65536
constexpr int SEL_NV = SEL_STRIP / 256;   // 16-byte loads per lane
template <int KSEL>
__global__ __launch_bounds__(SEL_MAXW * 64) void topk_select_rows_kernel(const float* __restrict__ scores, long lds_, int n,
                                                                        int Q, uint64_t* __restrict__ part,
                                                                        long part_row_stride, int slot,
                                                                        uint32_t* __restrict__ tau0, uint32_t* __restrict__ hist,
                                                                        int hist_buckets, int hist_shift,
                                                                        uint32_t* __restrict__ tau_copy, float est_z) {
    __shared__ uint64_t partial[SEL_MAXW][64];
    __shared__ float mom[SEL_MAXW][3];          // est_z != 0: per-wave count, sum, sum of squares of the strip's scores
    __shared__ uint64_t cand[SEL_MAXW][128];    // per wave: survivors of pass 2, compacted (up to two sorts' worth)
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int nw = blockDim.x >> 6;
    const int q = blockIdx.x;
    const int c0 = w * SEL_STRIP;
    const float* row = scores + (long)q * lds_ + c0;
    const int left = n - c0;                                   // columns of this strip (may be <= 0 or > SEL_STRIP)
    f32x4 v[SEL_NV];
    float mx = -INFINITY;
#pragma unroll
    for (int i = 0; i < SEL_NV; ++i) {
        const int c = (i * 64 + lane) * 4;
        // n and the row stride are multiples of 4 (checked by the launcher): a 16-byte chunk is all in or all out
        v[i] = c < left ? *(const f32x4*)(row + c) : (f32x4){-INFINITY, -INFINITY, -INFINITY, -INFINITY};
    }
#pragma unroll
    for (int i = 0; i < SEL_NV; ++i) mx = fmaxf(fmaxf(mx, fmaxf(v[i][0], v[i][1])), fmaxf(v[i][2], v[i][3]));
    if (est_z != 0.f) {
        // first two moments of the query's pre-pass scores (for the estimated admission level, below)
        float n_ = 0.f, s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < SEL_NV; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float x = v[i][e];
                const bool ok = x > -INFINITY;
                n_ += ok ? 1.f : 0.f;
                s1 += ok ? x : 0.f;
                s2 = ok ? fmaf(x, x, s2) : s2;
            }
        n_ = wave_sum(n_); s1 = wave_sum(s1); s2 = wave_sum(s2);
        if (lane == 0) { mom[w][0] = n_; mom[w][1] = s1; mom[w][2] = s2; }
    }
    // T = KSEL-th largest of the 64 lane maxima (order-preserving u32; NaN scores cannot occur: unit rows)
    uint32_t o = f32_orderable(mx);
#pragma unroll
    for (int k2 = 2; k2 <= 64; k2 <<= 1) {
#pragma unroll
        for (int j = k2 >> 1; j > 0; j >>= 1) {
            const uint32_t x = __shfl_xor(o, j, 64);
            const bool take_max = (((lane & j) == 0) == ((lane & k2) == 0));
            o = take_max ? (o > x ? o : x) : (o < x ? o : x);
        }
    }
    float tau = orderable_f32((uint32_t)__builtin_amdgcn_readlane((int)o, KSEL - 1));
    uint64_t run = 0ull;
    int cnt = 0;                           // wave-uniform
    // Pass 2.  Usually at most 64 scores of the strip reach T: every lane counts its own, one prefix sum over the
    // lanes gives each its place in the compacted list, and the list is sorted once.  (A ballot per element -- 128 of
    // them, each with a branch -- is what this kernel's time used to go into.)  More than 64 (ties, duplicates): the
    // chunked walk below.
    int mine = 0;
#pragma unroll
    for (int i = 0; i < SEL_NV; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) mine += (v[i][e] >= tau && v[i][e] > -INFINITY) ? 1 : 0;
    int incl = mine;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int t = __shfl_up(incl, o, 64);
        if (lane >= o) incl += t;
    }
    const int total = __builtin_amdgcn_readlane(incl, 63);
    // (up to 128: about 45 of a strip's 4096 scores reach T, 64 is 2.8 sigma away -- with 64 queries some wave of the launch
    //  passed it in most searches and the whole launch waited for its chunked walk: 28 us against 16 for one query)
    if (total <= 128) {
        int pos = incl - mine;
#pragma unroll
        for (int i = 0; i < SEL_NV; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float sv = v[i][e];
                if (sv >= tau && sv > -INFINITY) cand[w][pos++] = make_key(sv, (uint32_t)(c0 + (i * 64 + lane) * 4 + e));
            }
        cnt = total;
        if (total > 64) {
            cnt = 64;
            sel_flush<KSEL>(run, tau, cnt, cand[w], lane);          // the first 64; the rest below, from cand[w] + 64
            cnt = total - 64;
            sel_flush<KSEL>(run, tau, cnt, cand[w] + 64, lane);
        }
    } else {
#pragma unroll
    for (int i = 0; i < SEL_NV; ++i) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float sv = v[i][e];
            const bool take = sv >= tau && sv > -INFINITY;
            const unsigned long long mask = __ballot(take);
            if (mask == 0ull) continue;
            const int add = __popcll(mask);
            if (cnt + add > 64) sel_flush<KSEL>(run, tau, cnt, cand[w], lane);   // (entries kept under the old tau stay valid candidates)
            if (take) cand[w][cnt + __popcll(mask & ((1ull << lane) - 1ull))] = make_key(sv, (uint32_t)(c0 + (i * 64 + lane) * 4 + e));
            cnt += add;
            if (cnt == 64) sel_flush<KSEL>(run, tau, cnt, cand[w], lane);
        }
    }
    }
    if (cnt > 0) sel_flush<KSEL>(run, tau, cnt, cand[w], lane);
    partial[w][lane] = run;
    __syncthreads();
    if (w == 0) {
#pragma unroll 1
        for (int ow = 1; ow < nw; ++ow) {
            const uint64_t rev = partial[ow][63 - lane];
            const uint64_t m2 = run > rev ? run : rev;
            run = wave_bitonic_merge_desc(m2, lane);
            if (lane >= KSEL) run = 0ull;
        }
        if (lane < KSEL) part[(long)q * part_row_stride + (long)slot * KSEL + lane] = run;
        const uint64_t last = readlane_u64(run, KSEL - 1);
        uint32_t base = f32_orderable(last ? key_score(last) : -INFINITY);      // order-preserving u32
        if (est_z != 0.f) {
            // One shard of a row-sharded gallery (api.hip, revo_search_set_total_rows): start the scan not from this shard's
            // own KSEL-th best of the pre-pass rows but from an ESTIMATE of the score that the candidates of the WHOLE
            // gallery will have to reach -- mean + z sigma of this query's pre-pass scores, z from the quantile
            // min(64, 2 KSEL) / total rows (a Gaussian tail: what unit vectors in many dimensions give; a heavier tail
            // only makes the estimate low, which costs survivors, not results).  It is not a bound: the finish step counts
            // it as the score an unseen row of this shard can have (CertArgs::estimated), and the merge's certificate and
            // second round make the result exact whatever the estimate was.
            float n_ = 0.f, s1 = 0.f, s2 = 0.f;
            for (int ow = 0; ow < nw; ++ow) { n_ += mom[ow][0]; s1 += mom[ow][1]; s2 += mom[ow][2]; }
            if (n_ > 1.f) {
                const float mu = s1 / n_;
                const float var = fmaxf(s2 / n_ - mu * mu, 0.f);
                const uint32_t eo = f32_orderable(mu + est_z * sqrtf(var));
                base = eo > base ? eo : base;
            }
        }
        if (lane == 0) { tau0[q] = base; if (tau_copy) tau_copy[q] = base; }     // tau_copy: the scan's live bounds start here
        // seed the query's score histogram (origin = this bound) with the kept scores: the scan counts its
        // survivors into the same buckets, so "KSEL scores at or above an edge" includes the pre-pass rows
        if (hist && last && lane < KSEL && run && (uint32_t)(run >> 32) >= base) {
            uint32_t b = ((uint32_t)(run >> 32) - base) >> hist_shift;
            b = b < (uint32_t)(hist_buckets - 1) ? b : (uint32_t)(hist_buckets - 1);
            atomicAdd(hist + (long)q * hist_buckets + b, 1u);
        }
    }
}
int launch_topk_select_rows(const float* scores, long ld, int n, int Q, uint64_t* part, long part_row_stride, int slot,
                            uint32_t* tau0, int ksel, uint32_t* hist, int hist_buckets, int hist_shift, hipStream_t st,
                            uint32_t* tau_copy, float est_z) {
    if (Q <= 0) return 0;
    REVO_REQUIRE(n >= 1 && n <= SEL_STRIP * SEL_MAXW, "search: the pre-pass selection takes at most 65536 columns");
    REVO_REQUIRE(n % 4 == 0 && ld % 4 == 0 && (((uintptr_t)scores) & 15) == 0, "search: pre-pass score rows must be 16-byte aligned");
    const int nw = (n + SEL_STRIP - 1) / SEL_STRIP;
    if (ksel == 32)
        hipLaunchKernelGGL((topk_select_rows_kernel<32>), dim3(Q), dim3(nw * 64), 0, st, scores, ld, n, Q, part,
                           part_row_stride, slot, tau0, hist, hist_buckets, hist_shift, tau_copy, est_z);
    else
        hipLaunchKernelGGL((topk_select_rows_kernel<64>), dim3(Q), dim3(nw * 64), 0, st, scores, ld, n, Q, part,
                           part_row_stride, slot, tau0, hist, hist_buckets, hist_shift, tau_copy, est_z);
    REVO_HIP_CHECK(hipGetLastError());
    return 0;
}

// ------------------------------------------------- empty-gallery result ----
__global__ void topk_fill_empty_kernel(float* s, long long* i, int* c, int Q, int k) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < Q * k) { s[t] = -INFINITY; i[t] = -1ll; }
    if (t < Q) c[t] = 0;
}
int launch_topk_fill_empty(float* s, long long* i, int* c, int Q, int k, hipStream_t st) {
    const int n = Q * k;
    if (n <= 0) return 0;
    hipLaunchKernelGGL(topk_fill_empty_kernel, dim3((n + 255) / 256), dim3(256), 0, st, s, i, c, Q, k);
    REVO_HIP_CHECK(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------ the merge ----
// [P][Q][k] per-shard results (global indices, -inf/-1 padded) -> [Q][k].
// Global row ids are 64-bit here (a sharded gallery may hold more than 2^32 rows in total), so the merge
// orders 96-bit keys: (order-preserving score, ~index) -- score descending, then index ascending --
// held as  a = orderable(score) << 32 | high word of ~index,  b = low word of ~index.
// (the key lives in two plain registers per lane: as a struct it was kept in scratch memory and every step of the
//  sorting network went through it -- 194 us for 10 000 queries x 8 parts instead of ~20)
#define KEY96_LESS(xa, xb, ya, yb) ((xa) < (ya) || ((xa) == (ya) && (xb) < (yb)))
// one compare-exchange step with the lane at distance j; take_max: this lane keeps the larger key
#define KEY96_STEP(a, b, j, take_max)                                     \
    do {                                                                  \
        const uint64_t oa_ = shfl_xor_u64((a), (j));                      \
        const uint32_t ob_ = __shfl_xor((b), (j), 64);                    \
        const bool less_ = KEY96_LESS((a), (b), oa_, ob_);                \
        const bool swap_ = (take_max) == less_;                           \
        (a) = swap_ ? oa_ : (a);                                          \
        (b) = swap_ ? ob_ : (b);                                          \
    } while (0)
__global__ __launch_bounds__(256) void topk_merge_kernel(const float* __restrict__ scores, long score_part_stride,
                                                         const long long* __restrict__ idx, long idx_part_stride, int P, int Q, int k,
                                                         int has_thr, float thr, float* __restrict__ out_scores,
                                                         long long* __restrict__ out_idx, int* __restrict__ out_counts,
                                                         MergeCert mc) {
    const int lane = threadIdx.x & 63;
    const int q = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (q >= Q) return;
    const int total = P * k;
    uint64_t ra = 0ull;                        // running best 64, descending; all-zero = empty slot (a real entry has a
    uint32_t rb = 0u;                          // non-zero score word)
    for (int base = 0; base < total; base += 64) {
        const int e = base + lane;
        uint64_t va = 0ull;
        uint32_t vb = 0u;
        if (e < total) {
            const int pz = e / k, j = e - pz * k;
            const long off = (long)q * k + j;
            const long long gi = idx[(long)pz * idx_part_stride + off];
            if (gi >= 0) {
                const uint64_t ni = ~(uint64_t)gi;
                va = ((uint64_t)f32_orderable(scores[(long)pz * score_part_stride + off]) << 32) | (ni >> 32);
                vb = (uint32_t)ni;
            }
        }
        // sort the chunk, largest in lane 0
#pragma unroll
        for (int k2 = 2; k2 <= 64; k2 <<= 1) {
#pragma unroll
            for (int j = k2 >> 1; j > 0; j >>= 1) KEY96_STEP(va, vb, j, (((lane & j) == 0) == ((lane & k2) == 0)));
        }
        if (base == 0) {
            ra = va; rb = vb;                                       // first chunk: nothing to merge with
        } else {
            // chunk worst-first against the running list best-first: the element-wise maximum is the best 64 of both, bitonic
            const uint64_t wa = shfl_xor_u64(va, 63);
            const uint32_t wb = __shfl_xor(vb, 63, 64);
            const bool less = KEY96_LESS(ra, rb, wa, wb);
            ra = less ? wa : ra;
            rb = less ? wb : rb;
#pragma unroll
            for (int j = 32; j > 0; j >>= 1) KEY96_STEP(ra, rb, j, ((lane & j) == 0));
        }
    }
    const float sc = orderable_f32((uint32_t)(ra >> 32));
    const bool ok = (ra != 0ull || rb != 0u) && lane < k && (!has_thr || sc >= thr);
    const int cnt = __popcll(__ballot(ok));
    if (lane < k) {
        out_scores[(long)q * k + lane] = ok ? sc : -INFINITY;
        out_idx[(long)q * k + lane] = ok ? (long long)~((ra << 32) | (uint64_t)rb) : -1ll;
    }
    if (lane == 0) out_counts[q] = cnt;
    if (!mc.cert) return;
    // Exactness certificate over all shards (kernels.h): shard p published U_p + eps_p, the best fp32 score any of its
    // rows that were not re-scored can have.  If the score a row needs to change the merged result exceeds all of them,
    // the result is that of an exhaustive fp32 scoring; else the query goes to a second, collecting round on every shard.
    const uint32_t kth_hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(ra >> 32), k - 1);
    const uint32_t kth_lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)ra, k - 1);
    const uint32_t kth_b = (uint32_t)__builtin_amdgcn_readlane((int)rb, k - 1);
    const float sk = (kth_hi | kth_lo | kth_b) ? orderable_f32(kth_hi) : -INFINITY;
    const float need = has_thr ? fmaxf(sk, thr) : sk;
    float mx = -INFINITY;
    for (int pz = lane; pz < P; pz += 64) mx = fmaxf(mx, mc.cert[(long)pz * mc.cert_part_stride + q]);
    mx = wave_max(mx);
    if (mx == -INFINITY || need > mx) return;
    if (lane == 0) {
        const int j = atomicAdd(mc.unc_count, 1);
        mc.unc_q[j] = q;
        mc.unc_need[j] = need;
    }
}
#undef KEY96_STEP
#undef KEY96_LESS
int launch_topk_merge_strided(const float* scores, long score_part_stride, const long long* idx, long idx_part_stride, int P,
                              int Q, int k, int has_thr, float thr, float* out_scores, long long* out_idx, int* out_counts,
                              hipStream_t st, const MergeCert* mc) {
    REVO_REQUIRE(k >= 1 && k <= 64, "merge: need 1 <= k <= 64");
    REVO_REQUIRE(P >= 1, "merge: need at least one part");
    if (Q <= 0) return 0;
    MergeCert m{};
    if (mc) m = *mc;
    hipLaunchKernelGGL(topk_merge_kernel, dim3((Q + 3) / 4), dim3(256), 0, st, scores, score_part_stride, idx, idx_part_stride,
                       P, Q, k, has_thr, thr, out_scores, out_idx, out_counts, m);
    REVO_HIP_CHECK(hipGetLastError());
    return 0;
}
int launch_topk_merge(const float* scores, const long long* idx, int P, int Q, int k, int has_thr, float thr,
                      float* out_scores, long long* out_idx, int* out_counts, hipStream_t st) {
    return launch_topk_merge_strided(scores, (long)Q * k, idx, (long)Q * k, P, Q, k, has_thr, thr, out_scores, out_idx,
                                     out_counts, st);
}

}  // namespace revo
