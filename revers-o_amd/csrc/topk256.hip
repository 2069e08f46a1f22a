// Query x gallery scan on the 256 x 256 MFMA main loop (gemm256_core.h) with the top-k
// candidate selection fused behind it: the form used for every gallery that is not tiny.
//
// One workgroup = 256 queries x one contiguous slice of the gallery, walked in 256-row
// tiles.  After each tile a lane compares its 128 scores with the admission score of its
// 8 query rows (running KSEL-th best: a lower bound of the final one, so nothing that can
// end up in the top KSEL is ever dropped).  Survivors are rare after the first tiles; they
// are appended to a 2048-entry LDS queue as 64-bit entries  row | score | ~index.  When the
// queue holds 1792 entries (or the slice ends) the workgroup drains it: a bitonic sort of the queue groups the
// entries by query row (best first), and each row's best <= KSEL entries are merged into
// that row's candidate list, which lives in global memory (L2 resident, touched only at
// drains) because the 128 KiB main-loop image leaves no room for 256 lists in LDS.  The
// admission scores are seeded by a pre-pass over the first rows of the gallery (topk.hip),
// so even the first tile admits only a handful of entries per row.
// The next tile's first DMA is issued before the selection runs, so its HBM latency is
// hidden behind the compares.
#include "gemm256_core.h"
#include "kernels.h"

namespace revo {

constexpr int S256_QCAP = 2048;          // queue entries
constexpr int S256_DRAIN = 1792;         // drain once this many are queued (256 slots of slack for the next tile)
constexpr int S256_LDS = G256_LDS + S256_QCAP * 8 + 256 * 4 * 3 + 64 + 8 * 128 * 8 + 256 * 8;   // + queue, tau/start/end, ctrl, merge scratch, wkey

__device__ __forceinline__ uint64_t s256_entry(int row, float score, uint32_t relidx) {
    return ((uint64_t)row << 56) | ((uint64_t)f32_orderable(score) << 24) | (uint64_t)((~relidx) & 0xffffffu);
}
__device__ __forceinline__ uint64_t s256_entry_to_key(uint64_t e, uint32_t idx_base) {
    const uint32_t rel = (~(uint32_t)e) & 0xffffffu;
    const uint32_t ord = (uint32_t)(e >> 24);
    return ((uint64_t)ord << 32) | (uint64_t)(~(idx_base + rel));
}

__device__ __forceinline__ uint64_t s256_shfl_xor(uint64_t v, int m) {
    const uint32_t lo = __shfl_xor((uint32_t)v, m, 64), hi = __shfl_xor((uint32_t)(v >> 32), m, 64);
    return ((uint64_t)hi << 32) | lo;
}
__device__ __forceinline__ uint64_t s256_shfl_up1(uint64_t v) {
    const uint32_t lo = __shfl_up((uint32_t)v, 1, 64), hi = __shfl_up((uint32_t)(v >> 32), 1, 64);
    return ((uint64_t)hi << 32) | lo;
}

struct S256Lds {
    uint64_t* queue;     // [S256_QCAP]
    float* tau;          // [256] admission score per query row of the tile
    int* start;          // [256]
    int* end;            // [256]
    int* ctrl;           // [0] queue count (may exceed the capacity: overflow), [1..] spare
    uint64_t* scratch;   // [8][64] wave-private
    uint64_t* wkey;      // [256] key of the row's KSEL-th list entry (0 while the list is not full)
};

__device__ __forceinline__ uint32_t s256_sort_desc_u32(uint32_t v, int lane) {
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
__device__ __forceinline__ uint32_t s256_merge_desc_u32(uint32_t v, int lane) {   // v bitonic across the wave
#pragma unroll
    for (int j = 32; j > 0; j >>= 1) {
        const uint32_t o = __shfl_xor(v, j, 64);
        v = ((lane & j) == 0) ? (v > o ? v : o) : (v < o ? v : o);
    }
    return v;
}

// Cross-slice admission bound.  Every slice of a query publishes the scores of its best `top_m` list
// entries (gtop[q][slice][top_m], order-preserving u32, written at drains).  Those are distinct gallery
// rows, so the KSEL-th largest of their union is a lower bound of the query's final KSEL-th best score --
// and a tight one: the global top KSEL is spread over the slices, few slices hold more than top_m of it.
// One slice's own KSEL-th best only reaches the KSEL / slice-rows quantile (measured: ~6x more queued
// entries than necessary at 32 slices).  All 512 threads; one wave per row, rows strided by 8.
template <int KSEL>
__device__ __noinline__ void s256_refresh_bounds(const S256Lds& L, const uint32_t* gtop, int q0, int qvalid, int nvals,
                                                 int tid, unsigned long long* stats) {
    const int wave = tid >> 6, lane = tid & 63;
    for (int r = wave; r < qvalid; r += 8) {
        const uint32_t* row = gtop + (long)(q0 + r) * nvals;
        uint32_t run = 0u;                                        // lanes 0..31: best KSEL so far, descending
        for (int base = 0; base < nvals; base += 128) {
            const int i0 = base + lane * 2;
            uint32_t v0 = 0u, v1 = 0u;
            if (i0 < nvals) v0 = __hip_atomic_load(row + i0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (i0 + 1 < nvals) v1 = __hip_atomic_load(row + i0 + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const uint32_t a = s256_sort_desc_u32(v0 > v1 ? v0 : v1, lane);
            const uint32_t b = s256_sort_desc_u32(v0 > v1 ? v1 : v0, lane);
            // best 32 of the chunk: a[0..31] descending next to b[31..0] ascending is bitonic
            const uint32_t brev = __shfl(b, 63 - lane, 64);      // all lanes take part: a shuffle reads 0 from inactive lanes
            if (KSEL == 32) {
                uint32_t x = lane < 32 ? a : brev;
                x = s256_merge_desc_u32(x, lane);
                const uint32_t xrev = __shfl(x, 63 - lane, 64);
                uint32_t y = lane < 32 ? run : xrev;
                y = s256_merge_desc_u32(y, lane);
                run = lane < 32 ? y : 0u;
            } else {
                // best 64 of the 128: the element-wise maximum of a descending and an ascending run is bitonic
                uint32_t x = a > brev ? a : brev;
                x = s256_merge_desc_u32(x, lane);
                const uint32_t xrev = __shfl(x, 63 - lane, 64);
                uint32_t y = run > xrev ? run : xrev;
                run = s256_merge_desc_u32(y, lane);
            }
        }
        const uint32_t bound = (uint32_t)__builtin_amdgcn_readlane((int)run, KSEL - 1);
        if (lane == 0 && bound != 0u) {
            const float b = orderable_f32(bound);
            if (stats) atomicAdd(stats + (b > L.tau[r] ? 4 : 5), 1ull);
            if (b > L.tau[r]) L.tau[r] = b;
        } else if (lane == 0 && stats) {
            atomicAdd(stats + 6, 1ull);
        }
    }
    __syncthreads();
}

// All 512 threads.  Sort the queue (row, score, index descending), merge every row's best
// entries into its global list, refresh the admission scores, empty the queue.
template <int KSEL>
__device__ __noinline__ void s256_drain(const S256Lds& L, uint64_t* part, long part_row_stride, int q0, int qvalid,
                                        uint32_t idx_base, int tid, uint32_t* tau_g, uint32_t* gtop_mine, int gtop_stride,
                                        int top_m) {
    const int wave = tid >> 6, lane = tid & 63;
    __syncthreads();
    int n = L.ctrl[0];
    n = n < S256_QCAP ? n : S256_QCAP;
    int np2 = 64;
    while (np2 < n) np2 <<= 1;
    for (int i = n + tid; i < np2; i += 512) L.queue[i] = 0ull;
    if (tid < 256) { L.start[tid] = 0; L.end[tid] = 0; }
    __syncthreads();
    // bitonic sort, descending, np2 entries: comparator c touches (lo, lo | j)
    for (int k2 = 2; k2 <= np2; k2 <<= 1) {
        for (int j = k2 >> 1; j > 0; j >>= 1) {
            for (int c = tid; c < (np2 >> 1); c += 512) {
                const int lo = ((c & ~(j - 1)) << 1) | (c & (j - 1));
                const int hi = lo | j;
                const uint64_t a = L.queue[lo], b = L.queue[hi];
                const bool desc = (lo & k2) == 0;
                if (desc ? (a < b) : (a > b)) { L.queue[lo] = b; L.queue[hi] = a; }
            }
            __syncthreads();
        }
    }
    // row boundaries (entries of one row are contiguous, best first)
    for (int i = tid; i < n; i += 512) {
        const int r = (int)(L.queue[i] >> 56);
        if (i == 0 || (int)(L.queue[i - 1] >> 56) != r) L.start[r] = i;
        if (i == n - 1 || (int)(L.queue[i + 1] >> 56) != r) L.end[r] = i + 1;
    }
    __syncthreads();
    uint64_t* ws = L.scratch + wave * 128;
    // the lists live in global memory (L2): the next row's list is requested before the current row is
    // merged, so the round trips of a wave's ~32 rows overlap instead of adding up
    uint64_t nxt = 0ull;
    if (wave < qvalid && lane < KSEL) nxt = part[(long)(q0 + wave) * part_row_stride + lane];
    for (int r = wave; r < qvalid; r += 8) {
        uint64_t cur = nxt;                                     // lanes 0..KSEL-1: the row's list, best first
        if (r + 8 < qvalid && lane < KSEL) nxt = part[(long)(q0 + r + 8) * part_row_stride + lane];
        const int s0 = L.start[r];
        const int cnt = L.end[r] - s0;
        if (cnt <= 0) continue;
        uint64_t* list = part + (long)(q0 + r) * part_row_stride;
        // The row's queued entries are merged KSEL at a time, duplicates removed after every merge: a tile
        // that is computed again after a queue overflow re-queues entries the list already holds, and
        // truncating the queue side before de-duplication could push new entries out.
        for (int off = 0; off < cnt; off += KSEL) {
            const int c = (cnt - off) < KSEL ? (cnt - off) : KSEL;
            const uint64_t best = s256_entry_to_key(L.queue[s0 + off], idx_base);
            const uint64_t worst_kept = __builtin_amdgcn_readlane((uint32_t)cur, KSEL - 1) |
                                        ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((uint32_t)(cur >> 32), KSEL - 1) << 32);
            if (off > 0 && worst_kept != 0ull && best < worst_kept) break;   // nothing further down can enter
            if (KSEL == 32) {
                uint64_t v;
                if (lane < 32) {
                    v = cur;
                } else {
                    const int jx = 63 - lane;                   // lane 63 takes the chunk's best entry
                    v = jx < c ? s256_entry_to_key(L.queue[s0 + off + jx], idx_base) : 0ull;
                }
                // lanes 0..31 best-first, lanes 32..63 worst-first: bitonic -> sorted best-first
#pragma unroll
                for (int j = 32; j > 0; j >>= 1) {
                    const uint64_t o = s256_shfl_xor(v, j);
                    v = ((lane & j) == 0) ? (v > o ? v : o) : (v < o ? v : o);
                }
                const uint64_t prev = s256_shfl_up1(v);
                const bool keep = v != 0ull && (lane == 0 || v != prev);
                const unsigned long long km = __ballot(keep);
                const int pos = __popcll(km & ((1ull << lane) - 1ull));
                ws[lane] = 0ull;
                if (keep) ws[pos] = v;            // LDS operations of one wave complete in order
                cur = lane < KSEL ? ws[lane] : 0ull;
            } else {
                // 64-entry list + up to 64 queued entries: a 128-element bitonic merge held in two registers.
                // hi = element-wise maximum (the best 64, bitonic), lo = minimum (the rest, bitonic); both are
                // sorted, duplicates dropped across the whole 128, and the first 64 survivors are the new list.
                const int jx = 63 - lane;
                const uint64_t q1 = jx < c ? s256_entry_to_key(L.queue[s0 + off + jx], idx_base) : 0ull;   // worst-first
                uint64_t hi = cur > q1 ? cur : q1, lo = cur > q1 ? q1 : cur;
#pragma unroll
                for (int j = 32; j > 0; j >>= 1) {
                    const uint64_t oh = s256_shfl_xor(hi, j), ol = s256_shfl_xor(lo, j);
                    hi = ((lane & j) == 0) ? (hi > oh ? hi : oh) : (hi < oh ? hi : oh);
                    lo = ((lane & j) == 0) ? (lo > ol ? lo : ol) : (lo < ol ? lo : ol);
                }
                const uint64_t hprev = s256_shfl_up1(hi), lprev = s256_shfl_up1(lo);
                const uint64_t hlast = __builtin_amdgcn_readlane((uint32_t)hi, 63) |
                                       ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((uint32_t)(hi >> 32), 63) << 32);
                const bool keep_h = hi != 0ull && (lane == 0 || hi != hprev);
                const bool keep_l = lo != 0ull && lo != (lane == 0 ? hlast : lprev);
                const unsigned long long mh = __ballot(keep_h), ml = __ballot(keep_l);
                const unsigned long long below = (1ull << lane) - 1ull;
                const int nh = __popcll(mh);
                ws[lane] = 0ull;
                ws[64 + lane] = 0ull;
                if (keep_h) ws[__popcll(mh & below)] = hi;
                if (keep_l) ws[nh + __popcll(ml & below)] = lo;
                cur = ws[lane];
            }
        }
        if (lane < KSEL) list[lane] = cur;
        if (lane == KSEL - 1) L.wkey[r] = cur;                // 0 while the list is not full
        if (lane < top_m)                                       // publish this slice's best scores (see s256_refresh_bounds)
            __hip_atomic_store(gtop_mine + (long)(q0 + r) * gtop_stride + lane, (uint32_t)(cur >> 32), __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
        const uint32_t last = (uint32_t)__builtin_amdgcn_readlane((uint32_t)(cur >> 32), KSEL - 1);
        if (lane == 0 && last != 0u) {
            // this slice's KSEL-th best is a lower bound of the query's final KSEL-th best: publish it
            // for the other slices of the same query (stale reads only admit a few more candidates)
            const uint32_t seen = __hip_atomic_fetch_max(tau_g + q0 + r, last, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const float t = orderable_f32(seen > last ? seen : last);
            if (t > L.tau[r]) L.tau[r] = t;
        }
    }
    __syncthreads();
    if (tid == 0) L.ctrl[0] = 0;
    __syncthreads();
}

struct Scan256Args {
    const bf16_t* Qb; long ldq;
    const bf16_t* Gb; long ldg;
    int Q; long N; int D;
    long n_begin;                 // rows [0, n_begin) are covered by the pre-pass
    int splits;
    uint64_t* part;               // [Q][lists_per_query][KSEL]; this kernel owns slots [0, splits)
    int lists_per_query;
    uint32_t* tau_g;              // [Q] shared admission scores (order-preserving u32 of the score), seeded by the pre-pass
    int dbg;                      // timing experiments only: 1 = skip the selection (results are wrong)
    uint32_t* gtop;               // [Q][splits][top_m] best list scores of every slice (zeroed before the launch)
    int top_m;
    unsigned long long* stats;    // optional counters: [0] drains, [1] queued entries, [2] retry passes, [3] fragments scanned slowly
};

// ROWS: 0 = all 256 query rows of a tile may be valid; 64 / 128 = the whole search has at most that many
// queries (one query tile), and the main loop skips the MFMA work of the rows that cannot be valid.
template <int KSEL, int ROWS>
__global__ __launch_bounds__(G256_THREADS, 2) void topk_scan256_kernel(Scan256Args p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    S256Lds L;
    L.queue = (uint64_t*)(smem + G256_LDS);
    L.tau = (float*)(smem + G256_LDS + S256_QCAP * 8);
    L.start = (int*)(L.tau + 256);
    L.end = L.start + 256;
    L.ctrl = L.end + 256;
    L.scratch = (uint64_t*)(L.ctrl + 16);
    L.wkey = L.scratch + 8 * 128;

    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int lane = tid & 63;
    const int q0 = blockIdx.x * 256;
    const int sp = blockIdx.y;
    const int qvalid = (p.Q - q0) < 256 ? (p.Q - q0) : 256;

    const long span = p.N - p.n_begin;
    const long tiles = (span + 255) / 256;
    const long per = (tiles + p.splits - 1) / p.splits;
    const long t0 = sp * per;
    const long t1 = (t0 + per) < tiles ? (t0 + per) : tiles;
    const long row_begin = p.n_begin + t0 * 256;            // first gallery row of this slice
    const uint32_t idx_base = (uint32_t)row_begin;
    uint64_t* mypart = p.part + (long)sp * KSEL;
    const long part_row_stride = (long)p.lists_per_query * KSEL;

    if (tid < 256) L.tau[tid] = tid < qvalid ? orderable_f32(p.tau_g[q0 + tid]) : INFINITY;
    if (tid < 256) L.wkey[tid] = 0ull;
    if (tid == 0) L.ctrl[0] = 0;
    __syncthreads();
    if (t0 >= t1) return;

    G256Operand A, B;
    g256_operand_init(A, p.Qb, p.ldq, p.Q, q0, wave, lane);
    g256_operand_init(B, p.Gb + row_begin * p.ldg, p.ldg, p.N - row_begin, 0, wave, lane);
    g256_issue_prologue(A, B, smem, p.D, wave);

    // Normal mode: one pass per tile (groups == 1).  If a pass admits more entries than the queue holds
    // (a badly seeded or adversarially ordered gallery), the tile is recomputed in 2, 4, ... 32 column
    // groups, one pass and one drain per group; at 32 groups a pass can admit at most 256 x 8 = 2048
    // entries, so the retry always terminates.  Entries queued twice are removed when lists are merged.
    long t = t0;
    int groups = 1, grp = 0;
    // (Measured with the debug counters: ~8100 entries are queued per workgroup at Q = 10k, N = 1M, 32
    //  slices.  That number is set by how tight a bound ONE slice's KSEL-th best can give, about the
    //  KSEL / slice-rows quantile; draining earlier or more often does not lower it.)
    const int drain_thr = S256_DRAIN;
    while (t < t1) {
        const long n0 = p.n_begin + t * 256;
        {
            f32x4 acc[8][4];
#pragma unroll
            for (int m = 0; m < 8; ++m)
#pragma unroll
                for (int n = 0; n < 4; ++n) acc[m][n] = (f32x4){0.f, 0.f, 0.f, 0.f};
            // few queries: most of the tile's MFMA work would multiply zero rows (separate kernel
            // instantiations: inside one kernel a second main loop costs the main path its register allocation)
            gemm256_mainloop<ROWS>(A, B, smem, p.D, wave, lane, acc);

            if (groups == 1 && t + 1 < t1) {
                // next gallery tile: rebased descriptors (any gallery size), DMA in flight during the selection
                g256_operand_init(B, p.Gb + (n0 + 256) * p.ldg, p.ldg, p.N - (n0 + 256), 0, wave, lane);
                g256_issue_prologue(A, B, smem, p.D, wave);
            }
            asm volatile("" : "+v"(lane) :: "memory");
            const int lr = lane & 15, lq = lane >> 4;
            const int rbase = (wave >> 2) * 128 + lr;            // + m * 16
            const int cbase = (wave & 3) * 64 + lq * 4;          // + n * 16 + j
            const long left = p.N - n0;
            const uint32_t rel0 = (uint32_t)(n0 - row_begin);
            float taum[8];
            unsigned hitm = 0;             // bit m: some lane of this wave has a candidate in row fragment m
            if (p.dbg & 1) {
                asm volatile("" :: "v"(acc[0][0]), "v"(acc[7][3]));
            } else
#pragma unroll
            for (int m = 0; m < 8; ++m) {
                taum[m] = L.tau[rbase + m * 16];
                float mx = -INFINITY;
#pragma unroll
                for (int n = 0; n < 4; ++n)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        // columns past the gallery end become NaN: never admitted, ignored by fmaxf
                        if (left < 256 && cbase + n * 16 + j >= left) acc[m][n][j] = __builtin_nanf("");
                        mx = fmaxf(mx, acc[m][n][j]);
                    }
                if (__ballot(mx >= taum[m]) != 0ull) hitm |= 1u << m;
            }
            if (hitm && !(p.dbg & 4)) {
#pragma unroll
                for (int m = 0; m < 8; ++m) {
                    if (!(hitm & (1u << m))) continue;          // wave-uniform
                    if (p.stats && lane == 0) atomicAdd(p.stats + 3, 1ull);
                    // the row's current KSEL-th entry as (score, index); an empty slot admits everything
                    const uint64_t wk = L.wkey[rbase + m * 16];
                    const float ws = wk ? key_score(wk) : -INFINITY;
                    const uint32_t widx = wk ? key_index(wk) : 0xffffffffu;
#pragma unroll
                    for (int n = 0; n < 4; ++n)
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const float v = acc[m][n][j];
                            const int col = cbase + n * 16 + j;
                            // admission score (shared across slices) first, then the strict test against the
                            // row's own KSEL-th entry: an equal score enters only with a smaller index, so
                            // ties cannot keep the queue full forever.  Survivors are rare: each lane queues
                            // its own (one LDS atomic per survivor) under a mostly empty exec mask.
                            if (v >= taum[m] && (col & (groups - 1)) == grp &&
                                (v > ws || (v == ws && idx_base + rel0 + col < widx))) {
                                const int pos = atomicAdd(&L.ctrl[0], 1);
                                if (pos < S256_QCAP) L.queue[pos] = s256_entry(rbase + m * 16, v, rel0 + col);
                            }
                        }
                }
            }
        }
        // the accumulators are dead from here on (the drain is a real call)
        __syncthreads();
        const int qc = L.ctrl[0];
        // pick up what the other slices of these queries have learnt meanwhile (ordered before the next
        // selection by the barriers of the next main loop)
        // (every fourth tile: the load is an L2 round trip that four waves would otherwise sit on after every tile)
        if ((t & 3) == 0 && tid < qvalid) {
            const float tg = orderable_f32(__hip_atomic_load(p.tau_g + q0 + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
            if (tg > L.tau[tid]) L.tau[tid] = tg;
        }
        const bool overflow = qc > S256_QCAP;
        if (groups == 1 && !overflow && (t & 15) == 15)
            s256_refresh_bounds<KSEL>(L, p.gtop, q0, qvalid, p.splits * p.top_m, tid, p.stats);
        if (p.stats && tid == 0) {
            if (overflow || groups > 1) atomicAdd(p.stats + 2, 1ull);
            if (overflow || groups > 1 || qc >= drain_thr || t + 1 >= t1) { atomicAdd(p.stats + 0, 1ull); atomicAdd(p.stats + 1, (unsigned long long)qc); }
        }
        if (groups == 1 && !overflow) {
            if (qc >= drain_thr || t + 1 >= t1) {
                s256_drain<KSEL>(L, mypart, part_row_stride, q0, qvalid, idx_base, tid, p.tau_g, p.gtop + sp * p.top_m, p.splits * p.top_m,
                           p.top_m);
            }
            ++t;
            continue;
        }
        // retry mode (or entering it): merge what was queued, then recompute this tile / its next column group
        s256_drain<KSEL>(L, mypart, part_row_stride, q0, qvalid, idx_base, tid, p.tau_g, p.gtop + sp * p.top_m, p.splits * p.top_m,
                           p.top_m);
        if (overflow) {
            groups = groups < 32 ? groups * 2 : 32;
            grp = 0;
        } else if (++grp == groups) {
            groups = 1;
            grp = 0;
            ++t;                                                  // tile complete
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // a DMA issued for another tile must not land on top
        __syncthreads();
        if (t < t1) {
            const long nn = p.n_begin + t * 256;
            g256_operand_init(B, p.Gb + nn * p.ldg, p.ldg, p.N - nn, 0, wave, lane);
            g256_issue_prologue(A, B, smem, p.D, wave);
        }
    }
}

static int g_scan_dbg = 0;
static unsigned long long* g_scan_stats = nullptr;
unsigned long long* topk_scan256_stats() {
    if (!g_scan_stats) { (void)hipMalloc((void**)&g_scan_stats, 64); (void)hipMemset(g_scan_stats, 0, 64); }
    return g_scan_stats;
}
void topk_scan256_set_debug(int d) { g_scan_dbg = d; }
// per-slice scores published for the cross-slice bound: enough that splits x top_m comfortably exceeds KSEL
int topk_scan256_top_m(int splits, int ksel) { return (splits >= 128 ? 1 : (splits > 32 ? 2 : 4)) * (ksel / 32); }

int launch_topk_scan256(const bf16_t* Qb, long ldq, const bf16_t* Gb, long ldg, int Q, long N, int D, long n_begin,
                        int splits, uint64_t* part, int lists_per_query, uint32_t* tau_g, uint32_t* gtop, int ksel,
                        hipStream_t st) {
    REVO_REQUIRE(ksel == 32 || ksel == 64, "search: the 256 x 256 scan keeps 32 or 64 candidates per query");
    REVO_REQUIRE(D % 64 == 0 && ldq % 8 == 0 && ldg % 8 == 0, "search: D must be a multiple of 64");
    REVO_REQUIRE(N < (1ll << 32), "search: a shard holds at most 2^32 rows");
    REVO_REQUIRE(256l * ldg * 2 < (1l << 31) && 256l * ldq * 2 < (1l << 31), "search: row too long for the DMA window");
    const long tiles = (N - n_begin + 255) / 256;
    const long per = (tiles + splits - 1) / splits;
    REVO_REQUIRE(per * 256 <= (1l << 24), "search: a gallery slice holds at most 2^24 rows; use more splits");
    if (Q <= 0 || N <= n_begin) return 0;
    Scan256Args a{Qb, ldq, Gb, ldg, Q, N, D, n_begin, splits, part, lists_per_query, tau_g, g_scan_dbg, gtop,
                  topk_scan256_top_m(splits, ksel), (g_scan_dbg & 2) ? topk_scan256_stats() : nullptr};
    const dim3 grid((Q + 255) / 256, splits), block(G256_THREADS);
#define S256_LAUNCH(KS, RW)                                                                                    \
    do {                                                                                                       \
        static bool attr_ = false;                                                                             \
        if (!attr_) {                                                                                          \
            REVO_HIP_CHECK(hipFuncSetAttribute((const void*)topk_scan256_kernel<KS, RW>,                        \
                                               hipFuncAttributeMaxDynamicSharedMemorySize, S256_LDS));         \
            attr_ = true;                                                                                      \
        }                                                                                                      \
        hipLaunchKernelGGL((topk_scan256_kernel<KS, RW>), grid, block, S256_LDS, st, a);                       \
    } while (0)
    const int rows_mode = Q <= 64 ? 64 : (Q <= 128 ? 128 : 0);
    if (ksel == 32) {
        if (rows_mode == 64) S256_LAUNCH(32, 64);
        else if (rows_mode == 128) S256_LAUNCH(32, 128);
        else S256_LAUNCH(32, 0);
    } else {
        if (rows_mode == 64) S256_LAUNCH(64, 64);
        else if (rows_mode == 128) S256_LAUNCH(64, 128);
        else S256_LAUNCH(64, 0);
    }
#undef S256_LAUNCH
    REVO_HIP_CHECK(hipGetLastError());
    return 0;
}

// (query tiles x splits) should fill 256 CUs in whole rounds; slices of >= 32 tiles amortise the drains
int topk_scan256_splits(int Q, long rows) {
    const int qtiles = (Q + 255) / 256;
    const long tiles = (rows + 255) / 256;
    if (tiles <= 0) return 1;
    int best = 1;
    double best_score = -1.0;
    for (int s = 1; s <= 512; ++s) {
        if (s > tiles) break;
        const long per = (tiles + s - 1) / s;
        if (s > 1 && per < 3) break;         // (short slices are fine: the pre-pass and the cross-slice bound seed the admission scores)
        const long wgs = (long)qtiles * s;
        const long rounds = (wgs + 255) / 256;
        const double eff = (double)wgs / (double)(rounds * 256);
        // prefer full rounds; among equals prefer fewer, longer slices (fewer lists, fewer drains)
        const double score = eff - 1e-4 * s;
        if (score > best_score) { best_score = score; best = s; }
    }
    return best;
}

}  // namespace revo
