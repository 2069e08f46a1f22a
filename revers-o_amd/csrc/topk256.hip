// Query x gallery scan on the 256 x 256 MFMA main loop (gemm256_core.h) with the top-k
// candidate selection fused behind it: the form used for every gallery that is not tiny.
//
// One workgroup = 256 queries x one contiguous slice of the gallery, walked in 256-row
// tiles.  After each tile a lane compares its 128 scores with the admission score of its
// 8 query rows (a lower bound of the query's final KSEL-th best score, so nothing that can
// end up in the top KSEL is ever dropped).  Survivors are rare; a survivor is staged in LDS
// (row | score | index), and after the workgroup's barrier thread i moves staged entry i
//   * to its row's SEGMENT of this slice in global memory ([Q][slices][2 KSEL] keys, slot from
//     a per-row LDS counter, fire-and-forget store: no sorting in the scan), and
//   * into the query's global score HISTOGRAM (64 buckets of 2^17 fp32 ulps above the pre-pass
//     bound, shared by all slices of the query: one non-returning L2 atomic)
// -- at most one store and one atomic instruction per wave and tile, which the next main
// loop's operand-DMA waits never have to wait for.  Every few tiles a workgroup re-reads the
// histograms of its 256 queries: the lower edge of the highest bucket with KSEL or more
// scores at or above it is a valid admission bound that reflects what ALL slices have seen so
// far -- close to the bound a sequential scan would have (measured: ~4x fewer survivors than
// per-slice running bounds give, and no sorted lists to maintain).  A separate kernel picks
// each query's best KSEL out of its segments.  Only when a row's segment is full do its
// survivors go to a 1024-entry LDS queue, which is drained by sorting it and merging each
// row's entries with its segment (the slow, exact path: adversarially ordered galleries, huge
// tie groups); if the queue or the staging buffer overflows, the tile is recomputed by column
// groups.  The admission scores are seeded by a pre-pass over the first rows of the gallery
// (topk.hip).  The next tile's first DMA is issued before the selection runs, so its HBM
// latency is hidden behind the compares.
#include "gemm256_core.h"
#include "kernels.h"

namespace revo {

constexpr int S256_QCAP = 1024;          // overflow-queue entries (survivors whose row segment is full)
constexpr int S256_DRAIN = 512;          // drain once this many are queued
constexpr int S256_STG = 1024;           // staging entries: the survivors of ONE pass over a tile
constexpr int S256_MAXGROUPS = 64;       // retry ladder: 256 rows x 4 columns = 1024 survivors per pass at most
constexpr int S256_NB = 64;              // histogram buckets per query
constexpr int S256_SH = 17;              // bucket width: 2^17 ulps of the fp32 score (1.6 % of the value; 64 buckets = one binade)
// Scope of the histogram traffic.  Workgroup scope makes the counters live in the L2 of the issuing XCD (atomics
// always execute in L2; the load only skips the CU's L1): ~0.6 us instead of a trip over the fabric to memory, which
// is what agent scope costs on a part with one L2 per XCD, and what every wave then waits for at the first
// vmcnt wait of its next main loop.  The price: the eight XCDs hold eight partial copies of a histogram.  That is
// safe -- a copy is "what memory held when the line was fetched" plus this XCD's own increments, so it never
// exceeds the true count and the derived bound only lags -- and nearly free: the launcher maps all slices of a
// query tile to one XCD in every phase of 8 a query tiles (fewer than 8 query tiles, or left over: each XCD tightens on
// its own share of the gallery, and with few queries the scan is HBM-bound anyway).
#define S256_HIST_SCOPE __HIP_MEMORY_SCOPE_WORKGROUP
// main-loop image + queue + staging + tau / base / start / end / cnt (256 x 4 B each) + ctrl + merge scratch + wkey
constexpr int S256_LDS = G256_LDS + S256_QCAP * 8 + S256_STG * 8 + 256 * 4 * 5 + 64 + 8 * 128 * 8 + 256 * 8;
static_assert(S256_LDS <= 163840, "the scan needs more LDS than a CU has");

// LDS byte offsets of the selection state behind the main-loop image (the dynamic LDS block starts at 0: no
// static LDS in these kernels; gemm256_core.h addresses its fragments the same way)
constexpr uint32_t S256_STG_OFF = G256_LDS + S256_QCAP * 8;
constexpr uint32_t S256_TAU_OFF = S256_STG_OFF + S256_STG * 8;
constexpr uint32_t S256_CNT_OFF = S256_TAU_OFF + 256 * 4 * 4;
constexpr uint32_t S256_CTRL_OFF = S256_CNT_OFF + 256 * 4;      // [0] overflow-queue count, [1] running total of staged survivors
constexpr uint32_t S256_BASE_OFF = S256_TAU_OFF + 256 * 4;
constexpr uint32_t S256_WKEY_OFF = S256_CTRL_OFF + 64 + 8 * 128 * 8;
// atomicAdd(&cnt, 1) on LDS as raw ISA.  Through the compiler every LDS atomic here is preceded by
// s_waitcnt vmcnt(0): the next tile's operand DMA (buffer_load ... lds) is in flight and the waitcnt pass cannot
// tell that it writes a different part of LDS -- so every survivor waited for that DMA AND for the global store and
// histogram atomic of the survivor before it (a full L2 round trip each: this was most of the selection's cost).
__device__ __forceinline__ int s256_lds_inc(uint32_t lds_byte_addr) {
    int old;
    asm volatile("ds_add_rtn_u32 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=v"(old) : "v"(lds_byte_addr), "v"(1) : "memory");
    return old;
}
// the same for an 8-byte LDS store (the compiler's own LDS stores wait for vmcnt(0) here for the same reason)
__device__ __forceinline__ void s256_lds_store64(uint32_t lds_byte_addr, uint64_t v) {
    asm volatile("ds_write_b64 %0, %1" :: "v"(lds_byte_addr), "v"(v) : "memory");
}
// Plain LDS reads at constant offsets (the dynamic LDS block starts at 0).  Going through the generic pointers of
// S256Lds made the compiler keep them alive across the main loop, spill one, and reload it from scratch behind an
// s_waitcnt vmcnt(0) in the middle of the selection.
__device__ __forceinline__ uint32_t s256_lds_u32(uint32_t off) { return *(__attribute__((address_space(3))) const uint32_t*)(uintptr_t)off; }
__device__ __forceinline__ uint64_t s256_lds_u64(uint32_t off) { return *(__attribute__((address_space(3))) const uint64_t*)(uintptr_t)off; }
__device__ __forceinline__ float s256_lds_f32(uint32_t off) { return *(__attribute__((address_space(3))) const float*)(uintptr_t)off; }
// workgroup barrier that orders LDS traffic only (__syncthreads() also drains vmcnt: the next tile's operand DMA and
// the selection's fire-and-forget global stores would have to land first)
__device__ __forceinline__ void s256_barrier_lds() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

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
__device__ __forceinline__ uint64_t s256_readlane(uint64_t v, int l) {
    return (uint64_t)(uint32_t)__builtin_amdgcn_readlane((uint32_t)v, l) |
           ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((uint32_t)(v >> 32), l) << 32);
}
// the 64 lane values sorted, largest in lane 0
__device__ __forceinline__ uint64_t s256_sort_desc(uint64_t v, int lane) {
#pragma unroll
    for (int k2 = 2; k2 <= 64; k2 <<= 1) {
#pragma unroll
        for (int j = k2 >> 1; j > 0; j >>= 1) {
            const uint64_t o = s256_shfl_xor(v, j);
            const bool take_max = (((lane & j) == 0) == ((lane & k2) == 0));
            v = take_max ? (v > o ? v : o) : (v < o ? v : o);
        }
    }
    return v;
}
// v bitonic across the wave -> sorted, largest in lane 0
__device__ __forceinline__ uint64_t s256_bitonic_merge_desc(uint64_t v, int lane) {
#pragma unroll
    for (int j = 32; j > 0; j >>= 1) {
        const uint64_t o = s256_shfl_xor(v, j);
        v = ((lane & j) == 0) ? (v > o ? v : o) : (v < o ? v : o);
    }
    return v;
}

// a 64-bit key that another wave of this workgroup (or an earlier drain) stored: two 4-byte agent-scope loads (the
// stores are complete by the time this runs, so the halves cannot be torn)
__device__ __forceinline__ uint64_t s256_load_key_coherent(const uint64_t* p) {
    const uint32_t* q = (const uint32_t*)p;
    const uint32_t lo = __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const uint32_t hi = __hip_atomic_load(q + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return ((uint64_t)hi << 32) | lo;
}

// One step of "list := best KSEL distinct keys of (list u new keys)".
//   KSEL = 32: cur = the list in lanes 0..31 (best first, 0 = empty), fresh = up to 32 new keys in lanes 32..63,
//              WORST FIRST (lane 63 holds the best new key).
//   KSEL = 64: cur = the list in all 64 lanes, fresh = up to 64 new keys in all lanes, worst first.
// Keys may repeat (a tile that is recomputed after a queue overflow finds its survivors again): equal keys
// are dropped after the merge and the survivors are compacted through the wave's LDS scratch `ws` (128 slots;
// LDS operations of one wave complete in order).
template <int KSEL>
__device__ __forceinline__ uint64_t s256_merge_step(uint64_t cur, uint64_t fresh, uint64_t* ws_, int lane) {
    // The compaction passes values BETWEEN lanes through LDS: lane a stores to slot p, lane p loads slot p.  To the
    // compiler that is one thread storing to ws[lane] and ws[p] and loading ws[lane] -- it may (and, depending on
    // what this is inlined into, did) answer the load from the thread's own stores, i.e. "0 unless p == lane".
    // volatile keeps the three LDS operations as written; the hardware runs one wave's LDS operations in order.
    volatile uint64_t* ws = ws_;
    const unsigned long long below = (1ull << lane) - 1ull;
    if (KSEL == 32) {
        uint64_t v = lane < 32 ? cur : fresh;       // best-first then worst-first: bitonic
        v = s256_bitonic_merge_desc(v, lane);
        const uint64_t prev = s256_shfl_up1(v);
        const bool keep = v != 0ull && (lane == 0 || v != prev);
        const unsigned long long km = __ballot(keep);
        ws[lane] = 0ull;
        if (keep) ws[__popcll(km & below)] = v;
        return lane < KSEL ? ws[lane] : 0ull;
    } else {
        // a 128-element bitonic merge held in two registers: hi = element-wise maximum (the best 64, bitonic),
        // lo = minimum (the rest, bitonic); both are sorted, duplicates dropped across the whole 128, and the
        // first 64 survivors are the new list
        uint64_t hi = cur > fresh ? cur : fresh, lo = cur > fresh ? fresh : cur;
        hi = s256_bitonic_merge_desc(hi, lane);
        lo = s256_bitonic_merge_desc(lo, lane);
        const uint64_t hprev = s256_shfl_up1(hi), lprev = s256_shfl_up1(lo);
        const uint64_t hlast = s256_readlane(hi, 63);
        const bool keep_h = hi != 0ull && (lane == 0 || hi != hprev);
        const bool keep_l = lo != 0ull && lo != (lane == 0 ? hlast : lprev);
        const unsigned long long mh = __ballot(keep_h), ml = __ballot(keep_l);
        const int nh = __popcll(mh);
        ws[lane] = 0ull;
        ws[64 + lane] = 0ull;
        if (keep_h) ws[__popcll(mh & below)] = hi;
        if (keep_l) ws[nh + __popcll(ml & below)] = lo;
        return ws[lane];
    }
}

struct S256Lds {
    uint64_t* queue;     // [S256_QCAP] overflow queue
    uint64_t* stage;     // [S256_STG] survivors of the current pass (row | score | index), flushed after the barrier
    float* tau;          // [256] admission score per query row of the tile
    uint32_t* base;      // [256] order-preserving u32 of the pre-pass bound: histogram origin of the row
    int* start;          // [256]
    int* end;            // [256]
    int* cnt;            // [256] entries appended to the row's segment so far (may run past the capacity: those went to the queue)
    int* ctrl;           // [0] queue count (may exceed the capacity: overflow), [1] running total of staged survivors (never reset)
    uint64_t* scratch;   // [8][128] wave-private
    uint64_t* wkey;      // [256] key of the row's KSEL-th best entry as of its last drain (0 = never drained / fewer than KSEL)
};

// Admission bounds from the global histograms.  hist[q][b] counts the scores of query q that any slice
// has appended so far (plus the pre-pass list) whose order-preserving u32 lies in
// [base + (b << SH), base + ((b + 1) << SH))  (the last bucket is open-ended).  If the buckets b.. hold at
// least KSEL scores, KSEL distinct gallery rows score at least the lower edge of bucket b: a valid lower
// bound of the query's final KSEL-th best.  Counts only ever lag (a store may not have landed yet): stale
// reads give weaker bounds, never wrong ones.  One wave per row, rows strided by 8.
// the admission score for a (valid) bound t: with a margin, max(pre-pass bound, t - margin) (CertArgs, kernels.h)
__device__ __forceinline__ float s256_admit(float t, uint32_t base, float mg) {
    return mg > 0.f ? fmaxf(orderable_f32(base), t - mg) : t;
}
template <int KSEL, bool MARGIN>
__device__ __noinline__ void s256_refresh_hist(const S256Lds& L, const uint32_t* hist, const uint32_t* tau_g, int q0,
                                               int qvalid, int tid, const float* marg) {
    const int wave = tid >> 6, lane = tid & 63;
    // the drain-published bounds of this wave's 32 rows (rows wave + 8 i), one per lane
    uint32_t tgv = 0u;
    float mgv = 0.f;
    if (lane < 32 && wave + 8 * lane < qvalid) {
        tgv = __hip_atomic_load(tau_g + q0 + wave + 8 * lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if constexpr (MARGIN) mgv = marg[q0 + wave + 8 * lane];
    }
    // all 32 rows' counters are requested before any is used: one L2 round trip per refresh, not one per batch
    uint32_t h[32];
#pragma unroll
    for (int u = 0; u < 32; ++u) {
        const int r = wave + 8 * u;
        h[u] = r < qvalid ? __hip_atomic_load(hist + (long)(q0 + r) * S256_NB + lane, __ATOMIC_RELAXED, S256_HIST_SCOPE) : 0u;
    }
#pragma unroll
    for (int u = 0; u < 32; ++u) {
        const int r = wave + 8 * u;
        if (r >= qvalid) break;                                   // wave-uniform
        // suffix sums: s[lane] = h[lane] + h[lane + 1] + ... + h[63]
        uint32_t sfx = h[u];
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t t = __shfl_down(sfx, o, 64);
            sfx += (lane + o < 64) ? t : 0u;
        }
        const unsigned long long m = __ballot(sfx >= (uint32_t)KSEL);
        if (lane == 0) {
            float t = orderable_f32((uint32_t)__builtin_amdgcn_readlane((int)tgv, u));
            if (m) {
                const int b = 63 - __builtin_clzll(m);
                const float th = orderable_f32(L.base[r] + ((uint32_t)b << S256_SH));
                t = th > t ? th : t;
            }
            if constexpr (MARGIN)
                t = s256_admit(t, L.base[r], __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, mgv), u)));
            if (t > L.tau[r]) L.tau[r] = t;
        }
    }
    __syncthreads();
}

// All 512 threads; the slow, exact path.  Sort the overflow queue (row, score, index descending) and, for
// every row that has queued entries, merge them with the row's segment into the row's best KSEL distinct
// keys: those go back to the head of the segment (sorted, best first), the rest of it is free again.
template <int KSEL, bool MARGIN>
__device__ __noinline__ void s256_drain(const S256Lds& L, uint64_t* seg0, long seg_row_stride, int q0, int qvalid,
                                        uint32_t idx_base, int tid, uint32_t* tau_g, const float* marg, int* dropflag) {
    constexpr int SEG = 2 * KSEL;
    const int wave = tid >> 6, lane = tid & 63;
    // The segments were written by other waves of this workgroup since this CU last read them (appends by the
    // flush, list heads by earlier drains).  __syncthreads() makes every wave wait for its own stores; the CU's
    // vector L1 is NOT kept coherent with them (a line it holds from the previous read-back stays as it was), so
    // it is invalidated here (acquire fence: buffer_inv) and the read-back uses 4-byte sc1 loads.
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
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
    for (int r = wave; r < qvalid; r += 8) {
        const int s0 = L.start[r];
        const int cnt = L.end[r] - s0;
        if (cnt <= 0) continue;
        uint64_t* seg = seg0 + (long)r * seg_row_stride;
        int c = L.cnt[r];
        c = c < SEG ? c : SEG;
        // the segment as it stands (appended by any wave of this workgroup, unsorted): sort it, drop repeats
        uint64_t cur;
        if (KSEL == 32) {
            uint64_t v = lane < c ? s256_load_key_coherent(seg + lane) : 0ull;
            v = s256_sort_desc(v, lane);
            // as a merge step with an empty list: keeps the best 32 distinct keys
            const uint64_t lo32 = s256_shfl_xor(v, 63);          // lanes 32..63 <- v[31..0]: the best 32, worst first
            cur = s256_merge_step<32>(0ull, lane >= 32 ? lo32 : 0ull, ws, lane);
            // the other 32 (v[32..63]) can only matter if the best 32 held repeats
            const uint64_t rest = s256_shfl_xor(v, 31);          // lanes 32..63 <- v[63..32]
            if (s256_readlane(cur, 31) == 0ull) cur = s256_merge_step<32>(cur, lane >= 32 ? rest : 0ull, ws, lane);
        } else {
            uint64_t v0 = lane < c ? s256_load_key_coherent(seg + lane) : 0ull;
            uint64_t v1 = lane + 64 < c ? s256_load_key_coherent(seg + 64 + lane) : 0ull;
            v0 = s256_sort_desc(v0, lane);
            v1 = s256_sort_desc(v1, lane);
            cur = s256_merge_step<64>(0ull, s256_shfl_xor(v0, 63), ws, lane);
            cur = s256_merge_step<64>(cur, s256_shfl_xor(v1, 63), ws, lane);
        }
        // the row's queued entries, KSEL at a time (best chunks first)
        for (int off = 0; off < cnt; off += KSEL) {
            const int cc = (cnt - off) < KSEL ? (cnt - off) : KSEL;
            const uint64_t best = s256_entry_to_key(L.queue[s0 + off], idx_base);
            const uint64_t worst_kept = s256_readlane(cur, KSEL - 1);
            if (worst_kept != 0ull && best < worst_kept) break;   // nothing further down can enter
            const int jx = 63 - lane;                             // lane 63 takes the chunk's best entry
            const uint64_t fresh = jx < cc ? s256_entry_to_key(L.queue[s0 + off + jx], idx_base) : 0ull;
            cur = s256_merge_step<KSEL>(cur, fresh, ws, lane);
        }
        if (lane < KSEL) seg[lane] = cur;
        const int kept = __popcll(__ballot(cur != 0ull));
        const uint64_t last = s256_readlane(cur, KSEL - 1);
        if (lane == 0) {
            L.cnt[r] = kept;
            L.wkey[r] = last;                                     // 0 while fewer than KSEL
            if (last != 0ull) {
                // this slice's KSEL-th best is a lower bound of the query's final KSEL-th best: publish it
                const uint32_t lo = (uint32_t)(last >> 32);
                (void)__hip_atomic_fetch_max(tau_g + q0 + r, lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                float t = orderable_f32(lo);
                if constexpr (MARGIN) t = s256_admit(t, L.base[r], marg[q0 + r]);
                if (t > L.tau[r]) L.tau[r] = t;
            }
            // a drained segment keeps its best KSEL keys only, and from here on the row admits against them: rows inside
            // the admission margin are no longer all kept
            if constexpr (MARGIN) dropflag[q0 + r] = 1;
        }
    }
    __syncthreads();
    if (tid == 0) L.ctrl[0] = 0;
    __syncthreads();
}

constexpr int S256_PHASES = 8;
struct Scan256Args {
    const bf16_t* Qb; long ldq;
    const bf16_t* Gb; long ldg;
    int Q; long N; int D;
    long n_begin;                 // rows [0, n_begin) are covered by the pre-pass
    int splits;                   // segment slots per query: the most slices any phase has
    // The launch is a sequence of up to S256_PHASES phases; phase i holds the blocks [first[i], first[i + 1]) and deals
    // them as (query tile q0[i] + j % qn[i], slice j / qn[i]) of ns[i] slices (launch_topk_scan256 says which phases)
    int nph;
    int ph_first[S256_PHASES + 1], ph_q0[S256_PHASES], ph_qn[S256_PHASES], ph_ns[S256_PHASES];
    uint64_t* seg;                // [Q][splits][2 KSEL] appended keys (score desc / index asc order, unsorted)
    int* seg_cnt;                 // [Q][splits] valid entries of each segment (written when the slice is done)
    uint32_t* tau_g;              // [Q] shared admission scores (order-preserving u32), seeded by the pre-pass, raised by drains
    const uint32_t* tau_base;     // [Q] the pre-pass bound, constant during the scan: origin of the histogram buckets
    uint32_t* hist;               // [Q][S256_NB] score histogram shared by all slices (seeded with the pre-pass list)
    const float* marg;            // [Q] admission margins (null: none): every score >= max(pre-pass bound, bound - marg) is kept
    int* dropflag;                // [Q] set when a drain or a recomputed tile touched the query's segments (with marg)
    int dbg;                      // timing experiments only (REVO_EXPERIMENTS): 1 = skip the selection, 4 = skip the slow path, 8 = no global stores / atomics from the selection, 16 = no refreshes (wrong results)
    unsigned long long* stats;    // optional counters (REVO_EXPERIMENTS): [0] drains, [1] queued entries, [2] retry passes, [3] fragments scanned slowly, [4] appended entries, [5] refreshes
};

// ROWS: 0 = all 256 query rows of a tile may be valid; 64 / 128 / 192 = the whole search has at most that many
// queries (one query tile), and the main loop skips the MFMA work of the rows that cannot be valid.
// MARGIN: the scan admits against max(pre-pass bound, bound - marg[q]) and flags the queries whose segments a drain or a
// recomputed tile touched (CertArgs, kernels.h).  A template parameter, not a run-time switch: the 256-row form runs at
// the 256-VGPR limit, and with the margin's pointers and branches compiled into it the plain scan spilled 16 VGPRs and
// lost 5 % (10 000 queries) to 17 % (256 queries) -- measured in round 4 before this was split off.
template <int KSEL, int ROWS, bool MARGIN = false>
__global__ __launch_bounds__(G256_THREADS, 2) void topk_scan256_kernel(Scan256Args p) {
    constexpr int SEG = 2 * KSEL;
    // up to 128 queries the loop runs at the HBM rate: operands are requested a K-tile and a half ahead (gemm256_core.h)
    constexpr bool DEEP = ROWS == 64 || ROWS == 128;
    // ROWS != 0: the whole search is ONE query tile, so every gallery row is read by one workgroup, once: non-temporal DMA
    // (measured, 1 M x 1024, whole search: 1 query 0.416-0.427 -> 0.400-0.401 ms, 64 queries 0.441-0.451 -> 0.427-0.431,
    //  128 queries 0.521 -> 0.493-0.499; the bytes no longer push the queries, bounds and segments out of L2 / the Infinity Cache).
    //  With several query tiles the workgroups of an XCD that hold the same slice re-read it from L2: default policy.
#ifndef S256_ONE_TILE_AUX
#define S256_ONE_TILE_AUX 2
#endif
    constexpr int BAUX = ROWS != 0 ? S256_ONE_TILE_AUX : 0;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    S256Lds L;
    L.queue = (uint64_t*)(smem + G256_LDS);
    L.stage = (uint64_t*)(smem + S256_STG_OFF);
    L.tau = (float*)(smem + S256_TAU_OFF);
    L.base = (uint32_t*)(L.tau + 256);
    L.start = (int*)(L.base + 256);
    L.end = L.start + 256;
    L.cnt = L.end + 256;
    L.ctrl = L.cnt + 256;
    L.scratch = (uint64_t*)(L.ctrl + 16);
    L.wkey = L.scratch + 8 * 128;
    static_assert(S256_CNT_OFF == S256_TAU_OFF + (256 + 256 + 256 + 256) * 4, "cnt follows tau, base, start, end");
    static_assert(S256_WKEY_OFF + 256 * 8 == S256_LDS, "wkey is the last block");

#ifndef REVO_EXPERIMENTS
    // the product build has no timing switches and no counters: these fold to constants
    p.dbg = 0;
    p.stats = nullptr;
#endif
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int lane = tid & 63;
    // 1-D grid in phases: block b of phase i (scalar compares against the phases' first blocks) -> local j = b - first[i],
    // query tile q0[i] + j % qn[i], slice j / qn[i] of ns[i].  Blocks are dealt to the 8 XCDs round-robin and every phase
    // starts at a multiple of 8, so in a phase of 8 a query tiles all slices of a query tile run on one XCD (shared L2:
    // histograms, query rows) and the a workgroups that hold the same slice of the XCD's a query tiles stream the same
    // gallery rows side by side.
    // (constant indices only: a run-time index into the by-value argument block makes hipcc copy it to scratch)
    int ph_first = p.ph_first[0], ph_q0 = p.ph_q0[0], ph_qn = p.ph_qn[0], nsl = p.ph_ns[0];
#pragma unroll
    for (int i = 1; i < S256_PHASES; ++i)
        if (i < p.nph && (int)blockIdx.x >= p.ph_first[i]) { ph_first = p.ph_first[i]; ph_q0 = p.ph_q0[i]; ph_qn = p.ph_qn[i]; nsl = p.ph_ns[i]; }
    const int jloc = (int)blockIdx.x - ph_first;
    // (an integer division is a vector-unit sequence: its wave-uniform results are moved to scalar registers here, or the
    //  slice bounds derived from them sit in vector registers across the main loop -- at the 256-register limit, in scratch)
    const int sp = __builtin_amdgcn_readfirstlane(jloc / ph_qn);
    const int qtile = ph_q0 + (jloc - sp * ph_qn);
    const int q0 = qtile * 256;
    if (q0 >= p.Q) return;
    const int qvalid = (p.Q - q0) < 256 ? (p.Q - q0) : 256;

    // slices: the tiles are dealt out as evenly as possible (the first `rem` slices take one more)
    const long span = p.N - p.n_begin;
    const long tiles = (span + 255) / 256;
    // (32-bit and scalar: tiles < 2^24; the hardware has no scalar 64-bit order compare, so as `long` these bounds lived in
    //  vector registers across the main loop)
    const int per = __builtin_amdgcn_readfirstlane((int)tiles / nsl), rem = (int)tiles - per * nsl;
    const int t0 = sp * per + (sp < rem ? sp : rem);
    const int t1 = t0 + per + (sp < rem ? 1 : 0);
    const long row_begin = p.n_begin + (long)t0 * 256;            // first gallery row of this slice
    const uint32_t idx_base = (uint32_t)row_begin;
    const long seg_row_stride = (long)p.splits * SEG;
    uint64_t* myseg = p.seg + ((long)q0 * p.splits + sp) * SEG;      // row r of the tile: + r * seg_row_stride
    uint32_t* myhist = p.hist + (long)q0 * S256_NB;

    if (tid < 256) {
        const uint32_t b0 = tid < qvalid ? p.tau_base[q0 + tid] : 0u;
        float t0_ = tid < qvalid ? orderable_f32(p.tau_g[q0 + tid]) : INFINITY;
        if constexpr (MARGIN) { if (tid < qvalid) t0_ = s256_admit(t0_, b0, p.marg[q0 + tid]); }
        L.tau[tid] = t0_;
        L.base[tid] = b0;
        L.wkey[tid] = 0ull;
        L.cnt[tid] = 0;
    }
    if (tid == 0) { L.ctrl[0] = 0; L.ctrl[1] = 0; }
    __syncthreads();
    // (phases differ in their slice counts: a query's slots beyond its phase's slices are closed by slice 0)
    if (sp == 0 && tid < qvalid)
        for (int s2 = nsl; s2 < p.splits; ++s2) p.seg_cnt[(long)(q0 + tid) * p.splits + s2] = 0;
    if (t0 >= t1) {
        if (tid < qvalid) p.seg_cnt[(long)(q0 + tid) * p.splits + sp] = 0;
        return;
    }

    G256Operand A, B;
    g256_operand_init(A, p.Qb, p.ldq, p.Q, q0, wave, lane);
    g256_operand_init(B, p.Gb + row_begin * p.ldg, p.ldg, p.N - row_begin, 0, wave, lane);
    if constexpr (DEEP) g256_issue_prologue_deep<BAUX>(A, B, smem, p.D, wave); else g256_issue_prologue<BAUX>(A, B, smem, p.D, wave);
    // Slices that start after others have run (later rounds of workgroups on this CU) begin with what those have
    // learnt, not with the pre-pass bound: one refresh while the first operands are in flight.  (Without it every
    // slice's first tile admitted about one score per row: at 24 tiles per slice a third of all slow fragments.)
    if (!(p.dbg & 17)) {
        if (p.stats && tid == 0) atomicAdd(p.stats + 5, 1ull);
        s256_refresh_hist<KSEL, MARGIN>(L, p.hist, p.tau_g, q0, qvalid, tid, MARGIN ? p.marg : nullptr);
    }

    // Normal mode: one pass per tile (groups == 1).  If a pass pushes more entries to the overflow queue than
    // it holds, or stages more survivors than the staging buffer holds (an adversarially ordered gallery), the tile
    // is recomputed in 2, 4, ... 64 column groups, one pass and one drain per group; at 64 groups a pass can
    // admit at most 256 x 4 = 1024 entries, so the retry always terminates.  Entries found twice are dropped when lists are merged (drain, final reduce).
    int t = t0;
    int groups = 1, grp = 0;
    uint32_t staged_before = 0;            // survivors staged by earlier passes (the LDS total is never reset)
    while (t < t1) {
        const long n0 = p.n_begin + (long)t * 256;
        {
            f32x4 acc[8][4];
#pragma unroll
            for (int m = 0; m < 8; ++m)
#pragma unroll
                for (int n = 0; n < 4; ++n) acc[m][n] = (f32x4){0.f, 0.f, 0.f, 0.f};
            // few queries: most of the tile's MFMA work would multiply zero rows (separate kernel
            // instantiations: inside one kernel a second main loop costs the main path its register allocation)
            gemm256_mainloop<ROWS, false, DEEP, 0, BAUX>(A, B, smem, p.D, wave, lane, acc);

            if (groups == 1 && t + 1 < t1) {
                // next gallery tile: rebased descriptors (any gallery size), DMA in flight during the selection
                g256_operand_init(B, p.Gb + (n0 + 256) * p.ldg, p.ldg, p.N - (n0 + 256), 0, wave, lane);
                if constexpr (DEEP) g256_issue_prologue_deep<BAUX>(A, B, smem, p.D, wave); else g256_issue_prologue<BAUX>(A, B, smem, p.D, wave);
            }
            asm volatile("" : "+v"(lane) :: "memory");
            const int lr = lane & 15, lq = lane >> 4;
            const int rbase = (wave >> 2) * 128 + lr;            // + m * 16
            const int cbase = (wave & 3) * 64 + lq * 4;          // + n * 16 + j
            const long left = p.N - n0;
            const uint32_t rel0 = (uint32_t)(n0 - row_begin);
            float taum[8];
            unsigned hitm = 0;             // bit m: some lane of this wave has a candidate in row fragment m
            if (left < 256) {
                // the gallery's last, ragged tile: columns past the end become NaN (never admitted, ignored by
                // fmaxf).  One uniform branch per tile; inside the loops below the same test cost two scalar
                // instructions and a branch per score.
#pragma unroll
                for (int n = 0; n < 4; ++n)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const bool past = cbase + n * 16 + j >= left;
#pragma unroll
                        for (int m = 0; m < 8; ++m) acc[m][n][j] = past ? __builtin_nanf("") : acc[m][n][j];
                    }
            }
            if (p.dbg & 1) {
                asm volatile("" :: "v"(acc[0][0]), "v"(acc[7][3]));
            } else {
#pragma unroll
                for (int m = 0; m < 8; ++m) taum[m] = s256_lds_f32(S256_TAU_OFF + (rbase + m * 16) * 4);
#pragma unroll
                for (int m = 0; m < 8; ++m) {
                    float mx = -INFINITY;
#pragma unroll
                    for (int n = 0; n < 4; ++n)
#pragma unroll
                        for (int j = 0; j < 4; ++j) mx = fmaxf(mx, acc[m][n][j]);
                    if (__ballot(mx >= taum[m]) != 0ull) hitm |= 1u << m;
                }
            }
            if (hitm && !(p.dbg & 4)) {
#pragma unroll
                for (int m = 0; m < 8; ++m) {
                    if (!(hitm & (1u << m))) continue;          // wave-uniform
                    if (p.stats && lane == 0) atomicAdd(p.stats + 3, 1ull);
                    const int row = rbase + m * 16;
                    // the row's KSEL-th best as of its last drain, as (score, index); none yet admits everything
                    const uint64_t wk = s256_lds_u64(S256_WKEY_OFF + row * 8);
                    const float ws = wk ? key_score(wk) : -INFINITY;
                    const uint32_t widx = wk ? key_index(wk) : 0xffffffffu;
#pragma unroll
                    for (int n = 0; n < 4; ++n)
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const float v = acc[m][n][j];
                            const int col = cbase + n * 16 + j;
                            // admission score (shared across slices) first, then the strict test against the
                            // row's own KSEL-th entry: an equal score enters only with a smaller index, so a
                            // huge tie group cannot keep the queue full forever.  Survivors are rare: each lane
                            // stages its own (row | score | index) in LDS under a mostly empty exec mask; nothing
                            // here touches global memory.
                            const bool pass = v >= taum[m];
                            if (__ballot(pass) == 0ull) continue;       // wave-uniform: most elements of a hit fragment fail too
                            if (pass && (col & (groups - 1)) == grp &&
                                (v > ws || (v == ws && idx_base + rel0 + col < widx))) {
                                const uint32_t pos = (uint32_t)s256_lds_inc(S256_CTRL_OFF + 4) - staged_before;
                                if (pos < (uint32_t)S256_STG) s256_lds_store64(S256_STG_OFF + pos * 8, s256_entry(row, v, rel0 + col));
                            }
                        }
                }
            }
        }
        // the accumulators are dead from here on (refresh and drain are real calls)
        s256_barrier_lds();
        // Flush.  The pass's survivors sit in the staging buffer; thread i moves entry i to its row's segment in
        // global memory (slot from the row's LDS counter) and counts it in the query's histogram.  A tile has a few
        // dozen survivors, so a wave issues at most ONE store and ONE atomic instruction per tile, and those fit
        // inside the four vector-memory operations the next main loop's first wait leaves outstanding anyway.
        // (Placed by the lane that found them, 2 x 7 operations per wave and tile were in flight at that wait, and it
        //  -- and then the whole workgroup at the barrier behind it -- sat there until they had been acknowledged.)
        const uint32_t staged_total = s256_lds_u32(S256_CTRL_OFF + 4);
        const uint32_t staged = staged_total - staged_before;
        staged_before = staged_total;
        {
            const uint32_t nst = staged < (uint32_t)S256_STG ? staged : (uint32_t)S256_STG;
            for (uint32_t i = tid; i < nst; i += 512) {
                const uint64_t e = s256_lds_u64(S256_STG_OFF + i * 8);
                const int row = (int)(e >> 56);
                const int slot = s256_lds_inc(S256_CNT_OFF + row * 4);
                if (slot < SEG) {
                    if (!(p.dbg & 8)) {
                        myseg[(long)row * seg_row_stride + slot] = s256_entry_to_key(e, idx_base);
                        const uint32_t so = (uint32_t)(e >> 24), bo = s256_lds_u32(S256_BASE_OFF + row * 4);
                        if (groups == 1 && (!MARGIN || so >= bo)) {   // a recomputed tile must not be counted twice; a score
                                                               // admitted by the margin only may lie below the histogram's origin
                            uint32_t b = (so - bo) >> S256_SH;
                            b = b < (uint32_t)(S256_NB - 1) ? b : (uint32_t)(S256_NB - 1);
                            (void)__hip_atomic_fetch_add(myhist + (long)row * S256_NB + b, 1u, __ATOMIC_RELAXED, S256_HIST_SCOPE);
                        }
                    }
                    if (p.stats) atomicAdd(p.stats + 4, 1ull);
                } else {
                    const int pos = s256_lds_inc(S256_CTRL_OFF);
                    if (pos < S256_QCAP) s256_lds_store64(G256_LDS + pos * 8, e);
                }
            }
        }
        s256_barrier_lds();
        const int qc = (int)s256_lds_u32(S256_CTRL_OFF);
        const bool overflow = qc > S256_QCAP || staged > (uint32_t)S256_STG;
        if (p.stats && tid == 0) {
            if (overflow || groups > 1) atomicAdd(p.stats + 2, 1ull);
            if (overflow || groups > 1 || qc >= S256_DRAIN || (t + 1 >= t1 && qc > 0)) { atomicAdd(p.stats + 0, 1ull); atomicAdd(p.stats + 1, (unsigned long long)qc); }
        }
        if (groups == 1 && !overflow) {
            if (qc >= S256_DRAIN || (t + 1 >= t1 && qc > 0))
                s256_drain<KSEL, MARGIN>(L, myseg, seg_row_stride, q0, qvalid, idx_base, tid, p.tau_g, MARGIN ? p.marg : nullptr, MARGIN ? p.dropflag : nullptr);
            // what all slices of these queries have learnt meanwhile: after the first two tiles (the bound moves
            // fastest early on: it follows KSEL / rows seen), then every fourth tile, every 16th from tile 32 on
            // (each refresh is an L2 round trip plus ~3 us of wave scans)
            const int tl = t - t0;
            if (t + 1 < t1 && !(p.dbg & 17) && (tl < 2 || ((tl & 3) == 3 && tl < 32) || (tl & 15) == 15)) {
                if (p.stats && tid == 0) atomicAdd(p.stats + 5, 1ull);
                s256_refresh_hist<KSEL, MARGIN>(L, p.hist, p.tau_g, q0, qvalid, tid, MARGIN ? p.marg : nullptr);
            }
            ++t;
            continue;
        }
        // retry mode (or entering it): merge what was queued, then recompute this tile / its next column group
        s256_drain<KSEL, MARGIN>(L, myseg, seg_row_stride, q0, qvalid, idx_base, tid, p.tau_g, MARGIN ? p.marg : nullptr, MARGIN ? p.dropflag : nullptr);
        // a recomputed tile appends its survivors a second time: these queries' segments may hold repeated keys
        if constexpr (MARGIN) { if (tid < qvalid) p.dropflag[q0 + tid] = 1; }
        if (overflow) {
            groups = groups < S256_MAXGROUPS ? groups * 2 : S256_MAXGROUPS;
            grp = 0;
        } else if (++grp == groups) {
            groups = 1;
            grp = 0;
            ++t;                                                  // tile complete
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // a DMA issued for another tile must not land on top
        __syncthreads();
        if (t < t1) {
            const long nn = p.n_begin + (long)t * 256;
            g256_operand_init(B, p.Gb + nn * p.ldg, p.ldg, p.N - nn, 0, wave, lane);
            if constexpr (DEEP) g256_issue_prologue_deep<BAUX>(A, B, smem, p.D, wave); else g256_issue_prologue<BAUX>(A, B, smem, p.D, wave);
        }
    }
    if (tid < qvalid) {
        const int c = L.cnt[tid];
        p.seg_cnt[(long)(q0 + tid) * p.splits + sp] = c < SEG ? c : SEG;
    }
}

// ---------------------------------------------------------------- the final selection ----
// One workgroup of RW waves per query: the query's best KSEL distinct keys out of the pre-pass list and the
// segments of all slices.  Wave w packs the segments of slices w, w + RW, ... into 64-key chunks (whole
// segments, so that a repeated key always sits in one chunk or in consecutive ones), sorts each chunk
// and merges it into its running list; the RW partial lists meet in LDS and wave 0 merges them.
// With few slices (up to 32; a shard of a row-sharded gallery: 6) one wave per query does all of it, four queries per workgroup:
// their segments pack into one or two chunks, against one sort per slice and RW - 1 merges in the 8-wave form.
constexpr int S256_RW = 8;
template <int KSEL>
__device__ __forceinline__ uint64_t s256_fold_chunk(uint64_t run, uint64_t chunk, uint64_t* ws, int lane) {
    chunk = s256_sort_desc(chunk, lane);
    if (KSEL == 32) {
        // best 32 of the chunk, then (only if they can matter) the other 32
        run = s256_merge_step<32>(run, s256_shfl_xor(chunk, 63), ws, lane);          // lanes 32..63 <- chunk[31..0]
        const uint64_t worst = s256_readlane(run, 31), next = s256_readlane(chunk, 32);
        if (next != 0ull && (worst == 0ull || next > worst))
            run = s256_merge_step<32>(run, s256_shfl_xor(chunk, 31), ws, lane);      // lanes 32..63 <- chunk[63..32]
        return run;
    } else {
        return s256_merge_step<64>(run, s256_shfl_xor(chunk, 63), ws, lane);
    }
}
template <int KSEL, int RW>
__global__ __launch_bounds__((RW == 1 ? 4 : RW) * 64) void topk_reduce_segs_kernel(const uint64_t* __restrict__ seg,
                                                                                  const int* __restrict__ seg_cnt, int splits,
                                                                                  const uint64_t* __restrict__ prelist,
                                                                                  uint64_t* __restrict__ out, int Q,
                                                                                  uint32_t* __restrict__ bounds, int top_m) {
    constexpr int SEG = 2 * KSEL;
    constexpr int NWV = RW == 1 ? 4 : RW;               // waves per workgroup
    __shared__ uint64_t partial[RW == 1 ? 1 : RW][64];
    __shared__ uint64_t wsb[NWV][128];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int w = RW == 1 ? 0 : wv;                     // this wave's share of the query's slices: w, w + RW, ...
    const long q = RW == 1 ? (long)blockIdx.x * 4 + wv : (long)blockIdx.x;
    if (RW == 1 && q >= Q) return;
    uint64_t* ws = wsb[wv];
    uint64_t run = (w == 0 && lane < KSEL) ? prelist[q * KSEL + lane] : 0ull;
    int fill = 0;                                      // wave-uniform
    const int* cq = seg_cnt + q * splits;
    const uint64_t* sq = seg + q * (long)splits * SEG;
    // A chunk is first laid out from the counts alone -- lane l takes entry src_off of its slices, whole segments as before --
    // and then fetched by ONE gather.  (A load per non-empty slice, each waited for before the next could be merged into the
    // chunk register, was a chain of 20-30 memory round trips per wave: 18 of a one-query search's 440 us.)
    long src_off = 0;                                  // this lane's entry of the chunk being laid out: offset into sq
    auto fold = [&]() {
        const uint64_t chunk = lane < fill ? sq[src_off] : 0ull;
        run = s256_fold_chunk<KSEL>(run, chunk, ws, lane);
        fill = 0;
    };
    // this wave's counts, one slice per lane (slices w + RW * lane), 64 slices at a time
    for (int sb = w; sb < splits; sb += RW * 64) {
        const int my = sb + RW * lane;
        const int cl = my < splits ? cq[my] : 0;
        const int nlan = (splits - sb + RW - 1) / RW;
        for (int i = 0; i < (nlan < 64 ? nlan : 64); ++i) {
            int c = __builtin_amdgcn_readlane(cl, i);
            if (c <= 0) continue;
            const long sp = (long)(sb + RW * i) * SEG;
            int done = 0;
            while (done < c) {
                int take = c - done;
                if (fill + take > 64) {
                    if (fill > 0) fold();
                    take = take < 64 ? take : 64;
                }
                if (lane >= fill && lane < fill + take) src_off = sp + done + lane - fill;
                fill += take;
                done += take;
            }
        }
    }
    if (fill > 0) fold();
    if constexpr (RW == 1) {
        if (lane < KSEL) out[q * KSEL + lane] = run;
        // row-sharded search: the scan scores of the best top_m candidates, published for the bound exchange (was a launch of its own)
        if (bounds && lane < top_m) bounds[q * top_m + lane] = (uint32_t)(run >> 32);
        return;
    }
    partial[w][lane] = run;
    __syncthreads();
    if (w == 0) {
#pragma unroll 1
        for (int o = 1; o < RW; ++o) {
            const uint64_t other = partial[o][63 - lane];              // worst first; KSEL = 32: lanes 32..63 <- entries 31..0
            run = s256_merge_step<KSEL>(run, other, ws, lane);
        }
        if (lane < KSEL) out[q * KSEL + lane] = run;
        if (bounds && lane < top_m) bounds[q * top_m + lane] = (uint32_t)(run >> 32);
    }
}
int launch_topk_reduce_segs(const uint64_t* seg, const int* seg_cnt, int splits, const uint64_t* prelist, uint64_t* out,
                            int Q, int ksel, hipStream_t st, uint32_t* bounds, int top_m) {
    if (Q <= 0) return 0;
    if (splits <= 32) {
        const dim3 grid((unsigned)((Q + 3) / 4)), block(4 * 64);
        if (ksel == 32) hipLaunchKernelGGL((topk_reduce_segs_kernel<32, 1>), grid, block, 0, st, seg, seg_cnt, splits, prelist, out, Q, bounds, top_m);
        else hipLaunchKernelGGL((topk_reduce_segs_kernel<64, 1>), grid, block, 0, st, seg, seg_cnt, splits, prelist, out, Q, bounds, top_m);
    } else {
        const dim3 grid((unsigned)Q), block(S256_RW * 64);
        if (ksel == 32) hipLaunchKernelGGL((topk_reduce_segs_kernel<32, S256_RW>), grid, block, 0, st, seg, seg_cnt, splits, prelist, out, Q, bounds, top_m);
        else hipLaunchKernelGGL((topk_reduce_segs_kernel<64, S256_RW>), grid, block, 0, st, seg, seg_cnt, splits, prelist, out, Q, bounds, top_m);
    }
    REVO_HIP_CHECK(hipGetLastError());
    return 0;
}

#ifdef REVO_EXPERIMENTS
static int g_scan_dbg = 0;
static unsigned long long* g_scan_stats = nullptr;
unsigned long long* topk_scan256_stats() {
    if (!g_scan_stats) { (void)hipMalloc((void**)&g_scan_stats, 64); (void)hipMemset(g_scan_stats, 0, 64); }
    return g_scan_stats;
}
void topk_scan256_set_debug(int d) { g_scan_dbg = d; }
#else
constexpr int g_scan_dbg = 0;
static unsigned long long* topk_scan256_stats() { return nullptr; }
#endif
int topk_scan256_hist_buckets() { return S256_NB; }
int topk_scan256_hist_shift() { return S256_SH; }

// Slices per query tile for up to 7 query tiles (and for the last, unpinned phase of a larger launch).  Workgroups run
// one per CU in rounds; a slice costs its tiles plus about one tile time of fixed work (pipeline fill, refreshes, the
// tail), and the slices are dealt out evenly, so the phase takes about  rounds x (ceil(tiles / s) + 1)  tile times.
// Fewer, longer slices on a tie.
static int scan256_best_splits(int qtiles, long tiles, double* cost_out) {
    int best = 1;
    double best_cost = 1e300;
    for (int s = 1; s <= 512; ++s) {
        if (s > tiles) break;
        const long per = (tiles + s - 1) / s;
        if (s > 1 && per < 3) break;
        const long rounds = ((long)qtiles * s + 255) / 256;
        const double cost = (double)rounds * ((double)per + 1.0);
        if (cost < best_cost - 1e-9) { best_cost = cost; best = s; }
    }
    if (cost_out) *cost_out = best_cost;
    return best;
}
// The phases of a scan launch.  With 8 query tiles and more, query tiles are pinned to XCDs (kernel comment) and the
// slices of an XCD's query tiles have to line up, so an XCD's 32 CUs are all busy only when it holds a = 1, 2, 4, 8, 16 or
// 32 query tiles (32 / a slices each).  The query tiles are therefore taken in phases of 8 a, largest first -- 39 query
// tiles: 32 as 4 per XCD x 8 slices, then the other 7 with the slices of a small launch -- each phase one even round of
// workgroups; the hardware starts a phase's blocks as the previous phase's end.  (Rounds 2-4 ran all query tiles in one
// phase, 5 per XCD x 6 slices = 30 of 32 CUs and 24 on the eighth XCD: 615 tile times where 39 x 3 875 / 256 = 590 is even,
// 78 against 69.5 on a shard of an eighth.)
struct Scan256Plan { int nph, splits; int q0[S256_PHASES], qn[S256_PHASES], ns[S256_PHASES]; double cost; };
static void scan256_plan(int qtiles, long tiles, Scan256Plan& pl) {
    pl = Scan256Plan{};
    int q = 0;
    auto add = [&](int qn, int ns, double cost) {
        pl.q0[pl.nph] = q; pl.qn[pl.nph] = qn; pl.ns[pl.nph] = ns; ++pl.nph;
        pl.splits = ns > pl.splits ? ns : pl.splits;
        pl.cost += cost;
        q += qn;
    };
    while (qtiles - q >= 8 && pl.nph < S256_PHASES - 1) {
        int a = 1;
        while (a < 32 && 16 * a <= qtiles - q) a *= 2;                 // the largest power of two with 8 a <= what is left
        int rounds = 1;
        if (a == 32) rounds = (qtiles - q) / 256;                       // whole rounds of one query tile per CU
        long ns = 32 / a;
        if (ns > tiles) ns = tiles;
        while (ns > 1 && (tiles + ns - 1) / ns < 3) --ns;               // (a gallery of a few tiles: fewer, longer slices)
        add(8 * a * rounds, (int)ns, (double)rounds * ((double)((tiles + ns - 1) / ns) + 1.0));
    }
    if (q < qtiles) {
        double c = 0.0;
        const int ns = scan256_best_splits(qtiles - q, tiles, &c);
        add(qtiles - q, ns, c);
    }
}
int launch_topk_scan256(const bf16_t* Qb, long ldq, const bf16_t* Gb, long ldg, int Q, long N, int D, long n_begin,
                        int splits, uint64_t* seg, int* seg_cnt, uint32_t* tau_g, const uint32_t* tau_base, uint32_t* hist,
                        int ksel, hipStream_t st, const float* marg, int* dropflag) {
    REVO_REQUIRE(ksel == 32 || ksel == 64, "search: the 256 x 256 scan keeps 32 or 64 candidates per query");
    REVO_REQUIRE(D % 64 == 0 && ldq % 8 == 0 && ldg % 8 == 0, "search: D must be a multiple of 64");
    REVO_REQUIRE(N < (1ll << 32), "search: a shard holds at most 2^32 rows");
    REVO_REQUIRE(256l * ldg * 2 < (1l << 31) && 256l * ldq * 2 < (1l << 31), "search: row too long for the DMA window");
    REVO_REQUIRE(splits >= 1 && splits <= 65535, "search: bad slice count");
    const long tiles = (N - n_begin + 255) / 256;
    if (Q <= 0 || N <= n_begin) return 0;
    const int qtiles = (Q + 255) / 256;
    Scan256Plan pl;
    scan256_plan(qtiles, tiles, pl);
    REVO_REQUIRE(splits == pl.splits, "search: the slice count is not the one topk_scan256_splits() gave");
    Scan256Args a{};
    a.Qb = Qb; a.ldq = ldq; a.Gb = Gb; a.ldg = ldg; a.Q = Q; a.N = N; a.D = D; a.n_begin = n_begin; a.splits = splits;
    a.nph = pl.nph;
    long blocks = 0;
    for (int i = 0; i < pl.nph; ++i) {
        const long per = (tiles + pl.ns[i] - 1) / pl.ns[i];
        REVO_REQUIRE(per * 256 <= (1l << 24), "search: a gallery slice holds at most 2^24 rows; use more splits");
        a.ph_first[i] = (int)blocks; a.ph_q0[i] = pl.q0[i]; a.ph_qn[i] = pl.qn[i]; a.ph_ns[i] = pl.ns[i];
        blocks += (long)pl.qn[i] * pl.ns[i];
        if (i + 1 < pl.nph) blocks = (blocks + 7) / 8 * 8;       // (pinned phases are multiples of 8 blocks anyway)
    }
    REVO_REQUIRE(blocks < (1l << 31), "search: too many scan workgroups");
    a.ph_first[pl.nph] = (int)blocks;
    a.seg = seg; a.seg_cnt = seg_cnt; a.tau_g = tau_g; a.tau_base = tau_base; a.hist = hist; a.marg = marg; a.dropflag = dropflag;
    a.dbg = g_scan_dbg; a.stats = (g_scan_dbg & 2) ? topk_scan256_stats() : nullptr;
    const dim3 grid((unsigned)blocks), block(G256_THREADS);
#define S256_LAUNCH_M(KS, RW, MG)                                                                              \
    do {                                                                                                       \
        REVO_FUNC_LDS((topk_scan256_kernel<KS, RW, MG>), S256_LDS);                                              \
        hipLaunchKernelGGL((topk_scan256_kernel<KS, RW, MG>), grid, block, S256_LDS, st, a);                   \
    } while (0)
    // the margin form exists for 64-candidate scans only (searches with k > 25: api.hip)
    REVO_REQUIRE(!marg || (ksel == 64 && dropflag), "search: the admission margin goes with 64 candidates and drop flags");
#define S256_LAUNCH(KS, RW)                                                                                    \
    do {                                                                                                       \
        if (KS == 64 && RW != 192 && marg) S256_LAUNCH_M(64, (RW == 192 ? 0 : RW), true);                      \
        else S256_LAUNCH_M(KS, RW, false);                                                                     \
    } while (0)
    // (the 192-row form is not built with the margin: hipcc 7.2 spills 464 VGPRs in that combination; 129..192 queries of a
    //  margin scan take the 256-row form)
    const int rows_mode = Q <= 64 ? 64 : (Q <= 128 ? 128 : ((Q <= 192 && !marg) ? 192 : 0));
    if (ksel == 32) {
        if (rows_mode == 64) S256_LAUNCH(32, 64);
        else if (rows_mode == 128) S256_LAUNCH(32, 128);
        else if (rows_mode == 192) S256_LAUNCH(32, 192);
        else S256_LAUNCH(32, 0);
    } else {
        if (rows_mode == 64) S256_LAUNCH(64, 64);
        else if (rows_mode == 128) S256_LAUNCH(64, 128);
        else if (rows_mode == 192) S256_LAUNCH(64, 192);
        else S256_LAUNCH(64, 0);
    }
#undef S256_LAUNCH
#undef S256_LAUNCH_M
    REVO_HIP_CHECK(hipGetLastError());
    return 0;
}

// the phases of the launch for Q queries over `rows` scanned rows (host logic only; tests/test_host_logic.py through the
// experiment library): out[0] = phases, out[1] = segment slots per query, then per phase: first block, first query tile,
// query tiles, slices.  Returns the number of blocks.
long topk_scan256_plan_dump(int Q, long rows, long* out, int cap) {
    const long tiles = (rows + 255) / 256;
    Scan256Plan pl;
    scan256_plan((Q + 255) / 256, tiles < 1 ? 1 : tiles, pl);
    long blocks = 0;
    if (cap >= 2) { out[0] = pl.nph; out[1] = pl.splits; }
    for (int i = 0; i < pl.nph; ++i) {
        if (cap >= 2 + 4 * (i + 1)) { out[2 + 4 * i] = blocks; out[3 + 4 * i] = pl.q0[i]; out[4 + 4 * i] = pl.qn[i]; out[5 + 4 * i] = pl.ns[i]; }
        blocks += (long)pl.qn[i] * pl.ns[i];
        if (i + 1 < pl.nph) blocks = (blocks + 7) / 8 * 8;
    }
    return blocks;
}
int topk_scan256_splits(int Q, long rows) {
    const long tiles = (rows + 255) / 256;
    if (tiles <= 0) return 1;
    Scan256Plan pl;
    scan256_plan((Q + 255) / 256, tiles, pl);
    return pl.splits;
}
// A query count that leaves a mostly empty last query tile (10 000 = 39 x 256 + 16) pays a whole tile row of MFMA work
// for those few queries, and one more query tile can halve the slices the busiest XCD has room for (2064 queries =
// 9 query tiles: one XCD holds two of them; 2048 + a tail of 16: one each).  The ragged
// tail (<= 128 queries) then goes to its own launch in the 64- / 128-row mode of the kernel, which runs at the HBM
// rate: about 0.006 tile times per gallery tile.  Returns the number of queries of the main launch (Q = no split).
int topk_scan256_main_queries(int Q, long rows) {
    const long tiles = (rows + 255) / 256;
    const int rem = Q % 256;
    if (tiles <= 0 || Q <= 256 || rem == 0 || rem > 128) return Q;
    Scan256Plan whole, main;
    scan256_plan((Q + 255) / 256, tiles, whole);
    scan256_plan(Q / 256, tiles, main);
    const double tail = 0.006 * (double)tiles + 0.5;
    return main.cost + tail < whole.cost ? Q - rem : Q;
}

// ---------------------------------------------------------------- the collect pass ----
// Fallback of the exactness certificate (topk_exact.hip): for the queries the finish step could not certify, every
// gallery row whose bf16 score reaches the query's bound lb (= the score a row needs in fp32 to enter the result,
// minus the rigorous bound of |bf16 score - fp32 score|) is appended to the query's collect list; the fp32 re-score of
// that list is then exact, whatever the scan's ksel candidates were.  Same main loop as the scan, and its selection
// with everything adaptive removed: the bound is constant, survivors of a pass are staged in LDS and thread i appends
// entry i to its query's list (slot from a global counter; a list that runs full only keeps counting: that query goes
// to the brute-force pass).  A pass that stages more than the buffer holds is discarded and the tile recomputed by
// column groups, as in the scan.  The number of queries is read from device memory (the launch is sized for all Q
// queries of the search; workgroups of query tiles past the count exit at once): no host round trip in a search.
constexpr int C256_LDS = G256_LDS + S256_STG * 8 + 256 * 4 + 64;
constexpr uint32_t C256_STG_OFF = G256_LDS;
constexpr uint32_t C256_TAU_OFF = C256_STG_OFF + S256_STG * 8;
constexpr uint32_t C256_CTRL_OFF = C256_TAU_OFF + 256 * 4;          // [0] running total of staged survivors

// the finest column classes (col & 63) that a pass (groups, grp) of the retry ladder covers: col & (groups - 1) == grp
__device__ inline uint64_t c256_class_mask(int groups, int grp) {
    uint64_t m = 0;
    for (int f = grp; f < S256_MAXGROUPS; f += groups) m |= 1ull << f;
    return m;
}
static_assert(S256_MAXGROUPS == 64, "the collect pass keeps one bit per finest column class in a 64-bit mask");

template <int ROWS>
__global__ __launch_bounds__(G256_THREADS, 2) void topk_collect256_kernel(Collect256Args p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* tau = (float*)(smem + C256_TAU_OFF);
    uint32_t* ctrl = (uint32_t*)(smem + C256_CTRL_OFF);
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int lane = tid & 63;
    const int nq = *p.n_q;
    // three instantiations are launched when the search has more than 64 queries; the entry count picks the one that
    // works: <= 64 entries run at the HBM rate (ROWS = 64), <= 128 nearly (ROWS = 128), more at the MFMA rate
    if (ROWS == 64 ? nq > 64 : (ROWS == 128 ? (nq <= 64 || nq > 128) : (nq <= 128 && p.small_modes))) return;
    const int sp = blockIdx.x;                              // one workgroup per gallery slice; it walks the query tiles
    const long tiles = (p.N + 255) / 256;
    const long per = tiles / p.splits, rem = tiles - per * p.splits;
    const long t0 = sp * per + (sp < rem ? sp : rem);
    const long t1 = t0 + per + (sp < rem ? 1 : 0);
    if (t0 >= t1) return;
    const long row_begin = t0 * 256;
    const uint32_t idx_base = (uint32_t)row_begin;
    for (int q0 = 0; q0 < nq; q0 += 256) {
    const int qvalid = (nq - q0) < 256 ? (nq - q0) : 256;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                                        // the previous query tile's last reads of tau / the stage
    if (tid < 256) tau[tid] = tid < qvalid ? p.lb[q0 + tid] : INFINITY;
    if (tid == 0) ctrl[0] = 0u;
    __syncthreads();

    G256Operand A, B;
    g256_operand_init(A, p.Qb, p.ldq, nq, q0, wave, lane);
    g256_operand_init(B, p.Gb + row_begin * p.ldg, p.ldg, p.N - row_begin, 0, wave, lane);
    g256_issue_prologue(A, B, smem, p.D, wave);

    long t = t0;
    int groups = 1, grp = 0;
    // column classes (col & 63: the finest level of the ladder) of the CURRENT tile whose survivors are already in the
    // lists.  A pass that did not overflow appends at once; when a later group of the same tile overflows and the
    // ladder deepens, the classes of the passes before it must not be appended again (the exact finish re-scores
    // every list entry and assumes each row appears once: a repeated row would take two places of a result).
    uint64_t done = 0;
    uint32_t staged_before = 0;
    while (t < t1) {
        const long n0 = t * 256;
        {
            f32x4 acc[8][4];
#pragma unroll
            for (int m = 0; m < 8; ++m)
#pragma unroll
                for (int n = 0; n < 4; ++n) acc[m][n] = (f32x4){0.f, 0.f, 0.f, 0.f};
            gemm256_mainloop<ROWS>(A, B, smem, p.D, wave, lane, acc);
            if (groups == 1 && t + 1 < t1) {
                g256_operand_init(B, p.Gb + (n0 + 256) * p.ldg, p.ldg, p.N - (n0 + 256), 0, wave, lane);
                g256_issue_prologue(A, B, smem, p.D, wave);
            }
            asm volatile("" : "+v"(lane) :: "memory");
            const int lr = lane & 15, lq = lane >> 4;
            const int rbase = (wave >> 2) * 128 + lr;
            const int cbase = (wave & 3) * 64 + lq * 4;
            const long left = p.N - n0;
            const uint32_t rel0 = (uint32_t)(n0 - row_begin);
            if (left < 256) {
#pragma unroll
                for (int n = 0; n < 4; ++n)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const bool past = cbase + n * 16 + j >= left;
#pragma unroll
                        for (int m = 0; m < 8; ++m) acc[m][n][j] = past ? __builtin_nanf("") : acc[m][n][j];
                    }
            }
            float taum[8];
            unsigned hitm = 0;
#pragma unroll
            for (int m = 0; m < 8; ++m) taum[m] = s256_lds_f32(C256_TAU_OFF + (rbase + m * 16) * 4);
#pragma unroll
            for (int m = 0; m < 8; ++m) {
                float mx = -INFINITY;
#pragma unroll
                for (int n = 0; n < 4; ++n)
#pragma unroll
                    for (int j = 0; j < 4; ++j) mx = fmaxf(mx, acc[m][n][j]);
                if (__ballot(mx >= taum[m]) != 0ull) hitm |= 1u << m;
            }
            if (hitm) {
#pragma unroll
                for (int m = 0; m < 8; ++m) {
                    if (!(hitm & (1u << m))) continue;          // wave-uniform
                    const int row = rbase + m * 16;
#pragma unroll
                    for (int n = 0; n < 4; ++n)
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const float v = acc[m][n][j];
                            const int col = cbase + n * 16 + j;
                            const bool pass = v >= taum[m];
                            if (__ballot(pass) == 0ull) continue;
                            if (pass && (col & (groups - 1)) == grp && !((done >> (col & 63)) & 1ull)) {
                                const uint32_t pos = (uint32_t)s256_lds_inc(C256_CTRL_OFF) - staged_before;
                                if (pos < (uint32_t)S256_STG) s256_lds_store64(C256_STG_OFF + pos * 8, s256_entry(row, v, rel0 + col));
                            }
                        }
                }
            }
        }
        s256_barrier_lds();
        const uint32_t staged_total = s256_lds_u32(C256_CTRL_OFF);
        const uint32_t staged = staged_total - staged_before;
        staged_before = staged_total;
        const bool overflow = staged > (uint32_t)S256_STG;
        if (!overflow) {
            for (uint32_t i = tid; i < staged; i += 512) {
                const uint64_t e = s256_lds_u64(C256_STG_OFF + i * 8);
                const int row = (int)(e >> 56);
                const int slot = atomicAdd(p.cnt + q0 + row, 1);
                if (slot < p.cap) p.col[(long)(q0 + row) * p.cap + slot] = s256_entry_to_key(e, idx_base);
            }
        }
        s256_barrier_lds();
        if (groups == 1 && !overflow) { ++t; continue; }
        if (overflow) {
            groups = groups < S256_MAXGROUPS ? groups * 2 : S256_MAXGROUPS;
            grp = 0;
        } else {
            done |= c256_class_mask(groups, grp);            // this pass's columns are in the lists now
            ++grp;
        }
        // the next group of the (possibly deeper) ladder that still has columns to append
        while (grp < groups && (c256_class_mask(groups, grp) & ~done) == 0ull) ++grp;
        if (grp == groups) {
            groups = 1;
            grp = 0;
            done = 0;
            ++t;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (t < t1) {
            const long nn = t * 256;
            g256_operand_init(B, p.Gb + nn * p.ldg, p.ldg, p.N - nn, 0, wave, lane);
            g256_issue_prologue(A, B, smem, p.D, wave);
        }
    }
    }
}

// slices of the collect pass: enough to fill the chip with ONE query tile (the usual case: a handful of uncertified
// queries), at least three gallery tiles per slice, and never more than 2^24 rows in a slice (24-bit staging index)
int topk_collect256_splits(long N) {
    const long tiles = (N + 255) / 256;
    if (tiles <= 0) return 1;
    long s = tiles / 3;
    s = s < 1 ? 1 : (s > 256 ? 256 : s);
    const long min_s = (tiles + 65535) / 65536;
    return (int)(s < min_s ? min_s : s);
}
int launch_topk_collect256(const Collect256Args& a_in, int max_queries, hipStream_t st) {
    Collect256Args a = a_in;
    REVO_REQUIRE(a.D % 64 == 0 && a.ldq % 8 == 0 && a.ldg % 8 == 0, "search: D must be a multiple of 64");
    REVO_REQUIRE(a.N < (1ll << 32), "search: a shard holds at most 2^32 rows");
    REVO_REQUIRE(256l * a.ldg * 2 < (1l << 31) && 256l * a.ldq * 2 < (1l << 31), "search: row too long for the DMA window");
    if (max_queries <= 0 || a.N <= 0) return 0;
    a.splits = topk_collect256_splits(a.N);
    const dim3 grid((unsigned)a.splits), block(G256_THREADS);
#define C256_LAUNCH(RW)                                                                           \
    do {                                                                                          \
        REVO_FUNC_LDS((topk_collect256_kernel<RW>), C256_LDS);                                    \
        hipLaunchKernelGGL((topk_collect256_kernel<RW>), grid, block, C256_LDS, st, a);           \
    } while (0)
    a.small_modes = 1;
    C256_LAUNCH(64);
    if (max_queries > 64) C256_LAUNCH(128);
    if (max_queries > 128) C256_LAUNCH(0);
#undef C256_LAUNCH
    REVO_HIP_CHECK(hipGetLastError());
    return 0;
}

}  // namespace revo
