// 256 x 256 x 64 bf16 MFMA main loop, 8 waves (2 M x 4 N, 128 x 64 per wave),
// one workgroup per CU (128 KiB LDS), shared by the ViT linear layers and the
// query x gallery scan.   C[M,N] = A[M,K] . B[N,K]^T, both operands K-contiguous.
//
// Schedule (per K-tile t living in LDS stage t&1; fragments: A-lo/A-hi = the wave's
// first/second 64 rows, B-lo/B-hi = its first/second 32 columns):
//
//            MFMA cluster (16x)     LDS reads issued          LDS-DMA issued
//   even P0  A-lo x B-lo           A-hi(t)                   B half 0 of t+1 -> other stage
//        P1  A-hi x B-lo           B-hi(t)                   B half 1 of t+1 -> other stage
//        P2  A-hi x B-hi           -            [barrier]    A half 0 of t+2 -> this stage
//        P3  A-lo x B-hi           first operands of t+1     A half 1 of t+2 -> this stage
//                                  (after vmcnt(4)+barrier)
//   odd tiles run the mirrored order (A-hi first) so that the registers a cluster
//   frees are the ones the next reads fill: every ds_read is issued one cluster
//   ahead of its use, and the DMA queue never drains inside the loop (counted
//   vmcnt(4) once per K-tile leaves two half-tiles in flight across the barrier).
//
// LDS image: A stage 0 | A stage 1 | B stage 0 | B stage 1, each [256 rows][128 B] (the two stages of
// an operand are 32 KiB apart, inside the 16-bit immediate of ds_read); 16-byte chunks are
// XOR-swizzled with ((row>>1)&7) on the DMA *source* address and on the fragment
// read (the DMA destination is lane-linear).  Rows past the matrix edge are
// fetched through a bounds-checked buffer descriptor and read as zeros.
#pragma once
#include "common.h"

namespace revo {

constexpr int G256_THREADS = 512;
constexpr int G256_LDS = 131072;        // A stage 0 | A stage 1 | B stage 0 | B stage 1, 32 KiB each
#define G256_A(smem, s) ((smem) + (s) * 32768)
#define G256_B(smem, s) ((smem) + 65536 + (s) * 32768)

struct G256Operand {
    __amdgpu_buffer_rsrc_t rsrc[4];   // one bounds-checked window per 64-row block of the tile (SGPRs)
    uint32_t voff;                    // byte offset of this lane's 16-byte chunk inside a block at k = 0
};

// base: first element of the operand, rows: its row count, row0: first row of this tile.
// Block j's descriptor covers rows [row0 + 64 j, min(row0 + 64 j + 64, rows)): a lane whose
// row lies past the matrix edge addresses beyond num_records and the DMA writes zeros.
// One VGPR of addressing per operand; every window is < 4 GiB whatever the operand size.
// tile_rows < 256 (a 192- or 208-row tile): the rows past it get empty windows -- their DMA moves no data and writes zeros.
__device__ __forceinline__ void g256_operand_init(G256Operand& op, const bf16_t* base, long ld, long rows, int row0,
                                                  int wave, int lane, int tile_rows = 256) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        long left = rows - (row0 + 64 * j);
        left = left < 0 ? 0 : (left > 64 ? 64 : left);
        if (left > tile_rows - 64 * j) left = tile_rows - 64 * j < 0 ? 0 : tile_rows - 64 * j;
        op.rsrc[j] = __builtin_amdgcn_make_buffer_rsrc((void*)(base + (long)(row0 + 64 * j) * ld), 0,
                                                       (int)(left * ld * 2), 0x00020000);
    }
    const int r = wave * 8 + (lane >> 3);                  // row inside a 64-row block
    const int c = (lane & 7) ^ ((r >> 1) & 7);             // (row>>1)&7 is the same in every 64-row block
    op.voff = (uint32_t)(r * ld * 2 + c * 16);
}

// one half-tile (128 rows x 64 k): two 16-byte-per-lane DMA instructions per thread
// AUX: the loads' cache policy (0 = default; 2 = non-temporal: bytes that ONE workgroup reads ONCE -- the gallery stream of a
// scan with a single query tile, topk256.hip; never operands that other workgroups re-read from L2)
template <int AUX = 0>
__device__ __forceinline__ void g256_issue_half(const G256Operand& op, int half, int kbyte, char* region, int wave) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        char* dst = region + half * 16384 + i * 8192 + wave * 1024;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(op.rsrc[half * 2 + i], (__attribute__((address_space(3))) void*)dst, 16,
                                                 op.voff, kbyte, 0, AUX);
    }
}

struct G256Frags {
    bf16x8 alo[4][2], ahi[4][2], blo[2][2], bhi[2][2];   // [fragment][k sub-step]
};

// Per-lane LDS byte offsets of the fragment reads: [stage][k sub-step], for the wave's first
// A row / first B row.  Everything else is an immediate (m * 2048, +64 rows = 8192, ...), so
// the whole main loop addresses LDS through these four registers (made opaque so that the
// compiler does not re-derive a register per immediate and spill at the 256-VGPR limit).
struct G256Addr {
    uint32_t a[2], b[2];    // [k sub-step], stage 0; stage 1 is +32768 (an immediate)
};
__device__ __forceinline__ void g256_addr_init(G256Addr& ad, int wave, int lane) {
    const int sw = (lane >> 1) & 7, lr = lane & 15, lq = lane >> 4;
    const int wr = wave >> 2, wc = wave & 3;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
        const uint32_t x = (uint32_t)(((kk * 4 + lq) ^ sw) << 4);
        ad.a[kk] = (uint32_t)((wr * 128 + lr) * 128) + x;
        ad.b[kk] = (uint32_t)(65536 + (wc * 64 + lr) * 128) + x;
        asm volatile("" : "+v"(ad.a[kk]), "+v"(ad.b[kk]));
    }
}
typedef __attribute__((address_space(3))) const bf16x8* lds_frag_ptr;
// ROWS: 0 = A-lo / B-lo, 64 = A-hi (rows +64), 32 = B-hi (rows +32)
template <int ROWS, int STAGE>
__device__ __forceinline__ void g256_read_a(bf16x8 (&dst)[4][2], const uint32_t (&base)[2]) {
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
            dst[m][kk] = *(lds_frag_ptr)(uintptr_t)(base[kk] + STAGE * 32768 + (ROWS + m * 16) * 128);
}
template <int ROWS, int STAGE>
__device__ __forceinline__ void g256_read_b(bf16x8 (&dst)[2][2], const uint32_t (&base)[2]) {
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
            dst[n][kk] = *(lds_frag_ptr)(uintptr_t)(base[kk] + STAGE * 32768 + (ROWS + n * 16) * 128);
}

// acc[m][n]: lane l owns row m*16 + (l&15), columns n*16 + (l>>4)*4 + {0..3} (operands swapped in the MFMA)
template <int M0, int N0>
__device__ __forceinline__ void g256_cluster(const bf16x8 (&a)[4][2], const bf16x8 (&b)[2][2], f32x4 (&acc)[8][4]) {
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int n = 0; n < 2; ++n)
                acc[M0 + m][N0 + n] =
                    __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[n][kk], a[m][kk], acc[M0 + m][N0 + n], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
}

// first fragment only (the 16 extra rows of a 208-row tile)
template <int ROWS, int STAGE>
__device__ __forceinline__ void g256_read_a1(bf16x8 (&dst)[4][2], const uint32_t (&base)[2]) {
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) dst[0][kk] = *(lds_frag_ptr)(uintptr_t)(base[kk] + STAGE * 32768 + ROWS * 128);
}
template <int M0, int N0>
__device__ __forceinline__ void g256_cluster1(const bf16x8 (&a)[4][2], const bf16x8 (&b)[2][2], f32x4 (&acc)[8][4]) {
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int n = 0; n < 2; ++n)
            acc[M0][N0 + n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[n][kk], a[0][kk], acc[M0][N0 + n], 0, 0, 0);
}

#define G256_FENCE() __builtin_amdgcn_sched_barrier(0)

// DMA for the first 1.5 K-tiles of an output tile (tile 0 complete, A of tile 1).  May be
// issued while the previous output tile's epilogue is still running: the LDS image is free
// once gemm256_mainloop has returned (it ends with a barrier behind every wave's last read).
template <int BAUX = 0>
__device__ __forceinline__ void g256_issue_prologue(const G256Operand& A, const G256Operand& B, char* smem, int K,
                                                    int wave) {
    g256_issue_half(A, 0, 0, G256_A(smem, 0), wave);
    g256_issue_half(A, 1, 0, G256_A(smem, 0), wave);
    g256_issue_half<BAUX>(B, 0, 0, G256_B(smem, 0), wave);
    g256_issue_half<BAUX>(B, 1, 0, G256_B(smem, 0), wave);
    if (K > 64) {
        g256_issue_half(A, 0, 128, G256_A(smem, 1), wave);
        g256_issue_half(A, 1, 128, G256_A(smem, 1), wave);
    }
}
// for gemm256_mainloop<ROWS, RT, DEEP = true>: the first TWO K-tiles complete
template <int BAUX = 0>
__device__ __forceinline__ void g256_issue_prologue_deep(const G256Operand& A, const G256Operand& B, char* smem, int K,
                                                         int wave) {
    g256_issue_prologue<BAUX>(A, B, smem, K, wave);
    if (K > 64) {
        g256_issue_half<BAUX>(B, 0, 128, G256_B(smem, 1), wave);
        g256_issue_half<BAUX>(B, 1, 128, G256_B(smem, 1), wave);
    }
}

// K % 64 == 0, K >= 64.  acc must be zero-initialised (or hold the running sum) by the caller,
// and g256_issue_prologue(A, B, ...) must have been issued by this wave (any vector-memory
// operations issued after it only make the first wait below more conservative).
// ROWS = 64 / 128: only A rows 0..63 / 0..127 of the tile carry data (a search with few queries): the
// waves of the second wave-row (and, at 64, the A-hi halves of the first) skip their LDS reads and
// MFMAs (their accumulators stay 0) but keep issuing DMA and meeting the barriers, so the loop runs
// at the DMA / HBM rate instead of the MFMA rate.
// ROWS = 192: a 192-row output tile (the linear layers use it where 192-row tiles make whole rounds of workgroups and
// 256-row ones do not): the second wave-row skips its A-hi halves only.  Waves w and w + 4 share a SIMD, so every SIMD
// carries one full and one half wave tile: 3/4 of the MFMA work of a 256-row tile, evenly spread.  With `tall` (wave
// uniform) the tile has 208 rows: the second wave-row also takes the first fragment of its A-hi half (rows 192..207),
// 1/16 of a tile's work more -- how the launcher places a few leftover rows without a launch of their own.
// RT = true: the row mode is a run-time, wave-uniform choice instead (rt_lo / rt_hi: this wave computes its first / second
// 64 rows; `tall`: only the first fragment of its second 64) -- the phased persistent kernel, whose workgroups cut their
// first tile into two pieces of different heights (gemm.hip gemm256pp_kernel).  Same instruction stream per active part,
// so every row's result has the bits the fixed modes give it.
// DEEP = true (the HBM-bound row modes of the scan): BOTH operands of K-tile t + 2 are requested behind the barrier of
// tile t's third phase, when every wave has finished reading the stage they go to -- a K-tile and a half ahead of their
// use -- and nothing in the first two phases; the wait leaves those eight DMA instructions in flight (vmcnt(8)).  In the
// ordinary schedule B of tile t + 1 is requested half a tile ahead: with MFMA work for 64 query rows only a K-tile takes
// about as long as the CU's share of HBM delivers its 32 KiB (1.3 us), less than a DMA's latency, and the loop ran at
// one gallery K-tile per latency: 5.3 of the 6.3 TB/s a copy reaches.
// QS > 0 (queued stores: gemm.hip gemm256q_kernel): behind g256_issue_prologue the caller has issued exactly QS more
// vector-memory instructions -- the previous output tile's stores -- which the first wait leaves in flight:
// vmcnt(4 + QS) instead of vmcnt(4).  They retire with the wait at the end of K-tile 0 (B of tile 1, requested behind
// them inside the loop, is what that wait is for): one K-tile after they were issued instead of before the first MFMA.
// Nothing inside the loop changes.  (Requesting tile 1's B ahead of the stores as well gives them two K-tiles, but tile 0
// must then skip its B requests and count differently: with those run-time tests of t == 0 hipcc peeled the first trip,
// parked an accumulator in scratch around it and reloaded it behind an s_waitcnt vmcnt(0) -- once per output tile, a
// drain of the DMA queue where this mode is meant to remove one.)  Needs K >= 128.
template <int ROWS = 0, bool RT = false, bool DEEP = false, int QS = 0, int BAUX = 0>
__device__ __forceinline__ void gemm256_mainloop(const G256Operand& A, const G256Operand& B, char* smem, int K,
                                                 int wave, int lane, f32x4 (&acc)[8][4], bool tall = false,
                                                 bool rt_lo = true, bool rt_hi = true) {
    static_assert(ROWS == 0 || ROWS == 64 || ROWS == 128 || ROWS == 192, "");
    static_assert(!RT || ROWS == 0, "the run-time row mode has no compile-time one");
    static_assert(QS == 0 || (!DEEP && QS + 4 <= 63), "queued stores: the ordinary schedule, and a count vmcnt can hold");
    const bool wact = wave < 4;
#define G256_LO(...) do { if (RT ? rt_lo : (ROWS == 0 || ROWS == 192 || wact)) { __VA_ARGS__; } } while (0)
#define G256_HI(...) do { if (RT ? rt_hi : (ROWS == 0 || ((ROWS == 128 || ROWS == 192) && wact))) { __VA_ARGS__; } } while (0)
#define G256_HT(...) do { if (RT ? tall : (ROWS == 192 && !wact && tall)) { __VA_ARGS__; } } while (0)
    const int nt = K >> 6;
    G256Frags f;
    G256Addr ad;
    g256_addr_init(ad, wave, lane);

    if (QS) {
        asm volatile("s_waitcnt vmcnt(%0)" :: "n"(4 + QS) : "memory");
    } else if (nt > 1) {
        if (DEEP) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    G256_LO(g256_read_a<0, 0>(f.alo, ad.a););
    G256_LO(g256_read_b<0, 0>(f.blo, ad.b););
    G256_FENCE();

    // two K-tiles per trip; an odd trailing tile is peeled so the loop has a single exit
    int t = 0;
    for (; t + 1 < nt; t += 2) {
        // ------------------------------------------------------------ even tile t, stage 0
        {
            const bool n1 = t + 1 < nt, n2 = t + 2 < nt;
            // P0
            G256_HI(g256_read_a<64, 0>(f.ahi, ad.a););
            G256_HT(g256_read_a1<64, 0>(f.ahi, ad.a););
            if (!DEEP && n1) g256_issue_half<BAUX>(B, 0, (t + 1) * 128, G256_B(smem, 1), wave);
            G256_FENCE();
            G256_LO(g256_cluster<0, 0>(f.alo, f.blo, acc););
            G256_FENCE();
            // P1
            G256_LO(g256_read_b<32, 0>(f.bhi, ad.b););
            if (!DEEP && n1) g256_issue_half<BAUX>(B, 1, (t + 1) * 128, G256_B(smem, 1), wave);
            G256_FENCE();
            G256_HI(g256_cluster<4, 0>(f.ahi, f.blo, acc););
            G256_HT(g256_cluster1<4, 0>(f.ahi, f.blo, acc););
            G256_FENCE();
            // P2: every wave has retired its A reads of this stage -> refill its A halves
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            if (n2) g256_issue_half(A, 0, (t + 2) * 128, G256_A(smem, 0), wave);
            if (DEEP && n2) g256_issue_half<BAUX>(B, 0, (t + 2) * 128, G256_B(smem, 0), wave);
            G256_FENCE();
            G256_HI(g256_cluster<4, 2>(f.ahi, f.bhi, acc););
            G256_HT(g256_cluster1<4, 2>(f.ahi, f.bhi, acc););
            G256_FENCE();
            // P3: publish tile t+1, start reading it
            if (n2) g256_issue_half(A, 1, (t + 2) * 128, G256_A(smem, 0), wave);
            if (DEEP && n2) g256_issue_half<BAUX>(B, 1, (t + 2) * 128, G256_B(smem, 0), wave);
            if (n1) {
                if (n2) { if (DEEP) asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); }
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                G256_HI(g256_read_a<64, 1>(f.ahi, ad.a););          // odd tiles start with A-hi
                G256_HT(g256_read_a1<64, 1>(f.ahi, ad.a););
                G256_LO(g256_read_b<0, 1>(f.blo, ad.b););
            }
            G256_FENCE();
            G256_LO(g256_cluster<0, 2>(f.alo, f.bhi, acc););
            G256_FENCE();
        }
        // ------------------------------------------------------------ odd tile t+1, stage 1
        {
            const int u = t + 1;
            const bool n1 = u + 1 < nt, n2 = u + 2 < nt;
            // P0'
            G256_LO(g256_read_a<0, 1>(f.alo, ad.a););
            if (!DEEP && n1) g256_issue_half<BAUX>(B, 0, (u + 1) * 128, G256_B(smem, 0), wave);
            G256_FENCE();
            G256_HI(g256_cluster<4, 0>(f.ahi, f.blo, acc););
            G256_HT(g256_cluster1<4, 0>(f.ahi, f.blo, acc););
            G256_FENCE();
            // P1'
            G256_LO(g256_read_b<32, 1>(f.bhi, ad.b););
            if (!DEEP && n1) g256_issue_half<BAUX>(B, 1, (u + 1) * 128, G256_B(smem, 0), wave);
            G256_FENCE();
            G256_LO(g256_cluster<0, 0>(f.alo, f.blo, acc););
            G256_FENCE();
            // P2'
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            if (n2) g256_issue_half(A, 0, (u + 2) * 128, G256_A(smem, 1), wave);
            if (DEEP && n2) g256_issue_half<BAUX>(B, 0, (u + 2) * 128, G256_B(smem, 1), wave);
            G256_FENCE();
            G256_LO(g256_cluster<0, 2>(f.alo, f.bhi, acc););
            G256_FENCE();
            // P3'
            if (n2) g256_issue_half(A, 1, (u + 2) * 128, G256_A(smem, 1), wave);
            if (DEEP && n2) g256_issue_half<BAUX>(B, 1, (u + 2) * 128, G256_B(smem, 1), wave);
            if (n1) {
                if (n2) { if (DEEP) asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); }
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                G256_LO(g256_read_a<0, 0>(f.alo, ad.a););               // even tiles start with A-lo
                G256_LO(g256_read_b<0, 0>(f.blo, ad.b););
            }
            G256_FENCE();
            G256_HI(g256_cluster<4, 2>(f.ahi, f.bhi, acc););
            G256_HT(g256_cluster1<4, 2>(f.ahi, f.bhi, acc););
            G256_FENCE();
        }
    }
    if (t < nt) {
        // ------------------------------------------------------------ even tile t, stage 0
        {
            const bool n1 = t + 1 < nt, n2 = t + 2 < nt;
            // P0
            G256_HI(g256_read_a<64, 0>(f.ahi, ad.a););
            G256_HT(g256_read_a1<64, 0>(f.ahi, ad.a););
            if (!DEEP && n1) g256_issue_half<BAUX>(B, 0, (t + 1) * 128, G256_B(smem, 1), wave);
            G256_FENCE();
            G256_LO(g256_cluster<0, 0>(f.alo, f.blo, acc););
            G256_FENCE();
            // P1
            G256_LO(g256_read_b<32, 0>(f.bhi, ad.b););
            if (!DEEP && n1) g256_issue_half<BAUX>(B, 1, (t + 1) * 128, G256_B(smem, 1), wave);
            G256_FENCE();
            G256_HI(g256_cluster<4, 0>(f.ahi, f.blo, acc););
            G256_HT(g256_cluster1<4, 0>(f.ahi, f.blo, acc););
            G256_FENCE();
            // P2: every wave has retired its A reads of this stage -> refill its A halves
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            if (n2) g256_issue_half(A, 0, (t + 2) * 128, G256_A(smem, 0), wave);
            if (DEEP && n2) g256_issue_half<BAUX>(B, 0, (t + 2) * 128, G256_B(smem, 0), wave);
            G256_FENCE();
            G256_HI(g256_cluster<4, 2>(f.ahi, f.bhi, acc););
            G256_HT(g256_cluster1<4, 2>(f.ahi, f.bhi, acc););
            G256_FENCE();
            // P3: publish tile t+1, start reading it
            if (n2) g256_issue_half(A, 1, (t + 2) * 128, G256_A(smem, 0), wave);
            if (DEEP && n2) g256_issue_half<BAUX>(B, 1, (t + 2) * 128, G256_B(smem, 0), wave);
            if (n1) {
                if (n2) { if (DEEP) asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); }
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                G256_HI(g256_read_a<64, 1>(f.ahi, ad.a););          // odd tiles start with A-hi
                G256_HT(g256_read_a1<64, 1>(f.ahi, ad.a););
                G256_LO(g256_read_b<0, 1>(f.blo, ad.b););
            }
            G256_FENCE();
            G256_LO(g256_cluster<0, 2>(f.alo, f.bhi, acc););
            G256_FENCE();
        }
    }
    // all waves are past their last LDS read before the caller reuses the LDS image
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
#undef G256_LO
#undef G256_HI
#undef G256_HT
}

}  // namespace revo
