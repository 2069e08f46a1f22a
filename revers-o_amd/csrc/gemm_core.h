// bf16 MFMA GEMM main loop shared by the ViT linear layers (gemm.hip) and the
// query x gallery scan (topk.hip):   C[M,N] = A[M,K] . B[N,K]^T
// Both operands are K-contiguous (activations [rows][K], nn.Linear weights
// [out][in], gallery rows [N][D]), so one LDS image serves both.
//
// Tile: BM x BN x 64 per workgroup of 256 threads (4 waves), fp32 accumulate in
// v_mfma_f32_16x16x32_bf16.  Staging is direct global->LDS DMA
// (global_load_lds_dwordx4): one wave-instruction lands 8 tile rows x 128 B.
// The LDS image is lane-linear (DMA constraint); bank conflicts of the
// ds_read_b128 fragment reads are removed by XOR-swizzling the 16-byte chunk
// index with ((row>>1)&7) on the *global source* address and again on the read
// (cdna_hip_programming.md rule 21).
#pragma once
#include "common.h"

namespace revo {

constexpr int GEMM_BK = 64;        // bf16 elements per K-step: one 128-B LDS row
constexpr int GEMM_THREADS = 256;

__device__ __forceinline__ void glds16(const void* gsrc, void* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

// XCD-aware tile order: blocks b and b+8 share an XCD (and its L2); give each
// XCD a contiguous run of tile ids.  Bijective for any grid size.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, x = bid & 7;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (bid >> 3);
}

// Per-thread source pointers for one operand tile of ROWS rows.
template <int ROWS>
struct TileLoader {
    static constexpr int NI = ROWS / 32;   // DMA instructions per wave per K-step
    const bf16_t* src[NI];
    // rows >= nrows re-read the last valid row: their products are discarded by the epilogue
    __device__ __forceinline__ void init(const bf16_t* base, long ld, int row0, int nrows, int wave, int lane) {
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int r = (wave * NI + i) * 8 + (lane >> 3);
            const int c = (lane & 7) ^ ((r >> 1) & 7);
            int gr = row0 + r;
            gr = gr < nrows ? gr : nrows - 1;
            src[i] = base + (long)gr * ld + c * 8;
        }
    }
    __device__ __forceinline__ void issue(char* lds_tile, int k0, int wave) const {
#pragma unroll
        for (int i = 0; i < NI; ++i) glds16(src[i] + k0, lds_tile + (wave * NI + i) * 1024);
    }
};

// One 64-deep K-step on the wave's (MF*16) x (NF*16) sub-tile.
// acc[m][n] holds C^T fragments: lane l owns row (m*16 + (l&15)) and the four
// consecutive columns n*16 + (l>>4)*4 + {0..3}  (A and B swapped in the MFMA so
// that every lane's four results are adjacent in a C row).
template <int MF, int NF>
__device__ __forceinline__ void mma_kstep(const char* ldsA, const char* ldsB, int rowA0, int rowB0, int lane,
                                          f32x4 (&acc)[MF][NF]) {
    const int sw = (lane >> 1) & 7;
    const int lr = lane & 15, lq = lane >> 4;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
        const int ch = ((kk * 4 + lq) ^ sw) * 16;
        bf16x8 a[MF], b[NF];
#pragma unroll
        for (int m = 0; m < MF; ++m) a[m] = *(const bf16x8*)(ldsA + (rowA0 + m * 16 + lr) * 128 + ch);
#pragma unroll
        for (int n = 0; n < NF; ++n) b[n] = *(const bf16x8*)(ldsB + (rowB0 + n * 16 + lr) * 128 + ch);
#pragma unroll
        for (int m = 0; m < MF; ++m)
#pragma unroll
            for (int n = 0; n < NF; ++n)
                acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[n], a[m], acc[m][n], 0, 0, 0);
    }
}

// Double-buffered K loop: DMA of step t+1 is in flight while step t is multiplied.
// smem: 2 x (BM + BN) x 128 bytes.  K % 64 == 0.
template <int BM, int BN, int MF, int NF>
__device__ __forceinline__ void gemm_mainloop(const TileLoader<BM>& la, const TileLoader<BN>& lb, char* smem, int K,
                                              int wave, int lane, int rowA0, int rowB0, f32x4 (&acc)[MF][NF]) {
    constexpr int A_BYTES = BM * 128, STAGE = (BM + BN) * 128;
    const int nt = K / GEMM_BK;
    la.issue(smem, 0, wave);
    lb.issue(smem + A_BYTES, 0, wave);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int t = 0; t < nt; ++t) {
        char* cur = smem + (t & 1) * STAGE;
        char* nxt = smem + ((t + 1) & 1) * STAGE;
        if (t + 1 < nt) {
            la.issue(nxt, (t + 1) * GEMM_BK, wave);
            lb.issue(nxt + A_BYTES, (t + 1) * GEMM_BK, wave);
        }
        mma_kstep<MF, NF>(cur, cur + A_BYTES, rowA0, rowB0, lane, acc);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
}

// The same K loop with an NST-deep ring of stages instead of two: NST - 1 K-steps of DMA are in flight while one is
// multiplied.  For problems with about one tile per CU (one image: 577 rows; the leftover rows of a big GEMM) the K loop
// is a chain of DMA latencies (~1.1 us per step with two buffers: 18 us for K = 1024 whatever the tile computes); five
// loads in flight turn it into ~0.25 us per step.  The wait is a COUNTED vmcnt (a wave's DMA loads retire in order):
// step t may start when at most `ahead` later stages are still outstanding; the barrier is a raw s_barrier
// (__syncthreads() would drain every outstanding load).  smem: NST x (BM + BN) x 128 bytes.  K % 64 == 0.
template <int N>
__device__ __forceinline__ void wait_vmcnt_n() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <int BM, int BN, int MF, int NF, int NST>
__device__ __forceinline__ void gemm_mainloop_ring(const TileLoader<BM>& la, const TileLoader<BN>& lb, char* smem, int K,
                                                   int wave, int lane, int rowA0, int rowB0, f32x4 (&acc)[MF][NF]) {
    constexpr int A_BYTES = BM * 128, STAGE = (BM + BN) * 128;
    constexpr int NLD = TileLoader<BM>::NI + TileLoader<BN>::NI;      // DMA instructions per wave and stage
    static_assert(NST >= 3 && NST <= 6 && (NST - 2) * NLD <= 63, "vmcnt holds 6 bits");
    const int nt = K / GEMM_BK;
#pragma unroll
    for (int s = 0; s < NST - 1; ++s)
        if (s < nt) {
            la.issue(smem + s * STAGE, s * GEMM_BK, wave);
            lb.issue(smem + s * STAGE + A_BYTES, s * GEMM_BK, wave);
        }
    int buf = 0;                                                      // stage that holds K-step t
    for (int t = 0; t < nt; ++t) {
        const int left = nt - 1 - t;
        const int ahead = left < NST - 2 ? left : NST - 2;             // later stages that may still be in flight
        switch (ahead) {
            case 0: wait_vmcnt_n<0>(); break;
            case 1: wait_vmcnt_n<NLD>(); break;
            case 2: wait_vmcnt_n<(NST > 3 ? 2 : 0) * NLD>(); break;
            case 3: wait_vmcnt_n<(NST > 4 ? 3 : 0) * NLD>(); break;
            default: wait_vmcnt_n<(NST > 5 ? 4 : 0) * NLD>(); break;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();          // everybody's share of stage t has landed; everybody is done reading stage t - 1
        const int tn = t + NST - 1;
        if (tn < nt) {                          // refill the stage that was read in step t - 1
            const int nb = buf == 0 ? NST - 1 : buf - 1;
            la.issue(smem + nb * STAGE, tn * GEMM_BK, wave);
            lb.issue(smem + nb * STAGE + A_BYTES, tn * GEMM_BK, wave);
        }
        mma_kstep<MF, NF>(smem + buf * STAGE, smem + buf * STAGE + A_BYTES, rowA0, rowB0, lane, acc);
        buf = buf + 1 == NST ? 0 : buf + 1;
    }
}

}  // namespace revo
