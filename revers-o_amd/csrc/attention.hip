// Multi-head attention of the PE body (K6: softmax(q k^T / sqrt(hd)) v, flash
// style, never materialising S x S) and the single-probe attention-pool head
// (K9).  Replaces the SDPA / nn.MultiheadAttention calls inside upstream
// encode_image (reference call site core_system.py:442).
//
// Body kernel: one workgroup = up to 8 waves of one (image, head); each wave owns
// 32 query rows.  K/V tiles of 64 keys arrive by LDS-DMA into a ring of swizzled
// LDS images.  Scores are computed transposed (S^T = K . Q^T,
// v_mfma_f32_32x32x16_bf16) so a lane owns ONE query column: the online-softmax
// row statistics are lane-local (one cross-half exchange), and the exponentiated
// accumulator is fed straight back as the B operand of O^T = V^T . P^T with no
// LDS round trip.  V^T fragments come from ds_read_b64_tr_b16 (hardware
// transpose) of the row-major V tile.
#include "kernels.h"
#include <type_traits>

namespace revo {

// V^T fragments by inline asm.  Through the builtin, the compiler cannot tell these reads from the LDS-DMA that is
// filling the NEXT ring slots and puts s_waitcnt vmcnt(0) in front of the first of them in every iteration: the K/V
// tiles requested a few hundred cycles earlier are drained and their latency is paid once per key tile.  The asm reads
// are invisible to that bookkeeping; their own wait is att_v_wait (operands tied so that no consumer moves above it).
template <int OFF>
__device__ __forceinline__ uint64_t att_tr_read(uint32_t lds_addr) {
    uint64_t v;
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(lds_addr), "n"(OFF));
    return v;
}
// the 2 k-steps x DB d-blocks of one key block of a tile: v[s2][d][lo/hi]; OFF0 = ring slot + key block (bytes)
template <int ROWB, int DB, int OFF0>
__device__ __forceinline__ void att_v_issue(uint64_t (&v)[2][DB][2], const uint32_t (&va)[DB]) {
#define ATT_V1(S2, D)                                                          \
    v[S2][D][0] = att_tr_read<OFF0 + (16 * S2) * ROWB>(va[D]);                 \
    v[S2][D][1] = att_tr_read<OFF0 + (16 * S2 + 8) * ROWB>(va[D]);
    ATT_V1(0, 0) ATT_V1(0, 1)
    if constexpr (DB == 3) { ATT_V1(0, 2) }
    ATT_V1(1, 0) ATT_V1(1, 1)
    if constexpr (DB == 3) { ATT_V1(1, 2) }
#undef ATT_V1
}
// LEFT: LDS reads issued after v's that may stay in flight
template <int DB, int LEFT = 0>
__device__ __forceinline__ void att_v_wait(uint64_t (&v)[2][DB][2]) {
    if constexpr (DB == 2) {
        asm volatile("s_waitcnt lgkmcnt(%8)"
                     : "+v"(v[0][0][0]), "+v"(v[0][0][1]), "+v"(v[0][1][0]), "+v"(v[0][1][1]), "+v"(v[1][0][0]),
                       "+v"(v[1][0][1]), "+v"(v[1][1][0]), "+v"(v[1][1][1])
                     : "n"(LEFT));
    } else {
        asm volatile("s_waitcnt lgkmcnt(%12)"
                     : "+v"(v[0][0][0]), "+v"(v[0][0][1]), "+v"(v[0][1][0]), "+v"(v[0][1][1]), "+v"(v[0][2][0]),
                       "+v"(v[0][2][1]), "+v"(v[1][0][0]), "+v"(v[1][0][1]), "+v"(v[1][1][0]), "+v"(v[1][1][1]),
                       "+v"(v[1][2][0]), "+v"(v[1][2][1])
                     : "n"(LEFT));
    }
}
// K fragments the same way (ds_read_b128), four reads deep: a fragment's registers are refilled right behind the MFMA
// that consumed them and the MFMAs wait with counts.  Left to the compiler under the 128-VGPR budget, each read was
// issued into the same four registers right before the MFMA that consumes it: eight exposed LDS latencies per tile.
typedef uint32_t att_u32x4 __attribute__((ext_vector_type(4)));
template <int OFF>
__device__ __forceinline__ att_u32x4 att_read_b128(uint32_t lds_addr) {
    att_u32x4 v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(lds_addr), "n"(OFF));
    return v;
}
template <int LEFT>
__device__ __forceinline__ void att_k_wait(att_u32x4& k) {
    asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(k) : "n"(LEFT));
}
__device__ __forceinline__ bf16x8 att_v_frag(uint64_t lo, uint64_t hi) {
    typedef uint64_t u64x2 __attribute__((ext_vector_type(2)));
    const u64x2 c = {lo, hi};
    return __builtin_bit_cast(bf16x8, c);
}

// LDS-DMA helpers (plain device functions: called straight from the body of a __global__ template,
// these target builtins make hipcc 7.2 drop the kernel's host stub).
struct AttDmaSrc {
    __amdgpu_buffer_rsrc_t rsrc;
};
__device__ __forceinline__ void att_dma_init(AttDmaSrc& src, const void* base, int bytes) {
    src.rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, bytes, 0x00020000);
}
__device__ __forceinline__ void att_dma_issue(const AttDmaSrc& src, char* dst, uint32_t voff, uint32_t soff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(src.rsrc, (__attribute__((address_space(3))) void*)dst, 16, voff, soff, 0, 0);
}

// Body attention.  NW waves per workgroup, 32 query rows per wave.  Key rows [k_lo, S) are
// tiled; when k_lo == 1 the class-token key (row 0) is folded in as a rank-1 prelude
// (m = s_cls, l = 1, O = v_cls) so that L14's 576 patch keys are exactly 9 unmasked tiles.
// Query rows are then taken in rotated order (1, 2, ..., S-1, 0): the patch rows tile the
// waves exactly and the class-token query lands in a wave of the last workgroup that would
// otherwise idle (q_rot).
// (second launch bound: 4 waves per SIMD = two 8-wave workgroups per CU at head_dim 64, i.e. at most 128 VGPRs)
// DBG: knock-out study (compile time; instantiated only in librevo_exp.so and selected by REVO_ATTN_DBG; WRONG RESULTS,
// scripts/attn_knockout.sh): 1 = no K/V DMA in the loop and no wait for it, 2 = no per-tile barrier, 4 = no exponentials
// (scores packed as they are), 8 = no PV MFMAs, 16 = no score MFMAs, 32 = no row sums (the 16 v_pk_add_f32 per tile),
// 64 = no scale-and-shift before the exponentials (the 16 v_pk_fma_f32 per tile)
// STAG (head_dim 64, 8 waves; experiment, round 5): waves 4-7 run half a key tile behind waves 0-3 -- their P.V of tile t - 1
// moves behind the barrier of tile t, so that on every SIMD one wave's matrix segment lies beside its partner's softmax
// (the MI355X guide's stagger for same-program partners).  Needs a fourth ring slot (tile t - 1's V must outlive the
// barrier of tile t: 64 KiB per workgroup); same arithmetic in the same order per row: same bits.
template <int HD, int NW, int DBG = 0, bool STAG = false>
__global__ __launch_bounds__(NW * 64, (HD == 64 && NW >= 6) ? 4 : 1) void attn_fwd_kernel(const bf16_t* __restrict__ qkv, long ld,
                                                          bf16_t* __restrict__ out, long ldo, int S, int H, float c,
                                                          int q_rot, int k_lo
#ifdef REVO_EXPERIMENTS
                                                          , unsigned long long* clk     // diagnostic (attention_set_clock_buffer): [workgroup][2] shader-clock / 100 MHz ticks of its lifetime
#endif
                                                          ) {
    static_assert(HD == 64 || HD == 96, "body attention kernel is built for head_dim 64 (B16, L14) and 96 (G14)");
#ifdef REVO_EXPERIMENTS
    unsigned long long clk_t0 = 0, clk_r0 = 0;
    if (clk) { clk_t0 = __builtin_amdgcn_s_memtime(); clk_r0 = __builtin_amdgcn_s_memrealtime(); }
#endif
    constexpr int dbg = DBG;
    constexpr int KS = HD / 16;          // k-steps over d for S^T
    constexpr int DB = HD / 32;          // 32-wide d blocks of O^T
    constexpr int CH = HD / 8;           // 16-byte data chunks per K/V row (8 or 12)
    // LDS rows: 128 B for head_dim 64; 256 B (192 B used) for head_dim 96 so that the XOR swizzles below
    // stay inside a power-of-two chunk space.  Swizzle of the 16-byte chunk index, chosen per layout so
    // that the ds_read_b128 K reads and the ds_read_b64_tr_b16 V reads are bank-conflict free:
    //   128-B rows: K chunk ^= (key>>1)&7,  V chunk ^= ((key>>1)&1)<<2
    //   256-B rows: K chunk ^= key&15,      V chunk ^= (key&3)<<2
    constexpr int ROWB = HD == 64 ? 128 : 256;
    constexpr int TILE = 64 * ROWB;      // one operand of one key tile: 8 or 16 KB
    constexpr int BUF = 2 * TILE;        // K then V
    static_assert(!STAG || (HD == 64 && NW == 8), "");
    constexpr int LOOK = HD == 64 ? 2 : 1;  // key tiles requested ahead of the one being worked on
    constexpr int NBUF = LOOK + 1 + (STAG ? 1 : 0);   // ring depth: 48 KiB (hd 64; 64 KiB staggered) / 64 KiB (hd 96) per workgroup
    constexpr int NI = BUF / 1024;       // DMA wave-instructions per tile (1 KiB each)
    constexpr int NPW = (NI + NW - 1) / NW;   // per wave per tile; when NW does not divide NI the surplus
    constexpr int NDUMMY = NPW * NW - NI;     // instructions read out of bounds (zeros) into a spare KiB each
    constexpr int LPR = ROWB / 16;       // 16-byte chunks per LDS row
    constexpr int RPI = 64 / LPR;        // rows per DMA instruction
    __shared__ __attribute__((aligned(16))) char lds[NBUF * BUF + NDUMMY * 1024];

    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int r = lane & 31, hh = lane >> 5;
    // XCD-aware order.  Blocks L, L+8, L+16, ... share an XCD and start close in time: the query blocks of
    // one (image, head) are mapped to such a run, so its K/V rows are fetched into one XCD's L2 once instead
    // of once per query block on different XCDs (L14: 3 query blocks, 505 -> ~230 MB of fabric reads).
    int bh, qblk;
    {
        const int nq = gridDim.x, nbh = gridDim.y, L = blockIdx.y * nq + blockIdx.x;
        const int full = (nbh / 8) * 8;                 // (image, head) pairs that form whole groups of 8
        if (L < full * nq) {
            const int xcd = L & 7, slot = L >> 3;
            bh = (slot / nq) * 8 + xcd;
            qblk = slot % nq;
        } else {
            const int r = L - full * nq;                // the last < 8 pairs: plain order
            bh = full + r / nq;
            qblk = r % nq;
        }
    }
    const int b = bh / H, h = bh - b * H;
    const int W = H * HD;
    const long rowbase = (long)b * S;
    const int q0 = qblk * (NW * 32) + wave * 32;             // position in the (rotated) row order
    const bool wave_active = q0 < S;
    const bool row_valid = q0 + r < S;
    const int qpos = row_valid ? q0 + r : S - 1;
    const int qrow = q_rot ? (qpos + 1 < S ? qpos + 1 : 0) : qpos;   // sequence row served by this lane
    const int qrow_c = qrow;

    const bf16_t* kg = qkv + W + h * HD;
    const bf16_t* vg = qkv + 2 * W + h * HD;

    bf16x8 qf[KS];
    {
        const bf16_t* qp = qkv + (rowbase + qrow_c) * ld + h * HD + 8 * hh;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) qf[ks] = *(const bf16x8*)(qp + 16 * ks);
    }

    // ---- K/V tiles go global -> LDS by LDS-DMA (buffer_load ... lds): no staging registers, no
    // address arithmetic and no ds_write in the key loop.  Tile u lives in ring buffer u % NBUF as
    // [64 K rows | 64 V rows]; a DMA instruction fills 1 KiB (RPI rows), lane = 16-byte chunk, the
    // swizzle is applied on the source side.  Rows past the last key read as zeros through the
    // bounds-checked descriptor (one per image and head, starting at key k_lo's K row).
    const int nkeys = S - k_lo;
    const int nt = (nkeys + 63) / 64;
    AttDmaSrc dma_src;
    att_dma_init(dma_src, kg + (rowbase + k_lo) * ld, (int)(((long)(nkeys - 1) * ld + W + HD) * 2));
    uint32_t dma_voff[NPW];
#pragma unroll
    for (int i = 0; i < NPW; ++i) {
        const int j = wave + NW * i;                       // 1-KiB block of the (K | V) image
        const int row = j * RPI + lane / LPR;              // 0..127
        const int cp = lane % LPR;
        const int isv = row >= 64, key = row & 63;
        const int swz = HD == 64 ? (isv ? (((key >> 1) & 1) << 2) : ((key >> 1) & 7))
                                 : (isv ? ((key & 3) << 2) : (key & 15));
        const int ch = cp ^ swz;
        dma_voff[i] = (ch < CH && j < NI) ? (uint32_t)(((long)key * ld + (isv ? W : 0)) * 2 + ch * 16) : 0x80000000u;
    }
    const uint32_t tile_bytes = (uint32_t)(64 * ld * 2);
#define ATT_ISSUE_TILE(slot, t)                                                                 \
    do {                                                                                        \
        char* dst_ = lds + (slot) * BUF + wave * 1024;                                          \
        _Pragma("unroll") for (int i = 0; i < NPW; ++i) {                                       \
            char* d_ = (NDUMMY == 0 || wave + NW * i < NI) ? dst_ + i * NW * 1024               \
                                                           : lds + NBUF * BUF + (wave + NW * i - NI) * 1024; \
            att_dma_issue(dma_src, d_, dma_voff[i], (uint32_t)(t) * tile_bytes);                \
        }                                                                                       \
    } while (0)
    // wait until tile u has landed, given that tiles up to min(u + LOOK - 1, nt - 1) have been requested
#define ATT_WAIT_TILE(u)                                                                                  \
    do {                                                                                                  \
        const int inflight_ = ((u) + LOOK - 1 < nt ? LOOK - 1 : nt - 1 - (u));                             \
        if (inflight_ >= 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NPW) : "memory");                    \
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                             \
    } while (0)

    // ---- per-lane LDS fragment addresses (buffer 0), swizzles resolved once
    uint32_t kaddr[KS], vaddr[DB];
    {
        const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const char*)lds;
        const int sw = HD == 64 ? ((r >> 1) & 7) : (r & 15);      // same for key r and key r + 32
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) kaddr[ks] = lds0 + r * ROWB + (((2 * ks + hh) ^ sw) << 4);
        const int g16 = lane >> 4, li = lane & 15;
        const int tq = li >> 2, tp = li & 3;
        // V swizzle of keys 4*hh + tq (+ multiples of 8): moves whole 64-byte d-blocks
        const int sv = HD == 64 ? ((tq >> 1) & 1) : tq;
#pragma unroll
        for (int d = 0; d < DB; ++d)
            vaddr[d] = lds0 + TILE + (4 * hh + tq) * ROWB + ((((d ^ sv) * 4 + (g16 & 1) * 2 + (tp >> 1))) << 4) + (tp & 1) * 8;
    }

    f32x16 oacc[DB];
    float m_run = -INFINITY, l_run = 0.f;
#pragma unroll
    for (int d = 0; d < DB; ++d)
#pragma unroll
        for (int i = 0; i < 16; ++i) oacc[d][i] = 0.f;

    // ring discipline: tiles 0 .. LOOK-1 are requested here; the iteration that starts tile t requests tile t + LOOK
    // behind its barrier, i.e. once every wave is done with the tile whose buffer it takes (t - 1; staggered: t - 2,
    // whose P.V the late waves finished in the step of tile t - 1)
#pragma unroll
    for (int u = 0; u < LOOK; ++u)
        if (u < nt) ATT_ISSUE_TILE(u, u);

    if (k_lo == 1 && wave_active) {
        // rank-1 prelude with key row 0: this lane holds q[d] for d = 16*ks + 8*hh + j
        const bf16_t* k0p = kg + rowbase * ld + 8 * hh;
        float sdot = 0.f;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const s16x8 kk = *(const s16x8*)(k0p + 16 * ks);
            const s16x8 qq = __builtin_bit_cast(s16x8, qf[ks]);
#pragma unroll
            for (int j = 0; j < 8; ++j)
                sdot = fmaf(bf16_to_f32((bf16_t)qq[j]), bf16_to_f32((bf16_t)kk[j]), sdot);
        }
        sdot += __shfl_xor(sdot, 32, 64);
        m_run = sdot * c;
        l_run = hh == 0 ? 1.0f : 0.0f;       // the two halves' partial sums are added at the end
        const bf16_t* v0p = vg + rowbase * ld;
#pragma unroll
        for (int d = 0; d < DB; ++d)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const s16x4 vv = *(const s16x4*)(v0p + d * 32 + 8 * g + 4 * hh);
#pragma unroll
                for (int j = 0; j < 4; ++j) oacc[d][4 * g + j] = bf16_to_f32((bf16_t)vv[j]);
            }
    }

    // The query fragments were requested before the first K/V tiles.  Use them here once: the compiler then places its
    // wait for them in front of the loop.  Left pending, its bookkeeping carries them around the back edge and puts
    // vmcnt(3..0) in front of the score MFMAs of EVERY iteration, which drains the K/V tiles just requested.
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) asm volatile("" : "+v"(qf[ks]));
    const bool late = STAG && wave >= NW / 2;                 // wave-uniform
    uint32_t pw[2][8];                                        // the tile's probabilities, packed bf16 (late waves: kept across the barrier)
    uint64_t vt0[2][DB][2], vt1[2][DB][2];
    // O^T += V^T . P^T for the 64 keys of the tile whose ring slot starts at byte SBX; issued0: block 0's V^T fragments are
    // already requested (ahead of the exponentials)
    auto pv_tile = [&](auto sbx_c, const bool issued0) __attribute__((always_inline)) {
        constexpr int SBX = decltype(sbx_c)::value;
        if (!issued0) att_v_issue<ROWB, DB, SBX>(vt0, vaddr);
        att_v_wait<DB>(vt0);
        att_v_issue<ROWB, DB, SBX + 32 * ROWB>(vt1, vaddr);
#define ATT_PV(KB, VT)                                                                                  \
    _Pragma("unroll") for (int s2 = 0; s2 < 2; ++s2) {                                                  \
        uint4 pk;                                                                                       \
        pk.x = pw[KB][4 * s2 + 0];                                                                      \
        pk.y = pw[KB][4 * s2 + 1];                                                                      \
        pk.z = pw[KB][4 * s2 + 2];                                                                      \
        pk.w = pw[KB][4 * s2 + 3];                                                                      \
        const bf16x8 pb = __builtin_bit_cast(bf16x8, pk);                                               \
        /* k index (h, j) of this step is key 16*s2 + 8*(j>>2) + 4*h + (j&3) of the block */            \
        _Pragma("unroll") for (int d = 0; d < DB; ++d)                                                  \
            oacc[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(att_v_frag(VT[s2][d][0], VT[s2][d][1]), pb, oacc[d], 0, 0, 0); \
    }
        if constexpr (!(dbg & 8)) { ATT_PV(0, vt0); }
        att_v_wait<DB>(vt1);
        if constexpr (!(dbg & 8)) { ATT_PV(1, vt1); }
        else {      // knock-out build: keep the operands alive
            const uint32_t keep_ = pw[0][0] ^ pw[1][7] ^ (uint32_t)vt0[0][0][0] ^ (uint32_t)vt1[1][DB - 1][1];
            asm volatile("" :: "v"(keep_));
        }
#undef ATT_PV
    };
    // One key tile.  The ring slot is a compile-time constant (the loop below is unrolled NBUF times), so every LDS
    // address is a per-lane register fixed at kernel entry plus an immediate.
    auto tile_step = [&](auto slot_c, const int t) __attribute__((always_inline)) {
        constexpr int SLOT = decltype(slot_c)::value;
        constexpr int SB = SLOT * BUF;
        if constexpr (!(dbg & 1)) ATT_WAIT_TILE(t);
        if constexpr (!(dbg & 2)) __builtin_amdgcn_s_barrier();           // tile t is in LDS for every wave; every wave is done with tile t-1
        if constexpr (!(dbg & 1)) if (t + LOOK < nt) ATT_ISSUE_TILE((SLOT + LOOK) % NBUF, t + LOOK);
        if (!wave_active) return;
        if constexpr (STAG) {
            // late waves: the previous tile's P.V first (its probabilities were kept in pw, its V tile is still in the ring)
            if (late && t > 0) pv_tile(std::integral_constant<int, ((SLOT + NBUF - 1) % NBUF) * BUF>{}, false);
        }
        f32x16 sacc[2];
        att_u32x4 kf[4];
        // S^T = K . Q^T for the 64 keys of the tile, masked past the last key
        auto scores = [&]() __attribute__((always_inline)) {
            constexpr int NR = 2 * KS;          // fragment i: key block i / KS, k-step i % KS
#define ATT_KOFF(i) (SB + ((i) / KS) * 32 * ROWB)
            kf[0] = att_read_b128<ATT_KOFF(0)>(kaddr[0]);
            kf[1] = att_read_b128<ATT_KOFF(1)>(kaddr[1]);
            kf[2] = att_read_b128<ATT_KOFF(2)>(kaddr[2]);
            kf[3] = att_read_b128<ATT_KOFF(3)>(kaddr[3]);
#pragma unroll
            for (int kblk = 0; kblk < 2; ++kblk)
#pragma unroll
                for (int i = 0; i < 16; ++i) sacc[kblk][i] = 0.f;
#define ATT_KSTEP(i)                                                                                              \
    if constexpr ((i) < NR) {                                                                                     \
        if constexpr ((i) + 4 <= NR) att_k_wait<3>(kf[(i) % 4]);                                                  \
        else att_k_wait<NR - 1 - (i)>(kf[(i) % 4]);                                                               \
        sacc[(i) / KS] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, kf[(i) % 4]), qf[(i) % KS], \
                                                                 sacc[(i) / KS], 0, 0, 0);                          \
        if constexpr ((i) + 4 < NR) kf[(i) % 4] = att_read_b128<ATT_KOFF((i) + 4)>(kaddr[((i) + 4) % KS]);        \
    }
            ATT_KSTEP(0) ATT_KSTEP(1) ATT_KSTEP(2) ATT_KSTEP(3) ATT_KSTEP(4) ATT_KSTEP(5)
            ATT_KSTEP(6) ATT_KSTEP(7) ATT_KSTEP(8) ATT_KSTEP(9) ATT_KSTEP(10) ATT_KSTEP(11)
#undef ATT_KSTEP
#undef ATT_KOFF
            if (t == nt - 1 && (nkeys & 63)) {
                asm volatile("" ::: "memory"); /* a real branch: if-converted, this is 32 selects per key tile */
#pragma unroll
                for (int kblk = 0; kblk < 2; ++kblk)
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        const int key_ = t * 64 + kblk * 32 + (i & 3) + 8 * (i >> 2) + 4 * hh;
                        if (key_ >= nkeys) sacc[kblk][i] = -INFINITY;
                    }
            }
        };
        if constexpr (!(dbg & 16)) scores();
        else { _Pragma("unroll") for (int kb_ = 0; kb_ < 2; ++kb_) _Pragma("unroll") for (int i_ = 0; i_ < 16; ++i_) sacc[kb_][i_] = (float)(lane + i_) * 1e-3f; }
        // V^T fragments of key block 0 are requested ahead of the exponentials, block 1's before block 0's MFMAs
        if (!(STAG && late)) att_v_issue<ROWB, DB, SB>(vt0, vaddr);
        // Optimistic softmax.  p = 2^(s c - m) is computed against the running reference m on
        // register pairs (v_pk_fma_f32 / v_pk_add_f32 / v_cvt_pk_bf16_f32) WITHOUT first taking the
        // tile maximum: the softmax VALU work, not the MFMAs, bounds this kernel at head_dim 64.
        // fp32 and bf16 share the exponent range, so a reference that lags the true maximum by up
        // to 2^80 loses no precision; only when a row sum leaves that range (or m is still -inf)
        // does the wave take the maximum, move the reference, rescale and redo the exponentials.
        f32x2 ps2;
        auto exps = [&]() __attribute__((always_inline)) {
            const f32x2 c2_ = {c, c}, m2_ = {m_run, m_run};
            ps2 = (f32x2){0.f, 0.f};
#pragma unroll
            for (int kblk = 0; kblk < 2; ++kblk)
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const f32x2 sv_ = {sacc[kblk][2 * j], sacc[kblk][2 * j + 1]};
                    const f32x2 e_ = (dbg & 64) ? sv_ : sv_ * c2_ - m2_;
                    const f32x2 pv_ = {__builtin_amdgcn_exp2f(e_.x), __builtin_amdgcn_exp2f(e_.y)};
                    if constexpr (!(dbg & 32)) ps2 += pv_;
                    pw[kblk][j] = pack_bf16x2(pv_.x, pv_.y);
                }
        };
        if constexpr (!(dbg & 4)) exps();
        else {
            ps2 = (f32x2){1.f, 1.f};
            _Pragma("unroll") for (int kb_ = 0; kb_ < 2; ++kb_)
                _Pragma("unroll") for (int j_ = 0; j_ < 8; ++j_) pw[kb_][j_] = pack_bf16x2(sacc[kb_][2 * j_], sacc[kb_][2 * j_ + 1]);
        }
        float ps = ps2.x + ps2.y;
        if (!__all(ps < 0x1p80f)) {
            asm volatile("" ::: "memory");
            scores();           // rare path: the scores were consumed in place, compute them again
            float mx = -INFINITY;
#pragma unroll
            for (int kblk = 0; kblk < 2; ++kblk)
#pragma unroll
                for (int i = 0; i < 16; ++i) mx = fmaxf(mx, sacc[kblk][i]);
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64)) * c;
            const float m_new = fmaxf(m_run, mx);
            const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);      // 0 when m_run was -inf
            m_run = m_new;
            l_run *= alpha;
#pragma unroll
            for (int d = 0; d < DB; ++d)
#pragma unroll
                for (int i = 0; i < 16; ++i) oacc[d][i] *= alpha;
            exps();
            ps = ps2.x + ps2.y;
        }
        l_run += ps;
        if (!(STAG && late)) pv_tile(std::integral_constant<int, SB>{}, true);
    };
#ifdef REVO_ATTN_PRIO
    // experiment (MI355X guide, "static priority for the younger half"): waves 4-7 are the arbitration losers of every segment
    if (REVO_ATTN_PRIO == 1 ? wave >= NW / 2 : wave < NW / 2) __builtin_amdgcn_s_setprio(1);
#endif
    for (int t = 0; t < nt; t += NBUF) {
        tile_step(std::integral_constant<int, 0>{}, t);
        if (NBUF > 1 && t + 1 < nt) tile_step(std::integral_constant<int, 1 % NBUF>{}, t + 1);
        if (NBUF > 2 && t + 2 < nt) tile_step(std::integral_constant<int, 2 % NBUF>{}, t + 2);
        if (NBUF > 3 && t + 3 < nt) tile_step(std::integral_constant<int, 3 % NBUF>{}, t + 3);
    }
    if constexpr (STAG) {
        // the late waves' last P.V (its V tile stays in the ring: nothing is requested behind the last tile)
        if (late && wave_active && nt > 0) {
            switch ((nt - 1) % NBUF) {
                case 0: pv_tile(std::integral_constant<int, 0>{}, false); break;
                case 1: pv_tile(std::integral_constant<int, BUF>{}, false); break;
                case 2: pv_tile(std::integral_constant<int, 2 * BUF>{}, false); break;
                default: pv_tile(std::integral_constant<int, 3 * BUF>{}, false); break;
            }
        }
    }

    if constexpr (HD == 64) {
        // Output through a wave-private LDS slab (the K/V ring is free now) so that global stores are whole
        // 128-byte rows, 8 rows per instruction; stored straight from the accumulator layout a store
        // instruction touches 32 rows with 16 bytes each.
        __builtin_amdgcn_s_barrier();                // every wave is out of the ring
        if (wave_active) {
            constexpr int RS = 144;                  // 64 bf16 + 16 bytes of padding per row
            char* slab = lds + wave * (32 * RS);
            const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
            const float inv = 1.0f / l_tot;
#pragma unroll
            for (int d = 0; d < DB; ++d)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    uint2 o;
                    o.x = pack_bf16x2(oacc[d][4 * g + 0] * inv, oacc[d][4 * g + 1] * inv);
                    o.y = pack_bf16x2(oacc[d][4 * g + 2] * inv, oacc[d][4 * g + 3] * inv);
                    *(uint2*)(slab + r * RS + (d * 32 + 8 * g + 4 * hh) * 2) = o;
                }
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const int rl = it * 8 + (lane >> 3);
                const uint4 v = *(const uint4*)(slab + rl * RS + (lane & 7) * 16);
                const int qp = q0 + rl;
                if (qp < S) {
                    const int qr = q_rot ? (qp + 1 < S ? qp + 1 : 0) : qp;
                    *(uint4*)(out + (rowbase + qr) * ldo + h * HD + (lane & 7) * 8) = v;
                }
            }
        }
    } else if (wave_active) {
        const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
        if (row_valid) {
            const float inv = 1.0f / l_tot;
            bf16_t* op = out + (rowbase + qrow) * ldo + h * HD;
#pragma unroll
            for (int d = 0; d < DB; ++d)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    uint2 o;
                    o.x = pack_bf16x2(oacc[d][4 * g + 0] * inv, oacc[d][4 * g + 1] * inv);
                    o.y = pack_bf16x2(oacc[d][4 * g + 2] * inv, oacc[d][4 * g + 3] * inv);
                    *(uint2*)(op + d * 32 + 8 * g + 4 * hh) = o;
                }
        }
    }
#undef ATT_ISSUE_TILE
#undef ATT_WAIT_TILE
#ifdef REVO_EXPERIMENTS
    if (clk && threadIdx.x == 0) {
        const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
        unsigned long long* dst = clk + 2 * ((size_t)blockIdx.y * gridDim.x + blockIdx.x);
        dst[0] = t1 - clk_t0;
        dst[1] = r1 - clk_r0;
    }
#endif
}



// ---- The same body attention on v_mfma_f32_16x16x32_bf16 (head_dim 64; round 6, the MFMA-shape experiment the CDNA guide
// asks for: "the chip can hold a higher clock on one shape than on the other -- build both at the same output tile per wave,
// keep the faster by wall").  Same workgroup, same K/V ring and DMA, same 32 query rows and 64-key tiles per wave, same
// optimistic softmax; what changes is who holds what:
//   S^T block = 16 keys x 16 queries: lane (li = lane & 15, lq = lane >> 4) holds query li of query block nb (two per wave)
//     and keys 16 mb + 4 lq + i, i = 0..3, of key block mb (four per tile): a query's 64 scores of a tile sit in FOUR lanes
//     (lq = 0..3), 16 each.  Row sums stay per-lane partials (added over the four lanes once, at the end); the reference m
//     is per query block.
//   P^T as the B operand of O^T = V^T . P^T needs, per lane, 8 key slots k = 8 lq + j of a 32-key step: slot j of step kk :=
//     key 16 (2 kk + (j >> 2)) + 4 lq + (j & 3) -- exactly the keys this lane's accumulators of blocks 2 kk and 2 kk + 1 hold:
//     the packed probabilities go straight back in, no shuffle.  V^T fragments follow the same slot order: two
//     ds_read_b64_tr_b16 per (step, 16-wide d block), each transposing 4 keys x 16 d.
//   V rows are swizzled for that read (chunk ^= ((key >> 1) & 3) << 1: the eight rows a 32-lane LDS cycle touches fall into
//     four different 32-byte windows per row parity); K rows keep (key >> 1) & 7 (conflict-free for this read order too).
// acc += A . B with the accumulator IN PLACE (vDst == SrcC), as inline asm.  Through the builtin hipcc 7.2 picks registers for
// v_mfma_f32_16x16x32_bf16 that the hardware does not honour under load: (a) with a literal 0 as SrcC it gave the first MFMA
// of a chain the registers of its own A operand as destination (v_mfma v[50:53], v[50:53], v[10:13], 0 -- the 16x16 forms
// carry no early-clobber in LLVM), (b) it continued a chain into a DIFFERENT destination (v_mfma v[94:97], .., v[66:69]) with
// one s_nop between producer and consumer.  Either way a few rows per launch came out different from run to run (up to 1 M
// of 38 M output elements with constant P and V; profiles/r06_attention_mfma_shape.json).  In place, back to back, is the
// form the body GEMM runs billions of times.  The asm hides the MFMA from the hazard recogniser: att16_mfma_settle() puts
// the wait states between the last MFMA and the first vector instruction that reads an accumulator.
__device__ __forceinline__ void att16_mfma(f32x4& acc, const bf16x8& a, const bf16x8& b) {
    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
}
__device__ __forceinline__ void att16_mfma_settle() { asm volatile("s_nop 15\n\ts_nop 15" ::: "memory"); }

template <int NW, int DBG = 0>
__global__ __launch_bounds__(NW * 64, NW >= 6 ? 4 : 1) void attn16_fwd_kernel(const bf16_t* __restrict__ qkv, long ld,
                                                          bf16_t* __restrict__ out, long ldo, int S, int H, float c,
                                                          int q_rot, int k_lo
#ifdef REVO_EXPERIMENTS
                                                          , unsigned long long* clk
#endif
                                                          ) {
#ifdef REVO_EXPERIMENTS
    unsigned long long clk_t0 = 0, clk_r0 = 0;
    if (clk) { clk_t0 = __builtin_amdgcn_s_memtime(); clk_r0 = __builtin_amdgcn_s_memrealtime(); }
#endif
    constexpr int HD = 64, ROWB = 128, TILE = 64 * ROWB, BUF = 2 * TILE, LOOK = 2, NBUF = LOOK + 1;
    constexpr int NI = BUF / 1024, NPW = (NI + NW - 1) / NW, NDUMMY = NPW * NW - NI, LPR = ROWB / 16, RPI = 64 / LPR;
    __shared__ __attribute__((aligned(16))) char lds[NBUF * BUF + NDUMMY * 1024];
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int li = lane & 15, lq = lane >> 4;
    int bh, qblk;
    {
        const int nq = gridDim.x, nbh = gridDim.y, L = blockIdx.y * nq + blockIdx.x;
        const int full = (nbh / 8) * 8;
        if (L < full * nq) {
            const int xcd = L & 7, slot = L >> 3;
            bh = (slot / nq) * 8 + xcd;
            qblk = slot % nq;
        } else {
            const int rr = L - full * nq;
            bh = full + rr / nq;
            qblk = rr % nq;
        }
    }
    const int b = bh / H, h = bh - b * H;
    const int W = H * HD;
    const long rowbase = (long)b * S;
    const int q0 = qblk * (NW * 32) + wave * 32;
    const bool wave_active = q0 < S;
    const bf16_t* kg = qkv + W + h * HD;
    const bf16_t* vg = qkv + 2 * W + h * HD;

    bf16x8 qf[2][2];                          // [query block][k-step over d]: query 16 nb + li, d = 32 ks + 8 lq + j
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) {
        const int qp = q0 + nb * 16 + li < S ? q0 + nb * 16 + li : S - 1;
        const int qrow = q_rot ? (qp + 1 < S ? qp + 1 : 0) : qp;
        const bf16_t* qptr = qkv + (rowbase + qrow) * ld + h * HD + 8 * lq;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) qf[nb][ks] = *(const bf16x8*)(qptr + 32 * ks);
    }

    const int nkeys = S - k_lo;
    const int nt = (nkeys + 63) / 64;
    AttDmaSrc dma_src;
    att_dma_init(dma_src, kg + (rowbase + k_lo) * ld, (int)(((long)(nkeys - 1) * ld + W + HD) * 2));
    uint32_t dma_voff[NPW];
#pragma unroll
    for (int i = 0; i < NPW; ++i) {
        const int j = wave + NW * i;
        const int row = j * RPI + lane / LPR;
        const int cp = lane % LPR;
        const int isv = row >= 64, key = row & 63;
        const int swz = isv ? (((key >> 1) & 3) << 1) : ((key >> 1) & 7);
        const int ch = cp ^ swz;
        dma_voff[i] = (j < NI) ? (uint32_t)(((long)key * ld + (isv ? W : 0)) * 2 + ch * 16) : 0x80000000u;
    }
    const uint32_t tile_bytes = (uint32_t)(64 * ld * 2);
#define ATT_ISSUE_TILE(slot, t)                                                                 \
    do {                                                                                        \
        char* dst_ = lds + (slot) * BUF + wave * 1024;                                          \
        _Pragma("unroll") for (int i = 0; i < NPW; ++i) {                                       \
            char* d_ = (NDUMMY == 0 || wave + NW * i < NI) ? dst_ + i * NW * 1024               \
                                                           : lds + NBUF * BUF + (wave + NW * i - NI) * 1024; \
            att_dma_issue(dma_src, d_, dma_voff[i], (uint32_t)(t) * tile_bytes);                \
        }                                                                                       \
    } while (0)
#define ATT_WAIT_TILE(u)                                                                                  \
    do {                                                                                                  \
        const int inflight_ = ((u) + LOOK - 1 < nt ? LOOK - 1 : nt - 1 - (u));                             \
        if (inflight_ >= 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NPW) : "memory");                    \
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                             \
    } while (0)

    uint32_t kaddr[2], vaddr[4];
    {
        const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const char*)lds;
        const int sw = (li >> 1) & 7;                         // the same for keys li, li + 16, ...
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) kaddr[ks] = lds0 + li * ROWB + (((4 * ks + lq) ^ sw) << 4);
        const int tq = li >> 2, tp = li & 3;
        const int sv = ((2 * lq + (tq >> 1)) & 3) << 1;       // swizzle of keys 4 lq + tq (+ multiples of 16)
#pragma unroll
        for (int d = 0; d < 4; ++d)
            vaddr[d] = lds0 + TILE + (4 * lq + tq) * ROWB + (((2 * d + (tp >> 1)) ^ sv) << 4) + (tp & 1) * 8;
    }

    f32x4 oacc[4][2];                         // [d block][query block]: d = 16 db + 4 lq + i
    float m_run[2] = {-INFINITY, -INFINITY}, l_run[2] = {0.f, 0.f};
#pragma unroll
    for (int d = 0; d < 4; ++d)
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) oacc[d][nb] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int u = 0; u < LOOK; ++u)
        if (u < nt) ATT_ISSUE_TILE(u, u);

    if (k_lo == 1 && wave_active) {
        // rank-1 prelude with key row 0: this lane holds q[d] for d = 32 ks + 8 lq + j; a query's dot product is spread over its four lanes
        const bf16_t* k0p = kg + rowbase * ld + 8 * lq;
        float sdot[2] = {0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const s16x8 kk = *(const s16x8*)(k0p + 32 * ks);
#pragma unroll
            for (int nb = 0; nb < 2; ++nb) {
                const s16x8 qq = __builtin_bit_cast(s16x8, qf[nb][ks]);
#pragma unroll
                for (int j = 0; j < 8; ++j) sdot[nb] = fmaf(bf16_to_f32((bf16_t)qq[j]), bf16_to_f32((bf16_t)kk[j]), sdot[nb]);
            }
        }
        const bf16_t* v0p = vg + rowbase * ld;
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) {
            sdot[nb] += __shfl_xor(sdot[nb], 16, 64);
            sdot[nb] += __shfl_xor(sdot[nb], 32, 64);
            m_run[nb] = sdot[nb] * c;
            l_run[nb] = lq == 0 ? 1.0f : 0.0f;               // the four lanes' partial sums are added at the end
        }
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            const s16x4 vv = *(const s16x4*)(v0p + d * 16 + 4 * lq);
#pragma unroll
            for (int nb = 0; nb < 2; ++nb)
#pragma unroll
                for (int j = 0; j < 4; ++j) oacc[d][nb][j] = bf16_to_f32((bf16_t)vv[j]);
        }
    }
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) asm volatile("" : "+v"(qf[nb][ks]));

    uint32_t pw[4][2][2];                     // [key block][query block][pair]: packed probabilities
    uint64_t vt0[4][2], vt1[4][2];            // V^T fragments of one 32-key step: [d block][first / second 16 keys]
    auto v_issue = [&](auto off_c, uint64_t (&v)[4][2]) __attribute__((always_inline)) {
        constexpr int OFF = decltype(off_c)::value;       // ring slot + 32-key step (bytes)
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            v[d][0] = att_tr_read<OFF>(vaddr[d]);
            v[d][1] = att_tr_read<OFF + 16 * ROWB>(vaddr[d]);
        }
    };
    auto v_wait = [&](uint64_t (&v)[4][2]) __attribute__((always_inline)) {
        asm volatile("s_waitcnt lgkmcnt(0)"
                     : "+v"(v[0][0]), "+v"(v[0][1]), "+v"(v[1][0]), "+v"(v[1][1]), "+v"(v[2][0]), "+v"(v[2][1]), "+v"(v[3][0]), "+v"(v[3][1]));
    };
    auto pv_step = [&](auto kk_c, uint64_t (&v)[4][2]) __attribute__((always_inline)) {
        constexpr int KK = decltype(kk_c)::value;
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) {
            uint4 pk;
            pk.x = pw[2 * KK][nb][0];
            pk.y = pw[2 * KK][nb][1];
            pk.z = pw[2 * KK + 1][nb][0];
            pk.w = pw[2 * KK + 1][nb][1];
            bf16x8 pb = __builtin_bit_cast(bf16x8, pk);
            asm volatile("s_nop 1" : "+v"(pb));      // (the packs that produce pb are vector-unit writes: see scores())
#pragma unroll
            for (int d = 0; d < 4; ++d)
                att16_mfma(oacc[d][nb], (DBG & 2) ? att_v_frag(0x3f803f803f803f80ull, 0x3f803f803f803f80ull) : att_v_frag(v[d][0], v[d][1]), pb);
        }
    };
    auto tile_step = [&](auto slot_c, const int t) __attribute__((always_inline)) {
        constexpr int SLOT = decltype(slot_c)::value;
        constexpr int SB = SLOT * BUF;
        ATT_WAIT_TILE(t);
        __builtin_amdgcn_s_barrier();
        if (t + LOOK < nt) ATT_ISSUE_TILE((SLOT + LOOK) % NBUF, t + LOOK);
        if (!wave_active) return;
        f32x4 sacc[4][2];
        att_u32x4 kf[4];
        auto scores = [&]() __attribute__((always_inline)) {
            // fragment i: key block i / 2, k-step i % 2; each feeds two MFMAs (the two query blocks)
#define ATT_KOFF(i) (SB + ((i) / 2) * 16 * ROWB)
            kf[0] = att_read_b128<ATT_KOFF(0)>(kaddr[0]);
            kf[1] = att_read_b128<ATT_KOFF(1)>(kaddr[1]);
            kf[2] = att_read_b128<ATT_KOFF(2)>(kaddr[0]);
            kf[3] = att_read_b128<ATT_KOFF(3)>(kaddr[1]);
#pragma unroll
            for (int mb = 0; mb < 4; ++mb)
#pragma unroll
                for (int nb = 0; nb < 2; ++nb) sacc[mb][nb] = (f32x4){0.f, 0.f, 0.f, 0.f};
            // (the zeros are vector-unit writes and the asm MFMAs that read them are invisible to the hazard recogniser: two
            //  wait states in between, once per tile)
            asm volatile("s_nop 1" : "+v"(sacc[0][0]), "+v"(sacc[0][1]), "+v"(sacc[1][0]), "+v"(sacc[1][1]), "+v"(sacc[2][0]),
                                     "+v"(sacc[2][1]), "+v"(sacc[3][0]), "+v"(sacc[3][1]));
#define ATT_KSTEP(i)                                                                                              \
    {                                                                                                             \
        if constexpr ((i) + 4 <= 8) att_k_wait<3>(kf[(i) % 4]);                                                   \
        else att_k_wait<8 - 1 - (i)>(kf[(i) % 4]);                                                                \
        _Pragma("unroll") for (int nb = 0; nb < 2; ++nb)                                                          \
            att16_mfma(sacc[(i) / 2][nb], __builtin_bit_cast(bf16x8, kf[(i) % 4]), qf[nb][(i) % 2]);               \
        if constexpr ((i) + 4 < 8) kf[(i) % 4] = att_read_b128<ATT_KOFF((i) + 4)>(kaddr[(i) % 2]);                \
    }
            ATT_KSTEP(0) ATT_KSTEP(1) ATT_KSTEP(2) ATT_KSTEP(3) ATT_KSTEP(4) ATT_KSTEP(5) ATT_KSTEP(6) ATT_KSTEP(7)
#undef ATT_KSTEP
#undef ATT_KOFF
            att16_mfma_settle();
            if (t == nt - 1 && (nkeys & 63)) {
                asm volatile("" ::: "memory");
#pragma unroll
                for (int mb = 0; mb < 4; ++mb)
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int key_ = t * 64 + mb * 16 + 4 * lq + i;
                        if (key_ >= nkeys) { sacc[mb][0][i] = -INFINITY; sacc[mb][1][i] = -INFINITY; }
                    }
            }
        };
        scores();
        v_issue(std::integral_constant<int, SB>{}, vt0);            // keys 0..31 of the tile, ahead of the exponentials
        f32x2 ps2[2];
        auto exps = [&]() __attribute__((always_inline)) {
            const f32x2 c2_ = {c, c};
#pragma unroll
            for (int nb = 0; nb < 2; ++nb) {
                const f32x2 m2_ = {m_run[nb], m_run[nb]};
                ps2[nb] = (f32x2){0.f, 0.f};
#pragma unroll
                for (int mb = 0; mb < 4; ++mb)
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        const f32x2 sv_ = {sacc[mb][nb][2 * j], sacc[mb][nb][2 * j + 1]};
                        const f32x2 e_ = sv_ * c2_ - m2_;
                        const f32x2 pv_ = {__builtin_amdgcn_exp2f(e_.x), __builtin_amdgcn_exp2f(e_.y)};
                        ps2[nb] += pv_;
                        pw[mb][nb][j] = (DBG & 1) ? 0x3f803f80u : pack_bf16x2(pv_.x, pv_.y);
                    }
            }
        };
        exps();
        float ps[2] = {ps2[0].x + ps2[0].y, ps2[1].x + ps2[1].y};
        if (!__all(ps[0] < 0x1p80f && ps[1] < 0x1p80f)) {
            asm volatile("" ::: "memory");
            scores();           // rare path: the scores were consumed in place, compute them again
#pragma unroll
            for (int nb = 0; nb < 2; ++nb) {
                float mx = -INFINITY;
#pragma unroll
                for (int mb = 0; mb < 4; ++mb)
#pragma unroll
                    for (int i = 0; i < 4; ++i) mx = fmaxf(mx, sacc[mb][nb][i]);
                mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
                mx = fmaxf(mx, __shfl_xor(mx, 32, 64)) * c;
                const float m_new = fmaxf(m_run[nb], mx);
                const float alpha = __builtin_amdgcn_exp2f(m_run[nb] - m_new);      // 0 when m_run was -inf
                m_run[nb] = m_new;
                l_run[nb] *= alpha;
#pragma unroll
                for (int d = 0; d < 4; ++d) oacc[d][nb] *= alpha;       // (the previous tile's P.V MFMAs settled long ago: a whole scores() lies in between)
            }
            exps();
            ps[0] = ps2[0].x + ps2[0].y;
            ps[1] = ps2[1].x + ps2[1].y;
        }
        l_run[0] += ps[0];
        l_run[1] += ps[1];
        v_wait(vt0);
        v_issue(std::integral_constant<int, SB + 32 * ROWB>{}, vt1);
        pv_step(std::integral_constant<int, 0>{}, vt0);
        v_wait(vt1);
        pv_step(std::integral_constant<int, 1>{}, vt1);
    };
    for (int t = 0; t < nt; t += NBUF) {
        tile_step(std::integral_constant<int, 0>{}, t);
        if (t + 1 < nt) tile_step(std::integral_constant<int, 1>{}, t + 1);
        if (t + 2 < nt) tile_step(std::integral_constant<int, 2>{}, t + 2);
    }

    att16_mfma_settle();
    __builtin_amdgcn_s_barrier();                // every wave is out of the ring
    if (wave_active) {
        constexpr int RS = 144;
        char* slab = lds + wave * (32 * RS);
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) {
            float l_tot = l_run[nb] + __shfl_xor(l_run[nb], 16, 64);
            l_tot += __shfl_xor(l_tot, 32, 64);
            const float inv = 1.0f / l_tot;
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                uint2 o;
                o.x = pack_bf16x2(oacc[d][nb][0] * inv, oacc[d][nb][1] * inv);
                o.y = pack_bf16x2(oacc[d][nb][2] * inv, oacc[d][nb][3] * inv);
                *(uint2*)(slab + (nb * 16 + li) * RS + (d * 16 + 4 * lq) * 2) = o;
            }
        }
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int rl = it * 8 + (lane >> 3);
            const uint4 v = *(const uint4*)(slab + rl * RS + (lane & 7) * 16);
            const int qp = q0 + rl;
            if (qp < S) {
                const int qr = q_rot ? (qp + 1 < S ? qp + 1 : 0) : qp;
                *(uint4*)(out + (rowbase + qr) * ldo + h * HD + (lane & 7) * 8) = v;
            }
        }
    }
#undef ATT_ISSUE_TILE
#undef ATT_WAIT_TILE
#ifdef REVO_EXPERIMENTS
    if (clk && threadIdx.x == 0) {
        const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
        unsigned long long* dst = clk + 2 * ((size_t)blockIdx.y * gridDim.x + blockIdx.x);
        dst[0] = t1 - clk_t0;
        dst[1] = r1 - clk_r0;
    }
#endif
}

#ifdef REVO_EXPERIMENTS
static unsigned long long* g_attn_clk = nullptr;
void attention_set_clock_buffer(unsigned long long* buf) { g_attn_clk = buf; }
#define ATT_CLK_ARG , g_attn_clk
#else
#define ATT_CLK_ARG
#endif
static int g_attn_shape16 = 0;     // 1 = head_dim 64 on attn16_fwd_kernel (v_mfma_f32_16x16x32_bf16)
void attention_set_shape16(int on) { g_attn_shape16 = on; }
template <int HD, int NW>
static void launch_attn_nw(const bf16_t* qkv, long ld, bf16_t* out, long ldo, int B, int S, int H, float c, int q_rot,
                           int k_lo, hipStream_t st) {
    const int rows = S;
    dim3 grid((rows + NW * 32 - 1) / (NW * 32), B * H), block(NW * 64);
#ifdef REVO_EXPERIMENTS
    if constexpr (HD == 64 && NW == 8) {
        if (getenv("REVO_ATTN_STAG")) {            // A/B of the half-tile stagger (scripts/experiments/r5_attn_stagger.sh)
            hipLaunchKernelGGL((attn_fwd_kernel<HD, NW, 0, true>), grid, block, 0, st, qkv, ld, out, ldo, S, H, c, q_rot, k_lo ATT_CLK_ARG);
            return;
        }
        if (const char* e = getenv("REVO_ATTN_DBG")) {
            switch (atoi(e)) {
#define ATT_DBG_CASE(D) case D: hipLaunchKernelGGL((attn_fwd_kernel<HD, NW, D>), grid, block, 0, st, qkv, ld, out, ldo, S, H, c, q_rot, k_lo ATT_CLK_ARG); return;
                ATT_DBG_CASE(1) ATT_DBG_CASE(2) ATT_DBG_CASE(3) ATT_DBG_CASE(4) ATT_DBG_CASE(7) ATT_DBG_CASE(8) ATT_DBG_CASE(16)
                ATT_DBG_CASE(24) ATT_DBG_CASE(28) ATT_DBG_CASE(31) ATT_DBG_CASE(32) ATT_DBG_CASE(64) ATT_DBG_CASE(96)
#undef ATT_DBG_CASE
                default: break;
            }
        }
    }
#endif
#ifdef REVO_EXPERIMENTS           // (the 16x16x32 kernel is an experiment: measured and not adopted, profiles/r06_attention_mfma_shape.json)
    if constexpr (HD == 64) {
        if (g_attn_shape16) {
            if constexpr (NW == 8) {
                if (const char* e = getenv("REVO_ATTN16_DBG")) {
                    switch (atoi(e)) {
                        case 1: hipLaunchKernelGGL((attn16_fwd_kernel<NW, 1>), grid, block, 0, st, qkv, ld, out, ldo, S, H, c, q_rot, k_lo ATT_CLK_ARG); return;
                        case 2: hipLaunchKernelGGL((attn16_fwd_kernel<NW, 2>), grid, block, 0, st, qkv, ld, out, ldo, S, H, c, q_rot, k_lo ATT_CLK_ARG); return;
                        case 3: hipLaunchKernelGGL((attn16_fwd_kernel<NW, 3>), grid, block, 0, st, qkv, ld, out, ldo, S, H, c, q_rot, k_lo ATT_CLK_ARG); return;
                        default: break;
                    }
                }
            }
            hipLaunchKernelGGL((attn16_fwd_kernel<NW>), grid, block, 0, st, qkv, ld, out, ldo, S, H, c, q_rot, k_lo ATT_CLK_ARG);
            return;
        }
    }
#endif
    hipLaunchKernelGGL((attn_fwd_kernel<HD, NW>), grid, block, 0, st, qkv, ld, out, ldo, S, H, c, q_rot, k_lo ATT_CLK_ARG);
}

static int g_attn_force_nw = 0;   // timing experiments only
void attention_force_nw(int nw) { g_attn_force_nw = nw; }
// has_cls: row 0 of every sequence is the class token (true for every PE-Core variant with use_cls)
int launch_attention_ex(const bf16_t* qkv, long ld, bf16_t* out, long ldo, int B, int S, int H, int hd, int has_cls,
                        hipStream_t st) {
    REVO_REQUIRE(hd == 64 || hd == 96, "attention: head_dim must be 64 (PE-Core B16 / L14) or 96 (G14)");
    REVO_REQUIRE(ld % 8 == 0 && ldo % 4 == 0, "attention: strides must keep 16-byte alignment");
    if (B <= 0 || S <= 0) return 0;
    const float scale = 1.0f / sqrtf((float)hd);
    const float c = scale * 1.44269504088896340736f;
    // split off the class-token key when that saves a key tile (L14: 577 keys = 10 tiles, 576 = 9, no
    // masked tile); the class-token query then goes last in the rotated row order
    const int lo = (has_cls && S > 1 && (S - 1 + 63) / 64 < (S + 63) / 64) ? 1 : 0;
    const int rows = S;
    // Waves per workgroup.  A workgroup's time per key tile is set by the staging / barrier /
    // softmax latency chain, not by how many of its waves hold query rows (measured, L14, current kernel:
    // 8 waves 0.171 ms, 7 waves 0.187 ms, 6 waves 0.235 ms, 4 waves 0.184 ms although 8 waves pad 577 rows to 768), so take
    // the fewest workgroups per (image, head) and, among equals, the most waves (16 waves per CU).
    int best = 4, best_blocks = 1 << 30;
    for (int nw : {8, 7, 6, 4}) {
        const int blocks = (rows + nw * 32 - 1) / (nw * 32);
        if (blocks < best_blocks) { best_blocks = blocks; best = nw; }
    }
    if (g_attn_force_nw) best = g_attn_force_nw;
    if (hd == 64) {
        switch (best) {
            case 8: launch_attn_nw<64, 8>(qkv, ld, out, ldo, B, S, H, c, lo, lo, st); break;
            case 7: launch_attn_nw<64, 7>(qkv, ld, out, ldo, B, S, H, c, lo, lo, st); break;
            case 6: launch_attn_nw<64, 6>(qkv, ld, out, ldo, B, S, H, c, lo, lo, st); break;
            default: launch_attn_nw<64, 4>(qkv, ld, out, ldo, B, S, H, c, lo, lo, st); break;
        }
    } else {
        // 64 KB of LDS per workgroup: two 8-wave workgroups per CU
        if (best >= 6) launch_attn_nw<96, 8>(qkv, ld, out, ldo, B, S, H, c, lo, lo, st);
        else launch_attn_nw<96, 4>(qkv, ld, out, ldo, B, S, H, c, lo, lo, st);
    }
    REVO_HIP_CHECK(hipGetLastError());
    return 0;
}
int launch_attention(const bf16_t* qkv, long ld, bf16_t* out, long ldo, int B, int S, int H, int hd, hipStream_t st) {
    return launch_attention_ex(qkv, ld, out, ldo, B, S, H, hd, 1, st);
}


}  // namespace revo
