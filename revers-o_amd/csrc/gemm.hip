// ViT linear layers on MFMA: C = A . W^T with the elementwise tail of each layer
// fused into the epilogue (bias, exact-erf GELU, LayerScale + residual add into
// the fp32 residual stream, patch-embed position add).  K2/K4/K7/K8/K10 of
// SURVEY.md §2b; replaces the F.linear / conv2d calls the upstream PE module
// dispatches from encode_image (reference call site core_system.py:442).
#include <type_traits>
#include "gemm_core.h"
#include "gemm256_core.h"
#include "kernels.h"

namespace revo {

// Exact-erf GELU on register pairs.  erf(z) = 1 - 1/(1 + a1 z + ... + a6 z^6)^16 for z >= 0
// (Abramowitz-Stegun 7.1.28, |eps| <= 3e-7; ~2e-6 once the 16th power is taken in fp32): one
// reciprocal and no exponential per element, everything else on the packed fp32 pipe
// (v_pk_fma_f32 / v_pk_mul_f32), about half the VALU issue slots of the 7.1.26 form.  The result
// feeds a bf16 store (2^-9 relative), three orders of magnitude coarser than the approximation.
//   gelu(x) = 0.5 x (1 + erf(x / sqrt 2)) = 0.5 (x + |x| (1 - r)),   r = 1 / poly(|x| / sqrt 2)^16
__device__ __forceinline__ f32x2 gelu_erf2(f32x2 x) {
    const f32x2 ax = {__builtin_fabsf(x.x), __builtin_fabsf(x.y)};
    const f32x2 z = ax * 0.70710678118654752f;
    f32x2 p = z * 0.0000430638f + 0.0002765672f;
    p = p * z + 0.0001520143f;
    p = p * z + 0.0092705272f;
    p = p * z + 0.0422820123f;
    p = p * z + 0.0705230784f;
    p = p * z + 1.0f;
    p = p * p;
    p = p * p;
    p = p * p;
    p = p * p;
    const f32x2 r = {__builtin_amdgcn_rcpf(p.x), __builtin_amdgcn_rcpf(p.y)};   // rcp(inf) = 0 for large |x|
    return (x + ax * (1.0f - r)) * 0.5f;
}
__device__ __forceinline__ f32x4 gelu_erf4(f32x4 v) {
    const f32x2 lo = gelu_erf2((f32x2){v[0], v[1]}), hi = gelu_erf2((f32x2){v[2], v[3]});
    return (f32x4){lo.x, lo.y, hi.x, hi.y};
}
// The same function on four f32x4 at once, written stage by stage over the eight register pairs: one element's GELU is a
// chain of ~15 DEPENDENT packed operations (hipcc 7.2 puts an s_nop between each two of them), and fed one f32x4 after
// the other the epilogue of fc1 is bound by that chain's latency, not by the VALU's rate -- 6.75 us per 256 x 256 tile
// of which 5 are this function (profiles/r05_gemm_phase_groups.json).  Eight independent chains side by side fill each
// other's gaps.  Per element the operations and their order are gelu_erf2's: same bits.
#ifndef REVO_LNC_ABLATE       // timing-only ablations of the folded consumer (var build): 1 no merge, 2 no epilogue math, 4 no statistics DMA
#define REVO_LNC_ABLATE 0
#endif
#ifndef REVO_GELU_WIDTH
#define REVO_GELU_WIDTH 2
#endif
template <int NV>
__device__ __forceinline__ void gelu_erf4xn(f32x4 (&v)[NV]) {
    constexpr int NP = 2 * NV;
    f32x2 x[NP], ax[NP], z[NP], q[NP];
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        x[i] = (f32x2){v[i >> 1][(i & 1) * 2], v[i >> 1][(i & 1) * 2 + 1]};
        ax[i] = (f32x2){__builtin_fabsf(x[i].x), __builtin_fabsf(x[i].y)};
    }
#pragma unroll
    for (int i = 0; i < NP; ++i) z[i] = ax[i] * 0.70710678118654752f;
#pragma unroll
    for (int i = 0; i < NP; ++i) q[i] = z[i] * 0.0000430638f + 0.0002765672f;
#pragma unroll
    for (int i = 0; i < NP; ++i) q[i] = q[i] * z[i] + 0.0001520143f;
#pragma unroll
    for (int i = 0; i < NP; ++i) q[i] = q[i] * z[i] + 0.0092705272f;
#pragma unroll
    for (int i = 0; i < NP; ++i) q[i] = q[i] * z[i] + 0.0422820123f;
#pragma unroll
    for (int i = 0; i < NP; ++i) q[i] = q[i] * z[i] + 0.0705230784f;
#pragma unroll
    for (int i = 0; i < NP; ++i) q[i] = q[i] * z[i] + 1.0f;
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int i = 0; i < NP; ++i) q[i] = q[i] * q[i];
#pragma unroll
    for (int i = 0; i < NP; ++i) q[i] = (f32x2){__builtin_amdgcn_rcpf(q[i].x), __builtin_amdgcn_rcpf(q[i].y)};
#pragma unroll
    for (int i = 0; i < NP; ++i) q[i] = 1.0f - q[i];
#pragma unroll
    for (int i = 0; i < NP; ++i) x[i] = x[i] + ax[i] * q[i];
#pragma unroll
    for (int i = 0; i < NP; ++i) x[i] = x[i] * 0.5f;
#pragma unroll
    for (int i = 0; i < NV; ++i) v[i] = (f32x4){x[2 * i].x, x[2 * i].y, x[2 * i + 1].x, x[2 * i + 1].y};
}

// ---- LayerNorm folded into the GEMMs around it (kernels.h GemmArgs::lnf_* / lnc_*) ----------
// sum over the 16 lanes of a DPP row (lanes 16 r .. 16 r + 15): quad butterflies, then the two mirrors; every lane gets a sum
__device__ __forceinline__ float row16_sum(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, true));    // quad_perm [1,0,3,2]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xf, 0xf, true));    // quad_perm [2,3,0,1]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xf, 0xf, true));   // row_half_mirror
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xf, 0xf, true));   // row_mirror
    return v;
}
// P partial (mean, M2) pairs of equal-sized column slices of one row, in slot order -> (rstd, -mean * rstd).  One fixed
// order of operations wherever a row's statistics are merged (every consumer form gives a row the same bits).
constexpr int LNF_SLICE = 256;      // columns per partial = one column tile of the 256 x 256 kernel
// (`part` is read twice -- LDS or L1-resident global memory -- rather than copied: a run-time-indexed local array would live in scratch)
__device__ __forceinline__ void lnf_merge(const float2* part, int P, float eps, float& rstd, float& mr) {
    float sm = 0.f, sq = 0.f;
    for (int i = 0; i < P; ++i) { const float2 t = part[i]; sm += t.x; sq += t.y; }
    const float mean = sm / (float)P;
    float dd = 0.f;
    for (int i = 0; i < P; ++i) { const float d = part[i].x - mean; dd = fmaf(d, d, dd); }
    const float var = fmaf((float)LNF_SLICE, dd, sq) / (float)(LNF_SLICE * P);
    rstd = rsqrtf(var + eps);
    mr = -mean * rstd;
}

template <int EPI, int MF, int NF>
__device__ __forceinline__ void gemm_epilogue(const GemmArgs& p, int m_base, int n_base, int lane,
                                              f32x4 (&acc)[MF][NF]) {
    const int lr = lane & 15, lq = lane >> 4;
    const bool lnc = (EPI == EPI_BF16 || EPI == EPI_BF16_GELU || EPI == EPI_BF16_ROPE) && p.lnc_stats != nullptr;
#pragma unroll
    for (int m = 0; m < MF; ++m) {
        const int row = m_base + m * 16 + lr;
        if (row >= p.M) continue;
        float ln_rstd = 1.f, ln_mr = 0.f;
        if (lnc)        // the row's statistics straight from global memory (the small-tile kernels: a few leftover rows)
            lnf_merge(p.lnc_stats + (long)row * p.lnc_parts, p.lnc_parts, p.lnc_eps, ln_rstd, ln_mr);
        long orow = row;
        int prow = 0;
        if (EPI == EPI_PATCH) {
            const int b = row / p.G2, g = row - b * p.G2;
            orow = (long)b * p.S + p.cls + g;
            prow = p.cls + g;
        }
#pragma unroll
        for (int n = 0; n < NF; ++n) {
            const int col = n_base + n * 16 + lq * 4;
            if (col >= p.N) continue;
            f32x4 v = acc[m][n];
            if (lnc) {
                const f32x4 c = *(const f32x4*)(p.lnc_c + col);
                const f32x4 b = p.bias ? *(const f32x4*)(p.bias + col) : (f32x4){0.f, 0.f, 0.f, 0.f};
                v = v * ln_rstd + (c * ln_mr + b);          // rstd (acc - mean c) + b', the 256 x 256 epilogue's operation order
            } else
            if (p.bias) {
                const f32x4 b = *(const f32x4*)(p.bias + col);
                v += b;
            }
            if (EPI == EPI_BF16_ROPE) {
                {
                    // K5 on the accumulator layout: a lane holds two interleaved pairs of one token.  The values are
                    // rounded to bf16 first, as the row-coalesced epilogue of the 256 x 256 kernel does (it rotates what
                    // it reads back from its bf16 slab) and as the stand-alone RoPE kernel sees them: same bits on
                    // every path.  Branch-free on purpose (columns past rope_cols rotate by the identity): with the
                    // rotation inside a divergent `if`, hipcc 7.2's code for this epilogue gave run-to-run different
                    // results in one element per fragment once the grid exceeded one resident set of workgroups.
                    const bool rot = col < p.rope_cols;
                    const float2* t = p.rope_cs + (long)(row % p.rope_S) * (p.rope_hd >> 1) + ((col % p.rope_hd) >> 1);
                    const f32x4 tc = *(const f32x4*)t;
#pragma unroll
                    for (int h2 = 0; h2 < 2; ++h2) {
                        const float x0 = bf16_to_f32(f32_to_bf16(v[2 * h2])), x1 = bf16_to_f32(f32_to_bf16(v[2 * h2 + 1]));
                        const float cx = rot ? tc[2 * h2] : 1.0f, cy = rot ? tc[2 * h2 + 1] : 0.0f;
                        v[2 * h2] = x0 * cx - x1 * cy;
                        v[2 * h2 + 1] = x1 * cx + x0 * cy;
                    }
                }
            }
            if (EPI == EPI_BF16 || EPI == EPI_BF16_GELU || EPI == EPI_BF16_ROPE) {
                if (EPI == EPI_BF16_GELU) v = gelu_erf4(v);
                uint2 o;
                o.x = pack_bf16x2(v[0], v[1]);
                o.y = pack_bf16x2(v[2], v[3]);
                *(uint2*)((bf16_t*)p.C + orow * p.ldc + col) = o;
            } else if (EPI == EPI_RESID_F32) {
                float* dst = (float*)p.C + orow * p.ldc + col;
                f32x4 x = *(const f32x4*)dst;
                if (p.gamma) {
                    const f32x4 g = *(const f32x4*)(p.gamma + col);
                    v *= g;
                }
                x += v;
                *(f32x4*)dst = x;
            } else if (EPI == EPI_F32) {
                *(f32x4*)((float*)p.C + orow * p.ldc + col) = v;
            } else if (EPI == EPI_PATCH) {
                const f32x4 pe = *(const f32x4*)(p.pos + (long)prow * p.N + col);
                v += pe;
                *(f32x4*)((float*)p.C + orow * p.ldc + col) = v;
            }
        }
    }
}

// 32 rows (accumulator fragments 2*QT, 2*QT+1) of a wave's fp32 sub-tile: transpose through
// the wave's LDS slab, then whole 256-byte row segments to/from global memory.  For the residual
// epilogue the old values of the quarter arrive in `res` (loaded by gemm256_load_resid one quarter
// ahead, so that only the first of the four HBM round trips of a tile is exposed).
// Branch-free: rows / columns past the matrix edge re-read the last valid row / column chunk (the values are never
// stored).  With the loads inside per-lane branches the waitcnt pass lost track of which load a register came from
// and put s_waitcnt vmcnt(0) in front of every one of the 16 store steps of the second half of a tile -- each of
// them then waited for the previous step's store to be acknowledged by L2.
// nt: non-temporal (the folded-LayerNorm form: nothing reads the fp32 rows before the next residual GEMM, a whole GEMM
// later -- the bf16 copy written beside them is what the next kernel streams, and should be what stays in the caches)
template <int QT>
__device__ __forceinline__ void gemm256_load_resid(const GemmArgs& p, int m_base, int n_base, int lane, f32x4 (&res)[8],
                                                   bool nt = false) {
    int gcol = n_base + (lane & 15) * 4;
    gcol = gcol < p.N ? gcol : p.N - 4;
#pragma unroll
    for (int it = 0; it < 8; ++it) {
        int grow = m_base + QT * 32 + it * 4 + (lane >> 4);
        grow = grow < p.M ? grow : p.M - 1;
        const f32x4* src = (const f32x4*)((const float*)p.C + (long)grow * p.ldc + gcol);
        res[it] = nt ? __builtin_nontemporal_load(src) : *src;
    }
}
// lnst (EPI_RESID_F32, optional; LDS): the LayerNorm that follows is folded into the GEMMs around it -- every new row
// segment is also stored as bf16 (the next GEMM's A operand) and its (mean, M2) over this wave's 64 columns goes to
// lnst[row in the wave's 128 rows * 4] (the workgroup merges the four waves' slices behind its next barrier).
template <int EPI, int QT>
__device__ __forceinline__ void gemm256_epilogue_f32_quarter(const GemmArgs& p, char* slab, int m_base, int n_base,
                                                             int lane, f32x4 (&acc)[8][4], const f32x4 (&bias4)[4],
                                                             const f32x4 (&gamma4)[4], const f32x4 (&res)[8],
                                                             float2* lnst = nullptr) {
    constexpr int RS = 272;   // 64 fp32 + 16 bytes of padding per slab row
    asm volatile("" : "+v"(lane) :: "memory");   // keep this quarter's address arithmetic inside it
    const int lr = lane & 15, lq = lane >> 4;
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < 4; ++n) {
            f32x4 v = acc[QT * 2 + m][n] + bias4[n];
            if (EPI == EPI_RESID_F32) v *= gamma4[n];
            *(f32x4*)(slab + (m * 16 + lr) * RS + (n * 16 + lq * 4) * 4) = v;
        }
    const int gcol = n_base + (lane & 15) * 4;
    float keep_m = 0.f, keep_q = 0.f;
#pragma unroll
    for (int it = 0; it < 8; ++it) {
        const int rl = it * 4 + (lane >> 4);
        f32x4 v = *(const f32x4*)(slab + rl * RS + (lane & 15) * 16);
        const int grow = m_base + QT * 32 + rl;
        if (EPI == EPI_RESID_F32) v += res[it];
        if (grow < p.M && gcol < p.N) {
            float* dst = (float*)p.C + (long)grow * p.ldc + gcol;
            if (EPI == EPI_RESID_F32 && lnst) __builtin_nontemporal_store(v, (f32x4*)dst);
            else *(f32x4*)dst = v;
            if (EPI == EPI_RESID_F32 && lnst) {
                uint2 o;
                o.x = pack_bf16x2(v[0], v[1]);
                o.y = pack_bf16x2(v[2], v[3]);
                *(uint2*)(p.lnf_xb + (long)grow * p.lnf_ldxb + gcol) = o;
            }
        }
        if (EPI == EPI_RESID_F32 && lnst) {           // wave-uniform
            // two-pass statistics of the row's 64 columns held by its 16 lanes (rows past the edge: never read back)
            const float mean = row16_sum((v[0] + v[1]) + (v[2] + v[3])) * (1.0f / 64.0f);
            const float d0 = v[0] - mean, d1 = v[1] - mean, d2 = v[2] - mean, d3 = v[3] - mean;
            const float q = row16_sum(fmaf(d0, d0, d1 * d1) + fmaf(d2, d2, d3 * d3));
            if ((lane & 15) == it) { keep_m = mean; keep_q = q; }      // lane (it, lq) keeps row it * 4 + lq of this quarter
        }
    }
    if (EPI == EPI_RESID_F32 && lnst && (lane & 15) < 8)
        lnst[(QT * 32 + (lane & 15) * 4 + (lane >> 4)) * 4] = make_float2(keep_m, keep_q);
    asm volatile("" ::: "memory");
}

// Epilogue of the 256 x 256 kernel.  A lane's accumulator fragment is 4 columns of one
// row, so storing it directly issues 32 narrow stores per lane that each touch 16 rows
// (measured: ~7 us per tile, store-issue bound).  Instead every wave transposes its
// 128 x 64 sub-tile through a wave-private LDS slab (free after the main loop) and
// reads/writes global memory in whole 128-byte (bf16) or 256-byte (fp32) row segments,
// 16 bytes per lane.  Bias, GELU and LayerScale are applied before the transpose, the
// residual add after it (on the coalesced rows).
// SKIP_DEAD (192-row tiles): a wave whose second 64 rows lie past p.M (the caller passes the tile's row limit as p.M)
// skips them altogether instead of masking their stores.
// ---- the same residual epilogue with 8 columns per lane and the stream optionally in two bf16 planes (kernels.h xp_*)
// res[it][2]: the old values of step `it` (8 rows per step, 8 lanes per row): 8 fp32, or 8 bf16 hi + 8 bf16 lo
template <int QT, bool PIN>
__device__ __forceinline__ void gemm256_resid8_load(const GemmArgs& p, int m_base, int n_base, int lane, uint4 (&res)[4][2]) {
    typedef uint32_t nt_u32x4 __attribute__((ext_vector_type(4)));
    asm volatile("" : "+v"(lane));       // (derived here, not ahead of the main loop: hoisted, the row / column offsets were parked in scratch)
    int gcol = n_base + (lane & 7) * 8;
    gcol = gcol < p.N ? gcol : p.N - 8;
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        int grow = m_base + QT * 32 + it * 8 + (lane >> 3);
        grow = grow < p.M ? grow : p.M - 1;
        const nt_u32x4 *s0, *s1;
        if (PIN) {
            s0 = (const nt_u32x4*)(p.xp_hi + (long)grow * p.xp_ld + gcol);
            s1 = (const nt_u32x4*)(p.xp_lo + (long)grow * p.xp_ld + gcol);
        } else {
            s0 = (const nt_u32x4*)((const float*)p.C + (long)grow * p.ldc + gcol);
            s1 = s0 + 1;
        }
        res[it][0] = __builtin_bit_cast(uint4, __builtin_nontemporal_load(s0));
        res[it][1] = __builtin_bit_cast(uint4, __builtin_nontemporal_load(s1));
    }
}
__device__ __forceinline__ float row8_sum(float v) {      // over the 8 lanes that share a row here (half a DPP row)
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xf, 0xf, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xf, 0xf, true));
    return v;
}
// PIN / POUT: the old values come from / the new ones go to the planes; ST: row statistics (and, with fp32 rows out, their bf16 copy)
template <int QT, bool PIN, bool POUT, bool ST>
__device__ __forceinline__ void gemm256_resid8_quarter(const GemmArgs& p, char* slab, int m_base, int n_base, int lane,
                                                       f32x4 (&acc)[8][4], const uint4 (&res)[4][2], float2* lnst) {
    constexpr int RS = 272;
    typedef uint32_t nt_u32x4 __attribute__((ext_vector_type(4)));
    asm volatile("" : "+v"(lane) :: "memory");
    const int lr = lane & 15, lq = lane >> 4;
    // (acc already holds gamma * (acc + bias): the caller scales it in place once, so that the bias and gain vectors are
    //  not live across the four quarters -- with them the quarters spilled accumulator registers to scratch)
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < 4; ++n) *(f32x4*)(slab + (m * 16 + lr) * RS + (n * 16 + lq * 4) * 4) = acc[QT * 2 + m][n];
    const int gcol = n_base + (lane & 7) * 8;
    float keep_m = 0.f, keep_q = 0.f;
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int rl = it * 8 + (lane >> 3);
        const f32x4 v0 = *(const f32x4*)(slab + rl * RS + (lane & 7) * 32), v1 = *(const f32x4*)(slab + rl * RS + (lane & 7) * 32 + 16);
        float x[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
        const uint32_t a[4] = {res[it][0].x, res[it][0].y, res[it][0].z, res[it][0].w};
        const uint32_t b[4] = {res[it][1].x, res[it][1].y, res[it][1].z, res[it][1].w};
        if (PIN) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {      // old value = hi + lo (exact in fp32)
                x[2 * j] += __uint_as_float(a[j] << 16) + __uint_as_float(b[j] << 16);
                x[2 * j + 1] += __uint_as_float(a[j] & 0xffff0000u) + __uint_as_float(b[j] & 0xffff0000u);
            }
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) { x[j] += __uint_as_float(a[j]); x[4 + j] += __uint_as_float(b[j]); }
        }
        const int grow = m_base + QT * 32 + rl;
        if (grow < p.M && gcol < p.N) {
            uint32_t hi[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) hi[j] = pack_bf16x2(x[2 * j], x[2 * j + 1]);
            if (POUT) {
                uint32_t lo[4];
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    lo[j] = pack_bf16x2(x[2 * j] - __uint_as_float(hi[j] << 16), x[2 * j + 1] - __uint_as_float(hi[j] & 0xffff0000u));
                *(uint4*)(p.xp_hi + (long)grow * p.xp_ld + gcol) = make_uint4(hi[0], hi[1], hi[2], hi[3]);
                __builtin_nontemporal_store((nt_u32x4){lo[0], lo[1], lo[2], lo[3]}, (nt_u32x4*)(p.xp_lo + (long)grow * p.xp_ld + gcol));
            } else {
                float* dst = (float*)p.C + (long)grow * p.ldc + gcol;
                __builtin_nontemporal_store((f32x4){x[0], x[1], x[2], x[3]}, (f32x4*)dst);
                __builtin_nontemporal_store((f32x4){x[4], x[5], x[6], x[7]}, (f32x4*)(dst + 4));
                if (ST) *(uint4*)(p.lnf_xb + (long)grow * p.lnf_ldxb + gcol) = make_uint4(hi[0], hi[1], hi[2], hi[3]);
            }
        }
        if (ST) {             // two-pass statistics of the row's 64 columns held by its 8 lanes
            const float mean = row8_sum(((x[0] + x[1]) + (x[2] + x[3])) + ((x[4] + x[5]) + (x[6] + x[7]))) * (1.0f / 64.0f);
            float q = 0.f;
#pragma unroll
            for (int j = 0; j < 8; ++j) { const float d = x[j] - mean; q = fmaf(d, d, q); }
            q = row8_sum(q);
            if ((lane & 7) == it) { keep_m = mean; keep_q = q; }       // lane (it, row) keeps row it * 8 + (lane >> 3)
        }
    }
    if (ST && (lane & 7) < 4) lnst[(QT * 32 + (lane & 7) * 8 + (lane >> 3)) * 4] = make_float2(keep_m, keep_q);
    asm volatile("" ::: "memory");
}

// lnst: gemm256_epilogue_f32_quarter (producer side of a folded LayerNorm).  lnmr (consumer side, bf16 epilogues; LDS):
// (rstd, -mean rstd) of the tile's 256 rows, indexed from the wave's first row.
// XP (EPI_RESID_F32): the residual stream in planes (kernels.h xp_*): 0 = fp32 rows, 1 = fp32 in / planes out, 2 = planes
// in / planes out, 3 = planes in / fp32 out, no statistics (the last fc2 of a forward).  A compile-time choice, one kernel
// instantiation each: with run-time tests (or four copies) of the formats inside one kernel hipcc spilled 35-160 registers.
template <int EPI, bool SKIP_DEAD = false, int XP = 0>
__device__ __forceinline__ void gemm256_epilogue(const GemmArgs& p, char* slab, int m_base, int n_base, int lane,
                                                 f32x4 (&acc)[8][4], float2* lnst = nullptr, const float2* lnmr = nullptr) {
    static_assert(EPI == EPI_BF16 || EPI == EPI_BF16_GELU || EPI == EPI_F32 || EPI == EPI_RESID_F32 ||
                  EPI == EPI_BF16_ROPE, "");
    // The main loop runs at the 256-VGPR limit: keep every epilogue value from being
    // computed (or loaded) ahead of it by making the lane id opaque here.
    asm volatile("" : "+v"(lane) :: "memory");
    const int lq = lane >> 4;
    f32x4 bias4[4], gamma4[4];
#pragma unroll
    for (int n = 0; n < 4; ++n) {
        const int col = n_base + n * 16 + lq * 4;
        bias4[n] = (p.bias && col < p.N) ? *(const f32x4*)(p.bias + col) : (f32x4){0.f, 0.f, 0.f, 0.f};
        if (EPI == EPI_RESID_F32)
            gamma4[n] = (p.gamma && col < p.N) ? *(const f32x4*)(p.gamma + col) : (f32x4){1.f, 1.f, 1.f, 1.f};
    }
    if (EPI == EPI_BF16 || EPI == EPI_BF16_GELU || EPI == EPI_BF16_ROPE) {
        constexpr int RS = 144;   // 64 bf16 + 16 bytes of padding per slab row
        // folded LayerNorm (consumer side): out = rstd (acc - mean c) + b'  -- column sums of the rounded weights per lane
        f32x4 c4[4];
        if (lnmr) {
#pragma unroll
            for (int n = 0; n < 4; ++n) {
                const int col = n_base + n * 16 + lq * 4;
                c4[n] = col < p.N ? *(const f32x4*)(p.lnc_c + col) : (f32x4){0.f, 0.f, 0.f, 0.f};
            }
        }
        f32x4 rtab_next[8][2];
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            if (SKIP_DEAD && m_base + half * 64 >= p.M) break;          // wave-uniform
            asm volatile("" : "+v"(lane) :: "memory");
            const int lr = lane & 15, lq = lane >> 4;
            // RoPE table entries of a half's 8 store steps, requested before the LDS transpose so that
            // their latency is not paid once per step (a lane holds 4 interleaved pairs of one token).  The second half's
            // are requested behind the first half's fragment phase (its accumulators are dead by then) and BEFORE the first
            // half's stores: loads and stores retire in order, and behind those stores the wait for them was a drain.
            auto rope_load = [&](int hf, f32x4 (&rt)[8][2]) {
                const int gcol_ = n_base + (lane & 7) * 8;
                int tok_ = (m_base + hf * 64 + (lane >> 3)) % p.rope_S;        // one division per 64 rows; then +8 per step
                const float2* cs_ = p.rope_cs + ((gcol_ % p.rope_hd) >> 1);
#pragma unroll
                for (int it = 0; it < 8; ++it) {
                    const float2* t = cs_ + (long)tok_ * (p.rope_hd >> 1);
                    rt[it][0] = *(const f32x4*)t;
                    rt[it][1] = *(const f32x4*)(t + 2);
                    tok_ += 8;
                    if (tok_ >= p.rope_S) tok_ -= p.rope_S;
                }
            };
            f32x4 rtab[8][2];
            if (EPI == EPI_BF16_ROPE) {
                if (half == 0) rope_load(0, rtab);
                else {
#pragma unroll
                    for (int it = 0; it < 8; ++it) { rtab[it][0] = rtab_next[it][0]; rtab[it][1] = rtab_next[it][1]; }
                }
            }
            // fragment phase: bias (or the folded LayerNorm's rstd (acc - mean c) + b'), GELU, bf16, transpose through the slab.
            // The folded form is a separate copy of the loop (a per-fragment test of lnmr cut the GELU chains into
            // one basic block each).
            auto frag_phase = [&](auto lnc_tag) {
                constexpr bool LNC = decltype(lnc_tag)::value;
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    float2 st = make_float2(1.f, 0.f);
                    if (LNC) st = lnmr[half * 64 + m * 16 + lr];
                    if (EPI == EPI_BF16_GELU) {
                        constexpr int GV = REVO_GELU_WIDTH;    // fragments side by side: 2 GV independent GELU chains (gelu_erf4xn)
#pragma unroll
                        for (int n0 = 0; n0 < 4; n0 += GV) {
                            f32x4 v[GV];
#pragma unroll
                            for (int n = 0; n < GV; ++n) {
                                if (LNC && !(REVO_LNC_ABLATE & 2)) v[n] = acc[half * 4 + m][n0 + n] * st.x + (c4[n0 + n] * st.y + bias4[n0 + n]);
                                else v[n] = acc[half * 4 + m][n0 + n] + bias4[n0 + n];
                            }
                            gelu_erf4xn<GV>(v);
#pragma unroll
                            for (int n = 0; n < GV; ++n) {
                                uint2 o;
                                o.x = pack_bf16x2(v[n][0], v[n][1]);
                                o.y = pack_bf16x2(v[n][2], v[n][3]);
                                *(uint2*)(slab + (m * 16 + lr) * RS + ((n0 + n) * 16 + lq * 4) * 2) = o;
                            }
                        }
                    } else {
#pragma unroll
                        for (int n = 0; n < 4; ++n) {
                            f32x4 v;
                            if (LNC && !(REVO_LNC_ABLATE & 2)) v = acc[half * 4 + m][n] * st.x + (c4[n] * st.y + bias4[n]);
                            else v = acc[half * 4 + m][n] + bias4[n];
                            uint2 o;
                            o.x = pack_bf16x2(v[0], v[1]);
                            o.y = pack_bf16x2(v[2], v[3]);
                            *(uint2*)(slab + (m * 16 + lr) * RS + (n * 16 + lq * 4) * 2) = o;
                        }
                    }
                }
            };
            if (lnmr) frag_phase(std::true_type{});
            else frag_phase(std::false_type{});
            if (EPI == EPI_BF16_ROPE && half == 0 && !(SKIP_DEAD && m_base + 64 >= p.M)) rope_load(1, rtab_next);
            const int gcol = n_base + (lane & 7) * 8;
#pragma unroll
            for (int it = 0; it < 8; ++it) {
                const int rl = it * 8 + (lane >> 3);
                uint4 v = *(const uint4*)(slab + rl * RS + (lane & 7) * 16);
                const int grow = m_base + half * 64 + rl;
                if (EPI == EPI_BF16_ROPE) {
                    if (gcol < p.rope_cols) {       // K5 on the coalesced rows
                        uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const float x0 = bf16_to_f32((bf16_t)(w[j] & 0xffff)), x1 = bf16_to_f32((bf16_t)(w[j] >> 16));
                            const float cx = rtab[it][j >> 1][(j & 1) * 2], cy = rtab[it][j >> 1][(j & 1) * 2 + 1];
                            w[j] = pack_bf16x2(x0 * cx - x1 * cy, x1 * cx + x0 * cy);
                        }
                        v = make_uint4(w[0], w[1], w[2], w[3]);
                    }
                }
                // qkv and the MLP hidden activations (226 / 302 MB at PE-L14, batch 64) are written once and read once by the
                // next kernel: stored non-temporally they leave more of L2 / Infinity Cache to the operands and to the
                // fp32 residual stream (measured in the step: fc1 -1.4 %, fc2 -1.3 %, qkv -1 %)
                typedef uint32_t nt_u32x4 __attribute__((ext_vector_type(4)));
                if (EPI == EPI_BF16_ROPE || EPI == EPI_BF16_GELU) {
                    if (grow < p.M && gcol < p.N)
                        __builtin_nontemporal_store(__builtin_bit_cast(nt_u32x4, v), (nt_u32x4*)((bf16_t*)p.C + (long)grow * p.ldc + gcol));
                } else
                if (grow < p.M && gcol < p.N) *(uint4*)((bf16_t*)p.C + (long)grow * p.ldc + gcol) = v;
            }
            asm volatile("" ::: "memory");
        }
    } else {
        // Old values of two quarters at a time.  Loads and stores share one in-order-per-type counter (vmcnt), so a
        // load result that is needed while stores are in flight costs a full drain of both: prefetching a quarter
        // "one ahead" between the stores of the previous ones made every use such a drain.  Two batches of loads,
        // each issued when no load result is outstanding, leave two drains per tile.
        if (SKIP_DEAD && m_base >= p.M) return;                         // wave-uniform: a short piece's second wave-row
        if constexpr (EPI == EPI_RESID_F32 && XP != 0) {
            // the stream in (or going into / coming out of) two bf16 planes: 8 columns per lane.  Old values in three
            // batches (quarter 0 | quarters 1, 2 | quarter 3), each requested when no load result is outstanding (two
            // batches like the 4-column path below: the same step time, scripts/experiments/r5_resid8_batches.sh).
            constexpr bool PIN = XP >= 2, POUT = XP <= 2, ST = XP <= 2;
#pragma unroll
            for (int m = 0; m < 8; ++m)
#pragma unroll
                for (int n = 0; n < 4; ++n) acc[m][n] = (acc[m][n] + bias4[n]) * gamma4[n];
            asm volatile("" ::: "memory");
            uint4 pa[4][2], pb[4][2];
            gemm256_resid8_load<0, PIN>(p, m_base, n_base, lane, pa);
            gemm256_resid8_quarter<0, PIN, POUT, ST>(p, slab, m_base, n_base, lane, acc, pa, lnst);
            gemm256_resid8_load<1, PIN>(p, m_base, n_base, lane, pa);
            gemm256_resid8_load<2, PIN>(p, m_base, n_base, lane, pb);       // (unconditional: rows past the piece are clamped, never stored)
            gemm256_resid8_quarter<1, PIN, POUT, ST>(p, slab, m_base, n_base, lane, acc, pa, lnst);
            if (SKIP_DEAD && m_base + 64 >= p.M) return;                // wave-uniform
            gemm256_resid8_quarter<2, PIN, POUT, ST>(p, slab, m_base, n_base, lane, acc, pb, lnst);
            gemm256_resid8_load<3, PIN>(p, m_base, n_base, lane, pa);
            gemm256_resid8_quarter<3, PIN, POUT, ST>(p, slab, m_base, n_base, lane, acc, pa, lnst);
            return;
        }
        f32x4 ra[8], rb[8];
        const bool nt = lnst != nullptr;
        if (EPI == EPI_RESID_F32) gemm256_load_resid<0>(p, m_base, n_base, lane, ra, nt);
        if (EPI == EPI_RESID_F32) gemm256_load_resid<1>(p, m_base, n_base, lane, rb, nt);
        gemm256_epilogue_f32_quarter<EPI, 0>(p, slab, m_base, n_base, lane, acc, bias4, gamma4, ra, lnst);
        gemm256_epilogue_f32_quarter<EPI, 1>(p, slab, m_base, n_base, lane, acc, bias4, gamma4, rb, lnst);
        if (SKIP_DEAD && m_base + 64 >= p.M) return;                    // wave-uniform
        if (EPI == EPI_RESID_F32) gemm256_load_resid<2>(p, m_base, n_base, lane, ra, nt);
        if (EPI == EPI_RESID_F32) gemm256_load_resid<3>(p, m_base, n_base, lane, rb, nt);
        gemm256_epilogue_f32_quarter<EPI, 2>(p, slab, m_base, n_base, lane, acc, bias4, gamma4, ra, lnst);
        gemm256_epilogue_f32_quarter<EPI, 3>(p, slab, m_base, n_base, lane, acc, bias4, gamma4, rb, lnst);
    }
}

// ---- folded LayerNorm in the 256 x 256 kernels: LDS beyond the operand image / the slabs
//   consumer: raw partial statistics of the tile's 256 rows (DMA, [256][parts] float2, <= 12 KiB), then (rstd, -mean rstd) per row
//   producer: the waves' 64-column partials [256 rows][4] float2 (8 KiB), merged per 256-column tile behind the epilogue
constexpr int LNC_RAW_BYTES = 256 * 6 * 8 + 1024, LNC_MR_BYTES = 256 * 8;    // parts <= 6 in these kernels (width <= 1536); + the last piece's spill-over
constexpr int G256P_LN_OFF = 98304 + 5 * 9216;                                // persistent kernel: above the last slab (144384)
static_assert(G256P_LN_OFF + LNC_RAW_BYTES + LNC_MR_BYTES <= 163840, "");
constexpr int G256_LDS_LN = G256_LDS + LNC_RAW_BYTES + LNC_MR_BYTES;          // one-tile kernel: above the operand image
constexpr int G256Q_BC_OFF = G256_LDS_LN + LNC_MR_BYTES;                      // queued-stores kernel: (rstd, -mean rstd) in two buffers, then
constexpr int G256Q_LDS = G256Q_BC_OFF + 2 * 2048;                            // bias | column sums of a tile's 256 columns, two buffers
static_assert(G256Q_LDS <= 163840, "");
// DMA of the statistics of rows [m0, m0 + 256): one 1-KiB piece per wave-instruction, pieces dealt over the 8 waves.  Issued
// BEFORE operand DMA the main loop's counted waits cover (vmcnt retires in issue order), read after the main loop's barriers.
// (lane ids are made opaque in these helpers: the main loop runs at the 256-VGPR limit, and an address the compiler computes
//  ahead of it is spilled and reloaded behind it -- a scratch load whose wait also drains the DMA queue and every older store)
template <class ARGS>      // GemmArgs, or the same struct read through the kernarg segment (gemm256q_kernel)
__device__ __forceinline__ void lnc_issue_stats(const ARGS& p, int m0, char* lds_raw, int wave, int /*lane*/) {
    if (wave >= 4) return;
    int lane;                                                                                  // (see lnc_merge_rows)
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane) :: "memory");
    // wave w requests the slots of rows 64 w .. 64 w + 63 -- the rows its own threads merge (lnc_merge_rows): its own
    // vmcnt wait is then all the ordering the merge needs (a piece that runs past the wave's share re-writes the next
    // wave's first bytes with the same values)
    const int P = p.lnc_parts;
    long left = ((long)p.M - m0) * P * 8;
    const long full = 256l * P * 8;
    left = left < 0 ? 0 : (left > full ? full : left);
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(p.lnc_stats + (long)m0 * P), 0, (int)left, 0x00020000);
    const int share = 512 * P, pieces = (share + 1023) >> 10;
    for (int j = 0; j < pieces; ++j) {
        const int off = wave * share + j * 1024;
        if (off + 1024 <= LNC_RAW_BYTES)             // (always, for parts <= 6: the raw area includes the last piece's spill-over KiB)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(lds_raw + off), 16,
                                                     (uint32_t)(off + lane * 16), 0, 0, 0);
    }
}
// BEFORE the main loop: one thread per tile row merges the row's partials into (rstd, -mean rstd).  No barrier of its own:
// a wave reads only slots it requested itself (behind its own counted wait: the statistics are older than the K-tile
// DMA the main loop's first wait leaves in flight), and the epilogue reads lds_mr behind the main loop's barriers.
// tele (wave-uniform; the tile is the launch's column tile 0 and the caller asked for telemetry): rows_left = M - m0.
// VM: vector-memory instructions this wave has issued BEHIND its statistics pieces that may stay in flight (the ordinary
// kernels: the four halves of K-tile 1's A operand; the queued-stores kernel: the previous tile's sixteen stores).
template <int VM = 4, class ARGS = GemmArgs>
__device__ __forceinline__ void lnc_merge_rows(const ARGS& p, const char* lds_raw, float2* lds_mr, int wave, int /*lane*/, bool one_ktile,
                                               bool tele = false, int rows_left = 0) {
    if (wave >= 4) return;
    // (the lane id is read from the hardware here: derived from the kernel's `lane` it became one more value alive across
    //  the main loop, and hipcc spilled accumulator registers INSIDE the K loop of the RoPE kernel)
    // (as asm: the builtin's result is loop-invariant to hipcc, which may compute it once in front of a persistent kernel's tile
    //  loop, park it in scratch around the main loop and reload it here behind an s_waitcnt vmcnt(0) -- a drain of the previous
    //  tile's stores and of the next tile's DMA)
    int lane;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane) :: "memory");
    if (one_ktile) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(%0)" :: "n"(VM) : "memory");
    const int t = wave * 64 + lane;
    if (t < 256) {
        float r, m;
        // widths 1024 and 1536 unrolled (16-byte reads, no loops, no division by a run-time count); the same operations in the
        // same order as lnf_merge
        auto merge_n = [&](auto np) {
            constexpr int NP = decltype(np)::value;
            f32x4 v[NP / 2];
#pragma unroll
            for (int i = 0; i < NP / 2; ++i) v[i] = *(const f32x4*)(lds_raw + t * (NP * 8) + i * 16);
            float sm = 0.f, sq = 0.f;
#pragma unroll
            for (int i = 0; i < NP; ++i) { sm += v[i >> 1][(i & 1) * 2]; sq += v[i >> 1][(i & 1) * 2 + 1]; }
            const float mean = sm / (float)NP;
            float dd = 0.f;
#pragma unroll
            for (int i = 0; i < NP; ++i) { const float d = v[i >> 1][(i & 1) * 2] - mean; dd = fmaf(d, d, dd); }
            const float var = fmaf((float)LNF_SLICE, dd, sq) / (float)(LNF_SLICE * NP);
            r = rsqrtf(var + p.lnc_eps);
            m = -mean * r;
        };
        if (p.lnc_parts == 4) merge_n(std::integral_constant<int, 4>{});
        else if (p.lnc_parts == 6) merge_n(std::integral_constant<int, 6>{});
        else lnf_merge((const float2*)lds_raw + t * p.lnc_parts, p.lnc_parts, p.lnc_eps, r, m);
        lds_mr[t] = make_float2(r, m);
        if (tele) {
            // one wave = 64 rows: one to three no-return atomics per wave of a column-tile-0 workgroup (rows past the matrix edge
            // read zero statistics: |mean| rstd = 0, not counted)
            const int live = rows_left - wave * 64 < 0 ? 0 : (rows_left - wave * 64 > 64 ? 64 : rows_left - wave * 64);
            const float am = lane < live ? __builtin_fabsf(m) : 0.f;
            const unsigned long long big = __builtin_amdgcn_ballot_w64(am > LNC_TELE_RATIO);
            const unsigned long long huge = __builtin_amdgcn_ballot_w64(am > 4.0f * LNC_TELE_RATIO);
            if (lane == 0 && live > 0) {                  // (ballots and counts are scalar: no vector register lives past here)
                // the counters' address in vector registers DEFINED HERE: addressed off the scalar pointer, the atomics took a
                // zero offset register that hipcc hoisted out of the persistent loop and parked in scratch around it -- and
                // every scratch access near the DMA queue is a drain of it
                unsigned long long* tp = p.lnc_tele;
                asm volatile("" : "+v"(tp));
                atomicAdd(tp, (unsigned long long)live);
                if (big) atomicAdd(tp + 1, (unsigned long long)__builtin_popcountll(big));
                if (huge) atomicAdd(tp + 2, (unsigned long long)__builtin_popcountll(huge));
            }
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
}
// producer: the four waves' 64-column partials of every row -> one (mean, M2) per row and 256-column tile, slot `tn` of the row
__device__ __forceinline__ void lnf_store_tile_stats(const GemmArgs& p, const float2* lds_st, int m0, int mend, int tn, int wave,
                                                     int lane) {
    asm volatile("" : "+v"(lane) :: "memory");
    const int t = wave * 64 + lane;
    if (t < 256 && m0 + t < mend) {
        const float2 a = lds_st[t * 4], b = lds_st[t * 4 + 1], c = lds_st[t * 4 + 2], d = lds_st[t * 4 + 3];
        const float mean = ((a.x + b.x) + (c.x + d.x)) * 0.25f;
        const float da = a.x - mean, db = b.x - mean, dc = c.x - mean, dd = d.x - mean;
        const float m2 = ((a.y + b.y) + (c.y + d.y)) + 64.0f * (fmaf(da, da, db * db) + fmaf(dc, dc, dd * dd));
        p.lnf_stats[(long)(m0 + t) * (p.N / LNF_SLICE) + tn] = make_float2(mean, m2);
    }
}

// 128 x BN x 64 tile (BN = 128 or 64), 4 waves as 2 (M) x 2 (N), each 64 x BN/2.  BN = 64 halves the work
// per workgroup so that two or three are resident per CU and hide each other's latency chains when the
// problem has about as many 128 x 128 tiles as there are CUs (the out-proj leftover rows).
template <int EPI, int BN = 128>
__global__ __launch_bounds__(GEMM_THREADS) void gemm128_kernel(GemmArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int BM = 128, MF = 4, NF = BN / 32;
    const int tiles_n = (p.N + BN - 1) / BN;
    const int tiles_m = (p.M + BM - 1) / BM;
    const int s = xcd_remap(blockIdx.x, tiles_m * tiles_n);
    const int tm = s / tiles_n, tn = s - tm * tiles_n;
    const int m0 = tm * BM, n0 = tn * BN;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const int wr = wave >> 1, wc = wave & 1;

    TileLoader<BM> la;
    TileLoader<BN> lb;
    la.init(p.A, p.lda, m0, p.M, wave, lane);
    lb.init(p.B, p.ldb, n0, p.N, wave, lane);

    f32x4 acc[MF][NF];
#pragma unroll
    for (int m = 0; m < MF; ++m)
#pragma unroll
        for (int n = 0; n < NF; ++n) acc[m][n] = (f32x4){0.f, 0.f, 0.f, 0.f};

    gemm_mainloop<BM, BN, MF, NF>(la, lb, smem, p.K, wave, lane, wr * 64, wc * (BN / 2), acc);
    gemm_epilogue<EPI, MF, NF>(p, m0 + wr * 64, n0 + wc * (BN / 2), lane, acc);
}

// The 128 x 64 tile with a six-deep DMA ring (gemm_core.h gemm_mainloop_ring), 144 KiB LDS: one workgroup per CU with
// five K-steps of loads in flight.  For problems of about one tile per CU: one image (577 rows), the leftover rows of
// the residual GEMMs at large batch.
constexpr int G128R_NST = 6;
template <int EPI>
__global__ __launch_bounds__(GEMM_THREADS) void gemm128r_kernel(GemmArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int BM = 128, BN = 64, MF = 4, NF = BN / 32;
    const int tiles_n = (p.N + BN - 1) / BN;
    const int tiles_m = (p.M + BM - 1) / BM;
    const int s = xcd_remap(blockIdx.x, tiles_m * tiles_n);
    const int tm = s / tiles_n, tn = s - tm * tiles_n;
    const int m0 = tm * BM, n0 = tn * BN;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const int wr = wave >> 1, wc = wave & 1;
    if (EPI == EPI_F32 && p.ksplit > 1) {
        // split-K partial (one image's fc2: 80 tiles with 64 K-steps each): blockIdx.y owns K-tiles [y nt / S, (y+1) nt / S) and
        // writes plain fp32 sums to its own output plane; splitk_reduce_resid_kernel adds the planes in a fixed order
        const int nt = p.K >> 6, t0 = (int)blockIdx.y * nt / p.ksplit, t1 = ((int)blockIdx.y + 1) * nt / p.ksplit;
        p.A += t0 * 64;
        p.B += t0 * 64;
        p.K = (t1 - t0) * 64;
        p.C = (float*)p.C + (long)blockIdx.y * p.c_split_stride;
    }
    TileLoader<BM> la;
    TileLoader<BN> lb;
    la.init(p.A, p.lda, m0, p.M, wave, lane);
    lb.init(p.B, p.ldb, n0, p.N, wave, lane);
    f32x4 acc[MF][NF];
#pragma unroll
    for (int m = 0; m < MF; ++m)
#pragma unroll
        for (int n = 0; n < NF; ++n) acc[m][n] = (f32x4){0.f, 0.f, 0.f, 0.f};
    gemm_mainloop_ring<BM, BN, MF, NF, G128R_NST>(la, lb, smem, p.K, wave, lane, wr * 64, wc * (BN / 2), acc);
    gemm_epilogue<EPI, MF, NF>(p, m0 + wr * 64, n0 + wc * (BN / 2), lane, acc);
}

// 256 x 256 x 64 tile, 8 waves as 2 (M) x 4 (N), each 128 x 64; see gemm256_core.h.
// DBG (compile time, scripts/gemm_ksweep.py only): 1 = no epilogue stores, 2 = no main loop.
template <int EPI, int DBG>
__global__ __launch_bounds__(G256_THREADS, 2) void gemm256_kernel(GemmArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // 2-D XCD map: blocks b, b+8, ... share an XCD (and its 4 MiB L2).  XCD (xi, xj) of a
    // gx x gy arrangement owns the M-tile stripe xi and the N-tile stripe xj, so a weight
    // stripe stays L2 resident while the XCD sweeps its activation rows (p.gy is chosen by
    // the launcher so that the stripe fits); inside the region N runs fastest.
    const int tiles_n = (p.N + 255) / 256;
    const int tiles_m = (p.M + 255) / 256;
    const int gy = p.gy, gx = 8 / gy;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int xi = xcd / gy, xj = xcd - xi * gy;
    const int pm = (tiles_m + gx - 1) / gx, pn = (tiles_n + gy - 1) / gy;
    const int m_lo = xi * pm, n_lo = xj * pn;
    const int m_cnt = (tiles_m - m_lo) < pm ? (tiles_m - m_lo) : pm;
    const int n_cnt = (tiles_n - n_lo) < pn ? (tiles_n - n_lo) : pn;
    if (m_cnt <= 0 || n_cnt <= 0 || slot >= m_cnt * n_cnt) return;
    const int tm = m_lo + slot / n_cnt, tn = n_lo + slot % n_cnt;
    const int m0 = tm * 256, n0 = tn * 256;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;

    if (DBG & 4) {
        // split-K partial: blockIdx.y owns K-tiles [y nt / S, (y+1) nt / S) and writes plain fp32 sums to
        // its own output plane (the launcher's reduce kernel adds the planes in a fixed order)
        const int nt = p.K >> 6, t0 = (int)blockIdx.y * nt / p.ksplit, t1 = ((int)blockIdx.y + 1) * nt / p.ksplit;
        p.A += t0 * 64;
        p.B += t0 * 64;
        p.K = (t1 - t0) * 64;
        p.C = (float*)p.C + (long)blockIdx.y * p.c_split_stride;
    }
    G256Operand A, B;
    g256_operand_init(A, p.A, p.lda, p.M, m0, wave, lane);
    g256_operand_init(B, p.B, p.ldb, p.N, n0, wave, lane);

    f32x4 acc[8][4];
#pragma unroll
    for (int m = 0; m < 8; ++m)
#pragma unroll
        for (int n = 0; n < 4; ++n) acc[m][n] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const bool lnc = (EPI == EPI_BF16 || EPI == EPI_BF16_GELU || EPI == EPI_BF16_ROPE) && p.lnc_stats != nullptr;
    float2* lds_mr = (float2*)(smem + G256_LDS + LNC_RAW_BYTES);
    if (!(DBG & 2)) {
        if (lnc) lnc_issue_stats(p, m0, smem + G256_LDS, wave, lane);
        g256_issue_prologue(A, B, smem, p.K, wave);
        if (lnc) lnc_merge_rows(p, smem + G256_LDS, lds_mr, wave, lane, p.K <= 64, p.lnc_tele != nullptr && n0 == 0, p.M - m0);
        gemm256_mainloop(A, B, smem, p.K, wave, lane, acc);
    }
    if (DBG & 1) {
        if (acc[0][0][0] != 12345.678f) return;     // timing-only build: keep acc live, store nothing
    }
    if constexpr (EPI == EPI_PATCH) {
        gemm_epilogue<EPI, 8, 4>(p, m0 + (wave >> 2) * 128, n0 + (wave & 3) * 64, lane, acc);
    } else {
        // bf16 rows are written in 16-byte column chunks: needs N % 8 == 0 and ldc % 8 == 0 (else direct stores)
        // (a runtime condition for every variant: with the direct epilogue compiled out, hipcc 7.2
        //  allocates the fp32 variants' main loop so badly that the accumulators spill)
        const bool wide = (p.N & 7) == 0 && (p.ldc & 7) == 0;
        if (wide) gemm256_epilogue<EPI>(p, smem + wave * 16384, m0 + (wave >> 2) * 128, n0 + (wave & 3) * 64, lane, acc, nullptr,
                                        lnc ? lds_mr + (wave >> 2) * 128 : nullptr);
        else gemm_epilogue<EPI, 8, 4>(p, m0 + (wave >> 2) * 128, n0 + (wave & 3) * 64, lane, acc);
    }
}


// Persistent form of the 256 x 256 kernel: one workgroup per CU walks its share of the XCD's tile
// region.  The DMA for the next tile's first K-tile is issued BEFORE the epilogue of the current
// one, so the pipeline fill (and the launch of a fresh workgroup) no longer sits between a tile's
// last store and the next tile's first MFMA.  LDS while an epilogue runs: A stage 0 and B stage 0
// are being filled; the wave-private transpose slabs live in A stage 1 (waves 0-2) and in B stage
// 1 plus the 32 KiB above the main-loop image (waves 3-7).
// BMR = 192: 192-row tiles (gemm256_mainloop<192>), chosen by the launcher where they make whole rounds.
constexpr int G256P_LDS = 163840;
template <int EPI, int BMR = 256, int XP = 0>
__global__ __launch_bounds__(G256_THREADS, 2) void gemm256p_kernel(GemmArgs p, int nslot) {
    static_assert(BMR == 256 || BMR == 192, "");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tiles_n = (p.N + 255) / 256;
    // 192-row form: p.t192_tiles tile rows, the last p.t192_tall of them 208 rows tall (set by the launcher)
    const int tiles_m = BMR == 256 ? (p.M + 255) / 256 : p.t192_tiles;
    const int tall0 = tiles_m - p.t192_tall;
    const int gy = p.gy, gx = 8 / gy;
    const int xcd = blockIdx.x & 7;
    const int xi = __builtin_amdgcn_readfirstlane(xcd / gy), xj = xcd - xi * gy;
    const int pm = __builtin_amdgcn_readfirstlane((tiles_m + gx - 1) / gx), pn = __builtin_amdgcn_readfirstlane((tiles_n + gy - 1) / gy);
    const int m_lo = xi * pm, n_lo = xj * pn;
    const int m_cnt = (tiles_m - m_lo) < pm ? (tiles_m - m_lo) : pm;
    const int n_cnt = (tiles_n - n_lo) < pn ? (tiles_n - n_lo) : pn;
    const int total = (m_cnt > 0 && n_cnt > 0) ? m_cnt * n_cnt : 0;
    int slot = blockIdx.x >> 3;
    if (slot >= total) return;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    char* slab = smem + (wave < 3 ? 32768 + wave * 9216 : 98304 + (wave - 3) * 9216);
#ifdef REVO_EXPERIMENTS
    if (p.stagger_cycles > 0) {
        // timing experiment: phase groups.  Workgroup (slot % groups) starts (that / groups) of a tile time late.
        const int grp = (blockIdx.x >> 3) % p.stagger_groups;
        const long t0 = (long)__builtin_amdgcn_s_memrealtime();     // constant 100 MHz clock
        const long wait = (long)p.stagger_cycles * grp / p.stagger_groups;
        while ((long)__builtin_amdgcn_s_memrealtime() - t0 < wait) __builtin_amdgcn_s_sleep(8);
    }
#endif

    // row origin / height of tile row mi
    auto tile_row0 = [&](int mi) { return BMR == 256 ? mi * 256 : mi * 192 + (mi > tall0 ? (mi - tall0) * 16 : 0); };
    // (integer divisions are vector-unit sequences: their wave-uniform results are moved to scalar registers at once)
    int sq = __builtin_amdgcn_readfirstlane(slot / n_cnt);
    int mi = m_lo + sq;
    int m0 = tile_row0(mi), n0 = (n_lo + (slot - sq * n_cnt)) * 256;
    bool tall = BMR != 256 && mi >= tall0;
    // folded LayerNorm (kernels.h): consumer side of the bf16 epilogues, producer side of the residual epilogue
    const bool lnc = (EPI == EPI_BF16 || EPI == EPI_BF16_GELU || EPI == EPI_BF16_ROPE) && p.lnc_stats != nullptr;
    const bool lnf = EPI == EPI_RESID_F32 && p.lnf_stats != nullptr;
    char* lds_raw = smem + G256P_LN_OFF;
    float2* lds_mr = (float2*)(smem + G256P_LN_OFF + LNC_RAW_BYTES);
    float2* lds_st = (float2*)(smem + G256P_LN_OFF);
    if (lnc) lnc_issue_stats(p, m0, lds_raw, wave, lane);
    G256Operand A, B;
    g256_operand_init(A, p.A, p.lda, p.M, m0, wave, lane, BMR == 256 ? 256 : (tall ? 208 : 192));
    g256_operand_init(B, p.B, p.ldb, p.N, n0, wave, lane);
    g256_issue_half(A, 0, 0, G256_A(smem, 0), wave);
    g256_issue_half(A, 1, 0, G256_A(smem, 0), wave);
    g256_issue_half(B, 0, 0, G256_B(smem, 0), wave);
    g256_issue_half(B, 1, 0, G256_B(smem, 0), wave);
    for (;;) {
        if (p.K > 64) {
            g256_issue_half(A, 0, 128, G256_A(smem, 1), wave);
            g256_issue_half(A, 1, 128, G256_A(smem, 1), wave);
        }
        // this tile's rows: (rstd, -mean rstd) into LDS while its second K-tile is on its way (no barrier of its own)
        if (lnc && !(REVO_LNC_ABLATE & 1)) lnc_merge_rows(p, lds_raw, lds_mr, wave, lane, p.K <= 64, p.lnc_tele != nullptr && n0 == 0, p.M - m0);
        f32x4 acc[8][4];
#pragma unroll
        for (int m = 0; m < 8; ++m)
#pragma unroll
            for (int n = 0; n < 4; ++n) acc[m][n] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if constexpr (BMR == 256) gemm256_mainloop(A, B, smem, p.K, wave, lane, acc);
        else gemm256_mainloop<192>(A, B, smem, p.K, wave, lane, acc, tall);

        // the epilogue's lane id comes from the hardware, here: `lane` kept alive across the main loop is parked in scratch by
        // hipcc and reloaded at the head of every epilogue, behind a wait for all vector memory
        // (not in the plain bf16 and RoPE forms: there the same change made hipcc park four accumulator registers across the
        //  main loop's last K-tile instead)
        int lane_e = lane;
        if constexpr (EPI != EPI_BF16 && EPI != EPI_BF16_ROPE)
            asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane_e));
        const int mb = m0 + (wave >> 2) * 128, nb = n0 + (wave & 3) * 64;
        GemmArgs pe = p;                                   // the epilogue's row limit: the end of this tile
        if (BMR != 256) {
            const int mend = m0 + (tall ? 208 : 192);
            pe.M = mend < p.M ? mend : p.M;
        }
        const int m0_done = m0, tn_done = n0 >> 8;
        slot += nslot;
        const bool more = slot < total;
        if (more) {
            sq = __builtin_amdgcn_readfirstlane(slot / n_cnt);
            mi = m_lo + sq;
            m0 = tile_row0(mi);
            n0 = (n_lo + (slot - sq * n_cnt)) * 256;
            tall = BMR != 256 && mi >= tall0;
            // the next tile's row statistics travel with its first K-tile: the raw slots are free again (merged before this
            // tile's main loop, whose barriers lie in between) and the DMA has the whole epilogue to land
            if (lnc && !(REVO_LNC_ABLATE & 4)) lnc_issue_stats(p, m0, lds_raw, wave, lane);
            g256_operand_init(A, p.A, p.lda, p.M, m0, wave, lane, BMR == 256 ? 256 : (tall ? 208 : 192));
            g256_operand_init(B, p.B, p.ldb, p.N, n0, wave, lane);
            g256_issue_half(A, 0, 0, G256_A(smem, 0), wave);
            g256_issue_half(A, 1, 0, G256_A(smem, 0), wave);
            g256_issue_half(B, 0, 0, G256_B(smem, 0), wave);
            g256_issue_half(B, 1, 0, G256_B(smem, 0), wave);
        }
        if constexpr (EPI == EPI_PATCH) {
            gemm_epilogue<EPI, 8, 4>(pe, mb, nb, lane_e, acc);
        } else {
            const bool wide = (p.N & 7) == 0 && (p.ldc & 7) == 0;
            if (wide) gemm256_epilogue<EPI, BMR != 256, XP>(pe, slab, mb, nb, lane_e, acc,
                                                            lnf ? lds_st + (wave >> 2) * 128 * 4 + (wave & 3) : nullptr,
                                                            lnc ? lds_mr + (wave >> 2) * 128 : nullptr);
            else gemm_epilogue<EPI, 8, 4>(pe, mb, nb, lane_e, acc);
        }
        if (lnf) {
            // the four waves' 64-column partials of every row of this tile are in LDS: one slot per row and column tile
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();                  // (also: every wave is out of its slab)
            lnf_store_tile_stats(pe, lds_st, m0_done, pe.M, tn_done, wave, lane_e);
            if (!more) break;
            continue;                                      // the main loop's barriers separate these reads from the next tile's writes
        }
        if (!more) break;
        __builtin_amdgcn_s_barrier();      // every wave is out of its slab before A stage 1 is refilled
    }
}


// ---- queued-stores kernel: the rows the tiles leave over
template <int EPI>
__device__ __forceinline__ void gemm256q_tail(char* smem, int wave) {
    const __attribute__((address_space(4))) GemmArgs* pk0 = (const __attribute__((address_space(4))) GemmArgs*)__builtin_amdgcn_kernarg_segment_ptr();
    if (pk0->qtail_rows <= 0) return;
    const bool lnc = pk0->lnc_stats != nullptr;
    // ---- The rows the tiles leave over: rows [M, M + qtail_rows), at most 64 (fc1 of PE-L14 at batch 64: 36 928 = 144 x 256 + 64).
    // As a launch of their own (gemm_skinny_kernel) they were 13 us + a launch boundary behind every fc1: 0.54 GFLOP, i.e. nothing
    // but latency, with the whole chip waiting.  Here every workgroup, behind its last tile and while others still work on
    // theirs, takes 16-column strips: eight waves split K, each with all its loads in flight at once (the leftover rows of A are
    // 128 KB that every workgroup reads: L2), four 16 x 16 accumulators per wave, the eight partial sums added through LDS in a
    // fixed order, then the tile epilogue's arithmetic (folded LayerNorm from the rows' statistics in memory, bias, GELU).
    {
        const __attribute__((address_space(4))) GemmArgs& q = *pk0;
        int lane_t;
        asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane_t));
        const int li = lane_t & 15, lq = lane_t >> 4;
        const int tail = q.qtail_rows, Kw = q.K >> 3;                    // K columns per wave (a multiple of 32: launcher)
        const __amdgpu_buffer_rsrc_t ars = __builtin_amdgcn_make_buffer_rsrc((void*)(q.A + (long)q.M * q.lda), 0, (int)(tail * q.lda * 2), 0x00020000);
        float* part = (float*)smem;                                       // [8 waves][64 rows][16 columns] fp32 partial sums: 32 KiB
        for (int strip = blockIdx.x; strip * 16 < q.N; strip += gridDim.x) {
            __builtin_amdgcn_s_barrier();                                 // (the image is free: every wave is out of its last main loop / the previous strip's sums)
            f32x4 acc4[4];
#pragma unroll
            for (int m = 0; m < 4; ++m) acc4[m] = (f32x4){0.f, 0.f, 0.f, 0.f};
            const bf16_t* wrow = q.B + (long)(strip * 16 + li) * q.ldb + wave * Kw + lq * 8;
            const int aoff = (li * (int)q.lda + wave * Kw + lq * 8) * 2;
            for (int k0 = 0; k0 < Kw; k0 += 128) {                        // four 32-wide steps at a time: 20 loads per lane in flight
                bf16x8 wf[4], af[4][4];
#pragma unroll
                for (int s4 = 0; s4 < 4; ++s4) {
                    if (k0 + s4 * 32 < Kw) {
                        wf[s4] = *(const bf16x8*)(wrow + k0 + s4 * 32);
#pragma unroll
                        for (int m = 0; m < 4; ++m)
                            af[s4][m] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(ars, aoff + (m * 16 * (int)q.lda + k0 + s4 * 32) * 2, 0, 0));
                    }
                }
#pragma unroll
                for (int s4 = 0; s4 < 4; ++s4)
                    if (k0 + s4 * 32 < Kw) {
#pragma unroll
                        for (int m = 0; m < 4; ++m) acc4[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[s4], af[s4][m], acc4[m], 0, 0, 0);
                    }
            }
#pragma unroll
            for (int m = 0; m < 4; ++m) *(f32x4*)(part + (wave * 64 + m * 16 + li) * 16 + lq * 4) = acc4[m];
            __builtin_amdgcn_s_barrier();
            if (wave < 4) {
                const int tt = wave * 64 + lane_t;
                const int row = tt >> 2, cq = tt & 3;
                f32x4 v = *(const f32x4*)(part + row * 16 + cq * 4);
#pragma unroll
                for (int w = 1; w < 8; ++w) v += *(const f32x4*)(part + (w * 64 + row) * 16 + cq * 4);
                if (row < tail) {
                    const int col = strip * 16 + cq * 4;
                    const long grow = (long)q.M + row;
                    const f32x4 b4 = q.bias ? *(const f32x4*)(q.bias + col) : (f32x4){0.f, 0.f, 0.f, 0.f};
                    if (lnc) {
                        float rstd, mr;
                        lnf_merge(q.lnc_stats + grow * q.lnc_parts, q.lnc_parts, q.lnc_eps, rstd, mr);
                        if (q.lnc_tele && strip == 0 && cq == 0) {        // the fold's telemetry (revo_vit_stats): these rows once, by strip 0
                            unsigned long long* tp = q.lnc_tele;
                            asm volatile("" : "+v"(tp));
                            const float am = __builtin_fabsf(mr);
                            atomicAdd(tp, 1ull);
                            if (am > LNC_TELE_RATIO) atomicAdd(tp + 1, 1ull);
                            if (am > 4.0f * LNC_TELE_RATIO) atomicAdd(tp + 2, 1ull);
                        }
                        const f32x4 c = *(const f32x4*)(q.lnc_c + col);
                        v = v * rstd + (c * mr + b4);                     // rstd (acc - mean c) + b', the tile epilogue's operation order
                    } else {
                        v += b4;
                    }
                    if (EPI == EPI_BF16_GELU) v = gelu_erf4(v);
                    uint2 o;
                    o.x = pack_bf16x2(v[0], v[1]);
                    o.y = pack_bf16x2(v[2], v[3]);
                    *(uint2*)((bf16_t*)q.C + grow * q.ldc + col) = o;
                }
            }
        }
    }
}

// ---- Persistent 256 x 256 kernel with QUEUED STORES (bf16 epilogues: plain, GELU, RoPE; 256-row tiles, N % 256 == 0, K >= 128).
// gemm256p_kernel's phases run in series on every CU at the same moments: a tile's stores are pushed out (3.4 us at the
// per-CU store rate), and the next main loop's first wait for an operand requested behind them also waits for every one
// of them to be acknowledged (loads, DMA and stores retire in order: ~1.8 us of drain inside qkv's main loop,
// profiles/r05_gemm_phase_groups.json).  Here
//   * the epilogue never touches the operand image: a tile's values go from the accumulators to 16-byte row chunks in
//     REGISTERS (bias / folded LayerNorm / GELU / RoPE on the accumulator layout, v_cvt_pk_bf16_f32, then two
//     v_permlane16_swap_b32 per fragment pair: lanes l and l ^ 16 hold columns c..c+3 and c+4..c+7 of one row and trade
//     halves, so that each ends up with 8 consecutive columns of one fragment) -- no LDS slab, no transposing reads;
//   * so the next tile's first K-tile and a half (both operands of K-tile 0, A of K-tile 1) are requested right behind the
//     main loop's last barrier, before any epilogue arithmetic, and have the whole epilogue to land;
//   * the tile's 16 stores per wave are issued back to back at the very end, as bounds-checked buffer stores (rows past
//     the matrix edge are dropped by the descriptor: the instruction count never depends on the data), and
//     gemm256_mainloop<..., QS = 16> enters on a counted wait that leaves them in flight through its first K-tile.
// Queue of one wave, oldest first, when the main loop starts:  [K-tile 0: 8] [A of K-tile 1: 4] [stores: 16].  The first
// tile of a workgroup issues 16 dropped stores (an empty descriptor) so that every tile meets the same counts.  LDS: the operand image, and above it the folded LayerNorm's statistics as in gemm256_kernel.
// Per element the arithmetic is gemm256_epilogue's (same operations, same order: bias or rstd * acc + (c * mr + b'),
// GELU, bf16 rounding BEFORE the rotation); only the route to memory differs -- 64-byte row segments per store
// instruction instead of 128-byte ones, two adjacent instructions completing each 128-byte line.
constexpr int G256Q_STORES = 16;
template <int EPI>
__global__ __launch_bounds__(G256_THREADS, 2) void gemm256q_kernel(GemmArgs p, int nslot) {
    static_assert(EPI == EPI_BF16 || EPI == EPI_BF16_GELU || EPI == EPI_BF16_ROPE, "");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tiles_n = p.N >> 8;
    const int tiles_m = (p.M + 255) >> 8;
    const int gy = p.gy, gx = 8 / gy;
    const int xcd = blockIdx.x & 7;
    const int xi = __builtin_amdgcn_readfirstlane(xcd / gy), xj = xcd - xi * gy;
    const int pm = __builtin_amdgcn_readfirstlane((tiles_m + gx - 1) / gx), pn = __builtin_amdgcn_readfirstlane((tiles_n + gy - 1) / gy);
    const int m_lo = xi * pm, n_lo = xj * pn;
    const int m_cnt = (tiles_m - m_lo) < pm ? (tiles_m - m_lo) : pm;
    const int n_cnt = (tiles_n - n_lo) < pn ? (tiles_n - n_lo) : pn;
    const int total = (m_cnt > 0 && n_cnt > 0) ? m_cnt * n_cnt : 0;
    int slot = blockIdx.x >> 3;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (slot >= total) {                  // (an XCD region with fewer tiles than workgroups: its share of the leftover rows is still due)
        gemm256q_tail<EPI>(smem, wave);
        return;
    }
    const int lane = threadIdx.x & 63;

    int sq = __builtin_amdgcn_readfirstlane(slot / n_cnt);
    int m0 = (m_lo + sq) * 256, n0 = (n_lo + (slot - sq * n_cnt)) * 256;
    const bool lnc = p.lnc_stats != nullptr;
    char* lds_raw = smem + G256_LDS;
    // (rstd, -mean rstd) of a tile's rows, two buffers used in turn: the merge for tile i + 1 (behind tile i's stores) may
    // overtake another wave's reads for tile i (at the head of its epilogue) -- no barrier between a tile's stores and the
    // next main loop
    float2* lds_mr = (float2*)(smem + G256_LDS + LNC_RAW_BYTES);
    float2* lds_mr_next = lds_mr + 256;
    // bias and column sums of a tile's 256 columns travel to LDS with the tile's first operands (one 1-KiB DMA each, waves 4
    // and 5; an EMPTY descriptor -- the vector is absent -- delivers zeros): the epilogue then starts without a load to wait
    // for.  Two buffers in turn, like lds_mr: a wave may request tile i + 1's while another still reads tile i's.
    char* lds_bc = smem + G256Q_BC_OFF;
    char* lds_bc_next = lds_bc + 2048;
    auto issue_bias_csum = [&](const float* bias, const float* csum, int N, int n0_, char* dst, int lane_) {
        if (wave == 4 || wave == 5) {                       // wave-uniform
            const float* src = wave == 4 ? bias : csum;
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(src + n0_), 0, src ? (N - n0_ < 256 ? N - n0_ : 256) * 4 : 0, 0x00020000);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(dst + (wave - 4) * 1024), 16,
                                                     (uint32_t)(lane_ * 16), 0, 0, 0);
        }
    };
    if (lnc) lnc_issue_stats(p, m0, lds_raw, wave, lane);
    issue_bias_csum(p.bias, lnc ? p.lnc_c : nullptr, p.N, n0, lds_bc, lane);
    G256Operand A, B;
    g256_operand_init(A, p.A, p.lda, p.M, m0, wave, lane);
    g256_operand_init(B, p.B, p.ldb, p.N, n0, wave, lane);
    g256_issue_prologue(A, B, smem, p.K, wave);
    G256_FENCE();
    {   // the first tile's stand-ins for a previous tile's stores: an empty descriptor drops them
        const __amdgpu_buffer_rsrc_t none = __builtin_amdgcn_make_buffer_rsrc(p.C, 0, 0, 0x00020000);
        typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#pragma unroll
        for (int i = 0; i < G256Q_STORES; ++i)          // (distinct offsets: identical stores would be merged into one)
            __builtin_amdgcn_raw_buffer_store_b128((u32x4){0u, 0u, 0u, 0u}, none, i * 16, 0, 0);
    }
    G256_FENCE();
#ifdef REVO_EXPERIMENTS
    int tiles_done = 0;
#endif
    // (rstd, -mean rstd) of the first tile's rows; the wait leaves the 16 youngest instructions in flight, here the stand-ins.
    // Without a folded LayerNorm every row gets (1, 0): the epilogue below has ONE form, rstd * acc + (c * mr + b) with c = 0
    // -- the bits of acc + b -- instead of two forms joined by selects whose constant sides (zero vectors) hipcc kept in a
    // dozen registers across the main loop.
    if (lnc) lnc_merge_rows<G256Q_STORES>(p, lds_raw, lds_mr, wave, lane, false, p.lnc_tele != nullptr && n0 == 0, p.M - m0);
    else if (threadIdx.x < 512) lds_mr[threadIdx.x] = make_float2(1.f, 0.f);        // both buffers (read behind the main loop's barriers)
    for (;;) {
        f32x4 acc[8][4];
#pragma unroll
        for (int m = 0; m < 8; ++m)
#pragma unroll
            for (int n = 0; n < 4; ++n) acc[m][n] = (f32x4){0.f, 0.f, 0.f, 0.f};
#ifdef REVO_EXPERIMENTS
        // diagnostic stamps (scripts/gemm_qstamps.py): 100 MHz clock at main loop begin / end, epilogue arithmetic done, stores issued
        unsigned long long t_a = 0, t_b = 0, t_c = 0, c_a = 0;
        if (p.stamps) { t_a = __builtin_amdgcn_s_memrealtime(); c_a = __builtin_amdgcn_s_memtime(); }
#endif
        gemm256_mainloop<0, false, false, G256Q_STORES>(A, B, smem, p.K, wave, lane, acc);
#ifdef REVO_EXPERIMENTS
        if (p.stamps) { t_b = __builtin_amdgcn_s_memrealtime(); c_a = __builtin_amdgcn_s_memtime() - c_a; }      // shader-clock cycles of the main loop
#endif

        // ------------------------------------------------------------------ epilogue, in registers
        // The kernel's arguments are read again from the kernarg segment here (through a pointer the compiler cannot see
        // through): kept in scalar registers across the main loop, the pointers and sizes the epilogue and the next tile's
        // set-up need -- some forty registers on top of the two operands' eight descriptors -- overflowed the scalar file
        // into vector-register lanes, and the main loop, which runs at the 256-VGPR limit, spilled accumulators for them.
        const __attribute__((address_space(4))) GemmArgs* pk = (const __attribute__((address_space(4))) GemmArgs*)__builtin_amdgcn_kernarg_segment_ptr();
        asm volatile("" : "+s"(pk));
        const __attribute__((address_space(4))) GemmArgs& q = *pk;
        int lane_e;
        asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane_e));
        const int lr = lane_e & 15, lq = lane_e >> 4;
        const int wr = wave >> 2, wc = wave & 3;
        const int m0_done = m0, n0_done = n0;
        const int nb = n0_done + wc * 64;                           // this wave's first column
        slot += nslot;
        const bool more = slot < total;
        // (0) (rstd, -mean rstd) of this lane's eight fragment rows, out of LDS before ANY vector-memory request of the
        //     epilogue: hipcc guards an LDS read with s_waitcnt vmcnt(0) while DMA it knows of may be outstanding (it cannot
        //     tell the operand image from the statistics, nor see the main loop's counted waits) -- here that wait is free
        float2 st8[8];
#pragma unroll
        for (int m = 0; m < 8; ++m) st8[m] = lds_mr[wr * 128 + m * 16 + lr];
        //     ... and the tile's bias and column sums (requested with its first operands)
        f32x4 bias4[4], c4[4];
#pragma unroll
        for (int n = 0; n < 4; ++n) {
            bias4[n] = *(const f32x4*)(lds_bc + (wc * 64 + n * 16 + lq * 4) * 4);
            c4[n] = *(const f32x4*)(lds_bc + 1024 + (wc * 64 + n * 16 + lq * 4) * 4);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        G256_FENCE();
        // (1) what the RoPE form needs from memory, requested BEFORE the next tile's operands
        // RoPE table: entry (token, pair) as (cos, sin); a lane's fragment holds two pairs of one token: one 16-byte load
        // per fragment, through a descriptor (32-bit offsets).  Token of fragment row m: tok0 + 16 m (mod S).
        const bool rot = EPI == EPI_BF16_ROPE && nb < q.rope_cols;            // wave-uniform (rope_cols % 64 == 0: launcher)
        __amdgpu_buffer_rsrc_t rtab = __builtin_amdgcn_make_buffer_rsrc((void*)q.rope_cs, 0, EPI == EPI_BF16_ROPE ? q.rope_S * (q.rope_hd >> 1) * 8 : 0, 0x00020000);
        int tok = 0, cpair[4] = {0, 0, 0, 0};
        if (EPI == EPI_BF16_ROPE) {
            tok = (m0_done + wr * 128 + lr) % q.rope_S;
            int c0 = (nb + lq * 4) % q.rope_hd;
#pragma unroll
            for (int n = 0; n < 4; ++n) {
                cpair[n] = (c0 >> 1) * 8;                                     // byte offset of the pair inside a token's row
                c0 += 16;
                if (c0 >= q.rope_hd) c0 -= q.rope_hd;
            }
        }
        const int tok_row_bytes = (q.rope_hd >> 1) * 8;
        f32x4 rt[2][2][4];                                                    // [buffer][m of the quarter][fragment]
        auto rope_load = [&](auto bufc) {                                     // the next quarter (two fragment rows) into buffer `buf`
            constexpr int buf = decltype(bufc)::value;
#pragma unroll
            for (int mm = 0; mm < 2; ++mm) {
#pragma unroll
                for (int n = 0; n < 4; ++n)
                    rt[buf][mm][n] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rtab, tok * tok_row_bytes + cpair[n], 0, 0));
                tok += 16;
                if (tok >= q.rope_S) tok -= q.rope_S;
            }
        };
        using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>;
        using I2 = std::integral_constant<int, 2>; using I3 = std::integral_constant<int, 3>;
        if (rot) { rope_load(I0{}); rope_load(I1{}); }
        // ... and waited for HERE, before any DMA is in flight: hipcc's wait counts leave LDS-DMA instructions out, so a
        // wait it places for one of these loads behind the requests below (vmcnt(5), (4), ... (0), as it did) is a wait for
        // the DMA as well.  The empty asm statements are "uses": the waits land in front of them.
        if (rot) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int mm = 0; mm < 2; ++mm)
#pragma unroll
                    for (int n = 0; n < 4; ++n) asm volatile("" : "+v"(rt[i][mm][n]));
        }
        G256_FENCE();
        // (2) the next tile: statistics, then its first K-tile and a half (the image is free: the main loop ended behind a
        //     barrier, and nothing below touches it)
        if (more) {
            sq = __builtin_amdgcn_readfirstlane(slot / n_cnt);
            m0 = (m_lo + sq) * 256;
            n0 = (n_lo + (slot - sq * n_cnt)) * 256;
            if (lnc) lnc_issue_stats(q, m0, lds_raw, wave, lane_e);
            issue_bias_csum(q.bias, lnc ? q.lnc_c : nullptr, q.N, n0, lds_bc_next, lane_e);
            g256_operand_init(A, q.A, q.lda, q.M, m0, wave, lane_e);
            g256_operand_init(B, q.B, q.ldb, q.N, n0, wave, lane_e);
            g256_issue_prologue(A, B, smem, q.K, wave);
        }
        G256_FENCE();
        // (3) accumulators -> bf16 row chunks.  out[m][P]: row m * 16 + lr, 8 columns from (2 P + (lq & 1)) * 16 + (lq >> 1) * 8
        typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
        typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
        // (4) ... and each quarter's four stores right behind its arithmetic: a CU takes stores at ~30 GB/s (128 KB per tile:
        //     4-5 us, profiles/r06_gemm_queued_stores.json), several times longer than the arithmetic in front of them -- issued
        //     as a block of 16 at the end, every wave sat in its store instructions while nothing else ran.  One descriptor
        //     per tile: rows past the matrix edge lie beyond num_records and are dropped, so all 16 instructions are issued
        //     (and counted) whatever the tile holds.
        const long rows_here = (long)q.M - m0_done < 256 ? (long)q.M - m0_done : 256;
#ifdef REVO_EXPERIMENTS
        const bool drop_stores = (q.stagger_groups & 0x100) != 0;          // timing experiment: the stores are issued but move nothing
#else
        constexpr bool drop_stores = false;
#endif
        const __amdgpu_buffer_rsrc_t crs = __builtin_amdgcn_make_buffer_rsrc(
            (void*)((bf16_t*)q.C + (long)m0_done * q.ldc + n0_done), 0, drop_stores ? 0 : (int)((rows_here - 1) * q.ldc * 2 + 512), 0x00020000);
        const int rstep = (int)(q.ldc * 32);                                   // 16 rows, bytes
        int voff = (wr * 128 + lr) * (int)(q.ldc * 2) + (wc * 64 + (lq & 1) * 16 + (lq >> 1) * 8) * 2;
        auto quarter = [&](auto qc, auto bufc) {
            u32x4 out[2][2];
            constexpr int q = decltype(qc)::value, buf = decltype(bufc)::value;
#pragma unroll
            for (int mm = 0; mm < 2; ++mm) {
                constexpr int dummy_ = 0; (void)dummy_;
                const int m = q * 2 + mm;
                const float2 st = st8[m];
                uint32_t d[4][2];
#pragma unroll
                for (int n0f = 0; n0f < 4; n0f += 2) {
                    f32x4 v[2];
#pragma unroll
                    for (int n = 0; n < 2; ++n) v[n] = acc[m][n0f + n] * st.x + (c4[n0f + n] * st.y + bias4[n0f + n]);
                    if (EPI == EPI_BF16_GELU) gelu_erf4xn<2>(v);
#pragma unroll
                    for (int n = 0; n < 2; ++n) {
                        uint32_t w0 = pack_bf16x2(v[n][0], v[n][1]), w1 = pack_bf16x2(v[n][2], v[n][3]);
                        if (EPI == EPI_BF16_ROPE) {
                            if (rot) {                 // K5 on the bf16-rounded values, as every other path rotates them
                                const f32x4 tc = rt[buf][mm][n0f + n];
                                const float a0 = __uint_as_float(w0 << 16), a1 = __uint_as_float(w0 & 0xffff0000u);
                                const float b0 = __uint_as_float(w1 << 16), b1 = __uint_as_float(w1 & 0xffff0000u);
                                w0 = pack_bf16x2(a0 * tc[0] - a1 * tc[1], a1 * tc[0] + a0 * tc[1]);
                                w1 = pack_bf16x2(b0 * tc[2] - b1 * tc[3], b1 * tc[2] + b0 * tc[3]);
                            }
                        }
                        d[n0f + n][0] = w0;
                        d[n0f + n][1] = w1;
                    }
                }
#pragma unroll
                for (int P = 0; P < 2; ++P) {
                    // rows 1 and 3 of the first register trade places with rows 0 and 2 of the second (rows of 16 lanes): an even
                    // row keeps its own half of fragment 2 P and receives its neighbour's; an odd row likewise for 2 P + 1
                    const u32x2 s0 = __builtin_amdgcn_permlane16_swap(d[2 * P][0], d[2 * P + 1][0], false, false);
                    const u32x2 s1 = __builtin_amdgcn_permlane16_swap(d[2 * P][1], d[2 * P + 1][1], false, false);
                    out[mm][P] = (u32x4){s0[0], s1[0], s0[1], s1[1]};
                }
            }
            G256_FENCE();
#pragma unroll
            for (int mm = 0; mm < 2; ++mm) {
                // qkv and the MLP hidden activations are written once and read once by the next kernel: non-temporal
                // (aux = 2), as gemm256_epilogue stores them
                constexpr int aux = (EPI == EPI_BF16_ROPE || EPI == EPI_BF16_GELU) ? 2 : 0;
                __builtin_amdgcn_raw_buffer_store_b128(out[mm][0], crs, voff, 0, aux);
                __builtin_amdgcn_raw_buffer_store_b128(out[mm][1], crs, voff + 64, 0, aux);
                voff += rstep;
            }
            G256_FENCE();
        };
#ifdef REVO_EXPERIMENTS
        if (q.stamps) t_c = __builtin_amdgcn_s_memrealtime();          // (epilogue set-up done: loads waited for, DMA requested)
#endif
        quarter(I0{}, I0{});
        if (rot) rope_load(I0{});
        quarter(I1{}, I1{});
        if (rot) rope_load(I1{});
        quarter(I2{}, I0{});
        quarter(I3{}, I1{});
        G256_FENCE();
#ifdef REVO_EXPERIMENTS
        if (q.stamps && wave == 0 && lane_e == 0 && tiles_done < q.stamp_items) {
            unsigned long long* dst = q.stamps + ((size_t)blockIdx.x * q.stamp_items + tiles_done) * 4;
            dst[0] = t_a; dst[1] = t_b; dst[2] = (q.stagger_groups & 0x200) ? c_a : t_c; dst[3] = __builtin_amdgcn_s_memrealtime();
        }
        ++tiles_done;
#endif
        if (!more) break;
        if (lnc) {
            // the next tile's rows, into the other buffer: the wait leaves the 16 youngest instructions -- the stores -- in
            // flight; the statistics pieces are the oldest
            lnc_merge_rows<G256Q_STORES>(q, lds_raw, lds_mr_next, wave, lane_e, false, q.lnc_tele != nullptr && n0 == 0, q.M - m0);
        }
        { float2* t = lds_mr; lds_mr = lds_mr_next; lds_mr_next = t; }
        { char* t = lds_bc; lds_bc = lds_bc_next; lds_bc_next = t; }
    }
    gemm256q_tail<EPI>(smem, wave);
}

#ifdef REVO_EXPERIMENTS
// Phased form of the persistent kernel -- EXPERIMENT LIBRARY ONLY: measured in round 5 and not adopted (DESIGN_HISTORY.md,
// profiles/r05_gemm_phase_groups.json: bit-identical, 5-12 % slower on three of the four body GEMMs, +-0 on out-proj).
// All 256 workgroups of gemm256p_kernel run their main loops and their
// epilogues at the same moments: the epilogues' HBM traffic (stores of bf16 tiles, read-modify-write of the fp32
// residual stream) is paid while no MFMA runs, and the main loops run while HBM idles ("phases in series").
// Starting some workgroups late de-phases them, but costs exactly the delay.  Here the shift is free: a workgroup of
// phase group g does the first h_g rows of its FIRST tile at the start and the remaining rows of that tile at the very
// end (h_g = 64, 128, 192, 256 for four groups), whole tiles in between.  Every group does the same total work; through
// all middle rounds the groups sit a quarter of a tile apart, so at any moment only a fraction of the CUs is in its
// epilogue.  Tile -> workgroup mapping and every row's arithmetic are those of gemm256p_kernel (a row's result does
// not depend on which piece computed it: same K order, same epilogue), so the output is bit-identical.
// The piece heights are run-time, wave-uniform row modes of ONE main-loop instantiation (gemm256_mainloop<0, true>).
template <int EPI, int BMR = 256>
__global__ __launch_bounds__(G256_THREADS, 2) void gemm256pp_kernel(GemmArgs p, int nslot, int groups) {
    static_assert(BMR == 256 || BMR == 192, "");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tiles_n = (p.N + 255) / 256;
    const int tiles_m = BMR == 256 ? (p.M + 255) / 256 : p.t192_tiles;
    const int tall0 = tiles_m - (BMR == 256 ? 0 : p.t192_tall);
    const int gy = p.gy, gx = 8 / gy;
    const int xcd = blockIdx.x & 7;
    const int xi = xcd / gy, xj = xcd - xi * gy;
    const int pm = (tiles_m + gx - 1) / gx, pn = (tiles_n + gy - 1) / gy;
    const int m_lo = xi * pm, n_lo = xj * pn;
    const int m_cnt = (tiles_m - m_lo) < pm ? (tiles_m - m_lo) : pm;
    const int n_cnt = (tiles_n - n_lo) < pn ? (tiles_n - n_lo) : pn;
    const int total = (m_cnt > 0 && n_cnt > 0) ? m_cnt * n_cnt : 0;
    const int slot0 = blockIdx.x >> 3;
    if (slot0 >= total) return;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const int wr = wave >> 2;
    char* slab = smem + (wave < 3 ? 32768 + wave * 9216 : 98304 + (wave - 3) * 9216);

    auto tile_row0 = [&](int mi) { return BMR == 256 ? mi * 256 : mi * 192 + (mi > tall0 ? (mi - tall0) * 16 : 0); };
    // this workgroup's tiles: slots slot0, slot0 + nslot, ...; phase group by the tile ROW of the first one (the
    // workgroups that share an A stripe stay in step and keep meeting in L2)
    const int ntiles = (total - slot0 + nslot - 1) / nslot;
    const int grp = (slot0 / n_cnt) % groups;
    const int h = 64 * (((grp + 1) * (BMR / 64) + groups - 1) / groups);        // 64 .. BMR
    const bool first_tall = BMR != 256 && (m_lo + slot0 / n_cnt) >= tall0;
    const bool split = h < BMR && ntiles >= 2 && !first_tall;
    const int nitems = ntiles + (split ? 1 : 0);

    int m0, n0, rows;
    auto item = [&](int it) {
        const bool last_piece = split && it == nitems - 1;
        const int s = last_piece ? slot0 : slot0 + it * nslot;
        const int mi = m_lo + s / n_cnt;
        n0 = (n_lo + s % n_cnt) * 256;
        m0 = tile_row0(mi);
        rows = BMR == 256 ? 256 : (mi >= tall0 ? 208 : 192);
        if (split && it == 0) rows = h;
        if (last_piece) { m0 += h; rows = BMR - h; }
    };
    item(0);
    G256Operand A, B;
    g256_operand_init(A, p.A, p.lda, p.M, m0, wave, lane, rows);
    g256_operand_init(B, p.B, p.ldb, p.N, n0, wave, lane);
    g256_issue_half(A, 0, 0, G256_A(smem, 0), wave);
    g256_issue_half(A, 1, 0, G256_A(smem, 0), wave);
    g256_issue_half(B, 0, 0, G256_B(smem, 0), wave);
    g256_issue_half(B, 1, 0, G256_B(smem, 0), wave);
    for (int it = 0;;) {
        if (p.K > 64) {
            g256_issue_half(A, 0, 128, G256_A(smem, 1), wave);
            g256_issue_half(A, 1, 128, G256_A(smem, 1), wave);
        }
        f32x4 acc[8][4];
#pragma unroll
        for (int m = 0; m < 8; ++m)
#pragma unroll
            for (int n = 0; n < 4; ++n) acc[m][n] = (f32x4){0.f, 0.f, 0.f, 0.f};
        // row mode of this piece: which 64-row halves of its 128 rows this wave computes
        const bool do_lo = wr == 0 ? true : rows > 128;
        const bool do_hi = wr == 0 ? rows > 64 : rows >= 256;
        const bool do_ht = wr == 1 && rows == 208;
#ifdef REVO_EXPERIMENTS
        unsigned long long t_a = 0, t_b = 0;
        if (p.stamps) t_a = __builtin_amdgcn_s_memrealtime();
#endif
        gemm256_mainloop<0, true>(A, B, smem, p.K, wave, lane, acc, do_ht, do_lo, do_hi);
#ifdef REVO_EXPERIMENTS
        if (p.stamps) t_b = __builtin_amdgcn_s_memrealtime();
#endif

        const int mb = m0 + wr * 128, nb = n0 + (wave & 3) * 64;
        GemmArgs pe = p;                                   // the epilogue's row limit: the end of this piece
        {
            const int mend = m0 + rows;
            pe.M = mend < p.M ? mend : p.M;
        }
#ifdef REVO_EXPERIMENTS
        const int it_done = it, rows_done = rows;
#endif
        ++it;
        const bool more = it < nitems;
        if (more) {
            item(it);
            g256_operand_init(A, p.A, p.lda, p.M, m0, wave, lane, rows);
            g256_operand_init(B, p.B, p.ldb, p.N, n0, wave, lane);
            g256_issue_half(A, 0, 0, G256_A(smem, 0), wave);
            g256_issue_half(A, 1, 0, G256_A(smem, 0), wave);
            g256_issue_half(B, 0, 0, G256_B(smem, 0), wave);
            g256_issue_half(B, 1, 0, G256_B(smem, 0), wave);
        }
        if constexpr (EPI == EPI_PATCH) {
            gemm_epilogue<EPI, 8, 4>(pe, mb, nb, lane, acc);
        } else {
            const bool wide = (p.N & 7) == 0 && (p.ldc & 7) == 0;
            if (wide) gemm256_epilogue<EPI, true>(pe, slab, mb, nb, lane, acc);
            else gemm_epilogue<EPI, 8, 4>(pe, mb, nb, lane, acc);
        }
#ifdef REVO_EXPERIMENTS
        if (p.stamps && wave == 0 && lane == 0 && it_done < p.stamp_items) {
            // diagnostic build only: when this workgroup's main loop began / ended and when its epilogue had issued its last store
            unsigned long long* dst = p.stamps + ((size_t)blockIdx.x * p.stamp_items + it_done) * 4;
            dst[0] = t_a; dst[1] = t_b; dst[2] = __builtin_amdgcn_s_memrealtime(); dst[3] = (unsigned long long)rows_done;
        }
#endif
        if (!more) break;
        __builtin_amdgcn_s_barrier();      // every wave is out of its slab before A stage 1 is refilled
    }
}
#endif   // REVO_EXPERIMENTS (phased persistent kernel)

// Skinny GEMM for M <= 64 (the attention-pool head at batch <= 64: four layers whose time is the
// streaming of 2-8 MB of weights).  The tiled kernels give such a problem N/128 workgroups and a
// serial K loop; here a workgroup owns 16 output columns, its NW (4 or 16) waves split K, operands go
// global -> registers in MFMA fragment layout with eight k-steps of loads in flight, and the
// partial accumulators meet in LDS.
// MY = 1: a workgroup covers all (up to 64) rows, four accumulator fragments per wave.  MY = 4: blockIdx.y picks one
// 16-row fragment -- four times the workgroups, each reading a quarter of A: the leftover rows of a batch-64 layer (64 x
// 1024..4096 x 1024..4096) are bound by what ONE CU can pull through its L2 port (every workgroup of the MY = 1 form
// reads all of A: 640 KB at K = 4096, 20 us), not by the chip.
template <int EPI, int NW, int DEPTH, int MY>
__global__ __launch_bounds__(NW * 64) void gemm_skinny_kernel(GemmArgs p) {
    constexpr int MFR = MY == 1 ? 4 : 1;
    __shared__ f32x4 red[NW - 1][MFR][64];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int lr = lane & 15, lq = lane >> 4;
    const int n0 = blockIdx.x * 16;
    const int m0 = MY == 1 ? 0 : (int)blockIdx.y * 16;
    const int kw = p.K / NW;                       // this wave's K range
    const int nrow = n0 + lr < p.N ? n0 + lr : p.N - 1;
    const bf16_t* bp = p.B + (long)nrow * p.ldb + wave * kw + lq * 8;
    const bf16_t* ap[MFR];
#pragma unroll
    for (int m = 0; m < MFR; ++m) {
        const int row = m0 + m * 16 + lr < p.M ? m0 + m * 16 + lr : p.M - 1;
        ap[m] = p.A + (long)row * p.lda + wave * kw + lq * 8;
    }
    f32x4 acc[MFR][1];
#pragma unroll
    for (int m = 0; m < MFR; ++m) acc[m][0] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // DEPTH x 64 k per trip, every load of a trip requested before its first MFMA: the K loop is a chain of global-load
    // latencies (the launcher checks K % (64 NW DEPTH) == 0)
    for (int k = 0; k < kw; k += 64 * DEPTH) {
        bf16x8 b[2 * DEPTH], a[2 * DEPTH][MFR];
#pragma unroll
        for (int u = 0; u < 2 * DEPTH; ++u) {
            b[u] = *(const bf16x8*)(bp + k + u * 32);
#pragma unroll
            for (int m = 0; m < MFR; ++m) a[u][m] = *(const bf16x8*)(ap[m] + k + u * 32);
        }
#pragma unroll
        for (int u = 0; u < 2 * DEPTH; ++u)
#pragma unroll
            for (int m = 0; m < MFR; ++m)
                acc[m][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[u], a[u][m], acc[m][0], 0, 0, 0);
    }
    if (wave > 0) {
#pragma unroll
        for (int m = 0; m < MFR; ++m) red[wave - 1][m][lane] = acc[m][0];
    }
    __syncthreads();
    if (wave == 0) {
#pragma unroll
        for (int w = 0; w < NW - 1; ++w)
#pragma unroll
            for (int m = 0; m < MFR; ++m) acc[m][0] += red[w][m][lane];
        gemm_epilogue<EPI, MFR, 1>(p, m0, n0, lane, acc);
    }
}

static const char* check_args(const GemmArgs& a) {
    if (a.K <= 0 || a.K % GEMM_BK) return "gemm: K must be a positive multiple of 64";
    if (a.N % 4) return "gemm: N must be a multiple of 4";
    if (a.M <= 0 || a.N <= 0) return "gemm: empty problem";
    if (a.lda % 8 || a.ldb % 8) return "gemm: lda/ldb must be multiples of 8 (16-byte rows)";
    if (a.ldc % 4) return "gemm: ldc must be a multiple of 4";
    if (((uintptr_t)a.A | (uintptr_t)a.B | (uintptr_t)a.C) & 15) return "gemm: operands must be 16-byte aligned";
    return nullptr;
}

// XCD arrangement of the 256 x 256 kernels: N stripes per 8 XCDs (the other factor of 8 = M stripes).  Re-measured in round 6 with
// the current kernels, alternated per layer shape (scripts/gemm_gy_sweep.py, profiles/r06_gemm_gy_sweep.json): up to 7 column
// tiles every XCD sweeps all of N for its M stripe; from 8 on two N stripes (L14 qkv, 12 tiles: -2.9 % against the four
// stripes round 4 chose; fc1, 16: -1.8 %; G14 qkv, 18: -3.4 %); four only from 24 on (G14 fc1, 35 tiles: -1.7 %).
static int xcd_stripes(int tiles_n) { return tiles_n >= 24 ? 4 : (tiles_n >= 8 ? 2 : 1); }
static int g_force_gy = 0;     // timing experiments only: force the XCD arrangement (1, 2, 4, 8)
void gemm_force_gy(int gy) { g_force_gy = gy; }
template <int EPI, int DBG>
static int launch_256d(const GemmArgs& a, hipStream_t st) {
    REVO_FUNC_LDS((gemm256_kernel<EPI, DBG>), G256_LDS_LN);
    // XCD arrangement (measured on MI355X, scripts/gemm_gy.py): with few N tiles every XCD sweeps
    // all of N for its M stripe (gy = 1); from 12 N tiles on, four N stripes keep a weight stripe
    // L2 resident and cut the fabric traffic (qkv 933 -> 1122, fc1 859 -> 920, 8192^3 1322 -> 1453 TF)
    GemmArgs b = a;
    const int tiles_m = (a.M + 255) / 256, tiles_n = (a.N + 255) / 256;
    int gy = xcd_stripes(tiles_n);
    if (g_force_gy) gy = g_force_gy;
    b.gy = gy;
    const int gx = 8 / gy;
    const int region = ((tiles_m + gx - 1) / gx) * ((tiles_n + gy - 1) / gy);
    hipLaunchKernelGGL((gemm256_kernel<EPI, DBG>), dim3(8 * region), dim3(G256_THREADS), G256_LDS_LN, st, b);
    REVO_HIP_CHECK(hipGetLastError());
    return 0;
}

#ifdef REVO_EXPERIMENTS
static int g_stagger_cycles = 0, g_stagger_groups = 2;
void gemm_set_stagger(int cycles, int groups) { g_stagger_cycles = cycles; g_stagger_groups = groups < 1 ? 1 : groups; }
#endif
static int g_persistent = 1;   // timing experiments only: 0 = one workgroup per tile
void gemm_set_persistent(int on) { g_persistent = on; }
#ifdef REVO_EXPERIMENTS
// phase groups of the persistent kernel (gemm256pp_kernel): 0 / 1 = off (gemm256p_kernel), 2..4 forced
static int g_phase_groups = 0;
void gemm_set_phase_groups(int g) { g_phase_groups = g < 0 ? 0 : (g > 4 ? 4 : g); }
static unsigned long long* g_stamps = nullptr; static int g_stamp_items = 0;
void gemm_set_stamps(unsigned long long* buf, int items) { g_stamps = buf; g_stamp_items = items; }
#endif
template <int EPI, int BMR = 256>
static int launch_256p(const GemmArgs& a, hipStream_t st) {
    GemmArgs b = a;
#ifdef REVO_EXPERIMENTS
    b.stagger_cycles = g_stagger_cycles; b.stagger_groups = g_stagger_groups;
    b.stamps = g_stamps; b.stamp_items = g_stamp_items;
#endif
    const int tiles_m = BMR == 256 ? (a.M + 255) / 256 : a.t192_tiles, tiles_n = (a.N + 255) / 256;
    int gy = xcd_stripes(tiles_n);
    if (g_force_gy) gy = g_force_gy;
    b.gy = gy;
    const int gx = 8 / gy;
    const int region = ((tiles_m + gx - 1) / gx) * ((tiles_n + gy - 1) / gy);
    int nslot = region < 32 ? region : 32;                // 32 CUs per XCD
#ifdef REVO_EXPERIMENTS
    // (burst-size study, scripts/experiments/r5_gemm_half_chip.py: fewer persistent workgroups per XCD)
    if (const char* e = getenv("REVO_GEMM_NSLOT")) { const int v = atoi(e); if (v >= 1 && v < nslot) nslot = v; }
    {
        // phase groups (gemm256pp_kernel), forced by scripts/gemm_phase_ab.py only; the stamps live in that kernel too
        // (one group = every workgroup in step)
        int groups = g_phase_groups > 1 ? g_phase_groups : 1;
        if (BMR != 256 && groups > 3) groups = 3;
        if (groups > 1 || b.stamps != nullptr) {
            REVO_FUNC_LDS((gemm256pp_kernel<EPI, BMR>), G256P_LDS);
            hipLaunchKernelGGL((gemm256pp_kernel<EPI, BMR>), dim3(8 * nslot), dim3(G256_THREADS), G256P_LDS, st, b, nslot, groups);
            REVO_HIP_CHECK(hipGetLastError());
            return 0;
        }
    }
#endif
    if constexpr (EPI == EPI_RESID_F32) {
        // the residual stream in planes: one kernel instantiation per format pair (gemm256_epilogue, XP)
        const int xp = (b.xp_in || b.xp_out) ? (b.xp_in ? (b.xp_out ? 2 : 3) : 1) : 0;
        if (xp && ((b.N & 7) || (b.ldc & 7))) { revo_set_error("gemm: planes need the row-coalesced epilogue"); return -2; }
#define REVO_LAUNCH_XP(X)                                                                                              \
    case X:                                                                                                             \
        REVO_FUNC_LDS((gemm256p_kernel<EPI, BMR, X>), G256P_LDS);                                                       \
        hipLaunchKernelGGL((gemm256p_kernel<EPI, BMR, X>), dim3(8 * nslot), dim3(G256_THREADS), G256P_LDS, st, b, nslot); \
        break;
        switch (xp) { REVO_LAUNCH_XP(1) REVO_LAUNCH_XP(2) REVO_LAUNCH_XP(3) default: break; }
#undef REVO_LAUNCH_XP
        if (xp) { REVO_HIP_CHECK(hipGetLastError()); return 0; }
    }
    REVO_FUNC_LDS((gemm256p_kernel<EPI, BMR>), G256P_LDS);
    hipLaunchKernelGGL((gemm256p_kernel<EPI, BMR>), dim3(8 * nslot), dim3(G256_THREADS), G256P_LDS, st, b, nslot);
    REVO_HIP_CHECK(hipGetLastError());
    return 0;
}
#ifdef REVO_EXPERIMENTS
static int g_dbg = 0;          // timing experiments only (EPI_BF16, librevo_exp.so): 1 = no epilogue stores, 2 = no main loop
void gemm_set_debug(int d) { g_dbg = d; }
#endif
// the persistent kernel with queued stores (gemm256q_kernel): bf16 epilogues on whole 256-column tiles
static int g_qtail = 1;        // timing experiments only: 0 = the leftover rows behind the queued-stores kernel as a launch of their own
static int g_qstores = 1;      // timing experiments only: 0 = gemm256p_kernel for every epilogue
void gemm_set_qstores(int on) { g_qstores = on & 31; g_qtail = !(on & 32); }      // (+ 32: no fused leftover rows)
#ifdef REVO_EXPERIMENTS
constexpr bool G256Q_ROPE_BUILT = true;       // the RoPE form of the queued-stores kernel: measured, not adopted; experiment build only
#else
constexpr bool G256Q_ROPE_BUILT = false;
#endif
template <int EPI>
static bool use_256q(const GemmArgs& a) {
    if (!(EPI == EPI_BF16 || EPI == EPI_BF16_GELU || (EPI == EPI_BF16_ROPE && G256Q_ROPE_BUILT)) || !g_qstores) return false;
    if (a.N % 256 || a.K < 128 || (a.ldc & 7) || 256l * a.ldc * 2 >= (1l << 31)) return false;
    if (EPI == EPI_BF16_ROPE && (a.rope_cols % 64 || (long)a.rope_S * (a.rope_hd >> 1) * 8 >= (1l << 31))) return false;
    // RoPE: measured, not adopted (profiles/r06_gemm_queued_stores.json): the rotation's 32 table loads per lane must be
    // waited for behind the next tile's DMA requests, and hipcc's wait counts leave LDS-DMA out -- every such wait is a
    // wait for the DMA too; and with its short arithmetic the form gains nothing from stores issued early (the plain
    // bf16 form: +-0).  qkv stays on gemm256p_kernel unless the switch says 3.
    if (EPI == EPI_BF16_ROPE && (g_qstores & 7) != 3) return false;
    return true;
}
template <int EPI>
static int launch_256q(const GemmArgs& a, hipStream_t st) {
    if constexpr (EPI == EPI_BF16 || EPI == EPI_BF16_GELU || (EPI == EPI_BF16_ROPE && G256Q_ROPE_BUILT)) {
        GemmArgs b = a;
#ifdef REVO_EXPERIMENTS
        b.stamps = g_stamps; b.stamp_items = g_stamp_items;
        b.stagger_cycles = 0;
        b.stagger_groups = ((g_qstores & 7) == 2 ? 0x100 : 0)        // 2: the stores are issued but dropped (timing only)
                           | ((g_qstores & 8) ? 0x200 : 0);          // + 8: stamp item 2 = shader-clock cycles of the main loop (its clock)
#endif
        const int tiles_m = (a.M + 255) / 256, tiles_n = a.N / 256;
        int gy = xcd_stripes(tiles_n);
        if (g_force_gy) gy = g_force_gy;
        b.gy = gy;
        const int gx = 8 / gy;
        const int region = ((tiles_m + gx - 1) / gx) * ((tiles_n + gy - 1) / gy);
        const int nslot = region < 32 ? region : 32;          // 32 CUs per XCD
        REVO_FUNC_LDS((gemm256q_kernel<EPI>), G256Q_LDS);
        hipLaunchKernelGGL((gemm256q_kernel<EPI>), dim3(8 * nslot), dim3(G256_THREADS), G256Q_LDS, st, b, nslot);
        REVO_HIP_CHECK(hipGetLastError());
        return 0;
    } else {
        revo_set_error("gemm: internal: queued stores with a non-bf16 epilogue");
        return -3;
    }
}
template <int EPI>
static int launch_256(const GemmArgs& a, hipStream_t st) {
#ifdef REVO_EXPERIMENTS
    if (EPI == EPI_BF16 && g_dbg) {
        switch (g_dbg) {
            case 1: return launch_256d<EPI_BF16, 1>(a, st);
            case 2: return launch_256d<EPI_BF16, 2>(a, st);
            default: return launch_256d<EPI_BF16, 3>(a, st);
        }
    }
#endif
    if (g_persistent && a.K >= 128 && (long)((a.M + 255) / 256) * ((a.N + 255) / 256) > 256) {
#ifdef REVO_EXPERIMENTS
        if (use_256q<EPI>(a) && (a.qtail_rows > 0 || (g_phase_groups <= 1 && g_stagger_cycles == 0))) return launch_256q<EPI>(a, st);      // (its own stamps: revo_debug_gemm_stamps)
#else
        if (use_256q<EPI>(a)) return launch_256q<EPI>(a, st);
#endif
        return launch_256p<EPI>(a, st);
    }
    return launch_256d<EPI, 0>(a, st);
}

static int g_force_tile = 0;   // 0 = heuristic, 128 / 256 = forced (tests, A/B timing)
void gemm_force_tile(int t) { g_force_tile = t; }

static int g_min_tiles256 = 100;
void gemm_set_min_tiles256(int n) { g_min_tiles256 = n; }
static bool use_256(const GemmArgs& a) {
    if (256l * a.lda * 2 >= (1l << 31) || 256l * a.ldb * 2 >= (1l << 31)) return false;   // 32-bit DMA offsets per tile window
    if (g_force_tile == 256) return true;
    if (g_force_tile == 128) return false;
    // the big tile needs enough tiles to occupy the chip: below ~100 of them (a couple of images) the
    // 128-row kernels spread the same work over more CUs
    return a.M >= 1024 && a.N >= 256 && (long)((a.M + 255) / 256) * ((a.N + 255) / 256) >= g_min_tiles256;
}

bool gemm_uses_wide_epilogue(int M, int N, long lda, long ldb, long ldc) {
    GemmArgs a{};
    a.M = M; a.N = N; a.lda = lda; a.ldb = ldb;
    return use_256(a) && (N & 7) == 0 && (ldc & 7) == 0;
}

template <int EPI>
static int launch_128(const GemmArgs& a, hipStream_t st);

static int g_ring = 1, g_ring_max_tiles = 256;     // the ring kernel on (timing experiments: off) / its largest problem: one tile per CU
void gemm_set_ring(int on, int max_tiles) { g_ring = on; if (max_tiles > 0) g_ring_max_tiles = max_tiles; }
static int g_tail_split = 1;   // timing experiments only: 0 disables the tail split below
void gemm_set_tail_split(int on) { g_tail_split = on; }

// ---- split-K tail of the residual GEMMs -------------------------------------------------------
// The rows the whole rounds of the 256 x 256 kernel leave over (PE-L14 at batch 64, fc2: 68 tiles)
// occupy a quarter of the CUs for a full tile time: the K loop is a latency chain (~1.3 us per
// K-tile), so what helps is fewer K-tiles per CU, not smaller tiles.  Each tile's K range is cut in
// S parts that run on different CUs and write fp32 partial sums; a second kernel adds them in a
// fixed order (deterministic, unlike atomics) and applies bias, LayerScale and the residual add.
__global__ __launch_bounds__(256) void splitk_reduce_resid_kernel(const float* __restrict__ ws, int S, long plane,
                                                                  int rows, int N, const float* __restrict__ bias,
                                                                  const float* __restrict__ gamma,
                                                                  float* __restrict__ x, long ldc) {
    const long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i >= (long)rows * N) return;
    const int r = (int)(i / N), c = (int)(i - (long)r * N);
    f32x4 v = *(const f32x4*)(ws + i);
    for (int s = 1; s < S; ++s) v += *(const f32x4*)(ws + (long)s * plane + i);
    if (bias) v += *(const f32x4*)(bias + c);
    if (gamma) v *= *(const f32x4*)(gamma + c);
    float* dst = x + (long)r * ldc + c;
    *(f32x4*)dst = *(const f32x4*)dst + v;
}
// The same reduction, one wave per row, followed by the LayerNorm of the updated row (elementwise.hip layernorm8_kernel's
// arithmetic: 8 consecutive elements per lane and step, two-pass fp32 statistics): x_new = x + gamma * (sum_s ws_s + bias)
// is stored AND normalised into h (bf16) from the registers that hold it.  One-image forwards only (a few hundred rows).
template <int MAXG>
__global__ __launch_bounds__(256) void splitk_reduce_resid_ln_kernel(const float* __restrict__ ws, int S, long plane, int rows, int N,
                                                                     const float* __restrict__ bias, const float* __restrict__ gamma,
                                                                     float* __restrict__ x, long ldc, const float* __restrict__ lw,
                                                                     const float* __restrict__ lb, float eps,
                                                                     bf16_t* __restrict__ h, long ldo) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int ngroup = N >> 3;
    f32x4 v[MAXG][2];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < MAXG; ++i) {
        const int g = lane + 64 * i;
        if (g < ngroup) {
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) {
                const long e = (long)row * N + g * 8 + hh * 4;
                f32x4 t = *(const f32x4*)(ws + e);
                for (int p2 = 1; p2 < S; ++p2) t += *(const f32x4*)(ws + (long)p2 * plane + e);
                if (bias) t += *(const f32x4*)(bias + g * 8 + hh * 4);
                if (gamma) t *= *(const f32x4*)(gamma + g * 8 + hh * 4);
                float* dst = x + (long)row * ldc + g * 8 + hh * 4;
                t = *(const f32x4*)dst + t;
                *(f32x4*)dst = t;
                v[i][hh] = t;
            }
            s += ((v[i][0][0] + v[i][0][1]) + (v[i][0][2] + v[i][0][3])) + ((v[i][1][0] + v[i][1][1]) + (v[i][1][2] + v[i][1][3]));
        } else {
            v[i][0] = v[i][1] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
    }
    const float mean = wave_sum(s) / (float)N;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < MAXG; ++i) {
        const int g = lane + 64 * i;
        if (g < ngroup) {
#pragma unroll
            for (int hh = 0; hh < 2; ++hh)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float d = v[i][hh][j] - mean;
                    q = fmaf(d, d, q);
                }
        }
    }
    const float rstd = rsqrtf(wave_sum(q) / (float)N + eps);
#pragma unroll
    for (int i = 0; i < MAXG; ++i) {
        const int g = lane + 64 * i;
        if (g < ngroup) {
            float y[8];
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) {
                const f32x4 gm = lw ? *(const f32x4*)(lw + g * 8 + hh * 4) : (f32x4){1.f, 1.f, 1.f, 1.f};
                const f32x4 be = lb ? *(const f32x4*)(lb + g * 8 + hh * 4) : (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int j = 0; j < 4; ++j) y[hh * 4 + j] = fmaf((v[i][hh][j] - mean) * rstd, gm[j], be[j]);
            }
            uint4 o;
            o.x = pack_bf16x2(y[0], y[1]);
            o.y = pack_bf16x2(y[2], y[3]);
            o.z = pack_bf16x2(y[4], y[5]);
            o.w = pack_bf16x2(y[6], y[7]);
            *(uint4*)(h + (long)row * ldo + g * 8) = o;
        }
    }
}
static int g_splitk = 1;     // timing experiments only: 0 disables the split-K tail
void gemm_set_splitk(int on) { g_splitk = on; }
// returns 1 when it took the problem, 0 when the caller should use the ordinary path
static int try_splitk_tail(const GemmArgs& a, hipStream_t st) {
    if (!g_splitk || !a.ws || a.N % 4 || a.K < 1024) return 0;
    const int tiles_m = (a.M + 255) / 256, tiles_n = (a.N + 255) / 256, nt = a.K / 64;
    {
        // one image: so few tiles that even 128 x 64 ones leave most CUs idle -- the ring kernel on K thirds (80 tiles x 3);
        // with the LayerNorm that follows folded into the reduce this also pays at K = 1024 (out-proj: 5 K-steps per part)
        const int tiles64 = ((a.M + 127) / 128) * ((a.N + 63) / 64);
        const bool with_ln = a.ln_out && a.ln_fused && a.N % 8 == 0 && a.N <= 2048 && a.ln_ldo % 8 == 0;     // (ln_w / ln_b null: no affine)
        int S = g_ring && a.N % 64 == 0 ? 256 / tiles64 : 0;
        S = S > 8 ? 8 : S;
        while (S > 1 && nt / S < (with_ln ? 4 : 8)) --S;
        if (a.K < 2048 && !with_ln) S = 0;
        const long plane = (long)a.M * a.N;
        if (S >= 2 && plane * S <= a.ws_elems) {
            constexpr int LDS = G128R_NST * (128 + 64) * 128;
            REVO_FUNC_LDS((gemm128r_kernel<EPI_F32>), LDS);
            GemmArgs b = a;
            b.C = a.ws; b.ldc = a.N; b.bias = nullptr; b.gamma = nullptr;
            b.ksplit = S; b.c_split_stride = plane;
            hipLaunchKernelGGL((gemm128r_kernel<EPI_F32>), dim3(tiles64, S), dim3(GEMM_THREADS), LDS, st, b);
            REVO_HIP_CHECK(hipGetLastError());
            if (with_ln) {
                const dim3 grid((unsigned)((a.M + 3) / 4));
                const int groups = (a.N / 8 + 63) / 64;
#define RLN_LAUNCH(G) hipLaunchKernelGGL((splitk_reduce_resid_ln_kernel<G>), grid, dim3(256), 0, st, a.ws, S, plane, a.M, a.N, a.bias, \
                                         a.gamma, (float*)a.C, a.ldc, a.ln_w, a.ln_b, a.ln_eps, a.ln_out, a.ln_ldo)
                if (groups <= 1) RLN_LAUNCH(1);
                else if (groups <= 2) RLN_LAUNCH(2);
                else if (groups <= 3) RLN_LAUNCH(3);
                else RLN_LAUNCH(4);
#undef RLN_LAUNCH
                REVO_HIP_CHECK(hipGetLastError());
                *a.ln_fused = 1;
                return 1;
            }
            const long quads = plane / 4;
            hipLaunchKernelGGL(splitk_reduce_resid_kernel, dim3((unsigned)((quads + 255) / 256)), dim3(256), 0, st, a.ws, S, plane,
                               a.M, a.N, a.bias, a.gamma, (float*)a.C, a.ldc);
            REVO_HIP_CHECK(hipGetLastError());
            return 1;
        }
    }
    if (a.K < 2048) return 0;
    // XCD arrangement that spreads these few tiles most evenly (an XCD has 32 CUs: its tiles x S must fit)
    int gy = 1, per = 1 << 30;
    for (int g : {1, 2, 4, 8}) {
        const int pm = (tiles_m + 8 / g - 1) / (8 / g), pn = (tiles_n + g - 1) / g;
        if (pm * pn < per) { per = pm * pn; gy = g; }
    }
    int S = 32 / per;
    S = S > 8 ? 8 : S;
    while (S > 1 && nt / S < 8) --S;
    if (S < 2) return 0;
    const long plane = (long)a.M * a.N;
    if (plane * S > a.ws_elems) return 0;
    REVO_FUNC_LDS((gemm256_kernel<EPI_F32, 4>), G256_LDS);
    GemmArgs b = a;
    b.C = a.ws; b.ldc = a.N; b.bias = nullptr; b.gamma = nullptr;
    b.ksplit = S; b.c_split_stride = plane;
    b.gy = gy;
    hipLaunchKernelGGL((gemm256_kernel<EPI_F32, 4>), dim3(8 * per, S), dim3(G256_THREADS), G256_LDS, st, b);
    REVO_HIP_CHECK(hipGetLastError());
    const long quads = plane / 4;
    hipLaunchKernelGGL(splitk_reduce_resid_kernel, dim3((unsigned)((quads + 255) / 256)), dim3(256), 0, st, a.ws, S, plane,
                       a.M, a.N, a.bias, a.gamma, (float*)a.C, a.ldc);
    REVO_HIP_CHECK(hipGetLastError());
    return 1;
}

static bool use_skinny(const GemmArgs& a) { return g_force_tile == 0 && a.M <= 64 && a.K % 256 == 0 && a.N >= 256; }

// 192-row tiles.  The 256 x 256 kernel's tiles go in rounds of 256 (one workgroup per CU); PE-L14 at batch 64 has
// 36 928 rows = 144.25 tile rows, which for the two residual GEMMs (four tile columns) is 2.27 rounds: two whole ones
// and leftover rows that cost small-tile kernels more than half a round.  With 192-row tiles the same rows are
// 192 tile rows x 4 = exactly three rounds of 3/4 the work each (gemm256_mainloop<192>); the 64 rows that are still
// left make the last four tile rows 208 rows tall (1/16 of a tile's work more for 16 of the 768 workgroup-tiles: a
// launch of their own cost 10-17 us).  Returns the number of tile rows (and how many are tall), or 0 when the 256-row
// plan is no worse.
static int g_rows192 = 1;      // timing experiments only: 0 disables the 192-row form
void gemm_set_rows192(int on) { g_rows192 = on; }
static int plan_rows192(const GemmArgs& a, bool can_split_tail, int* tall) {
    if (!g_rows192 || g_force_tile || a.K < 128 || a.M % 16) return 0;
    // T tile rows of 192, the last e of them 16 rows taller: M = 192 T + 16 e
    const long tn = (a.N + 255) / 256, T = a.M / 192, e = (a.M - T * 192) / 16;
    if (e > T || e > 8) return 0;
    const long tiles192 = T * tn, rounds192 = (tiles192 + 255) / 256;
    if (tiles192 <= 256 || rounds192 * 256 - tiles192 > tiles192 / 32) return 0;      // whole rounds only
    // the 256-row plan in tile times: its whole rounds, plus a last round or what the tail split makes of it
    const long tiles = ((a.M + 255) / 256) * tn, full = tiles / 256, rem = tiles % 256;
    double cost256 = (double)full;
    if (rem) cost256 += (can_split_tail && g_tail_split && full > 0 && rem <= 160) ? (rem > 77 ? rem / 128.0 : 0.6) : 1.0;
    const double cost192 = rounds192 * 0.75 * 1.06 + (e ? 0.0625 : 0.0);
    if (cost192 >= cost256) return 0;
    *tall = (int)e;
    return (int)T;
}
template <int EPI>
static int launch_skinny(const GemmArgs& a, hipStream_t st) {
    // few output columns (N / 16 workgroups would leave most CUs idle, each pulling all of A through one L2 port): one
    // workgroup per 16 x 16 fragment, all loads of 256 k per wave in flight -- 64 x 1024 x 4096: 21 -> 9 us, x 1024: 8 -> 6
    // (same bits: per element the same K order and the same fixed-order sum of the waves' parts)
    // (not for fc1's 64 x 4096 x 1024 leftover rows: measured in the step, round 5: fc1 7.70 -> 7.79 ms with the fragment form)
    if (a.K % 1024 == 0 && a.N <= 2048)
        hipLaunchKernelGGL((gemm_skinny_kernel<EPI, 4, 4, 4>), dim3((a.N + 15) / 16, (a.M + 15) / 16), dim3(256), 0, st, a);
    else
        hipLaunchKernelGGL((gemm_skinny_kernel<EPI, 4, 1, 1>), dim3((a.N + 15) / 16), dim3(256), 0, st, a);
    REVO_HIP_CHECK(hipGetLastError());
    return 0;
}

static int g_lnf = 1;          // timing experiments only: 0 = never fold a LayerNorm into the residual GEMMs (the caller runs the kernel)
static int g_ln_planes = 1;    // ... 0 = folded, but the residual stream stays fp32 rows (+ a bf16 copy)
void gemm_set_ln_fold(int on) { g_lnf = on != 0; g_ln_planes = on != 2; }
bool gemm_ln_planes_enabled() { return g_lnf && g_ln_planes; }
// Which launch form launch_t<EPI_RESID_F32> takes for this problem, as far as the folded LayerNorm is concerned: 1 = whole
// rounds of 192-row tiles, 2 = whole rounds of 256-row tiles on the persistent kernel (both cover ALL rows with the
// row-coalesced epilogue: they can write bf16(x) + row statistics and keep the stream in planes), 0 = any other form.
static int resid_fold_form(const GemmArgs& a, int* t192 = nullptr, int* tall = nullptr) {
    if (!g_lnf || g_force_tile || a.N % LNF_SLICE || a.N / LNF_SLICE > 6 || (a.ldc & 7)) return 0;
    if (a.prefer256 || 256l * a.lda * 2 >= (1l << 31) || 256l * a.ldb * 2 >= (1l << 31)) return 0;
    const long tm = (a.M + 255) / 256, tn = (a.N + 255) / 256, tiles = tm * tn;
    if (a.M > 64 && a.K >= 1024 && tiles <= 96) return 0;              // the split-K forms of small batches
    if (use_skinny(a) || !use_256(a)) return 0;
    int tl = 0;
    if (const int t = plan_rows192(a, true, &tl)) {
        if (t192) *t192 = t;
        if (tall) *tall = tl;
        return 1;
    }
    const long full = tiles / 256 * 256, rem = tiles - full;
    if (g_tail_split && full > 0 && rem > 0 && rem <= 160) {
        const long m_tiles_main = full / tn;
        if (m_tiles_main >= 1 && m_tiles_main < tm) return 0;           // main rows + leftover rows: two launches
    }
    return (g_persistent && a.K >= 128 && tiles > 256) ? 2 : 0;
}
bool gemm_resid_folds(int M, int N, int K, long lda, long ldb, long ldc) {
    GemmArgs a{};
    a.M = M; a.N = N; a.K = K; a.lda = lda; a.ldb = ldb; a.ldc = ldc;
    return resid_fold_form(a) != 0;
}
template <int EPI>
static int launch_t(const GemmArgs& a_in, hipStream_t st) {
    GemmArgs a = a_in;
    // Producer side of a folded LayerNorm (kernels.h): only the two launch forms below that cover ALL rows with the
    // persistent 256-row kernel's row-coalesced epilogue write bf16(x) and the row statistics; every other form leaves
    // *lnf_done alone and the caller runs the LayerNorm kernel.
    const bool lnf_ok = EPI == EPI_RESID_F32 && g_lnf && a.lnf_stats && a.lnf_xb && a.lnf_done && a.N % LNF_SLICE == 0 &&
                        a.N / LNF_SLICE <= 6 && (a.ldc & 7) == 0 && a.lnf_ldxb % 8 == 0 && g_force_tile == 0;
    bf16_t* const lnf_xb = a.lnf_xb;
    float2* const lnf_stats = a.lnf_stats;
    a.lnf_xb = nullptr; a.lnf_stats = nullptr;
    if constexpr (EPI == EPI_RESID_F32) {
        // the residual stream in planes (kernels.h xp_*): only the two whole-rows forms read / write them
        const bool planes = a.xp_in || a.xp_out;
        if (planes || lnf_ok) {
            int t192 = 0, tall = 0;
            const int form = resid_fold_form(a, &t192, &tall);
            if (planes && (!form || !a.xp_hi || !a.xp_lo || a.xp_ld % 8 || (a.xp_out && (!lnf_ok || a.xp_hi != lnf_xb)))) {
                revo_set_error("gemm: the residual stream in planes needs a launch form that folds (gemm_resid_folds) and, for xp_out, "
                               "the folded LayerNorm's outputs with xp_hi == lnf_xb");
                return -2;
            }
            if (form) {
                GemmArgs a1 = a;
                a1.ln_out = nullptr;
                if (lnf_ok) { a1.lnf_xb = lnf_xb; a1.lnf_stats = lnf_stats; }
                int rc;
                if (form == 1) {
                    a1.t192_tiles = t192; a1.t192_tall = tall;
                    rc = launch_256p<EPI, 192>(a1, st);
                } else {
                    rc = launch_256p<EPI>(a1, st);
                }
                if (rc == 0 && lnf_ok) *a.lnf_done = 1;
                return rc;
            }
        }
    }
    a.xp_in = a.xp_out = 0;
    if (a.prefer256 && g_force_tile == 0 && 256l * a.lda * 2 < (1l << 31) && 256l * a.ldb * 2 < (1l << 31))
        return launch_256<EPI>(a, st);
    if constexpr (EPI == EPI_RESID_F32) {
        // a few tiles with a long K (fc2 of one to a few images): the K loop is a latency chain, cut it across CUs
        if (g_force_tile == 0 && a.M > 64 && (a.K >= 2048 || (a.K >= 1024 && a.ln_out)) && (long)((a.M + 255) / 256) * ((a.N + 255) / 256) <= 96 &&
            256l * a.lda * 2 < (1l << 31) && 256l * a.ldb * 2 < (1l << 31)) {
            const int rc2 = try_splitk_tail(a, st);
            if (rc2 < 0) return rc2;
            if (rc2 == 1) return 0;
        }
    }
    if constexpr (EPI != EPI_PATCH && EPI != EPI_BF16_ROPE) {
        if (use_skinny(a)) return launch_skinny<EPI>(a, st);
    }
    if (use_256(a)) {
        // Tail split.  The 256 x 256 kernel runs one workgroup per CU, so its tiles go in rounds of
        // 256; a last round that is mostly empty costs a full tile time (PE-L14 at batch 64: out-proj
        // and fc2 have 580 tiles = 2.27 rounds, fc1 2320 = 9.06).  When the last round would be
        // less than ~60 % full, the 256 x 256 kernel takes the M rows that make whole rounds and
        // the remaining rows go to the 128 x 128 kernel (two workgroups per CU, 4x shorter tiles).
        if constexpr (EPI == EPI_RESID_F32) {
            int tall = 0;
            if (const int t192 = plan_rows192(a, true, &tall)) {
                GemmArgs a1 = a;
                a1.ln_out = nullptr;
                a1.t192_tiles = t192; a1.t192_tall = tall;
                return launch_256p<EPI, 192>(a1, st);
            }
        }
        const long tm = (a.M + 255) / 256, tn = (a.N + 255) / 256;
        const long tiles = tm * tn, full = tiles / 256 * 256, rem = tiles - full;
        if (g_tail_split && g_force_tile == 0 && EPI != EPI_PATCH && EPI != EPI_BF16_ROPE && full > 0 && rem > 0 &&
            rem <= 160) {
            const long m_tiles_main = full / tn;
            if (m_tiles_main >= 1 && m_tiles_main < tm) {
                const int m_main = (int)(m_tiles_main * 256);
                GemmArgs a1 = a, a2 = a;
                a1.ln_out = a2.ln_out = nullptr;      // a LayerNorm can only be folded into a launch form that covers ALL rows
                a1.M = m_main;
                a2.M = a.M - m_main;
                a2.A = a.A + (long)m_main * a.lda;
                if (a2.lnc_stats) a2.lnc_stats += (long)m_main * a.lnc_parts;     // the leftover rows' statistics
                const long esz = (EPI == EPI_BF16 || EPI == EPI_BF16_GELU) ? 2 : 4;
                a2.C = (char*)a.C + (long)m_main * a.ldc * esz;
                if constexpr (EPI == EPI_BF16 || EPI == EPI_BF16_GELU) {
                    // at most 64 leftover rows behind whole rounds of the queued-stores kernel: done by that launch (qtail_rows)
                    if (a2.M <= 64 && a.K % 256 == 0 && a.N % 16 == 0 && g_qtail && use_256q<EPI>(a1) && g_persistent &&
                        (long)((a1.M + 255) / 256) * ((a1.N + 255) / 256) > 256 && (long)a2.M * a.lda * 2 < (1l << 31)) {
                        a1.qtail_rows = a2.M;
                        return launch_256<EPI>(a1, st);
                    }
                }
                const int rc = launch_256<EPI>(a1, st);
                if (rc) return rc;
                if (use_skinny(a2)) return launch_skinny<EPI>(a2, st);      // fc1 at batch 64: 64 rows are left over
                if constexpr (EPI == EPI_RESID_F32) {
                    const int rc2 = try_splitk_tail(a2, st);
                    if (rc2 < 0) return rc2;
                    if (rc2 == 1) return 0;
                }
                return launch_128<EPI>(a2, st);
            }
        }
        return launch_256<EPI>(a, st);
    }
    return launch_128<EPI>(a, st);
}

template <int EPI>
static int launch_128(const GemmArgs& a, hipStream_t st) {
    REVO_FUNC_LDS((gemm128_kernel<EPI, 128>), 2 * (128 + 128) * 128);
    const int tiles = ((a.M + 127) / 128) * ((a.N + 127) / 128);
    // at most about one 128 x 128 tile per CU: one latency-bound workgroup each; 128 x 64 tiles halve the work per
    // K step of a workgroup and put two or three of them on a CU
    if (g_force_tile == 0 && tiles <= 384 && a.K >= 512 && a.N % 64 == 0) {
        const int tiles64 = ((a.M + 127) / 128) * ((a.N + 63) / 64);
        if (g_ring && tiles64 <= g_ring_max_tiles) {
            // about one tile per CU: the K loop is a chain of DMA latencies -- six-deep ring, one workgroup per CU
            constexpr int LDS = G128R_NST * (128 + 64) * 128;
            REVO_FUNC_LDS((gemm128r_kernel<EPI>), LDS);
            hipLaunchKernelGGL((gemm128r_kernel<EPI>), dim3(tiles64), dim3(GEMM_THREADS), LDS, st, a);
            REVO_HIP_CHECK(hipGetLastError());
            return 0;
        }
        hipLaunchKernelGGL((gemm128_kernel<EPI, 64>), dim3(tiles64), dim3(GEMM_THREADS), 2 * (128 + 64) * 128, st, a);
    } else {
        hipLaunchKernelGGL((gemm128_kernel<EPI, 128>), dim3(tiles), dim3(GEMM_THREADS), 2 * (128 + 128) * 128, st, a);
    }
    REVO_HIP_CHECK(hipGetLastError());
    return 0;
}

// Few rows against very many columns with a plain fp32 output (the search pre-pass of up to 128 queries): 128 x 64 tiles
// on the six-deep ring, a round or two of workgroups whose K loops keep five steps of loads in flight (one 256 x 256 tile
// per CU is a chain of sixteen DMA latencies).
int launch_gemm_f32_ring(const GemmArgs& a, hipStream_t st) {
    if (const char* e = check_args(a)) {
        revo_set_error(e);
        return -2;
    }
    REVO_REQUIRE(a.N % 64 == 0 && a.K % 64 == 0, "gemm ring: N and K must be multiples of 64");
    const int tiles64 = ((a.M + 127) / 128) * ((a.N + 63) / 64);
    constexpr int LDS = G128R_NST * (128 + 64) * 128;
    REVO_FUNC_LDS((gemm128r_kernel<EPI_F32>), LDS);
    GemmArgs b = a;
    b.ksplit = 1;
    hipLaunchKernelGGL((gemm128r_kernel<EPI_F32>), dim3(tiles64), dim3(GEMM_THREADS), LDS, st, b);
    REVO_HIP_CHECK(hipGetLastError());
    return 0;
}

int launch_gemm(int epi, const GemmArgs& a, hipStream_t st) {
    if (const char* e = check_args(a)) {
        revo_set_error(e);
        return -2;
    }
    if (a.lnc_stats && (!a.lnc_c || a.lnc_parts < 1 || a.lnc_parts > 6 || (epi != EPI_BF16 && epi != EPI_BF16_GELU && epi != EPI_BF16_ROPE))) {
        revo_set_error("gemm: a folded LayerNorm (lnc_*) needs a bf16 epilogue, the column sums and 1..6 statistics slots per row");
        return -2;
    }
    switch (epi) {
        case EPI_BF16: return launch_t<EPI_BF16>(a, st);
        case EPI_BF16_GELU: return launch_t<EPI_BF16_GELU>(a, st);
        case EPI_RESID_F32: return launch_t<EPI_RESID_F32>(a, st);
        case EPI_F32: return launch_t<EPI_F32>(a, st);
        case EPI_PATCH: return launch_t<EPI_PATCH>(a, st);
        case EPI_BF16_ROPE:
            if (!a.rope_cs || a.rope_S <= 0 || a.rope_hd <= 0 || a.rope_hd % 8 || a.rope_cols % a.rope_hd || a.rope_cols > a.N) {
                revo_set_error("gemm: the RoPE epilogue needs a table, rope_hd % 8 == 0 and rope_cols a multiple of rope_hd");
                return -2;
            }
            return launch_t<EPI_BF16_ROPE>(a, st);
    }
    revo_set_error("gemm: unknown epilogue");
    return -2;
}

}  // namespace revo
