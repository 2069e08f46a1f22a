// Multi-head attention of the PE body (K6: softmax(q k^T / sqrt(hd)) v, flash
// style, never materialising S x S) and the single-probe attention-pool head
// (K9).  Replaces the SDPA / nn.MultiheadAttention calls inside upstream
// encode_image (reference call site core_system.py:442).
//
// Body kernel: one workgroup = 4 waves = 128 query rows of one (image, head);
// each wave owns 32 query rows.  K/V tiles of 64 keys are register-staged into
// a double-buffered LDS image.  Scores are computed transposed (S^T = K . Q^T,
// v_mfma_f32_32x32x16_bf16) so a lane owns ONE query column: the online-softmax
// row statistics are lane-local (one cross-half exchange), and the exponentiated
// accumulator is fed straight back as the B operand of O^T = V^T . P^T with no
// LDS round trip.  V^T fragments come from ds_read_b64_tr_b16 (hardware
// transpose) of the row-major V tile.
#include "kernels.h"

namespace revo {

__device__ __forceinline__ bf16x8 tr_pair(const char* lo, const char* hi) {
    const s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)lo);
    const s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)hi);
    const s16x8 c = __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(bf16x8, c);
}

template <int HD>
__global__ __launch_bounds__(256) void attn_fwd_kernel(const bf16_t* __restrict__ qkv, long ld,
                                                       bf16_t* __restrict__ out, long ldo, int S, int H, float c) {
    static_assert(HD == 64, "body attention kernel is built for head_dim 64");
    constexpr int KS = HD / 16;          // k-steps over d for S^T
    constexpr int DB = HD / 32;          // 32-wide d blocks of O^T
    constexpr int ROWB = HD * 2;         // bytes per K/V tile row (128)
    constexpr int CH = HD / 8;           // 16-byte chunks per row (8)
    constexpr int TILE = 64 * ROWB;      // 8 KB
    constexpr int NLD = (64 * CH) / 256; // chunks per thread per operand (2)
    __shared__ __attribute__((aligned(16))) char lds[4 * TILE];   // K0 K1 V0 V1

    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63;
    const int r = lane & 31, hh = lane >> 5;
    const int b = blockIdx.y / H, h = blockIdx.y - b * H;
    const int W = H * HD;
    const long rowbase = (long)b * S;
    const int q0 = blockIdx.x * 128 + wave * 32;
    const bool wave_active = q0 < S;
    const int qrow = q0 + r;
    const int qrow_c = qrow < S ? qrow : S - 1;

    const bf16_t* kg = qkv + W + h * HD;
    const bf16_t* vg = qkv + 2 * W + h * HD;

    bf16x8 qf[KS];
    {
        const bf16_t* qp = qkv + (rowbase + qrow_c) * ld + h * HD + 8 * hh;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) qf[ks] = *(const bf16x8*)(qp + 16 * ks);
    }

    // staging assignment: chunk id = tid + 256*i -> (key, ch)
    int st_key[NLD], st_ch[NLD];
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
        const int id = tid + 256 * i;
        st_key[i] = id / CH;
        st_ch[i] = id - st_key[i] * CH;
    }
    uint4 kreg[NLD], vreg[NLD];
    auto load_tile = [&](int t) {
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            int key = t * 64 + st_key[i];
            key = key < S ? key : S - 1;
            const long off = (rowbase + key) * ld + st_ch[i] * 8;
            kreg[i] = *(const uint4*)(kg + off);
            vreg[i] = *(const uint4*)(vg + off);
        }
    };
    auto store_tile = [&](int buf) {
        char* kb = lds + buf * TILE;
        char* vb = lds + (2 + buf) * TILE;
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int key = st_key[i], ch = st_ch[i];
            *(uint4*)(kb + key * ROWB + ((ch ^ ((key >> 1) & 7)) << 4)) = kreg[i];
            *(uint4*)(vb + key * ROWB + ((ch ^ (((key >> 1) & 1) << 2)) << 4)) = vreg[i];
        }
    };

    f32x16 oacc[DB];
#pragma unroll
    for (int d = 0; d < DB; ++d)
#pragma unroll
        for (int i = 0; i < 16; ++i) oacc[d][i] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;

    const int nt = (S + 63) / 64;
    load_tile(0);
    store_tile(0);
    __syncthreads();

    // per-lane constant pieces of the transposed-read address
    const int g16 = lane >> 4, li = lane & 15;
    const int tq = li >> 2, tp = li & 3;

    for (int t = 0; t < nt; ++t) {
        const int buf = t & 1;
        if (t + 1 < nt) load_tile(t + 1);
        if (wave_active) {
            const char* kb = lds + buf * TILE;
            const char* vb = lds + (2 + buf) * TILE;
            f32x16 sacc[2];
#pragma unroll
            for (int kblk = 0; kblk < 2; ++kblk) {
#pragma unroll
                for (int i = 0; i < 16; ++i) sacc[kblk][i] = 0.f;
                const int key = kblk * 32 + r;
                const char* krow = kb + key * ROWB;
                const int sw = (r >> 1) & 7;
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    const bf16x8 a = *(const bf16x8*)(krow + (((2 * ks + hh) ^ sw) << 4));
                    sacc[kblk] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, qf[ks], sacc[kblk], 0, 0, 0);
                }
            }
            if (t == nt - 1 && (S & 63)) {
#pragma unroll
                for (int kblk = 0; kblk < 2; ++kblk)
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        const int key = t * 64 + kblk * 32 + (i & 3) + 8 * (i >> 2) + 4 * hh;
                        if (key >= S) sacc[kblk][i] = -INFINITY;
                    }
            }
            float mx = -INFINITY;
#pragma unroll
            for (int kblk = 0; kblk < 2; ++kblk)
#pragma unroll
                for (int i = 0; i < 16; ++i) mx = fmaxf(mx, sacc[kblk][i]);
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            const float m_new = fmaxf(m_run, mx * c);
            const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
            m_run = m_new;
            float ps = 0.f;
#pragma unroll
            for (int kblk = 0; kblk < 2; ++kblk)
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const float p = __builtin_amdgcn_exp2f(fmaf(sacc[kblk][i], c, -m_new));
                    sacc[kblk][i] = p;
                    ps += p;
                }
            l_run = fmaf(l_run, alpha, ps);
#pragma unroll
            for (int d = 0; d < DB; ++d)
#pragma unroll
                for (int i = 0; i < 16; ++i) oacc[d][i] *= alpha;

#pragma unroll
            for (int kblk = 0; kblk < 2; ++kblk) {
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    uint4 pk;
                    pk.x = pack_bf16x2(sacc[kblk][8 * s2 + 0], sacc[kblk][8 * s2 + 1]);
                    pk.y = pack_bf16x2(sacc[kblk][8 * s2 + 2], sacc[kblk][8 * s2 + 3]);
                    pk.z = pack_bf16x2(sacc[kblk][8 * s2 + 4], sacc[kblk][8 * s2 + 5]);
                    pk.w = pack_bf16x2(sacc[kblk][8 * s2 + 6], sacc[kblk][8 * s2 + 7]);
                    const bf16x8 pb = __builtin_bit_cast(bf16x8, pk);
                    // k index (h, j) of this step is key 16*s2 + 8*(j>>2) + 4*h + (j&3) of the block
                    const int key_lo = kblk * 32 + 16 * s2 + 4 * hh + tq;
                    const int key_hi = key_lo + 8;
#pragma unroll
                    for (int d = 0; d < DB; ++d) {
                        const int chunk = d * 4 + (g16 & 1) * 2 + (tp >> 1);
                        const int within = (tp & 1) * 8;
                        const char* alo = vb + key_lo * ROWB + ((chunk ^ (((key_lo >> 1) & 1) << 2)) << 4) + within;
                        const char* ahi = vb + key_hi * ROWB + ((chunk ^ (((key_hi >> 1) & 1) << 2)) << 4) + within;
                        const bf16x8 a = tr_pair(alo, ahi);
                        oacc[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, pb, oacc[d], 0, 0, 0);
                    }
                }
            }
        }
        if (t + 1 < nt) store_tile(buf ^ 1);
        __syncthreads();
    }

    if (wave_active && qrow < S) {
        const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
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
    } else if (wave_active) {
        // keep the exchange convergent for lanes whose row is past S
        (void)__shfl_xor(l_run, 32, 64);
    }
}

int launch_attention(const bf16_t* qkv, long ld, bf16_t* out, long ldo, int B, int S, int H, int hd, hipStream_t st) {
    REVO_REQUIRE(hd == 64, "attention: only head_dim 64 is built (PE-Core B16/L14 body)");
    REVO_REQUIRE(ld % 8 == 0 && ldo % 4 == 0, "attention: strides must keep 16-byte alignment");
    if (B <= 0 || S <= 0) return 0;
    const float c = (1.0f / sqrtf((float)hd)) * 1.44269504088896340736f;
    dim3 grid((S + 127) / 128, B * H), block(256);
    hipLaunchKernelGGL((attn_fwd_kernel<64>), grid, block, 0, st, qkv, ld, out, ldo, S, H, c);
    REVO_HIP_CHECK(hipGetLastError());
    return 0;
}

// ------------------------------------------------------ attention pool -----
// One workgroup per (image, pool head): scores of the single probe against all
// S keys, softmax, weighted sum of V.  q is pre-projected and pre-scaled fp32.
template <int MAXS>
__global__ __launch_bounds__(256) void pool_attn_kernel(const float* __restrict__ q, const bf16_t* __restrict__ kv,
                                                        long ld, bf16_t* __restrict__ out, long ldo, int S, int H,
                                                        int hd) {
    __shared__ float sc[MAXS];
    __shared__ float red[8];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int b = blockIdx.x / H, h = blockIdx.x - b * H;
    const int W = H * hd;
    const bf16_t* kbase = kv + (long)b * S * ld + h * hd;
    const bf16_t* vbase = kbase + W;
    const float* qh = q + h * hd;
    // scores: one wave per key, lanes stride over d in pairs
    for (int s = wave; s < S; s += 4) {
        const bf16_t* kr = kbase + (long)s * ld;
        float acc = 0.f;
        for (int d = lane * 2; d < hd; d += 128) {
            const uint32_t w2 = *(const uint32_t*)(kr + d);
            acc = fmaf(qh[d], bf16_to_f32((bf16_t)(w2 & 0xffff)), acc);
            acc = fmaf(qh[d + 1], bf16_to_f32((bf16_t)(w2 >> 16)), acc);
        }
        acc = wave_sum(acc);
        if (lane == 0) sc[s] = acc;
    }
    __syncthreads();
    float mx = -INFINITY;
    for (int s = tid; s < S; s += 256) mx = fmaxf(mx, sc[s]);
    mx = wave_max(mx);
    if (lane == 0) red[wave] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    float sum = 0.f;
    for (int s = tid; s < S; s += 256) {
        const float p = __expf(sc[s] - mx);
        sc[s] = p;
        sum += p;
    }
    sum = wave_sum(sum);
    if (lane == 0) red[4 + wave] = sum;
    __syncthreads();
    const float inv = 1.0f / (red[4] + red[5] + red[6] + red[7]);
    for (int d = tid; d < hd; d += 256) {
        float acc = 0.f;
        for (int s = 0; s < S; ++s) acc = fmaf(sc[s], bf16_to_f32(vbase[(long)s * ld + d]), acc);
        out[(long)b * ldo + h * hd + d] = f32_to_bf16(acc * inv);
    }
}

int launch_pool_attention(const float* q, const bf16_t* kv, long ld, bf16_t* out, long ldo, int B, int S, int H,
                          int hd, hipStream_t st) {
    REVO_REQUIRE(S <= 1024, "pool attention: sequence longer than 1024 tokens");
    REVO_REQUIRE(hd % 2 == 0, "pool attention: head_dim must be even");
    if (B <= 0) return 0;
    hipLaunchKernelGGL((pool_attn_kernel<1024>), dim3(B * H), dim3(256), 0, st, q, kv, ld, out, ldo, S, H, hd);
    REVO_HIP_CHECK(hipGetLastError());
    return 0;
}

}  // namespace revo
