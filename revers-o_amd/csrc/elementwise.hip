// HBM-bound row kernels of the embed path: LayerNorm (K3), patchify / im2col (K1
// staging), cls row (K2), 2-D RoPE (K5), L2 normalise (K11; core_system.py:447),
// dtype conversion.  All loads/stores are 8-16 B per lane, one wave per row where a
// row reduction is needed.
#include "kernels.h"

namespace revo {

// ------------------------------------------------------------ LayerNorm ----
// One wave per row; the row lives in registers (W <= 64*4*MAXC), two-pass fp32
// statistics exactly as torch.nn.functional.layer_norm computes them.
// lg (optional, fp32 output only): the attention-pool logits of the normalised row against the H probe-key vectors
// (head.hip): logits[(row / S * H + h) * S + row % S] = y . qk_h + ck_h, computed from the registers that hold the row
// (ln_post feeds nothing but the pool: its rows are read once more instead of twice)
template <int MAXC, bool OUT_BF16>
__global__ __launch_bounds__(256) void layernorm_kernel(const float* __restrict__ x, long ldx,
                                                        const float* __restrict__ w, const float* __restrict__ b,
                                                        float eps, int rows, int W, void* __restrict__ out, long ldo,
                                                        LnLogits lg) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* xr = x + (long)row * ldx;
    const int nchunk = W >> 2;
    f32x4 v[MAXC];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
        const int c = lane + 64 * i;
        if (c < nchunk) {
            v[i] = *(const f32x4*)(xr + c * 4);
            s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
        } else {
            v[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
    }
    const float mean = wave_sum(s) / (float)W;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
        const int c = lane + 64 * i;
        if (c < nchunk) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float d = v[i][j] - mean;
                q = fmaf(d, d, q);
            }
        }
    }
    const float rstd = rsqrtf(wave_sum(q) / (float)W + eps);
    float lacc[LN_MAXH];
#pragma unroll
    for (int h = 0; h < LN_MAXH; ++h) lacc[h] = 0.f;
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
        const int c = lane + 64 * i;
        if (c < nchunk) {
            const f32x4 g = w ? *(const f32x4*)(w + c * 4) : (f32x4){1.f, 1.f, 1.f, 1.f};          // (null: no affine)
            const f32x4 be = b ? *(const f32x4*)(b + c * 4) : (f32x4){0.f, 0.f, 0.f, 0.f};
            f32x4 y;
#pragma unroll
            for (int j = 0; j < 4; ++j) y[j] = fmaf((v[i][j] - mean) * rstd, g[j], be[j]);
            if (!OUT_BF16 && lg.qk) {
#pragma unroll
                for (int h = 0; h < LN_MAXH; ++h)
                    if (h < lg.H) {
                        const f32x4 k4 = *(const f32x4*)(lg.qk + (long)h * W + c * 4);
                        lacc[h] = fmaf(y[0], k4[0], lacc[h]);
                        lacc[h] = fmaf(y[1], k4[1], lacc[h]);
                        lacc[h] = fmaf(y[2], k4[2], lacc[h]);
                        lacc[h] = fmaf(y[3], k4[3], lacc[h]);
                    }
            }
            if (OUT_BF16) {
                uint2 o;
                o.x = pack_bf16x2(y[0], y[1]);
                o.y = pack_bf16x2(y[2], y[3]);
                *(uint2*)((bf16_t*)out + (long)row * ldo + c * 4) = o;
            } else {
                *(f32x4*)((float*)out + (long)row * ldo + c * 4) = y;
            }
        }
    }
    if (!OUT_BF16 && lg.qk) {
        const long bb = row / lg.S, ss = row - bb * lg.S;
#pragma unroll
        for (int h = 0; h < LN_MAXH; ++h)
            if (h < lg.H) {
                const float t = wave_sum(lacc[h]);
                if (lane == 0) lg.logits[(bb * lg.H + h) * lg.S + ss] = t + lg.ck[h];
            }
    }
}

// ln_post with the pool logits for a BATCH: R rows per wave.  The logits read H probe-key vectors against every row -- in
// the one-row form 128 sixteen-byte loads per lane and row, 128 KB per row through the CU's vector L1 (64 bytes a cycle):
// at batch 64 that, not the 302 MB the kernel moves, set its time (110 us).  Here a wave holds R rows in registers and
// every key chunk it loads serves all of them (HM = heads compiled in).  Per row the arithmetic and its order are the one-row
// kernel's: same bits.
template <int MAXC, int R, int HM>
__global__ __launch_bounds__(256) void layernorm_logits_kernel(const float* __restrict__ x, long ldx,
                                                               const float* __restrict__ w, const float* __restrict__ b,
                                                               float eps, int rows, int W, float* __restrict__ out, long ldo,
                                                               LnLogits lg) {
    const int lane = threadIdx.x & 63;
    const int row0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * R;
    if (row0 >= rows) return;
    const int nchunk = W >> 2;
    f32x4 v[R][MAXC];
    float mean[R], rstd[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int row = row0 + r < rows ? row0 + r : rows - 1;            // (a wave's rows past the end repeat the last one; not stored)
        const float* xr = x + (long)row * ldx;
#pragma unroll
        for (int i = 0; i < MAXC; ++i) {
            const int c = lane + 64 * i;
            v[r][i] = c < nchunk ? *(const f32x4*)(xr + c * 4) : (f32x4){0.f, 0.f, 0.f, 0.f};
        }
    }
    // the rows' statistics: R sums in one transposing butterfly, handed back to every lane through scalar registers
    // (lane of sum r: index bit k of r is lane bit 5 - k)
    auto all_rows = [&](float (&part)[R], float (&tot)[R]) {
        int ix;
        const float t = wave_sum_transposed<R>(part, lane, ix);
#pragma unroll
        for (int r = 0; r < R; ++r) {
            int src = 0;
#pragma unroll
            for (int k = 0; (1 << k) < R; ++k) src |= ((r >> k) & 1) << (5 - k);
            tot[r] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(t), src));
        }
    };
    {
        float part[R], tot[R];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            float s = 0.f;
#pragma unroll
            for (int i = 0; i < MAXC; ++i) {
                const int c = lane + 64 * i;
                if (c < nchunk) s += (v[r][i][0] + v[r][i][1]) + (v[r][i][2] + v[r][i][3]);
            }
            part[r] = s;
        }
        all_rows(part, tot);
#pragma unroll
        for (int r = 0; r < R; ++r) mean[r] = tot[r] / (float)W;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            float q = 0.f;
#pragma unroll
            for (int i = 0; i < MAXC; ++i) {
                const int c = lane + 64 * i;
                if (c < nchunk) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float d = v[r][i][j] - mean[r];
                        q = fmaf(d, d, q);
                    }
                }
            }
            part[r] = q;
        }
        all_rows(part, tot);
#pragma unroll
        for (int r = 0; r < R; ++r) rstd[r] = rsqrtf(tot[r] / (float)W + eps);
    }
    float lacc[R][HM];
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
        for (int h = 0; h < HM; ++h) lacc[r][h] = 0.f;
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
        const int c = lane + 64 * i;
        if (c < nchunk) {
            const f32x4 g = w ? *(const f32x4*)(w + c * 4) : (f32x4){1.f, 1.f, 1.f, 1.f};
            const f32x4 be = b ? *(const f32x4*)(b + c * 4) : (f32x4){0.f, 0.f, 0.f, 0.f};
            f32x4 y[R];
#pragma unroll
            for (int r = 0; r < R; ++r) {
#pragma unroll
                for (int j = 0; j < 4; ++j) y[r][j] = fmaf((v[r][i][j] - mean[r]) * rstd[r], g[j], be[j]);
                if (row0 + r < rows) *(f32x4*)(out + (long)(row0 + r) * ldo + c * 4) = y[r];
            }
#pragma unroll
            for (int h = 0; h < HM; ++h)
                if (h < lg.H) {
                    const f32x4 k4 = *(const f32x4*)(lg.qk + (long)h * W + c * 4);
#pragma unroll
                    for (int r = 0; r < R; ++r) {
                        lacc[r][h] = fmaf(y[r][0], k4[0], lacc[r][h]);
                        lacc[r][h] = fmaf(y[r][1], k4[1], lacc[r][h]);
                        lacc[r][h] = fmaf(y[r][2], k4[2], lacc[r][h]);
                        lacc[r][h] = fmaf(y[r][3], k4[3], lacc[r][h]);
                    }
                }
        }
    }
    // the R x HM sums of the wave in one transposing butterfly (common.h: wave_sum's addition tree, R x HM + 0 shuffles instead of
    // 6 R x HM -- as R x HM full reductions they were a third of this kernel's time), one store instruction for all of them
    static_assert(R * HM <= 32, "");
    float flat[R * HM];
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
        for (int h = 0; h < HM; ++h) flat[r * HM + h] = lacc[r][h];
    int idx;
    const float t = wave_sum_transposed<R * HM>(flat, lane, idx);
    constexpr int SAME = 64 / (R * HM);            // lanes that hold the same sum
    const int r = idx / HM, h = idx - r * HM;
    if ((lane & (SAME - 1)) == 0 && h < lg.H && row0 + r < rows) {
        const long bb = (row0 + r) / lg.S, ss = (row0 + r) - bb * lg.S;
        lg.logits[(bb * lg.H + h) * lg.S + ss] = t + lg.ck[h];
    }
}

// Same statistics, 8 consecutive elements per lane and step: two adjacent 16-byte loads and ONE 16-byte
// bf16 store (8-byte stores run at 0.5-0.7x the 16-byte rate).  W % 8 == 0, bf16 output.
template <int MAXG>
__global__ __launch_bounds__(256) void layernorm8_kernel(const float* __restrict__ x, long ldx,
                                                         const float* __restrict__ w, const float* __restrict__ b,
                                                         float eps, int rows, int W, bf16_t* __restrict__ out, long ldo) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* xr = x + (long)row * ldx;
    const int ngroup = W >> 3;
    f32x4 v[MAXG][2];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < MAXG; ++i) {
        const int g = lane + 64 * i;
        if (g < ngroup) {
            v[i][0] = *(const f32x4*)(xr + g * 8);
            v[i][1] = *(const f32x4*)(xr + g * 8 + 4);
            s += ((v[i][0][0] + v[i][0][1]) + (v[i][0][2] + v[i][0][3])) + ((v[i][1][0] + v[i][1][1]) + (v[i][1][2] + v[i][1][3]));
        } else {
            v[i][0] = v[i][1] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
    }
    const float mean = wave_sum(s) / (float)W;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < MAXG; ++i) {
        const int g = lane + 64 * i;
        if (g < ngroup) {
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float d = v[i][h][j] - mean;
                    q = fmaf(d, d, q);
                }
        }
    }
    const float rstd = rsqrtf(wave_sum(q) / (float)W + eps);
#pragma unroll
    for (int i = 0; i < MAXG; ++i) {
        const int g = lane + 64 * i;
        if (g < ngroup) {
            float y[8];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const f32x4 gm = w ? *(const f32x4*)(w + g * 8 + h * 4) : (f32x4){1.f, 1.f, 1.f, 1.f};    // (null: no affine)
                const f32x4 be = b ? *(const f32x4*)(b + g * 8 + h * 4) : (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int j = 0; j < 4; ++j) y[h * 4 + j] = fmaf((v[i][h][j] - mean) * rstd, gm[j], be[j]);
            }
            uint4 o;
            o.x = pack_bf16x2(y[0], y[1]);
            o.y = pack_bf16x2(y[2], y[3]);
            o.z = pack_bf16x2(y[4], y[5]);
            o.w = pack_bf16x2(y[6], y[7]);
            *(uint4*)(out + (long)row * ldo + g * 8) = o;
        }
    }
}

int launch_layernorm(const float* x, long ldx, const float* w, const float* b, float eps, int rows, int W, void* out,
                     long ldo, int out_is_bf16, hipStream_t st, const LnLogits* logits) {
    LnLogits lg{};
    if (logits) {
        REVO_REQUIRE(!out_is_bf16 && logits->H >= 1 && logits->H <= LN_MAXH && logits->S >= 1, "layernorm: pool logits need fp32 output and at most 16 heads");
        lg = *logits;
    }
    REVO_REQUIRE(W % 4 == 0 && W <= 2048, "layernorm: W must be a multiple of 4 and <= 2048");
    REVO_REQUIRE(ldx % 4 == 0 && ldo % 4 == 0, "layernorm: row strides must be multiples of 4");
    if (rows <= 0) return 0;
    dim3 grid((rows + 3) / 4), block(256);
    if (out_is_bf16 && W % 8 == 0 && ldx % 4 == 0 && ldo % 8 == 0 && (((uintptr_t)out) & 15) == 0) {
        const int groups = (W / 8 + 63) / 64;
        bf16_t* o = (bf16_t*)out;
        if (groups <= 1) hipLaunchKernelGGL((layernorm8_kernel<1>), grid, block, 0, st, x, ldx, w, b, eps, rows, W, o, ldo);
        else if (groups <= 2) hipLaunchKernelGGL((layernorm8_kernel<2>), grid, block, 0, st, x, ldx, w, b, eps, rows, W, o, ldo);
        else if (groups <= 3) hipLaunchKernelGGL((layernorm8_kernel<3>), grid, block, 0, st, x, ldx, w, b, eps, rows, W, o, ldo);
        else hipLaunchKernelGGL((layernorm8_kernel<4>), grid, block, 0, st, x, ldx, w, b, eps, rows, W, o, ldo);
        REVO_HIP_CHECK(hipGetLastError());
        return 0;
    }
    const int chunks = (W / 4 + 63) / 64;
    // ln_post of a batch: four rows per wave (the keys of the pool logits are loaded once for the four)
    if (lg.qk && !out_is_bf16 && rows >= 4096 && lg.H <= 8 && chunks <= 6) {
        dim3 grid4((rows + 15) / 16);
        if (chunks <= 3) hipLaunchKernelGGL((layernorm_logits_kernel<3, 4, 8>), grid4, block, 0, st, x, ldx, w, b, eps, rows, W, (float*)out, ldo, lg);
        else if (chunks <= 4) hipLaunchKernelGGL((layernorm_logits_kernel<4, 4, 8>), grid4, block, 0, st, x, ldx, w, b, eps, rows, W, (float*)out, ldo, lg);
        else hipLaunchKernelGGL((layernorm_logits_kernel<6, 4, 8>), grid4, block, 0, st, x, ldx, w, b, eps, rows, W, (float*)out, ldo, lg);
        REVO_HIP_CHECK(hipGetLastError());
        return 0;
    }
#define LN_LAUNCH(MC)                                                                                              \
    if (out_is_bf16)                                                                                               \
        hipLaunchKernelGGL((layernorm_kernel<MC, true>), grid, block, 0, st, x, ldx, w, b, eps, rows, W, out, ldo, lg); \
    else                                                                                                           \
        hipLaunchKernelGGL((layernorm_kernel<MC, false>), grid, block, 0, st, x, ldx, w, b, eps, rows, W, out, ldo, lg);
    if (chunks <= 1) { LN_LAUNCH(1) }
    else if (chunks <= 3) { LN_LAUNCH(3) }
    else if (chunks <= 4) { LN_LAUNCH(4) }
    else if (chunks <= 6) { LN_LAUNCH(6) }
    else { LN_LAUNCH(8) }
#undef LN_LAUNCH
    REVO_HIP_CHECK(hipGetLastError());
    return 0;
}

// ---------------------------------------- LayerNorm folded into a linear ----
// y = LayerNorm(x; gamma, beta) . W^T + b  ==  ((x - mean) rstd) . (gamma . W)^T + (b + W beta).  One wave per output row j:
//   out[j][k] = bf16(gamma[k] W[j][k]),   csum[j] = sum_k float(out[j][k])  (the ROUNDED weights: what the MFMA multiplies;
//   the GEMM epilogue subtracts mean * csum[j]),   bias_out[j] = bias[j] + sum_k beta[k] W[j][k].   Sums in fp64.
__global__ __launch_bounds__(256) void fold_ln_linear_kernel(const float* __restrict__ w, const float* __restrict__ gamma,
                                                             const float* __restrict__ beta, const float* __restrict__ bias,
                                                             int rows, int cols, bf16_t* __restrict__ out, long ld,
                                                             float* __restrict__ csum, float* __restrict__ bias_out) {
    const int lane = threadIdx.x & 63;
    const int j = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (j >= rows) return;
    double cs = 0.0, bs = 0.0;
    for (int k = lane; k < cols; k += 64) {
        const float wv = w[(long)j * cols + k];
        const bf16_t r = f32_to_bf16(wv * gamma[k]);
        out[(long)j * ld + k] = r;
        cs += (double)bf16_to_f32(r);
        bs += (double)beta[k] * (double)wv;
    }
    for (int k = cols + lane; k < ld; k += 64) out[(long)j * ld + k] = 0;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { cs += __shfl_xor(cs, o, 64); bs += __shfl_xor(bs, o, 64); }
    if (lane == 0) { csum[j] = (float)cs; bias_out[j] = (float)((double)bias[j] + bs); }
}
int launch_fold_ln_linear(const float* w, const float* gamma, const float* beta, const float* bias, int rows, int cols,
                          bf16_t* out, long ld, float* csum, float* bias_out, hipStream_t st) {
    REVO_REQUIRE(rows > 0 && cols > 0 && ld >= cols, "fold_ln_linear: bad sizes");
    hipLaunchKernelGGL(fold_ln_linear_kernel, dim3((rows + 3) / 4), dim3(256), 0, st, w, gamma, beta, bias, rows, cols, out, ld,
                       csum, bias_out);
    REVO_HIP_CHECK(hipGetLastError());
    return 0;
}

// residual stream in two bf16 planes (kernels.h GemmArgs::xp_*) -> fp32 rows: x = hi + lo
__global__ __launch_bounds__(256) void planes_to_f32_kernel(const bf16_t* __restrict__ hi, const bf16_t* __restrict__ lo, long ldp,
                                                            float* __restrict__ x, long ldx, long rows, int W) {
    const int chunks = W >> 3;
    const long gid = (long)blockIdx.x * 256 + threadIdx.x;
    if (gid >= rows * chunks) return;
    const long r = gid / chunks;
    const int c = (int)(gid - r * chunks) * 8;
    const uint4 a = *(const uint4*)(hi + r * ldp + c), b = *(const uint4*)(lo + r * ldp + c);
    const uint32_t aw[4] = {a.x, a.y, a.z, a.w}, bw[4] = {b.x, b.y, b.z, b.w};
    float v[8];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        v[2 * j] = __uint_as_float(aw[j] << 16) + __uint_as_float(bw[j] << 16);
        v[2 * j + 1] = __uint_as_float(aw[j] & 0xffff0000u) + __uint_as_float(bw[j] & 0xffff0000u);
    }
    *(f32x4*)(x + r * ldx + c) = (f32x4){v[0], v[1], v[2], v[3]};
    *(f32x4*)(x + r * ldx + c + 4) = (f32x4){v[4], v[5], v[6], v[7]};
}
int launch_planes_to_f32(const bf16_t* hi, const bf16_t* lo, long ldp, float* x, long ldx, long rows, int W, hipStream_t st) {
    REVO_REQUIRE(W % 8 == 0 && ldp % 8 == 0 && ldx % 4 == 0, "planes_to_f32: widths must be multiples of 8");
    const long total = rows * (W / 8);
    if (total <= 0) return 0;
    hipLaunchKernelGGL(planes_to_f32_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, hi, lo, ldp, x, ldx, rows, W);
    REVO_HIP_CHECK(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------- patchify ----
// im2col for the SPLIT-PRECISION patch GEMM (api.hip): the row of a patch holds PARTS copies / parts of its Kp values.
//   u8 images   (PARTS = 2): ( a | a ),            a = 2 v - 255: an odd integer of at most 8 bits, EXACT in bf16
//                            (ToTensor + Normalize(0.5, 0.5) is (2 v - 255) / 255; the 1 / 255 lives in the weights)
//   f32 images  (PARTS = 3): ( hi | hi | lo ),     hi = bf16(255 x), lo = bf16(255 x - hi)
// against weight rows ( hi | lo | hi ) of w / 255: a.w_hi + a.w_lo (+ a_lo.w_hi), accumulated in fp32 by the MFMA
// chain -- the patch embedding to ~2^-16 instead of 2^-8 for twice (three times) 0.2 % of the tower's FLOPs.  (The
// embedded tokens feed all 24 blocks: their rounding was a fifth of the residual stream's final error.)
// Each thread produces 8 consecutive k of one patch row (one 16-byte store per part).
template <bool U8>
__global__ __launch_bounds__(256) void patchify_kernel(const void* __restrict__ images, int B, int img, int P, int G,
                                                       bf16_t* __restrict__ out, long Kp) {
    constexpr int PARTS = U8 ? 2 : 3;
    const long ld = Kp * PARTS;
    const int chunks = (int)(Kp >> 3);
    const long gid = (long)blockIdx.x * 256 + threadIdx.x;
    const long total = (long)B * G * G * chunks;
    if (gid >= total) return;
    const int c8 = (int)(gid % chunks);
    const long prow = gid / chunks;
    const int gx = (int)(prow % G);
    const int gy = (int)((prow / G) % G);
    const int b = (int)(prow / ((long)G * G));
    const int PP = P * P, Kreal = 3 * PP;
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int k = c8 * 8 + j;
        float f = 0.f;
        if (k < Kreal) {
            const int c = k / PP, rem = k - c * PP;
            const int py = rem / P, px = rem - py * P;
            const long off = (((long)b * 3 + c) * img + (gy * P + py)) * img + (gx * P + px);
            if (U8) f = 2.0f * (float)((const uint8_t*)images)[off] - 255.0f;
            else f = ((const float*)images)[off] * 255.0f;
        }
        v[j] = f;
    }
    uint4 o;
    o.x = pack_bf16x2(v[0], v[1]);
    o.y = pack_bf16x2(v[2], v[3]);
    o.z = pack_bf16x2(v[4], v[5]);
    o.w = pack_bf16x2(v[6], v[7]);
    bf16_t* dst = out + prow * ld + c8 * 8;
    *(uint4*)dst = o;
    *(uint4*)(dst + Kp) = o;
    if (!U8) {
        const uint32_t hw[4] = {o.x, o.y, o.z, o.w};
        float lo[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) lo[j] = v[j] - bf16_to_f32((bf16_t)(hw[j >> 1] >> ((j & 1) * 16)));
        uint4 l;
        l.x = pack_bf16x2(lo[0], lo[1]);
        l.y = pack_bf16x2(lo[2], lo[3]);
        l.z = pack_bf16x2(lo[4], lo[5]);
        l.w = pack_bf16x2(lo[6], lo[7]);
        *(uint4*)(dst + 2 * Kp) = l;
    }
}

// u8 images, one workgroup per band of G patches (image b, patch row gy): the band's 3 x P pixel rows are contiguous runs of
// `img` bytes -- loaded with 16-byte loads into LDS (the gather form above reads every pixel with a byte load of its own, 37
// cache lines per wave instruction: 56-62 us at batch 64 for 22 MB in, 94 MB out) -- then every thread builds 8-value chunks
// of patch rows from LDS bytes through a table of the k -> (channel, py, px) offsets (no divisions per element).
// The values are the exact integers 2 v - 255: the same bytes as the gather form.  img % 16 == 0, 16-byte aligned images.
__global__ __launch_bounds__(256) void patchify_band_u8_kernel(const uint8_t* __restrict__ images, int img, int P, int G,
                                                               bf16_t* __restrict__ out, long Kp) {
    extern __shared__ __attribute__((aligned(16))) uint8_t pband[];       // [3 P][img] bytes, then int koff[Kreal]
    const int b = blockIdx.x / G, gy = blockIdx.x - b * G;
    const int PP = P * P, Kreal = 3 * PP, rows = 3 * P, v16 = img >> 4;
    int* koff = (int*)(pband + (((long)rows * img + 15) & ~15l));
    for (int i = threadIdx.x; i < rows * v16; i += 256) {
        const int r = i / v16, q = i - r * v16;
        const int c = r / P, py = r - c * P;
        const uint8_t* src = images + (((long)b * 3 + c) * img + (gy * P + py)) * img;
        *(uint4*)(pband + (long)r * img + q * 16) = *(const uint4*)(src + q * 16);
    }
    for (int k = threadIdx.x; k < Kreal; k += 256) {
        const int c = k / PP, rem = k - c * PP;
        const int py = rem / P, px = rem - py * P;
        koff[k] = (c * P + py) * img + px;
    }
    __syncthreads();
    const int chunks = (int)(Kp >> 3);
    const long ld = Kp * 2;
    for (int it = threadIdx.x; it < G * chunks; it += 256) {
        const int gx = it / chunks, c8 = it - gx * chunks;
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int k = c8 * 8 + j;
            v[j] = k < Kreal ? 2.0f * (float)pband[koff[k] + gx * P] - 255.0f : 0.f;
        }
        uint4 o;
        o.x = pack_bf16x2(v[0], v[1]);
        o.y = pack_bf16x2(v[2], v[3]);
        o.z = pack_bf16x2(v[4], v[5]);
        o.w = pack_bf16x2(v[6], v[7]);
        bf16_t* dst = out + ((long)(b * G + gy) * G + gx) * ld + c8 * 8;
        *(uint4*)dst = o;
        *(uint4*)(dst + Kp) = o;
    }
}

// out rows: [B*G*G][parts * Kp] with parts = 2 (u8) or 3 (f32); see the kernel
int launch_patchify(const void* images, int is_u8, int B, int img, int P, bf16_t* out, long Kp, hipStream_t st) {
    REVO_REQUIRE(img % P == 0, "patchify: image size must be a multiple of the patch size");
    REVO_REQUIRE(Kp % 8 == 0 && Kp >= 3 * P * P, "patchify: bad part width");
    const int G = img / P;
    const long total = (long)B * G * G * (Kp / 8);
    if (total <= 0) return 0;
    dim3 grid((unsigned)((total + 255) / 256)), block(256);
    const size_t band_lds = (((size_t)3 * P * img + 15) & ~(size_t)15) + (size_t)3 * P * P * 4;
    bool band = is_u8 && img % 16 == 0 && (((uintptr_t)images) & 15) == 0 && band_lds <= 64 * 1024 && B * G >= 64;
#ifdef REVO_EXPERIMENTS
    if (getenv("REVO_PATCHIFY_GATHER")) band = false;      // parity of the two forms (tests)
#endif
    if (band) {
        hipLaunchKernelGGL(patchify_band_u8_kernel, dim3((unsigned)(B * G)), block, band_lds, st, (const uint8_t*)images, img, P, G, out, Kp);
        REVO_HIP_CHECK(hipGetLastError());
        return 0;
    }
    if (is_u8) hipLaunchKernelGGL((patchify_kernel<true>), grid, block, 0, st, images, B, img, P, G, out, Kp);
    else hipLaunchKernelGGL((patchify_kernel<false>), grid, block, 0, st, images, B, img, P, G, out, Kp);
    REVO_HIP_CHECK(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------- cls rows ----
__global__ void cls_rows_kernel(float* x, long ldx, const float* cls, const float* pos, int B, int S, int W) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * W) return;
    const int b = i / W, c = i - b * W;
    x[(long)b * S * ldx + c] = cls[c] + pos[c];
}
int launch_cls_rows(float* x, long ldx, const float* cls, const float* pos, int B, int S, int W, hipStream_t st) {
    const int n = B * W;
    if (n <= 0) return 0;
    hipLaunchKernelGGL(cls_rows_kernel, dim3((n + 255) / 256), dim3(256), 0, st, x, ldx, cls, pos, B, S, W);
    REVO_HIP_CHECK(hipGetLastError());
    return 0;
}

// ----------------------------------------------------------------- RoPE ----
// cs: [S][hd/2] (cos, sin) per interleaved pair.  One thread rotates 4 pairs
// (16 bytes) of q or k.
__global__ __launch_bounds__(256) void rope_kernel(bf16_t* __restrict__ qkv, long ld, const float2* __restrict__ cs,
                                                   long rows, int S, int W, int hd) {
    const int chunks_per_row = (2 * W) >> 3;
    const long gid = (long)blockIdx.x * 256 + threadIdx.x;
    if (gid >= rows * chunks_per_row) return;
    const long row = gid / chunks_per_row;
    const int c8 = (int)(gid - row * chunks_per_row);
    const int col = c8 * 8;                 // column inside [q | k]
    const int s = (int)(row % S);
    const int within = col % hd;            // column inside the head (same for q and k thirds: W % hd == 0)
    bf16_t* ptr = qkv + row * ld + col;
    uint4 raw = *(const uint4*)ptr;
    uint32_t wds[4] = {raw.x, raw.y, raw.z, raw.w};
    const float2* t = cs + (long)s * (hd >> 1) + (within >> 1);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float x0 = bf16_to_f32((bf16_t)(wds[j] & 0xffff));
        const float x1 = bf16_to_f32((bf16_t)(wds[j] >> 16));
        const float2 c = t[j];
        const float y0 = x0 * c.x - x1 * c.y;
        const float y1 = x1 * c.x + x0 * c.y;
        wds[j] = pack_bf16x2(y0, y1);
    }
    *(uint4*)ptr = make_uint4(wds[0], wds[1], wds[2], wds[3]);
}
int launch_rope(bf16_t* qkv, long ld, const float2* cs, int rows, int S, int W, int heads, hipStream_t st) {
    const int hd = W / heads;
    REVO_REQUIRE(hd % 8 == 0 && W % 8 == 0 && ld % 8 == 0, "rope: head_dim, W and ld must be multiples of 8");
    const long total = (long)rows * ((2 * W) / 8);
    if (total <= 0) return 0;
    hipLaunchKernelGGL(rope_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, qkv, ld, cs, (long)rows, S,
                       W, hd);
    REVO_HIP_CHECK(hipGetLastError());
    return 0;
}

// ---------------------------------------------------------- conversions ----
__global__ __launch_bounds__(256) void f32_to_bf16_kernel(const float* __restrict__ src, long ld_src,
                                                          bf16_t* __restrict__ dst, long ld_dst, long rows, int cols) {
    const int chunks = (int)(ld_dst >> 3);
    const long gid = (long)blockIdx.x * 256 + threadIdx.x;
    if (gid >= rows * chunks) return;
    const long r = gid / chunks;
    const int c = (int)(gid - r * chunks) * 8;
    float v[8];
    const float* s = src + r * ld_src + c;
    if (c + 8 <= cols && ((((uintptr_t)s) & 15) == 0)) {
        const f32x4 a = *(const f32x4*)s, b = *(const f32x4*)(s + 4);
        v[0] = a[0]; v[1] = a[1]; v[2] = a[2]; v[3] = a[3];
        v[4] = b[0]; v[5] = b[1]; v[6] = b[2]; v[7] = b[3];
    } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = (c + j < cols) ? s[j] : 0.f;
    }
    uint4 o;
    o.x = pack_bf16x2(v[0], v[1]);
    o.y = pack_bf16x2(v[2], v[3]);
    o.z = pack_bf16x2(v[4], v[5]);
    o.w = pack_bf16x2(v[6], v[7]);
    *(uint4*)(dst + r * ld_dst + c) = o;
}
int launch_f32_to_bf16(const float* src, long ld_src, bf16_t* dst, long ld_dst, long rows, int cols, hipStream_t st) {
    REVO_REQUIRE(ld_dst % 8 == 0 && ld_dst >= cols, "f32_to_bf16: bad destination stride");
    const long total = rows * (ld_dst / 8);
    if (total <= 0) return 0;
    hipLaunchKernelGGL(f32_to_bf16_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, src, ld_src, dst,
                       ld_dst, rows, cols);
    REVO_HIP_CHECK(hipGetLastError());
    return 0;
}


// --------------------------------------------------------- L2 normalise ----
// e / ||e||_2 with no epsilon (core_system.py:447); an all-zero row stays zero
// (qdrant local mode guards the division the same way).
// row_stats (optional) [rows][2]: ( ||bf16(y) - y||_2 , ||bf16(y)||_2 ) of every output row y; max_stats (optional) [2]:
// running maxima, as fp32 bit patterns (non-negative floats order like unsigned integers), of ||y||_2 and of
// ||bf16(y) - y||_2 over all rows ever passed -- the quantities the search's exactness certificate is built from
// (rigorous bound of |bf16-scan score - fp32 score|).  normalize == 0 copies the rows unscaled.
__global__ __launch_bounds__(256) void l2norm_rows_kernel(const float* __restrict__ src, long ld_src,
                                                          float* __restrict__ dst_f32, long ld_f32,
                                                          bf16_t* __restrict__ dst_bf16, long ld_bf16, long rows,
                                                          int D, int normalize, float* __restrict__ row_stats,
                                                          uint32_t* __restrict__ max_stats, uint32_t* __restrict__ zero_a,
                                                          long zero_a_words, uint32_t* __restrict__ zero_b, long zero_b_words) {
    // (a search's first kernel also clears the counters and histograms its later kernels add to: two launches fewer)
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < zero_a_words; i += (long)gridDim.x * 256) zero_a[i] = 0;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < zero_b_words; i += (long)gridDim.x * 256) zero_b[i] = 0;
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* s = src + row * ld_src;
    float inv = 1.0f;
    float ny = 0.f, nb = 0.f, ne = 0.f;
    constexpr int NV = 32;                    // rows of up to 64 NV = 2048 elements are held in registers
    if (D <= 64 * NV) {
        // every load of the row requested before the first is used, and the row read once: with a load inside each step of
        // the two fma chains a one-row call (the search's query) was 32 memory round trips in a row, 12 us of a 430 us
        // search.  Same element order per lane as the loops below: same bits.
        float v[NV];
#pragma unroll
        for (int i = 0; i < NV; ++i) v[i] = lane + 64 * i < D ? s[lane + 64 * i] : 0.f;
        if (normalize) {
            float q = 0.f;
#pragma unroll
            for (int i = 0; i < NV; ++i)
                if (lane + 64 * i < D) q = fmaf(v[i], v[i], q);
            q = wave_sum(q);
            inv = q > 0.f ? 1.0f / sqrtf(q) : 1.0f;
        }
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = lane + 64 * i;
            if (c < D) {
                const float y = v[i] * inv;
                const bf16_t yb = f32_to_bf16(y);
                if (dst_f32) dst_f32[row * ld_f32 + c] = y;
                if (dst_bf16) dst_bf16[row * ld_bf16 + c] = yb;
                const float fb = bf16_to_f32(yb), d = fb - y;     // the difference of two neighbouring floats is exact
                ny = fmaf(y, y, ny); nb = fmaf(fb, fb, nb); ne = fmaf(d, d, ne);
            }
        }
    } else {
        if (normalize) {
            float q = 0.f;
            for (int c = lane; c < D; c += 64) q = fmaf(s[c], s[c], q);
            q = wave_sum(q);
            inv = q > 0.f ? 1.0f / sqrtf(q) : 1.0f;
        }
        for (int c = lane; c < D; c += 64) {
            const float y = s[c] * inv;
            const bf16_t yb = f32_to_bf16(y);
            if (dst_f32) dst_f32[row * ld_f32 + c] = y;
            if (dst_bf16) dst_bf16[row * ld_bf16 + c] = yb;
            const float fb = bf16_to_f32(yb), d = fb - y;
            ny = fmaf(y, y, ny); nb = fmaf(fb, fb, nb); ne = fmaf(d, d, ne);
        }
    }
    if (!row_stats && !max_stats) return;
    ny = sqrtf(wave_sum(ny)); nb = sqrtf(wave_sum(nb)); ne = sqrtf(wave_sum(ne));
    if (lane == 0) {
        if (row_stats) { row_stats[row * 2] = ne; row_stats[row * 2 + 1] = nb; }
        if (max_stats) {
            // NaN / inf rows (never produced by the embed path) would poison the maxima: they are left out
            if (ny == ny && ny < INFINITY) atomicMax(max_stats, __float_as_uint(ny));
            if (ne == ne && ne < INFINITY) atomicMax(max_stats + 1, __float_as_uint(ne));
        }
    }
}
int launch_l2norm_rows(const float* src, long ld_src, float* dst_f32, long ld_f32, bf16_t* dst_bf16, long ld_bf16,
                       long rows, int D, hipStream_t st, int normalize, float* row_stats, uint32_t* max_stats,
                       uint32_t* zero_a, long zero_a_words, uint32_t* zero_b, long zero_b_words) {
    if (rows <= 0) return 0;
    hipLaunchKernelGGL(l2norm_rows_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, st, src, ld_src, dst_f32,
                       ld_f32, dst_bf16, ld_bf16, rows, D, normalize, row_stats, max_stats, zero_a, zero_a_words, zero_b,
                       zero_b_words);
    REVO_HIP_CHECK(hipGetLastError());
    return 0;
}

}  // namespace revo
