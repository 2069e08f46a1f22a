// The attention-pool head of the PE tower (K9-K10 of SURVEY.md section 2b; upstream AttentionPooling + proj behind
// pe_model.encode_image, core_system.py:341/:442) in fp32.
//
// Why fp32, and why it costs nothing.  The head works on ONE vector per image: every rounding of that vector goes
// straight into the embedding, whereas the roundings of the 577 token rows of the body average out in the pooling.
// Measured on PE-Core-L14-336 against the fp32 oracle (tests/test_gpu_l14_error_budget.py): embedding error 5.97e-3
// with a bf16 head, of which the 24 blocks of the body account for 3.06e-3 -- the head was the larger half.  And the
// head's one big GEMM, the K/V projection of all tokens (2.4 GFLOP per image), is not needed at all:
//   * the pool's query is the learned probe, input independent, so the logits  q_h . (W_k x_s + b_k)  are
//     (W_k,h^T q_h) . x_s + q_h . b_k,h : one [heads x W] matrix prepared at load time against the ln_post rows;
//   * the values enter only through the softmax-weighted sum, and  sum_s p_s (W_v x_s + b_v) = W_v (sum_s p_s x_s) + b_v :
//     pool the ln_post rows first (per head), then ONE skinny fp32 GEMM per image.
// Both identities are exact in real arithmetic; in fp32 the head now agrees with the oracle to ~1e-6.  What is left
// are four skinny GEMMs ([batch, W] operands, fp32 weights streamed once: 40 MB for L14) and two passes over the
// fp32 ln_post rows.
#include "kernels.h"

namespace revo {

// ---------------------------------------------------------------- load time ----
// qk[h][i] = sum_{o in head h} q[o] * Wk[o][i],  ck[h] = sum_{o in head h} q[o] * bk[o]     (q already scaled by hd^-1/2)
__global__ __launch_bounds__(256) void probe_qk_kernel(const float* __restrict__ q, const float* __restrict__ Wk,
                                                       const float* __restrict__ bk, int W, int heads,
                                                       float* __restrict__ qk, float* __restrict__ ck) {
    const int hd = W / heads;
    const int h = blockIdx.y;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < W) {
        float acc = 0.f;
        for (int o = h * hd; o < (h + 1) * hd; ++o) acc = fmaf(q[o], Wk[(long)o * W + i], acc);
        qk[(long)h * W + i] = acc;
    }
    if (blockIdx.x == 0 && threadIdx.x < 64) {
        float acc = 0.f;
        for (int o = h * hd + (int)threadIdx.x; o < (h + 1) * hd; o += 64) acc = fmaf(q[o], bk[o], acc);
        acc = wave_sum(acc);
        if (threadIdx.x == 0) ck[h] = acc;
    }
}
int launch_probe_qk(const float* q, const float* Wk, const float* bk, int W, int heads, float* qk, float* ck, hipStream_t st) {
    hipLaunchKernelGGL(probe_qk_kernel, dim3((W + 255) / 256, heads), dim3(256), 0, st, q, Wk, bk, W, heads, qk, ck);
    REVO_HIP_CHECK(hipGetLastError());
    return 0;
}

constexpr int PL_MAXH = LN_MAXH;        // heads a thread of the accumulation carries (the logits come from the ln_post kernel)

// --------------------------------------------------------------- pooled rows ----
// u[(b * H + h) * W + c] = sum_s softmax_s(logits[b, h, :])[s] * x[b * S + s][c]
// grid (W / (64 CPL), B): a workgroup owns 64 x CPL columns of one image; its sixteen waves take the token rows
// s = w, w + 16, ... (every element of x is read once, by one lane; eight rows in flight per lane -- with four waves and four
// rows one image took 94 us), each thread carries all H heads; the partial sums meet in LDS in a fixed order.  The softmax
// of the image's H x S logits is recomputed by every workgroup (a few thousand exps).
// CPL = columns per lane: 1 for a few images (B x W / 64 workgroups: one image keeps 16), 4 for a batch (16-byte loads, and
// the H probabilities of a row -- LDS broadcasts -- are read once per FOUR elements: at batch 64 the one-column form spent its
// time on them, 90 us for 151 MB).  A column's sum is taken in the same order either way: bit-identical results.
constexpr int PA_WAVES = 16;
constexpr int PA_HC = 4;          // heads per pass of the final reduction (LDS: PA_WAVES x PA_HC x 64 CPL floats)
template <int CPL>
__global__ __launch_bounds__(PA_WAVES * 64) void pool_accumulate_kernel(const float* __restrict__ x, long ldx,
                                                              const float* __restrict__ logits, int S, int W, int H,
                                                              float* __restrict__ u) {
    extern __shared__ float psm[];                    // [H][S] probabilities, then [PA_WAVES][PA_HC][64 CPL] partial sums
    float* red = psm + (((long)H * S + 3) & ~3l);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int b = blockIdx.y, c = (blockIdx.x * 64 + lane) * CPL;
    for (int h = w; h < H; h += PA_WAVES) {
        const float* lg = logits + ((long)b * H + h) * S;
        float m = -INFINITY;
        for (int s = lane; s < S; s += 64) m = fmaxf(m, lg[s]);
        m = wave_max(m);
        float sum = 0.f;
        for (int s = lane; s < S; s += 64) {
            const float e = expf(lg[s] - m);
            psm[(long)h * S + s] = e;
            sum += e;
        }
        sum = wave_sum(sum);
        const float inv = 1.0f / sum;
        for (int s = lane; s < S; s += 64) psm[(long)h * S + s] *= inv;
    }
    __syncthreads();
    float acc[PL_MAXH][CPL];
#pragma unroll
    for (int h = 0; h < PL_MAXH; ++h)
#pragma unroll
        for (int j = 0; j < CPL; ++j) acc[h][j] = 0.f;
    if (c < W) {
        const float* xc = x + (long)b * S * ldx + c;
        auto load = [&](int s, float (&v)[CPL]) {
            if constexpr (CPL == 4) {
                const f32x4 t = *(const f32x4*)(xc + (long)s * ldx);
                v[0] = t[0]; v[1] = t[1]; v[2] = t[2]; v[3] = t[3];
            } else {
                v[0] = xc[(long)s * ldx];
            }
        };
        int s = w;
        for (; s + 7 * PA_WAVES < S; s += 8 * PA_WAVES) {
            float v[8][CPL];
#pragma unroll
            for (int r = 0; r < 8; ++r) load(s + r * PA_WAVES, v[r]);
#pragma unroll
            for (int r = 0; r < 8; ++r)
#pragma unroll
                for (int h = 0; h < PL_MAXH; ++h)
                    if (h < H) {
                        const float pr = psm[(long)h * S + s + r * PA_WAVES];
#pragma unroll
                        for (int j = 0; j < CPL; ++j) acc[h][j] = fmaf(pr, v[r][j], acc[h][j]);
                    }
        }
        for (; s < S; s += PA_WAVES) {
            float v[CPL];
            load(s, v);
#pragma unroll
            for (int h = 0; h < PL_MAXH; ++h)
                if (h < H) {
                    const float pr = psm[(long)h * S + s];
#pragma unroll
                    for (int j = 0; j < CPL; ++j) acc[h][j] = fmaf(pr, v[j], acc[h][j]);
                }
        }
    }
    // the sixteen waves' partial sums, PA_HC heads at a time: thread t adds column t % (64 CPL) of head h0 + t / (64 CPL) ... in wave order
    constexpr int NC = 64 * CPL;
#pragma unroll
    for (int h0 = 0; h0 < PL_MAXH; h0 += PA_HC) {
        if (h0 >= H) break;
        if (h0) __syncthreads();
#pragma unroll
        for (int hh = 0; hh < PA_HC; ++hh)
#pragma unroll
            for (int j = 0; j < CPL; ++j) red[((long)w * PA_HC + hh) * NC + lane * CPL + j] = acc[h0 + hh][j];
        __syncthreads();
        for (int t = threadIdx.x; t < PA_HC * NC; t += PA_WAVES * 64) {
            const int hh = t / NC, cc = t - hh * NC;
            const int col = blockIdx.x * NC + cc;
            if (h0 + hh < H && col < W) {
                float sum = red[((long)0 * PA_HC + hh) * NC + cc];
#pragma unroll
                for (int o = 1; o < PA_WAVES; ++o) sum += red[((long)o * PA_HC + hh) * NC + cc];
                u[((long)b * H + h0 + hh) * W + col] = sum;
            }
        }
    }
}
int launch_pool_head_rows(const float* x, long ldx, int B, int S, int W, int H, const float* logits, float* u, hipStream_t st) {
    REVO_REQUIRE(H >= 1 && H <= PL_MAXH && W % 4 == 0 && ldx % 4 == 0, "pool head: at most 16 heads, width a multiple of 4");
    const long rows = (long)B * S;
    if (rows <= 0) return 0;
    // a batch: four columns per lane once that still gives most CUs a workgroup
    const bool wide = (long)B * ((W + 255) / 256) >= 192;
    const int cpl = wide ? 4 : 1;
    const size_t lds = ((((size_t)H * S + 3) & ~(size_t)3) + PA_WAVES * (size_t)PA_HC * 64 * cpl) * 4;
    REVO_REQUIRE(lds <= 160 * 1024, "pool head: sequence too long for the probability table in LDS");
    if (wide) {
        REVO_FUNC_LDS(pool_accumulate_kernel<4>, (int)lds);
        hipLaunchKernelGGL(pool_accumulate_kernel<4>, dim3((W + 255) / 256, B), dim3(PA_WAVES * 64), lds, st, x, ldx, logits, S, W, H, u);
    } else {
        REVO_FUNC_LDS(pool_accumulate_kernel<1>, (int)lds);
        hipLaunchKernelGGL(pool_accumulate_kernel<1>, dim3((W + 63) / 64, B), dim3(PA_WAVES * 64), lds, st, x, ldx, logits, S, W, H, u);
    }
    REVO_HIP_CHECK(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------- skinny fp32 GEMM ----
// C[M][N] (+)= epi(A[M][K] . Wt[N][K]^T + bias), everything fp32, M = batch (tens of rows), the weights streamed once.
// A workgroup owns 16 output columns; its four waves split K; v_mfma_f32_16x16x4_f32 with the operands straight from
// global memory: a lane loads 16 bytes (4 consecutive k) of its row, and MFMA j of a 16-k chunk contracts the j-th
// element of every lane's quadruple (any partition of k does: everything is summed).  The waves' partial tiles meet in
// LDS and are added in a fixed order (deterministic).  Rows are taken 64 at a time.
// Grouped A (the value projection of the pool): output columns [g * group_cols, (g + 1) * group_cols) read their A rows
// at A + g * a_group_stride (one pooled row per head).
// (K is always cut into the same NW = 16 ranges, summed in the same order, whatever M is: a row's result does not
//  depend on how many other rows -- images -- are in the batch, bit for bit.)
constexpr int SK_NW = 16;
template <int EPI, int MB>   // EPI 0: C = acc + bias, 1: C = gelu_erf(acc + bias), 2: C += acc + bias; MB 16-row blocks per pass
__global__ __launch_bounds__(SK_NW * 64) void gemm_f32_skinny_kernel(const float* __restrict__ A, long lda, long a_group_stride,
                                                                  int group_cols, const float* __restrict__ Wt, long ldw,
                                                                  const float* __restrict__ bias, int M, int N, int K,
                                                                  float* __restrict__ C, long ldc) {
    constexpr int NW = SK_NW;
    extern __shared__ float part_raw[];
    float (*part)[MB * 16][17] = (float (*)[MB * 16][17])part_raw;           // [NW][MB * 16][17]
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int r = lane & 15, kq = lane >> 4;
    const int n0 = blockIdx.x * 16;
    const float* Ab = A + (group_cols > 0 ? (long)(n0 / group_cols) * a_group_stride : 0);
    const int nr = n0 + r < N ? n0 + r : N - 1;                      // a ragged last column block re-reads a valid row
    const float* wrow = Wt + (long)nr * ldw + kq * 4;
    const int kper = ((K / 16 + NW - 1) / NW) * 16;                  // k range of a wave, a multiple of 16
    const int k0 = w * kper, k1 = (k0 + kper) < K ? (k0 + kper) : K;
    // (grid.y > 1: this workgroup takes the 16 rows [16 y, 16 y + 16) -- launch_skinny_f32)
    const int rows_y = gridDim.y > 1 ? 16 : M;
    const int m_begin = (int)blockIdx.y * rows_y, m_end = m_begin + rows_y < M ? m_begin + rows_y : M;
    for (int m0 = m_begin; m0 < m_end; m0 += MB * 16) {
        f32x4 acc[MB];
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) acc[mb] = (f32x4){0.f, 0.f, 0.f, 0.f};
        const float* arow[MB];
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) {
            const int m = m0 + mb * 16 + r;
            arow[mb] = Ab + (long)(m < M ? m : M - 1) * lda + kq * 4;
        }
        int k = k0;
        for (; k + 16 < k1; k += 32) {                                // two 16-k chunks per trip: their loads are all issued first
            const f32x4 wv0 = *(const f32x4*)(wrow + k), wv1 = *(const f32x4*)(wrow + k + 16);
            f32x4 av0[MB], av1[MB];
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) { av0[mb] = *(const f32x4*)(arow[mb] + k); av1[mb] = *(const f32x4*)(arow[mb] + k + 16); }
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int mb = 0; mb < MB; ++mb)
                    acc[mb] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv0[j], av0[mb][j], acc[mb], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int mb = 0; mb < MB; ++mb)
                    acc[mb] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv1[j], av1[mb][j], acc[mb], 0, 0, 0);
        }
        for (; k < k1; k += 16) {
            const f32x4 wv = *(const f32x4*)(wrow + k);
            f32x4 av[MB];
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) av[mb] = *(const f32x4*)(arow[mb] + k);
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int mb = 0; mb < MB; ++mb)
                    acc[mb] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[j], av[mb][j], acc[mb], 0, 0, 0);
        }
        // lane holds C[row mb*16 + (lane & 15)][cols (lane >> 4) * 4 + 0..3] of its K share
#pragma unroll
        for (int mb = 0; mb < MB; ++mb)
#pragma unroll
            for (int v = 0; v < 4; ++v) part[w][mb * 16 + r][kq * 4 + v] = acc[mb][v];
        __syncthreads();
        for (int e = threadIdx.x; e < MB * 16 * 16; e += NW * 64) {
            const int row = e >> 4, col = e & 15;
            const int m = m0 + row, n = n0 + col;
            if (m < m_end && n < N) {
                float t = part[0][row][col];
#pragma unroll
                for (int o = 1; o < NW; ++o) t += part[o][row][col];     // fixed order: deterministic
                if (bias) t += bias[n];
                float* cp = C + (long)m * ldc + n;
                if (EPI == 1) t = 0.5f * t * (1.0f + erff(t * 0.70710678118654752440f));
                if (EPI == 2) t += *cp;
                *cp = t;
            }
        }
        __syncthreads();
    }
}
template <int EPI>
static void launch_skinny_f32(const float* A, long lda, long a_group_stride, int group_cols, const float* Wt, long ldw,
                              const float* bias, int M, int N, int K, float* C, long ldc, hipStream_t st) {
    dim3 grid((unsigned)((N + 15) / 16));
    // sixteen waves split K (the weights' latency chain is what costs); up to 16 rows in one pass, else 64 rows at a time
    // -- unless 16 columns per workgroup leave most CUs without one (N = 1024: 64 workgroups, and the fp32 MFMAs of 64 CUs were
    // the kernel's time: the pool's fc2 at batch 64 took 47 us): then a workgroup takes 16 ROWS as well, grid.y = M / 16; the
    // weights are read once from HBM and M / 16 - 1 times from L2.  A row's K split and order do not depend on the form.
    if (M > 16 && (N + 15) / 16 < 192) {
        grid.y = (unsigned)((M + 15) / 16);
        hipLaunchKernelGGL((gemm_f32_skinny_kernel<EPI, 1>), grid, dim3(SK_NW * 64), SK_NW * 16 * 17 * 4, st, A, lda, a_group_stride,
                           group_cols, Wt, ldw, bias, M, N, K, C, ldc);
        return;
    }
    if (M <= 16) {
        hipLaunchKernelGGL((gemm_f32_skinny_kernel<EPI, 1>), grid, dim3(SK_NW * 64), SK_NW * 16 * 17 * 4, st, A, lda, a_group_stride,
                           group_cols, Wt, ldw, bias, M, N, K, C, ldc);
    } else {
        constexpr int LDS = SK_NW * 64 * 17 * 4;
        static bool done[REVO_MAX_DEVICES] = {};
        int dev = 0;
        if (hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < REVO_MAX_DEVICES && !done[dev]) {
            (void)hipFuncSetAttribute((const void*)(gemm_f32_skinny_kernel<EPI, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
            done[dev] = true;
        }
        hipLaunchKernelGGL((gemm_f32_skinny_kernel<EPI, 4>), grid, dim3(SK_NW * 64), LDS, st, A, lda, a_group_stride, group_cols, Wt,
                           ldw, bias, M, N, K, C, ldc);
    }
}
int launch_gemm_f32_skinny(int epi, const float* A, long lda, long a_group_stride, int group_cols, const float* Wt, long ldw,
                           const float* bias, int M, int N, int K, float* C, long ldc, hipStream_t st) {
    REVO_REQUIRE(epi >= 0 && epi <= 2, "fp32 gemm: epilogue 0..2");
    REVO_REQUIRE(K % 16 == 0 && lda % 4 == 0 && ldw % 4 == 0, "fp32 gemm: K must be a multiple of 16, rows 16-byte aligned");
    REVO_REQUIRE(group_cols == 0 || (group_cols % 16 == 0 && a_group_stride % 4 == 0), "fp32 gemm: bad A grouping");
    if (M <= 0 || N <= 0) return 0;
    if (epi == 0) launch_skinny_f32<0>(A, lda, a_group_stride, group_cols, Wt, ldw, bias, M, N, K, C, ldc, st);
    else if (epi == 1) launch_skinny_f32<1>(A, lda, a_group_stride, group_cols, Wt, ldw, bias, M, N, K, C, ldc, st);
    else launch_skinny_f32<2>(A, lda, a_group_stride, group_cols, Wt, ldw, bias, M, N, K, C, ldc, st);
    REVO_HIP_CHECK(hipGetLastError());
    return 0;
}

// dst[c][r] = src[r][c]  (fp32; visual.proj is stored [W][D], the GEMMs want [D][W])
__global__ __launch_bounds__(256) void transpose_f32_kernel(const float* __restrict__ src, int rows, int cols,
                                                            float* __restrict__ dst) {
    __shared__ float tile[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
    for (int i = ty; i < 32; i += 8)
        if (r0 + i < rows && c0 + tx < cols) tile[i][tx] = src[(long)(r0 + i) * cols + c0 + tx];
    __syncthreads();
    for (int i = ty; i < 32; i += 8)
        if (c0 + i < cols && r0 + tx < rows) dst[(long)(c0 + i) * rows + r0 + tx] = tile[tx][i];
}
int launch_transpose_f32(const float* src, int rows, int cols, float* dst, hipStream_t st) {
    hipLaunchKernelGGL(transpose_f32_kernel, dim3((cols + 31) / 32, (rows + 31) / 32), dim3(256), 0, st, src, rows, cols, dst);
    REVO_HIP_CHECK(hipGetLastError());
    return 0;
}

// fp32 [rows][cols] -> bf16 [rows][3 * ld]: ( hi | lo | hi ) of scale * src, hi = bf16(v), lo = bf16(v - hi), zero padded
// to ld per part: the patch-embedding weights for the split-precision patch GEMM (api.hip)
__global__ __launch_bounds__(256) void split_hi_lo_hi_kernel(const float* __restrict__ src, long rows, int cols, float scale,
                                                             bf16_t* __restrict__ dst, long ld) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= rows * ld) return;
    const long r = i / ld;
    const int c = (int)(i - r * ld);
    bf16_t hi = 0, lo = 0;
    if (c < cols) {
        const float v = src[r * cols + c] * scale;
        hi = f32_to_bf16(v);
        lo = f32_to_bf16(v - bf16_to_f32(hi));
    }
    bf16_t* d = dst + r * 3 * ld;
    d[c] = hi; d[ld + c] = lo; d[2 * ld + c] = hi;
}
int launch_split_hi_lo_hi(const float* src, long rows, int cols, float scale, bf16_t* dst, long ld, hipStream_t st) {
    const long n = rows * ld;
    if (n <= 0) return 0;
    hipLaunchKernelGGL(split_hi_lo_hi_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, src, rows, cols, scale, dst, ld);
    REVO_HIP_CHECK(hipGetLastError());
    return 0;
}

}  // namespace revo
