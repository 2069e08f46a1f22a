// ViT linear layers on MFMA: C = A . W^T with the elementwise tail of each layer
// fused into the epilogue (bias, exact-erf GELU, LayerScale + residual add into
// the fp32 residual stream, patch-embed position add).  K2/K4/K7/K8/K10 of
// SURVEY.md §2b; replaces the F.linear / conv2d calls the upstream PE module
// dispatches from encode_image (reference call site core_system.py:442).
#include "gemm_core.h"
#include "kernels.h"

namespace revo {

// erf to ~1.5e-7 absolute (Abramowitz-Stegun 7.1.26): the result feeds a bf16
// store (2^-9 relative), so this is "exact" GELU at the output precision.
__device__ __forceinline__ float erf_fast(float x) {
    const float ax = fabsf(x);
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, ax, 1.0f));
    float p = fmaf(1.061405429f, t, -1.453152027f);
    p = fmaf(p, t, 1.421413741f);
    p = fmaf(p, t, -0.284496736f);
    p = fmaf(p, t, 0.254829592f);
    p *= t;
    const float e = __expf(-ax * ax);
    const float r = fmaf(-p, e, 1.0f);
    return copysignf(r, x);
}
__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erf_fast(x * 0.70710678118654752f)); }

template <int EPI, int MF, int NF>
__device__ __forceinline__ void gemm_epilogue(const GemmArgs& p, int m_base, int n_base, int lane,
                                              f32x4 (&acc)[MF][NF]) {
    const int lr = lane & 15, lq = lane >> 4;
#pragma unroll
    for (int m = 0; m < MF; ++m) {
        const int row = m_base + m * 16 + lr;
        if (row >= p.M) continue;
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
            if (p.bias) {
                const f32x4 b = *(const f32x4*)(p.bias + col);
                v += b;
            }
            if (EPI == EPI_BF16 || EPI == EPI_BF16_GELU) {
                if (EPI == EPI_BF16_GELU) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] = gelu_erf(v[j]);
                }
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

// 128 x 128 x 64 tile, 4 waves as 2 (M) x 2 (N), each 64 x 64.
template <int EPI>
__global__ __launch_bounds__(GEMM_THREADS) void gemm128_kernel(GemmArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int BM = 128, BN = 128, MF = 4, NF = 4;
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

    gemm_mainloop<BM, BN, MF, NF>(la, lb, smem, p.K, wave, lane, wr * 64, wc * 64, acc);
    gemm_epilogue<EPI, MF, NF>(p, m0 + wr * 64, n0 + wc * 64, lane, acc);
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

template <int EPI>
static int launch_t(const GemmArgs& a, hipStream_t st) {
    constexpr int LDS = 2 * (128 + 128) * 128;
    static bool attr_done = false;
    if (!attr_done) {
        REVO_HIP_CHECK(hipFuncSetAttribute((const void*)gemm128_kernel<EPI>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
        attr_done = true;
    }
    const int tiles = ((a.M + 127) / 128) * ((a.N + 127) / 128);
    hipLaunchKernelGGL(gemm128_kernel<EPI>, dim3(tiles), dim3(GEMM_THREADS), LDS, st, a);
    REVO_HIP_CHECK(hipGetLastError());
    return 0;
}

int launch_gemm(int epi, const GemmArgs& a, hipStream_t st) {
    if (const char* e = check_args(a)) {
        revo_set_error(e);
        return -2;
    }
    switch (epi) {
        case EPI_BF16: return launch_t<EPI_BF16>(a, st);
        case EPI_BF16_GELU: return launch_t<EPI_BF16_GELU>(a, st);
        case EPI_RESID_F32: return launch_t<EPI_RESID_F32>(a, st);
        case EPI_F32: return launch_t<EPI_F32>(a, st);
        case EPI_PATCH: return launch_t<EPI_PATCH>(a, st);
    }
    revo_set_error("gemm: unknown epilogue");
    return -2;
}

}  // namespace revo
