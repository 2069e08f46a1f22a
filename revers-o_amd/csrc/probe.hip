// Calibration probes of bench.py (include/revo.h revo_probe_*): two fixed kernels that do not change from round to round,
// run right before the timed region, so that a bench line carries the speed of the BOX it was taken on.  The MI355X boxes
// of one pool differ by several per cent on MFMA-dense loops (the chip holds its clock down under load, by a
// device-dependent amount: MI355X_MICROARCH.md, DVFS give-back item 5); without a same-run yardstick a change of the code
// and a change of the silicon look the same in the headline number.
//   probe_mfma_kernel: register-resident v_mfma_f32_16x16x32_bf16 loop, no memory traffic inside the loop, random
//                      operands (zeros clock ~20 % higher: ibid. item 1), four workgroups of four waves per CU.
//   probe_copy_kernel: 16 bytes per lane, grid-stride: the HBM copy rate (read + write).
#include "common.h"
#include "../../include/revo.h"

namespace revo {

__global__ __launch_bounds__(256) void probe_mfma_kernel(const bf16x8* __restrict__ src, int n_frag, float* __restrict__ sink, int iters) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    bf16x8 a[4], b[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        a[i] = src[(int)(((long)t * 8 + i) % n_frag)];
        b[i] = src[(int)(((long)t * 8 + 4 + i) % n_frag)];
    }
    f32x4 acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int j = 0; j < 8; ++j)
                acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[(j + r) & 3], b[(j >> 1) & 3], acc[j], 0, 0, 0);
    }
    f32x4 s = acc[0];
#pragma unroll
    for (int j = 1; j < 8; ++j) s += acc[j];
    sink[t] = (s[0] + s[1]) + (s[2] + s[3]);
}

__global__ __launch_bounds__(256) void probe_copy_kernel(const uint4* __restrict__ src, uint4* __restrict__ dst, long n16) {
    // a workgroup copies contiguous 64-KiB chunks (4096 x 16 bytes), eight loads per lane in flight before the first store
    // (one load per trip at a 16-MiB stride measured 4.3-4.8 TB/s: latency and DRAM page misses, not bandwidth)
    const long chunks = n16 >> 12;
    for (long c = blockIdx.x; c < chunks; c += gridDim.x) {
        const uint4* s = src + (c << 12) + threadIdx.x;
        uint4* d = dst + (c << 12) + threadIdx.x;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            uint4 v[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = s[(h * 8 + i) * 256];
#pragma unroll
            for (int i = 0; i < 8; ++i) d[(h * 8 + i) * 256] = v[i];
        }
    }
    for (long i = (chunks << 12) + (long)blockIdx.x * 256 + threadIdx.x; i < n16; i += (long)gridDim.x * 256) dst[i] = src[i];
}

}  // namespace revo

extern "C" int32_t revo_probe_mfma(const void* src_bf16, int64_t n_elems, float* sink, int32_t blocks, int32_t iters, void* stream) {
    REVO_REQUIRE(src_bf16 && sink && n_elems >= 8 * 8 && blocks >= 1 && iters >= 1, "probe_mfma: bad arguments");
    REVO_REQUIRE(((uintptr_t)src_bf16 & 15) == 0, "probe_mfma: the operand buffer must be 16-byte aligned");
    hipLaunchKernelGGL(revo::probe_mfma_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const bf16x8*)src_bf16,
                       (int)(n_elems / 8 > (1 << 30) ? (1 << 30) : n_elems / 8), sink, iters);
    REVO_HIP_CHECK(hipGetLastError());
    return 0;
}
extern "C" int64_t revo_probe_mfma_flops(int32_t blocks, int32_t iters) {
    return (int64_t)blocks * 4 /*waves*/ * iters * 32 /*MFMAs per trip*/ * (2ll * 16 * 16 * 32);
}
extern "C" int32_t revo_probe_copy(void* dst, const void* src, int64_t bytes, void* stream) {
    REVO_REQUIRE(dst && src && bytes >= 16 && bytes % 16 == 0, "probe_copy: bad arguments");
    REVO_REQUIRE((((uintptr_t)dst | (uintptr_t)src) & 15) == 0, "probe_copy: buffers must be 16-byte aligned");
    hipLaunchKernelGGL(revo::probe_copy_kernel, dim3(256 * 8), dim3(256), 0, (hipStream_t)stream, (const uint4*)src, (uint4*)dst,
                       (long)(bytes / 16));
    REVO_HIP_CHECK(hipGetLastError());
    return 0;
}
