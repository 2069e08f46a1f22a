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

#ifdef REVO_EXPERIMENTS
// ---- experiment (scripts/gridbar_probe.py; verdict round 5, item 8): what does a grid-wide barrier cost against a kernel
// boundary?  A stage = every workgroup writes n16 x 16 bytes of its own slice, then reads the slice of the workgroup
// half the grid away (another XCD), which is only correct after a device-wide release / acquire.  The barrier is a
// monotonic counter in device memory: one atomic per workgroup, wave 0 spins (s_sleep) until the counter reaches
// stage x grid, bounded by the 100-MHz clock (5 ms): a grid that is not co-resident ends with the error flag set instead
// of hanging.  The chain kernel does one stage per launch.
__device__ __forceinline__ bool probe_grid_barrier(unsigned* ctr, unsigned target, unsigned* err) {
    __syncthreads();
    if (threadIdx.x == 0) {
        __atomic_thread_fence(__ATOMIC_RELEASE);            // agent scope: this workgroup's stores leave its XCD's L2
        __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(1);
            if (__builtin_amdgcn_s_memrealtime() - t0 > 500000ull) { *err = 1u; break; }
        }
    }
    __syncthreads();
    __atomic_thread_fence(__ATOMIC_ACQUIRE);
    return true;
}
// the same synchronisation without read-modify-writes: every workgroup publishes the stage number in its own word of a
// packed flag array, wave 0 of every workgroup polls all of the words (nb / 64 loads per lane per poll)
__device__ __forceinline__ void probe_flag_barrier(unsigned* flags, unsigned stage, int nb, unsigned* err) {
    __syncthreads();
    if (threadIdx.x < 64) {
        const int lane = threadIdx.x;
        if (lane == 0) {
            __atomic_thread_fence(__ATOMIC_RELEASE);
            __hip_atomic_store(flags + blockIdx.x, stage, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        for (;;) {
            bool ok = true;
            for (int j = lane; j < nb; j += 64) ok = ok && (__hip_atomic_load(flags + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= stage);
            if (__all(ok)) break;
            if (__builtin_amdgcn_s_memrealtime() - t0 > 500000ull) { if (lane == 0) *err = 1u; break; }
            __builtin_amdgcn_s_sleep(1);
        }
    }
    __syncthreads();
    __atomic_thread_fence(__ATOMIC_ACQUIRE);
}
// XCD-hierarchical form: the workgroups of one XCD arrive at that XCD's counter; the one that arrives last writes the XCD's L2
// back ONCE (release), arrives at the top counter, waits for all XCDs, and publishes the stage in the XCD's generation word;
// the others poll that word and invalidate their own CU's L1.  bar: [0] top counter, [32 x (1 + x)] counter of XCD x,
// [32 x (9 + x)] generation of XCD x, [32 x (17 + x)] census of XCD x (workgroups counted by the first, flat barrier).
__device__ __forceinline__ int probe_xcc_id() { return (int)(__builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) & 15u); }
__device__ __forceinline__ void probe_xcd_barrier(unsigned* bar, unsigned stage, int xcc, unsigned n_local, unsigned n_xcc, unsigned* err) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        const unsigned prev = __hip_atomic_fetch_add(bar + 32 * (1 + xcc), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (prev + 1 == stage * n_local) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            __hip_atomic_fetch_add(bar, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            while (__hip_atomic_load(bar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < stage * n_xcc) {
                __builtin_amdgcn_s_sleep(1);
                if (__builtin_amdgcn_s_memrealtime() - t0 > 500000ull) { *err = 1u; break; }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            __hip_atomic_store(bar + 32 * (9 + xcc), stage, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            while (__hip_atomic_load(bar + 32 * (9 + xcc), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < stage) {
                __builtin_amdgcn_s_sleep(1);
                if (__builtin_amdgcn_s_memrealtime() - t0 > 500000ull) { *err = 1u; break; }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
}
__device__ __forceinline__ float probe_stage(float4* buf, int n16, int it, int bid, int nb) {
    float4* mine = buf + (size_t)bid * n16;
    for (int i = threadIdx.x; i < n16; i += blockDim.x) mine[i] = (float4){(float)(it + bid), 1.f, 2.f, (float)i};
    return 0.f;
}
__device__ __forceinline__ float probe_read(const float4* buf, int n16, int bid, int nb) {
    const float4* other = buf + (size_t)((bid + nb / 2) % nb) * n16;
    float s = 0.f;
    for (int i = threadIdx.x; i < n16; i += blockDim.x) s += other[i].x;
    return s;
}
template <int FLAGS>
__global__ __launch_bounds__(512) void probe_gridbar_kernel(unsigned* ctr, unsigned* err, float4* buf, int n16, int iters,
                                                             float* sink, unsigned long long* stamps) {
    const int bid = blockIdx.x, nb = gridDim.x;
    float s = 0.f;
    int bad = 0;
    int xcc = 0;
    unsigned n_local = 0, n_xcc = 0;
    if (FLAGS == 2) {
        // census: who shares my XCD (one flat barrier per launch)
        xcc = probe_xcc_id();
        if (threadIdx.x == 0) __hip_atomic_fetch_add(ctr + 640 + 32 * (17 + xcc), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        probe_grid_barrier(ctr, (unsigned)nb, err);
        n_local = __hip_atomic_load(ctr + 640 + 32 * (17 + xcc), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        for (int x = 0; x < 16; ++x) n_xcc += __hip_atomic_load(ctr + 640 + 32 * (17 + x), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) ? 1u : 0u;
    }
    for (int it = 0; it < iters; ++it) {
        probe_stage(buf, n16, it, bid, nb);
        if (stamps && threadIdx.x == 0 && bid == 0) stamps[2 * it] = __builtin_amdgcn_s_memrealtime();
        if (FLAGS == 2) probe_xcd_barrier(ctr + 640, (unsigned)(it * 2 + 1), xcc, n_local, n_xcc, err);
        else if (FLAGS) probe_flag_barrier(ctr + 64, (unsigned)(it * 2 + 1), nb, err);
        else probe_grid_barrier(ctr, (unsigned)(it * 2 + 1) * nb, err);
        if (stamps && threadIdx.x == 0 && bid == 0) stamps[2 * it + 1] = __builtin_amdgcn_s_memrealtime();
        const float r = probe_read(buf, n16, bid, nb);
        // every lane checks what it read: the neighbour's stage-`it` values
        const float want = (float)(it + (bid + nb / 2) % nb);
        int cnt = 0;
        for (int i = threadIdx.x; i < n16; i += blockDim.x) ++cnt;
        if (r != want * (float)cnt) bad = 1;
        s += r;
        if (FLAGS == 2) probe_xcd_barrier(ctr + 640, (unsigned)(it * 2 + 2), xcc, n_local, n_xcc, err);
        else if (FLAGS) probe_flag_barrier(ctr + 64, (unsigned)(it * 2 + 2), nb, err);      // the slice is rewritten by the next stage
        else probe_grid_barrier(ctr, (unsigned)(it * 2 + 2) * nb, err);
    }
    if (bad) *err = 2u;
    sink[bid * blockDim.x + threadIdx.x] = s;
}
__global__ __launch_bounds__(512) void probe_chain_kernel(float4* buf, int n16, int it, float* sink, int phase) {
    const int bid = blockIdx.x, nb = gridDim.x;
    if (phase == 0) probe_stage(buf, n16, it, bid, nb);
    else sink[bid * blockDim.x + threadIdx.x] += probe_read(buf, n16, bid, nb);
}
#endif

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

#ifdef REVO_EXPERIMENTS
// mode 0: one launch with `iters` stages separated by grid barriers (2 per stage; a counter), mode 2: the same with a flag word
// per workgroup instead of the counter, mode 3: XCD-hierarchical counters; mode 1: 2 x iters launches (write / read).
// ctr_err: 2 words + 62 unused + 512 flag words + 64 unused + 33 x 32 words of the hierarchical form (zeroed here), stamps: 2 x iters 100-MHz clock words of workgroup 0 around each first barrier, or null
extern "C" int32_t revo_probe_gridbar(int32_t mode, void* ctr_err, void* buf, int32_t n16, int32_t blocks, int32_t threads,
                                      int32_t iters, float* sink, void* stamps, void* stream) {
    REVO_REQUIRE(ctr_err && buf && sink && n16 >= 1 && blocks >= 2 && blocks <= 512 && iters >= 1, "probe_gridbar: bad arguments");
    REVO_REQUIRE(threads == 256 || threads == 512, "probe_gridbar: 256 or 512 threads");
    hipStream_t st = (hipStream_t)stream;
    if (mode == 0 || mode == 2 || mode == 3) {
        int dev = 0, cus = 0;
        REVO_HIP_CHECK(hipGetDevice(&dev));
        REVO_HIP_CHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
        REVO_REQUIRE(blocks <= cus * (threads == 256 ? 2 : 1), "probe_gridbar: the grid must be co-resident");
        REVO_HIP_CHECK(hipMemsetAsync(ctr_err, 0, (640 + 33 * 32) * 4, st));
        if (mode == 0)
            hipLaunchKernelGGL(revo::probe_gridbar_kernel<0>, dim3(blocks), dim3(threads), 0, st, (unsigned*)ctr_err, (unsigned*)ctr_err + 1,
                               (float4*)buf, n16, iters, sink, (unsigned long long*)stamps);
        else if (mode == 3)
            hipLaunchKernelGGL(revo::probe_gridbar_kernel<2>, dim3(blocks), dim3(threads), 0, st, (unsigned*)ctr_err, (unsigned*)ctr_err + 1,
                               (float4*)buf, n16, iters, sink, (unsigned long long*)stamps);
        else
            hipLaunchKernelGGL(revo::probe_gridbar_kernel<1>, dim3(blocks), dim3(threads), 0, st, (unsigned*)ctr_err, (unsigned*)ctr_err + 1,
                               (float4*)buf, n16, iters, sink, (unsigned long long*)stamps);
    } else {
        for (int it = 0; it < iters; ++it) {
            hipLaunchKernelGGL(revo::probe_chain_kernel, dim3(blocks), dim3(threads), 0, st, (float4*)buf, n16, it, sink, 0);
            hipLaunchKernelGGL(revo::probe_chain_kernel, dim3(blocks), dim3(threads), 0, st, (float4*)buf, n16, it, sink, 1);
        }
    }
    REVO_HIP_CHECK(hipGetLastError());
    return 0;
}
#endif
