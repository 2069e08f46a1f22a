#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <algorithm>
#include <vector>
__device__ __forceinline__ uint32_t sort_desc_u32(uint32_t v, int lane) {
#pragma unroll
    for (int k2 = 2; k2 <= 64; k2 <<= 1) {
#pragma unroll
        for (int j = k2 >> 1; j > 0; j >>= 1) {
            const uint32_t o = __shfl_xor(v, j, 64);
            const bool take_max = (((lane & j) == 0) == ((lane & k2) == 0));
            v = take_max ? (v > o ? v : o) : (v < o ? v : o);
        }
    }
    return v;
}
__device__ __forceinline__ uint32_t merge_desc_u32(uint32_t v, int lane) {
#pragma unroll
    for (int j = 32; j > 0; j >>= 1) {
        const uint32_t o = __shfl_xor(v, j, 64);
        v = ((lane & j) == 0) ? (v > o ? v : o) : (v < o ? v : o);
    }
    return v;
}
__global__ void k(const uint32_t* row, int nvals, uint32_t* out) {
    const int lane = threadIdx.x;
    uint32_t run = 0u;
    for (int base = 0; base < nvals; base += 128) {
        const int i0 = base + lane * 2;
        uint32_t v0 = 0u, v1 = 0u;
        if (i0 < nvals) v0 = __hip_atomic_load(row + i0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (i0 + 1 < nvals) v1 = __hip_atomic_load(row + i0 + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const uint32_t a = sort_desc_u32(v0 > v1 ? v0 : v1, lane);
        const uint32_t b = sort_desc_u32(v0 > v1 ? v1 : v0, lane);
        const uint32_t brev = __shfl(b, 63 - lane, 64);      // all lanes take part: a shuffle reads 0 from inactive lanes
            uint32_t x = lane < 32 ? a : brev;
        x = merge_desc_u32(x, lane);
        const uint32_t xrev = __shfl(x, 63 - lane, 64);
            uint32_t y = lane < 32 ? run : xrev;
        y = merge_desc_u32(y, lane);
        run = lane < 32 ? y : 0u;
    }
    out[lane] = run;
}
int main() {
    for (int nvals : {128, 256, 100}) {
        std::vector<uint32_t> h(nvals);
        for (int i = 0; i < nvals; ++i) h[i] = (uint32_t)(((i * 2654435761u) >> 7) | 0x80000000u);
        uint32_t *d, *o; hipMalloc(&d, nvals * 4); hipMalloc(&o, 256);
        hipMemcpy(d, h.data(), nvals * 4, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, nvals, o);
        uint32_t r[64]; hipMemcpy(r, o, 256, hipMemcpyDeviceToHost);
        std::sort(h.begin(), h.end(), std::greater<uint32_t>());
        int bad = 0; for (int i = 0; i < 32; ++i) bad += r[i] != h[i];
        printf("nvals %d: mismatches %d  r31=%u ref31=%u\n", nvals, bad, r[31], h[31]);
    }
}
