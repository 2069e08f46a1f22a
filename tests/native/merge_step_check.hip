// Device-level check of s256_merge_step / s256_sort_desc (topk256.hip): 64..128 keys with many duplicates -> the best
// 64 distinct keys, against std::set on the host; the KSEL = 32 form likewise.  These helpers pass values between
// lanes through LDS; as plain (non-volatile) accesses the compiler was free to answer a lane's load from that lane's
// own stores, and in some inlining contexts did (a true top-50 neighbour went missing in an adversarially ordered
// gallery).  Compiled and run by tests/test_gpu_kernels.py::test_merge_step_device_check.
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <set>
#include <algorithm>
#include <functional>
#include "../../revers-o_amd/csrc/topk256.hip"   // the device functions under test
void revo_set_error(const std::string&) {}
using namespace revo;
__global__ void k(const uint64_t* in, uint64_t* out, int c) {
    __shared__ uint64_t ws[128];
    const int lane = threadIdx.x;
    uint64_t v0 = lane < c ? in[lane] : 0ull;
    uint64_t v1 = lane + 64 < c ? in[64 + lane] : 0ull;
    v0 = s256_sort_desc(v0, lane);
    v1 = s256_sort_desc(v1, lane);
    uint64_t cur = s256_merge_step<64>(0ull, s256_shfl_xor(v0, 63), ws, lane);
    cur = s256_merge_step<64>(cur, s256_shfl_xor(v1, 63), ws, lane);
    out[lane] = cur;
}
__global__ void k32(const uint64_t* in, uint64_t* out, int c) {
    __shared__ uint64_t ws[128];
    const int lane = threadIdx.x;
    uint64_t v = lane < c ? in[lane] : 0ull;
    v = s256_sort_desc(v, lane);
    // as in s256_drain: the shuffles run with all lanes active (a shuffle under a partial exec mask reads inactive lanes)
    const uint64_t best32 = s256_shfl_xor(v, 63);      // lanes 32..63 <- v[31..0]
    const uint64_t rest32 = s256_shfl_xor(v, 31);      // lanes 32..63 <- v[63..32]
    uint64_t cur = s256_merge_step<32>(0ull, lane >= 32 ? best32 : 0ull, ws, lane);
    if (s256_readlane(cur, 31) == 0ull) cur = s256_merge_step<32>(cur, lane >= 32 ? rest32 : 0ull, ws, lane);
    out[lane] = cur;
}
int main() {
    uint64_t *din, *dout;
    hipMalloc(&din, 128 * 8); hipMalloc(&dout, 64 * 8);
    srand(1);
    int bad = 0;
    for (int trial = 0; trial < 2000; ++trial) {
        const int c = 64 + rand() % 65;                 // 64..128 entries
        const int distinct = 20 + rand() % 100;         // few distinct values -> many duplicates
        std::vector<uint64_t> pool(distinct), in(128, 0);
        for (auto& x : pool) x = ((uint64_t)(0x3f000000u + rand() % 100000) << 32) | (uint32_t)~(uint32_t)(rand() % 50000);
        for (int i = 0; i < c; ++i) in[i] = pool[rand() % distinct];
        hipMemcpy(din, in.data(), 128 * 8, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, din, dout, c);
        std::vector<uint64_t> out(64);
        hipMemcpy(out.data(), dout, 64 * 8, hipMemcpyDeviceToHost);
        std::set<uint64_t, std::greater<uint64_t>> s(in.begin(), in.begin() + c);
        std::vector<uint64_t> ref(s.begin(), s.end());
        ref.resize(64, 0);
        if (ref != out) {
            if (bad < 3) {
                printf("trial %d c %d distinct %d MISMATCH\n", trial, c, distinct);
                for (int i = 0; i < 64; ++i) if (ref[i] != out[i]) { printf("  first diff at %d: ref %llx out %llx\n", i, (unsigned long long)ref[i], (unsigned long long)out[i]); break; }
            }
            ++bad;
        }
    }
    int bad32 = 0;
    for (int trial = 0; trial < 2000; ++trial) {
        const int c = 1 + rand() % 64;
        const int distinct = 5 + rand() % 80;
        std::vector<uint64_t> pool(distinct), in(128, 0);
        for (auto& x : pool) x = ((uint64_t)(0x3f000000u + rand() % 100000) << 32) | (uint32_t)~(uint32_t)(rand() % 50000);
        for (int i = 0; i < c; ++i) in[i] = pool[rand() % distinct];
        hipMemcpy(din, in.data(), 128 * 8, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k32, dim3(1), dim3(64), 0, 0, din, dout, c);
        std::vector<uint64_t> out(64);
        hipMemcpy(out.data(), dout, 64 * 8, hipMemcpyDeviceToHost);
        std::set<uint64_t, std::greater<uint64_t>> s(in.begin(), in.begin() + c);
        std::vector<uint64_t> ref(s.begin(), s.end());
        ref.resize(32, 0); ref.resize(64, 0);
        if (ref != out) ++bad32;
    }
    printf("bad trials: %d of 2000 (KSEL 64), %d of 2000 (KSEL 32)\n", bad, bad32);
    bad += bad32;
    return bad != 0;
}
