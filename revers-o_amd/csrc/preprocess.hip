// Device-side image preprocessing: crop + squash-resize of decoded uint8 RGB frames to the model
// resolution.  Output is the uint8 CHW batch that
// revo_vit_forward(image_dtype = 1) consumes; ToTensor + Normalize stay fused in its patchify kernel.
//
// The arithmetic is Pillow's 8-bit separable resample (what the reference's
// transforms.get_image_transform(336), core_system.py:200, runs on PIL images at :335 / :439):
// triangle filter with support max(scale, 1), coefficients normalised in double precision and
// rounded to 22 fractional bits, int32 accumulation, uint8 intermediate after the horizontal
// pass.  Results are bit-identical to Image.crop(box).resize((S, S), Image.BILINEAR); the
// coefficient kernel therefore runs in fp64 with contraction off.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/revo.h"
#include "common.h"

namespace revo {

constexpr int RS_PRECISION = 22;

struct CropJobDev {
    const uint8_t* src;
    long stride;
    int x0, y0, cw, ch;
};

// bounds[job][axis][S][2], coeffs[job][axis][S][KS]; axis 0 = horizontal (crop width), 1 = vertical
__global__ void resize_coeffs_kernel(const CropJobDev* __restrict__ jobs, int S, int KS, int* __restrict__ bounds,
                                     int* __restrict__ coeffs) {
#pragma clang fp contract(off)
    const int job = blockIdx.x >> 1, axis = blockIdx.x & 1;
    const int in_size = axis ? jobs[job].ch : jobs[job].cw;
    const double scale = (double)(float)in_size / (double)S;
    const double filterscale = scale < 1.0 ? 1.0 : scale;
    const double support = filterscale;          // bilinear: support 1.0 * filterscale
    const double ss = 1.0 / filterscale;
    int* bnd = bounds + ((long)blockIdx.x * S) * 2;
    int* kk = coeffs + (long)blockIdx.x * S * KS;
    for (int xx = threadIdx.x; xx < S; xx += blockDim.x) {
        const double center = 0.0 + ((double)xx + 0.5) * scale;
        int xmin = (int)(center - support + 0.5);
        if (xmin < 0) xmin = 0;
        int xmax = (int)(center + support + 0.5);
        if (xmax > in_size) xmax = in_size;
        xmax -= xmin;
        double ww = 0.0;
        for (int x = 0; x < xmax; ++x) {
            double a = ((double)(x + xmin) - center + 0.5) * ss;
            if (a < 0.0) a = -a;
            ww += a < 1.0 ? 1.0 - a : 0.0;
        }
        int* k = kk + (long)xx * KS;
        for (int x = 0; x < KS; ++x) {
            double w = 0.0;
            if (x < xmax) {
                double a = ((double)(x + xmin) - center + 0.5) * ss;
                if (a < 0.0) a = -a;
                w = a < 1.0 ? 1.0 - a : 0.0;
                if (ww != 0.0) w = w / ww;
            }
            const double s = w * (double)(1 << RS_PRECISION);
            k[x] = w < 0.0 ? (int)(-0.5 + s) : (int)(0.5 + s);
        }
        bnd[xx * 2] = xmin;
        bnd[xx * 2 + 1] = xmax;
    }
}

__device__ __forceinline__ uint8_t rs_clip8(int v) {
    v >>= RS_PRECISION;
    return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
}

// tmp[job][row][xo][3] <- src rows of the crop; grid (ceil(S/256), max crop height, jobs)
__global__ __launch_bounds__(256) void resize_h_kernel(const CropJobDev* __restrict__ jobs, int S, int KS,
                                                       const int* __restrict__ bounds, const int* __restrict__ coeffs,
                                                       uint8_t* __restrict__ tmp, long tmp_job_stride) {
    const int job = blockIdx.z, row = blockIdx.y;
    const CropJobDev j = jobs[job];
    const int xo = blockIdx.x * 256 + threadIdx.x;
    if (row >= j.ch || xo >= S) return;
    const long tab = (long)(job * 2) * S;
    const int lo = bounds[(tab + xo) * 2], n = bounds[(tab + xo) * 2 + 1];
    const int* k = coeffs + (tab + xo) * KS;
    const uint8_t* p = j.src + (long)(j.y0 + row) * j.stride + (long)(j.x0 + lo) * 3;
    int s0 = 1 << (RS_PRECISION - 1), s1 = s0, s2 = s0;
    for (int x = 0; x < n; ++x) {
        const int w = k[x];
        s0 += (int)p[x * 3] * w;
        s1 += (int)p[x * 3 + 1] * w;
        s2 += (int)p[x * 3 + 2] * w;
    }
    uint8_t* o = tmp + job * tmp_job_stride + ((long)row * S + xo) * 3;
    o[0] = rs_clip8(s0);
    o[1] = rs_clip8(s1);
    o[2] = rs_clip8(s2);
}

// out[job][c][yo][xo] <- tmp; grid (ceil(S/256), S, jobs)
__global__ __launch_bounds__(256) void resize_v_kernel(const CropJobDev* __restrict__ jobs, int S, int KS,
                                                       const int* __restrict__ bounds, const int* __restrict__ coeffs,
                                                       const uint8_t* __restrict__ tmp, long tmp_job_stride,
                                                       uint8_t* __restrict__ out) {
    const int job = blockIdx.z, yo = blockIdx.y;
    const int xo = blockIdx.x * 256 + threadIdx.x;
    if (xo >= S) return;
    const long tab = (long)(job * 2 + 1) * S;
    const int lo = bounds[(tab + yo) * 2], n = bounds[(tab + yo) * 2 + 1];
    const int* k = coeffs + (tab + yo) * KS;
    const uint8_t* p = tmp + job * tmp_job_stride + ((long)lo * S + xo) * 3;
    int s0 = 1 << (RS_PRECISION - 1), s1 = s0, s2 = s0;
    for (int y = 0; y < n; ++y) {
        const int w = k[y];
        const uint8_t* q = p + (long)y * S * 3;
        s0 += (int)q[0] * w;
        s1 += (int)q[1] * w;
        s2 += (int)q[2] * w;
    }
    uint8_t* o = out + (long)job * 3 * S * S + (long)yo * S + xo;
    o[0] = rs_clip8(s0);
    o[(long)S * S] = rs_clip8(s1);
    o[2l * S * S] = rs_clip8(s2);
}

struct PreWorkspace {
    CropJobDev* jobs = nullptr;
    int jobs_cap = 0;
    int* bounds = nullptr;
    long bounds_cap = 0;
    int* coeffs = nullptr;
    long coeffs_cap = 0;
    uint8_t* tmp = nullptr;
    long tmp_cap = 0;
    // the job table travels through a ring of pinned host buffers (the call does not wait for its copy: a buffer is
    // reused only after the copy that read it has run), and calls are ordered by the stream they are enqueued on
    static constexpr int RING = 8;
    CropJobDev* pin[RING] = {};
    int pin_cap[RING] = {};
    hipEvent_t pin_ev[RING] = {};
    int ring = 0;
    hipStream_t last_stream = nullptr;
    bool has_last = false;
};
static PreWorkspace g_ws[16];
static std::mutex g_ws_mutex;

template <typename T>
static int grow(T*& p, long& cap, long need) {
    if (need <= cap) return 0;
    if (p) REVO_HIP_CHECK(hipFree(p));
    p = nullptr;
    cap = 0;
    const long want = need + need / 4;
    REVO_HIP_CHECK(hipMalloc((void**)&p, want * sizeof(T)));
    cap = want;
    return 0;
}

static int crop_resize(const revo_crop_job* jobs, int n, int S, uint8_t* out, hipStream_t st) {
    int dev = 0;
    REVO_HIP_CHECK(hipGetDevice(&dev));
    REVO_REQUIRE(dev >= 0 && dev < 16, "crop_resize: device index out of range");
    std::vector<CropJobDev> host(n);
    int max_ch = 0, KS = 3;
    for (int i = 0; i < n; ++i) {
        const revo_crop_job& j = jobs[i];
        REVO_REQUIRE(j.src != nullptr, "crop_resize: null source image");
        REVO_REQUIRE(j.height > 0 && j.width > 0 && j.height <= 65535, "crop_resize: source image must be 1..65535 rows");
        REVO_REQUIRE(j.row_stride >= (int64_t)j.width * 3, "crop_resize: row_stride smaller than width * 3");
        REVO_REQUIRE(j.x0 >= 0 && j.y0 >= 0 && j.x1 <= j.width && j.y1 <= j.height && j.x0 < j.x1 && j.y0 < j.y1,
                     "crop_resize: box must satisfy 0 <= x0 < x1 <= width and 0 <= y0 < y1 <= height");
        host[i] = CropJobDev{j.src, (long)j.row_stride, j.x0, j.y0, j.x1 - j.x0, j.y1 - j.y0};
        max_ch = std::max(max_ch, host[i].ch);
        for (int side : {host[i].cw, host[i].ch}) {
            const double scale = (double)(float)side / (double)S;
            const int ks = (int)std::ceil(scale < 1.0 ? 1.0 : scale) * 2 + 1;
            KS = std::max(KS, ks);
        }
    }
    std::lock_guard<std::mutex> lock(g_ws_mutex);
    PreWorkspace& w = g_ws[dev];
    // The device workspace (job table, filter tables, the intermediate rows) is shared by the calls on this device:
    // calls on ONE stream are ordered by it; a call on another stream first waits for the previous one's.
    if (w.has_last && w.last_stream != st) REVO_HIP_CHECK(hipStreamSynchronize(w.last_stream));
    w.last_stream = st; w.has_last = true;
    if (n > w.jobs_cap) {
        if (w.jobs) REVO_HIP_CHECK(hipFree(w.jobs));          // (hipFree waits for the device: nothing still reads it)
        w.jobs = nullptr;
        w.jobs_cap = 0;
        REVO_HIP_CHECK(hipMalloc((void**)&w.jobs, sizeof(CropJobDev) * (size_t)(n + 64)));
        w.jobs_cap = n + 64;
    }
    const long tmp_job_stride = (long)max_ch * S * 3;
    if (int rc = grow(w.bounds, w.bounds_cap, (long)n * 2 * S * 2)) return rc;
    if (int rc = grow(w.coeffs, w.coeffs_cap, (long)n * 2 * S * KS)) return rc;
    if (int rc = grow(w.tmp, w.tmp_cap, (long)n * tmp_job_stride)) return rc;
    const int slot = w.ring++ % PreWorkspace::RING;
    if (w.pin_ev[slot]) REVO_HIP_CHECK(hipEventSynchronize(w.pin_ev[slot]));     // eight calls back: long done
    else REVO_HIP_CHECK(hipEventCreateWithFlags(&w.pin_ev[slot], hipEventDisableTiming));
    if (n > w.pin_cap[slot]) {
        if (w.pin[slot]) REVO_HIP_CHECK(hipHostFree(w.pin[slot]));
        w.pin[slot] = nullptr; w.pin_cap[slot] = 0;
        REVO_HIP_CHECK(hipHostMalloc((void**)&w.pin[slot], sizeof(CropJobDev) * (size_t)(n + 64), hipHostMallocDefault));
        w.pin_cap[slot] = n + 64;
    }
    memcpy(w.pin[slot], host.data(), sizeof(CropJobDev) * (size_t)n);
    REVO_HIP_CHECK(hipMemcpyAsync(w.jobs, w.pin[slot], sizeof(CropJobDev) * (size_t)n, hipMemcpyHostToDevice, st));
    REVO_HIP_CHECK(hipEventRecord(w.pin_ev[slot], st));
    hipLaunchKernelGGL(resize_coeffs_kernel, dim3(n * 2), dim3(256), 0, st, w.jobs, S, KS, w.bounds, w.coeffs);
    REVO_HIP_CHECK(hipGetLastError());
    const int bx = (S + 255) / 256;
    hipLaunchKernelGGL(resize_h_kernel, dim3(bx, max_ch, n), dim3(256), 0, st, w.jobs, S, KS, w.bounds, w.coeffs, w.tmp,
                       tmp_job_stride);
    REVO_HIP_CHECK(hipGetLastError());
    hipLaunchKernelGGL(resize_v_kernel, dim3(bx, S, n), dim3(256), 0, st, w.jobs, S, KS, w.bounds, w.coeffs, w.tmp,
                       tmp_job_stride, out);
    REVO_HIP_CHECK(hipGetLastError());
    return 0;                       // asynchronous: an ingest queues batch after batch without waiting for the device
}

}  // namespace revo

extern "C" int32_t revo_preprocess_crop_resize(const revo_crop_job* jobs, int32_t n, int32_t out_size, uint8_t* out,
                                               void* stream) {
    try {
        REVO_REQUIRE(n >= 0, "crop_resize: negative job count");
        if (n == 0) return 0;
        REVO_REQUIRE(jobs != nullptr && out != nullptr, "crop_resize: null argument");
        REVO_REQUIRE(out_size > 0 && out_size <= 4096, "crop_resize: out_size must be in 1..4096");
        REVO_REQUIRE(n <= 65535, "crop_resize: at most 65535 jobs per call");
        return revo::crop_resize(jobs, n, out_size, out, (hipStream_t)stream);
    } catch (const std::exception& e) {
        revo_set_error(std::string("exception: ") + e.what());
        return -3;
    }
}
