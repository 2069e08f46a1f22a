// C ABI of librevo (include/revo.h): model and gallery handles, the embed forward
// schedule, the search pipeline, error plumbing and the per-kernel-class profiler.
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/revo.h"
#include "kernels.h"

// ------------------------------------------------------------ error state --
static thread_local std::string g_err;
void revo_set_error(const std::string& msg) { g_err = msg; }
extern "C" const char* revo_last_error(void) { return g_err.c_str(); }
extern "C" int32_t revo_version(void) { return 100; }

#define API_BEGIN try {
#define API_END                                            \
    }                                                      \
    catch (const std::exception& e) {                      \
        revo_set_error(std::string("exception: ") + e.what()); \
        return -3;                                         \
    }                                                      \
    catch (...) {                                          \
        revo_set_error("unknown exception");               \
        return -3;                                         \
    }
#define CHECK_RC(expr)            \
    do {                          \
        int _rc = (expr);         \
        if (_rc) return _rc;      \
    } while (0)

// A handle is bound to the device it was created on: every entry point that takes one makes that
// device current for the duration of the call (allocations, kernel attributes and launches all go
// to the current device) and puts the caller's device back afterwards.
namespace {
struct DeviceGuard {
    int prev = -1; bool switched = false; hipError_t err = hipSuccess;
    explicit DeviceGuard(int device) {
        err = hipGetDevice(&prev);
        if (err == hipSuccess && prev != device) { err = hipSetDevice(device); switched = err == hipSuccess; }
    }
    ~DeviceGuard() { if (switched) (void)hipSetDevice(prev); }
};
}  // namespace
#define REVO_ON_DEVICE(dev)                 \
    DeviceGuard dg_(dev);                   \
    REVO_HIP_CHECK(dg_.err)

extern "C" int32_t revo_sync(void* stream) {
    REVO_HIP_CHECK(hipStreamSynchronize((hipStream_t)stream));
    return 0;
}

// --------------------------------------------------------------- profiler --
namespace {
struct ProfRec { std::string cls; hipEvent_t a, b; };
struct Profiler {
    int on = 0;      // 0 off, 1 every kernel class, 2 only the body GEMMs (gemm_qkv / out / fc1 / fc2), 3 every 4th of those
    unsigned sample[4] = {0, 0, 0, 0};
    std::mutex mu;
    std::vector<ProfRec> recs;
    std::vector<hipEvent_t> pool;
    std::map<std::string, std::pair<long, double>> acc;
    hipEvent_t get() {
        if (!pool.empty()) { hipEvent_t e = pool.back(); pool.pop_back(); return e; }
        hipEvent_t e; (void)hipEventCreate(&e); return e;
    }
    void drain() {
        for (auto& r : recs) {
            float ms = 0.f;
            if (hipEventSynchronize(r.b) == hipSuccess && hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess) {
                auto& s = acc[r.cls];
                s.first += 1; s.second += ms;
            }
            pool.push_back(r.a); pool.push_back(r.b);
        }
        recs.clear();
    }
} g_prof;

struct ProfScope {
    bool active; hipStream_t st; ProfRec rec;
    ProfScope(const char* cls, hipStream_t s) : active(g_prof.on != 0), st(s) {
        if (active && g_prof.on >= 2) {
            const int which = !strcmp(cls, "gemm_qkv") ? 0 : !strcmp(cls, "gemm_out") ? 1 : !strcmp(cls, "gemm_fc1") ? 2 :
                              !strcmp(cls, "gemm_fc2") ? 3 : -1;
            active = which >= 0;
            // mode 3: every fourth launch of each class (all layers have the same shapes): a quarter of the event cost
            if (active && g_prof.on == 3) active = (g_prof.sample[which]++ & 3) == 0;
        }
        if (!active) return;
        std::lock_guard<std::mutex> lk(g_prof.mu);
        rec.cls = cls; rec.a = g_prof.get(); rec.b = g_prof.get();
        (void)hipEventRecord(rec.a, st);
    }
    ~ProfScope() {
        if (!active) return;
        (void)hipEventRecord(rec.b, st);
        std::lock_guard<std::mutex> lk(g_prof.mu);
        g_prof.recs.push_back(rec);
    }
};
}  // namespace

extern "C" int32_t revo_prof_enable(int32_t on) {
    g_prof.on = on < 0 ? 0 : (on > 3 ? 1 : on);
    for (auto& x : g_prof.sample) x = 0;
    return 0;
}
extern "C" int32_t revo_prof_reset(void) {
    std::lock_guard<std::mutex> lk(g_prof.mu);
    g_prof.drain();
    g_prof.acc.clear();
    return 0;
}
extern "C" int32_t revo_prof_report(char* buf, int32_t capacity) {
    std::lock_guard<std::mutex> lk(g_prof.mu);
    g_prof.drain();
    std::string s = "{";
    bool first = true;
    for (auto& kv : g_prof.acc) {
        if (!first) s += ", ";
        first = false;
        s += "\"" + kv.first + "\": {\"launches\": " + std::to_string(kv.second.first) +
             ", \"ms\": " + std::to_string(kv.second.second) + "}";
    }
    s += "}";
    if ((int)s.size() + 1 > capacity) { revo_set_error("prof_report: buffer too small"); return -2; }
    memcpy(buf, s.c_str(), s.size() + 1);
    return 0;
}

// ------------------------------------------------------------- small ops ---
namespace revo {
// q[o] = scale * (bias[o] + sum_i w[o][i] * probe[i])   -- the pool head's query is input independent
__global__ __launch_bounds__(256) void probe_q_kernel(const float* __restrict__ w, const float* __restrict__ bias,
                                                      const float* __restrict__ probe, int W, float scale,
                                                      float* __restrict__ q) {
    const int lane = threadIdx.x & 63;
    const int o = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (o >= W) return;
    float acc = 0.f;
    for (int i = lane; i < W; i += 64) acc = fmaf(w[(long)o * W + i], probe[i], acc);
    acc = wave_sum(acc);
    if (lane == 0) q[o] = (acc + bias[o]) * scale;
}
}  // namespace revo

// ------------------------------------------------------------ the model ----
struct DevBuf {
    void* p = nullptr;
    int alloc(size_t bytes) {
        REVO_HIP_CHECK(hipMalloc(&p, bytes ? bytes : 16));
        return 0;
    }
    void release() { if (p) { (void)hipFree(p); p = nullptr; } }
};

// ln_1 / ln_2 are folded into the weights of the GEMM that follows them: w_qkv = bf16(ln_1.weight . W),
// b_qkv = b + W ln_1.bias, c_qkv[j] = sum_k w_qkv[j][k]; the same for fc1 with ln_2.  The forward feeds those GEMMs either
// bf16(x) with the row statistics (folded form) or the normalised row without gain and shift (LayerNorm kernel).
struct LayerW {
    bf16_t *w_qkv, *w_o, *w_fc1, *w_fc2;
    float *b_qkv, *b_o, *b_fc1, *b_fc2, *c_qkv, *c_fc1, *ls1, *ls2;
};

constexpr size_t SPLITK_WS_ELEMS = 16u << 20;     // 64 MiB of fp32
struct revo_vit {
    revo_vit_cfg cfg;
    int device = 0, max_batch = 0, S = 0, G2 = 0, Kp = 0, hd = 0, phd = 0, PM = 0, debug_layers = -1;
    std::vector<void*> owned;
    // weights
    bf16_t* w_patch = nullptr; float *cls = nullptr, *pos = nullptr, *lnpre_w = nullptr, *lnpre_b = nullptr,
            *lnpost_w = nullptr, *lnpost_b = nullptr;
    std::vector<LayerW> layers;
    // attention-pool head, all fp32 (head.hip): probe query folded into the key projection (qk, ck), value / out / MLP /
    // proj weights as uploaded
    float *q_probe = nullptr, *qk = nullptr, *ck = nullptr, *w_v = nullptr, *b_v = nullptr, *w_po = nullptr, *b_po = nullptr;
    float *pln_w = nullptr, *pln_b = nullptr, *w_pfc1 = nullptr, *b_pfc1 = nullptr, *w_pfc2 = nullptr, *b_pfc2 = nullptr,
          *w_projT = nullptr;
    float2* rope_cs = nullptr;
    // workspace
    bf16_t *patches = nullptr, *h = nullptr, *qkv = nullptr, *att = nullptr, *mlp = nullptr;
    bf16_t* xlo = nullptr;             // low plane of the residual stream while it is kept as (h, xlo) = (bf16(x), bf16(x - hi))
    float *x = nullptr, *pool_logits = nullptr, *pool_u = nullptr, *pool_att = nullptr, *pool_o = nullptr, *pool_h = nullptr,
          *pool_m = nullptr, *feat = nullptr;
    float* splitk_ws = nullptr;        // fp32 partial planes of the split-K residual GEMMs
    float2* ln_stats = nullptr;        // [rows][width / 256] (mean, M2) per 256-column slice: the folded LayerNorm's row statistics
    int ln_parts = 0;                  // width / 256 when the fold applies (width % 256 == 0, <= 6 slices), else 0
    unsigned long long* ln_tele = nullptr;   // [4] telemetry of the folded LayerNorm's consumers (kernels.h GemmArgs::lnc_tele; revo_vit_stats)

    template <class T> int dalloc(T** out, size_t count) {
        void* p = nullptr;
        REVO_HIP_CHECK(hipMalloc(&p, count * sizeof(T) + 256));
        owned.push_back(p);
        *out = (T*)p;
        return 0;
    }
    ~revo_vit() {
        for (void* p : owned) (void)hipFree(p);
    }
};

namespace {
struct WeightMap {
    std::map<std::string, const revo_tensor*> m;
    const revo_tensor* get(const std::string& name, int64_t numel) {
        auto it = m.find(name);
        if (it == m.end()) { revo_set_error("missing weight tensor: " + name); return nullptr; }
        if (it->second->numel != numel) {
            revo_set_error("weight " + name + ": expected " + std::to_string(numel) + " elements, got " +
                           std::to_string(it->second->numel));
            return nullptr;
        }
        return it->second;
    }
};

// fp32 (host or device) -> device fp32
int up_f32(revo_vit* v, WeightMap& wm, const std::string& name, int64_t numel, float** out) {
    const revo_tensor* t = wm.get(name, numel);
    if (!t) return -2;
    CHECK_RC(v->dalloc(out, (size_t)numel));
    REVO_HIP_CHECK(hipMemcpy(*out, t->data, (size_t)numel * 4, hipMemcpyDefault));
    return 0;
}
// linear layer behind a LayerNorm: fp32 [rows][cols] weights, [cols] gain / shift, [rows] bias (host or device) ->
// bf16(gamma . W), column sums of the rounded weights, bias + W beta
int up_folded(revo_vit* v, WeightMap& wm, float* stage, float* gb, const std::string& wname, const std::string& bname,
              const std::string& ln, int64_t rows, int64_t cols, bf16_t** w_out, float** c_out, float** b_out) {
    const revo_tensor *tw = wm.get(wname, rows * cols), *tb = wm.get(bname, rows), *tg = wm.get(ln + ".weight", cols),
                      *te = wm.get(ln + ".bias", cols);
    if (!tw || !tb || !tg || !te) return -2;
    CHECK_RC(v->dalloc(w_out, (size_t)(rows * cols)));
    CHECK_RC(v->dalloc(c_out, (size_t)rows));
    CHECK_RC(v->dalloc(b_out, (size_t)rows));
    REVO_HIP_CHECK(hipMemcpy(stage, tw->data, (size_t)(rows * cols) * 4, hipMemcpyDefault));
    REVO_HIP_CHECK(hipMemcpy(gb, tg->data, (size_t)cols * 4, hipMemcpyDefault));
    REVO_HIP_CHECK(hipMemcpy(gb + cols, te->data, (size_t)cols * 4, hipMemcpyDefault));
    REVO_HIP_CHECK(hipMemcpy(gb + 2 * cols, tb->data, (size_t)rows * 4, hipMemcpyDefault));
    CHECK_RC(revo::launch_fold_ln_linear(stage, gb, gb + cols, gb + 2 * cols, (int)rows, (int)cols, *w_out, cols, *c_out, *b_out, 0));
    REVO_HIP_CHECK(hipStreamSynchronize(0));
    return 0;
}
// fp32 [rows][cols] (host or device) -> device bf16 [rows][ld] zero padded
int up_bf16(revo_vit* v, float* stage, const float* src, int64_t rows, int64_t cols, int64_t ld, bf16_t** out) {
    CHECK_RC(v->dalloc(out, (size_t)(rows * ld)));
    REVO_HIP_CHECK(hipMemcpy(stage, src, (size_t)(rows * cols) * 4, hipMemcpyDefault));
    CHECK_RC(revo::launch_f32_to_bf16(stage, cols, *out, ld, rows, (int)cols, 0));
    REVO_HIP_CHECK(hipStreamSynchronize(0));
    return 0;
}
}  // namespace

extern "C" int32_t revo_vit_create(const revo_vit_cfg* cfg, const revo_tensor* weights, int32_t n_weights,
                                   int32_t device, int32_t max_batch, revo_vit** out) {
    API_BEGIN
    REVO_REQUIRE(cfg && weights && out, "vit_create: null argument");
    REVO_REQUIRE(max_batch >= 1, "vit_create: max_batch must be >= 1");
    const revo_vit_cfg& c = *cfg;
    REVO_REQUIRE(c.image_size % c.patch_size == 0, "vit_create: image_size must be a multiple of patch_size");
    REVO_REQUIRE(c.width % c.heads == 0 && c.width % c.pool_heads == 0, "vit_create: width must divide into heads");
    REVO_REQUIRE(c.width % 64 == 0 && c.mlp_dim % 64 == 0, "vit_create: width and mlp_dim must be multiples of 64");
    REVO_REQUIRE(c.out_dim % 4 == 0, "vit_create: out_dim must be a multiple of 4");
    REVO_REQUIRE(c.width / c.heads == 64 || c.width / c.heads == 96,
                 "vit_create: body head_dim must be 64 (PE-Core B16 / L14) or 96 (G14)");
    REVO_REQUIRE(c.image_size > 0 && c.patch_size > 0 && c.layers >= 1 && c.heads >= 1 && c.pool_heads >= 1 && n_weights >= 0,
                 "vit_create: sizes must be positive");
    std::unique_ptr<revo_vit> v(new revo_vit());
    v->cfg = c; v->device = device; v->max_batch = max_batch;
#ifdef REVO_EXPERIMENTS
    if (const char* e = getenv("REVO_LN_FOLD")) revo::gemm_set_ln_fold(atoi(e));     // A/B runs of bench.py (scripts/): 0, 1 or 2
#endif
    const int W = c.width, M = c.mlp_dim, D = c.out_dim, P = c.patch_size, G = c.image_size / P;
    v->G2 = G * G; v->S = v->G2 + (c.use_cls ? 1 : 0);
    v->hd = W / c.heads; v->phd = W / c.pool_heads;
    // the pool head's MLP has its own width (upstream AttentionPooling: mlp_ratio 4), not the tower's mlp_dim
    const int PM = c.pool_mlp_dim > 0 ? c.pool_mlp_dim : 4 * W;
    REVO_REQUIRE(PM % 64 == 0, "vit_create: pool_mlp_dim must be a multiple of 64");
    v->PM = PM;
    const int Kreal = 3 * P * P;
    v->Kp = (Kreal + 63) / 64 * 64;
    const int S = v->S;

    // The checkpoint is validated as a whole before the device is touched: every tensor the architecture needs must
    // be present with the right element count (a checkpoint of another variant fails here, by name, not half-way
    // through the upload).
    WeightMap wm;
    for (int i = 0; i < n_weights; ++i) {
        REVO_REQUIRE(weights[i].name && weights[i].data, "vit_create: weight entry with a null name or data pointer");
        wm.m[weights[i].name] = &weights[i];
    }
    {
        std::vector<std::pair<std::string, int64_t>> need = {
            {"visual.conv1.weight", (int64_t)W * Kreal}, {"visual.positional_embedding", (int64_t)S * W},
            {"visual.ln_pre.weight", W}, {"visual.ln_pre.bias", W}, {"visual.ln_post.weight", W}, {"visual.ln_post.bias", W},
            {"visual.attn_pool.probe", W}, {"visual.attn_pool.attn.in_proj_weight", (int64_t)3 * W * W},
            {"visual.attn_pool.attn.in_proj_bias", 3 * W}, {"visual.attn_pool.attn.out_proj.weight", (int64_t)W * W},
            {"visual.attn_pool.attn.out_proj.bias", W}, {"visual.attn_pool.layernorm.weight", W},
            {"visual.attn_pool.layernorm.bias", W}, {"visual.attn_pool.mlp.c_fc.weight", (int64_t)PM * W},
            {"visual.attn_pool.mlp.c_fc.bias", PM}, {"visual.attn_pool.mlp.c_proj.weight", (int64_t)W * PM},
            {"visual.attn_pool.mlp.c_proj.bias", W}, {"visual.proj", (int64_t)W * D}};
        if (c.use_cls) need.push_back({"visual.class_embedding", W});
        for (int i = 0; i < c.layers; ++i) {
            const std::string p = "visual.transformer.resblocks." + std::to_string(i) + ".";
            for (const char* n : {"ln_1.weight", "ln_1.bias", "ln_2.weight", "ln_2.bias", "attn.out_proj.bias", "mlp.c_proj.bias"})
                need.push_back({p + n, W});
            need.push_back({p + "attn.in_proj_weight", (int64_t)3 * W * W});
            need.push_back({p + "attn.in_proj_bias", 3 * W});
            need.push_back({p + "attn.out_proj.weight", (int64_t)W * W});
            need.push_back({p + "mlp.c_fc.weight", (int64_t)M * W});
            need.push_back({p + "mlp.c_fc.bias", M});
            need.push_back({p + "mlp.c_proj.weight", (int64_t)W * M});
            if (c.use_ls) { need.push_back({p + "ls_1.gamma", W}); need.push_back({p + "ls_2.gamma", W}); }
        }
        for (auto& kv : need)
            if (!wm.get(kv.first, kv.second)) return -2;
        // ... and nothing else: a tensor the forward would not read (LayerScale gains handed to a handle created with
        // use_ls = 0, a tensor of another architecture) means a different model, not one to run without it
        if (wm.m.size() != need.size()) {
            std::map<std::string, int64_t> want(need.begin(), need.end());
            for (auto& kv : wm.m)
                if (!want.count(kv.first)) {
                    revo_set_error("unexpected weight tensor: " + kv.first + " (not read by this architecture" +
                                   (c.use_ls ? ")" : "; LayerScale needs cfg.use_ls = 1)"));
                    return -2;
                }
        }
    }
    REVO_ON_DEVICE(device);

    // staging buffer for fp32 -> bf16 conversion: the largest matrix
    size_t stage_elems = (size_t)std::max({(long)3 * W * W, (long)M * W, (long)PM * W, (long)W * Kreal, (long)W * D});
    float* stage = nullptr;
    REVO_HIP_CHECK(hipMalloc((void**)&stage, stage_elems * 4));
    struct StageGuard { float* p; ~StageGuard() { (void)hipFree(p); } } sg{stage};
    float* gb = nullptr;               // gain | shift | bias of the layer being folded
    REVO_HIP_CHECK(hipMalloc((void**)&gb, (size_t)(2 * W + std::max(3 * W, M)) * 4));
    StageGuard sg2{gb};

    const revo_tensor* t;
    if (!(t = wm.get("visual.conv1.weight", (int64_t)W * Kreal))) return -2;
    // split-precision patch embedding: rows ( hi | lo | hi ) of w / 255 (elementwise.hip patchify_kernel)
    CHECK_RC(v->dalloc(&v->w_patch, (size_t)W * 3 * v->Kp));
    REVO_HIP_CHECK(hipMemcpy(stage, t->data, (size_t)W * Kreal * 4, hipMemcpyDefault));
    CHECK_RC(revo::launch_split_hi_lo_hi(stage, W, Kreal, 1.0f / 255.0f, v->w_patch, v->Kp, 0));
    REVO_HIP_CHECK(hipStreamSynchronize(0));
    if (c.use_cls) CHECK_RC(up_f32(v.get(), wm, "visual.class_embedding", W, &v->cls));
    CHECK_RC(up_f32(v.get(), wm, "visual.positional_embedding", (int64_t)S * W, &v->pos));
    CHECK_RC(up_f32(v.get(), wm, "visual.ln_pre.weight", W, &v->lnpre_w));
    CHECK_RC(up_f32(v.get(), wm, "visual.ln_pre.bias", W, &v->lnpre_b));
    CHECK_RC(up_f32(v.get(), wm, "visual.ln_post.weight", W, &v->lnpost_w));
    CHECK_RC(up_f32(v.get(), wm, "visual.ln_post.bias", W, &v->lnpost_b));
    v->layers.resize(c.layers);
    for (int i = 0; i < c.layers; ++i) {
        const std::string p = "visual.transformer.resblocks." + std::to_string(i) + ".";
        LayerW& L = v->layers[i];
        memset(&L, 0, sizeof(L));
        CHECK_RC(up_folded(v.get(), wm, stage, gb, p + "attn.in_proj_weight", p + "attn.in_proj_bias", p + "ln_1", 3 * W, W,
                           &L.w_qkv, &L.c_qkv, &L.b_qkv));
        if (!(t = wm.get(p + "attn.out_proj.weight", (int64_t)W * W))) return -2;
        CHECK_RC(up_bf16(v.get(), stage, t->data, W, W, W, &L.w_o));
        CHECK_RC(up_f32(v.get(), wm, p + "attn.out_proj.bias", W, &L.b_o));
        CHECK_RC(up_folded(v.get(), wm, stage, gb, p + "mlp.c_fc.weight", p + "mlp.c_fc.bias", p + "ln_2", M, W,
                           &L.w_fc1, &L.c_fc1, &L.b_fc1));
        if (!(t = wm.get(p + "mlp.c_proj.weight", (int64_t)W * M))) return -2;
        CHECK_RC(up_bf16(v.get(), stage, t->data, W, M, M, &L.w_fc2));
        CHECK_RC(up_f32(v.get(), wm, p + "mlp.c_proj.bias", W, &L.b_fc2));
        if (c.use_ls) {
            CHECK_RC(up_f32(v.get(), wm, p + "ls_1.gamma", W, &L.ls1));
            CHECK_RC(up_f32(v.get(), wm, p + "ls_2.gamma", W, &L.ls2));
        }
    }
    // attention-pool head: fp32 weights; the probe's query and the key projection folded into qk / ck (head.hip)
    {
        const std::string p = "visual.attn_pool.";
        float *probe = nullptr, *wi = nullptr, *bi = nullptr;
        CHECK_RC(up_f32(v.get(), wm, p + "probe", W, &probe));
        CHECK_RC(up_f32(v.get(), wm, p + "attn.in_proj_weight", (int64_t)3 * W * W, &wi));
        CHECK_RC(up_f32(v.get(), wm, p + "attn.in_proj_bias", 3 * W, &bi));
        CHECK_RC(v->dalloc(&v->q_probe, (size_t)W));
        hipLaunchKernelGGL(revo::probe_q_kernel, dim3((W + 3) / 4), dim3(256), 0, 0, wi, bi, probe, W,
                           1.0f / sqrtf((float)v->phd), v->q_probe);
        REVO_HIP_CHECK(hipGetLastError());
        CHECK_RC(v->dalloc(&v->qk, (size_t)c.pool_heads * W));
        CHECK_RC(v->dalloc(&v->ck, (size_t)c.pool_heads));
        CHECK_RC(revo::launch_probe_qk(v->q_probe, wi + (size_t)W * W, bi + W, W, c.pool_heads, v->qk, v->ck, 0));
        v->w_v = wi + (size_t)2 * W * W;
        v->b_v = bi + 2 * W;
        REVO_HIP_CHECK(hipStreamSynchronize(0));
        CHECK_RC(up_f32(v.get(), wm, p + "attn.out_proj.weight", (int64_t)W * W, &v->w_po));
        CHECK_RC(up_f32(v.get(), wm, p + "attn.out_proj.bias", W, &v->b_po));
        CHECK_RC(up_f32(v.get(), wm, p + "layernorm.weight", W, &v->pln_w));
        CHECK_RC(up_f32(v.get(), wm, p + "layernorm.bias", W, &v->pln_b));
        CHECK_RC(up_f32(v.get(), wm, p + "mlp.c_fc.weight", (int64_t)PM * W, &v->w_pfc1));
        CHECK_RC(up_f32(v.get(), wm, p + "mlp.c_fc.bias", PM, &v->b_pfc1));
        CHECK_RC(up_f32(v.get(), wm, p + "mlp.c_proj.weight", (int64_t)W * PM, &v->w_pfc2));
        CHECK_RC(up_f32(v.get(), wm, p + "mlp.c_proj.bias", W, &v->b_pfc2));
    }
    if (!(t = wm.get("visual.proj", (int64_t)W * D))) return -2;
    CHECK_RC(v->dalloc(&v->w_projT, (size_t)D * W));
    REVO_HIP_CHECK(hipMemcpy(stage, t->data, (size_t)W * D * 4, hipMemcpyDefault));
    CHECK_RC(revo::launch_transpose_f32(stage, W, D, v->w_projT, 0));
    REVO_HIP_CHECK(hipStreamSynchronize(0));

    // 2-D rope table: [S][hd/2] (cos, sin); x-axis pairs first, then y-axis; cls row unrotated
    {
        const int hd = v->hd, d = hd / 2, nf = d / 2;
        std::vector<float2> cs((size_t)S * (hd / 2));
        const int start = c.use_cls ? 1 : 0;
        for (int s = 0; s < S; ++s) {
            for (int pi = 0; pi < hd / 2; ++pi) {
                double ang = 0.0;
                if (!(c.use_cls && s == 0)) {
                    const int g = s - (c.use_cls ? 1 : 0);
                    const int gy = g / G, gx = g % G;
                    const bool xaxis = pi < nf;
                    const int f = xaxis ? pi : pi - nf;
                    const double freq = 1.0 / pow((double)c.rope_theta, (double)(2 * f) / (double)d);
                    ang = (double)((xaxis ? gx : gy) + start) * freq;
                }
                cs[(size_t)s * (hd / 2) + pi] = make_float2((float)cos(ang), (float)sin(ang));
            }
        }
        CHECK_RC(v->dalloc(&v->rope_cs, cs.size()));
        REVO_HIP_CHECK(hipMemcpy(v->rope_cs, cs.data(), cs.size() * sizeof(float2), hipMemcpyHostToDevice));
    }

    // workspace for max_batch images
    const size_t rows = (size_t)max_batch * S, B = (size_t)max_batch;
    CHECK_RC(v->dalloc(&v->patches, (size_t)max_batch * v->G2 * 3 * v->Kp));
    CHECK_RC(v->dalloc(&v->x, rows * W));
    CHECK_RC(v->dalloc(&v->h, rows * W));
    CHECK_RC(v->dalloc(&v->qkv, rows * 3 * W));
    CHECK_RC(v->dalloc(&v->att, rows * W));
    CHECK_RC(v->dalloc(&v->mlp, rows * M));
    CHECK_RC(v->dalloc(&v->pool_logits, B * c.pool_heads * S));
    CHECK_RC(v->dalloc(&v->pool_u, B * c.pool_heads * W));
    CHECK_RC(v->dalloc(&v->pool_att, B * W));
    CHECK_RC(v->dalloc(&v->pool_o, B * W));
    CHECK_RC(v->dalloc(&v->pool_h, B * W));
    CHECK_RC(v->dalloc(&v->pool_m, B * PM));
    CHECK_RC(v->dalloc(&v->feat, B * D));
    CHECK_RC(v->dalloc(&v->splitk_ws, SPLITK_WS_ELEMS));
    v->ln_parts = (W % 256 == 0 && W / 256 <= 6) ? W / 256 : 0;
    if (v->ln_parts) CHECK_RC(v->dalloc(&v->ln_stats, rows * (size_t)v->ln_parts));
    if (v->ln_parts) CHECK_RC(v->dalloc(&v->xlo, rows * W));
    CHECK_RC(v->dalloc(&v->ln_tele, 4));
    REVO_HIP_CHECK(hipMemset(v->ln_tele, 0, 4 * sizeof(unsigned long long)));
    REVO_HIP_CHECK(hipDeviceSynchronize());
    *out = v.release();
    return 0;
    API_END
}

extern "C" int32_t revo_vit_destroy(revo_vit* vit) {
    API_BEGIN
    if (vit) { DeviceGuard dg(vit->device); delete vit; }
    return 0;
    API_END
}
extern "C" int32_t revo_vit_seq_len(const revo_vit* vit) { return vit ? vit->S : -1; }
extern "C" int32_t revo_vit_stats(revo_vit* vit, double* out4, int32_t reset, void* stream) {
    API_BEGIN
    REVO_REQUIRE(vit && out4, "vit_stats: null argument");
    REVO_ON_DEVICE(vit->device);
    unsigned long long c[4] = {0, 0, 0, 0};
    REVO_HIP_CHECK(hipStreamSynchronize((hipStream_t)stream));
    REVO_HIP_CHECK(hipMemcpy(c, vit->ln_tele, sizeof(c), hipMemcpyDeviceToHost));
    if (reset) REVO_HIP_CHECK(hipMemset(vit->ln_tele, 0, sizeof(c)));
    out4[0] = (double)c[0]; out4[1] = (double)c[1]; out4[2] = (double)c[2]; out4[3] = (double)revo::LNC_TELE_RATIO;
    return 0;
    API_END
}
#ifdef REVO_EXPERIMENTS   // parity-test hooks: librevo_exp.so only (include/revo.h)
extern "C" int32_t revo_vit_set_debug_layers(revo_vit* vit, int32_t n) {
    REVO_REQUIRE(vit, "null handle");
    vit->debug_layers = n;
    return 0;
}
extern "C" int32_t revo_vit_read_residual(revo_vit* vit, int32_t batch, float* dst, void* stream) {
    REVO_REQUIRE(vit && dst && batch >= 1 && batch <= vit->max_batch, "read_residual: bad arguments");
    REVO_ON_DEVICE(vit->device);
    REVO_HIP_CHECK(hipMemcpyAsync(dst, vit->x, (size_t)batch * vit->S * vit->cfg.width * 4, hipMemcpyDeviceToDevice,
                                  (hipStream_t)stream));
    return 0;
}

extern "C" int32_t revo_vit_read_tap(revo_vit* vit, int32_t which, int32_t batch, void* dst, void* stream) {
    REVO_REQUIRE(vit && dst && batch >= 1 && batch <= vit->max_batch, "read_tap: bad arguments");
    REVO_ON_DEVICE(vit->device);
    const size_t rows = (size_t)batch * vit->S, W = vit->cfg.width;
    const void* src = nullptr;
    size_t bytes = 0;
    switch (which) {
        case 0: src = vit->x; bytes = rows * W * 4; break;                  // fp32 residual stream
        case 1: src = vit->x; bytes = rows * W * 4; break;                  // fp32 ln_post output (in place in the residual buffer; after a whole forward)
        case 2: src = vit->pool_o; bytes = (size_t)batch * W * 4; break;    // fp32 attention-pool output (after its MLP residual, before proj)
        default: REVO_REQUIRE(false, "read_tap: which must be 0 (residual), 1 (ln_post output) or 2 (pooled)");
    }
    REVO_HIP_CHECK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return 0;
}
#endif

namespace {
// ln (optional, residual epilogue): the LayerNorm that follows; *ln_fused = 1 if the GEMM's launch form did it (kernels.h)
// The LayerNorm behind a residual GEMM, without gain and shift (those live in the next GEMM's weights).  Whichever the
// launch form can do: *fused = 1: `out` holds the normalised rows (one-image forms: the split-K reduce does it);
// *folded = 1: `out` holds bf16(x) and `stats` the rows' partial statistics (folded form, kernels.h GemmArgs::lnf_*);
// neither: the caller runs the LayerNorm kernel.
// planes: keep the residual stream as (out, lo) = (bf16(x), bf16(x - bf16(x))) from here on (kernels.h GemmArgs::xp_*)
struct LnAfter { float eps; bf16_t* out; long ldo; int* fused; float2* stats; int* folded; bf16_t* lo; int x_in_planes, planes_out; };
// consumer side of the folded form: A = bf16(x), the epilogue applies rstd (acc - mean c) + bias
struct LnBefore { const float2* stats; int parts; const float* c; float eps; };
int gemm(const char* cls, int epi, const bf16_t* A, long lda, const bf16_t* B, long ldb, int M, int N, int K, void* C,
         long ldc, const float* bias, const float* gamma, hipStream_t st, float* ws = nullptr, long ws_elems = 0,
         const LnAfter* ln = nullptr) {
    revo::GemmArgs a{};
    a.A = A; a.lda = lda; a.B = B; a.ldb = ldb; a.M = M; a.N = N; a.K = K; a.C = C; a.ldc = ldc;
    a.bias = bias; a.gamma = gamma; a.ws = ws; a.ws_elems = ws_elems;
    if (ln) {
        a.ln_eps = ln->eps; a.ln_ldo = ln->ldo; a.ln_fused = ln->fused;
        if (ln->fused) a.ln_out = ln->out;
        if (ln->stats) { a.lnf_xb = ln->out; a.lnf_ldxb = ln->ldo; a.lnf_stats = ln->stats; a.lnf_done = ln->folded; }
        if (ln->lo) { a.xp_hi = ln->out; a.xp_lo = ln->lo; a.xp_ld = ln->ldo; a.xp_in = ln->x_in_planes; a.xp_out = ln->planes_out; }
    }
    ProfScope ps(cls, st);
    return revo::launch_gemm(epi, a, st);
}
}  // namespace

// Forward of images [b0, b0 + B) of the call's batch on stream st; every workspace buffer is
// addressed at the rows of those images.
static int vit_forward_range(revo_vit* vv, const void* images_all, int32_t image_dtype, int b0, int B, float* out_all,
                             int32_t normalize, hipStream_t st) {
    const revo_vit_cfg& c = vv->cfg;
    const int W = c.width, Md = c.mlp_dim, D = c.out_dim, S = vv->S, PM = vv->PM;
    const int rows = B * S;
    using namespace revo;
    // views of the workspace at this range
    struct View {
        bf16_t *patches, *h, *qkv, *att, *mlp;
        float *x, *pool_logits, *pool_u, *pool_att, *pool_o, *pool_h, *pool_m, *feat;
    } w;
    const size_t r0 = (size_t)b0 * S;
    const int PH = c.pool_heads;
    w.patches = vv->patches + (size_t)b0 * vv->G2 * 3 * vv->Kp;
    w.x = vv->x + r0 * W; w.h = vv->h + r0 * W; w.qkv = vv->qkv + r0 * 3 * W; w.att = vv->att + r0 * W;
    w.mlp = vv->mlp + r0 * Md;
    w.pool_logits = vv->pool_logits + (size_t)b0 * PH * S; w.pool_u = vv->pool_u + (size_t)b0 * PH * W;
    w.pool_att = vv->pool_att + (size_t)b0 * W; w.pool_o = vv->pool_o + (size_t)b0 * W;
    w.pool_h = vv->pool_h + (size_t)b0 * W; w.pool_m = vv->pool_m + (size_t)b0 * PM;
    w.feat = vv->feat + (size_t)b0 * D;
    const size_t img_elems = (size_t)3 * c.image_size * c.image_size;
    const void* images = (const char*)images_all + (size_t)b0 * img_elems * (image_dtype == 1 ? 1 : 4);
    float* out = out_all + (size_t)b0 * D;
    struct FW {
        // names used by the schedule below: workspace from the view, everything else from the handle
        bf16_t *patches, *h, *qkv, *att, *mlp;
        float *x, *pool_logits, *pool_u, *pool_att, *pool_o, *pool_h, *pool_m, *feat;
        int Kp, G2, hd, phd, debug_layers;
        bf16_t* w_patch; float *pos, *cls, *lnpre_w, *lnpre_b, *lnpost_w, *lnpost_b;
        const std::vector<LayerW>& layers;
        float *qk, *ck, *w_v, *b_v, *w_po, *b_po, *pln_w, *pln_b, *w_pfc1, *b_pfc1, *w_pfc2, *b_pfc2, *w_projT;
        float2* rope_cs;
    } fw{w.patches, w.h, w.qkv, w.att, w.mlp, w.x, w.pool_logits, w.pool_u, w.pool_att, w.pool_o, w.pool_h, w.pool_m, w.feat,
         vv->Kp, vv->G2, vv->hd, vv->phd, vv->debug_layers, vv->w_patch, vv->pos, vv->cls, vv->lnpre_w, vv->lnpre_b,
         vv->lnpost_w, vv->lnpost_b, vv->layers, vv->qk, vv->ck, vv->w_v, vv->b_v, vv->w_po, vv->b_po, vv->pln_w,
         vv->pln_b, vv->w_pfc1, vv->b_pfc1, vv->w_pfc2, vv->b_pfc2, vv->w_projT, vv->rope_cs};
    const FW* v = &fw;

    {   // K1 + K2: patch embed GEMM in split precision (patchify_kernel: the row of a patch holds its values twice --
        // exact integers for u8 images -- or as hi | hi | lo; the weight rows are hi | lo | hi of w / 255), position add, cls row
        const int parts = image_dtype == 1 ? 2 : 3;
        { ProfScope ps("patchify", st);
          CHECK_RC(launch_patchify(images, image_dtype == 1, B, c.image_size, c.patch_size, v->patches, v->Kp, st)); }
        GemmArgs a{};
        a.A = v->patches; a.lda = (long)parts * v->Kp; a.B = v->w_patch; a.ldb = 3l * v->Kp; a.M = B * v->G2; a.N = W;
        a.K = parts * v->Kp;
        a.C = v->x; a.ldc = W; a.pos = v->pos; a.S = S; a.G2 = v->G2; a.cls = c.use_cls ? 1 : 0;
        { ProfScope ps("gemm_patch", st); CHECK_RC(launch_gemm(EPI_PATCH, a, st)); }
        if (c.use_cls) { ProfScope ps("elementwise", st); CHECK_RC(launch_cls_rows(v->x, W, v->cls, v->pos, B, S, W, st)); }
    }
    if (v->debug_layers == -2) return 0;   // parity hook: the embedded tokens (patch embed + position + class token), before ln_pre
    { ProfScope ps("layernorm", st);
      CHECK_RC(launch_layernorm(v->x, W, v->lnpre_w, v->lnpre_b, c.ln_eps, rows, W, v->x, W, 0, st)); }

    const int nl = v->debug_layers < 0 ? c.layers : std::min(v->debug_layers, c.layers);
    // ln_1 / ln_2 live in the weights of qkv / fc1 (LayerW): what those GEMMs read from `h` is, depending on what the
    // residual GEMM in front of them could do,
    //   H_XB    bf16(x) + the rows' partial statistics in ln_stats: the folded form (batch-sized forwards: the residual
    //           epilogue that holds the new row writes both; the consuming epilogue applies rstd (acc - mean c) + b'),
    //   H_NORM  the normalised rows (one-image forwards: the split-K reduce of the residual GEMM normalises its rows;
    //           otherwise the LayerNorm kernel, which is also what block 0's ln_1 takes behind ln_pre).
    enum { H_NONE = 0, H_NORM = 1, H_XB = 2 };
    int h_state = H_NONE;
    float2* stats = vv->ln_parts ? vv->ln_stats + r0 * vv->ln_parts : nullptr;
    // Between folded residual GEMMs the stream itself lives in two bf16 planes, (h, xlo) = (bf16(x), bf16(x - bf16(x))):
    // the high plane IS the next GEMM's A operand, so the residual epilogues move 4 + 4 bytes per element instead of fp32
    // rows plus a bf16 copy (4 + 4 + 2).  Entered by the first folding out-proj (fp32 in, planes out), left by the last
    // fc2 (planes in, fp32 out: ln_post and the head read fp32); only when both residual GEMMs of a block fold.
    bf16_t* xlo = (vv->xlo && gemm_ln_planes_enabled() && gemm_resid_folds(rows, W, W, W, W, W) && gemm_resid_folds(rows, W, Md, Md, Md, W))
                      ? vv->xlo + r0 * W : nullptr;
    bool x_planes = false;
    auto ln_before = [&](const float* csum, GemmArgs& a) {
        if (h_state == H_XB) { a.lnc_stats = stats; a.lnc_parts = vv->ln_parts; a.lnc_c = csum; a.lnc_eps = c.ln_eps; a.lnc_tele = vv->ln_tele; }
    };
    auto normalise_if_needed = [&]() -> int {
        if (h_state != H_NONE) return 0;
        if (x_planes) { revo_set_error("vit_forward: internal: a LayerNorm kernel was asked for while the stream is in planes"); return -3; }
        ProfScope ps("layernorm", st);
        CHECK_RC(launch_layernorm(v->x, W, nullptr, nullptr, c.ln_eps, rows, W, v->h, W, 1, st));
        h_state = H_NORM;
        return 0;
    };
    for (int i = 0; i < nl; ++i) {
        const LayerW& L = v->layers[i];
        CHECK_RC(normalise_if_needed());
        {
            // K4 + K5: bias and the 2-D rotary embedding of q and k in the GEMM epilogue (every tile shape)
            GemmArgs a{};
            a.A = v->h; a.lda = W; a.B = L.w_qkv; a.ldb = W; a.M = rows; a.N = 3 * W; a.K = W; a.C = v->qkv;
            a.ldc = 3 * W; a.bias = L.b_qkv; a.rope_cs = v->rope_cs; a.rope_S = S; a.rope_hd = v->hd;
            a.rope_cols = 2 * W;
            ln_before(L.c_qkv, a);
            ProfScope ps("gemm_qkv", st);
            CHECK_RC(launch_gemm(EPI_BF16_ROPE, a, st));
        }
        { ProfScope ps("attention", st);
          CHECK_RC(launch_attention_ex(v->qkv, 3 * W, v->att, W, B, S, c.heads, v->hd, c.use_cls, st)); }
        int fused = 0, folded = 0;
        const LnAfter ln2{c.ln_eps, v->h, W, &fused, stats, &folded, xlo, x_planes ? 1 : 0, xlo ? 1 : 0};
        CHECK_RC(gemm("gemm_out", EPI_RESID_F32, v->att, W, L.w_o, W, rows, W, W, v->x, W, L.b_o, L.ls1, st, vv->splitk_ws,
                      (long)SPLITK_WS_ELEMS, &ln2));
        h_state = folded ? H_XB : (fused ? H_NORM : H_NONE);
        x_planes = xlo && folded;
#ifdef REVO_EXPERIMENTS
        // cache-state experiment: re-touch bf16(x) (a device copy into the dead qkv buffer) before the GEMM that streams it
        if (folded && getenv("REVO_LNFOLD_TOUCH")) {
            ProfScope ps("touch", st);
            REVO_HIP_CHECK(hipMemcpyAsync(v->qkv, v->h, (size_t)rows * W * 2, hipMemcpyDeviceToDevice, st));
        }
#endif
        CHECK_RC(normalise_if_needed());
        {
            GemmArgs a{};
            a.A = v->h; a.lda = W; a.B = L.w_fc1; a.ldb = W; a.M = rows; a.N = Md; a.K = W; a.C = v->mlp; a.ldc = Md;
            a.bias = L.b_fc1;
            ln_before(L.c_fc1, a);
            ProfScope ps("gemm_fc1", st);
            CHECK_RC(launch_gemm(EPI_BF16_GELU, a, st));
        }
        // (splitk_ws: scratch for the split-K forms of fc2 -- leftover rows at large batch, the whole GEMM at small batch)
        fused = 0; folded = 0;
        const bool last = i + 1 >= nl;
        // the last fc2 has no LayerNorm of the body behind it: it brings the stream back to fp32 rows (ln_post, the taps)
        const LnAfter ln1n{c.ln_eps, v->h, W, last ? nullptr : &fused, last ? nullptr : stats, last ? nullptr : &folded, xlo,
                           x_planes ? 1 : 0, (xlo && !last) ? 1 : 0};
        CHECK_RC(gemm("gemm_fc2", EPI_RESID_F32, v->mlp, Md, L.w_fc2, Md, rows, W, Md, v->x, W, L.b_fc2, L.ls2, st,
                      vv->splitk_ws, (long)SPLITK_WS_ELEMS, (!last || x_planes) ? &ln1n : nullptr));
        h_state = last ? H_NONE : (folded ? H_XB : (fused ? H_NORM : H_NONE));
        x_planes = !last && xlo && folded;
#ifdef REVO_EXPERIMENTS
        if (folded && !last && getenv("REVO_LNFOLD_TOUCH")) {
            ProfScope ps("touch", st);
            REVO_HIP_CHECK(hipMemcpyAsync(v->qkv, v->h, (size_t)rows * W * 2, hipMemcpyDeviceToDevice, st));
        }
#endif
    }
    if (v->debug_layers >= 0) return 0;   // parity hook: residual stream only

    // ln_post (fp32, in place) -> attention pool -> proj -> normalise: all fp32 (head.hip says why)
    { ProfScope ps("layernorm", st);
      // (the pool's logits come out of the same kernel: the normalised row is in registers there)
      const LnLogits lg{v->qk, v->ck, v->pool_logits, PH, S};
      CHECK_RC(launch_layernorm(v->x, W, v->lnpost_w, v->lnpost_b, c.ln_eps, rows, W, v->x, W, 0, st, &lg)); }
    { ProfScope ps("pool_attention", st);
      CHECK_RC(launch_pool_head_rows(v->x, W, B, S, W, PH, v->pool_logits, v->pool_u, st)); }
    { ProfScope ps("gemm_pool", st);
      // values: head h's output columns from head h's pooled row
      CHECK_RC(launch_gemm_f32_skinny(0, v->pool_u, (long)PH * W, W, v->phd, v->w_v, W, v->b_v, B, W, W, v->pool_att, W, st));
      CHECK_RC(launch_gemm_f32_skinny(0, v->pool_att, W, 0, 0, v->w_po, W, v->b_po, B, W, W, v->pool_o, W, st)); }
    { ProfScope ps("layernorm", st);
      CHECK_RC(launch_layernorm(v->pool_o, W, v->pln_w, v->pln_b, c.ln_eps, B, W, v->pool_h, W, 0, st)); }
    float* feat = normalize ? v->feat : out;
    { ProfScope ps("gemm_pool", st);
      CHECK_RC(launch_gemm_f32_skinny(1, v->pool_h, W, 0, 0, v->w_pfc1, W, v->b_pfc1, B, PM, W, v->pool_m, PM, st));
      CHECK_RC(launch_gemm_f32_skinny(2, v->pool_m, PM, 0, 0, v->w_pfc2, PM, v->b_pfc2, B, W, PM, v->pool_o, W, st));
      CHECK_RC(launch_gemm_f32_skinny(0, v->pool_o, W, 0, 0, v->w_projT, W, nullptr, B, D, W, feat, D, st)); }
    if (normalize) {
        ProfScope ps("l2norm", st);
        CHECK_RC(launch_l2norm_rows(v->feat, D, out, D, nullptr, 0, B, D, st));
    }
    return 0;
}

extern "C" int32_t revo_vit_forward(revo_vit* v, const void* images, int32_t image_dtype, int32_t batch, float* out,
                                    int32_t normalize, void* stream) {
    API_BEGIN
    REVO_REQUIRE(v && images && out, "vit_forward: null argument");
    REVO_REQUIRE(batch >= 1 && batch <= v->max_batch, "vit_forward: batch exceeds max_batch of the handle");
    REVO_REQUIRE(image_dtype == 0 || image_dtype == 1, "vit_forward: image_dtype must be 0 (f32) or 1 (u8)");
    REVO_ON_DEVICE(v->device);
    hipStream_t st = (hipStream_t)stream;
    return vit_forward_range(v, images, image_dtype, 0, batch, out, normalize, st);
    API_END
}

// ------------------------------------------------------------ the gallery --
struct revo_gallery {
    int D = 0, device = 0, keep_f32 = 1;
    int64_t capacity = 0, size = 0;
    bf16_t* gb = nullptr;      // [capacity][D] normalised rows, scan copy
    float* gf = nullptr;       // [capacity][D] normalised rows, fp32 master (re-score + persistence)
    // per-call workspace, grown on demand
    float* qf = nullptr; bf16_t* qb = nullptr; uint32_t* tau0 = nullptr; int q_cap = 0;
    uint64_t* part = nullptr; size_t part_cap = 0;
    float* stage = nullptr; size_t stage_cap = 0;
    // candidates of the last scan (inside `part`), consumed by the finish step
    const uint64_t* cand = nullptr; long cand_stride = 0; int cand_Q = 0, cand_ksel = 0;
    // the last scan ran with an admission margin: its segments can answer uncertified queries (CertArgs, kernels.h)
    revo::SegSrc segs[2] = {}; int nsegs = 0; const uint64_t* prelist = nullptr;
    float* marg = nullptr; int* dropflag = nullptr;
    int64_t total_rows = 0;        // revo_search_set_total_rows: this handle is one shard of a gallery of that many rows (0: the whole)
    bool cand_estimated = false;   // the last scan started from estimated admission scores (CertArgs::estimated)
    // exactness certificate (kernels.h): per-query rounding norms of the last search's queries, running maxima over the
    // gallery's rows, the fallback workspace (sized with q_cap) and the handle's mode
    float* qstat = nullptr; uint32_t* gstat = nullptr;
    char* xbuf = nullptr; revo::ExactWs xw{};
    int mode = 0;
    const uint32_t* seed_bounds = nullptr;   // experiment build only (revo_debug_seed_bounds): admission bounds from outside
    ~revo_gallery() {
        (void)hipFree(gb); (void)hipFree(gf); (void)hipFree(qf); (void)hipFree(qb); (void)hipFree(part);
        (void)hipFree(tau0); (void)hipFree(marg); (void)hipFree(dropflag);
        (void)hipFree(stage);
        (void)hipFree(qstat); (void)hipFree(gstat); (void)hipFree(xbuf);
    }
    revo::CertArgs cert_args(float* cert_out) const {
        revo::CertArgs c{};
        c.qstat = qstat; c.gstat = gstat; c.mode = mode; c.ws = xw; c.Qb = qb; c.ldq = D; c.cert_out = cert_out;
        c.nsegs = nsegs; c.segs[0] = segs[0]; c.segs[1] = segs[1]; c.seg_ksel = cand_ksel; c.prelist = prelist;
        c.tau_base = tau0; c.marg = marg; c.dropflag = dropflag; c.estimated = cand_estimated ? 1 : 0;
        return c;
    }
};

extern "C" int32_t revo_gallery_create(int32_t dim, int64_t capacity, int32_t device, int32_t keep_f32,
                                       revo_gallery** out) {
    API_BEGIN
    REVO_REQUIRE(out, "gallery_create: null argument");
    REVO_REQUIRE(dim >= 64 && dim % 64 == 0, "gallery_create: dim must be a positive multiple of 64");
    REVO_REQUIRE(capacity >= 1 && capacity < (1ll << 32), "gallery_create: capacity must be in [1, 2^32)");
    REVO_ON_DEVICE(device);
    std::unique_ptr<revo_gallery> g(new revo_gallery());
    g->D = dim; g->device = device; g->capacity = capacity; g->keep_f32 = keep_f32 != 0;
    REVO_HIP_CHECK(hipMalloc((void**)&g->gb, (size_t)capacity * dim * 2));
    if (g->keep_f32) REVO_HIP_CHECK(hipMalloc((void**)&g->gf, (size_t)capacity * dim * 4));
    REVO_HIP_CHECK(hipMalloc((void**)&g->gstat, 8));
    REVO_HIP_CHECK(hipMemset(g->gstat, 0, 8));
    *out = g.release();
    return 0;
    API_END
}
extern "C" int32_t revo_gallery_destroy(revo_gallery* g) {
    API_BEGIN
    if (g) { DeviceGuard dg(g->device); delete g; }
    return 0;
    API_END
}
extern "C" int64_t revo_gallery_size(const revo_gallery* g) { return g ? g->size : -1; }
extern "C" int32_t revo_search_set_total_rows(revo_gallery* g, int64_t total_rows) {
    REVO_REQUIRE(g && total_rows >= 0, "search_set_total_rows: null handle or negative row count");
    g->total_rows = total_rows;
    return 0;
}
extern "C" int32_t revo_gallery_clear(revo_gallery* g) {
    REVO_REQUIRE(g, "null handle");
    REVO_ON_DEVICE(g->device);
    REVO_HIP_CHECK(hipMemset(g->gstat, 0, 8));        // the row maxima of the certificate start over with the rows
    g->size = 0;
    return 0;
}

extern "C" int32_t revo_gallery_append(revo_gallery* g, const float* vecs, int64_t n, int32_t normalize,
                                       int32_t src_on_device, void* stream) {
    API_BEGIN
    REVO_REQUIRE(g && (vecs || n == 0), "gallery_append: null argument");
    REVO_REQUIRE(n >= 0 && g->size + n <= g->capacity, "gallery_append: exceeds the capacity given at create");
    if (n == 0) return 0;
    REVO_ON_DEVICE(g->device);
    hipStream_t st = (hipStream_t)stream;
    const int D = g->D;
    const int64_t chunk_rows = std::max<int64_t>(1, (64ll << 20) / (D * 4));
    for (int64_t done = 0; done < n; done += chunk_rows) {
        const int64_t m = std::min(chunk_rows, n - done);
        const float* src = vecs + done * D;
        if (!src_on_device) {
            const size_t need = (size_t)m * D * 4;
            if (g->stage_cap < need) {
                (void)hipFree(g->stage); g->stage = nullptr; g->stage_cap = 0;
                REVO_HIP_CHECK(hipMalloc((void**)&g->stage, need));
                g->stage_cap = need;
            }
            REVO_HIP_CHECK(hipMemcpyAsync(g->stage, src, need, hipMemcpyHostToDevice, st));
            src = g->stage;
        }
        const int64_t row0 = g->size + done;
        bf16_t* db = g->gb + row0 * D;
        float* df = g->keep_f32 ? g->gf + row0 * D : nullptr;
        ProfScope ps("gallery_append", st);
        // one kernel either way: fp32 master row, bf16 scan row, and the row's share of the certificate's maxima
        // (max ||g||, max ||bf16(g) - g||; with normalize = 0 the rows are stored as given, whatever their length)
        CHECK_RC(revo::launch_l2norm_rows(src, D, df, D, db, D, m, D, st, normalize ? 1 : 0, nullptr, g->gstat));
        if (!src_on_device) REVO_HIP_CHECK(hipStreamSynchronize(st));   // staging buffer is reused
    }
    g->size += n;
    return 0;
    API_END
}

extern "C" int32_t revo_gallery_read(revo_gallery* g, int64_t start, int64_t n, float* dst, int32_t dst_on_device) {
    API_BEGIN
    REVO_REQUIRE(g && dst, "gallery_read: null argument");
    REVO_REQUIRE(g->keep_f32, "gallery_read: gallery was created without the fp32 master copy");
    REVO_REQUIRE(start >= 0 && n >= 0 && start + n <= g->size, "gallery_read: range outside the gallery");
    if (n == 0) return 0;
    REVO_ON_DEVICE(g->device);
    REVO_HIP_CHECK(hipMemcpy(dst, g->gf + start * g->D, (size_t)n * g->D * 4,
                             dst_on_device ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost));
    return 0;
    API_END
}

#ifdef REVO_EXPERIMENTS
// experiment (bound exchange between shards): raise the scan's admission bounds to values handed in
__global__ void seed_bounds_kernel(uint32_t* tau, const uint32_t* seed, int Q) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < Q && seed[i] > tau[i]) tau[i] = seed[i];
}
#endif
// Rows of the pre-pass of the 256 x 256 scan: about one round of GEMM tiles, at most a quarter of the gallery
static long search_prepass_rows(int Q, long N) {
#ifdef REVO_EXPERIMENTS
    // (sweep of the pre-pass size, scripts/: 2048 / 4096 / 8192 / 16384 rows at 10 000 queries -> 4.22 / 3.43 / 3.36 /
    //  3.50 ms per search of a 125 k-row shard: 8192 it is)
    if (const char* e = getenv("REVO_NPRE")) { const long v = atol(e) / 256 * 256; if (v >= 1024 && v <= N / 4) return v; }
#endif
    // (few queries, 1 M rows, whole search in ms at 1 / 64 queries: 8192 rows 0.557 / 0.600, 16 384 0.555 / 0.596, 32 768
    //  0.529 / 0.566, 65 536 0.564 / 0.596: a weaker seed costs the HBM-rate scan more survivors than the rows save)
    const long qtiles = (Q + 255) / 256;
    long n_pre = (65536 / qtiles) / 256 * 256;
    n_pre = n_pre > 32768 ? 32768 : (n_pre < 8192 ? 8192 : n_pre);
    if (n_pre > N / 4) n_pre = (N / 4) / 256 * 256;
    const long cap_pre = ((512l << 20) / (4l * Q)) / 256 * 256;
    if (n_pre > cap_pre) n_pre = cap_pre;
    if (n_pre < 1024) n_pre = 1024;
    return n_pre;
}
constexpr long SEARCH_SMALL_ROWS = 16384;     // below this the 128 x 128 scan with LDS lists takes the gallery
constexpr long SEARCH_WIDE_ROWS = 1l << 22;   // from here on the unsharded search keeps 64 candidates per query for every k

// Phase 1 of a search: normalise the queries, scan the gallery (bf16 MFMA scores) and leave each query's best
// ksel candidates, sorted best first, in the handle (cand / cand_stride).  The gallery must not be empty.
// z with P(standard normal > z) = p (Acklam's rational approximation, |relative error| < 1.2e-9; 0 < p < 0.5 here)
static double upper_normal_quantile(double p) {
    static const double a[] = {-3.969683028665376e+01, 2.209460984245205e+02, -2.759285104469687e+02, 1.383577518672690e+02,
                               -3.066479806614716e+01, 2.506628277459239e+00};
    static const double b[] = {-5.447609879822406e+01, 1.615858368580409e+02, -1.556989798598866e+02, 6.680131188771972e+01,
                               -1.328068155288572e+01};
    static const double c[] = {-7.784894002430293e-03, -3.223964580411365e-01, -2.400758277161838e+00, -2.549732539343734e+00,
                               4.374664141464968e+00, 2.938163982698783e+00};
    static const double d[] = {7.784695709041462e-03, 3.224671290700398e-01, 2.445134137142996e+00, 3.754408661907416e+00};
    if (p < 0.02425) {
        const double q = std::sqrt(-2.0 * std::log(p));
        return -(((((c[0] * q + c[1]) * q + c[2]) * q + c[3]) * q + c[4]) * q + c[5]) / ((((d[0] * q + d[1]) * q + d[2]) * q + d[3]) * q + 1.0);
    }
    const double q = (1.0 - p) - 0.5, r = q * q;
    return (((((a[0] * r + a[1]) * r + a[2]) * r + a[3]) * r + a[4]) * r + a[5]) * q /
           (((((b[0] * r + b[1]) * r + b[2]) * r + b[3]) * r + b[4]) * r + 1.0);
}

static int search_candidates(revo_gallery* g, const float* queries, int Q, int ksel, hipStream_t st,
                             uint32_t* bounds = nullptr, int top_m = 0, bool margin = false) {
    using namespace revo;
    const int D = g->D;
    const long N = g->size;
    if (g->q_cap < Q) {
        REVO_HIP_CHECK(hipStreamSynchronize(st));
        (void)hipFree(g->qf); (void)hipFree(g->qb); (void)hipFree(g->tau0); (void)hipFree(g->qstat); (void)hipFree(g->xbuf);
        (void)hipFree(g->marg); (void)hipFree(g->dropflag);
        g->qf = nullptr; g->qb = nullptr; g->tau0 = nullptr; g->qstat = nullptr; g->xbuf = nullptr; g->q_cap = 0;
        g->marg = nullptr; g->dropflag = nullptr;
        g->xw = ExactWs{};
        REVO_HIP_CHECK(hipMalloc((void**)&g->qf, (size_t)Q * D * 4));
        REVO_HIP_CHECK(hipMalloc((void**)&g->qb, (size_t)Q * D * 2));
        REVO_HIP_CHECK(hipMalloc((void**)&g->tau0, (size_t)Q * 4 * 2));   // pre-pass bounds | live bounds
        REVO_HIP_CHECK(hipMalloc((void**)&g->qstat, (size_t)Q * 8));
        REVO_HIP_CHECK(hipMalloc((void**)&g->marg, (size_t)Q * 4));
        REVO_HIP_CHECK(hipMalloc((void**)&g->dropflag, (size_t)Q * 4));
        if (g->keep_f32) {
            // fallback workspace of the exactness certificate: counters | unc_q | unc_lb | col_cnt | over_j | qb_u | col
            auto up256 = [](size_t x) { return (x + 255) / 256 * 256; };
            const size_t o_q = 256, o_lb = o_q + up256((size_t)Q * 4), o_cnt = o_lb + up256((size_t)Q * 4),
                         o_over = o_cnt + up256((size_t)Q * 4), o_orow = o_over + up256((size_t)Q * 4),
                         o_qb = o_orow + up256((size_t)Q * 4),
                         o_col = o_qb + up256((size_t)Q * D * 2), total = o_col + (size_t)Q * EXACT_COL_CAP * 8;
            REVO_HIP_CHECK(hipMalloc((void**)&g->xbuf, total));
            REVO_HIP_CHECK(hipMemsetAsync(g->xbuf, 0, 256, st));
            g->xw.ctr = (int*)g->xbuf; g->xw.unc_q = (int*)(g->xbuf + o_q); g->xw.unc_lb = (float*)(g->xbuf + o_lb);
            g->xw.col_cnt = (int*)(g->xbuf + o_cnt); g->xw.over_j = (int*)(g->xbuf + o_over);
            g->xw.qb_u = (bf16_t*)(g->xbuf + o_qb); g->xw.ldqb = D; g->xw.col = (uint64_t*)(g->xbuf + o_col);
            g->xw.cap = Q; g->xw.orow = (int*)(g->xbuf + o_orow);
        }
        g->q_cap = Q;
    }
    auto need_part = [&](size_t bytes) -> int {
        if (g->part_cap < bytes) {
            REVO_HIP_CHECK(hipStreamSynchronize(st));
            (void)hipFree(g->part); g->part = nullptr; g->part_cap = 0;
            REVO_HIP_CHECK(hipMalloc((void**)&g->part, bytes));
            g->part_cap = bytes;
        }
        return 0;
    };
    g->cand = nullptr; g->cand_Q = 0; g->nsegs = 0; g->prelist = nullptr; g->cand_estimated = false;
    // the admission margin only pays where the certificate is expected to fail (see revo_search_topk) and only the
    // 256 x 256 scan has segments; it needs the fp32 rows (no certificate without them)
    margin = margin && g->keep_f32 && N >= SEARCH_SMALL_ROWS;
    // query normalisation (+ rounding norms); the same kernel clears the certificate's counters and -- 256 x 256 scan -- the
    // score histograms (two memset launches fewer per search: a quarter of a one-query search is launches)
    auto prep = [&](uint32_t* hist, long hist_words) -> int {
        ProfScope ps("search_prep", st);
        CHECK_RC(launch_l2norm_rows(queries, D, g->qf, D, g->qb, D, Q, D, st, 1, g->qstat, nullptr,
                                    g->xbuf ? (uint32_t*)g->xw.ctr : nullptr, g->xbuf ? 8 : 0, hist, hist_words));
        if (margin) CHECK_RC(launch_cert_margin(g->qstat, g->gstat, D, Q, g->marg, g->dropflag, st));
        return 0;
    };

    if (N >= SEARCH_SMALL_ROWS) {
        // ---- 256 x 256 scan.  Pre-pass: a plain GEMM of the queries against the first n_pre rows and a
        // per-row selection seed the admission scores; the fused scan covers rows [n_pre, N).
        // Pre-pass size: about one round of 256 x 256 GEMM tiles; with few queries (short gallery slices
        // per CU) up to 32 k rows, which seed the 0.1 % quantile.
        const long n_pre = search_prepass_rows(Q, N);
        // the scan runs as one launch, or as a main launch of whole query tiles plus one for a ragged tail of queries
        const int q_main = topk_scan256_main_queries(Q, N - n_pre);
        struct Part { int q0, nq, splits; size_t seg_off, cnt_off; } parts[2] = {{0, q_main, 0, 0, 0}, {q_main, Q - q_main, 0, 0, 0}};
        const int nparts = q_main < Q ? 2 : 1;
        const int NB = topk_scan256_hist_buckets();
        // workspace: histograms (zeroed) | per part: segment counts, segments | pre-pass lists | final lists | pre-pass scores
        auto up256 = [](size_t x) { return (x + 255) / 256 * 256; };
        const size_t hist_off = 0, hist_bytes = (size_t)Q * NB * 4;
        size_t off = up256(hist_off + hist_bytes);
        for (int i = 0; i < nparts; ++i) {
            Part& pt = parts[i];
            pt.splits = topk_scan256_splits(pt.nq, N - n_pre);
            pt.cnt_off = off; off = up256(off + (size_t)pt.nq * pt.splits * 4);
            pt.seg_off = off; off = up256(off + (size_t)pt.nq * pt.splits * (2 * ksel) * 8);
        }
        const size_t pre_off = off, pre_bytes = (size_t)Q * ksel * 8;
        const size_t fin_off = up256(pre_off + pre_bytes), fin_bytes = (size_t)Q * ksel * 8;
        const size_t sco_off = up256(fin_off + fin_bytes);
        CHECK_RC(need_part(sco_off + (size_t)Q * n_pre * 4));
        char* wsb = (char*)g->part;
        uint32_t* hist = (uint32_t*)(wsb + hist_off);
        uint64_t* prelist = (uint64_t*)(wsb + pre_off);
        uint64_t* final_lists = (uint64_t*)(wsb + fin_off);
        float* pre_scores = (float*)(wsb + sco_off);
        uint32_t* tau_base = g->tau0, *tau_live = g->tau0 + g->q_cap;
        CHECK_RC(prep(hist, (long)(hist_bytes / 4)));
        {
            ProfScope ps("topk_prepass", st);
            GemmArgs ga{};
            ga.A = g->qb; ga.lda = D; ga.B = g->gb; ga.ldb = D; ga.M = Q; ga.N = (int)n_pre; ga.K = D;
            ga.C = pre_scores; ga.ldc = n_pre; ga.prefer256 = 1;
            // (few queries: the skinny form instead -- measured: pre-pass 48 -> 37 us at one query, 50 -> 57 at 64; nothing in the search)
            // Up to 128 queries: 128 x 64 tiles on the six-deep ring (two rounds of workgroups with five K-steps of loads in
            // flight; one 256 x 256 tile per CU is a chain of sixteen DMA latencies): pre-pass 48 -> 40 us at one query,
            // 49 -> 42 at 64, 52 -> 46 at 128
            bool ring = Q <= 128 && n_pre % 64 == 0;
#ifdef REVO_EXPERIMENTS
            if (getenv("REVO_PREPASS_256")) ring = false;
#endif
            if (ring) CHECK_RC(launch_gemm_f32_ring(ga, st));
            else CHECK_RC(launch_gemm(EPI_F32, ga, st));
            // One shard of a larger gallery, in the two-phase search: what its candidates have to reach is decided by ALL
            // shards' rows (the finish step re-scores only candidates among the best min(64, 2 ksel) of the whole gallery),
            // but its scan can only learn its own rows' scores -- at an eighth of the rows its admission bound sits at an
            // 8 x higher quantile and a third of its tile fragments still hold a survivor.  So it
            // starts from an ESTIMATE of the whole gallery's level, extrapolated from its own pre-pass scores
            // (topk_select_rows_kernel); an estimate, not a bound: the protocol's certificate and second round cover it.
            float est_z = 0.f;
            if (bounds && !margin && g->total_rows > N) {
                int j = 2 * ksel < 64 ? 2 * ksel : 64;
#ifdef REVO_EXPERIMENTS
                if (const char* e = getenv("REVO_EST_J")) j = atoi(e) > 0 ? atoi(e) : j;      // sweep of the estimate's rank (scripts/)
#endif
                est_z = (float)upper_normal_quantile((double)j / (double)g->total_rows);
                g->cand_estimated = true;
            }
            CHECK_RC(launch_topk_select_rows(pre_scores, n_pre, (int)n_pre, Q, prelist, ksel, 0, tau_base, ksel, hist, NB,
                                             topk_scan256_hist_shift(), st, tau_live, est_z));
#ifdef REVO_EXPERIMENTS
            if (g->seed_bounds) hipLaunchKernelGGL(seed_bounds_kernel, dim3((Q + 255) / 256), dim3(256), 0, st, tau_live, g->seed_bounds, Q);
#endif
        }
        { ProfScope ps("topk_scan", st);
          for (int i = 0; i < nparts; ++i) {
              const Part& pt = parts[i];
              CHECK_RC(launch_topk_scan256(g->qb + (size_t)pt.q0 * D, D, g->gb, D, pt.nq, N, D, n_pre, pt.splits,
                                           (uint64_t*)(wsb + pt.seg_off), (int*)(wsb + pt.cnt_off), tau_live + pt.q0,
                                           tau_base + pt.q0, hist + (size_t)pt.q0 * NB, ksel, st,
                                           margin ? g->marg + pt.q0 : nullptr, margin ? g->dropflag + pt.q0 : nullptr));
              if (margin) g->segs[i] = SegSrc{(const uint64_t*)(wsb + pt.seg_off), (const int*)(wsb + pt.cnt_off), pt.splits, pt.q0, pt.nq};
          }
          if (margin) { g->nsegs = nparts; g->prelist = prelist; } }
        { ProfScope ps("topk_reduce", st);
          for (int i = 0; i < nparts; ++i) {
              const Part& pt = parts[i];
              CHECK_RC(launch_topk_reduce_segs((const uint64_t*)(wsb + pt.seg_off), (const int*)(wsb + pt.cnt_off), pt.splits,
                                               prelist + (size_t)pt.q0 * ksel, final_lists + (size_t)pt.q0 * ksel, pt.nq, ksel,
                                               st, bounds ? bounds + (size_t)pt.q0 * top_m : nullptr, top_m));
          } }
        g->cand = final_lists; g->cand_stride = ksel;
    } else {
        // ---- small galleries: 128 x 128 scan with per-wave LDS lists
        const int splits = topk_scan_workspace_splits(Q, N);
        CHECK_RC(need_part((size_t)Q * splits * ksel * 8));
        CHECK_RC(prep(nullptr, 0));
        ScanArgs a{};
        a.Qb = g->qb; a.ldq = D; a.Gb = g->gb; a.ldg = D; a.Q = Q; a.N = N; a.D = D; a.ksel = ksel;
        a.splits = splits; a.part = g->part;
        { ProfScope ps("topk_scan", st); CHECK_RC(launch_topk_scan(a, st)); }
        { ProfScope ps("topk_reduce", st); CHECK_RC(launch_topk_reduce(g->part, Q, splits, ksel, st)); }
        g->cand = g->part; g->cand_stride = (long)splits * ksel;
        if (bounds) { ProfScope ps("topk_bounds", st); CHECK_RC(launch_topk_publish(g->cand, g->cand_stride, Q, top_m, bounds, st)); }
    }
    g->cand_Q = Q; g->cand_ksel = ksel;
    return 0;
}

// The fallback of the exactness certificate for the entries in the handle's workspace (count on the device): collect
// pass, exact re-score of what it collected, brute force for the entries whose lists overflowed.  Idle passes cost a
// few microseconds each.
static int search_fallback(revo_gallery* g, int max_entries, int k, int has_thr, float thr, long index_offset, int out_compact,
                           float* scores, long long* indices, int* counts, hipStream_t st) {
    using namespace revo;
    ProfScope ps("topk_exact", st);
    if (g->mode != 2) {
        Collect256Args ca{};
        ca.Qb = g->xw.qb_u; ca.ldq = g->xw.ldqb; ca.Gb = g->gb; ca.ldg = g->D; ca.N = g->size; ca.D = g->D;
        ca.n_q = g->xw.ctr; ca.lb = g->xw.unc_lb; ca.cnt = g->xw.col_cnt; ca.col = g->xw.col; ca.cap = EXACT_COL_CAP;
        CHECK_RC(launch_topk_collect256(ca, max_entries, st));
    }
    CHECK_RC(launch_topk_exact_finish(g->xw, max_entries, g->qf, g->D, g->gf, g->D, g->D, k, has_thr, thr, index_offset,
                                      g->mode == 2, out_compact, scores, indices, counts, st));
    return launch_topk_exact_bruteforce(g->xw, max_entries, g->qf, g->D, g->gf, g->D, g->size, g->D, k, has_thr, thr,
                                        index_offset, out_compact, scores, indices, counts, st);
}

// over-selection: the bf16 scan keeps ksel >= k + margin candidates, the fp32 re-score decides
static int search_ksel(int k) { return (k <= 16) ? 32 : 64; }
extern "C" int32_t revo_search_ksel(int32_t k) { return (k >= 1 && k <= 50) ? search_ksel(k) : -1; }
// k > 25: the scan runs with the admission margin (revo_search_topk); a shard's two-phase scan estimates the whole gallery's
// admission level only without it -- one rule, asked for by the host side that decides whether to exchange bounds
static bool search_uses_margin(int k) { return k > 25; }
extern "C" int32_t revo_search_estimates(int32_t k) { return (k >= 1 && k <= 50) ? (search_uses_margin(k) ? 0 : 1) : -1; }
extern "C" int32_t revo_search_plan(const revo_gallery* g, int32_t Q, int32_t k, int64_t* out4) {
    REVO_REQUIRE(g && out4 && Q >= 1 && k >= 1 && k <= 50, "search_plan: bad arguments");
    const long N = g->size;
    const bool big = N >= SEARCH_SMALL_ROWS;
    const long n_pre = big ? search_prepass_rows(Q, N) : 0;
    out4[0] = big ? 1 : 0;
    out4[1] = n_pre;
    out4[2] = big ? revo::topk_scan256_splits(revo::topk_scan256_main_queries(Q, N - n_pre), N - n_pre)
                  : revo::topk_scan_workspace_splits(Q, N);
    out4[3] = N >= SEARCH_WIDE_ROWS ? 64 : search_ksel(k);
    return 0;
}

extern "C" int32_t revo_search_topk(revo_gallery* g, const float* queries, int32_t Q, int32_t k, int32_t has_thr,
                                    float thr, int64_t index_offset, float* scores, int64_t* indices, int32_t* counts,
                                    void* stream) {
    API_BEGIN
    REVO_REQUIRE(g && scores && indices && counts && (queries || Q == 0), "search: null argument");
    REVO_REQUIRE(Q >= 0, "search: negative query count");
    REVO_REQUIRE(k >= 1 && k <= 50, "search: k must be in [1, 50]");
    if (Q == 0) return 0;
    REVO_ON_DEVICE(g->device);
    hipStream_t st = (hipStream_t)stream;
    using namespace revo;
    if (g->size == 0) return launch_topk_fill_empty(scores, (long long*)indices, counts, Q, k, st);
    // candidates per query: 32 for k <= 16, 64 beyond -- and 64 on very large galleries whatever k is: there a query
    // that fails its certificate costs a whole extra pass over the gallery (10 M x 1536: 5.9 ms next to a 7.7 ms scan),
    // and the wider list all but rules that out (the k-th to 64th score gap is 1.6 x the k-th to 32nd) for 0.1 ms of re-scores
    int ksel = g->size >= SEARCH_WIDE_ROWS ? 64 : search_ksel(k);
#ifdef REVO_EXPERIMENTS
    if (const char* e = getenv("REVO_KSEL")) ksel = atoi(e) == 64 ? 64 : ksel;         // candidate-list width study (scripts/)
#endif
    // k > 25: 64 candidates leave the certificate less room than its error bound on ordinary data (the 50th-to-64th score
    // gap of a random 1 M gallery is half of eps), so nearly every query fails it.  The scan then runs with an admission
    // margin of 2 eps: what an uncertified query needs is in its segments, and no second pass over the gallery is made
    // (Not for smaller k, not even on very large galleries where a second pass costs most of a search: measured on 10 M x
    //  1536 with 256 queries, the margin's extra survivors cost the scan 18 % on EVERY search -- 7.5 -> 8.9 ms -- to save a
    //  pass that the 64-candidate lists kept there for every k already make a rarity.)
    CHECK_RC(search_candidates(g, queries, Q, ksel, st, nullptr, 0, search_uses_margin(k)));
    const CertArgs ca = g->cert_args(nullptr);
    { ProfScope ps("topk_finish", st);
      CHECK_RC(launch_topk_finish(g->cand, g->cand_stride, ksel, g->qf, g->D, g->keep_f32 ? g->gf : nullptr, g->D, g->D, Q, k,
                                  has_thr, thr, index_offset, nullptr, 0, 0, scores, (long long*)indices, counts,
                                  g->keep_f32 ? &ca : nullptr, st)); }
    if (g->keep_f32 && g->mode != 3)
        CHECK_RC(search_fallback(g, Q, k, has_thr, thr, index_offset, 0, scores, (long long*)indices, counts, st));
    return 0;
    API_END
}

// ---- the same search in two phases, for a gallery that is row-sharded over several GPUs (include/revo.h)
extern "C" int32_t revo_search_candidates(revo_gallery* g, const float* queries, int32_t Q, int32_t k, int32_t top_m,
                                          uint32_t* bounds, void* stream) {
    API_BEGIN
    REVO_REQUIRE(g && bounds && (queries || Q == 0), "search_candidates: null argument");
    REVO_REQUIRE(Q >= 0, "search_candidates: negative query count");
    REVO_REQUIRE(k >= 1 && k <= 50, "search_candidates: k must be in [1, 50]");
    const int ksel = search_ksel(k);
    REVO_REQUIRE(top_m >= 1 && top_m <= ksel, "search_candidates: top_m must be in [1, revo_search_ksel(k)]");
    if (Q == 0) return 0;
    REVO_ON_DEVICE(g->device);
    hipStream_t st = (hipStream_t)stream;
    g->cand = nullptr; g->cand_Q = Q; g->cand_ksel = ksel;
    if (g->size == 0) {                                   // an empty shard publishes nothing
        REVO_HIP_CHECK(hipMemsetAsync(bounds, 0, (size_t)Q * top_m * 4, st));
        return 0;
    }
    // (the published scores come out of the final selection kernel: no launch of their own; k > 25: with the admission
    //  margin, so that the second round the merge's certificate then asks for needs no pass over the shard -- revo_search_topk)
    return search_candidates(g, queries, Q, ksel, st, bounds, top_m, search_uses_margin(k));
    API_END
}
extern "C" int32_t revo_search_finish(revo_gallery* g, int32_t Q, int32_t k, int32_t has_thr, float thr,
                                      int64_t index_offset, const uint32_t* all_bounds, int32_t parts, int32_t top_m,
                                      float* scores, int64_t* indices, int32_t* counts, float* cert, void* stream) {
    API_BEGIN
    REVO_REQUIRE(g && scores && indices && counts, "search_finish: null argument");
    REVO_REQUIRE(k >= 1 && k <= 50 && Q >= 0, "search_finish: bad k or query count");
    REVO_REQUIRE(Q == g->cand_Q && search_ksel(k) == g->cand_ksel,
                 "search_finish: no matching revo_search_candidates call on this handle");
    REVO_REQUIRE(!all_bounds || (parts >= 1 && top_m >= 1 && top_m <= g->cand_ksel), "search_finish: bad bounds layout");
    // A scan that started from an ESTIMATED admission level (revo_search_set_total_rows) has dropped rows on the strength of
    // that estimate: only the certificate (cert -> revo_topk_merge_packed -> revo_search_exact) makes the result exhaustive
    REVO_REQUIRE(cert || !g->cand_estimated,
                 "search_finish: this shard scanned against an estimated admission level (revo_search_set_total_rows): cert must "
                 "be given and checked by revo_topk_merge_packed, with revo_search_exact as the second round");
    if (Q == 0) return 0;
    REVO_ON_DEVICE(g->device);
    hipStream_t st = (hipStream_t)stream;
    using namespace revo;
    if (g->size == 0 || !g->cand) {
        // an empty shard holds no row that could change a result: its certificate bound is -inf
        if (cert) REVO_HIP_CHECK(hipMemsetD32Async((hipDeviceptr_t)cert, (int)0xff800000u, (size_t)Q, st));
        return launch_topk_fill_empty(scores, (long long*)indices, counts, Q, k, st);
    }
    ProfScope ps("topk_finish", st);
    const CertArgs ca = g->cert_args(cert);
    return launch_topk_finish(g->cand, g->cand_stride, g->cand_ksel, g->qf, g->D, g->keep_f32 ? g->gf : nullptr, g->D, g->D,
                              Q, k, has_thr, thr, index_offset, all_bounds, parts, top_m, scores, (long long*)indices,
                              counts, cert ? &ca : nullptr, st);
    API_END
}

// Second round of a row-sharded search: exact local results for the queries the merge step could not certify.
extern "C" int32_t revo_search_exact(revo_gallery* g, int32_t n, const int32_t* q_idx, const float* need, int32_t k,
                                     int32_t has_thr, float thr, int64_t index_offset, float* scores, int64_t* indices,
                                     int32_t* counts, void* stream) {
    API_BEGIN
    REVO_REQUIRE(g && scores && indices && counts && (n == 0 || (q_idx && need)), "search_exact: null argument");
    REVO_REQUIRE(k >= 1 && k <= 50 && n >= 0, "search_exact: bad k or entry count");
    REVO_REQUIRE(n <= g->cand_Q, "search_exact: more entries than queries in the last revo_search_candidates call");
    if (n == 0) return 0;
    REVO_ON_DEVICE(g->device);
    hipStream_t st = (hipStream_t)stream;
    using namespace revo;
    if (g->size == 0) return launch_topk_fill_empty(scores, (long long*)indices, counts, n, k, st);
    REVO_REQUIRE(g->keep_f32 && g->xbuf, "search_exact: the gallery was created without the fp32 master copy");
    const CertArgs ca = g->cert_args(nullptr);
    REVO_HIP_CHECK(hipMemsetAsync(g->xw.ctr, 0, 32, st));
    CHECK_RC(launch_topk_exact_prepare(g->xw, q_idx, need, n, ca, g->D, g->cand, g->cand_stride, g->cand_ksel, st));
    return search_fallback(g, n, k, has_thr, thr, index_offset, 1, scores, (long long*)indices, counts, st);
    API_END
}

#ifdef REVO_EXPERIMENTS   // librevo.so cannot be put into a non-exact mode
extern "C" int32_t revo_search_set_mode(revo_gallery* g, int32_t mode) {
    REVO_REQUIRE(g && mode >= 0 && mode <= 3, "search_set_mode: mode must be 0..3");
    g->mode = mode;
    return 0;
}
#endif
extern "C" int32_t revo_search_stats(revo_gallery* g, int32_t* out8, void* stream) {
    API_BEGIN
    REVO_REQUIRE(g && out8, "search_stats: null argument");
    int32_t* out4 = out8;
    for (int i = 0; i < 8; ++i) out8[i] = 0;
    if (!g->xbuf) { out4[0] = -1; return 0; }       // no fp32 master rows (or no search yet): nothing was certified
    REVO_ON_DEVICE(g->device);
    int c[8];
    REVO_HIP_CHECK(hipStreamSynchronize((hipStream_t)stream));
    REVO_HIP_CHECK(hipMemcpy(c, g->xw.ctr, sizeof(c), hipMemcpyDeviceToHost));
    out4[0] = c[0] + c[4] + c[5]; out4[1] = c[1]; out4[2] = c[2]; out4[3] = c[3]; out8[4] = c[5];
    return 0;
    API_END
}

extern "C" int32_t revo_topk_merge(const float* scores, const int64_t* indices, int32_t parts, int32_t Q, int32_t k,
                                   int32_t has_thr, float thr, float* out_scores, int64_t* out_indices,
                                   int32_t* out_counts, void* stream) {
    API_BEGIN
    REVO_REQUIRE(scores && indices && out_scores && out_indices && out_counts, "merge: null argument");
    ProfScope ps("topk_merge", (hipStream_t)stream);
    return revo::launch_topk_merge(scores, (const long long*)indices, parts, Q, k, has_thr, thr, out_scores,
                                   (long long*)out_indices, out_counts, (hipStream_t)stream);
    API_END
}

extern "C" int64_t revo_topk_packed_bytes(int32_t Q, int32_t k) {
    return Q < 0 || k < 1 ? -1 : (((int64_t)Q * k * 12 + (int64_t)Q * 4 + 15) / 16) * 16;
}
extern "C" int32_t revo_topk_merge_packed(const void* packed, int32_t parts, int32_t Q, int32_t k, int32_t has_thr, float thr,
                                          float* out_scores, int64_t* out_indices, int32_t* out_counts,
                                          int32_t* unc_count, int32_t* unc_q, float* unc_need, void* stream) {
    API_BEGIN
    REVO_REQUIRE(packed && out_scores && out_indices && out_counts, "merge: null argument");
    REVO_REQUIRE(Q >= 0 && k >= 1, "merge: bad sizes");
    REVO_REQUIRE(!unc_count || (unc_q && unc_need), "merge: the certificate needs all three of unc_count, unc_q, unc_need");
    const int64_t pb = revo_topk_packed_bytes(Q, k);
    ProfScope ps("topk_merge", (hipStream_t)stream);
    // part p: [Q][k] int64 indices, then [Q][k] fp32 scores, then [Q] fp32 certificate bounds
    revo::MergeCert mc{};
    if (unc_count) {
        REVO_HIP_CHECK(hipMemsetAsync(unc_count, 0, 4, (hipStream_t)stream));
        mc.cert = (const float*)((const char*)packed + (size_t)Q * k * 12); mc.cert_part_stride = pb / 4;
        mc.unc_count = unc_count; mc.unc_q = unc_q; mc.unc_need = unc_need;
    }
    return revo::launch_topk_merge_strided((const float*)((const char*)packed + (size_t)Q * k * 8), pb / 4,
                                           (const long long*)packed, pb / 8, parts, Q, k, has_thr, thr, out_scores,
                                           (long long*)out_indices, out_counts, (hipStream_t)stream, unc_count ? &mc : nullptr);
    API_END
}

// ------------------------------------------------------- single kernels ----
extern "C" int32_t revo_op_gemm(int32_t epi, const void* a, int64_t lda, const void* b, int64_t ldb, int32_t m,
                                int32_t n, int32_t k, void* c, int64_t ldc, const float* bias, const float* gamma,
                                void* stream) {
    API_BEGIN
    REVO_REQUIRE(epi >= 0 && epi <= 3, "op_gemm: epilogue must be 0..3");
    // the residual epilogue may cut the K range of its last, partly filled round of tiles (split-K tail): give it
    // the scratch the ViT forward would, for the duration of this call (stream-ordered allocation: nothing is kept
    // by the library between calls, nothing waits)
    constexpr long OP_WS_ELEMS = 16l << 20;
    hipStream_t st = (hipStream_t)stream;
    float* ws = nullptr;
    if (epi == revo::EPI_RESID_F32) REVO_HIP_CHECK(hipMallocAsync((void**)&ws, OP_WS_ELEMS * 4, st));
    const int rc = gemm("gemm_op", epi, (const bf16_t*)a, lda, (const bf16_t*)b, ldb, m, n, k, c, ldc, bias, gamma, st, ws,
                        ws ? OP_WS_ELEMS : 0);
    if (ws) REVO_HIP_CHECK(hipFreeAsync(ws, st));
    return rc;
    API_END
}
// The two halves of a LayerNorm folded into the GEMMs around it (kernels.h GemmArgs::lnf_* / lnc_*), one kernel each
extern "C" int32_t revo_op_gemm_resid_ln(const void* a, int64_t lda, const void* b, int64_t ldb, int32_t m, int32_t n, int32_t k,
                                         float* c, int64_t ldc, const float* bias, const float* gamma, void* xb, int64_t ldxb,
                                         void* stats, int32_t* done, void* xlo, int32_t x_in_planes, int32_t planes_out,
                                         void* stream) {
    API_BEGIN
    REVO_REQUIRE(a && b && c && xb && done, "op_gemm_resid_ln: null argument");
    REVO_REQUIRE(stats || (xlo && x_in_planes && !planes_out), "op_gemm_resid_ln: stats may be NULL only for planes in, fp32 out");
    REVO_REQUIRE(xlo || (!x_in_planes && !planes_out), "op_gemm_resid_ln: planes need the low plane");
    constexpr long OP_WS_ELEMS = 16l << 20;
    hipStream_t st = (hipStream_t)stream;
    float* ws = nullptr;
    REVO_HIP_CHECK(hipMallocAsync((void**)&ws, OP_WS_ELEMS * 4, st));
    revo::GemmArgs g{};
    g.A = (const bf16_t*)a; g.lda = lda; g.B = (const bf16_t*)b; g.ldb = ldb; g.M = m; g.N = n; g.K = k; g.C = c; g.ldc = ldc;
    g.bias = bias; g.gamma = gamma; g.ws = ws; g.ws_elems = OP_WS_ELEMS;
    int flag = 0;
    if (stats) { g.lnf_xb = (bf16_t*)xb; g.lnf_ldxb = ldxb; g.lnf_stats = (float2*)stats; g.lnf_done = &flag; }
    if (xlo) { g.xp_hi = (bf16_t*)xb; g.xp_lo = (bf16_t*)xlo; g.xp_ld = ldxb; g.xp_in = x_in_planes != 0; g.xp_out = planes_out != 0; }
    const int rc = revo::launch_gemm(revo::EPI_RESID_F32, g, st);
    REVO_HIP_CHECK(hipFreeAsync(ws, st));
    *done = flag;
    return rc;
    API_END
}
extern "C" int32_t revo_op_gemm_ln_in(int32_t epi, const void* a, int64_t lda, const void* b, int64_t ldb, int32_t m, int32_t n,
                                      int32_t k, void* c, int64_t ldc, const float* bias, const float* csum, const void* stats,
                                      int32_t parts, float eps, void* tele, void* stream) {
    API_BEGIN
    REVO_REQUIRE(epi == revo::EPI_BF16 || epi == revo::EPI_BF16_GELU, "op_gemm_ln_in: epilogue must be 0 (bf16) or 1 (bf16 + GELU)");
    REVO_REQUIRE(a && b && c && csum && stats, "op_gemm_ln_in: null argument");
    revo::GemmArgs g{};
    g.A = (const bf16_t*)a; g.lda = lda; g.B = (const bf16_t*)b; g.ldb = ldb; g.M = m; g.N = n; g.K = k; g.C = c; g.ldc = ldc;
    g.bias = bias; g.lnc_stats = (const float2*)stats; g.lnc_parts = parts; g.lnc_c = csum; g.lnc_eps = eps;
    g.lnc_tele = (unsigned long long*)tele;
    return revo::launch_gemm(epi, g, (hipStream_t)stream);
    API_END
}
extern "C" int32_t revo_op_gemm_rope(const void* a, int64_t lda, const void* b, int64_t ldb, int32_t m, int32_t n,
                                     int32_t k, void* c, int64_t ldc, const float* bias, const float* cos_sin,
                                     int32_t seq, int32_t head_dim, int32_t rope_cols, void* stream) {
    API_BEGIN
    revo::GemmArgs g{};
    g.A = (const bf16_t*)a; g.lda = lda; g.B = (const bf16_t*)b; g.ldb = ldb; g.M = m; g.N = n; g.K = k; g.C = c; g.ldc = ldc;
    g.bias = bias; g.rope_cs = (const float2*)cos_sin; g.rope_S = seq; g.rope_hd = head_dim; g.rope_cols = rope_cols;
    return revo::launch_gemm(revo::EPI_BF16_ROPE, g, (hipStream_t)stream);
    API_END
}
#ifdef REVO_EXPERIMENTS
// result-preserving kernel-variant switches (librevo_exp.so only; see revo.h)
extern "C" int32_t revo_op_set_variant(int32_t flags) {
    revo::gemm_force_gy((flags >> 4) & 15);
    revo::attention_force_nw((flags >> 8) & 15);
    revo::gemm_set_tail_split(((flags >> 12) & 1) ? 0 : 1);
    revo::gemm_set_persistent(((flags >> 16) & 1) ? 0 : 1);
    revo::gemm_set_splitk(((flags >> 17) & 1) ? 0 : 1);
    revo::gemm_set_min_tiles256(((flags >> 18) & 1) ? 0 : 100);
    revo::gemm_set_ring(((flags >> 19) & 1) ? 0 : 1, 0);
    revo::gemm_set_rows192(((flags >> 3) & 1) ? 0 : 1);
    revo::attention_set_shape16((flags >> 20) & 1);
    return 0;
}
// 0: every ln_1 / ln_2 runs as its own LayerNorm kernel (A/B timing and parity of the folded form against it); 1: default
extern "C" int32_t revo_op_set_ln_fold(int32_t on) {
    REVO_REQUIRE(on >= 0 && on <= 2, "set_ln_fold: 0 (LayerNorm kernels), 1 (folded, stream in planes: default) or 2 (folded, fp32 stream + bf16 copy)");
    revo::gemm_set_ln_fold(on);
    return 0;
}
// 0: the bf16 epilogues on gemm256p_kernel (a tile's stores drained before the next main loop: the round-5 kernels); 1: default,
// gemm256q_kernel (stores left in flight through the next tile's first K-tile).  Same results either way.
extern "C" int32_t revo_op_set_qstores(int32_t on) {
    revo::gemm_set_qstores(on);          // (2: queued, stores dropped -- timing experiment, WRONG RESULTS)
    return 0;
}
// phase groups of the persistent 256 x 256 GEMM (gemm.hip gemm256pp_kernel): 0 = the launcher's choice, 1 = off, 2..4 forced
extern "C" int32_t revo_op_set_phase_groups(int32_t groups) {
    REVO_REQUIRE(groups >= 0 && groups <= 4, "set_phase_groups: 0 (heuristic), 1 (off) or 2..4");
    revo::gemm_set_phase_groups(groups);
    return 0;
}
// diagnostic: device array [workgroups][2] of u64 the body attention kernel fills with (shader-clock ticks, 100 MHz ticks) per workgroup
extern "C" int32_t revo_debug_attention_clock(void* buf) {
    revo::attention_set_clock_buffer((unsigned long long*)buf);
    return 0;
}
// diagnostic: buf = device array [workgroups][items][4] of u64 (100 MHz stamps: main loop begin, main loop end, epilogue
// issued, rows of the piece) that the phased persistent kernel fills for its first `items` pieces; NULL = off
extern "C" int32_t revo_debug_gemm_stamps(void* buf, int32_t items) {
    revo::gemm_set_stamps((unsigned long long*)buf, buf ? items : 0);
    return 0;
}
extern "C" int32_t revo_op_set_gemm_tile(int32_t tile) {
    REVO_REQUIRE(tile == 0 || tile == 128 || tile == 256, "set_gemm_tile: 0, 128 or 256");
    revo::gemm_force_tile(tile);
    return 0;
}
// timing experiments (librevo_exp.so only): the variant bits plus the switches that skip work
extern "C" int32_t revo_op_set_gemm_debug(int32_t flags) {
    revo::gemm_set_debug(flags & 3);
    revo::gemm_set_stagger(((flags >> 28) & 15) * 200, 2 + ((flags >> 2) & 3));   // bits 28-31: stagger in 2 us steps (100 MHz clock), bits 2-3: groups - 2
    revo::topk_scan256_set_debug(((flags >> 13) & 7) | (((flags >> 20) & 255) << 3));
    return revo_op_set_variant(flags);
}
// the scan launch's phases for Q queries over `rows` scanned rows: host logic only, no device needed (CPU-tier tests)
extern "C" int64_t revo_debug_scan_plan(int32_t Q, int64_t rows, int64_t* out, int32_t cap) {
    if (Q < 1 || rows < 1 || !out || cap < 2) return -1;
    return (int64_t)revo::topk_scan256_plan_dump(Q, (long)rows, (long*)out, cap);
}
// whether the forward keeps the residual stream in planes for this many rows of this tower (reporting / tests)
extern "C" int32_t revo_debug_stream_in_planes(const revo_vit* v, int32_t batch) {
    if (!v || !v->xlo) return 0;
    const int W = v->cfg.width, Md = v->cfg.mlp_dim, rows = batch * v->S;
    return revo::gemm_ln_planes_enabled() && revo::gemm_resid_folds(rows, W, W, W, W, W) && revo::gemm_resid_folds(rows, W, Md, Md, Md, W);
}
// copy bytes [offset, offset + bytes) of the handle's search workspace to the host (debugging the scan's buffers)
extern "C" int64_t revo_debug_read_workspace(revo_gallery* g, int64_t offset, int64_t bytes, void* host_dst) {
    if (!g || !g->part) return -1;
    if (!host_dst) return (int64_t)g->part_cap;
    if (offset < 0 || bytes < 0 || (size_t)(offset + bytes) > g->part_cap) return -2;
    if (hipDeviceSynchronize() != hipSuccess) return -3;
    if (hipMemcpy(host_dst, (const char*)g->part + offset, (size_t)bytes, hipMemcpyDeviceToHost) != hipSuccess) return -3;
    return bytes;
}
extern "C" int32_t revo_debug_seed_bounds(revo_gallery* g, const uint32_t* bounds) {
    if (!g) return -1;
    g->seed_bounds = bounds;
    return 0;
}
extern "C" int32_t revo_debug_scan_stats(int64_t* out4) {
    REVO_HIP_CHECK(hipDeviceSynchronize());
    REVO_HIP_CHECK(hipMemcpy(out4, revo::topk_scan256_stats(), 64, hipMemcpyDeviceToHost));
    REVO_HIP_CHECK(hipMemset(revo::topk_scan256_stats(), 0, 64));
    return 0;
}
#endif
extern "C" int32_t revo_op_layernorm(const float* x, int64_t ldx, const float* w, const float* b, float eps,
                                     int32_t rows, int32_t width, void* out, int64_t ldo, int32_t out_is_bf16,
                                     void* stream) {
    API_BEGIN
    return revo::launch_layernorm(x, ldx, w, b, eps, rows, width, out, ldo, out_is_bf16, (hipStream_t)stream);
    API_END
}
extern "C" int32_t revo_op_layernorm_logits(const float* x, int64_t ldx, const float* w, const float* b, float eps, int32_t rows,
                                            int32_t width, float* out, int64_t ldo, const float* qk, const float* ck,
                                            int32_t heads, int32_t seq, float* logits, void* stream) {
    API_BEGIN
    REVO_REQUIRE(x && out && qk && ck && logits && rows >= 1 && seq >= 1 && rows % seq == 0, "op_layernorm_logits: bad arguments");
    const revo::LnLogits lg{qk, ck, logits, heads, seq};
    return revo::launch_layernorm(x, ldx, w, b, eps, rows, width, out, ldo, 0, (hipStream_t)stream, &lg);
    API_END
}
extern "C" int32_t revo_op_linear_f32(int32_t epi, const float* A, int64_t lda, const float* Wt, int64_t ldw, const float* bias,
                                      int32_t M, int32_t N, int32_t K, float* C, int64_t ldc, void* stream) {
    API_BEGIN
    REVO_REQUIRE(A && Wt && C && M >= 1 && N >= 1 && K >= 16 && lda >= K && ldw >= K && ldc >= N, "op_linear_f32: bad arguments");
    return revo::launch_gemm_f32_skinny(epi, A, lda, 0, 0, Wt, ldw, bias, M, N, K, C, ldc, (hipStream_t)stream);
    API_END
}
extern "C" int32_t revo_op_pool_rows(const float* x, int64_t ldx, const float* logits, int32_t batch, int32_t seq, int32_t width,
                                     int32_t heads, float* u, void* stream) {
    API_BEGIN
    REVO_REQUIRE(x && logits && u && batch >= 1 && seq >= 1 && width >= 4 && ldx >= width, "op_pool_rows: bad arguments");
    return revo::launch_pool_head_rows(x, ldx, batch, seq, width, heads, logits, u, (hipStream_t)stream);
    API_END
}
extern "C" int32_t revo_op_rope(void* qkv, int64_t ld, const float* cs, int32_t rows, int32_t seq, int32_t width,
                                int32_t heads, void* stream) {
    API_BEGIN
    return revo::launch_rope((bf16_t*)qkv, ld, (const float2*)cs, rows, seq, width, heads, (hipStream_t)stream);
    API_END
}
extern "C" int32_t revo_op_attention(const void* qkv, int64_t ld, void* out, int64_t ldo, int32_t batch, int32_t seq,
                                     int32_t heads, int32_t head_dim, void* stream) {
    API_BEGIN
    ProfScope ps("attention", (hipStream_t)stream);
    return revo::launch_attention((const bf16_t*)qkv, ld, (bf16_t*)out, ldo, batch, seq, heads, head_dim,
                                  (hipStream_t)stream);
    API_END
}
extern "C" int32_t revo_op_f32_to_bf16(const float* src, int64_t ld_src, void* dst, int64_t ld_dst, int64_t rows,
                                       int32_t cols, void* stream) {
    API_BEGIN
    return revo::launch_f32_to_bf16(src, ld_src, (bf16_t*)dst, ld_dst, rows, cols, (hipStream_t)stream);
    API_END
}
