// Host-side paths of the C ABI (include/revo.h) under AddressSanitizer + UBSan, with no GPU: argument validation,
// the checkpoint name / size map of revo_vit_create, error strings, and the graceful failure of everything that
// needs a device (status != 0 and a message, never a crash).  Built by `make -C revers-o_amd/csrc asan` (api.hip is
// compiled with -fsanitize=address,undefined on the host side; GPU sanitizers are not available on this pool) and
// run by tests/test_abi.py in the CPU tier.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/revo.h"

static int failures = 0;
#define EXPECT(cond)                                                                  \
    do {                                                                              \
        if (!(cond)) { std::printf("FAIL %s:%d  %s   [last error: %s]\n", __FILE__, __LINE__, #cond, revo_last_error()); ++failures; } \
    } while (0)
static bool err_has(const char* needle) { return std::strstr(revo_last_error(), needle) != nullptr; }

struct Ckpt {
    std::vector<std::string> names;
    std::vector<std::vector<float>> data;
    std::vector<revo_tensor> t;
    void add(const std::string& n, int64_t numel) { names.push_back(n); data.emplace_back((size_t)numel, 0.01f); }
    void finish() {
        t.resize(names.size());
        for (size_t i = 0; i < names.size(); ++i) { t[i].name = names[i].c_str(); t[i].data = data[i].data(); t[i].numel = (int64_t)data[i].size(); }
    }
};

static Ckpt tiny_ckpt(const revo_vit_cfg& c, int pool_mlp) {
    const int W = c.width, M = c.mlp_dim, D = c.out_dim, P = c.patch_size, G = c.image_size / P, S = G * G + (c.use_cls ? 1 : 0);
    Ckpt k;
    k.add("visual.conv1.weight", (int64_t)W * 3 * P * P);
    if (c.use_cls) k.add("visual.class_embedding", W);
    k.add("visual.positional_embedding", (int64_t)S * W);
    for (const char* n : {"visual.ln_pre.weight", "visual.ln_pre.bias", "visual.ln_post.weight", "visual.ln_post.bias"}) k.add(n, W);
    for (int i = 0; i < c.layers; ++i) {
        const std::string p = "visual.transformer.resblocks." + std::to_string(i) + ".";
        for (const char* n : {"ln_1.weight", "ln_1.bias", "ln_2.weight", "ln_2.bias", "attn.out_proj.bias", "mlp.c_proj.bias"}) k.add(p + n, W);
        k.add(p + "attn.in_proj_weight", (int64_t)3 * W * W); k.add(p + "attn.in_proj_bias", 3 * W);
        k.add(p + "attn.out_proj.weight", (int64_t)W * W);
        k.add(p + "mlp.c_fc.weight", (int64_t)M * W); k.add(p + "mlp.c_fc.bias", M); k.add(p + "mlp.c_proj.weight", (int64_t)W * M);
    }
    k.add("visual.attn_pool.probe", W);
    k.add("visual.attn_pool.attn.in_proj_weight", (int64_t)3 * W * W); k.add("visual.attn_pool.attn.in_proj_bias", 3 * W);
    k.add("visual.attn_pool.attn.out_proj.weight", (int64_t)W * W); k.add("visual.attn_pool.attn.out_proj.bias", W);
    k.add("visual.attn_pool.layernorm.weight", W); k.add("visual.attn_pool.layernorm.bias", W);
    k.add("visual.attn_pool.mlp.c_fc.weight", (int64_t)pool_mlp * W); k.add("visual.attn_pool.mlp.c_fc.bias", pool_mlp);
    k.add("visual.attn_pool.mlp.c_proj.weight", (int64_t)W * pool_mlp); k.add("visual.attn_pool.mlp.c_proj.bias", W);
    k.add("visual.proj", (int64_t)W * D);
    k.finish();
    return k;
}

int main() {
    EXPECT(revo_version() >= 100);
    EXPECT(revo_last_error() != nullptr);

    // ---- embed handle: configuration and checkpoint validation (all before the device is touched)
    revo_vit_cfg c{};
    c.image_size = 56; c.patch_size = 14; c.width = 128; c.layers = 2; c.heads = 2; c.mlp_dim = 512; c.out_dim = 64;
    c.pool_heads = 2; c.use_cls = 1; c.use_ls = 0; c.ln_eps = 1e-5f; c.rope_theta = 10000.f; c.pool_mlp_dim = 0;
    revo_vit* vit = nullptr;
    Ckpt good = tiny_ckpt(c, 4 * c.width);
    EXPECT(revo_vit_create(nullptr, good.t.data(), (int)good.t.size(), 0, 4, &vit) == -2 && err_has("null"));
    EXPECT(revo_vit_create(&c, good.t.data(), (int)good.t.size(), 0, 0, &vit) == -2 && err_has("max_batch"));
    { revo_vit_cfg b = c; b.image_size = 57; EXPECT(revo_vit_create(&b, good.t.data(), (int)good.t.size(), 0, 4, &vit) == -2 && err_has("multiple of patch_size")); }
    { revo_vit_cfg b = c; b.heads = 3; EXPECT(revo_vit_create(&b, good.t.data(), (int)good.t.size(), 0, 4, &vit) == -2); }
    { revo_vit_cfg b = c; b.heads = 4; EXPECT(revo_vit_create(&b, good.t.data(), (int)good.t.size(), 0, 4, &vit) == -2 && err_has("head_dim")); }
    { revo_vit_cfg b = c; b.pool_mlp_dim = 100; EXPECT(revo_vit_create(&b, good.t.data(), (int)good.t.size(), 0, 4, &vit) == -2 && err_has("pool_mlp_dim")); }
    {   // a tensor is missing: reported by name
        Ckpt k = tiny_ckpt(c, 4 * c.width);
        k.t.pop_back();
        EXPECT(revo_vit_create(&c, k.t.data(), (int)k.t.size(), 0, 4, &vit) == -2 && err_has("missing weight tensor: visual.proj"));
    }
    {   // the attention-pool MLP has its own width (4 * width upstream): a tower-mlp_dim-sized one is refused
        revo_vit_cfg b = c; b.mlp_dim = 384;                        // tower MLP 384, pool MLP must be 4 * 128 = 512
        Ckpt wrong = tiny_ckpt(b, 384);
        EXPECT(revo_vit_create(&b, wrong.t.data(), (int)wrong.t.size(), 0, 4, &vit) == -2 && err_has("visual.attn_pool.mlp.c_fc.weight"));
        Ckpt right = tiny_ckpt(b, 512);
        const int rc = revo_vit_create(&b, right.t.data(), (int)right.t.size(), 0, 4, &vit);
        EXPECT(rc == 0 || rc == -1);                                // passes validation; -1 = no device here
        if (rc == 0) EXPECT(revo_vit_destroy(vit) == 0);
    }
    {   // wrong element count
        Ckpt k = tiny_ckpt(c, 4 * c.width);
        k.t[0].numel -= 1;
        EXPECT(revo_vit_create(&c, k.t.data(), (int)k.t.size(), 0, 4, &vit) == -2 && err_has("expected"));
    }
    {   // null data pointer inside the table
        Ckpt k = tiny_ckpt(c, 4 * c.width);
        k.t[3].data = nullptr;
        EXPECT(revo_vit_create(&c, k.t.data(), (int)k.t.size(), 0, 4, &vit) == -2 && err_has("null name or data"));
    }
    {   // a valid checkpoint: without a GPU the create fails at the device, cleanly
        vit = nullptr;
        const int rc = revo_vit_create(&c, good.t.data(), (int)good.t.size(), 0, 4, &vit);
        EXPECT(rc == 0 || (rc == -1 && vit == nullptr && std::strlen(revo_last_error()) > 0));
        if (rc == 0) {
            EXPECT(revo_vit_seq_len(vit) == 17);
            float out[64];
            EXPECT(revo_vit_forward(vit, nullptr, 1, 1, out, 1, nullptr) == -2);
            EXPECT(revo_vit_forward(vit, out, 2, 1, out, 1, nullptr) == -2);
            EXPECT(revo_vit_forward(vit, out, 1, 5, out, 1, nullptr) == -2 && err_has("max_batch"));
            EXPECT(revo_vit_destroy(vit) == 0);
        }
    }
    EXPECT(revo_vit_destroy(nullptr) == 0);
    EXPECT(revo_vit_seq_len(nullptr) == -1);

    // ---- gallery + search
    revo_gallery* g = nullptr;
    EXPECT(revo_gallery_create(100, 10, 0, 1, &g) == -2 && err_has("multiple of 64"));
    EXPECT(revo_gallery_create(64, 0, 0, 1, &g) == -2 && err_has("capacity"));
    EXPECT(revo_gallery_create(64, (int64_t)1 << 33, 0, 1, &g) == -2);
    EXPECT(revo_gallery_create(64, 10, 0, 1, nullptr) == -2);
    {
        const int rc = revo_gallery_create(64, 10, 0, 1, &g);
        EXPECT(rc == 0 || rc == -1);
        if (rc == 0) {
            float v[64] = {1.f}, s[5]; int64_t i[5]; int32_t cnt[1];
            EXPECT(revo_gallery_append(g, v, 11, 1, 0, nullptr) == -2 && err_has("capacity"));
            EXPECT(revo_search_topk(g, v, 1, 0, 0, 0.f, 0, s, i, cnt, nullptr) == -2 && err_has("k must be"));
            EXPECT(revo_search_topk(g, v, 1, 51, 0, 0.f, 0, s, i, cnt, nullptr) == -2);
            EXPECT(revo_search_finish(g, 3, 5, 0, 0.f, 0, nullptr, 0, 0, s, i, cnt, nullptr, nullptr) == -2 && err_has("no matching"));
            EXPECT(revo_gallery_read(g, 0, 1, v, 0) == -2 && err_has("outside"));
            int32_t st4[8] = {7, 7, 7, 7, 7, 7, 7, 7};
            EXPECT(revo_search_stats(g, st4, nullptr) == 0 && st4[0] == -1 && st4[1] == 0);      // no search yet
            int32_t qi[1] = {0}; float nd[1] = {0.f};
            EXPECT(revo_search_exact(g, 2, qi, nd, 5, 0, 0.f, 0, s, i, cnt, nullptr) == -2 && err_has("more entries"));
            EXPECT(revo_search_exact(g, 1, qi, nd, 51, 0, 0.f, 0, s, i, cnt, nullptr) == -2);
            EXPECT(revo_gallery_destroy(g) == 0);
        }
    }
    EXPECT(revo_gallery_destroy(nullptr) == 0);
    EXPECT(revo_gallery_size(nullptr) == -1);
    EXPECT(revo_gallery_clear(nullptr) == -2 && err_has("null"));
    EXPECT(revo_gallery_append(nullptr, nullptr, 1, 1, 0, nullptr) == -2);
    EXPECT(revo_gallery_read(nullptr, 0, 0, nullptr, 0) == -2);
    EXPECT(revo_search_topk(nullptr, nullptr, 1, 5, 0, 0.f, 0, nullptr, nullptr, nullptr, nullptr) == -2);
    EXPECT(revo_search_candidates(nullptr, nullptr, 1, 5, 8, nullptr, nullptr) == -2);
    EXPECT(revo_search_finish(nullptr, 1, 5, 0, 0.f, 0, nullptr, 0, 0, nullptr, nullptr, nullptr, nullptr, nullptr) == -2);
    EXPECT(revo_search_plan(nullptr, 1, 5, nullptr) == -2);
    EXPECT(revo_search_ksel(1) == 32 && revo_search_ksel(16) == 32 && revo_search_ksel(17) == 64 && revo_search_ksel(50) == 64);
    EXPECT(revo_search_ksel(0) == -1 && revo_search_ksel(51) == -1);
    EXPECT(revo_topk_packed_bytes(10, 5) == 640 && revo_topk_packed_bytes(3, 1) == 48 && revo_topk_packed_bytes(-1, 5) == -1);
    EXPECT(revo_topk_merge(nullptr, nullptr, 1, 1, 1, 0, 0.f, nullptr, nullptr, nullptr, nullptr) == -2);
    EXPECT(revo_topk_merge_packed(nullptr, 1, 1, 1, 0, 0.f, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr) == -2);
    EXPECT(revo_search_exact(nullptr, 1, nullptr, nullptr, 5, 0, 0.f, 0, nullptr, nullptr, nullptr, nullptr) == -2);
    EXPECT(revo_search_stats(nullptr, nullptr, nullptr) == -2);

    // ---- single kernels and hooks: argument checks that do not need a device
    EXPECT(revo_op_gemm(9, nullptr, 0, nullptr, 0, 1, 1, 64, nullptr, 0, nullptr, nullptr, nullptr) == -2);
    {
        char buf[4];
        EXPECT(revo_prof_reset() == 0 && revo_prof_enable(0) == 0);
        EXPECT(revo_prof_report(buf, 1) == -2 && err_has("too small"));
        char big[64];
        EXPECT(revo_prof_report(big, 64) == 0 && std::strcmp(big, "{}") == 0);
    }
    EXPECT(revo_preprocess_crop_resize(nullptr, 1, 56, nullptr, nullptr) != 0);
    // round 6's entries: argument validation before anything touches a device
    EXPECT(revo_vit_stats(nullptr, nullptr, 0, nullptr) == -2);
    EXPECT(revo_probe_mfma(nullptr, 0, nullptr, 1, 1, nullptr) == -2 && err_has("probe_mfma"));
    EXPECT(revo_probe_copy(nullptr, nullptr, 16, nullptr) == -2 && err_has("probe_copy"));
    EXPECT(revo_probe_mfma_flops(1024, 10) == (int64_t)1024 * 4 * 10 * 32 * (2ll * 16 * 16 * 32));
    EXPECT(revo_op_pool_rows(nullptr, 0, nullptr, 1, 1, 4, 1, nullptr, nullptr) == -2 && err_has("op_pool_rows"));
    EXPECT(revo_op_layernorm_logits(nullptr, 0, nullptr, nullptr, 1e-5f, 1, 4, nullptr, 0, nullptr, nullptr, 1, 1, nullptr, nullptr) == -2 &&
           err_has("op_layernorm_logits"));
    EXPECT(revo_op_linear_f32(0, nullptr, 0, nullptr, 0, nullptr, 1, 1, 16, nullptr, 0, nullptr) == -2 && err_has("op_linear_f32"));

    if (failures) { std::printf("%d check(s) failed\n", failures); return 1; }
    std::printf("ALL OK\n");
    return 0;
}
