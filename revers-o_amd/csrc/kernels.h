// Internal launcher interface of librevo (not part of the C ABI; see include/revo.h).
#pragma once
#include "common.h"

namespace revo {

// ---------------------------------------------------------------- GEMM -----
enum GemmEpi {
    EPI_BF16 = 0,       // C bf16 = acc + bias
    EPI_BF16_GELU = 1,  // C bf16 = gelu_erf(acc + bias)
    EPI_RESID_F32 = 2,  // C f32 += gamma * (acc + bias)        (residual stream, in place)
    EPI_F32 = 3,        // C f32 = acc + bias
    EPI_PATCH = 4,      // C f32[(m/G2)*S + cls + m%G2] = acc + pos[cls + m%G2]
    EPI_BF16_ROPE = 5,  // EPI_BF16 + axial 2-D RoPE on columns [0, rope_cols) (q and k thirds of qkv); 256 kernel only
};

struct GemmArgs {
    const bf16_t* A; long lda;   // [M][lda]  activations, K-contiguous
    const bf16_t* B; long ldb;   // [N][ldb]  weights (nn.Linear layout), K-contiguous
    int M, N, K;
    void* C; long ldc;
    const float* bias;           // [N] or null
    const float* gamma;          // [N] LayerScale or null   (EPI_RESID_F32)
    const float* pos;            // [S][N]                    (EPI_PATCH)
    int S, G2, cls;              //                           (EPI_PATCH)
    int gy;                      // XCD arrangement of the 256 x 256 kernel (set by the launcher)
    const float2* rope_cs;       // [rope_S][rope_hd/2] (cos, sin)      (EPI_BF16_ROPE)
    int rope_S, rope_hd, rope_cols;
    int prefer256;               // caller's hint: few rows but very many columns (search pre-pass) -> 256 x 256 tiles
    float* ws; long ws_elems;    // optional scratch for the split-K tail of EPI_RESID_F32 (null = never split K)
    int ksplit; long c_split_stride;   // set by the launcher: K ranges per tile, fp32 elements between partial outputs
#ifdef REVO_EXPERIMENTS
    int stagger_cycles, stagger_groups;   // timing experiment (persistent kernel): phase groups, see gemm256p_kernel
#endif
};
// true when launch_gemm would run the 256 x 256 kernel with the row-coalesced epilogue for these sizes
bool gemm_uses_wide_epilogue(int M, int N, long lda, long ldb, long ldc);
int launch_gemm(int epi, const GemmArgs& a, hipStream_t st);
#ifdef REVO_EXPERIMENTS
void gemm_set_debug(int d);           // timing experiments (results are wrong): librevo_exp.so only
void gemm_set_stagger(int cycles, int groups);
#endif
void gemm_force_gy(int gy);
void gemm_set_tail_split(int on);
void gemm_force_tile(int t);   // 0 = heuristic, 128 or 256 = forced

// --------------------------------------------------------- elementwise -----
// out = LayerNorm(x) * w + b over rows of width W (fp32 statistics, two-pass).
int launch_layernorm(const float* x, long ldx, const float* w, const float* b, float eps, int rows, int W,
                     void* out, long ldo, int out_is_bf16, hipStream_t st);
// images NCHW (u8: normalised (v/255-0.5)/0.5 on the fly; f32: already normalised)
// -> im2col matrix [B*G*G][ld] bf16, k = c*P*P + py*P + px, zero padded to ld.
int launch_patchify(const void* images, int is_u8, int B, int img, int P, bf16_t* out, long ld, hipStream_t st);
// x[b*S + 0][:] = cls + pos[0]
int launch_cls_rows(float* x, long ldx, const float* cls, const float* pos, int B, int S, int W, hipStream_t st);
// in-place interleaved-pair rotation of the q and k thirds of qkv [rows][3W]
int launch_rope(bf16_t* qkv, long ld, const float2* cs, int rows, int S, int W, int heads, hipStream_t st);
// dst[r][0..cols) = bf16(src[r][0..cols)), dst[r][cols..ld_dst) = 0
int launch_f32_to_bf16(const float* src, long ld_src, bf16_t* dst, long ld_dst, long rows, int cols, hipStream_t st);
// row-wise L2 normalise; writes fp32 and/or bf16 copies (either may be null)
int launch_l2norm_rows(const float* src, long ld_src, float* dst_f32, long ld_f32, bf16_t* dst_bf16, long ld_bf16,
                       long rows, int D, hipStream_t st);
// dst[n][k] = src[k][n]   (fp32 -> bf16 transposed copy; visual.proj is stored [W][D])
int launch_transpose_f32_to_bf16(const float* src, int rows, int cols, bf16_t* dst, long ld_dst, hipStream_t st);

// ----------------------------------------------------------- attention -----
// qkv [B*S][ld] bf16 (q | k | v thirds, heads contiguous inside a third) -> out [B*S][ldo] bf16
int launch_attention(const bf16_t* qkv, long ld, bf16_t* out, long ldo, int B, int S, int H, int hd, hipStream_t st);
// has_cls: row 0 is the class token and may be scheduled apart from the patch rows (same result)
int launch_attention_ex(const bf16_t* qkv, long ld, bf16_t* out, long ldo, int B, int S, int H, int hd, int has_cls,
                        hipStream_t st);
void gemm_set_min_tiles256(int n);  // timing experiments only (default 100)
void gemm_set_splitk(int on);       // timing experiments only (1 = default)
void gemm_set_persistent(int on);   // timing experiments only (1 = default)
void attention_force_nw(int nw);   // timing experiments only (0 = heuristic)
// single-probe attention pool: q [W] fp32 (already projected and scaled), kv [B*S][ld] bf16 (k | v halves)
int launch_pool_attention(const float* q, const bf16_t* kv, long ld, bf16_t* out, long ldo, int B, int S, int H,
                          int hd, hipStream_t st);

// ---------------------------------------------------------------- top-k ----
struct ScanArgs {
    const bf16_t* Qb; long ldq;   // [Q][D] bf16 normalised queries
    const bf16_t* Gb; long ldg;   // [N][D] bf16 normalised gallery rows
    int Q; long N; int D;
    int ksel;                     // candidates kept per query (32 or 64)
    int splits;                   // gallery splits (grid.y); part is [Q][splits][ksel]
    uint64_t* part;               // out: candidate keys, sorted best-first, 0 = empty
};
int topk_scan_workspace_splits(int Q, long N);
int launch_topk_scan(const ScanArgs& a, hipStream_t st);
// merge `splits` sorted candidate lists per query into one list of ksel keys (in place into part[q][0])
int launch_topk_reduce(uint64_t* part, int Q, int splits, int ksel, hipStream_t st);
// exact fp32 re-score of the ksel candidates of each query, final order (score desc, index asc),
// threshold cut, write k results.  Gf may be null: then the bf16-scan scores are returned.
// all_bounds (optional): [parts][Q][top_m] order-preserving u32 scores published by every shard of a row-sharded
// gallery; only candidates at or above their ksel-th largest (a lower bound of the global ksel-th best) are re-scored.
int launch_topk_finish(const uint64_t* part, long part_stride, int ksel, const float* Qf, long ldqf, const float* Gf,
                       long ldgf, int D, int Q, int k, int has_thr, float thr, long idx_offset, const uint32_t* all_bounds,
                       int parts, int top_m, float* out_scores, long long* out_idx, int* out_counts, hipStream_t st);
// bounds[q][0..top_m) = order-preserving u32 scores of the query's best top_m candidates (0 = none)
int launch_topk_publish(const uint64_t* part, long part_stride, int Q, int top_m, uint32_t* bounds, hipStream_t st);
// 256 x 256 tile scan (topk256.hip): survivors are appended to per-(query, slice) segments of 2 * ksel keys and
// counted in per-query score histograms; launch_topk_reduce_segs then picks each query's best ksel keys
int topk_scan256_splits(int Q, long rows);
// queries the main launch of a search takes; the rest (a ragged tail of <= 128 queries) runs as a second launch in the
// small-query mode of the kernel when that is cheaper (Q = one launch)
int topk_scan256_main_queries(int Q, long rows);
int topk_scan256_hist_buckets();      // u32 counters per query in `hist`
int topk_scan256_hist_shift();        // bucket width = 2^shift ulps of the fp32 score above the pre-pass bound
#ifdef REVO_EXPERIMENTS
void topk_scan256_set_debug(int d);   // timing experiments (bit 0 / 2: results are wrong): librevo_exp.so only
unsigned long long* topk_scan256_stats();
#endif
// seg [Q][splits][2 ksel] keys, seg_cnt [Q][splits]; tau_g [Q] live admission bounds (in: = tau_base), tau_base [Q]
// the pre-pass bound (constant), hist [Q][buckets] zeroed and then seeded by launch_topk_select_rows
int launch_topk_scan256(const bf16_t* Qb, long ldq, const bf16_t* Gb, long ldg, int Q, long N, int D, long n_begin,
                        int splits, uint64_t* seg, int* seg_cnt, uint32_t* tau_g, const uint32_t* tau_base, uint32_t* hist,
                        int ksel, hipStream_t st);
// out[q][ksel] = best ksel distinct keys (sorted, best first) of prelist[q][ksel] and the query's segments
int launch_topk_reduce_segs(const uint64_t* seg, const int* seg_cnt, int splits, const uint64_t* prelist, uint64_t* out,
                            int Q, int ksel, hipStream_t st);
// pre-pass of the 256 x 256 scan: KSEL best columns of every row of a [Q][n] fp32 score matrix -> part[q][slot];
// tau0[q] = the KSEL-th score; hist (optional): the kept scores are counted into the query's histogram
int launch_topk_select_rows(const float* scores, long ld, int n, int Q, uint64_t* part, long part_row_stride, int slot,
                            uint32_t* tau0, int ksel, uint32_t* hist, int hist_buckets, int hist_shift, hipStream_t st);
// all-padding result for an empty gallery
int launch_topk_fill_empty(float* s, long long* i, int* c, int Q, int k, hipStream_t st);
// merge P per-shard result lists [P][Q][k] -> [Q][k]
int launch_topk_merge(const float* scores, const long long* idx, int P, int Q, int k, int has_thr, float thr,
                      float* out_scores, long long* out_idx, int* out_counts, hipStream_t st);
// the same with explicit distances (in elements) between the parts of the two arrays
int launch_topk_merge_strided(const float* scores, long score_part_stride, const long long* idx, long idx_part_stride, int P,
                              int Q, int k, int has_thr, float thr, float* out_scores, long long* out_idx, int* out_counts,
                              hipStream_t st);

}  // namespace revo
