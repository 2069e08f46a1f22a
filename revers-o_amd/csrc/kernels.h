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
    int t192_tiles, t192_tall;         // set by the launcher (192-row form): tile rows, how many of the last are 208 rows tall
    int ksplit; long c_split_stride;   // set by the launcher: K ranges per tile, fp32 elements between partial outputs
    // EPI_RESID_F32 only, optional: the LayerNorm that reads the updated residual rows next (ln_w != null).  If the
    // launcher takes the one-image form -- ring kernel on K parts + one reduce -- the reduce also normalises its rows
    // into ln_out (bf16) and *ln_fused is set to 1: the caller then skips that LayerNorm launch (a 577-row LayerNorm is
    // nothing but its launch: 5 us of a 90-us block).  Any other form leaves *ln_fused alone.
    // (ln_w == null with ln_out set: normalise only, no affine -- the towers' gains and shifts live in the weights of
    //  the GEMM that follows, api.hip "LayerNorm folded")
    const float* ln_w; const float* ln_b; float ln_eps; bf16_t* ln_out; long ln_ldo; int* ln_fused;
    // LayerNorm folded into the two GEMMs around it.  The LayerNorm's gain lives in the next GEMM's
    // weights (W' = bf16(gamma . W), bias b' = b + W beta), so what that GEMM needs from the row is bf16(x) as its A
    // operand and the row's mean and 1/std in its epilogue:   out = rstd * (bf16(x) . W'^T - mean * c) + b',  c_j = sum_k W'_jk.
    //  * producer (EPI_RESID_F32, the GEMM that writes the residual row): lnf_xb (bf16 [M][lnf_ldxb]) receives bf16(x_new),
    //    lnf_stats ([M][N / 256] (mean, M2) of each 256-column slice of the new row: one slot per column tile, fixed
    //    order, merged by the consumer).  Only launch forms that cover ALL rows with the 256 x 256 kernels' row-coalesced
    //    epilogue do this; they set *lnf_done = 1, any other form leaves it alone (the caller then runs the LayerNorm kernel).
    //  * consumer (EPI_BF16 / _GELU / _ROPE): lnc_stats ([M][lnc_parts] as written above; the normalised width is
    //    256 * lnc_parts), lnc_c ([N] column sums of the rounded weights), lnc_eps; `bias` is b'.
    bf16_t* lnf_xb; long lnf_ldxb; float2* lnf_stats; int* lnf_done;
    // Residual stream as two bf16 planes (EPI_RESID_F32, the two whole-rows persistent forms only):
    // between the folded residual GEMMs of one forward the stream is kept as hi = bf16(x), lo = bf16(x - hi) -- x to 2^-17
    // instead of 2^-24, still 256 times finer than the bf16 GEMM operands -- because hi IS the next GEMM's A operand: the
    // epilogue moves 4 + 4 bytes per element where fp32 rows plus their bf16 copy were 4 + 4 + 2.  xp_in: the old
    // values come from (xp_hi, xp_lo) instead of C; xp_out: the new values go there instead of to C (C is then stale;
    // xp_hi must be lnf_xb).  *lnf_done = 1 also says that xp_out was honoured.
    bf16_t* xp_hi; bf16_t* xp_lo; long xp_ld; int xp_in, xp_out;
    const float2* lnc_stats; int lnc_parts; const float* lnc_c; float lnc_eps;
    // Telemetry of the folded consumer (optional; device, 3 x u64): the A operand is bf16(x) UN-CENTRED, so the fold's
    // error against LayerNorm-then-round grows with |mean| / std of a row (tests/test_gpu_ln_fold.py records the curve).
    // The 256 x 256 kernels' column-tile-0 workgroups -- each row of the launch exactly once -- add: [0] rows merged,
    // [1] rows with |mean| * rstd > LNC_TELE_RATIO, [2] rows with |mean| * rstd > 4 * LNC_TELE_RATIO.
    unsigned long long* lnc_tele;
    // Set by the launcher (queued-stores kernel only): rows [M, M + qtail_rows) -- the few rows that whole rounds of 256-row
    // tiles leave over (fc1 of PE-L14 at batch 64: 64) -- are done by the same launch, one 16-column strip per workgroup
    // behind its last tile, instead of by a launch of their own (gemm.hip gemm256q_kernel)
    int qtail_rows;
#ifdef REVO_EXPERIMENTS
    int stagger_cycles, stagger_groups;   // timing experiment (persistent kernel): phase groups, see gemm256p_kernel
    unsigned long long* stamps; int stamp_items;   // diagnostic (gemm256pp_kernel): [workgroup][item][4] 100 MHz time stamps
#endif
};
constexpr float LNC_TELE_RATIO = 8.0f;
// true when launch_gemm would run the 256 x 256 kernel with the row-coalesced epilogue for these sizes
bool gemm_uses_wide_epilogue(int M, int N, long lda, long ldb, long ldc);
int launch_gemm(int epi, const GemmArgs& a, hipStream_t st);
// EPI_F32 on 128 x 64 ring tiles whatever the tile count (few rows x very many columns: the search pre-pass of few queries)
int launch_gemm_f32_ring(const GemmArgs& a, hipStream_t st);
// true when launch_gemm(EPI_RESID_F32) would take one of the two launch forms that can fold the LayerNorm behind it (and
// keep the residual stream in planes): the persistent 256-row kernel over ALL rows
bool gemm_resid_folds(int M, int N, int K, long lda, long ldb, long ldc);
// residual stream planes <-> fp32 (format changes at the ends of a forward's folded stretch; parity hooks)
int launch_planes_to_f32(const bf16_t* hi, const bf16_t* lo, long ldp, float* x, long ldx, long rows, int W, hipStream_t st);
#ifdef REVO_EXPERIMENTS
void gemm_set_debug(int d);           // timing experiments (results are wrong): librevo_exp.so only
void gemm_set_stagger(int cycles, int groups);
void gemm_set_stamps(unsigned long long* buf, int items);
void gemm_set_phase_groups(int g);    // 0 / 1 = off, 2..4 = forced (scripts/gemm_phase_ab.py)
#endif
void gemm_force_gy(int gy);
void gemm_set_tail_split(int on);
void gemm_force_tile(int t);   // 0 = heuristic, 128 or 256 = forced

// --------------------------------------------------------- elementwise -----
// out = LayerNorm(x) * w + b over rows of width W (fp32 statistics, two-pass).
// logits (optional, fp32 output): also the attention-pool logits of every normalised row (head.hip): row r = token r % S of
// image r / S; logits[(image * H + h) * S + token] = y . qk[h] + ck[h]
constexpr int LN_MAXH = 16;
struct LnLogits { const float* qk; const float* ck; float* logits; int H, S; };
int launch_layernorm(const float* x, long ldx, const float* w, const float* b, float eps, int rows, int W,
                     void* out, long ldo, int out_is_bf16, hipStream_t st, const LnLogits* logits = nullptr);
// (w / b null: normalise only.)  Weights of a linear layer with the LayerNorm in front of it folded in (elementwise.hip):
// out = bf16(gamma . W) [rows][ld], csum[j] = sum_k of the rounded row, bias_out = bias + W beta
int launch_fold_ln_linear(const float* w, const float* gamma, const float* beta, const float* bias, int rows, int cols,
                          bf16_t* out, long ld, float* csum, float* bias_out, hipStream_t st);
// images NCHW (u8 pixel values; f32: already normalised to [-1, 1]) -> im2col matrix for the split-precision patch GEMM:
// rows [B*G*G][parts * Kp] bf16, k = c*P*P + py*P + px zero padded to Kp, parts = 2 (u8: the exact integers 2v - 255,
// twice) or 3 (f32: hi | hi | lo of 255 x); the weights carry the 1/255 (elementwise.hip, head.hip split_hi_lo_hi)
int launch_patchify(const void* images, int is_u8, int B, int img, int P, bf16_t* out, long Kp, hipStream_t st);
// ---- the attention-pool head in fp32 (head.hip)
int launch_probe_qk(const float* q, const float* Wk, const float* bk, int W, int heads, float* qk, float* ck, hipStream_t st);
// u [B][H][W] = softmax-weighted sums of the fp32 ln_post rows x; the logits [B][H][S] of the probe come from the
// ln_post kernel (launch_layernorm's LnLogits)
int launch_pool_head_rows(const float* x, long ldx, int B, int S, int W, int H, const float* logits, float* u, hipStream_t st);
// C[M][N] (+)= epi(A . Wt^T + bias) in fp32, M small; epi 0 plain, 1 exact-erf GELU, 2 C += ; group_cols > 0: output
// columns [g * group_cols, ...) read A at A + g * a_group_stride
int launch_gemm_f32_skinny(int epi, const float* A, long lda, long a_group_stride, int group_cols, const float* Wt, long ldw,
                           const float* bias, int M, int N, int K, float* C, long ldc, hipStream_t st);
int launch_transpose_f32(const float* src, int rows, int cols, float* dst, hipStream_t st);
int launch_split_hi_lo_hi(const float* src, long rows, int cols, float scale, bf16_t* dst, long ld, hipStream_t st);
// x[b*S + 0][:] = cls + pos[0]
int launch_cls_rows(float* x, long ldx, const float* cls, const float* pos, int B, int S, int W, hipStream_t st);
// in-place interleaved-pair rotation of the q and k thirds of qkv [rows][3W]
int launch_rope(bf16_t* qkv, long ld, const float2* cs, int rows, int S, int W, int heads, hipStream_t st);
// dst[r][0..cols) = bf16(src[r][0..cols)), dst[r][cols..ld_dst) = 0
int launch_f32_to_bf16(const float* src, long ld_src, bf16_t* dst, long ld_dst, long rows, int cols, hipStream_t st);
// row-wise L2 normalise; writes fp32 and/or bf16 copies (either may be null)
// normalize = 0: rows copied unscaled.  row_stats [rows][2] (optional): (||bf16(y) - y||, ||bf16(y)||) per output row;
// max_stats [2] (optional): running maxima (fp32 bit patterns) of ||y|| and ||bf16(y) - y||  -- inputs of the search's
// exactness certificate
// zero_a / zero_b (optional): two word arrays the kernel also clears (a search's counters and histograms: its first kernel
// does what two memset launches did)
int launch_l2norm_rows(const float* src, long ld_src, float* dst_f32, long ld_f32, bf16_t* dst_bf16, long ld_bf16,
                       long rows, int D, hipStream_t st, int normalize = 1, float* row_stats = nullptr,
                       uint32_t* max_stats = nullptr, uint32_t* zero_a = nullptr, long zero_a_words = 0,
                       uint32_t* zero_b = nullptr, long zero_b_words = 0);

// ----------------------------------------------------------- attention -----
// qkv [B*S][ld] bf16 (q | k | v thirds, heads contiguous inside a third) -> out [B*S][ldo] bf16
int launch_attention(const bf16_t* qkv, long ld, bf16_t* out, long ldo, int B, int S, int H, int hd, hipStream_t st);
// has_cls: row 0 is the class token and may be scheduled apart from the patch rows (same result)
int launch_attention_ex(const bf16_t* qkv, long ld, bf16_t* out, long ldo, int B, int S, int H, int hd, int has_cls,
                        hipStream_t st);
void gemm_set_min_tiles256(int n);  // timing experiments only (default 100)
void gemm_set_ring(int on, int max_tiles);   // timing experiments only (default on, 256 tiles)
void gemm_set_splitk(int on);       // timing experiments only (1 = default)
void gemm_set_persistent(int on);   // timing experiments only (1 = default)
void gemm_set_qstores(int on);      // timing experiments only (1 = default): 0 = the bf16 epilogues on gemm256p_kernel (stores drained before the next main loop)
void gemm_set_rows192(int on);      // timing experiments only (1 = default)
void gemm_set_ln_fold(int on);      // timing experiments only (1 = default): 0 = LayerNorm kernels instead of the folded form, 2 = folded with an fp32 stream
bool gemm_ln_planes_enabled();
void attention_force_nw(int nw);   // timing experiments only (0 = heuristic)
void attention_set_shape16(int on); // head_dim 64: 1 = the v_mfma_f32_16x16x32_bf16 kernel (attn16_fwd_kernel), 0 = the 32x32x16 one
#ifdef REVO_EXPERIMENTS
// diagnostic: every workgroup of the body attention kernel writes (shader-clock ticks, 100 MHz ticks) of its lifetime to
// buf[workgroup][2] (device; null = off): the clock the chip holds under this kernel (MI355X_MICROARCH.md, DVFS item 6)
void attention_set_clock_buffer(unsigned long long* buf);
#endif

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
// ---- exactness certificate and its fallback (topk.hip: finish; topk256.hip: collect; topk_exact.hip: the rest) ----
// The scan selects on bf16-input scores; the finish step re-scores its ksel candidates in fp32.  A query is CERTIFIED
// when the score a row needs to enter the result exceeds the best bf16 score any un-re-scored row can have by more
// than eps, a rigorous bound of |bf16-scan score - fp32 score|.  Uncertified queries are handed
// to the collect pass (every row whose bf16 score is within eps of what is needed, re-scored exactly), and queries
// whose collect list overflows to a brute-force fp32 pass over the whole gallery.  All of it is enqueued on the
// stream with device-side counts: no host round trip.
constexpr int EXACT_COL_CAP = 2048;     // collected keys per uncertified query
constexpr int EXACT_L3_SLICES = 32;     // gallery slices of the brute-force pass (their 64-key lists reuse the query's collect buffer)
static_assert(EXACT_L3_SLICES * 64 == EXACT_COL_CAP, "the brute-force partial lists live in the collect buffer");
struct ExactWs {
    int* ctr;            // [0] uncertified queries of this search, [1] of those: collect list overflowed -> brute force,
                         // [2] queries the certificate was evaluated for, [3] rows collected (all lists), [4] mode 3: failed
                         // (counted only), [5] uncertified queries resolved from the scan's segments (numbered from the back)
    int* unc_q;          // [cap] query index of uncertified entry j (or the output row, see out_compact)
    float* unc_lb;       // [cap] bf16-score bound of entry j's collect pass
    bf16_t* qb_u;        // [cap][ldqb] its bf16 query row (compacted: the collect pass reads whole query tiles)
    long ldqb;
    int* col_cnt;        // [cap] rows appended to entry j's list (may exceed EXACT_COL_CAP: overflow)
    uint64_t* col;       // [cap][EXACT_COL_CAP] keys (bf16 score, row)
    int* over_j;         // [cap] entries that overflowed
    int* orow;           // [cap] out_compact results: the output row of entry j (its place in the caller's list)
    int cap;             // entries the arrays hold.  Entries the finish step fills from the scan's own segments (no gallery
                         // pass needed) are numbered from the BACK: cap - 1, cap - 2, ...; their count is ctr[5]
};
// The scan's segments of one launch part (queries [q0, q0 + nq) of the search): every survivor the scan appended
struct SegSrc { const uint64_t* seg; const int* cnt; int splits, q0, nq; };
struct CertArgs {
    const float* qstat;      // [Q][2] (||qb - qf||, ||qb||) from the query normalisation
    const uint32_t* gstat;   // [2] fp32 bit patterns: max ||g||, max ||gb - g|| over the gallery's rows
    int mode;                // 0 = certificate + fallback, 1 = every query takes the collect path, 2 = every query takes the
                             // brute-force path, 3 = certificate only (counted, no fallback: the pre-certificate behaviour)
    ExactWs ws;
    const bf16_t* Qb; long ldq;   // the search's bf16 query rows (source of qb_u)
    float* cert_out;         // row-sharded search: [Q] this shard's bound U + eps for the merge step's certificate (then no
                             // local decision is taken); null otherwise
    // Resolving an uncertified query WITHOUT another pass over the gallery (nsegs > 0: the scan ran with an admission
    // margin, launch_topk_scan256): the scan admitted every row scoring at least max(pre-pass bound, bound - marg[q]),
    // where `bound` is any lower bound of the query's ksel-th best scan score it held at the time -- so every row with
    // scan score >= max(base, U - marg[q]), U = the ksel-th best scan score, sits in the query's segments (or, a
    // pre-pass row, in prelist), unless a drain or a recomputed tile touched the query (dropflag).  If the bound the
    // collect pass would use reaches that level, the finish step fills the entry's list from there.
    SegSrc segs[2]; int nsegs;
    int seg_ksel;                // ksel of the scan (segments hold 2 * ksel keys)
    const uint64_t* prelist;     // [Q][ksel] the pre-pass rows' best keys
    const uint32_t* tau_base;    // [Q] the pre-pass bound (order-preserving u32) -- or, `estimated`, the estimate the scan started from
    int estimated;               // the scan's starting admission scores were estimates (a shard of a larger gallery): finish counts
                                 // them as the score an unseen row can have
    const float* marg;           // [Q] admission margins
    const int* dropflag;         // [Q] != 0: the query's segments may be incomplete or hold repeated keys
};
// marg[q] = 2 eps(q) (+ rounding slack), dropflag[q] = 0: the admission margins of a scan whose uncertified queries are
// to be resolved from its segments
int launch_cert_margin(const float* qstat, const uint32_t* gstat, int D, int Q, float* marg, int* dropflag, hipStream_t st);
// exact fp32 re-score of the ksel candidates of each query, final order (score desc, index asc),
// threshold cut, write k results.  Gf may be null: then the bf16-scan scores are returned (and nothing is certified).
// all_bounds (optional): [parts][Q][top_m] order-preserving u32 scores published by every shard of a row-sharded
// gallery; only candidates at or above their ksel-th largest (a lower bound of the global ksel-th best) are re-scored.
// cert (optional): evaluate the exactness certificate and hand uncertified queries to the fallback workspace.
int launch_topk_finish(const uint64_t* part, long part_stride, int ksel, const float* Qf, long ldqf, const float* Gf,
                       long ldgf, int D, int Q, int k, int has_thr, float thr, long idx_offset, const uint32_t* all_bounds,
                       int parts, int top_m, float* out_scores, long long* out_idx, int* out_counts, const CertArgs* cert,
                       hipStream_t st);
// rigorous bound of |bf16-scan score - fp32 re-score| for a query with (e_q, n_qb) against rows with (G, Eg); see DESIGN.md
__host__ __device__ inline float cert_eps(float e_q, float n_qb, float G, float Eg, int D) {
    const float gam1 = (float)D * 1.1920929e-7f;      // D * 2^-23: fp32 accumulation of the MFMA chain (truncation allowed)
    const float gam2 = (float)D * 5.9604645e-8f;      // D * 2^-24: the fp32 fma chain of the re-score
    const float eps = e_q * G + n_qb * Eg + gam1 * n_qb * (G + Eg) + gam2 * (n_qb + e_q) * G;
    return eps * 1.001f + 1e-30f;
}
struct Collect256Args {
    const bf16_t* Qb; long ldq;   // compacted bf16 query rows of the uncertified entries
    const bf16_t* Gb; long ldg;
    long N; int D;
    int splits, small_modes;      // set by the launcher
    const int* n_q;               // device: number of entries (query tiles past it exit)
    const float* lb;              // [entries] bf16-score bound
    int* cnt;                     // [entries] rows appended
    uint64_t* col;                // [entries][cap]
    int cap;
};
int launch_topk_collect256(const Collect256Args& a, int max_queries, hipStream_t st);
// Fallback passes over the entries ws.ctr[0] (device count; launches are sized for max_entries).
// exact_finish: fp32 re-score of every collected row, exact top-k; entries whose list overflowed go to ws.over_j.
// bruteforce: for those, the fp32 score of EVERY gallery row (same fma chain as the re-score), exact top-k.
// out_compact = 0: results go to row unc_q[j] of the outputs; 1: to row j (unc_q then only names the query row in Qf).
int launch_topk_exact_finish(const ExactWs& ws, int max_entries, const float* Qf, long ldqf, const float* Gf, long ldgf,
                             int D, int k, int has_thr, float thr, long idx_offset, int force_bruteforce, int out_compact,
                             float* out_scores, long long* out_idx, int* out_counts, hipStream_t st);
int launch_topk_exact_bruteforce(const ExactWs& ws, int max_entries, const float* Qf, long ldqf, const float* Gf, long ldgf,
                                 long N, int D, int k, int has_thr, float thr, long idx_offset, int out_compact,
                                 float* out_scores, long long* out_idx, int* out_counts, hipStream_t st);
// entries from an explicit list (the row-sharded search's second round): entry j = query q_idx[j], collect bound
// need[j] - eps(query, this gallery); sets ws.ctr[0] = n
int launch_topk_exact_prepare(const ExactWs& ws, const int* q_idx, const float* need, int n, const CertArgs& cert, int D,
                              const uint64_t* cand, long cand_stride, int ksel, hipStream_t st);
// (cand: the handle's candidate lists of the last scan, [Q][cand_stride], best first: with segments to draw from
//  (cert.nsegs > 0) an entry whose bound they cover is filled from them and numbered from the back, ExactWs::cap;
//  ws.ctr[0..7] must be zero when this runs)
// bounds[q][0..top_m) = order-preserving u32 scores of the query's best top_m candidates (0 = none)
int launch_topk_publish(const uint64_t* part, long part_stride, int Q, int top_m, uint32_t* bounds, hipStream_t st);
// 256 x 256 tile scan (topk256.hip): survivors are appended to per-(query, slice) segments of 2 * ksel keys and
// counted in per-query score histograms; launch_topk_reduce_segs then picks each query's best ksel keys
int topk_scan256_splits(int Q, long rows);
// queries the main launch of a search takes; the rest (a ragged tail of <= 128 queries) runs as a second launch in the
// small-query mode of the kernel when that is cheaper (Q = one launch)
int topk_scan256_main_queries(int Q, long rows);
long topk_scan256_plan_dump(int Q, long rows, long* out, int cap);   // the launch's phases (reporting / CPU tests)
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
                        int ksel, hipStream_t st, const float* marg = nullptr, int* dropflag = nullptr);
// marg / dropflag (optional, [Q]): admit every score >= max(pre-pass bound, bound - marg[q]) instead of >= bound, and flag
// the queries whose segments a drain or a recomputed tile touched (CertArgs: the finish step's segment collect)
// out[q][ksel] = best ksel distinct keys (sorted, best first) of prelist[q][ksel] and the query's segments;
// bounds (optional) [Q][top_m]: the order-preserving u32 scan scores of each query's best top_m candidates (what
// launch_topk_publish writes: the row-sharded search's published admission scores, without a launch of their own)
int launch_topk_reduce_segs(const uint64_t* seg, const int* seg_cnt, int splits, const uint64_t* prelist, uint64_t* out,
                            int Q, int ksel, hipStream_t st, uint32_t* bounds = nullptr, int top_m = 0);
// pre-pass of the 256 x 256 scan: KSEL best columns of every row of a [Q][n] fp32 score matrix -> part[q][slot];
// tau0[q] = the KSEL-th score; hist (optional): the kept scores are counted into the query's histogram
int launch_topk_select_rows(const float* scores, long ld, int n, int Q, uint64_t* part, long part_row_stride, int slot,
                            uint32_t* tau0, int ksel, uint32_t* hist, int hist_buckets, int hist_shift, hipStream_t st,
                            uint32_t* tau_copy = nullptr, float est_z = 0.f);
// (est_z != 0: tau0[q] = max(KSEL-th best score, mean + est_z * sigma of the row's scores): an estimated admission level)
// all-padding result for an empty gallery
int launch_topk_fill_empty(float* s, long long* i, int* c, int Q, int k, hipStream_t st);
// merge P per-shard result lists [P][Q][k] -> [Q][k]
int launch_topk_merge(const float* scores, const long long* idx, int P, int Q, int k, int has_thr, float thr,
                      float* out_scores, long long* out_idx, int* out_counts, hipStream_t st);
// the merge's share of the exactness certificate of a row-sharded search (all optional: cert == null switches it off)
struct MergeCert {
    const float* cert; long cert_part_stride;   // [P][Q] per-shard bounds U_p + eps_p written by the shards' finish steps
    int* unc_count;                             // device counter (zeroed by the caller): uncertified queries
    int* unc_q; float* unc_need;                // [Q] their indices (in no particular order) and the score a row needs to enter
};
// the same with explicit distances (in elements) between the parts of the two arrays
int launch_topk_merge_strided(const float* scores, long score_part_stride, const long long* idx, long idx_part_stride, int P,
                              int Q, int k, int has_thr, float thr, float* out_scores, long long* out_idx, int* out_counts,
                              hipStream_t st, const MergeCert* mc = nullptr);

}  // namespace revo
