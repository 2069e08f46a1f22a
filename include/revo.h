/* librevo -- C ABI of the MI355X-native embed + search hot path of revers-o.
 *
 * This is the drop-in boundary: plain pointers and sizes, no
 * torch types.  Each entry point names the reference interface it replaces
 * (file:line relative to the reference tree).  The reference itself has no FFI:
 * its hot path is two Python calls into third-party packages, so the entry points
 * below are what a ctypes binding in core_system.py would call instead of
 *   self.pe_model.encode_image(...)          core_system.py:341, :442
 *   self.vector_db.search(...)               core_system.py:659-664
 *   client.recreate_collection / upsert      core_system.py:600-603, :621
 * (binding stub: INTEGRATION.md).
 *
 * Conventions: every function returns 0 on success and a negative status on
 * failure; the message is available from revo_last_error() (thread-local).
 * Nothing throws across the boundary.  `stream` is a hipStream_t (NULL = the
 * default stream); work is enqueued asynchronously on it and the caller owns
 * ordering (revo_sync).  A handle is bound to the device it was created on: every
 * entry point that takes a handle makes that device current for the call and
 * restores the caller's device on return, so handles on different GPUs can be
 * used from one process; a handle must not be used from two threads at once;
 * distinct handles are independent.  All device pointers (and the stream) passed
 * with a handle must belong to the handle's device; functions without a handle
 * (revo_topk_merge, revo_op_*, revo_preprocess_*) run on the current device.
 */
#ifndef REVO_H
#define REVO_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct revo_vit revo_vit;
typedef struct revo_gallery revo_gallery;

/* Dimensions of a Perception-Encoder vision tower (the values the reference gets
 * implicitly from pe.CLIP.from_config("PE-Core-L14-336"), core_system.py:177-181). */
typedef struct revo_vit_cfg {
    int32_t image_size, patch_size, width, layers, heads, mlp_dim, out_dim, pool_heads;
    int32_t use_cls;    /* class token + (0,0) rope position */
    int32_t use_ls;     /* LayerScale tensors (ls_1.gamma / ls_2.gamma) present */
    float ln_eps, rope_theta;
    int32_t pool_mlp_dim; /* hidden width of the attention-pool head's MLP (upstream: 4 * width, whatever mlp_dim is); 0 = 4 * width */
} revo_vit_cfg;

/* One checkpoint tensor: upstream name ("visual.conv1.weight", ...), fp32 values
 * in host or device memory. */
typedef struct revo_tensor {
    const char* name;
    const float* data;
    int64_t numel;
} revo_tensor;

const char* revo_last_error(void);
int32_t revo_version(void);
int32_t revo_sync(void* stream);

/* ---- embed: replaces load_pe_model + encode_image (core_system.py:169-203, :341, :442) */
int32_t revo_vit_create(const revo_vit_cfg* cfg, const revo_tensor* weights, int32_t n_weights, int32_t device,
                        int32_t max_batch, revo_vit** out);
int32_t revo_vit_destroy(revo_vit* vit);
/* images: NCHW at the model resolution on the device; image_dtype 0 = fp32 already
 * normalised to [-1,1] (what self.preprocess yields, core_system.py:439), 1 = uint8
 * (normalised (v/255-0.5)/0.5 inside the patchify kernel).  out: [B, out_dim] fp32 on
 * the device; normalize != 0 applies embedding / embedding.norm() (core_system.py:447). */
int32_t revo_vit_forward(revo_vit* vit, const void* images, int32_t image_dtype, int32_t batch, float* out,
                         int32_t normalize, void* stream);
int32_t revo_vit_seq_len(const revo_vit* vit);
/* Run-time telemetry of the LayerNorm folded into the GEMMs around it (batch-sized forwards: ln_1 / ln_2 are not kernels;
 * qkv / fc1 read A = bf16(x) UN-CENTRED and apply rstd * (acc - mean * csum) + b' in their epilogues).  The form is the same
 * function as LayerNorm-then-GEMM, but its rounding error grows with |mean| / std of a row (bf16(x) spends its 8 bits on
 * the common offset): measured rms error against the fp64 LayerNorm + linear, relative to LayerNorm kernel -> bf16 -> GEMM:
 * 1.00 x at offset 0, 1.5 x at 2 sigma, 4.9 x at 8.5 sigma, 18.7 x at 33 sigma -- about 0.57 x the offset in sigmas
 * (tests/test_gpu_ln_fold.py::test_fold_error_against_the_row_offset records the curve and holds it to a bound).  The
 * reference normalises in the activations' precision before the matmul (oracle/pe_vit.py:146-155 <- core_system.py:341), so on
 * a trained checkpoint this is the number to look at first if embeddings drift: out4 = { rows the consuming GEMMs merged
 * statistics for since the last reset (each row of each folded LayerNorm of each forward, counted once; rows that a
 * separate leftover-row kernel takes -- none at the headline shapes -- are not sampled), of those: rows with |mean| * rstd > out4[3], rows with |mean| * rstd > 4 *
 * out4[3], the ratio that counts as large (8) }.  Synchronises `stream`; reset != 0 clears the counters. */
int32_t revo_vit_stats(revo_vit* vit, double* out4, int32_t reset, void* stream);

/* ---- gallery: replaces recreate_collection(size=D, COSINE) + upsert (core_system.py:600-622) */
int32_t revo_gallery_create(int32_t dim, int64_t capacity, int32_t device, int32_t keep_f32, revo_gallery** out);
int32_t revo_gallery_destroy(revo_gallery* g);
/* vecs: [n, dim] fp32, host (src_on_device = 0) or device memory.  normalize != 0
 * L2-normalises each row at insert, as qdrant does for Distance.COSINE. */
int32_t revo_gallery_append(revo_gallery* g, const float* vecs, int64_t n, int32_t normalize, int32_t src_on_device,
                            void* stream);
int64_t revo_gallery_size(const revo_gallery* g);
int32_t revo_gallery_clear(revo_gallery* g);
/* copy rows [start, start+n) of the fp32 master copy to dst (host or device); needs keep_f32 */
int32_t revo_gallery_read(revo_gallery* g, int64_t start, int64_t n, float* dst, int32_t dst_on_device);

/* ---- search: replaces vector_db.search(query_vector, limit, score_threshold) (core_system.py:659-664)
 * queries: [n_queries, dim] fp32 on the device (normalised internally, cosine semantics).
 * Results, best first under (score desc, index asc): scores [n_queries, k] fp32, indices
 * [n_queries, k] int64 (row index + index_offset), counts [n_queries] int32; entries past
 * counts[q] are -inf / -1.  has_threshold != 0 keeps only score >= threshold.
 *
 * EXACTNESS.  The reference's search is an exhaustive one (qdrant local mode: scores = G @ q over every row, then the
 * ranking).  This search returns exactly what an exhaustive scoring of the gallery's fp32 rows would -- the same rows
 * in the same order, every score an fp32 dot product in one fixed summation order -- although its scan selects on bf16
 * scores: each query carries a certificate (the k-th re-scored score must exceed the best bf16 score of any row that
 * was not re-scored by more than a rigorous bound of |bf16 score - fp32 score|, computed from the rounding norms of
 * the query and of the gallery's rows), and a query that fails it is re-done by a collecting pass over the gallery
 * (every row within that bound of what is needed, re-scored in fp32) and, if that list overflows, by a brute-force
 * fp32 pass.  Searches with k > 25 (64 candidates leave the certificate too little room on ordinary data) scan with an
 * admission margin of twice that bound: what an uncertified query needs is then among the rows its scan kept, and it is
 * re-done exactly WITHOUT another pass over the gallery.  All of it is enqueued on `stream`; revo_search_stats reports how often it happened.  (A gallery created
 * with keep_f32 = 0 has no fp32 rows: it returns the scan's own scores and certifies nothing.)
 * What "exactly" is relative to: the ranking is the exhaustive one UNDER THIS LIBRARY'S fp32 SUMMATION ORDER (one fused
 * multiply-add chain over the D products, k = 0 .. D - 1, of the fp32 query and the fp32 gallery row, both normalised in
 * fp32).  The reference accumulates the same products in float64 and rounds once (numpy G @ q on Python floats): two rows
 * whose true scores differ by less than the fp32 chain's rounding error (about D * 2^-24 * |score| <= 3e-7 at D = 1024) can
 * come out in either order there and here.  tests/test_gpu_search.py therefore compares with the float64 oracle up to
 * swaps of ADJACENT results inside that band (`near_tie`) -- a tolerance of the checker's arithmetic, not of this
 * search -- and asserts that the k-th place of no query of the headline 1 M x 1024 gallery falls inside it; exact
 * duplicates (scores equal to the bit) are always returned index-ascending. */
int32_t revo_search_topk(revo_gallery* g, const float* queries, int32_t n_queries, int32_t k, int32_t has_threshold,
                         float threshold, int64_t index_offset, float* scores, int64_t* indices, int32_t* counts,
                         void* stream);
/* ---- the same search in two phases, for a gallery that is row-sharded over several GPUs / ranks (one shard per
 * handle).  The reference has a single process and a single collection (core_system.py:659-664); this is the
 * scale-out of that call.  Per rank:
 *   1. revo_search_candidates: scan the shard (same kernels as revo_search_topk); the query's candidates stay in the
 *      handle, and bounds [n_queries, top_m] receives the scan scores of the best top_m of them as order-preserving
 *      uint32 (0 = none).  top_m <= revo_search_ksel(k); parts * top_m >= min(64, 2 * revo_search_ksel(k)) makes the bound below tight.
 *   2. all-gather `bounds` over the ranks -> all_bounds [parts, n_queries, top_m]   (RCCL; 4 * top_m bytes per query and rank)
 *   3. revo_search_finish: only candidates that can still be among the best j = min(64, 2 * revo_search_ksel(k)) of the
 *      WHOLE gallery (scan score at or above the j-th largest published score) are re-scored in fp32; results as
 *      revo_search_topk.  (Twice the unsharded search's candidates: with that margin the exactness certificate of the
 *      merge all but never fails on ordinary data, which saves the protocol's second round.)
 *      all_bounds may be NULL (re-score every candidate).  Same queries, k and stream as step 1.
 *   4. all-gather the per-rank results and revo_topk_merge_packed them (with revo_search_finish's cert block and the
 *      merge's unc_* outputs: the cross-shard exactness certificate; step 5 re-does what it cannot certify).
 *      revo_topk_merge on plain [parts, n_queries, k] arrays is valid only for shards that do NOT estimate (no
 *      revo_search_set_total_rows, or revo_search_estimates(k) == 0) and then certifies nothing across shards.
 * With steps 4 and 5 the merged result equals the unsharded revo_search_topk of the concatenated gallery.
 * The ordinary case is ONE exchange per search: shards that know the total row count (revo_search_set_total_rows) and
 * search for k <= 25 skip steps 2-3's exchange (revo_search_estimates(k) == 1: all_bounds = NULL on every rank). */
int32_t revo_search_ksel(int32_t k);     /* candidates the scan keeps per query for a top-k search (32 or 64) */
/* Tell a shard's handle how many rows the WHOLE row-sharded gallery has (0 = forget).  revo_search_candidates then starts
 * its scan from an estimate of the score a candidate of the whole gallery has to reach (extrapolated from the shard's own
 * first rows) instead of what the shard's own rows would admit: fewer survivors per tile, a faster scan.  The estimate is
 * not trusted: step 4's certificate counts it as the score an unseen row may have, and step 5 re-does what it cannot
 * certify -- results are the exhaustive search's whatever the estimate was, PROVIDED the certificate is used: once a total
 * is set, revo_search_finish refuses cert = NULL after an estimating scan (status -2), and revo_topk_merge_packed with its
 * unc_* outputs plus revo_search_exact are mandatory parts of the search.  revo_search_topk ignores the setting. */
int32_t revo_search_set_total_rows(revo_gallery* g, int64_t total_rows);
/* 1 if a two-phase search for the best k on shards that know the total row count scans against that estimate (then every
 * shard's list is already cut at the whole gallery's level and step 2's exchange may be skipped: pass all_bounds = NULL to
 * revo_search_finish on EVERY rank), 0 if not (k > 25: those scans run with the admission margin of revo_search_topk instead) */
int32_t revo_search_estimates(int32_t k);
/* how a search of n_queries against the gallery's current rows would run (reporting only): out4 = { 1 if the 256 x 256
 * scan takes it (0: the small-gallery scan), rows covered by the pre-pass GEMM, gallery slices per query tile, ksel } */
int32_t revo_search_plan(const revo_gallery* g, int32_t n_queries, int32_t k, int64_t* out4);
int32_t revo_search_candidates(revo_gallery* g, const float* queries, int32_t n_queries, int32_t k, int32_t top_m,
                               uint32_t* bounds, void* stream);
int32_t revo_search_finish(revo_gallery* g, int32_t n_queries, int32_t k, int32_t has_threshold, float threshold,
                           int64_t index_offset, const uint32_t* all_bounds, int32_t parts, int32_t top_m, float* scores,
                           int64_t* indices, int32_t* counts, float* cert, void* stream);
/* cert ([n_queries] fp32; optional only while the shard does not estimate, see revo_search_set_total_rows): this shard's share of the exactness certificate -- the best fp32 score any of its
 * rows that was NOT re-scored can have (scan score of the best such row + the error bound; -inf if every row was
 * re-scored).  All-gathered with the results (it is the third block of the packed layout below) and checked by
 * revo_topk_merge_packed against the merged k-th score.
 *   5. revo_search_exact: second round for the queries that check failed for (none on ordinary data): entry j is query
 *      q_idx[j] of the last revo_search_candidates call, need[j] the fp32 score a row must reach to change its merged
 *      result (both device arrays, identical on every rank: the merge step's unc_q / unc_need, sorted by query).  The
 *      shard collects every row whose scan score is within the error bound of need[j], re-scores it in fp32 and returns
 *      its exact local top-k in row j of scores / indices / counts ([n, k], [n, k], [n]); all-gathered and merged like
 *      step 4, these replace the queries' first-round results. */
int32_t revo_search_exact(revo_gallery* g, int32_t n, const int32_t* q_idx, const float* need, int32_t k,
                          int32_t has_threshold, float threshold, int64_t index_offset, float* scores, int64_t* indices,
                          int32_t* counts, void* stream);
/* counters of the handle's last search, read after `stream` has drained (host array of 8): out8 = { queries the
 * certificate failed for (-1: the gallery has no fp32 rows), of those: brute-forced, queries the certificate was
 * evaluated for, rows the exact passes re-scored, of the failed queries: resolved from what the scan had kept (no second
 * pass over the gallery: searches with k > 25 scan with an admission margin for that), 0, 0, 0 } */
int32_t revo_search_stats(revo_gallery* g, int32_t* out8, void* stream);
/* merge `parts` result sets laid out [parts, n_queries, k] (the all-gathered per-shard
 * results of a row-sharded gallery) into one [n_queries, k] set, same ordering rule. */
int32_t revo_topk_merge(const float* scores, const int64_t* indices, int32_t parts, int32_t n_queries, int32_t k,
                        int32_t has_threshold, float threshold, float* out_scores, int64_t* out_indices,
                        int32_t* out_counts, void* stream);

/* One result set packed for a single all-gather: [n_queries, k] int64 indices, [n_queries, k] fp32 scores, then
 * [n_queries] fp32 certificate bounds (revo_search_finish's cert), padded to revo_topk_packed_bytes(n_queries, k) bytes;
 * `packed` holds `parts` such blocks back to back.  unc_count / unc_q / unc_need (optional, device: [1], [n_queries],
 * [n_queries]): the merge checks every query's certificate over all shards and lists the queries that fail it (in no
 * particular order) with the score a row needs to enter their result -- the input of revo_search_exact. */
int64_t revo_topk_packed_bytes(int32_t n_queries, int32_t k);
int32_t revo_topk_merge_packed(const void* packed, int32_t parts, int32_t n_queries, int32_t k, int32_t has_threshold,
                               float threshold, float* out_scores, int64_t* out_indices, int32_t* out_counts,
                               int32_t* unc_count, int32_t* unc_q, float* unc_need, void* stream);

/* ---- single kernels, exposed for parity tests and micro-benchmarks (device pointers) */
int32_t revo_op_gemm(int32_t epilogue, const void* a_bf16, int64_t lda, const void* b_bf16, int64_t ldb, int32_t m,
                     int32_t n, int32_t k, void* c, int64_t ldc, const float* bias, const float* gamma, void* stream);
/* C bf16 = rope(A . B^T + bias): the QKV projection with the axial rotary embedding of the first
 * rope_cols columns fused into the epilogue (cos_sin: [seq][head_dim/2] (cos, sin) pairs; row r is token r % seq) */
int32_t revo_op_gemm_rope(const void* a_bf16, int64_t lda, const void* b_bf16, int64_t ldb, int32_t m, int32_t n, int32_t k,
                          void* c_bf16, int64_t ldc, const float* bias, const float* cos_sin, int32_t seq,
                          int32_t head_dim, int32_t rope_cols, void* stream);
/* The two halves of a LayerNorm folded into the GEMMs around it (how the forward runs ln_1 / ln_2 of batch-sized calls; the
 * gain and shift live in the consuming GEMM's weights: W' = bf16(gamma . W), bias' = bias + W beta, csum_j = sum_k W'_jk):
 *  - revo_op_gemm_resid_ln: the residual GEMM  C f32 [m][n] += gamma * (A . B^T + bias)  which, where its launch form covers all
 *    rows with the persistent 256 x 256 kernel (then *done = 1; else only C is written and *done = 0), also writes
 *    xb bf16 [m][ldxb] = bf16(new C) and stats [m][n / 256] pairs of float (mean, sum of squared deviations) of every
 *    256-column slice of the new row;
 *  - revo_op_gemm_ln_in: C bf16 = epilogue(rstd * (A . B^T - mean * csum) + bias), epilogue 0 = plain, 1 = exact-erf GELU, the
 *    row's mean and rstd = 1 / sqrt(var + eps) merged from `parts` slices of `stats` as written above. */
/*    With xlo (bf16 [m][ldxb]) the residual stream itself may be in two bf16 planes (xb_bf16, xlo) = (bf16(x), bf16(x - bf16(x))),
 *    as the forward keeps it between folded GEMMs: x_in_planes != 0 takes the old values from there instead of c, planes_out
 *    != 0 writes the new ones there instead of to c (needs a folding form, else status -2); stats may be NULL for planes in,
 *    fp32 out (the last residual GEMM of a forward). */
int32_t revo_op_gemm_resid_ln(const void* a_bf16, int64_t lda, const void* b_bf16, int64_t ldb, int32_t m, int32_t n, int32_t k,
                              float* c, int64_t ldc, const float* bias, const float* gamma, void* xb_bf16, int64_t ldxb,
                              void* stats, int32_t* done, void* xlo_bf16, int32_t x_in_planes, int32_t planes_out, void* stream);
int32_t revo_op_gemm_ln_in(int32_t epilogue, const void* a_bf16, int64_t lda, const void* b_bf16, int64_t ldb, int32_t m, int32_t n,
                           int32_t k, void* c_bf16, int64_t ldc, const float* bias, const float* csum, const void* stats,
                           int32_t parts, float eps, void* tele, void* stream);
/*    tele (optional; device, 3 x uint64, zeroed by the caller): the counters behind revo_vit_stats -- rows merged, rows with
 *    |mean| * rstd > 8, rows with |mean| * rstd > 32. */
/* ---- calibration probes (bench.py's `calibration` object; not on the reference's path: nothing there to cite).  Two fixed
 * kernels that say how fast the BOX is, so that bench lines taken on different MI355X devices can be compared:
 *  - revo_probe_mfma: `blocks` workgroups of four waves, each wave `iters` trips of 32 register-resident
 *    v_mfma_f32_16x16x32_bf16 on operands read once from src (bf16, n_elems of them, random data); sink receives one
 *    float per thread (blocks * 256).  revo_probe_mfma_flops(blocks, iters) = the launch's FLOPs.
 *  - revo_probe_copy: dst[0, bytes) = src[0, bytes), 16 bytes per lane (the HBM copy rate: 2 * bytes of traffic). */
int32_t revo_probe_mfma(const void* src_bf16, int64_t n_elems, float* sink, int32_t blocks, int32_t iters, void* stream);
int64_t revo_probe_mfma_flops(int32_t blocks, int32_t iters);
int32_t revo_probe_copy(void* dst, const void* src, int64_t bytes, void* stream);
#ifdef REVO_EXPERIMENTS
/* Kernel-variant selection and timing experiments: compiled only into librevo_exp.so (`make exp`: the same sources
 * with -DREVO_EXPERIMENTS; used by scripts/ and by the forced-tile runs of tests/test_gpu_kernels.py), never into
 * librevo.so, whose kernels are chosen by its size heuristics alone.  Process-global, not thread-safe.
 * Every variant computes the same result (split-K changes the fp32 summation order only). */
/* Parity-test hooks (tests/ load librevo_exp.so for them; the product library has no way to stop a forward early, to
 * read intermediate buffers, or to switch the exactness machinery of a search off):
 * run only the first n transformer blocks (n = -1: the whole forward; n = -2: stop after the patch embedding, before
 * ln_pre) and copy the fp32 residual stream [batch*seq, width] of the last forward to dst (device). */
int32_t revo_vit_set_debug_layers(revo_vit* vit, int32_t n_layers);
int32_t revo_vit_read_residual(revo_vit* vit, int32_t batch, float* dst, void* stream);
/* other intermediate buffers of the last forward (device to device): which = 0 the residual stream (fp32 [batch*seq, width]),
 * 1 the ln_post output after a whole forward (fp32 [batch*seq, width]: the head works in fp32), 2 the attention-pool
 * output after its MLP residual, before proj (fp32 [batch, width]) */
int32_t revo_vit_read_tap(revo_vit* vit, int32_t which, int32_t batch, void* dst, void* stream);
/* how the certificate treats the handle's searches: 0 = certificate + fallback (the product library's only behaviour),
 * 1 = every query takes the collecting pass, 2 = every query takes the brute-force pass (1 and 2: parity tests of the
 * fallback against the fast path), 3 = certificate evaluated and counted but no fallback (timing only: NOT exact) */
int32_t revo_search_set_mode(revo_gallery* g, int32_t mode);
/* phase groups of the persistent 256 x 256 GEMM (an experiment, measured in round 5 and not adopted): 0 / 1 = off (all
 * workgroups in step), 2..4 = that many groups, a workgroup of group g doing the first (g + 1) / groups of its first tile at the start and the
 * rest of that tile last.  Result-preserving (bit-identical). */
int32_t revo_op_set_phase_groups(int32_t groups);
/* 0: the persistent body GEMMs' bf16 epilogues on the round-5 kernel (LDS-transposed stores, drained before the next main loop);
 * 1 (default): the queued-stores kernel.  Same bits either way (A/B timing, parity of the two kernels). */
int32_t revo_op_set_qstores(int32_t on);
/* diagnostic: device array [workgroups][items][4] of uint64 the phased kernel fills with 100 MHz time stamps (main loop
 * begin, main loop end, epilogue issued) and the piece's rows, for its first `items` pieces per workgroup; NULL = off */
int32_t revo_debug_gemm_stamps(void* buf, int32_t items);
/* diagnostic: device array [workgroups][2] of uint64 the body attention kernel fills with the shader-clock ticks and the 100 MHz
 * ticks of each workgroup's lifetime (their ratio x 100 MHz = the clock the chip holds under this kernel); NULL = off */
int32_t revo_debug_attention_clock(void* buf);
/* 0 = size heuristic (default), 128 or 256 = force that GEMM tile */
int32_t revo_op_set_gemm_tile(int32_t tile);
/* bits 4-7 = force the XCD arrangement (N-stripes 1, 2, 4 or 8; 0 = heuristic), bits 8-11 = force the attention
 * waves per workgroup, bit 12 = disable the GEMM tail split, bit 16 = one workgroup per tile instead of the persistent
 * 256 x 256 GEMM, bit 17 = no split-K for the leftover rows of a residual GEMM, bit 18 = 256 x 256 tiles also for
 * problems with fewer than 100 of them, bit 19 = the two-buffer 128 x 64 kernel instead of its six-deep-ring form,
 * bit 20 = head_dim-64 attention on v_mfma_f32_16x16x32_bf16 instead of 32x32x16 (round 6's MFMA-shape experiment);
 * 0 = normal */
int32_t revo_op_set_variant(int32_t flags);
/* 0 = ln_1 / ln_2 as LayerNorm kernels in front of their GEMMs instead of the folded form (A/B timing, parity of one
 * against the other); 1 = default (folded, the residual stream in two bf16 planes between the folded GEMMs); 2 = folded
 * with the stream as fp32 rows plus a bf16 copy */
int32_t revo_op_set_ln_fold(int32_t on);
/* Timing experiments.  The variant bits above plus: bit 0 = skip the GEMM epilogue stores, bit 1 = skip the GEMM main loop,
 * bit 13 = skip the scan's selection, bit 15 = skip the scan's slow path (all four: WRONG RESULTS),
 * bit 14 = count scan events for revo_debug_scan_stats. */
int32_t revo_op_set_gemm_debug(int32_t flags);
/* The phases of the scan launch of a search of Q queries over `rows` scanned gallery rows (host logic only, no device):
 * out[0] = phases, out[1] = segment slots per query, then per phase: first block, first query tile, query tiles, slices.
 * Returns the number of workgroups of the launch (-1: bad arguments). */
int64_t revo_debug_scan_plan(int32_t Q, int64_t rows, int64_t* out, int32_t cap);
/* 1 if a forward of `batch` images keeps the residual stream in two bf16 planes between its folded GEMMs (reporting) */
int32_t revo_debug_stream_in_planes(const revo_vit* vit, int32_t batch);
/* copies bytes of the handle's search workspace to host_dst (host_dst NULL: returns the workspace size) */
int64_t revo_debug_read_workspace(revo_gallery* g, int64_t offset, int64_t bytes, void* host_dst);
/* experiment: bounds = device array [Q] of order-preserving u32 scan scores (the format revo_search_candidates publishes)
 * that the next searches on this handle take as lower limits of their admission bounds (NULL: off).  Stands in for a
 * bound exchanged between the shards before the scan; the caller guarantees each is <= the query's true ksel-th score. */
int32_t revo_debug_seed_bounds(revo_gallery* g, const uint32_t* bounds);
/* experiment (scripts/gridbar_probe.py): `iters` stages of "every workgroup writes n16 x 16 bytes, a device-wide
 * synchronisation, every workgroup reads another XCD's slice", as ONE launch with grid barriers (mode 0: a counter, mode 2: a flag word per workgroup, mode 3: XCD-hierarchical counters; the grid must be
 * co-resident; the spin is bounded: ctr_err[1] != 0 afterwards = timed out (1) or read stale data (2)) or as 2 x iters launches
 * (mode 1).  ctr_err: 1696 device words; stamps: 2 x iters uint64 (100-MHz clock of workgroup 0 around each barrier) or NULL. */
int32_t revo_probe_gridbar(int32_t mode, void* ctr_err, void* buf, int32_t n16, int32_t blocks, int32_t threads,
                           int32_t iters, float* sink, void* stamps, void* stream);
/* counters of the fused scan when debug bit 14 is set */
int32_t revo_debug_scan_stats(int64_t* out8);
#endif
/* (w / b NULL: normalise only -- how the forward runs the LayerNorms it cannot fold: their gain and shift live in the weights
 * of the linear layer behind them) */
int32_t revo_op_layernorm(const float* x, int64_t ldx, const float* w, const float* b, float eps, int32_t rows,
                          int32_t width, void* out, int64_t ldo, int32_t out_is_bf16, void* stream);
/* ln_post with the attention pool's logits out of the same pass (K9 + K10's scores): out = LayerNorm(x) in fp32 and
 * logits[(r / seq * heads + h) * seq + r % seq] = out[r] . qk[h] + ck[h] (qk [heads, width], ck [heads]; rows % seq == 0).
 * A row's results do not depend on how many rows the call has (a batch takes four rows per wave): same bits. */
int32_t revo_op_layernorm_logits(const float* x, int64_t ldx, const float* w, const float* b, float eps, int32_t rows,
                                 int32_t width, float* out, int64_t ldo, const float* qk, const float* ck, int32_t heads,
                                 int32_t seq, float* logits, void* stream);
/* the head's fp32 linear layers (head.hip gemm_f32_skinny_kernel; M = batch rows, weights [N, K] streamed once):
 * C = epi(A . Wt^T + bias), epi 0 = none, 1 = exact GELU, 2 = added to what C holds.  K % 16 == 0, rows 16-byte aligned.
 * K is cut into the same sixteen ranges, summed in the same order, whatever M is: a row's result does not depend on the
 * other rows of the call, bit for bit. */
int32_t revo_op_linear_f32(int32_t epi, const float* A, int64_t lda, const float* Wt, int64_t ldw, const float* bias, int32_t M,
                           int32_t N, int32_t K, float* C, int64_t ldc, void* stream);
/* the attention pool's weighted row sums (K10; head.hip): u[(b * heads + h) * width + c] = sum_s softmax_s(logits[b, h, :])[s] *
 * x[b * seq + s][c], all fp32; logits [batch, heads, seq].  A column's sum is taken in one fixed order whatever the batch
 * (the batch only decides how many columns a lane carries): bit-identical between a batch and its images one at a time. */
int32_t revo_op_pool_rows(const float* x, int64_t ldx, const float* logits, int32_t batch, int32_t seq, int32_t width,
                          int32_t heads, float* u, void* stream);
int32_t revo_op_rope(void* qkv_bf16, int64_t ld, const float* cos_sin, int32_t rows, int32_t seq, int32_t width,
                     int32_t heads, void* stream);
int32_t revo_op_attention(const void* qkv_bf16, int64_t ld, void* out_bf16, int64_t ldo, int32_t batch, int32_t seq,
                          int32_t heads, int32_t head_dim, void* stream);
int32_t revo_op_f32_to_bf16(const float* src, int64_t ld_src, void* dst_bf16, int64_t ld_dst, int64_t rows,
                            int32_t cols, void* stream);

/* ---- per-kernel-class device timing (HIP events on the launch stream) for bench.py.
 * on: 0 = off, 1 = every kernel class, 2 = only the four body GEMM classes (the roofline kernel),
 * 3 = every fourth launch of each of those classes (all layers have the same shapes);
 * an event pair costs a few microseconds of stream time per kernel, so the timed region uses 3. */
int32_t revo_prof_enable(int32_t on);
int32_t revo_prof_reset(void);
/* writes a JSON object {"class": {"launches": n, "ms": t}, ...} into buf */
int32_t revo_prof_report(char* buf, int32_t capacity);

/* ---- preprocessing: crop + squash-resize on the device ----------
 * Replaces the host-side  self.preprocess(image_pil.convert("RGB"))  resize of
 * core_system.py:335 / :439 (transform built at :200) for frames that are already decoded to
 * uint8 RGB in device memory, and implements the per-region crop the reference leaves as a
 * placeholder (core_system.py:406 "Use global for now"; crop idea at :687-690).
 * Bit-identical to PIL's  Image.crop(box).resize((S, S), Image.BILINEAR).
 * `jobs` is a HOST array; src pointers are device memory, interleaved RGB (H x W x 3).
 * out: device uint8 [n][3][out_size][out_size], ready for revo_vit_forward(image_dtype = 1).
 * Asynchronous on `stream` like everything else (`jobs` is read before the call returns; the source frames and `out`
 * must stay valid until the stream has run the work).  Calls on one device share a workspace: calls on one stream
 * are ordered by it, a call on another stream first waits for the previous call's stream. */
typedef struct revo_crop_job {
    const uint8_t* src;      /* device pointer to the top-left pixel of the source image */
    int32_t height, width;   /* source image size in pixels */
    int64_t row_stride;      /* bytes between source rows (>= width * 3) */
    int32_t x0, y0, x1, y1;  /* crop box, half-open [x0, x1) x [y0, y1); the whole image = 0, 0, width, height */
} revo_crop_job;
int32_t revo_preprocess_crop_resize(const revo_crop_job* jobs, int32_t n, int32_t out_size, uint8_t* out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* REVO_H */
