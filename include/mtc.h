/*
 * libmtc — C ABI of the MI355X-native Matryoshka-transcoder hot path.
 *
 * The reference (a1842806/matryoshka_transcoder) is pure Python on PyTorch and has no FFI
 * layer; its boundary is the Python surface of `MatryoshkaTranscoder` (src/models/sae.py:73-406)
 * and the step body of `train_transcoder` (src/training/training.py:172-183).  Every entry
 * point below names the reference lines it replaces.  The Python package
 * `matryoshka_transcoder_amd` binds these with ctypes.
 *
 * Conventions
 *   - plain C types only; every pointer is a DEVICE pointer unless the name ends in `_host`;
 *   - the caller owns every buffer (inputs, outputs, workspace); the library allocates nothing
 *     and keeps no state besides a thread-local error string;
 *   - every call is asynchronous on `stream` (a hipStream_t passed as void*) and never
 *     synchronises; scalars that depend on device results stay on the device;
 *   - return value 0 = success, negative = error (mtc_last_error() has the text);
 *   - matrices are row-major and contiguous.  The encoder weight is passed FEATURE-MAJOR:
 *     `W_encT[D][A_s]`, i.e. the transpose of the reference's `W_enc[A_s][D]` (the Python layer
 *     keeps the `W_enc` parameter as a transposed view of this storage, so state_dict shapes
 *     and values are the reference's);
 *   - `dtype` is the model dtype of parameters / activations: MTC_BF16 or MTC_F32.
 */
#ifndef MTC_H
#define MTC_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MTC_VERSION 1
#define MTC_MAX_PREFIX 16

enum { MTC_F32 = 0, MTC_BF16 = 1 };

/* Problem description shared by all ops. */
typedef struct mtc_desc {
    int32_t B;                       /* batch rows (tokens)                                    */
    int32_t A_s;                     /* source activation width  (multiple of 8)               */
    int32_t A_t;                     /* target activation width  (multiple of 8, <= 8192)      */
    int32_t D;                       /* dictionary size held by this process (local shard)     */
    int32_t G;                       /* number of nested prefix groups (<= MTC_MAX_PREFIX)     */
    int32_t prefix[MTC_MAX_PREFIX];  /* cumulative prefix sizes in LOCAL feature coordinates    */
    int32_t dtype;                   /* MTC_F32 | MTC_BF16                                     */
    int32_t unit_norm;               /* cfg["input_unit_norm"]                                 */
} mtc_desc;

/* Device-resident state of the global select (one per call chain; 64 bytes). */
typedef struct mtc_select_state {
    uint32_t key_prefix;   /* bits of tau fixed so far (float32 bit pattern, upper bits)      */
    uint32_t k_rem;        /* how many entries still to take inside the current bin           */
    uint32_t all_pos;      /* 1: fewer than k positives exist -> every positive is selected   */
    uint32_t n_gt;         /* entries strictly greater than tau                               */
    uint32_t tau_bits;     /* float32 bit pattern of tau (valid after the last level)         */
    uint32_t n_tie_take;   /* entries equal to tau that are selected                          */
    uint32_t min_sel_bits; /* float32 bits of the smallest selected value (atomicMin)         */
    uint32_t nnz;          /* number of selected entries                                      */
    uint32_t pad[8];
} mtc_select_state;

const char* mtc_last_error(void);
int mtc_version(void);

/* a2  preprocess_input (sae.py:139-147): x <- (x - mean_row) / (std_row_unbiased + 1e-5), each step
 * rounded to the model dtype as the reference's eager ops do.  mean/std: [B] in the model dtype. */
int mtc_row_normalize(const mtc_desc* d, const void* x, void* x_norm, void* mean, void* std,
                      void* stream);

/* a3  compute_activations, dense part (sae.py:173-174): acts = relu(round(round(x.W_enc) + b_enc)),
 * acts [B][D] in the model dtype.  bf16: MFMA 16x16x32, fp32 accumulate; fp32: f32 MFMA. */
int mtc_encode_fwd(const mtc_desc* d, const void* x, const void* W_encT, const void* b_enc,
                   void* acts, void* stream);

/* The training loop's encoder: the same launch also MARKS, per batch row and 128-feature block, which 16-feature
 * granules may hold an activation >= the select's carried bound carry[0] (mtc_select_global_topk_carry): the first pass
 * of sae.py:176-182's topk over the B*D activations happens in the epilogue that holds them anyway, and the select's
 * candidate pass (mtc_select_global_topk_carry_marked) reads only the flagged granules instead of all of acts.
 * marks: mtc_encode_marks_bytes(d) bytes = uint8 [B][ceil(D/128)] + the bound they were made for; 0 bytes = this shape /
 * dtype does not run the marking kernel (small problems, fp32): use mtc_encode_fwd and pass marks = NULL to the select. */
size_t mtc_encode_marks_bytes(const mtc_desc* d);
int mtc_encode_fwd_marked(const mtc_desc* d, const void* x, const void* W_encT, const void* b_enc,
                          void* acts, const uint32_t* carry, void* marks, void* stream);

/* a4 / a15  eval-mode encode (sae.py:189-191 inside encode, sae.py:363-387): the same encoder with
 * where(acts > threshold, acts, 0) applied in its epilogue, so the dense feature tensor `feats` [B][D] the API
 * returns is written once and never re-read.  threshold: device scalar in the model dtype (the module's buffer). */
int mtc_encode_threshold_infer(const mtc_desc* d, const void* x, const void* W_encT, const void* b_enc,
                               const void* threshold, void* feats, void* stream);

/* Encoder tile override (no counterpart in the reference: a tuning / test knob).  0 = automatic choice by problem
 * size (default; the environment variable MTC_ENCODE_TILE = 1 | 2 is read once at the first launch), 1 = always the
 * 128 x 128 MFMA kernel, 2 = always the 256 x 256 MFMA kernel (bf16, A_s a multiple of 64).  Returns the previous
 * setting, or -1 for an invalid argument.  Process-wide; not synchronised with launches on other threads. */
int mtc_encode_set_tile(int tile);

/* a3  global batch-TopK (sae.py:176-187) as an exact k-th-largest threshold select over the
 * B*D activations; a5 threshold EMA input (sae.py:358-360) falls out as min_sel_bits.
 * Phases are separate entry points so a feature-sharded caller can all-reduce `hist` between
 * them; mtc_select_global_topk() chains them for one GPU.
 *   level 0: float bits [30:19]   level 1: [18:7] (bf16: [18:16])   level 2: [6:0] (fp32 only)
 * hist: 4096 uint32 (caller zeroes it; mtc_select_hist adds into it). */
int mtc_select_levels(const mtc_desc* d);
int mtc_select_init(mtc_select_state* st, uint32_t k_total, void* stream);
int mtc_select_hist(const mtc_desc* d, const void* acts, int level, const mtc_select_state* st,
                    uint32_t* hist, void* stream);
int mtc_select_scan(const mtc_desc* d, int level, const uint32_t* hist, mtc_select_state* st,
                    void* stream);
/* bf16 shortcut for levels >= 1: after level 0 only 3 key bits remain, so ONE pass per row counts
 * the entries above the level-0 bin (row_above[B]) and the 8 sub-bin populations inside it
 * (row_sub[B][8]); mtc_select_finish_rows then fixes tau / the tie quota from the sub-bin totals
 * (global_sub[8]: all-reduced totals when sharded, NULL = sum row_sub locally) and derives the
 * per-row (> tau, == tau) counts without touching acts again. */
int mtc_select_rowsub(const mtc_desc* d, const void* acts, mtc_select_state* st, uint32_t* row_above,
                      uint32_t* row_sub, void* stream);
int mtc_select_finish_rows(const mtc_desc* d, mtc_select_state* st, const uint32_t* row_above,
                           const uint32_t* row_sub, const uint32_t* global_sub, uint32_t* row_gt,
                           uint32_t* row_eq, void* stream);
/* per-row counts of values > tau and == tau (or, threshold mode, > *threshold: sae.py:189-191;
 * a NaN *threshold selects every non-zero entry: dense -> CSR for decode()) */
int mtc_select_count(const mtc_desc* d, const void* acts, const mtc_select_state* st,
                     const void* threshold /* NULL in TopK mode; model-dtype scalar otherwise */,
                     uint32_t* row_gt, uint32_t* row_eq, void* stream);
/* row_ptr[B+1] and per-row tie quota from the counts.  tie_before (optional, [B]) = ties that
 * precede this shard's row b in the global (row, shard) order; NULL = exclusive scan of row_eq. */
int mtc_select_offsets(const mtc_desc* d, const uint32_t* row_gt, const uint32_t* row_eq,
                       const uint32_t* tie_before, mtc_select_state* st, int32_t* row_ptr,
                       uint32_t* tie_take, void* stream);
/* ordered compaction: col_idx ascending inside each row; vals in the model dtype */
int mtc_select_fill(const mtc_desc* d, const void* acts, const mtc_select_state* st,
                    const void* threshold, const int32_t* row_ptr, const uint32_t* tie_take,
                    int32_t* col_idx, void* vals, void* stream);
size_t mtc_select_workspace_bytes(const mtc_desc* d);
int mtc_select_global_topk(const mtc_desc* d, const void* acts, uint32_t k_total,
                           const void* threshold /* NULL = TopK mode */, void* workspace,
                           size_t workspace_bytes, mtc_select_state* st, int32_t* row_ptr,
                           int32_t* col_idx, void* vals, void* stream);

/* Training-loop variant of mtc_select_global_topk (TopK mode only; same outputs, bit for bit).
 * tau moves slowly between steps (sae.py:176-187 runs once per step on weights one Adam update
 * apart), so `carry` -- 8 uint32 on the device, zeroed once by the caller and then left alone --
 * keeps the previous tau minus a margin as a lower bound.  One pass over acts copies the entries
 * >= bound into a compact [B][min(D,1024)] matrix and the exact select runs on that; if the bound
 * turns out wrong (too few candidates, or a row with more than fit) the full-size passes run
 * instead.  Both sets of launches are enqueued and predicated on a device flag: no read-back.
 * bf16: a key has 16 bits, so the candidate pass also counts each row's candidates by their distance
 * from the bound in ulps (32 bins) and ONE single-workgroup launch derives tau, the tie quota, the
 * CSR offsets, the next bound (and the threshold EMA, below) from those counts; that form also
 * needs tau within 30 ulps of the bound, i.e. a rise of at most 22 ulps in one step.
 * carry[4] / carry[5] count the calls served by the carried bound / by the full-size passes. */
size_t mtc_select_carry_workspace_bytes(const mtc_desc* d);
int mtc_select_global_topk_carry(const mtc_desc* d, const void* acts, uint32_t k_total,
                                 void* workspace, size_t workspace_bytes, uint32_t* carry,
                                 mtc_select_state* st, int32_t* row_ptr, int32_t* col_idx,
                                 void* vals, void* stream);
/* The same with a5 folded in: threshold (the module's buffer, model dtype; NULL = leave it alone) receives the EMA update
 * threshold <- (1 - ema_lr) threshold + ema_lr min(selected) (sae.py:349-361) inside the select's single-workgroup tail
 * launch (per-row counts -> offsets -> next bound -> EMA are ONE launch here) instead of a mtc_threshold_ema launch. */
int mtc_select_global_topk_carry_ema(const mtc_desc* d, const void* acts, uint32_t k_total,
                                     void* workspace, size_t workspace_bytes, uint32_t* carry,
                                     mtc_select_state* st, int32_t* row_ptr, int32_t* col_idx,
                                     void* vals, void* threshold, float ema_lr, void* stream);

/* ... with the encoder's marks (mtc_encode_fwd_marked; NULL = stream all of acts).  Marks made for another bound than the
 * one carried now are ignored; the result is the same bit for bit either way. */
int mtc_select_global_topk_carry_marked(const mtc_desc* d, const void* acts, uint32_t k_total,
                                        void* workspace, size_t workspace_bytes, uint32_t* carry,
                                        mtc_select_state* st, int32_t* row_ptr, int32_t* col_idx,
                                        void* vals, void* threshold, float ema_lr, const void* marks,
                                        void* stream);

/* Helpers of the feature-sharded caller (no counterpart in the single-process reference; each replaces a
 * handful of tiny tensor ops between two collectives):
 *   mtc_select_offsets_sharded : mtc_select_offsets with the tie order of a sharded dictionary -- ties are
 *       granted in (row, shard, feature) order from all_eq [world][eq_stride], the all-gathered per-row tie
 *       counts; with eq_stride > B, all_eq[w][B] = shard w's smallest selected value (bits) and minmax_out
 *       (optional) receives the pair mtc_threshold_ema_global consumes -- no extra collective for the EMA;
 *   mtc_select_pack_minmax     : out2 = {min selected bits, -nnz} (int64) for ONE all-reduce(MIN);
 *   mtc_threshold_ema_global   : a5 from that reduced pair (skipped when nothing was selected anywhere);
 *   mtc_shard_nnz_abs          : out2 = {nnz of this shard, sum |a| of its entries} for one all-reduce(SUM)
 *       that feeds mtc_reduce_scalars(nnz_abs_override). */
int mtc_select_offsets_sharded(const mtc_desc* d, const uint32_t* row_gt, const uint32_t* all_eq,
                               int32_t eq_stride, int32_t world, int32_t rank, mtc_select_state* st,
                               int32_t* row_ptr, uint32_t* tie_take, int64_t* minmax_out,
                               void* stream);
int mtc_select_pack_minmax(const mtc_select_state* st, int64_t* out2, void* stream);

/* The global batch-TopK of sae.py:176-187 over a FEATURE-SHARDED dictionary (the reference is single-process),
 * in four phases with the caller's three fixed-shape collectives between them; nothing is read back.  The selection is
 * the single-GPU one bit for bit whatever the number of shards: ties at tau are granted in ascending GLOBAL flat index
 * (row, then feature = prefix group, shard, local feature).  d describes the LOCAL shard (d->D, d->prefix local).
 *   begin  : zeroes the exchange buffers, inits st; bf16: one pass over acts keeps the entries >= the carried bound
 *            (carry[0], as in mtc_select_global_topk_carry) in a compact matrix and histograms them into the 4096-bin
 *            WINDOW xbuf[bin = key16 - bound16] + {xbuf[4096] candidates, xbuf[4097] overflowing rows}.
 *            -> X1: all-reduce(SUM) of xbuf[0 .. 4104) as uint32.
 *            fp32: init only; the caller then runs mtc_select_hist / all-reduce / mtc_select_scan per level on
 *            xbuf + level * 4096 (xbuf holds mtc_select_sharded_xbuf_words() = 3 * 4096 words).
 *   window : (bf16) the reduced window decides tau / the tie quota, or sets carry[1] = 0 (bound failed somewhere: too
 *            few candidates globally, an overflowing row, tau in the clamped top bin); only then a second kernel fills
 *            hist_full[32768] (the whole bf16 key space of the local activations).
 *            -> X2: all-reduce(SUM) of hist_full (all zeros when the window served the call).
 *   counts : (bf16, carry[1] == 0: tau from the reduced hist_full) then per row the entries > tau and, per prefix group,
 *            the entries == tau -> pack[B * G + 1] (last word: the shard's smallest positive when fewer than k positives
 *            exist globally, else ~0).  -> X3: all-gather of pack.
 *   finish : row_ptr / tie quotas from all_pack [world][B * G + 1], st->nnz, threshold EMA (a5, sae.py:349-361) from the
 *            GLOBAL smallest selected value (threshold may be NULL), the next call's bound, and the ordered fill.
 * workspace: mtc_select_sharded_workspace_bytes(d), the SAME buffer for all four phases of one call. */
size_t mtc_select_sharded_workspace_bytes(const mtc_desc* d);
size_t mtc_select_sharded_xbuf_words(void);
size_t mtc_select_sharded_full_bins(void);
int mtc_select_sharded_begin(const mtc_desc* d, const void* acts, uint32_t k_total, void* workspace,
                             size_t workspace_bytes, uint32_t* carry /* NULL for fp32 */, mtc_select_state* st,
                             uint32_t* xbuf, uint32_t* hist_full /* NULL for fp32 */, uint32_t* pack,
                             const void* marks /* mtc_encode_fwd_marked's, or NULL */, void* stream);
int mtc_select_sharded_window(const mtc_desc* d, const void* acts, uint32_t k_total, uint32_t* carry,
                              mtc_select_state* st, const uint32_t* xbuf, uint32_t* hist_full, void* stream);
int mtc_select_sharded_counts(const mtc_desc* d, const void* acts, uint32_t k_total, void* workspace,
                              const uint32_t* carry, mtc_select_state* st, const uint32_t* hist_full,
                              uint32_t* pack, void* stream);
int mtc_select_sharded_finish(const mtc_desc* d, const void* acts, void* workspace, uint32_t* carry,
                              mtc_select_state* st, const uint32_t* all_pack, int32_t world, int32_t rank,
                              int32_t* row_ptr, int32_t* col_idx, void* vals, void* threshold, float ema_lr,
                              void* stream);
int mtc_threshold_ema_global(const mtc_desc* d, const int64_t* minmax2, void* threshold, float lr,
                             void* stream);
int mtc_shard_nnz_abs(const mtc_desc* d, const float* abs_rows, const mtc_select_state* st,
                      float* out2, void* stream);

/* a5  update_threshold (sae.py:349-361): threshold <- (1-lr)*threshold + lr*min_selected, evaluated
 * in the buffer dtype, skipped when nothing was selected. */
int mtc_threshold_ema(const mtc_desc* d, const mtc_select_state* st, void* threshold, float lr,
                      void* stream);

/* a6/a8/a10  nested prefix decode (sae.py:217-230) in sparse form, one wavefront per batch row:
 * x_hat_i = sum_{j < P_i} a_bj W_dec[j] + b_dec accumulated incrementally per prefix group,
 * de-normalised (sae.py:149-153) and rounded to the model dtype; fused per-row partials of the
 * per-prefix squared error (losses sae.py:236, FVU config.py:222-232) and of sum(x_t), sum(x_t^2).
 *   x_hat     [B][A_t] model dtype (last prefix)          row_stats [G+4][B] float:
 *   g         [B][A_t] model dtype, = 2/(B*A_t)*(x_hat-x_t)*std_row  rows 0..G-1  SSE of prefix i
 *             (NULL = forward only)                                  row G sum x_t, G+1 sum x_t^2,
 *   recons    optional [G][B][A_t] model dtype (every prefix)        row G+2 sum |a|, G+3 unused
 * Pure decode (a15, sae.py:389-406): x_target = g = row_stats = d_vals = NULL. */
int mtc_decode_fwd(const mtc_desc* d, const int32_t* row_ptr, const int32_t* col_idx,
                   const void* vals, const void* W_dec, const void* b_dec, const void* x_target,
                   const void* mean, const void* std, void* x_hat, void* g, float* row_stats,
                   void* recons, float* d_vals /* NULL, or [nnz]: fused a11 sweep, see
                   mtc_decode_bwd_vals */, float l1_term, void* stream);

/* feature-sharded variant.  mtc_decode_partial writes this shard's bias-free cumulative sums:
 *   partial_last [B][A_t] fp32 (last prefix: feeds the loss and the gradient), and
 *   partial_pre  [(B/rows_per_shard)][G-1][rows_per_shard][A_t] model dtype (earlier prefixes: FVU only),
 * grouped by the rank that will own the rows so that a reduce-scatter hands each rank one
 * contiguous block.  mtc_decode_finish then runs the decode epilogue on THIS rank's rows
 * (d->B = local rows, B_total = global batch for the 2/(B*A_t) factor): partial_pre
 * [G-1][rows][A_t], x_target / mean / std / x_hat / g / row_stats[G+4][rows] all local. */
int mtc_decode_partial(const mtc_desc* d, const int32_t* row_ptr, const int32_t* col_idx,
                       const void* vals, const void* W_dec, int32_t rows_per_shard,
                       float* partial_last, void* partial_pre,
                       float* abs_rows /* [B] sum |a| of the local entries, or NULL */, void* stream);
/* The same for ONE ROW BLOCK of a batch that is decoded block by block (so that the reduce-scatter of block i runs beside
 * the decode of block i+1): d->B = rows of the block, row_ptr / partial_last / abs_rows point at the block's first row
 * (row_ptr holds absolute offsets), partial_pre at the block's slot of ONE exchange buffer
 * [shard][block][G-1][rows_per_shard][A_t] whose shard stride (in elements) is pre_shard_stride (0 = dense). */
int mtc_decode_partial_strided(const mtc_desc* d, const int32_t* row_ptr, const int32_t* col_idx,
                               const void* vals, const void* W_dec, int32_t rows_per_shard,
                               int64_t pre_shard_stride, float* partial_last, void* partial_pre,
                               float* abs_rows, void* stream);
/* mtc_decode_finish may be called in two parts so that the exchange of the earlier prefixes (FVU metrics only) can run
 * beside the backward kernels: partial_pre = NULL finishes the last prefix only (x_hat, g, its SSE row, the target
 * moments); partial_last = NULL (x_hat / g may be NULL) writes only the SSE rows of prefixes 0 .. G-2. */
int mtc_decode_finish(const mtc_desc* d, int32_t B_total, const float* partial_last,
                      const void* partial_pre, const void* b_dec, const void* x_target,
                      const void* mean, const void* std, void* x_hat, void* g, float* row_stats,
                      void* recons, void* stream);

/* a8/a10/a14  scalar epilogue: reduces row_stats (and the optional aux partials) to the
 * reference's output scalars without leaving the device.
 *   scalars (float[16]): 0 l2  1 l1  2 l0  3 l1_norm  4 aux  5 loss  6 fvu  7 fvu_min  8 fvu_max
 *                        9 fvu_mean  10 var(x_t)  11 nnz  12..15: fp32 descriptors 0; bf16 descriptors 8 bf16 values
 *                        {l2, l1, l1_norm, 0 (aux while no feature is dead), loss of the aux-free form with every op
 *                        rounded (sae.py:245), 0, 0, 0}: the model-dtype scalars of the output dict without a conversion
 * l2/l1/l1_norm are rounded to the model dtype (sae.py:236-239), the rest are fp32.
 * n_rows_total/n_feat_total: global B and D (they differ from d->D when sharded). */
int mtc_reduce_scalars(const mtc_desc* d, const float* row_stats, const float* aux_rows /*[B] or NULL*/,
                       const mtc_select_state* st,
                       const float* nnz_abs_override /* NULL, or device float[2] {global nnz, global
                       sum |a|} when the selection is sharded */,
                       int32_t n_feat_total, float l1_coeff, float aux_penalty, float* scalars,
                       void* stream);

/* a11  d a_bj = round(g_b . W_dec[j]) (+ l1_coeff/(B*D), accumulated in the model dtype) for every selected entry, CSR
 * order.  Rounded to the model dtype like the reference's autograd holds it (the output of g @ W_dec^T; a no-op for
 * fp32 models), stored as float. */
int mtc_decode_bwd_vals(const mtc_desc* d, const int32_t* row_ptr, const int32_t* col_idx,
                        const void* W_dec, const void* g, float l1_term, float* d_vals,
                        void* stream);

/* a7  update_inactive_features (sae.py:165-169) + feature-major index of the selection.
 * col_ptr[D+1]; csc_row[nnz] batch row of each entry; csc_pos[nnz] its position in the CSR
 * arrays.  cnt (float[D]) += 1 where the feature did not fire, = 0 where it did (NULL = skip).
 * Entries of a feature are ordered by batch row (deterministic accumulation order downstream).
 * workspace: mtc_csc_workspace_bytes(). */
size_t mtc_csc_workspace_bytes(const mtc_desc* d);
int mtc_build_csc(const mtc_desc* d, const int32_t* row_ptr, const int32_t* col_idx,
                  void* workspace, size_t workspace_bytes, int32_t* col_ptr, int32_t* csc_row,
                  int32_t* csc_pos, float* cnt, void* stream);
/* The same in two enqueues on one workspace (mtc_build_csc = part 1, then part 2): part 1 leaves col_ptr and the updated
 * counters cnt (sae.py:165-169) behind, part 2 fills and orders csc_row / csc_pos.  A step with the auxiliary loss active
 * enqueues mtc_aux_dead_list between the two, so that the dead list (sae.py:330-331) is ready before the decode launch
 * that runs beside this chain has drained. */
int mtc_build_csc_part(const mtc_desc* d, const int32_t* row_ptr, const int32_t* col_idx,
                       void* workspace, size_t workspace_bytes, int32_t* col_ptr, int32_t* csc_row,
                       int32_t* csc_pos, float* cnt, int32_t part, void* stream);

/* a11  k-sparse backward (autograd of sae.py:173-245, training.py:172), one wavefront per feature:
 *   dW_dec[j]  = scale * sum_b a_bj g_b          dW_encT[j] = scale * sum_b d_a_bj x_b
 *   db_enc[j]  = scale * sum_b d_a_bj            db_dec     = scale * sum_b g_b
 * grads are dense [D][A] in the model dtype (rows of silent features are written as zeros);
 * `scale` points to the upstream d loss (device float, NULL = 1).  accumulate != 0 adds into the
 * existing gradient buffers (used for the aux contributions). */
int mtc_backward_sparse(const mtc_desc* d, const int32_t* col_ptr, const int32_t* csc_row,
                        const int32_t* csc_pos, const void* vals, const float* d_vals,
                        const void* g, const void* x_norm, const float* scale, void* dW_dec,
                        void* dW_encT, void* db_enc, int accumulate, void* stream);
size_t mtc_colsum_workspace_bytes(const mtc_desc* d);
int mtc_colsum(const mtc_desc* d, const void* g, const float* scale, void* workspace,
               size_t workspace_bytes, void* db_dec, int accumulate, void* stream);
/* No counterpart in the reference: the ONE entry point of this library that allocates nothing and takes no
 * workspace, for callers that cannot call hipMalloc (code under HIP graph capture, tight serving loops, harnesses
 * that forbid allocation).  Same result, bit for bit, as mtc_colsum (db_dec = scale * sum_b g_b, sae.py:217-245
 * backward); the per-slab partial sums live in a static device array inside the library (64 slabs x 8192 columns
 * fp32 = 2 MiB, held by every GPU the library runs on until the process ends).  A_t > 8192 is rejected with an error.
 * The array's device address is looked up on the first call on each device (hipGetSymbolAddress; make that first
 * call outside a graph capture).  All calls on one GPU share the array: calls that overlap on one GPU (different
 * streams) race; nothing needs clearing between calls (every partial read is written by the same call). */
int mtc_colsum_static(const mtc_desc* d, const void* g, const float* scale, void* db_dec, int accumulate,
                      void* stream);

/* a12 + a13  make_decoder_weights_and_grad_unit_norm (sae.py:155-163) fused with torch.optim.Adam
 * (training.py:60,178; betas, eps 1e-8, no weight decay), op-by-op rounding in the parameter
 * dtype like PyTorch's single-tensor Adam.  `step` is 1-based.  Moments are in the param dtype. */
typedef struct mtc_adam_hyper {
    double lr, beta1, beta2, eps;   /* host doubles: bias corrections are formed in double like PyTorch */
    int32_t step;
    int32_t pad;
} mtc_adam_hyper;
int mtc_unitnorm_adam_rows(const mtc_desc* d, void* W_dec, void* grad, void* exp_avg,
                           void* exp_avg_sq, const mtc_adam_hyper* h_host, int write_grad,
                           void* stream);
int mtc_adam_flat(int32_t dtype, size_t n, void* p, const void* grad, void* exp_avg,
                  void* exp_avg_sq, const mtc_adam_hyper* h_host, void* stream);
/* a11 + a12 + a13 fused (training.py:172-179 as one pass per parameter row): the gradient row of
 * feature j is built in registers (as mtc_backward_sparse does), rounded to the parameter dtype
 * like a materialised .grad, and consumed by [projection +] Adam immediately; dense gradients never
 * reach HBM.  _dec: W_dec rows with the unit-norm projection; _enc: W_encT rows and b_enc.
 * b_dec still goes through mtc_colsum + mtc_adam_flat.  Not usable while the aux loss is active
 * (its contributions must be added to the gradient first: use mtc_backward_sparse then).
 * _dec with d_vals_out != NULL also writes d a_bj = g_b . W_dec[j] + l1_term (pre-update W_dec) for
 * every selected entry at its CSR position -- the same values as mtc_decode_bwd_vals, from the one
 * gather of g the decoder gradient needs anyway; mtc_decode_fwd is then called with d_vals = NULL.
 * Call _dec before _enc (which consumes d_vals). */
int mtc_fused_backward_adam_dec(const mtc_desc* d, const int32_t* col_ptr, const int32_t* csc_row,
                                const int32_t* csc_pos, const void* vals, const void* g,
                                const float* scale, void* W_dec, void* exp_avg, void* exp_avg_sq,
                                const mtc_adam_hyper* h_host, float* d_vals_out, float l1_term,
                                void* stream);
int mtc_fused_backward_adam_enc(const mtc_desc* d, const int32_t* col_ptr, const int32_t* csc_row,
                                const int32_t* csc_pos, const float* d_vals, const void* x_norm,
                                const float* scale, void* W_encT, void* exp_avg, void* exp_avg_sq,
                                void* b_enc, void* b_exp_avg, void* b_exp_avg_sq,
                                const mtc_adam_hyper* h_host, void* stream);

/* The same two launches with a dense extra contribution for some rows (the aux branch's GEMM results over the dead
 * features, see mtc_aux_dense_fill): extra_row[D] holds -1 or the row of extra_rows [.][A] (model dtype) to add to
 * feature j's rounded sparse gradient -- rounded again, as `grad[dead] += ...` on a materialised gradient would be --
 * before projection / Adam; extra_bias[.] likewise for the b_enc element.  No counterpart entry in the reference: it
 * is autograd's accumulation of the two branches' gradients (sae.py:245 loss = l2 + l1 + aux). */
int mtc_fused_backward_adam_dec_extra(const mtc_desc* d, const int32_t* col_ptr, const int32_t* csc_row,
                                      const int32_t* csc_pos, const void* vals, const void* g, const float* scale,
                                      void* W_dec, void* exp_avg, void* exp_avg_sq, const mtc_adam_hyper* h,
                                      float* d_vals_out, float l1_term, const int32_t* extra_row, const void* extra_rows,
                                      int32_t d_vals_csc_order, void* stream);
int mtc_fused_backward_adam_enc_extra(const mtc_desc* d, const int32_t* col_ptr, const int32_t* csc_row,
                                      const int32_t* csc_pos, const float* d_vals, const void* x_norm, const float* scale,
                                      void* W_encT, void* exp_avg, void* exp_avg_sq, void* b_enc, void* b_exp_avg,
                                      void* b_exp_avg_sq, const mtc_adam_hyper* h, const int32_t* extra_row,
                                      const void* extra_rows, const void* extra_bias, int32_t d_vals_csc_order, void* stream);
/* d_vals_csc_order != 0 (both launches of one step alike): d a_bj is written / read at the entry's CSC index instead of
 * its CSR position -- the order both launches walk -- and nothing else may read that buffer as CSR-ordered.  extra_row /
 * extra_rows may be NULL with it (the plain fused step).  With it, bf16 and A_s a multiple of 64 the encoder launch runs
 * in its COLUMN-SLICED form when the problem is large enough: the parameter matrix is walked slice by slice (256 / 128 /
 * 64 columns) in an XCD-contiguous order so that the slice of the gather table x [B][A_s] stays in each XCD's L2 and
 * the k-fold row gathers stop crossing the fabric; same arithmetic per element.  mtc_fused_set_sliced: 0 = automatic
 * (default; environment MTC_FUSED_SLICED read once), 1 = never, 2 = whenever the shape allows; returns the previous
 * setting or -1.  Process-wide tuning / test knob, no counterpart in the reference. */
int mtc_fused_set_sliced(int mode);
/* Measurement aid, no counterpart in the reference and not on the product path: lower bound of a column-sliced form of
 * the decoder-row gradient + Adam launch (pass 0: gather + gradient slice + partial row sums into scratch
 * [D][A_t/256][2] floats; pass 1: gather again + update without projection).  tools/fused_ab.py times both passes next
 * to the row kernel that is used.  Overwrites W_dec / the moments it is given (pass 1): hand it copies. */
int mtc_diag_sliced_decoder_bound(const mtc_desc* d, const int32_t* col_ptr, const int32_t* csc_row,
                                  const float* coef_csc, const void* g, void* W_dec, void* exp_avg,
                                  void* exp_avg_sq, float* scratch, int32_t pass, void* stream);
/* projection only (the reference method called on its own, sae.py:155-163) */
int mtc_unitnorm_project(const mtc_desc* d, void* W_dec, void* grad, void* stream);

/* a9  get_auxiliary_loss (sae.py:318-347), sparse.  n_dead stays on the device (int32).
 *   mtc_aux_dead_list : dead_idx[<=D] ascending = features with cnt >= n_batches_to_dead, *n_dead_dev
 *   mtc_aux_select    : per row, the min(top_k_aux, n_dead) largest POSITIVE acts among the dead
 *                       columns: count, threshold bits and tie quota per row (ties: ascending feature)
 *   mtc_aux_fill      : CSR (row_ptr from mtc_select_offsets on the counts) of those entries
 *   mtc_aux_fwd       : x_aux = sum a W_dec[j] (no bias); e = round(x_aux) - (x_t - x_hat);
 *                       aux_rows[b] = sum e^2; g_aux = 2 aux_penalty/(B A_t) e; g += g_aux*std_row
 *                       (residual not detached); d_aux_vals = g_aux . W_dec[j]
 * The aux gradients then reuse mtc_build_csc + mtc_backward_sparse(accumulate = 1). */
int mtc_aux_dead_list(const mtc_desc* d, const float* cnt, float n_batches_to_dead, int32_t* dead_idx /* [D]; the tail
                      beyond the count is filled with feature 0 (valid to gather through) */,
                      int32_t* n_dead_dev, int32_t* row_map /* optional [D]: slot of the feature in dead_idx, or -1 */,
                      int32_t* next_bound_dev /* optional: #features with cnt >= n_batches_to_dead - 1, an upper bound of
                      the NEXT forward's count (a counter grows by at most 1 per forward): lets the caller size the next
                      step's launches without waiting for that step's own count */,
                      void* stream);

/* a9 in the reference's own dense form over the compacted dead set (sae.py:336-341: acts_aux @ W_dec[dead], and the
 * three products of its autograd), on the encoder's MFMA tiles -- no BLAS library on the path:
 *   mtc_gemm_nt          C[m][n] = sum_k A[row(m)][k] Bt[n][k]; fp32 accumulation, ONE rounding to the model dtype
 *                        (out_f32 = 0) or fp32 output (partial sums all-reduced over shards); row strides lda / ldb /
 *                        ldc in elements; a_rows (optional) gathers A's rows (row(m) = a_rows[m]).  MFMA tiles when
 *                        K % 64 == 0 (bf16; fp32: K % 32 == 0), N % 4 == 0 and 16-byte-aligned rows, else an FMA kernel.
 *   mtc_transpose_rows   dst[c][r] = src[row(r)][c], r < R, c < C; dst columns R <= r < R_pad zero-filled (k-tile padding)
 *   mtc_aux_mask_rowsum  d[j][b] = 0 where acts_auxT[j][b] <= 0 (TopK / ReLU mask), e_b[j] = sum_b d[j][b]
 *   mtc_rows_scatter_add dst[idx[j]][:] = round(dst[idx[j]][:] + src[j][:]) (unique idx), the dense `grad[dead] += rows`
 * Launch sizes may come from an upper bound of n_dead (mtc_aux_dead_list's next_bound): the list's tail is gatherable,
 * padded columns of acts_aux are zero, and mtc_aux_residual / mtc_rows_scatter_add take the true count from the device. */
int mtc_gemm_nt(int32_t dtype, int32_t M, int32_t N, int32_t K, const void* A, int64_t lda, const int32_t* a_rows,
                const void* Bt, int64_t ldb, void* C, int64_t ldc, int32_t out_f32, void* stream);
int mtc_transpose_rows(int32_t dtype, int32_t R, int32_t C, int32_t R_pad, const void* src, int64_t ld_src,
                       const int32_t* src_rows, void* dst, int64_t ld_dst, void* stream);
int mtc_aux_mask_rowsum(int32_t dtype, int32_t n, int32_t B, const void* acts_auxT, int64_t ld_a, void* d_actsT,
                        int64_t ld_d, void* e_b, void* stream);
int mtc_rows_scatter_add(int32_t dtype, int32_t n, int32_t A, const int32_t* idx, const void* src, int64_t ld_src,
                         void* dst, int64_t ld_dst, int32_t dst_rows /* an idx outside [0, dst_rows) is skipped */,
                         const int32_t* n_dev /* optional: only rows j < *n_dev */, void* stream);
int mtc_aux_select(const mtc_desc* d, const void* acts, const int32_t* dead_idx,
                   const int32_t* n_dead_dev, int32_t top_k_aux, uint32_t* row_cnt,
                   uint32_t* row_tau_bits, uint32_t* row_tie_quota, void* stream);
int mtc_aux_fill(const mtc_desc* d, const void* acts, const int32_t* dead_idx,
                 const int32_t* n_dead_dev, const uint32_t* row_tau_bits,
                 const uint32_t* row_tie_quota, const int32_t* row_ptr, int32_t* col_idx, void* vals,
                 void* stream);
int mtc_aux_fwd(const mtc_desc* d, const int32_t* row_ptr, const int32_t* col_idx, const void* vals,
                const void* W_dec, const void* x_target, const void* x_hat, const void* std,
                float aux_penalty, void* g, void* g_aux, float* aux_rows, float* d_aux_vals,
                void* stream);

/* a9, dense (GEMM) form -- what the reference itself computes (sae.py:332-341: acts_aux @ W_dec[dead]).  With
 * n_dead dead features the aux branch is four GEMMs over the compacted dead set ([B, n_dead] x [n_dead, A]); for
 * every n_dead <= D these cost less on MFMA than gathering k_aux decoder rows per batch row (B*k_aux*A*s bytes).
 * libmtc provides the selection, the fused elementwise parts and the products themselves (mtc_gemm_nt above):
 *   mtc_aux_dense_fill : acts_aux[B][n_pad] (model dtype) = the row's top-k_aux dead activations in dead-list
 *                        column order (threshold / tie quota from mtc_aux_select), zeros elsewhere and in the padding;
 *   mtc_aux_residual   : given x_hat_aux = acts_aux @ W_dec[dead]: e = x_hat_aux - (x_target - x_hat) (residual not
 *                        detached), aux_rows[b] = sum e^2, g_aux = (2 aux_penalty / (B A_t)) e, g += g_aux * std_row;
 *   mtc_aux_mask       : d_acts_aux[i] = 0 where acts_aux[i] <= 0 (backward of the TopK scatter + ReLU). */
int mtc_aux_dense_fill(const mtc_desc* d, const void* acts, const int32_t* dead_idx, const int32_t* n_dead_dev,
                       const uint32_t* row_tau_bits, const uint32_t* row_tie_quota, int32_t n_pad, void* acts_aux,
                       void* stream);
/* mtc_aux_select + mtc_aux_dense_fill in ONE pass over the activations (sae.py:332-336: the top-k_aux of the dead
 * columns scattered into a dense acts_aux): a workgroup per batch row holds the row's dead columns in LDS, so the
 * activations are gathered once instead of three times.  Same row_cnt / row_tau_bits / row_tie_quota / acts_aux as the
 * two calls (which it falls back to when n_pad * element size exceeds 40 KB or the rows do not take 8-byte stores). */
int mtc_aux_select_dense(const mtc_desc* d, const void* acts, const int32_t* dead_idx, const int32_t* n_dead_dev,
                         int32_t top_k_aux, int32_t n_pad, uint32_t* row_cnt, uint32_t* row_tau_bits,
                         uint32_t* row_tie_quota, void* acts_aux, void* stream);
int mtc_aux_residual(const mtc_desc* d, const void* x_hat_aux, const void* x_target, const void* x_hat, const void* std,
                     float aux_penalty, void* g, void* g_aux, float* aux_rows,
                     const int32_t* active_dev /* optional device int32: 0 = no dead feature after all (sae.py:331): aux_rows
                     and g_aux are zeroed and g is left alone */,
                     void* stream);
int mtc_aux_mask(int32_t dtype, uint64_t n, const void* acts_aux, void* d_acts_aux, void* stream);

/* a9 over a feature-sharded dictionary ("per-row top-k_aux merge").  Each rank runs mtc_aux_select on its
 * local dead columns, mtc_aux_row_candidates packs the values of that local top-k_aux into cand[B][k_pad] (zero padded)
 * for an all-gather; mtc_aux_select on the gathered [B][world*k_pad] matrix gives the global per-row threshold and tie
 * quota; mtc_aux_count_at counts the local entries above / at that threshold (the at-threshold counts are all-gathered
 * and the quota is granted in (rank, feature) order); mtc_aux_dense_fill then writes the local part of acts_aux. */
int mtc_aux_row_candidates(const mtc_desc* d, const void* acts, const int32_t* dead_idx, const int32_t* n_dead_dev,
                           const uint32_t* row_tau_bits, const uint32_t* row_tie_quota, int32_t k_pad, void* cand,
                           void* stream);
int mtc_aux_count_at(const mtc_desc* d, const void* acts, const int32_t* dead_idx, const int32_t* n_dead_dev,
                     const uint32_t* row_tau_bits, int32_t* row_gt, int32_t* row_eq, void* stream);
/* ... with the ties counted per prefix group (row_eq_groups [B][G], groups = d->prefix in local coordinates): what a
 * feature-sharded caller all-gathers to grant a row's ties in ascending GLOBAL feature order (group, shard, local
 * feature), so that the aux selection does not depend on the number of shards either. */
int mtc_aux_count_at_groups(const mtc_desc* d, const void* acts, const int32_t* dead_idx, const int32_t* n_dead_dev,
                            const uint32_t* row_tau_bits, int32_t* row_gt, int32_t* row_eq_groups, void* stream);

/* a15 helpers: CSR <-> dense [B][D] in the model dtype (encode returns dense, decode takes it) */
int mtc_densify(const mtc_desc* d, const int32_t* row_ptr, const int32_t* col_idx, const void* vals,
                int32_t max_col, void* dense, void* stream);

/* Streaming device copy (16 bytes per lane, non-temporal): the contiguous refill copy of the activation store
 * (transcoder_activation_store.py:157-191 concatenates chunks with torch.cat) and the HBM calibration of bench.py's
 * roofline.  bytes, dst and src multiples of 16. */
int mtc_copy16(void* dst, const void* src, size_t bytes, void* stream);

/* f1  activation-store batch assembly (transcoder_activation_store.py:184-207: DataLoader(TensorDataset(source,
 * target), batch_size, shuffle=True) -> next_batch()).  ONE launch gathers rows perm[0..n_rows) of the source buffer
 * [buffer_rows][A_s] and of the target buffer [buffer_rows][A_t] (model dtype, rows a multiple of 16 bytes) into
 * out_source [n_rows][A_s] / out_target [n_rows][A_t]: pairs travel together.  perm: int64 device indices (one slice
 * of a permutation of the buffer rows per batch).  An index outside [0, buffer_rows) is never dereferenced: that
 * output row is zeroed and *bad_index_count (optional device int32, caller-zeroed) incremented. */
int mtc_gather_row_pairs(int32_t dtype, int32_t n_rows, int32_t A_s, int32_t A_t, int64_t buffer_rows,
                         const void* source_buffer, const void* target_buffer, const int64_t* perm,
                         void* out_source, void* out_target, int32_t* bad_index_count, void* stream);

/* SURVEY 8f row 4 -- top-activating-sample tables (replaces ActivationSampleCollector.collect_batch_samples /
 * _add_sample, src/utils/activation_samples.py:83-183, called from training.py:138-152).
 * Input: the CSC view (col_ptr[D+1], csc_row, csc_pos -> vals in the model dtype) of the sparse activations of
 * n_rows = batch*seq_len token positions, and tokens[batch][seq_len] (int64, ids < 2^31).  Every entry with
 * value > storage_threshold counts into feat_count[D] / feat_max[D]; per feature the max_samples (1..256) largest
 * are kept across calls in top_val[D][M] (descending, -inf = empty) / top_slot[D][M] (a permutation of 0..M-1 per
 * feature; initialise to 0..M-1 and -inf), with per slot slot_meta[D][M][4] = {sequence position, window length,
 * token before the position or -1, flat row} and slot_ctx[D][M][2*(context_size/2)+1] = the token window
 * [pos - context_size/2, pos + context_size/2] clipped to the sequence.  Equal values: the older sample stays.
 * track (optional uint8[D]): features with 0 are skipped (the reference's feature_indices). */
int mtc_collect_top_samples(int32_t dtype, int32_t n_rows, int32_t D, int32_t seq_len, const int32_t* col_ptr,
                            const int32_t* csc_row, const int32_t* csc_pos, const void* vals, const int64_t* tokens,
                            const uint8_t* track, float storage_threshold, int32_t max_samples, int32_t context_size,
                            float* top_val, int32_t* top_slot, int32_t* slot_meta, int32_t* slot_ctx,
                            uint64_t* feat_count, float* feat_max, void* stream);

/* SURVEY 8b / 8e  the exchanges of the feature-sharded step (no counterpart in the single-process reference; its
 * multi-GPU layout is north_star's): thin wrappers over RCCL for hosts that do not use torch.distributed (the Python
 * package does, matryoshka_transcoder_amd/sharded.py: per radix level a 16 KiB histogram all-reduce, the per-row tie-count
 * all-gather, the reduce-scatter over batch rows of the partial reconstructions, the all-gather of the finished rows).
 * One communicator per process and GPU, made on HIP's current device; RCCL is bound at run time (the copy already in
 * the process, else librccl.so), so libmtc does not link against it.  Host side: the unique id is 128 bytes, made by
 * rank 0 and handed to the others by the launcher.  All calls are asynchronous on `stream`; counts are in elements. */
enum { MTC_COMM_F32 = 0, MTC_COMM_BF16 = 1, MTC_COMM_I32 = 2, MTC_COMM_U8 = 3, MTC_COMM_I64 = 4 };
enum { MTC_COMM_SUM = 0, MTC_COMM_MAX = 1, MTC_COMM_MIN = 2 };
#define MTC_COMM_ID_BYTES 128
typedef struct mtc_comm mtc_comm;
int mtc_comm_unique_id(void* id_host_128);
int mtc_comm_init(mtc_comm** out, int32_t world, int32_t rank, const void* id_host_128);
int mtc_comm_destroy(mtc_comm* c);
int mtc_comm_world(const mtc_comm* c);
int mtc_comm_rank(const mtc_comm* c);
int mtc_comm_all_reduce(mtc_comm* c, const void* send, void* recv, size_t count, int32_t dtype, int32_t op, void* stream);
/* recv [recv_count] = this rank's block of the element-wise reduction of every rank's send [world * recv_count] */
int mtc_comm_reduce_scatter(mtc_comm* c, const void* send, void* recv, size_t recv_count, int32_t dtype, int32_t op,
                            void* stream);
/* recv [world * send_count] = every rank's send [send_count], in rank order */
int mtc_comm_all_gather(mtc_comm* c, const void* send, void* recv, size_t send_count, int32_t dtype, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MTC_H */
