/*
 * include/mms.h -- C ABI of libmms_hip.so: the MI355X (gfx950) implementation
 * of the MMS metric-learning inner loop of lxmeng/mms_answer_selection.
 *
 * This is the drop-in boundary.  Each entry point replaces the body of one
 * Caffe virtual of the reference (the file:line it replaces is cited per
 * function; paths relative to the reference checkout).  A Caffe build binds
 * them from Forward_gpu/Backward_gpu with blob->gpu_data()/mutable_gpu_diff()
 * pointers (INTEGRATION.md shows the stub); tests bind them with ctypes.
 *
 * Conventions
 *  - Plain pointers and sizes only.  Every data pointer is DEVICE memory
 *    (hipMalloc / torch.cuda tensor), row-major contiguous, laid out exactly
 *    like the reference's blobs.  No allocation, no host synchronisation, no
 *    stream other than `stream` (a hipStream_t passed as void*; NULL = the
 *    null stream, which is what Caffe uses).  Safe to capture in a hipGraph.
 *  - Return value: MMS_OK or an MMS_ERR_* code; nothing is launched on error.
 *    (The C++ Layer mirror turns a non-zero code into the reference's
 *    CHECK/LOG(FATAL) abort.)
 *  - Deterministic: the same inputs give the same bits every run.  No floating-point atomics anywhere; the only
 *    atomics in the library are the INTEGER arrival words of the fused steps' in-launch loss sum (order-free:
 *    integer addition commutes), which live in the caller's workspace (mms_triplet_workspace_init).
 *  - Numerics: results whose summation order the reference's
 *    own source fixes (Euclidean SimCross forward/backward, PairRankLoss
 *    per-element terms and gradients) are BIT-IDENTICAL to the reference CPU
 *    code.  Results that pass through CBLAS in the reference (cosine,
 *    bilinear, SimMatrix) or a long scalar sum (the loss value) agree to
 *    1e-5 relative.
 *  - dist_mode: 0 cosine, 1 Euclidean 1/(1+||q-a||), 2 bilinear q W a^T
 *    (src/caffe/proto/caffe.proto:471-477; default 1).
 */
#ifndef MMS_H_
#define MMS_H_

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 0.2.0: mms_embed_simcross_forward_f32 gained `embed_bias`, mms_rank_workspace_bytes and the mms_layer_t option
 * table changed (round 2), the triplet workspace carries the arrival words and must be initialised (round 3).
 * A host built against another header must refuse to run: compare mms_version() with MMS_VERSION at start-up. */
#define MMS_VERSION 212

enum {
  MMS_OK = 0,
  MMS_ERR_INVALID_ARG = 1,   /* null pointer, negative size, unknown mode     */
  MMS_ERR_UNSUPPORTED = 2,   /* valid request this build cannot run           */
  MMS_ERR_WORKSPACE = 3,     /* workspace missing or too small                */
  MMS_ERR_LAUNCH = 4         /* hipGetLastError() != hipSuccess after launch  */
};

int mms_version(void);
const char* mms_error_string(int code);

/* ------------------------------------------------------------------------- *
 * Arithmetic of the Euclidean backward term  tt = dT*T*T*T*(q-a)/(T-1+1e-9)
 * (sim_cross_layer.cpp:214-216: float products divided by a DOUBLE, rounded once
 * to float).  The forward value T is bit-identical to the CPU code in both modes.
 *   MMS_EUCLID_BWD_FP32 (default): fp32 throughout, tt = (c*(q-a)) * fl32(1/den);
 *       at most 2 ulp from the reference's value (<= 1.2e-7 relative; the bar is
 *       1e-5).  Used by the kernels specialised for D = 100 / 200 / 300 in the
 *       one-word geometry (SimCross and the fused triplet step) and by the tiled
 *       word-grid backward; every other kernel always runs the reference mode.
 *       For batches of >= 512 narrow word grids (D <= 64) the word-grid backward also
 *       sums da over the rows j in two halves (two waves per pair) and adds the halves:
 *       one association away from the reference's j-ascending sum, within the same bar.
 *   MMS_EUCLID_BWD_REFERENCE: the reference's bits, everywhere.
 * The mode belongs to the CALLING THREAD (a Caffe host drives each GPU from its own thread, and so does its
 * `Caffe` singleton, src/caffe/common.cpp:13-15): mms_set_euclid_backward_mode changes it for the calling
 * thread only and is read when a call is issued.  A thread that never set it uses the process default, the
 * environment variable MMS_EUCLID_BWD=reference|fp32 (fp32 when unset).  The Layer mirror can pin a mode per
 * layer (include/mms_layer.h: mms_layer_set_euclid_backward_mode).
 * ------------------------------------------------------------------------- */
#define MMS_EUCLID_BWD_FP32 0
#define MMS_EUCLID_BWD_REFERENCE 1
int mms_set_euclid_backward_mode(int mode);
int mms_get_euclid_backward_mode(void);

/* Which comparison gates the hinge term in the PairRankLoss BACKWARD (the standalone entry point and the
 * fused triplet step).  The reference's two implementations disagree where margin - y*(a-b) is exactly 0:
 *   MMS_PAIRRANK_HINGE_CPU (default): `ordered > 0`,  PairRankLossLayer::Backward_cpu
 *                                     (src/caffe/layers/pair_rank_loss_layer.cpp:76) -- the parity target;
 *   MMS_PAIRRANK_HINGE_GPU:           `ordered >= 0`, the PairRankLossBackward kernel that Backward_gpu
 *                                     launches (src/caffe/layers/pair_rank_loss_layer.cu:51).
 * Set per calling thread, like the Euclidean mode above. */
/* Measurement aid (no reference counterpart): launches an EMPTY kernel of `workgroups` x 512 threads on
 * `stream`.  bench.py replays it under the same hipGraph protocol as the Layer-API sequence to report what a
 * launch costs by itself on this box (about 2.0 us per graph kernel node on MI355X), i.e. how much of a
 * two-launch step no kernel can remove. */
int mms_null_launch(int workgroups, void* stream);

/* out[0] = sum_i x[i]*y[i], all three on the device (one workgroup, fixed summation tree).  The Layer mirror's
 * Forward uses it for loss tops in GPU mode where the reference calls caffe_gpu_dot
 * (include/caffe/layer.hpp:469-481, src/caffe/util/math_functions.cu caffe_gpu_dot). */
int mms_dot_f32(int n, const float* x, const float* y, float* out, void* stream);

/* SplitLayer::Backward (src/caffe/layers/split_layer.cpp:38-57, split_layer.cu:19-34): the gradient of a blob that
 * feeds `ntop` consumers is the sum of their gradients, formed in consumer order: bottom = top_0 (ntop == 1: a copy),
 * (top_0 + top_1), then + top_2, ... -- the reference's caffe_gpu_add followed by caffe_gpu_axpy(1, ...), same
 * association, hence the same bits.  `top_diffs`: a HOST array of ntop DEVICE pointers; bottom_diff may be top_diffs[0]. */
int mms_split_backward_f32(int count, int ntop, const float* const* top_diffs, float* bottom_diff, void* stream);
int mms_dot_f64(int n, const double* x, const double* y, double* out, void* stream);

#define MMS_PAIRRANK_HINGE_CPU 0
#define MMS_PAIRRANK_HINGE_GPU 1
int mms_set_pairrank_hinge_mode(int mode);
int mms_get_pairrank_hinge_mode(void);

/* How PairRankLoss's loss scalar is summed (mms_pairrank_forward_f32 and the fused triplet step; per calling
 * thread, like the modes above).
 *   MMS_LOSS_SUM_FAST (default): an order-free sum of the bit-exact per-element terms (fixed-shape tree, or integer
 *       arithmetic inside the fused step): within 2e-6 of the exact mean, hence within 1e-5 of the reference's value
 *       wherever that value is itself within 1e-5 of the exact mean.
 *   MMS_LOSS_SUM_REFERENCE: Forward_cpu's own sum, ONE running fp32 accumulator over the terms in index order
 *       (pair_rank_loss_layer.cpp:41-49): the CPU code's bits, including its drift for thousands of near-constant
 *       terms (every add of ~2 to a partial sum in [4096, 8192) rounds the same way: 1-2e-5 relative at N = 4096).
 *       A dependent chain of `count` adds by one lane, ~3 ns each, in one extra one-workgroup launch; the fused
 *       step then sums by a second launch whatever its finish mode. */
#define MMS_LOSS_SUM_FAST 0
#define MMS_LOSS_SUM_REFERENCE 1
int mms_set_loss_sum_mode(int mode);
int mms_get_loss_sum_mode(void);

/* ------------------------------------------------------------------------- *
 * SimCross  (q (N,W1,D), a (N,W2,D) -> top (N, M|1, W1, W2))
 * ------------------------------------------------------------------------- */

/* Replaces SimCrossLayer<float>::Forward_cpu / Forward_gpu
 *   src/caffe/layers/sim_cross_layer.cpp:83-163, sim_cross_layer.cu:128-194.
 * W (M,D,D) and bias (M,W1,W2) are read for dist_mode 2 only (bias may be NULL
 * = bias_term false).  norm0 (N,W1) / norm1 (N,W2) are the layer's
 * data{0,1}_norm_ blobs: written for dist_mode 0 (they hold the NORM, like the
 * CPU code, :118,:126 -- not the squared norm the reference .cu caches),
 * ignored otherwise.  M is mesure_count (dist_mode 2) and must be 1 otherwise.
 * workspace: mms_simcross_workspace_bytes(...) bytes of device scratch. */
int mms_simcross_forward_f32(int dist_mode, int N, int W1, int W2, int D, int M,
                             const float* q, const float* a, const float* W,
                             const float* bias, float* top, float* norm0,
                             float* norm1, void* workspace,
                             size_t workspace_bytes, void* stream);

/* Replaces SimCrossLayer<float>::Backward_cpu / Backward_gpu
 *   src/caffe/layers/sim_cross_layer.cpp:166-307, sim_cross_layer.cu:197-243.
 * Reference-visible behaviour kept: dq and da are ALWAYS overwritten (zeroed
 * first, :176-177); if neither propagate_down flag is set they stay zero, if
 * either is set BOTH are computed (:201).  dist_mode 2: dW (M,D,D) is
 * overwritten (the reference zeroes W.diff itself, :256); dbias (M,W1,W2) is
 * ACCUMULATED into (:301-304) when bias_term != 0. */
int mms_simcross_backward_f32(int dist_mode, int N, int W1, int W2, int D, int M,
                              const float* q, const float* a, const float* W,
                              int bias_term, const float* top,
                              const float* top_diff, const float* norm0,
                              const float* norm1, int propagate_down0,
                              int propagate_down1, float* dq, float* da,
                              float* dW, float* dbias, void* workspace,
                              size_t workspace_bytes, void* stream);

/* The two calls above with their arguments in ONE block (same meaning, same checks, same results): for hosts
 * whose foreign-function calls cost per argument -- ctypes, JNI, cgo -- so that launching kernel by kernel keeps
 * ahead of the device without a hipGraph (a 21-argument ctypes call costs ~4 us of host time, one with a block
 * pointer ~1.3 us; a kernel of this path runs 3-5 us).  The block may be reused and edited between calls. */
typedef struct mms_simcross_args_f32 {
  int dist_mode, N, W1, W2, D, M;
  const float* q; const float* a; const float* W; const float* bias;   /* bias: forward only */
  float* top;                                                          /* forward: written; backward: read */
  float* norm0; float* norm1;
  /* backward only */
  int bias_term, propagate_down0, propagate_down1;
  const float* top_diff; float* dq; float* da; float* dW; float* dbias;
  void* workspace; size_t workspace_bytes;
} mms_simcross_args_f32;
int mms_simcross_forward_block_f32(const mms_simcross_args_f32* args, void* stream);
int mms_simcross_backward_block_f32(const mms_simcross_args_f32* args, void* stream);

/* Forward and Backward in ONE launch, for hosts that hold top_diff BEFORE the forward runs
 * (dist_mode 0/1: q and a are read once, top/dq/da written once; dist_mode 2 runs the two passes
 * back to back on `stream`).  Results are identical to the two calls above.  NOT reachable from
 * the reference's callers: a Caffe Net and `caffe time` alike run every layer's Forward, then every
 * layer's Backward (net.cpp:535-546, 581-591; tools/caffe.cpp:349-361), so a Layer binds the two
 * calls above -- that pair of launches is what bench.py reports; this entry point is a labelled
 * variant there. */
int mms_simcross_forward_backward_f32(int dist_mode, int N, int W1, int W2,
                                      int D, int M, const float* q,
                                      const float* a, const float* W,
                                      const float* bias, const float* top_diff,
                                      float* top, float* norm0, float* norm1,
                                      float* dq, float* da, float* dW,
                                      float* dbias, void* workspace,
                                      size_t workspace_bytes, void* stream);

/* fp16-STORAGE variants of the Euclidean sentence-vector path (W1 = W2 = 1,
 * dist_mode 1; BASELINE cfg 5): q, a, dq, da are IEEE half in HBM, the scores
 * and top_diff stay fp32, and all arithmetic is the fp32 reference arithmetic on
 * the exactly-widened inputs -- top equals the fp32 result on those inputs bit
 * for bit and dq/da are its correctly rounded (RNE) halves.  D % 8 == 0,
 * D <= 2048.  The reference has no fp16 instantiation (common.hpp:41-44). */
/* How the fp16-storage entry points sum the D squares of a pair (per calling thread):
 *   MMS_F16_DISTANCE_ORDERED (default): d-ascending fp32 sum, the order of sim_cross_layer.cpp:100-105 -- scores
 *       bit-identical to the fp32 layer run on the fp16-rounded inputs (and so is any ranking);
 *   MMS_F16_DISTANCE_TREE: a fixed tree sum (deterministic, ~1e-6 relative from the ordered sum; SURVEY 8(d) holds
 *       this configuration to 1e-3 against the fp32 oracle because the reference has no fp16 instantiation to
 *       reproduce).  2.3x faster at D = 1024: the ordered chain is 55 % of the ordered kernel. */
#define MMS_F16_DISTANCE_ORDERED 0
#define MMS_F16_DISTANCE_TREE 1
int mms_set_f16_distance_mode(int mode);
int mms_get_f16_distance_mode(void);
int mms_simcross_euclid_forward_f16(int N, int D, const void* q_f16, const void* a_f16,
                                    float* top, void* stream);
int mms_simcross_euclid_forward_backward_f16(int N, int D, const void* q_f16,
                                             const void* a_f16, const float* top_diff,
                                             float* top, void* dq_f16, void* da_f16,
                                             void* stream);
/* fp16-STORAGE cosine (dist_mode 0), W1 = W2 = 1 (round 3): q, a, dq, da IEEE half in HBM, scores / norms / top_diff fp32,
 * fp32 arithmetic on the exactly-widened inputs in the reference's form (sim_cross_layer.cpp:112-139, 226-250: norms
 * cached, T = q.a / nq / na); within 1e-3 relative of the fp32 layer run on the fp16-rounded inputs (the dot products
 * are cblas_sdot in the reference: no defined order; the gradients are rounded to half).  norm0 / norm1 may be NULL.
 * D % 8 == 0, D <= 2048. */
int mms_simcross_cosine_forward_f16(int N, int D, const void* q_f16, const void* a_f16, float* top, float* norm0,
                                    float* norm1, void* stream);
int mms_simcross_cosine_forward_backward_f16(int N, int D, const void* q_f16, const void* a_f16, const float* top_diff,
                                             float* top, float* norm0, float* norm1, void* dq_f16, void* da_f16,
                                             void* stream);

/* Device scratch needed by the three calls above (0 is possible). */
size_t mms_simcross_workspace_bytes(int dist_mode, int N, int W1, int W2, int D,
                                    int M);

/* ------------------------------------------------------------------------- *
 * SimMatrix  (q (N,K1), a (N,K2), W (K1,K2) -> top (N,1), s_i = q_i^T W a_i)
 * ------------------------------------------------------------------------- */

/* Replaces SimMatrixLayer<float>::Forward_cpu / Forward_gpu
 *   src/caffe/layers/sim_matrix_layer.cpp:53-65, sim_matrix_layer.cu:21-41.
 * qw_scratch (N,K2) receives Q*W.  The reference uses bottom[1]'s diff buffer
 * for it (:58); pass a_blob->mutable_gpu_diff() to reproduce that side effect. */
int mms_simmatrix_forward_f32(int N, int K1, int K2, const float* q,
                              const float* a, const float* W, float* top,
                              float* qw_scratch, void* stream);

/* The same forward for a caller that owns a workspace (mms_simmatrix_workspace_bytes, the backward's; no
 * initialisation needed): with it, and N >= 2048 pairs, Q*W runs on the BF16 matrix pipe at fp32 accuracy -- every
 * fp32 operand is the exact sum of three bf16 values, the six partial products of weight >= 2^-16 are accumulated in
 * fp32 (csrc/bx3_gemm.h; the three dropped are <= 2^-24 relative) -- 16 x the fp32 pipe's rate per instruction for 6 x
 * the instructions.  Results agree with the fp32-MFMA product of mms_simmatrix_forward_f32 to fp32 rounding (both are
 * inside the 1e-5 contract of this BLAS-backed layer: the reference calls cblas_sgemm, no defined order); an input
 * holding an infinity yields NaN where the fp32 product yields inf.  The backward calls below take the same route for
 * dq (and da) when handed this workspace.  mms_set_matrix_mode(1) pins every product to the fp32 pipe (process-wide;
 * 0 = default), mms_get_matrix_mode() reads it. */
int mms_simmatrix_forward_ws_f32(int N, int K1, int K2, const float* q,
                                 const float* a, const float* W, float* top,
                                 float* qw_scratch, void* workspace,
                                 size_t workspace_bytes, void* stream);
int mms_set_matrix_mode(int mode);
int mms_get_matrix_mode(void);

/* fp16-STORAGE SimMatrix scoring (no reference instantiation: common.hpp:41-44; cfg 5's "fp16 embeddings" with a learned
 * metric): q (N,K1) and a (N,K2) IEEE half in HBM, W (K1,K2) and top (N) fp32; top_i = a_i . (q_i W), the forward of
 * sim_matrix_layer.cpp:53-65 without its Q*W output.  Runs on the bf16 matrix pipe whatever the matrix mode: a half is
 * the exact sum of two bf16 values, so the products are formed exactly and accumulated in fp32 -- the fp32 layer's
 * result on the widened inputs to fp32 rounding (1e-5).  K1 % 8 == 0 (any length), K2 % 4 == 0, K2 <= 320, q 16-byte and a
 * 8-byte aligned, else MMS_ERR_UNSUPPORTED (there is no fp32 fallback).  workspace: mms_simmatrix_workspace_bytes. */
int mms_simmatrix_forward_f16(int N, int K1, int K2, const void* q_f16, const void* a_f16,
                              const float* W, float* top, void* workspace,
                              size_t workspace_bytes, void* stream);
/* The TRAINING pair of the same family: the forward also leaves Q*W (fp32, (N,K2)) in qw_scratch; the backward
 * (sim_matrix_layer.cpp:68-95) takes it back and writes dq (N,K1) and da (N,K2) as halves (rounded RNE from the fp32
 * values; either may be NULL = not propagated) and ACCUMULATES dW (K1,K2) in fp32 (NULL = not propagated).  K1 % 8 == 0
 * and K2 % 8 == 0 (both products take a half operand along their inner dimension), K1, K2 <= 320.  Against the fp32
 * layer on the widened inputs: scores and dW at 1e-5, dq / da at half precision (1e-3). */
int mms_simmatrix_forward_train_f16(int N, int K1, int K2, const void* q_f16, const void* a_f16,
                                    const float* W, float* top, float* qw_scratch, void* workspace,
                                    size_t workspace_bytes, void* stream);
int mms_simmatrix_backward_f16(int N, int K1, int K2, const void* q_f16, const void* a_f16,
                               const float* W, const float* qw, const float* top_diff,
                               void* dq_f16, void* da_f16, float* dW, void* workspace,
                               size_t workspace_bytes, void* stream);

/* Replaces SimMatrixLayer<float>::Backward_cpu / Backward_gpu
 *   src/caffe/layers/sim_matrix_layer.cpp:68-95, sim_matrix_layer.cu:43-46.
 * dW (K1,K2) is ACCUMULATED into when param_propagate_down (:73-80);
 * dq / da are overwritten when their propagate_down flag is set (:81-93) and
 * left untouched otherwise. */
int mms_simmatrix_backward_f32(int N, int K1, int K2, const float* q,
                               const float* a, const float* W,
                               const float* top_diff, int param_propagate_down,
                               int propagate_down0, int propagate_down1,
                               float* dq, float* da, float* dW, void* workspace,
                               size_t workspace_bytes, void* stream);

/* Same call for a caller that kept the forward's product: qw (N,K2) is what
 * mms_simmatrix_forward_f32 wrote to qw_scratch for the SAME q and W, unchanged
 * since.  da_j = dT_j * (W^T q_j) (:88) is row j of that product times dT_j, so
 * the call scales it instead of recomputing it -- same bits as the call above,
 * one GEMM fewer.  qw may be the da buffer itself (in place), which is where the
 * reference's forward leaves it (bottom[1]'s diff, :58). */
int mms_simmatrix_backward_cached_f32(int N, int K1, int K2, const float* q,
                                      const float* a, const float* W,
                                      const float* qw, const float* top_diff,
                                      int param_propagate_down,
                                      int propagate_down0, int propagate_down1,
                                      float* dq, float* da, float* dW,
                                      void* workspace, size_t workspace_bytes,
                                      void* stream);

size_t mms_simmatrix_workspace_bytes(int N, int K1, int K2);

/* ------------------------------------------------------------------------- *
 * PairRankLoss  (a, b, y of `count` = N*C elements -> scalar loss)
 * ------------------------------------------------------------------------- */

/* Replaces PairRankLossLayer<float>::Forward_cpu / Forward_gpu
 *   src/caffe/layers/pair_rank_loss_layer.cpp:26-52, pair_rank_loss_layer.cu:10-43.
 * ordered / similar are the layer's cached ordered_diff_ / similar_diff_ blobs
 * (count floats each, outputs).  loss is ONE device float. */
int mms_pairrank_forward_f32(int count, float margin, const float* a,
                             const float* b, const float* y, float* ordered,
                             float* similar, float* loss, void* workspace,
                             size_t workspace_bytes, void* stream);

/* Replaces PairRankLossLayer<float>::Backward_cpu / Backward_gpu
 *   src/caffe/layers/pair_rank_loss_layer.cpp:55-84, pair_rank_loss_layer.cu:45-83.
 * top_diff = top[0]->cpu_diff()[0] (the loss weight).  Uses the CPU code's
 * strict `ordered > 0` (:76), not the .cu's `>=`.  da / db are overwritten
 * when their flag is set.  (propagate_down[2] is a LOG(FATAL) in the
 * reference and is rejected one level up, in the Layer mirror.) */
int mms_pairrank_backward_f32(int count, float top_diff, const float* y,
                              const float* ordered, const float* similar,
                              int propagate_down0, int propagate_down1,
                              float* da, float* db, void* stream);

size_t mms_pairrank_workspace_bytes(int count);

/* ------------------------------------------------------------------------- *
 * Fused training step of the metric-learning inner loop (one launch, see below):
 *   s_pos = SimCross_euclid(q, a_pos), s_neg = SimCross_euclid(q, a_neg)   (N,1,1,1)
 *   loss  = PairRankLoss(s_pos, s_neg, y)                 (margin, loss_weight)
 *   backward through PairRankLoss and both SimCross layers:
 *   dq = dq(from pos) + dq(from neg)  (Caffe's Split layer sum), da_pos, da_neg.
 * Equivalent net: two SimCross layers sharing bottom q (sim_cross_layer.cpp)
 * feeding PairRankLoss (pair_rank_loss_layer.cpp), geometry W1 = W2 = 1.
 * Outputs equal the layer-by-layer result bitwise, except the loss scalar: it is within 2e-6 of the exact mean of
 * the (bit-exact) per-triplet terms, hence within 1e-5 of the reference's value wherever the reference's fp32
 * running sum is itself that close to the exact mean (for thousands of near-constant terms it drifts by 1-2e-5:
 * DESIGN.md 5); bit-identical to the reference's in MMS_LOSS_SUM_REFERENCE mode (above).
 * `loss` may be NULL: the scalar (a display value: no gradient depends on it) is then not reduced at all -- 7.6
 * instead of 9.6 us of kernel time at 4096 x 300.
 * workspace: mms_triplet_workspace_bytes(N) bytes of 8-byte-aligned device memory that belongs to this call
 * sequence alone: its head holds the arrival words of the in-launch loss sum (zero between launches), the rest one
 * term per triplet.  Call mms_triplet_workspace_init ONCE after allocating it (it zeroes the words, asynchronously
 * on `stream`) -- and again if a launch that used it died mid-way; a workspace sized for a larger N serves any
 * smaller one.  Launches that may be in flight together -- two streams, two captured graphs replayed
 * concurrently -- need a workspace each; nothing about the words is chosen by host state at call or capture time.
 * ------------------------------------------------------------------------- */
/* How the loss scalar of the fused step is summed (per calling thread):
 *   MMS_TRIPLET_FINISH_INLAUNCH (default): the step is ONE launch.  The per-triplet terms are added as integers in
 *       units of 2^-S (S = 42 - ceil(log2 N): 2^-30 at N = 4096), so the sum does not depend on the order in which
 *       workgroups finish, and the arrival count travels in the same 64-bit word as the sum (waves -> workgroup
 *       word in LDS -> one word per 16 workgroups -> top word; the wave that completes the top word writes the
 *       loss).  10.3 us per 4096 x 300 step HBM-cold against 11.2 for the second launch.  Domain: every term
 *       max(0, margin - y*(s_pos - s_neg)) + |(1 - y)*(s_pos - s_neg)| must be below 2^10 -- any margin and labels
 *       below a few hundred.  A term outside it makes the loss NaN (never a wrong number); scores and gradients do
 *       not depend on the mode.  Batches of more than 131072 triplets, and widths other than 100 / 200 / 300, use
 *       the second launch whatever the mode.
 *   MMS_TRIPLET_FINISH_LAUNCH: a second, one-workgroup launch adds the N per-triplet terms in a fixed order (no
 *       limit on the terms). */
#define MMS_TRIPLET_FINISH_LAUNCH 0
#define MMS_TRIPLET_FINISH_INLAUNCH 1
int mms_set_triplet_finish_mode(int mode);
int mms_get_triplet_finish_mode(void);
int mms_triplet_euclid_step_f32(int N, int D, float margin, float loss_weight,
                                const float* q, const float* a_pos,
                                const float* a_neg, const float* y,
                                float* s_pos, float* s_neg, float* loss,
                                float* dq, float* da_pos, float* da_neg,
                                void* workspace, size_t workspace_bytes,
                                void* stream);

size_t mms_triplet_workspace_bytes(int N);
int mms_triplet_workspace_init(void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------- *
 * Fused training step of the LEARNED metric (cfg 3's arithmetic):
 *   s_pos_i = q_i^T W a_pos_i, s_neg_i = q_i^T W a_neg_i          SimMatrix x 2, W shared (sim_matrix_layer.cpp:53-65)
 *   loss = PairRankLoss(s_pos, s_neg, y)                            (pair_rank_loss_layer.cpp:26-52)
 *   backward through PairRankLoss (:55-84) and both SimMatrix layers (sim_matrix_layer.cpp:68-95):
 *   dq = dq(pos) + dq(neg) (Net::Init's Split sum), da_pos, da_neg, dW += both branches' q_i a_i^T terms (param
 *   diffs accumulate, as in the layers).
 * Equivalent net: two SimMatrix layers with bottom q in common and `param { name }` in common, feeding PairRankLoss.
 * Three products instead of the layers' six: Q W once for both branches -- its epilogue forms both scores, the hinge
 * term and its gradients g+, g- per row, and writes da_pos = g+ (W^T q), da_neg = g- (W^T q) and B = g+ a_pos + g- a_neg;
 * then dq = B W^T and dW += Q^T B.  Neither Q W nor the (N, 1) score gradients reach HBM.  Scores, loss terms and
 * hinge decisions follow the reference's operation order given the scores; everything that passes through a product
 * agrees with the layer-by-layer result to 1e-5 (the reference's own products go through CBLAS).
 * Shapes the panel kernel does not serve (K2 > 304, sizes not multiples of 4, small N) run the layers one by one
 * inside the call: same results, no fusion.  `loss` may be NULL.  dW is accumulated into (zero it like
 * Net::ClearParamDiffs does).  workspace: mms_triplet_simmatrix_workspace_bytes(N, K1, K2), no initialisation needed.
 * ------------------------------------------------------------------------- */
size_t mms_triplet_simmatrix_workspace_bytes(int N, int K1, int K2);
int mms_triplet_simmatrix_step_f32(int N, int K1, int K2, float margin, float loss_weight,
                                   const float* q, const float* a_pos, const float* a_neg, const float* y,
                                   const float* W, float* s_pos, float* s_neg, float* loss,
                                   float* dq, float* da_pos, float* da_neg, float* dW,
                                   void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------- *
 * Ranking metrics over the scores (forward only).  These
 * define "ranking output": with the Euclidean scores bit-identical to the CPU
 * code, MAP/MRR/AUC computed here are bit-identical too wherever the sort order
 * is defined (ties between EQUAL scores are implementation-defined in the
 * reference's unstable std::sort; here they keep original order -- which is also what
 * libstdc++'s std::sort does for a bucket of at most 16 items, its stable insertion sort:
 * MAP / MRR over candidate groups that small match a libstdc++ build of the reference bit
 * for bit even when tied scores carry different labels).
 * ------------------------------------------------------------------------- */

/* Order of EQUAL scores that carry different labels (MAP / MRR / AUC; per calling thread).
 *   MMS_RANK_TIES_INPUT_ORDER (default): ties keep the input order -- also what libstdc++ does in buckets of at
 *       most 16 items.
 *   MMS_RANK_TIES_LIBSTDCXX: the order a libstdc++ build of the reference leaves them in -- buckets of more than
 *       16 items that contain such a tie are re-sorted by a step-by-step restatement of libstdc++'s std::sort
 *       (csrc/libstdcxx_sort.h: introsort with median-of-three partitioning, heap-sort fallback, final insertion
 *       sort; checked against the real std::sort by tests/test_libstdcxx_sort.py) on the bucket's items in their
 *       original order.  Sequential, one lane per bucket, hence opt-in (tools/rank_ties_probe.py, heavily tied
 *       scores): MAP + MRR over 1,517 candidates in groups of ~22: 147 instead of 91 us; AUC, whose one bucket is
 *       the whole input: 4.4 ms instead of 60 us at 1,517 items, 117 ms at 20,000.  Without such a tie the mode
 *       costs its detection pass only. */
#define MMS_RANK_TIES_INPUT_ORDER 0
#define MMS_RANK_TIES_LIBSTDCXX 1
int mms_set_rank_tie_mode(int mode);
int mms_get_rank_tie_mode(void);

/* Replaces MAPLayer<float>::Forward_cpu (src/caffe/layers/map_layer.cpp:41-100) and
 * MRRLayer<float>::Forward_cpu (src/caffe/layers/mrr_layer.cpp:38-79).
 * prob (n, fixed_axis+1): the score of item i is prob[i*(fixed_axis+1)+fixed_axis];
 * label (n) in {0,1}; group (n): items are bucketed by int(group[i]).
 * map_out / mrr_out / effective_out (device scalars) may each be NULL. */
int mms_rank_map_mrr_f32(int n, int fixed_axis, const float* prob, const float* label,
                         const float* group, float* map_out, float* mrr_out,
                         int* effective_out, void* workspace, size_t workspace_bytes,
                         void* stream);

/* Replaces AUCLayer<float>::Forward_cpu (src/caffe/layers/auc_layer.cpp:47-136) for
 * inner_num = 1: prob (n, dim), score = prob[i*dim + fixed_axis]. */
int mms_rank_auc_f32(int n, int dim, int fixed_axis, const float* prob, const float* label,
                     int has_ignore_label, int ignore_label, float* auc_out,
                     void* workspace, size_t workspace_bytes, void* stream);

/* The same layer for any label axis: prob (outer, channels, inner) -- bottom[0] with `axis` as its label axis,
 * outer = count(0, axis), inner = count(axis + 1) -- and label (outer, inner); item (o, j) scores
 * prob[o*channels*inner + fixed_axis*inner + j] (auc_layer.cpp:66-77).  Items whose label equals ignore_label are
 * skipped when has_ignore_label (:69-71).  Workspace: mms_rank_workspace_bytes(outer * inner). */
int mms_rank_auc_nd_f32(int outer, int channels, int inner, int fixed_axis, const float* prob,
                        const float* label, int has_ignore_label, int ignore_label, float* auc_out,
                        void* workspace, size_t workspace_bytes, void* stream);

/* Replaces RankAccuracyLayer<float>::Forward_cpu
 * (src/caffe/layers/rank_accuracy_layer.cpp:36-50): mean of [label*(a-b) > 0]. */
int mms_rank_accuracy_f32(int count, const float* a, const float* b, const float* label,
                          float* acc_out, void* workspace, size_t workspace_bytes,
                          void* stream);

size_t mms_rank_workspace_bytes(int n);

/* ------------------------------------------------------------------------- *
 * Embed -- the layer that produces SimCross's bottoms.
 * index (M) holds word ids as floats (Caffe feeds them as Dtype), weight (K,N),
 * bias (N) or NULL, top (M,N).
 * ------------------------------------------------------------------------- */

/* Replaces EmbedLayer<float>::Forward_cpu / Forward_gpu
 *   src/caffe/layers/embed_layer.cpp:135-152, embed_layer.cu:42-62. */
int mms_embed_forward_f32(int M, int N, int K, const float* index, const float* weight,
                          const float* bias, float* top, void* stream);

/* Replaces EmbedLayer<float>::Backward_cpu / Backward_gpu
 *   src/caffe/layers/embed_layer.cpp:155-180, embed_layer.cu:64-88.
 * weight_diff (K,N) is ACCUMULATED into in the CPU code's n-ascending order
 * (bit-identical; the reference's .cu uses atomicAdd).  bias_diff (N) likewise
 * accumulated (BLAS-ordered in the reference: 1e-5).  Either may be NULL
 * (= param_propagate_down false). */
int mms_embed_backward_f32(int M, int N, int K, const float* index, const float* top_diff,
                           float* weight_diff, float* bias_diff, void* workspace,
                           size_t workspace_bytes, void* stream);
/* Two Embed layers that share ONE table (and bias) -- network_v4's w2v_q / w2v_a, do_trec_qa_clean.py:452-467 --
 * back-propagated in one pass: exactly mms_embed_backward_f32(layer 0) followed by mms_embed_backward_f32(layer 1) into
 * the same weight_diff / bias_diff (pass the layer whose Backward the net runs first -- the LATER layer of the file --
 * as layer 0), with the same bits in weight_diff (a table row's additions keep the order "layer 0's rows ascending,
 * then layer 1's"), but the inverted index is built once over M0 + M1 indices, every touched table row is read and
 * written once, and the bias gradient is one column sum.  workspace: mms_embed_workspace_bytes(M0 + M1, N). */
int mms_embed_backward_pair_f32(int M0, int M1, int N, int K, const float* index0, const float* top_diff0,
                                const float* index1, const float* top_diff1, float* weight_diff, float* bias_diff,
                                void* workspace, size_t workspace_bytes, void* stream);
/* The same pair with the inverted index built in the FORWARD pass: it depends on the word ids alone, the one-workgroup
 * sort that builds it takes ~14 us, and beside the forward's gathers it costs nothing.
 *   mms_embed_forward_pair_f32: top0 = Embed(index0), top1 = Embed(index1) (embed_layer.cpp:135-152 twice) in ONE
 *       launch; with index_workspace != NULL (mms_embed_workspace_bytes(M0 + M1, N)) and
 *       mms_embed_pair_index_supported(M0, M1, K) it also writes the index of the ids in the order (index0, index1).
 *   mms_embed_backward_pair_indexed_f32: mms_embed_backward_pair_f32 minus the index build; index0 / index1 and the
 *       workspace must be the ones the forward call was given, in the same order, and the workspace untouched since.
 *       (For the reference's nets pass the LATER layer of the file as layer 0 in BOTH calls: Net::Backward reaches it first.) */
int mms_embed_pair_index_supported(int M0, int M1, int K);
int mms_embed_forward_pair_f32(int M0, int M1, int N, int K, const float* index0, const float* index1,
                               const float* weight, const float* bias, float* top0, float* top1, void* index_workspace,
                               size_t index_workspace_bytes, void* stream);
int mms_embed_backward_pair_indexed_f32(int M0, int M1, int N, int K, const float* index0, const float* top_diff0,
                                        const float* index1, const float* top_diff1, float* weight_diff, float* bias_diff,
                                        void* index_workspace, size_t index_workspace_bytes, void* stream);

size_t mms_embed_workspace_bytes(int M, int N);

/* Embed fused into SimCross's loads (forward / scoring; dist_mode 0 or 1):
 *   top == SimCross(Embed(index_q), Embed(index_a))
 * i.e. embed_layer.cpp:135-152 followed by sim_cross_layer.cpp:96-139, without the (N,W,D) blobs in
 * between.  index_q (N,W1) and index_a (N,W2) hold word ids as floats, clamped into [0,K) like
 * mms_embed_forward_f32; weight (K,D) is the table BOTH Embed layers read (the driver shares it by
 * parameter name, do_trec_qa_clean.py:462-467); embed_bias (D) is their shared bias blob or NULL
 * (bias_term false) -- the driver's Embed layers DO carry one (`bias_term` is left at its default,
 * the `#bias_term=False` of :462 is commented out), so a row is bias[d] + weight[id][d], one
 * rounding, as the layer's gemm gives it (:146-151); top (N,1,W1,W2); norm0 (N,W1) / norm1 (N,W2)
 * receive the row norms in dist_mode 0 (NULL otherwise).  Results are the bits of the two separate
 * calls (Euclid: the reference's CPU bits).  dist_mode 2 has its own entry point below (it needs
 * W and bias). */
int mms_embed_simcross_forward_f32(int dist_mode, int N, int W1, int W2, int D, int K,
                                   const float* index_q, const float* index_a,
                                   const float* weight, const float* embed_bias, float* top,
                                   float* norm0, float* norm1, void* stream);

/* The same fusion for dist_mode 2, the mode network_v4 scores with (do_trec_qa_clean.py:468):
 *   top (N,M,W1,W2) == SimCross_bilinear(Embed(index_q), Embed(index_a); W (M,D,D), bias (M,W1,W2) or NULL)
 * embed_layer.cpp:135-152 (embed_bias as above) followed by sim_cross_layer.cpp:140-161, ONE launch: the word-grid
 * forward kernels gather their q / a images from the table themselves.  Same kernels and operand values as
 * mms_embed_forward_f32 x2 + mms_simcross_forward_f32, hence the same bits (1e-5 vs the reference: BLAS order).
 * Covers the word-grid geometries of the fused forward (W1, W2 <= 48, D <= 64, and either N >= 512 or
 * N <= 256 with N*M <= 65535); anything else returns MMS_ERR_UNSUPPORTED -- run the two layers separately. */
int mms_embed_simcross_bilinear_forward_f32(int N, int W1, int W2, int D, int M, int K,
                                            const float* index_q, const float* index_a,
                                            const float* weight, const float* embed_bias,
                                            const float* W, const float* bias, float* top, void* stream);

/* ------------------------------------------------------------------------- *
 * Batch feed: dst[i,:] = src[perm[first+i],:], i < rows,
 * for a dataset (src_rows, row_elems) resident in HBM; perm (src_rows ints on
 * the device) or NULL for the identity; first + rows <= src_rows.
 * Replaces the per-row caffe_copy loop of HDF5DataLayer<float>::Forward_cpu/_gpu
 *   src/caffe/layers/hdf5_data_layer.cpp:124-151, hdf5_data_layer.cu:19-51.
 * ------------------------------------------------------------------------- */
int mms_feed_gather_rows_f32(int rows, int row_elems, int src_rows, const float* src, const int* perm,
                             int first, float* dst, void* stream);

/* ------------------------------------------------------------------------- *
 * double instantiation.  The reference instantiates every layer for float and
 * double (INSTANTIATE_CLASS, include/caffe/common.hpp:41-44); these are what a
 * Layer<double>::Forward_gpu / Backward_gpu binds.  Same argument meaning and
 * error behaviour as the _f32 functions above (no norm / workspace differences
 * except that SimMatrix backward and PairRankLoss need no workspace).  The
 * kernels are functional rather than tuned (csrc/f64_paths.hip); Euclidean
 * results, PairRankLoss (loss included) and dbias are bit-identical to the CPU
 * code, BLAS-backed results agree to ~1e-12.
 * ------------------------------------------------------------------------- */
size_t mms_simcross_workspace_bytes_f64(int dist_mode, int N, int W1, int W2, int D, int M);
int mms_simcross_forward_f64(int dist_mode, int N, int W1, int W2, int D, int M, const double* q,
                             const double* a, const double* W, const double* bias, double* top,
                             double* norm0, double* norm1, void* workspace, size_t workspace_bytes,
                             void* stream);
int mms_simcross_backward_f64(int dist_mode, int N, int W1, int W2, int D, int M, const double* q,
                              const double* a, const double* W, int bias_term, const double* top,
                              const double* top_diff, const double* norm0, const double* norm1,
                              int propagate_down0, int propagate_down1, double* dq, double* da,
                              double* dW, double* dbias, void* workspace, size_t workspace_bytes,
                              void* stream);
int mms_simmatrix_forward_f64(int N, int K1, int K2, const double* q, const double* a,
                              const double* W, double* top, double* qw_scratch, void* stream);
int mms_simmatrix_backward_f64(int N, int K1, int K2, const double* q, const double* a,
                               const double* W, const double* top_diff, int param_propagate_down,
                               int propagate_down0, int propagate_down1, double* dq, double* da,
                               double* dW, void* stream);
int mms_pairrank_forward_f64(int count, double margin, const double* a, const double* b,
                             const double* y, double* ordered, double* similar, double* loss,
                             void* stream);
int mms_pairrank_backward_f64(int count, double top_diff, const double* y, const double* ordered,
                              const double* similar, int propagate_down0, int propagate_down1,
                              double* da, double* db, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MMS_H_ */
