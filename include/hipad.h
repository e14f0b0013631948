/*
 * include/hipad.h -- C ABI of libhipad.so, the MI355X (gfx950) implementation of HiP-AD's
 * hot path.  Plain pointers and sizes only (no torch / ATen types); every pointer is a
 * DEVICE pointer unless stated otherwise; `stream` is a hipStream_t passed as void*.
 *
 * Each entry point names the reference interface it replaces
 * (reference = nullmax-vision/HiP-AD, paths relative to projects/mmdet3d_plugin/).
 *
 * Conventions
 *   - return value: 0 = HIPAD_OK, otherwise a negative HIPAD_E* code (the reference's
 *     launchers return void and never look at cudaGetLastError(); we do);
 *   - all launches are asynchronous on `stream`; nothing here allocates, frees or
 *     synchronises, so every call can be captured into a hipGraph;
 *   - buffers are owned by the caller and must be contiguous in the stated layout.
 */
#ifndef HIPAD_H_
#define HIPAD_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HIPAD_OK 0
#define HIPAD_EINVAL (-1)      /* bad dimension / null pointer / unsupported combination */
#define HIPAD_EWORKSPACE (-2)  /* workspace too small */
#define HIPAD_ELAUNCH (-3)     /* hipGetLastError() reported a launch failure */
#define HIPAD_ERANGE (-4)      /* sizes overflow the kernel's 32-bit index budget */

typedef void *hipad_stream_t;

/* ABI version (bumped on any signature change or added entry point; 3 = this header) and a static description string. */
int hipad_abi_version(void);
const char *hipad_status_string(int status);

/* ------------------------------------------------------------------------------------
 * deformable_aggregation forward.
 * Replaces:  void deformable_aggregation(float* output, const float* mc_ms_feat,
 *              const int* spatial_shape, const int* scale_start_index,
 *              const float* sample_location, const float* weights, int batch_size,
 *              int num_cams, int num_feat, int num_embeds, int num_scale, int num_anchors,
 *              int num_pts, int num_groups)
 *            ops/src/deformable_aggregation_cuda.cu:265-288 (kernel :129-187), called from
 *            ops/src/deformable_aggregation.cpp:31-62.
 * Same arguments in the same order and meaning, plus workspace and stream.
 *   feat   [bs, num_feat, C] f32 (C = num_embeds)
 *   spatial_shape [cams, scales, 2] i32 (h, w);  scale_start_index [cams, scales] i32
 *   loc    [bs, A, P, cams, 2] f32 (x, y normalised to the image)
 *   weights[bs, A, P, cams, scales, G] f32
 *   out    [bs, A, C] f32 -- OVERWRITTEN (the reference accumulates with atomicAdd into a
 *          tensor its shim zero-fills; the result is the same, no pre-zeroing needed here)
 * Deterministic: no atomics, fixed summation order.
 * workspace: hipad_daf_forward_workspace() bytes (may be 0 -> workspace may be NULL).
 * ---------------------------------------------------------------------------------- */
size_t hipad_daf_forward_workspace(int batch_size, int num_cams, int num_feat, int num_embeds,
                                   int num_scale, int num_anchors, int num_pts, int num_groups);

int hipad_daf_forward(float *out, const float *feat, const int32_t *spatial_shape,
                      const int32_t *scale_start_index, const float *loc, const float *weights,
                      int batch_size, int num_cams, int num_feat, int num_embeds, int num_scale,
                      int num_anchors, int num_pts, int num_groups, void *workspace,
                      size_t workspace_bytes, hipad_stream_t stream);

/* ------------------------------------------------------------------------------------
 * deformable_aggregation backward.
 * Replaces:  void deformable_aggregation_grad(const float* mc_ms_feat, ..., const float*
 *              grad_output, float* grad_mc_ms_feat, float* grad_sampling_location,
 *              float* grad_weights, int batch_size, ... int num_groups)
 *            ops/src/deformable_aggregation_cuda.cu:291-318 (kernel :190-262), called from
 *            ops/src/deformable_aggregation.cpp:86-124.
 *   grad_out  [bs, A, C];  grad_feat like feat;  grad_loc like loc;  grad_w like weights.
 * flags:
 *   0 (reference semantics)     all three gradients are ACCUMULATED into caller buffers
 *                               (the reference's caller zero-fills them,
 *                               ops/deformable_aggregation.py:55-57);
 *   HIPAD_DAF_OVERWRITE_LOC_W   grad_loc and grad_w are fully written by the kernel (zeros
 *                               for dropped samples) -- no memset needed; grad_feat is
 *                               still accumulated, so one buffer can collect all call sites.
 * grad_feat uses fp32 atomics (true scatter); grad_loc / grad_w are reduced inside the
 * wavefront and stored once (deterministic).  Any of grad_feat / grad_loc / grad_w may be
 * NULL to skip that gradient.
 * ---------------------------------------------------------------------------------- */
#define HIPAD_DAF_OVERWRITE_LOC_W 1
/*   HIPAD_DAF_ATOMIC_FEAT       use the one-pass kernel that scatters grad_feat with fp32
 *                               atomics (needs no workspace).  Default is the sorted path:
 *                               bilinear taps are counting-sorted by pyramid row and each row
 *                               is accumulated in registers from L2-resident grad_out rows, so
 *                               only rows straddling a 64-tap batch use atomics (see
 *                               hip-ad_amd/csrc/daf_bwd_sorted.hip).  The sorted path needs
 *                               hipad_daf_backward_workspace() bytes; that function returns 0
 *                               for shapes it does not cover (then the atomic kernel runs). */
#define HIPAD_DAF_ATOMIC_FEAT 2

size_t hipad_daf_backward_workspace(int batch_size, int num_cams, int num_feat, int num_embeds,
                                    int num_scale, int num_anchors, int num_pts, int num_groups);

int hipad_daf_backward(const float *feat, const int32_t *spatial_shape,
                       const int32_t *scale_start_index, const float *loc, const float *weights,
                       const float *grad_out, float *grad_feat, float *grad_loc, float *grad_w,
                       int batch_size, int num_cams, int num_feat, int num_embeds, int num_scale,
                       int num_anchors, int num_pts, int num_groups, int flags, void *workspace,
                       size_t workspace_bytes, hipad_stream_t stream);

/* bf16 feature rows (ours; no reference counterpart: the reference widens the fp16 pyramid to fp32 before the op,
 * models/sparse_detector.py:84-89).  Same contracts as hipad_daf_forward / hipad_daf_backward with `feat_bf16`
 * [batch_size][num_feat][256] bfloat16 -- the image encoder's own output dtype, so the values are the same and the
 * results are bit-identical to the fp32 entries on the widened tensor, at half the gather bytes.  grad_feat stays fp32.
 * Only the 256-channel kernels (num_embeds == 256, num_groups == 8 for the backward, sorted feature gradient);
 * anything else returns HIPAD_EINVAL. */
int hipad_daf_forward_bf16(float *out, const void *feat_bf16, const int32_t *spatial_shape,
                           const int32_t *scale_start_index, const float *loc, const float *weights,
                           int batch_size, int num_cams, int num_feat, int num_embeds, int num_scale,
                           int num_anchors, int num_pts, int num_groups, void *workspace, size_t workspace_bytes,
                           hipad_stream_t stream);
int hipad_daf_backward_bf16(const void *feat_bf16, const int32_t *spatial_shape,
                            const int32_t *scale_start_index, const float *loc, const float *weights,
                            const float *grad_out, float *grad_feat, float *grad_loc, float *grad_w,
                            int batch_size, int num_cams, int num_feat, int num_embeds, int num_scale,
                            int num_anchors, int num_pts, int num_groups, int flags, void *workspace,
                            size_t workspace_bytes, hipad_stream_t stream);

/* ------------------------------------------------------------------------------------
 * Feature gradient of SEVERAL aggregation calls in one pass (ours; the reference runs
 * deformable_aggregation_grad once per call site, ops/deformable_aggregation.py:41-75, each time zero-filling and
 * scattering into a fresh copy of the pyramid that autograd then adds up: 24 call sites per decoder forward,
 * models/sparse_onedecoder.py:867-887).
 * The calls share the pyramid geometry (batch_size, cams, num_feat, levels, 256 channels, 8 groups) and therefore the
 * feature-gradient buffer; each brings its own sampling locations, weights and output gradient:
 *   calls[k].loc      [bs, A_k, P_k, cams, 2] f32      calls[k].weights [bs, A_k, P_k, cams, L, 8] f32
 *   calls[k].grad_out [bs, A_k, 256] f32
 * grad_feat [bs, num_feat, 256] f32 is ACCUMULATED into, exactly as by `ncalls` calls of hipad_daf_backward with
 * grad_loc = grad_w = NULL (same taps, same products; the summation order inside a row differs, as it does between any
 * two runs of the reference's atomics).  One counting sort of the bilinear taps of ALL calls by pyramid row, one
 * accumulation pass: a row touched by several calls is read-modified-written once.
 * `calls` is a HOST array (ncalls <= HIPAD_DAF_MAX_CALLS); it is copied into the kernel arguments, the tensors it
 * points to must stay valid until the launches have run.  workspace: hipad_daf_backward_feat_multi_workspace() bytes.
 * ---------------------------------------------------------------------------------- */
#define HIPAD_DAF_MAX_CALLS 64
typedef struct hipad_daf_call {
  const float *loc;
  const float *weights;
  const float *grad_out;
  int32_t num_anchors;
  int32_t num_pts;
} hipad_daf_call;

/* Tuning knob of the counting sort (like hipad_daf_set_pairs_per_wave): chunks of 1024 (anchor, point) indices one
 * workgroup of the tap passes walks; 0 = automatic (about one workgroup per CU). */
void hipad_daf_set_tap_chunks(int chunks);
/* Tuning knob of the accumulation pass: consecutive batches of 64 sorted taps one wave walks with its row sum carried
 * along (rows inside a run are written without atomics); 1..64, anything else = the default (4). */
void hipad_daf_set_feat_run(int batches);
/* ... and the most workgroups of that pass's persistent grid; 64..65536, anything else = the default (8192). */
void hipad_daf_set_feat_blocks(int blocks);

size_t hipad_daf_backward_feat_multi_workspace(const hipad_daf_call *calls, int ncalls, int batch_size, int num_cams,
                                               int num_feat, int num_embeds, int num_scale, int num_groups);
int hipad_daf_backward_feat_multi(const hipad_daf_call *calls, int ncalls, float *grad_feat,
                                  const int32_t *spatial_shape, const int32_t *scale_start_index, int batch_size,
                                  int num_cams, int num_feat, int num_embeds, int num_scale, int num_groups,
                                  void *workspace, size_t workspace_bytes, hipad_stream_t stream);

/* ------------------------------------------------------------------------------------
 * Index work of the operator, exposed for bit-exact checks (no reference counterpart as a
 * function: it is the integer part of deformable_aggregation_cuda.cu:160-181 + :18-52).
 *   valid [bs*A*P*cams] u8;  taps [bs*A*P*cams*scales*4] i32 = (h_low, w_low, mask, row)
 *   mask bit k = corner k in bounds (k: 0 = (lo,lo), 1 = (lo,hi_w), 2 = (hi_h,lo), 3 = (hi,hi));
 *   row = b*num_feat + scale_start_index[cam, scale]; dropped samples write zeros.
 * ---------------------------------------------------------------------------------- */
int hipad_daf_taps(uint8_t *valid, int32_t *taps, const int32_t *spatial_shape,
                   const int32_t *scale_start_index, const float *loc, int batch_size,
                   int num_cams, int num_feat, int num_scale, int num_anchors, int num_pts,
                   hipad_stream_t stream);

/* ------------------------------------------------------------------------------------
 * 3D -> 2D projection of the key points, written in the aggregation op's location layout.
 * Replaces: DeformableFeatureAggregation.project_points (models/blocks.py:216-225) + the
 *           permute/reshape of models/blocks.py:144-145.
 *   key_points [bs, A, P, 3] f32;  projection_mat [bs, cams, 4, 4] f32 (row-major);
 *   image_wh [bs, cams, 2] f32 or NULL;  loc [bs, A, P, cams, 2] f32 (overwritten).
 * Arithmetic: ((m0*x + m1*y) + m2*z) + m3 per row without fma, clamp(z, min=1e-5), IEEE
 * divisions -- bit-exact with the reference's fp32 result on the committed fixtures.
 * backward: grad_key_points [bs, A, P, 3] (overwritten) from grad_loc; the projection
 * matrices are data and receive no gradient (as in the reference's use).
 * ---------------------------------------------------------------------------------- */
int hipad_project_points_forward(float *loc, const float *key_points, const float *projection_mat,
                                 const float *image_wh, int batch_size, int num_anchors, int num_pts,
                                 int num_cams, hipad_stream_t stream);
int hipad_project_points_backward(float *grad_key_points, const float *grad_loc,
                                  const float *key_points, const float *projection_mat,
                                  const float *image_wh, int batch_size, int num_anchors, int num_pts,
                                  int num_cams, hipad_stream_t stream);

/* ------------------------------------------------------------------------------------
 * Sampling-weight softmax fused with the re-layout.
 * Replaces: the tail of DeformableFeatureAggregation._get_weights (models/blocks.py:196-214:
 *           reshape -> softmax(dim=-2) -> reshape -> keep-mask) and the permute + contiguous of
 *           models/blocks.py:147-158.
 *   u [bs, A, n], v [bs, cams, n] f32 with n = L*P*G ordered ((l*P + p)*G + g): the logits are
 *     u[b,a,:] + v[b,cam,:] (the Linear applied to the anchor part and to the camera part);
 *   keep [bs, A, cams, P] f32 (0 or 1/(1-p_drop)) or NULL;
 *   weights [bs, A, P, cams, L, G] f32 (overwritten) -- the aggregation op's layout;
 *   stats [bs, A, G, 2] f32 (max, sum of exp) kept for the backward.  G must divide 256.
 *   u_per_cam != 0: u is [bs, A, cams, n] (the module without camera embedding,
 *     models/blocks.py:116-118) and v may be NULL.
 * backward: grad_u [bs, A, n] (overwritten), grad_v [bs, cams, n] (zeroed, then accumulated).
 * ---------------------------------------------------------------------------------- */
int hipad_weights_softmax_forward(float *weights, float *stats, const float *u, const float *v,
                                  const float *keep, int batch_size, int num_anchors, int num_cams,
                                  int num_scale, int num_pts, int num_groups, int u_per_cam,
                                  hipad_stream_t stream);
/* backward scratch (bytes): the per-(anchor, camera) logit gradients before they are summed over the anchors
 * into grad_v; 0 when there is no camera part (has_v == 0) */
/* Tuning knob: workgroups per (sample, anchor) of the two kernels (each repeats the reduction, writes its slice);
 * 1..16, anything else = automatic (2 when fewer than ~200 anchors of >= 8192 logits exist, else 1). */
void hipad_weights_softmax_set_split(int workgroups_per_anchor);
size_t hipad_weights_softmax_backward_workspace(int batch_size, int num_anchors, int num_cams, int num_scale,
                                                int num_pts, int num_groups, int has_v);
int hipad_weights_softmax_backward(float *grad_u, float *grad_v, const float *grad_weights,
                                   const float *stats, const float *u, const float *v,
                                   const float *keep, int batch_size, int num_anchors, int num_cams,
                                   int num_scale, int num_pts, int num_groups, int u_per_cam,
                                   void *workspace, size_t workspace_bytes, hipad_stream_t stream);

/* ------------------------------------------------------------------------------------
 * Multi-head attention core: out = dropout(softmax(q k^T * softmax_scale)) v per (batch, head).
 * Replaces: flash_attn_unpadded_kvpacked_func as called by FlashAttention.forward
 *           (models/attention.py:76-80, 91-95; flash-attn==2.7.0.post2 is a CUDA package that is
 *           neither vendored in the reference nor usable here), fixed-length case.
 *   q [B, Nq, H*D], k, v [B, Nk, H*D], out [B, Nq, H*D] f32 (row-major, heads contiguous in the
 *   last dim); D in {32, 64, 128}.  Operands are rounded to bf16 on load, fp32 accumulation on the
 *   matrix cores, fp32 softmax (the reference runs this block in fp16/bf16, attention.py:63).
 *   lse [B, H, Nq] f32: base-2 log-sum-exp of the scaled scores, needed by the backward (may be
 *   NULL for inference).  p_drop / seed: dropout on the probabilities (counter-based, the
 *   backward regenerates the same mask from the same seed).  seed_dev (device pointer or NULL): a
 *   step counter added to `seed` at run time, so a captured hipGraph draws a fresh mask per replay.
 * backward: dq [B,Nq,H*D], dk, dv [B,Nk,H*D] overwritten; delta_ws: B*H*Nq floats of scratch.
 * ---------------------------------------------------------------------------------- */
int hipad_attention_forward(float *out, float *lse, const float *q, const float *k, const float *v,
                            int batch, int heads, int num_q, int num_k, int head_dim,
                            float softmax_scale, float p_drop, unsigned seed, const unsigned *seed_dev,
                            hipad_stream_t stream);
int hipad_attention_backward(float *dq, float *dk, float *dv, float *delta_ws, const float *dout,
                             const float *out, const float *lse, const float *q, const float *k,
                             const float *v, int batch, int heads, int num_q, int num_k, int head_dim,
                             float softmax_scale, float p_drop, unsigned seed, const unsigned *seed_dev,
                            hipad_stream_t stream);

/* ------------------------------------------------------------------------------------
 * Linear layer on the matrix cores.
 * Replaces: the torch / mmcv ``Linear`` calls of the decoder blocks (reference
 *           models/blocks.py:32-42, 104-118, 367-396; models/attention.py:27-34, 127-143; the
 *           refinement heads and anchor encoders), i.e. cuBLAS GEMM + bias + ReLU + autograd
 *           accumulate kernels.
 *   forward : y[M,N] = relu?(x[M,K] weight[N,K]^T + bias[N])        (bias may be NULL)
 *   backward: dx[M,K] = g weight ;  dw[N,K] += g^T x ;  db[N] += colsum(g)
 *             with g = dy where y_relu > 0 (y_relu = the forward output of a ReLU layer, or NULL
 *             for no activation).  dx is overwritten; dw / db are ACCUMULATED with fp32 atomics
 *             (pass the parameter's gradient buffer: no separate accumulation pass); any of
 *             dx / (dw, db) may be NULL.
 * fp32 in memory, operands rounded to bf16 on load, fp32 accumulation (v_mfma_f32_16x16x32_bf16).
 * ---------------------------------------------------------------------------------- */
int hipad_linear_forward(float *y, const float *x, const float *weight, const float *bias, int M, int N,
                         int K, int relu, hipad_stream_t stream);
/* Linear + ReLU + LayerNorm forward in one launch (the [Linear, ReLU, LayerNorm] unit of the reference's
 * linear_relu_ln stacks, models/blocks.py:32-42): x_relu[M,N] = relu(x weight^T + bias) and y = LayerNorm(x_relu)
 * with mean / rstd [M] for the backward, which is hipad_layernorm_backward followed by hipad_linear_backward
 * (y_relu = x_relu).  Supported when hipad_linear_relu_ln_supported(N, K) and x / weight are 16-byte aligned. */
int hipad_linear_relu_ln_supported(int N, int K);
int hipad_linear_relu_ln_forward(float *y, float *x_relu, float *mean, float *rstd, const float *x,
                                 const float *weight, const float *bias, const float *gamma, const float *beta,
                                 int M, int N, int K, float eps, hipad_stream_t stream);
int hipad_linear_backward(float *dx, float *dw, float *db, const float *dy, const float *y_relu,
                          const float *x, const float *weight, int M, int N, int K, hipad_stream_t stream);

/* ------------------------------------------------------------------------------------
 * Key points of 3D-box queries, generated and projected into every camera in one launch
 * (hip-ad_amd/csrc/keypoints.hip).
 * Replaces: SparseBox3DKeyPointsGenerator.forward (reference models/det/blocks.py:183-224) chained with
 *           DeformableFeatureAggregation.project_points + permute (models/blocks.py:216-225, 144-145).
 *   anchor [bs, A, anchor_dim >= 8] = [x,y,z, log w,l,h, sin,cos, ...]; fix_scale [n_fix, 3]; learn [bs, A, n_learn*3]
 *   (the learnable_fc output BEFORE the sigmoid; NULL when n_learn == 0); projection_mat [bs, cams, 4, 4];
 *   image_wh [bs, cams, 2] or NULL -> loc [bs, A, n_fix + n_learn, cams, 2] (the aggregation op's layout);
 *   key_points [bs, A, P, 3] is written too when not NULL.
 *   backward: grad_anchor [bs, A, anchor_dim] OVERWRITTEN (zero-filled here, columns 0..7 accumulated), grad_learn
 *   overwritten.
 * ---------------------------------------------------------------------------------- */
int hipad_box_points_project_forward(float *loc, float *key_points, const float *anchor, const float *fix_scale,
                                     const float *learn, const float *projection_mat, const float *image_wh,
                                     int batch_size, int num_anchors, int n_fix, int n_learn, int num_cams,
                                     int anchor_dim, hipad_stream_t stream);
int hipad_box_points_project_backward(float *grad_anchor, float *grad_learn, const float *grad_loc,
                                      const float *anchor, const float *fix_scale, const float *learn,
                                      const float *projection_mat, const float *image_wh, int batch_size,
                                      int num_anchors, int n_fix, int n_learn, int num_cams, int anchor_dim,
                                      hipad_stream_t stream);

/* Same for poly-line queries (map elements, plan trajectories): SparsePoint3DKeyPointsGenerator.forward (reference
 * models/map/blocks.py:172-225) + projection.  anchor [bs, A, num_sample*2]; offset [bs, A, num_sample*num_heights*
 * num_learnable*2] (the learnable_fc output); heights [num_heights] (ground_height + fix_height) ->
 * loc [bs, A, num_sample*num_heights*num_learnable, cams, 2].  backward overwrites grad_anchor and grad_offset. */
int hipad_line_points_project_forward(float *loc, const float *anchor, const float *offset, const float *heights,
                                      const float *projection_mat, const float *image_wh, int batch_size,
                                      int num_anchors, int num_sample, int num_heights, int num_learnable, int num_cams,
                                      hipad_stream_t stream);
int hipad_line_points_project_backward(float *grad_anchor, float *grad_offset, const float *grad_loc, const float *anchor,
                                       const float *offset, const float *heights, const float *projection_mat,
                                       const float *image_wh, int batch_size, int num_anchors, int num_sample,
                                       int num_heights, int num_learnable, int num_cams, hipad_stream_t stream);

/* ------------------------------------------------------------------------------------
 * LayerNorm over the last dimension (hip-ad_amd/csrc/layernorm.hip).
 * Replaces: nn.LayerNorm in the decoder -- linear_relu_ln stacks (reference models/blocks.py:32-42), the
 *           "norm" ops of the decoder program (projects/configs/hipad_b2d_stage2.py:293), AsymmetricFFN's
 *           pre-norm (blocks.py:352-353).
 *   forward : y[M,N] = (x - mean) * rstd * gamma + beta ; mean / rstd [M] are written for the backward
 *             (either may be NULL); gamma / beta may be NULL (= 1 / 0).  N % 4 == 0, N <= 1024.
 *   backward: dx[M,N] overwritten (may be NULL); dgamma[N] / dbeta[N] ACCUMULATED with fp32 atomics (pass
 *             the parameters' gradient buffers; either may be NULL).
 * fp32, rstd = 1/sqrt(biased variance + eps): torch.nn.functional.layer_norm's arithmetic.
 * ---------------------------------------------------------------------------------- */
int hipad_layernorm_forward(float *y, float *mean, float *rstd, const float *x, const float *gamma,
                            const float *beta, int M, int N, float eps, hipad_stream_t stream);
int hipad_layernorm_backward(float *dx, float *dgamma, float *dbeta, const float *dy, const float *x,
                             const float *mean, const float *rstd, const float *gamma, int M, int N,
                             hipad_stream_t stream);

/* ------------------------------------------------------------------------------------
 * Minimum-cost assignment on the device (hip-ad_amd/csrc/assign.hip).
 * Replaces: scipy.optimize.linear_sum_assignment in the reference's target assignment
 *           (models/det/target.py:98-103, models/map/target.py:150-155): cost matrix copied to the host and
 *           solved there, once per sample, decoder layer and task.
 *   cost [batch, rows, cols] f32, rows = ground-truth items, cols = predictions (the reference's matrix
 *   transposed); n_rows [batch] int32: only the first n_rows[b] rows of problem b take part.
 *   col_of_row [batch, rows] int32 (out): the prediction assigned to each row, -1 for rows >= n_rows[b].
 *   rows <= 256, cols <= 2048, rows <= cols.  fp64 arithmetic on the fp32 costs (what SciPy does); the
 *   assignment equals SciPy's whenever the optimum is unique.  Costs must be finite (the callers map
 *   nan / -inf to 1e8 like the reference does before calling SciPy).
 * ---------------------------------------------------------------------------------- */
int hipad_linear_assignment(int *col_of_row, const float *cost, const int *n_rows, int batch, int rows,
                            int cols, hipad_stream_t stream);

/* ------------------------------------------------------------------------------------
 * Sigmoid focal loss, value and logit gradient in one pass (hip-ad_amd/csrc/losses.hip).
 * Replaces: mmdet==2.28.2 FocalLoss(use_sigmoid=True) as called by the reference's loss()
 *           (models/sparse_onedecoder.py:1146, 1201, 1302, 1360-1362).
 *   logits [rows, num_classes] f32; target [rows] int64 in [0, num_classes] (num_classes = background);
 *   weight [rows] or NULL; rows are layer-major, `layers` equal groups; avg_factor [layers] or NULL.
 *   loss_per_layer [layers] (out) = sum over the group of loss * weight / (avg_factor[l] + eps), or the group mean
 *   when avg_factor is NULL (mmdet's weight_reduce_loss, reduction 'mean'); grad_logits [rows, num_classes] (out)
 *   = d loss_per_layer[l(row)] / d logits.
 * ---------------------------------------------------------------------------------- */
int hipad_focal_loss_forward(float *loss_per_layer, float *grad_logits, const float *logits, const long long *target,
                             const float *weight, const float *avg_factor, long long rows, int num_classes,
                             int layers, float alpha, float gamma, hipad_stream_t stream);

/* ------------------------------------------------------------------------------------
 * The decoder's training objective in a handful of launches (hip-ad_amd/csrc/lossprog.hip).
 * Replaces: SparseOneDecoder.loss and its per-task pieces (models/sparse_onedecoder.py:1094-1579): samplers
 *           det/target.py:66-162, map/target.py:38-62 + 105-160 (+ map/match_cost.py:8-106), motion/target.py:5-35 + 71-99,
 *           plan/target.py:7-37 + 80-162; loss modules det/losses.py:11-93, map/loss.py:10-120 and mmdet 2.28.2's
 *           FocalLoss / L1Loss / CrossEntropyLoss(use_sigmoid) / GaussianFocalLoss / FocalLossCost.
 * All decoder layers at once: a prediction tensor is given as a table of per-layer DEVICE pointers (hipad_layer_ptrs,
 * entry l = the head's output of layer l, [bs, rows, width] contiguous f32); ground truth is padded to G items per
 * sample with a per-sample count (int32).  Every "grad_*" output is [layers, bs, rows, width] f32 and receives
 * d(sum of the task's loss terms) / d(prediction) -- every element is written; `terms` outputs are [n_terms][layers] f32
 * and are ACCUMULATED into (caller zero-fills once per step).  Loss term = sum over rows / (max(num_pos[l], 1) + eps)
 * or the plain mean, times the loss weight, exactly as mmdet's weight_reduce_loss does it.
 * The *_assign entries run cost matrix -> hipad_linear_assignment -> inverse map:
 *   matched [layers*bs, P] i32 = ground-truth item of a prediction or -1;
 *   count_out [2][layers] f32: [0] matched predictions whose target row is not all zero (the task's num_pos before the
 *   cross-rank mean), [1] all assigned items (the motion head's num_pos).
 * Scratch (cost, perm, n_rows, index) is caller-owned: cost [layers*bs, G, P] f32, perm same shape u8, n_rows
 * [layers*bs] i32, index [layers*bs, G] i32 (prediction of each ground-truth item, -1 for padding).
 * ---------------------------------------------------------------------------------- */
#define HIPAD_LOSS_MAX_LAYERS 8
#define HIPAD_LOSS_MAX_CLSWISE 4
#define HIPAD_LOSS_MAX_GROUPS 16
#define HIPAD_LOSS_MAX_INTERVALS 4
#define HIPAD_LOSS_MAX_BUCKETS 8

typedef struct hipad_layer_ptrs {
  const float *p[HIPAD_LOSS_MAX_LAYERS];
} hipad_layer_ptrs;

typedef struct hipad_det_loss_cfg {
  /* SparseBox3DTarget (det/target.py:66-162): cost = cls_weight * focal cost + box_weight * sum |box - target| * w * reg_w */
  float cost_alpha, cost_gamma, cost_eps, cost_cls_weight, cost_box_weight;
  float cost_reg_weights[10];
  int num_cls_wise;                                   /* class-wise replacement of the NaN-aware weight row */
  int cls_wise_label[HIPAD_LOSS_MAX_CLSWISE];
  float cls_wise_weights[HIPAD_LOSS_MAX_CLSWISE][10];
  /* losses: FocalLoss (w_cls) on all rows; L1 (w_box, per-dimension loss_reg_weights), centerness BCE (w_cns) and
   * yawness Gaussian focal (w_yns, gauss_alpha) on matched rows whose best class score passes cls_threshold (<= 0: off) */
  float focal_alpha, focal_gamma, w_cls, w_box, w_cns, w_yns, gauss_alpha, cls_threshold;
  float loss_reg_weights[10];
  int cns_index, yns_index;                           /* columns of the quality head */
} hipad_det_loss_cfg;

typedef struct hipad_map_loss_cfg {
  /* roi normalisation (map/target.py:64-76): (v - origin) / norm per coordinate */
  float origin_x, origin_y, norm_x, norm_y;
  /* MapQueriesCost: FocalLossCost (alpha 0.25, gamma 2) * cost_cls_weight + LinesL1Cost(beta, permute) * cost_reg_weight */
  float cost_cls_weight, cost_reg_weight, cost_beta;
  float focal_alpha, focal_gamma, w_cls, w_line, loss_beta, cls_threshold;
  float reg_weights[40];
} hipad_map_loss_cfg;

typedef struct hipad_motion_loss_cfg {
  float focal_alpha, focal_gamma, w_cls, w_reg;
} hipad_motion_loss_cfg;

typedef struct hipad_plan_loss_cfg {
  /* anchor group g: kind 0 = temp, 1 = spat (aligned to the reference group's winning mode), 2 = speed bucket;
   * gt_traj[g] [bs, T, 2], gt_mask[g] [bs, T] = the group's ground truth (speed groups: their interval's) */
  int kind[HIPAD_LOSS_MAX_GROUPS];
  const float *gt_traj[HIPAD_LOSS_MAX_GROUPS];
  const float *gt_mask[HIPAD_LOSS_MAX_GROUPS];
  int ref_group;
  int num_intervals;                                  /* speed intervals; interval i = interval_size[i] bucket groups */
  int interval_size[HIPAD_LOSS_MAX_INTERVALS];
  int interval_group[HIPAD_LOSS_MAX_INTERVALS][HIPAD_LOSS_MAX_BUCKETS];
  float bucket_lo[HIPAD_LOSS_MAX_INTERVALS][HIPAD_LOSS_MAX_BUCKETS], bucket_hi[HIPAD_LOSS_MAX_INTERVALS][HIPAD_LOSS_MAX_BUCKETS];
  const float *speed_traj, *speed_mask;               /* trajectory the ground-truth average speed is taken from */
  float speed_interval;                               /* seconds between its way-points */
  float focal_alpha, focal_gamma, w_cls, w_reg;
  const float *ego_status, *ego_status_mask;          /* [bs, S] (S = 0: no ego-status term) */
  float w_status;
} hipad_plan_loss_cfg;

int hipad_loss_det_assign(float *cost, int *n_rows, int *index, int *matched, float *count_out, const hipad_layer_ptrs *cls,
                          const hipad_layer_ptrs *box, const float *gt_boxes, const long long *labels, const int *count,
                          const hipad_det_loss_cfg *cfg, int layers, int bs, int P, int C, int D, int G, int gt_dim,
                          hipad_stream_t stream);
/* terms [4][layers]: class, box, centerness, yawness.  quality / grad_quality may be NULL with Q = 0.
 * grad_box_cns [layers, bs, P, 3]: the centerness term's gradient into the box centre (its target exp(-|centre error|)
 * is not detached in the reference, det/losses.py:55-60), kept apart from grad_box so that hipad_loss_scale can weight
 * the two terms separately. */
int hipad_loss_det(float *terms, float *grad_cls, float *grad_box, float *grad_box_cns, float *grad_quality,
                   const hipad_layer_ptrs *cls, const hipad_layer_ptrs *box, const hipad_layer_ptrs *quality, const int *matched, const float *num_pos,
                   const float *gt_boxes, const long long *labels, const hipad_det_loss_cfg *cfg, int layers, int bs, int P,
                   int C, int D, int Q, int G, int gt_dim, hipad_stream_t stream);
/* gt_pts [bs, G, num_permute, 40]; order [layers*bs, G] i32 (out) = point order of each matched ground-truth line */
int hipad_loss_map_assign(float *cost, unsigned char *perm, int *n_rows, int *index, int *matched, int *order,
                          float *count_out, const hipad_layer_ptrs *cls, const hipad_layer_ptrs *pts, const float *gt_pts,
                          const long long *labels, const int *count, const hipad_map_loss_cfg *cfg, int layers, int bs, int P,
                          int C, int pts_dim, int G, int num_permute, hipad_stream_t stream);
/* terms [2][layers]: class, line */
int hipad_loss_map(float *terms, float *grad_cls, float *grad_pts, const hipad_layer_ptrs *cls, const hipad_layer_ptrs *pts,
                   const int *matched, const int *order, const float *num_pos, const float *gt_pts, const long long *labels,
                   const hipad_map_loss_cfg *cfg, int layers, int bs, int P, int C, int pts_dim, int G, int num_permute,
                   hipad_stream_t stream);
/* cls [bs, A, M], reg [bs, A, M, T, 2] per layer; det_matched = the box head's `matched` (its LAST layer serves every
 * layer, sparse_onedecoder.py:1287); num_pos[l * num_pos_stride]; trajs [bs, G, T, 2], masks [bs, G, T].
 * terms [2][layers]: class, regression */
int hipad_loss_motion(float *terms, float *grad_cls, float *grad_reg, const hipad_layer_ptrs *cls, const hipad_layer_ptrs *reg,
                      const int *det_matched, const float *num_pos, int num_pos_stride, const float *trajs, const float *masks,
                      const hipad_motion_loss_cfg *cfg, int layers, int bs, int A, int M, int T, int G, hipad_stream_t stream);
/* cls [bs, num_groups * M], reg [bs, num_groups * M, T, 2] (way-point OFFSETS), status [bs, S] per layer (single driving
 * command).  terms [7][layers]: temp cls, temp reg, spat cls, spat reg, speed cls, speed reg, ego status */
int hipad_loss_plan(float *terms, float *grad_cls, float *grad_reg, float *grad_status, const hipad_layer_ptrs *cls,
                    const hipad_layer_ptrs *reg, const hipad_layer_ptrs *status, const hipad_plan_loss_cfg *cfg, int layers,
                    int bs, int num_groups, int M, int T, int S, hipad_stream_t stream);

/* Backward of the objective: out[segment] = grads[segment] * g_terms[term of the element's column] (one launch for all
 * segments; out may alias grads).  A segment is a [count / width, width] gradient tensor at `offset` floats into `grads`; term_table (device,
 * int8) maps column -> term at table_offset; extra_offset >= 0 adds grads[extra_offset + row * extra_cols + col] *
 * g_terms[extra_term] to the first extra_cols columns.  `segments` is a HOST array. */
#define HIPAD_LOSS_MAX_SEGMENTS 16
typedef struct hipad_loss_segment {
  long long offset, count, extra_offset;
  int width, table_offset, extra_cols, extra_term;
} hipad_loss_segment;
int hipad_loss_scale(float *out, const float *grads, const float *g_terms, const signed char *term_table,
                     const hipad_loss_segment *segments, int num_segments, hipad_stream_t stream);

/* ------------------------------------------------------------------------------------
 * Gradient clipping + AdamW over flat buffers (hip-ad_amd/csrc/optim.hip).
 * Replaces: the per-tensor optimiser step of the reference's training loop: mmcv OptimizerHook
 *           grad_clip (max_norm 25) + torch.optim.AdamW with the backbone at lr x0.5
 *           (projects/configs/hipad_b2d_stage2.py:629-641).
 *   param / grad / exp_avg / exp_avg_sq: n floats each, 16-byte aligned; elements [0, n_group0) step
 *   with lr0, the rest with lr1.  max_norm <= 0 disables clipping.  step_dev: device int32 holding the
 *   number of steps taken so far (incremented here).  norm_out_dev (may be NULL): TWO floats, [0] receives the
 *   total gradient norm BEFORE clipping, [1] the learning rate group 0 used.  zero_grad != 0 clears grad after
 *   it has been consumed.  workspace: hipad_adamw_workspace() bytes of device scratch.
 *   sched (may be NULL = constant): the reference's lr_config (mmcv LrUpdaterHook, by_epoch=False;
 *   projects/configs/hipad_b2d_stage2.py:643-649: CosineAnnealing, linear warm-up 500 iterations from 1/3,
 *   min_lr_ratio 1e-3).  The factor is evaluated IN the kernel from *step_dev, so a hipGraph-replayed step follows
 *   the schedule without any host update; hipad_lr_factor() is the same closed form on the host.
 *   shadow_bf16 (may be NULL): n bf16 values, 8-byte aligned; receives the updated parameters rounded to bf16
 *   (the operand format of the MFMA kernels, element order unchanged); hipad_shadow_bf16 fills it initially.
 * Arithmetic is torch.nn.utils.clip_grad_norm_ followed by torch.optim.AdamW (decoupled weight decay,
 * bias-corrected moments), fp32.
 * ---------------------------------------------------------------------------------- */
typedef struct hipad_lr_schedule {
  int policy;          /* 0 constant, 1 CosineAnnealing */
  int warmup_iters;    /* 0 = no warm-up; linear warm-up otherwise */
  float warmup_ratio;
  int max_iters;
  float min_lr_ratio;
} hipad_lr_schedule;
size_t hipad_adamw_workspace(void);
float hipad_lr_factor(const hipad_lr_schedule *sched, int iteration);
int hipad_shadow_bf16(unsigned short *dst, const float *src, long long n, hipad_stream_t stream);

/* hipad_accumulate_bf16: dst[k] += src[k] for up to HIPAD_ACC_MAX tensors in one launch.  Replaces: the per-layer
 *   `weight.grad.add_(bf16 weight gradient)` of the image encoder's convolutions under bf16 autocast (torch: the cast back
 *   to fp32 in ToCopyBackward + AccumulateGrad, one or two launches per convolution, ~60 convolutions per frame).
 *   dst: fp32, contiguous, sizes[0..3]; src: bf16 with ELEMENT strides[0..3] (channels-last weight gradients are read in
 *   place).  items: HOST array (the table is passed in the kernel arguments: nothing to upload, capturable). */
#define HIPAD_ACC_MAX 64
typedef struct hipad_acc_item {
  float *dst;
  const unsigned short *src;
  int32_t sizes[4];
  int32_t strides[4];
} hipad_acc_item;
int hipad_accumulate_bf16(const hipad_acc_item *items, int n_items, hipad_stream_t stream);
int hipad_adamw_step(float *param, float *grad, float *exp_avg, float *exp_avg_sq, long long n,
                     long long n_group0, float lr0, float lr1, float beta1, float beta2, float eps,
                     float weight_decay, float max_norm, int *step_dev, float *norm_out_dev,
                     void *workspace, size_t workspace_bytes, int zero_grad, const hipad_lr_schedule *sched,
                     unsigned short *shadow_bf16, hipad_stream_t stream);

/* ------------------------------------------------------------------------------------
 * MLP chains (hip-ad_amd/csrc/chain.hip; weight gradients: chain_dw_kernel in gemm.hip).
 * Replaces: the reference's linear_relu_ln stacks and heads -- ([Linear, ReLU] x in_loops, LayerNorm) x
 *           out_loops, Linear (+ mmcv Scale) (+ residual) -- models/blocks.py:32-42 used by the refinement
 *           heads (det/blocks.py:77-156, map/blocks.py:80-135, plan/blocks.py:16-157, motion/blocks.py:16-50,
 *           ego/blocks.py:14-75), the anchor encoders (det/blocks.py:22-74, map/blocks.py:18-42) and the small
 *           camera / command / target-point encoders: one launch per direction for a whole GROUP of chains
 *           instead of one per Linear / LayerNorm.
 * A chain: x = x0[r * ldx0 + c] (+ x1[r * ldx1 + c]); for every layer l: h = x W_l^T + b_l, ReLU if flags & 1,
 *   then x = LayerNorm(h) * gamma + beta if flags & 2 else x = h; out[r * ldo + c] = x * out_scale[c] +
 *   residual[r * ldr + c] for the last layer.  1 <= K, N <= 256; layer l + 1 has K = N of layer l.
 *   w: the bf16 weights W [N][K] in MFMA-fragment order P(W) (see hipad_pack_weights, which makes them), 16-byte
 *   aligned; bias / gamma / beta fp32 or NULL.
 * Training: `save` (fp32 scratch owned by the caller) receives, at the given float offsets, h (M x N: post-ReLU,
 *   pre-LayerNorm activation), y (M x N: LayerNorm output, only for flags & 2) and stats (M x 2: mean, rstd);
 *   xsum (M x K0, may be NULL) receives x0 + x1 when x1 is given.  With save == NULL nothing is kept (inference).
 * hipad_chain_backward_dx: reverse sweep.  dout (M rows, ldo) is the gradient of `out`; for every layer the gated
 *   gradient of its pre-activation is written to dy + off_dy (M x N) for hipad_chain_backward_dw, gamma / beta /
 *   out_scale gradients are ADDED atomically into dgamma / dbeta / dscale, and the input gradient is written to
 *   dx (M rows, lddx; NULL skips it).  wt: P(W^T), the transposed weights in fragment order.
 * hipad_chain_backward_dw: for every entry dw[N][K] += dy^T x, db[N] += column sums of dy (atomics; x: M rows, ldx).
 * ---------------------------------------------------------------------------------- */
#define HIPAD_CHAIN_MAX_LAYERS 6
#define HIPAD_CHAIN_MAX_CHAINS 8
#define HIPAD_CHAIN_MAX_DW 48
#define HIPAD_CHAIN_NONE 0xffffffffu
typedef struct hipad_chain_layer {
  const unsigned short *w;
  const float *bias, *gamma, *beta;
  unsigned off_h, off_y, off_stats;
  int K, N, flags;
  float eps;
} hipad_chain_layer;
typedef struct hipad_chain {
  const float *x0, *x1;
  float *xsum, *out;
  const float *out_scale, *residual;
  float *save;
  int ldx0, ldx1, ldo, ldr, M, nlayers;
  hipad_chain_layer layers[HIPAD_CHAIN_MAX_LAYERS];
} hipad_chain;
typedef struct hipad_chain_grad_layer {
  const unsigned short *wt;
  const float *gamma;
  float *dgamma, *dbeta;
  unsigned off_h, off_stats, off_dy;
  int K, N, flags;
  float eps;
} hipad_chain_grad_layer;
typedef struct hipad_chain_grad {
  const float *dout, *out_scale;
  float *dscale, *dx;
  const float *save;
  float *dy;
  int ldo, lddx, M, nlayers;
  hipad_chain_grad_layer layers[HIPAD_CHAIN_MAX_LAYERS];
} hipad_chain_grad;
typedef struct hipad_chain_dw {
  const float *dy, *x;
  float *dw, *db;
  int M, N, K, ldx;
} hipad_chain_dw;
int hipad_chain_forward(const hipad_chain *chains, int nchains, hipad_stream_t stream);
/* Diagnostic aid: when set (>= 256 x 2 u64 of device memory), thread 0 of workgroup 0 of every forward launch writes
 * (s_memtime, s_memrealtime) pairs at its phase boundaries there; NULL (the default) turns it off. */
void hipad_chain_debug_stamps(unsigned long long *device_buffer);
int hipad_chain_backward_dx(const hipad_chain_grad *chains, int nchains, hipad_stream_t stream);
int hipad_chain_backward_dw(const hipad_chain_dw *entries, int nentries, hipad_stream_t stream);
/* The chains' operand copies of a batch of fp32 matrices src[i] = [rows][cols] in one launch, in MFMA-fragment order:
 *   P(A [r][d]): blocks of 16 rows x 32 depth, block (tr, s) at element ((tr * ceil(d / 32) + s) * 512); inside a block
 *   lane = 16 * quad + l15 (0..63) owns the 8 consecutive elements A[16 tr + l15][32 s + 8 quad + 0..7]; elements outside
 *   A are zero.  Size: ceil(r / 16) * ceil(d / 32) * 512 bf16.
 * dst[i] = P(src[i]) (the chains' `w`), dst_t[i] = P(src[i]^T) (their `wt`); either table may be NULL.  All tables live
 * in DEVICE memory: n_mats pointers / dimensions and tile_start[n_mats + 1] = exclusive prefix sum of
 * ceil(rows / 32) * ceil(cols / 32); total_tiles = its last entry. */
int hipad_pack_weights(unsigned short *const *dst, unsigned short *const *dst_t, const float *const *src,
                       const int *rows, const int *cols, const int *tile_start, int n_mats, int total_tiles,
                       hipad_stream_t stream);

/* ------------------------------------------------------------------------------------
 * Decoder glue (hip-ad_amd/csrc/glue.hip).
 * hipad_chunk_mix.  Replaces: the query mixing of the planning refinement head (reference
 *   models/plan/blocks.py:120-135 -- "aligned" = sum of the temp / spat anchor groups' queries, each speed query =
 *   aligned + the speed groups of its interval; ~12 slice / add / cat launches forward, ~25 backward).
 *   x0, x1 (x1 may be NULL): (bs, in_chunks * rows, channels); out: (bs, out_chunks * rows, channels);
 *   out chunk g = sum_k weights[g * in_chunks + k] * (x0 chunk k + x1 chunk k); weights on the HOST (<= 16 x 16).
 *   The backward is the same call with the transposed table on the output gradient.  channels % 4 == 0.
 * hipad_motion_query_embed.  Replaces: get_motion_anchor (reference models/sparse_onedecoder.py:428-444: anchors of the
 *   arg-max class rotated by the box yaw) + the last way-point's gen_sineembed_for_position
 *   (models/attention.py:292-306), ~25 elementwise launches per decoder layer; no gradient flows through it.
 *   cls (n_anchor, num_classes) logits; box (n_anchor, box_dim); table (num_classes, modes, steps, 2);
 *   freq (half_dim) = 10000 ** (2 * (k / 2) / half_dim); out (n_anchor, modes, 2 * half_dim) = [embed(y) | embed(x)],
 *   embed(v)[k] = sin or cos (k odd) of v * 2 pi / freq[k], same operation order as the torch expression.
 * hipad_step_offsets.  Replaces: the planning branch's way-points -> per-step offsets (reference
 *   models/sparse_onedecoder.py, plan refinement: cat(wp[:1], wp[1:] - wp[:-1]) along the time axis; slice / subtract /
 *   concatenate launches forward, their slice-backward fills and joins backward).  x, out: (rows, steps, dims) fp32,
 *   out != x; out[t] = x[t] - x[t-1], out[0] = x[0]; adjoint != 0 applies the transposed map (a gradient's backward):
 *   out[t] = x[t] - x[t+1], out[steps-1] = x[steps-1].
 * hipad_add_rows / hipad_rows_sum.  Replaces: the planning branch's broadcast additions `embed + target-point embed +
 *   command embed + ego embed` and `feature + ego feature` (reference models/sparse_onedecoder.py, plan refinement: one
 *   launch per addend forward, one two-workgroup column reduction per addend backward).  base, out: (bs, n_rows,
 *   channels); rows0..2 (rows1 / rows2 may be NULL): (bs, channels), added to every row of their sample; channels % 4
 *   == 0, 16-byte aligned.  hipad_rows_sum: out (bs, channels) = sum over the rows of x (bs, n_rows, channels) -- the
 *   gradient of each row vector (the same for all of them); fixed summation order, no atomics.
 * hipad_keep_mask.  Replaces: the Bernoulli keep mask of DeformableFeatureAggregation's attn_drop (reference
 *   models/blocks.py:209-212; rand, compare, cast, rescale = four launches): out[i] = 1 / (1 - p_drop) with probability
 *   1 - p_drop else 0, drawn from (seed, *seed_dev, i) -- seed_dev (may be NULL) is a device step counter, so a replayed
 *   hipGraph draws a fresh mask every step.
 * hipad_dropout_add.  Replaces: `identity + dropout(x)` at the end of the attention and FFN blocks (reference
 *   models/attention.py:283-289, models/blocks.py:383-396; dropout + add = two launches forward): out[i] = base[i] +
 *   (keep(i) ? x[i] / (1 - p_drop) : 0), base may be NULL (then the call is the backward: x = output gradient).  The mask
 *   is a function of (seed, *seed_dev, i) as in hipad_keep_mask and is never stored.  n % 4 == 0, 16-byte aligned.
 * hipad_grid_mask.  Replaces: GridMask.forward (reference models/grid_mask.py:92-138; the detector's settings rotate = 1,
 *   offset = False): out = x * mask with the stripe mask evaluated per pixel from params = [apply, d, l, st_h, st_w] in
 *   device memory (this step's draw, uploaded outside a hipGraph).  x: (n, c, h, w) fp32 contiguous; out: fp32 or bf16
 *   (out_bf16) at out[n s0 + c s1 + y s2 + x s3] (element strides), e.g. bf16 channels-last for the first convolution.
 * ---------------------------------------------------------------------------------- */
#define HIPAD_MIX_MAX 16
int hipad_grid_mask(void *out, int out_bf16, const long long *out_strides, const float *x, const float *params, int n, int c,
                    int h, int w, int use_h, int use_w, int mode, hipad_stream_t stream);
int hipad_dropout_add(float *out, const float *x, const float *base, long long n, float p_drop, unsigned seed,
                      const unsigned *seed_dev, hipad_stream_t stream);
int hipad_keep_mask(float *out, long long n, float p_drop, unsigned seed, const unsigned *seed_dev, hipad_stream_t stream);
int hipad_chunk_mix(float *out, const float *x0, const float *x1, const float *weights, int bs, int in_chunks,
                    int out_chunks, int rows, int channels, hipad_stream_t stream);
int hipad_motion_query_embed(float *out, const float *cls, const float *box, const float *table, const float *freq,
                             long long n_anchor, int num_classes, int box_dim, int sin_col, int cos_col, int modes,
                             int steps, int half_dim, hipad_stream_t stream);
int hipad_add_rows(float *out, const float *base, const float *rows0, const float *rows1, const float *rows2, int bs,
                   int n_rows, int channels, hipad_stream_t stream);
int hipad_rows_sum(float *out, const float *x, int bs, int n_rows, int channels, hipad_stream_t stream);
int hipad_step_offsets(float *out, const float *x, long long rows, int steps, int dims, int adjoint, hipad_stream_t stream);

/* ------------------------------------------------------------------------------------
 * Dense-depth heads + loss on the flat pyramid (hip-ad_amd/csrc/depthloss.hip).  Replaces: DenseDepthNet.forward and
 * .loss (reference models/blocks.py:266-326) in training: per level a 1x1 convolution 256 -> 1 on the fp32-widened
 * level, exp, x focal / equal_focal, masked L1 against the sparse LiDAR depth, normalised per level by
 * max(1, n_valid * num_levels).
 *   feat: the frame's flat bf16 pyramid (bs, pyramid_rows, 256); level l occupies rows [row_offset, row_offset + cams *
 *   rows_per_cam) of every sample, camera-major (the layout the aggregation op reads).  gt: (bs * cams * rows_per_cam)
 *   fp32 per level, <= 0 = no target.  weight: 256 fp32, bias: 1 fp32.  focal: bs * cams or NULL.
 * forward: loss[0] = total, loss[1 + l] = the level's term; coef[l] and pred (one float per row of all levels, level
 *   after level) are kept for the backward; workspace: hipad_depth_loss_workspace() bytes, 8-byte aligned.  The error
 *   sums are 64-bit fixed-point integers: the value does not depend on the order of the atomics.
 * backward: grad_feat (bs, pyramid_rows, 256) fp32 += d loss / d feat (rows with a valid target only; plain
 *   read-modify-write: nothing else may write grad_feat concurrently); grad_weight / grad_bias of every level (may be
 *   NULL) += their gradients (fp32 atomics).  upstream: device scalar d total / d loss (NULL = 1).
 * ---------------------------------------------------------------------------------- */
#define HIPAD_DEPTH_MAX_LEVELS 4
typedef struct hipad_depth_level {
  const float *gt;
  const float *weight;
  const float *bias;
  float *grad_weight;
  float *grad_bias;
  int32_t rows_per_cam;
  int32_t row_offset;
} hipad_depth_level;
size_t hipad_depth_loss_workspace(void);
int hipad_depth_loss_forward(float *loss, float *coef, float *pred, void *workspace, size_t workspace_bytes,
                             const unsigned short *feat, long long pyramid_rows, const float *focal,
                             const hipad_depth_level *levels, int nlevels, int bs, int cams, float equal_focal,
                             float max_depth, float loss_weight, hipad_stream_t stream);
int hipad_depth_loss_backward(float *grad_feat, const float *pred, const float *coef, const float *upstream,
                              const unsigned short *feat, long long pyramid_rows, const hipad_depth_level *levels,
                              int nlevels, int bs, int cams, float max_depth, hipad_stream_t stream);

/* ------------------------------------------------------------------------------------
 * Image leg of the training data pipeline (hip-ad_amd/csrc/imgpipe.hip).  Replaces, for all camera images of one
 * sample and with the frames already in HBM as uint8 (n_img, src_h, src_w, 3):
 *   datasets/pipelines/augment.py:46-68      ResizeCropFlipImage._img_transform (PIL resize [bicubic] -> crop ->
 *                                            FLIP_LEFT_RIGHT -> rotate [nearest] -> float32)
 *   datasets/pipelines/transform.py:286-321  NormalizeMultiviewImage (mmcv.imnormalize)
 *   datasets/pipelines/transform.py:136-138  the HWC -> CHW transpose + stack of NuScenesSparse4DAdaptor
 * Geometry is bit-exact with Pillow (tests/test_imgpipe_*.py).  Host-side table builders (no GPU needed):
 *   hipad_resample_tables  Pillow's tap tables for resizing in_size -> out_size with the bicubic filter: bounds
 *                          (out_size, 2) = [first source index, tap count], coeffs (out_size, ksize) with 22 fractional
 *                          bits; returns ksize (> 0; call with NULL tables to size the buffers) or a negative status.
 *                          in_size == out_size gives identity taps (Pillow skips that pass).
 *   hipad_rotate_fixed     Image.rotate(angle_deg)'s inverse map about (width / 2, height / 2) as the six 16.16 fixed
 *                          point numbers Pillow's affine_fixed walks; returns 1 (a_out filled), 0 (angle % 360 == 0: the
 *                          copy path, nothing to apply) or a negative status (180 / 90 / 270 transpose paths, overflow).
 * Device entries:
 *   hipad_image_resize_rows  horizontal pass of source rows [row0, row0 + rows) -> tmp (n_img, rows, out_w, 3) uint8.
 *   hipad_image_finish       per output pixel: inverse rotate -> flip -> crop -> vertical pass on tmp -> [BGR->RGB]
 *                            -> [(v - mean) * (1 / std)] -> out[n * s[0] + c * s[1] + y * s[2] + x * s[3]] (element
 *                            strides: CHW fp32, HWC or channels-last).  bounds_v's first indices are relative to tmp's
 *                            first row (row0 already subtracted); crop_box = (x0, y0, x1, y1) in the resized image, zero
 *                            fill outside; rotate_a = the six numbers of hipad_rotate_fixed or NULL; mean / std both
 *                            NULL = raw float32 pixel values (ResizeCropFlipImage alone).
 * ---------------------------------------------------------------------------------- */
int hipad_resample_tables(int in_size, int out_size, int *bounds, int *coeffs);
int hipad_rotate_fixed(double angle_deg, int width, int height, int *a_out);
int hipad_image_resize_rows(unsigned char *tmp, const unsigned char *src, const int *bounds_h, const int *coeffs_h,
                            int ksize_h, int n_img, int src_h, int src_w, int row0, int rows, int out_w,
                            hipad_stream_t stream);
int hipad_image_finish(float *out, const long long *out_strides, const unsigned char *tmp, const int *bounds_v,
                       const int *coeffs_v, int ksize_v, int n_img, int tmp_rows, int resized_w, int resized_h,
                       const int *crop_box, int flip, const int *rotate_a, const float *mean, const float *std, int to_rgb,
                       hipad_stream_t stream);

/* ------------------------------------------------------------------------------------
 * Training-mode BatchNorm on bf16 channels-last activations, fused with the identity add and the ReLU that follow it
 * in a ResNet bottleneck (hip-ad_amd/csrc/batchnorm.hip).  Replaces, per norm layer of the image encoder (mmdet 2.28.2
 * ResNet / FPN as built by projects/configs/hipad_b2d_stage2.py:112-134 and run by models/sparse_detector.py:66-94),
 * the library's 3 + 3 BatchNorm launches plus the ReLU (and add) launches by two launches each way.
 *   x, y, residual, dy, dx, dres: (rows, channels) row-major bf16 = NHWC with rows = N*H*W, 16-byte aligned.
 *   channels in {64, 128, 256, 512, 1024, 2048} (hipad_bn_supported).
 *   sums / gsums: HIPAD_BN_REPLICAS x 2 x channels 64-bit words (8-byte aligned; declared float* for the allocator's sake:
 *                 2 floats per word), ZERO on entry.  Partial-sum copies in 64-bit fixed point -- integer accumulation is
 *                 order-independent, so the statistics (and everything computed from them) are bitwise reproducible
 *                 although the workgroups' atomics arrive in any order; consumed by the same call.
 *   save: 2 x channels floats (batch mean, 1 / sqrt(var + eps)) written by the forward for the backward.
 *   hipad_bn_forward:  y = relu?((x - mean) * rstd * gamma + beta (+ residual)); biased batch variance; when
 *                      running_mean / running_var are given they are updated in place with `momentum` (unbiased variance),
 *                      as torch.nn.BatchNorm2d does.
 *   hipad_bn_backward: dy' = dy gated by y > 0 (y = the forward output; NULL when no ReLU was fused);
 *                      dx = gamma * rstd * (dy' - mean(dy') - xhat * mean(dy' * xhat)); dres (may be NULL) = dy';
 *                      dgamma / dbeta (may be NULL) are ACCUMULATED into (+=): pass the parameters' gradient buffers.
 * ---------------------------------------------------------------------------------- */
#define HIPAD_BN_REPLICAS 4
int hipad_bn_supported(long long rows, int channels);
int hipad_bn_forward(void *y, float *save, float *sums, const void *x, const void *residual, const float *gamma,
                     const float *beta, float *running_mean, float *running_var, long long rows, int channels, float eps,
                     float momentum, int relu, hipad_stream_t stream);
/* The same with a GROUPED output: output rows [g * y_group_rows, (g + 1) * y_group_rows) are written y_group_stride_rows
 * rows apart -- the FPN's last norm layer writes each level straight into the flat pyramid the aggregation operator reads
 * (one group = one sample's block of the level; reference ops/__init__.py:33-103 copies the levels there instead). */
int hipad_bn_forward_grouped(void *y, float *save, float *sums, const void *x, const void *residual, const float *gamma,
                             const float *beta, float *running_mean, float *running_var, long long rows, int channels,
                             float eps, float momentum, int relu, long long y_group_rows, long long y_group_stride_rows,
                             hipad_stream_t stream);
int hipad_bn_backward(void *dx, void *dres, float *dgamma, float *dbeta, float *gsums, const void *dy, const void *y,
                      const void *x, const float *save, const float *gamma, long long rows, int channels,
                      hipad_stream_t stream);

/* Tuning knob (host side, process-wide): target number of (point, camera) pairs one
 * wavefront owns in the forward / backward kernels.  <=0 restores the default. */
void hipad_daf_set_pairs_per_wave(int fwd, int bwd);

#ifdef __cplusplus
}
#endif
#endif /* HIPAD_H_ */
