/* reni_hip.h -- C ABI of libreni_hip.so: the MI355X (gfx950) RENI forward / training hot path.
 *
 * This is the drop-in boundary underneath the reference's nn.Module surface.  The reference
 * (JADGardner/RENI) has no FFI of its own; the seam it offers is the call
 *     model_output = self.model(Z, directions)            src/lightning/RENI_module.py:105 (training)
 *                                                          src/lightning/RENI_module.py:78  (inference)
 * into src/models/RENI.py (SO2/SO3/None invariant encoding :23-60, SineLayer :63-87, decoder
 * :132-178) plus the loss of src/utils/loss_functions.py:6-71 and the autograd backward of both.
 * Each entry point below names the reference code it replaces.
 *
 * Conventions
 *  - every pointer is a DEVICE pointer on the current HIP device unless it says "host";
 *  - all tensors are fp32, row-major, contiguous unless a stride argument says otherwise;
 *  - `stream` is a hipStream_t passed as void* (0 = the null stream);
 *  - every function returns 0 on success or a negative RENI_E* code; the message of the last
 *    failure on the calling thread is available from reni_last_error();
 *  - the library allocates nothing per call: the caller owns params, gradients, I/O and the
 *    workspace (size from reni_workspace_bytes); the plan is immutable after creation.
 *
 * Flat parameter layout (`params`, `dparams`): the decoder's state_dict in the reference's own
 * order (src/models/RENI.py:132-178): net.0.linear.weight [H,F_in], net.0.linear.bias [H],
 * net.l.linear.weight [H,H], net.l.linear.bias [H] for l = 1..L, then the head weight [3,H] and
 * bias [3].  reni_param_count() gives the total.
 */
#ifndef RENI_HIP_H
#define RENI_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RENI_OK 0
#define RENI_EINVAL (-1)       /* bad argument / unsupported shape */
#define RENI_EWORKSPACE (-2)   /* workspace too small */
#define RENI_EHIP (-3)         /* a HIP runtime call failed */
#define RENI_EUNSUPPORTED (-4) /* configuration not supported by the compiled kernels */

/* reni_desc.equivariance : which invariant encoding (src/models/RENI.py:118-126) */
#define RENI_EQ_NONE 0
#define RENI_EQ_SO2 1
#define RENI_EQ_SO3 2
/* reni_desc.output_activation (src/models/RENI.py:173-176) */
#define RENI_ACT_NONE 0
#define RENI_ACT_TANH 1
#define RENI_ACT_EXP 2
/* reni_desc.dtype : arithmetic of the dense layers */
#define RENI_F32 0  /* fp32 MFMA (v_mfma_f32_32x32x2_f32), precise sin/cos: bit-for-bit an fmaf chain */
#define RENI_BF16 1 /* bf16 MFMA (v_mfma_f32_32x32x16_bf16), fp32 accumulate, fp32 sin argument    */

/* reni_desc.conditioning : how the latent code conditions the SIREN (src/models/RENI.py:862, 877-933) */
#define RENI_COND_CONCAT 0 /* Cond-by-Concat: RENIAutoDecoder / RENIVariationalAutoDecoder      RENI.py:90-399  */
#define RENI_COND_FILM 1   /* FiLM: RENIAutoDecoderFiLM / RENIVariationalAutoDecoderFiLM         RENI.py:407-858 */

/* loss_kind for reni_forward_loss_backward */
#define RENI_LOSS_MSE 0  /* RENITrainLoss      = WeightedMSE                  loss_functions.py:6-13,39-45 */
#define RENI_LOSS_TEST 1 /* RENITestLoss       = MSE + alpha*|Z|^2 + beta*WeightedCosine  :25-32,60-71     */

/* flags */
#define RENI_NEED_DW 1u /* produce decoder gradients (dparams)                      */
#define RENI_NEED_DZ 2u /* produce latent gradients (dZ)                            */
/* The loss weight is zero over whole regions (RENI_module.py:92-94: sineweight * mask, the notebook's inpainting masks): the library
 * may leave out work that cannot change the result -- 128-pixel tiles whose weights are all zero, and the statistics pass of
 * RENITestLoss's cosine term for images whose pixel-0 weight is zero (loss_functions.py:25-32 multiplies the term by that weight).
 * The regions are found on the device from `weight` in every call; results are the dense ones (the skipped terms are exact zeros).
 * Honoured by the frozen-decoder calls of the concat models (no RENI_NEED_DW, no output image requested; every width, fp32 and bf16),
 * ignored elsewhere. */
#define RENI_WEIGHT_SPARSE 4u
/* As RENI_WEIGHT_SPARSE, and the pixels with weight are PACKED into each image's first tiles (a position -> pixel list built on the
 * device per call), so a tile is left out unless it holds such pixels: Mask-3 at 128 x 256 keeps 19 % of the pixels in 148 of 256
 * tiles -- packed, in 49.  The same terms are then summed in another order: results equal the dense ones to fp32 rounding, not bit
 * for bit (run-to-run they stay bit-identical). */
#define RENI_WEIGHT_COMPACT 8u
/* reni_latent_step_rows_cached only: the lists' build reported no image with a live cosine term (summary[2] == 0: pixel 0 of every
 * image is masked, as in every mask of the reference's data/Masks) -- the statistics pass and its per-image kernel, which would visit
 * nothing and write zero coefficients, are not launched.  Same results, two launches fewer. */
#define RENI_WEIGHT_COS_CONSTANT 16u

typedef struct reni_plan reni_plan;

typedef struct reni_desc {
  int32_t equivariance;      /* RENI_EQ_*                                   RENI.py:118-126 */
  int32_t ndims;             /* latent rows ND (Z is [ND,3])                RENI.py:94      */
  int32_t hidden_features;   /* H: 32, 64, 128 or 256                       RENI.py:96      */
  int32_t hidden_layers;     /* L: number of hidden SineLayers after the first (L+1 sine layers) RENI.py:143 */
  int32_t out_features;      /* must be 3                                    RENI.py:98      */
  int32_t last_layer_linear; /* 1: linear head, 0: sine head                RENI.py:153-171 */
  int32_t output_activation; /* RENI_ACT_*                                  RENI.py:173-176 */
  float first_omega_0;       /*                                             RENI.py:139     */
  float hidden_omega_0;      /*                                             RENI.py:149     */
  int32_t dtype;             /* RENI_F32 | RENI_BF16 */
  int32_t conditioning;      /* RENI_COND_*.  FiLM: hidden_layers = siren_hidden_layers - 1 (RENI.py:563-568),
                                omegas unused, last_layer_linear = 1                     RENI.py:537-570 */
  int32_t mapping_layers;    /* FiLM: hidden layers of the mapping network (0 = the reni_film_model_* entry points are
                                not used)                                                RENI.py:482-496 */
  int32_t mapping_features;  /* FiLM: width of those layers                             RENI.py:482-496 */
} reni_desc;

/* Message of the last error raised on this thread ("" if none). */
const char* reni_last_error(void);

/* Replaces the constructor of RENIAutoDecoder / RENIVariationalAutoDecoder as far as the decoder
 * is concerned (src/models/RENI.py:91-178): validates the hyper-parameters, selects kernels. */
int reni_plan_create(const reni_desc* desc, reni_plan** out_plan);
void reni_plan_destroy(reni_plan* plan);

/* Number of fp32 elements of the flat decoder parameter buffer; F_in via reni_in_features. */
int64_t reni_param_count(const reni_plan* plan);
int32_t reni_in_features(const reni_plan* plan);

/* Bytes of workspace the calls below need for (B images) x (P directions).  `flags` as passed
 * to the call (0 for reni_forward).  It scales with B x P only through per-tile buffers (a tile = 128 directions of one image):
 *   every backward path            8 KB of per-tile partials (not kept by the bf16 H = 128 training instance: image runs instead)
 *   bf16, H = 128, L <= 5          32 KB per tile: the g_1 stream (concat and FiLM; FiLM adds one partial slot per image RUN:
 *                                  <= 3 x #CUs x runs x 0.33 MB, independent of B x P)
 *   bf16, H = 128, L > 5           hidden_layers x 70 KB per tile: the operand-image stream (k_dw_stream)
 *   bf16, H = 256                  (2 L + 1) x 64 KB per tile: every tile's phase stash + the g_l fragment stream (k_dw_frag)
 *   fp32, H = 256, concat          (2 L + 1) x 128 KB per tile: the same in fp32 (k_dw_frag32)
 * e.g. 64 images x 32768 directions (16 384 tiles): 0.7 GB at 5 x 128, 11 GB at 5 x 256 bf16, 23 GB at 5 x 256 fp32.
 * (H = 256 training: a call whose stream would exceed 16 GB -- RENI_FRAG_WS_CAP_MB overrides -- runs in equal chunks of whole images,
 * one after the other, and the workspace is sized for one chunk: the fp32 example above is two passes of 32 images, 11.5 GB each.)
 * The H <= 128 stash ring and the per-workgroup gradient partials do not grow with B x P. */
size_t reni_workspace_bytes(const reni_plan* plan, int64_t B, int64_t P, uint32_t flags);

/* out[B,P,3] = model(Z, D) under no_grad -- replaces InvariantRepresentation + self.net(x)
 * (src/models/RENI.py:225-233) as called from RENI.forward (src/lightning/RENI_module.py:75-78).
 * Z [B,ND,3]; D [B,P,3] with `d_batch_stride` elements between images (0 = one shared grid
 * [P,3], which is what RENI_module.py:77 materialises with .repeat). */
int reni_forward(const reni_plan* plan, int64_t B, int64_t P, const float* Z, const float* D,
                 int64_t d_batch_stride, const float* params, float* out, void* ws,
                 size_t ws_bytes, void* stream);

/* Fused model(Z,D) -> loss -> backward: replaces RENI_module.py:105 + the criterion call
 * (:117 / :126-128) + loss.backward() for the decoder and latents.
 *   target  : element (b,p,c) at target[b*tgt_strides[0] + p*tgt_strides[1] + c*tgt_strides[2]]
 *             (accepts the channel-planar view of RENI_module.py:83-84 without a copy);
 *   weight  : sineweight (x mask), element (b,p,c) at weight[b*w_strides[0] + ...]; a stride of 0
 *             broadcasts (the reference .repeat's one [1,P,3] grid, RENI_module.py:90-94);
 *   loss_kind, alpha, beta : RENI_LOSS_*; alpha/beta only for RENI_LOSS_TEST;
 *   out     : optional [B,P,3] model output (NULL to skip the store);
 *   loss_terms[4] : (loss, mse, prior, cosine) summed over the batch as the reference does;
 *   dZ [B,ND,3]   : d loss / d Z   (written when flags & RENI_NEED_DZ);
 *   dparams       : flat decoder gradient, OVERWRITTEN (written when flags & RENI_NEED_DW). */
int reni_forward_loss_backward(const reni_plan* plan, int64_t B, int64_t P, const float* Z,
                               const float* D, int64_t d_batch_stride, const float* params,
                               const float* target, const int64_t tgt_strides[3],
                               const float* weight, const int64_t w_strides[3], int32_t loss_kind,
                               float alpha, float beta, uint32_t flags, float* out,
                               float* loss_terms, float* dZ, float* dparams, void* ws,
                               size_t ws_bytes, void* stream);

/* The same with the batch's latents given as ROWS OF A TABLE: image b uses Z_table[idx[b]] ([n_rows][ndims][3], idx on the
 * device) -- `Z = self.model.Z[idx, :, :]` of the training step (RENI_module.py:97-103) happens inside the prologue kernel
 * instead of as a separate gather.  dZ is [B][ndims][3] in batch order, as above.  An idx[b] outside [0, n_rows) cannot be
 * reported without a host synchronisation: it reads no memory outside the table, and image b's outputs, the loss terms and
 * every gradient of the call come out NaN (the reference's `Z[idx]` raises a device-side assert). */
int reni_forward_loss_backward_rows(const reni_plan* plan, int64_t B, int64_t P, const float* Z_table, int64_t n_rows, const int64_t* idx,
                                    const float* D, int64_t d_batch_stride, const float* params, const float* target,
                                    const int64_t target_strides[3], const float* weight, const int64_t weight_strides[3],
                                    int32_t loss_kind, float alpha, float beta, uint32_t flags, float* out,
                                    float* loss_terms, float* dZ, float* dparams, void* workspace, size_t workspace_bytes,
                                    void* stream);

/* One whole training step of the reference's FIT_DECODER loop in one call (RENI_module.py:80-146 training_step, :178-192 the
 * optimiser over decoder + latent table; run.py's `trainer.fit` iteration): reni_forward_loss_backward_rows with
 * RENI_NEED_DW | RENI_NEED_DZ, THEN reni_adam_step2 -- same arguments, same results, bit for bit -- in fewer launches:
 *   - the optimiser step is one launch that also sums layer 1's weight-gradient partials (on the persistent bf16 path their
 *     reduction is otherwise a launch of its own behind k_reni_dw1);
 *   - the NEXT batch's prologue (`idx_next`: its latent rows gathered, A_b, layer-0 operands, the packed weight images of the
 *     UPDATED decoder) is run at the end of this call, so the next call starts with its main kernel.
 * `stage_state` (in / out, zero before the first call) says whether -- and into which of two copies -- the previous call staged this
 * call's prologue.  The caller resets it to zero whenever it changes B, P, `params` or `Z_table` between two calls, and passes as
 * idx what it announced as `idx_next`: the library checks the indices on the device -- on every path, persistent and generic kernels alike --
 * and SKIPS a step whose batch is not the staged one, loudly: loss terms and the latent gradient come out NaN, and parameters,
 * latent table and all four Adam moments stay exactly as they were (the optimiser launch reads the check's flag).  The staged copies
 * live in `workspace`: ANY other library call that is given the same workspace between two calls of this function (a validation
 * forward, a fused loss, another B or P) overwrites them -- reset `stage_state` to zero after such a call (reni_amd/ops.py counts
 * its workspace hand-outs and does).  idx_next = NULL: nothing is staged (the next call runs its own prologue).  Z_table and params are
 * updated in place; dZ [B,ND,3] and dparams receive the step's gradients as reni_forward_loss_backward_rows returns them.  One
 * process; the data-parallel step is reni_train_step_rows_dp below. */
int reni_train_step_rows(const reni_plan* plan, int64_t B, int64_t P, float* Z_table, int64_t n_rows, const int64_t* idx,
                         const int64_t* idx_next, const float* D, int64_t d_batch_stride, float* params, const float* target,
                         const int64_t target_strides[3], const float* weight, const int64_t weight_strides[3], int32_t loss_kind,
                         float alpha, float beta, float* m_dec, float* v_dec, float* m_lat, float* v_lat, float lr, float b1,
                         float b2, float eps, int64_t step, float grad_scale, uint32_t* stage_state, float* loss_terms, float* dZ,
                         float* dparams, void* workspace, size_t workspace_bytes, void* stream);

/* The same step for one rank of a DATA-PARALLEL job (BASELINE config 3; the reference: PyTorch-Lightning's DDP strategy, run.py:97-110,
 * whose all-reduce averages the shared decoder's gradients over the ranks between backward and optimizer.step): everything
 * reni_train_step_rows does, with the exchange INSIDE the call -- fwd + loss + bwd, the partial reductions, an in-place RCCL
 * all-reduce(sum) of the flat decoder gradient on `comm` (reni_rccl_comm_create; the caller's stream), the one optimiser launch with
 * `grad_scale` (pass 1 / world_size: the mean), the next batch's prologue.  Same launches as the one-process step plus layer 1's
 * partial reduction (its sum must be in the gradient buffer before the collective) plus the collective.  The latent table needs no
 * exchange: every image's row has exactly one owner (reni_amd/dist.py).  dparams returns the SUM over the ranks.
 * overlap != 0: the gradient of layers >= 2 + head (final before the ring kernel runs) is all-reduced on the library's own
 * communication stream beside the rest of the backward pass, the remainder (first layer, layer 1) behind it on the caller's stream --
 * two collectives on `comm`, issued in the same order on every rank; element for element the same sums.
 * With a one-rank communicator the results are bit-equal to reni_train_step_rows (tests/test_gpu_dist.py).
 * The skip decision of a staged step (see reni_train_step_rows) is all-reduced with MAX over `comm` in front of the gradient -- one
 * more 4-byte collective per call, issued on every rank whether or not its own step was staged: a batch that is not the staged one on
 * ANY rank skips the step on EVERY rank (the replicas stay replicas); on a skipped step the layer-1 slice of dparams, which this
 * call's optimiser launch would have written, is NaN like the rest of the gradient. */
int reni_train_step_rows_dp(const reni_plan* plan, int64_t B, int64_t P, float* Z_table, int64_t n_rows, const int64_t* idx,
                            const int64_t* idx_next, const float* D, int64_t d_batch_stride, float* params, const float* target,
                            const int64_t target_strides[3], const float* weight, const int64_t weight_strides[3],
                            int32_t loss_kind, float alpha, float beta, float* m_dec, float* v_dec, float* m_lat, float* v_lat,
                            float lr, float b1, float b2, float eps, int64_t step, float grad_scale, void* comm, int32_t overlap,
                            uint32_t* stage_state, float* loss_terms, float* dZ, float* dparams, void* workspace,
                            size_t workspace_bytes, void* stream);

/* Backward for an arbitrary upstream gradient dout[B,P,3] (generic autograd use of
 * model(Z,D)); the forward is recomputed inside the same fused kernel. */
int reni_backward(const reni_plan* plan, int64_t B, int64_t P, const float* Z, const float* D,
                  int64_t d_batch_stride, const float* params, const float* dout, uint32_t flags,
                  float* dZ, float* dparams, void* ws, size_t ws_bytes, void* stream);

/* ---- FiLM conditioning (src/models/RENI.py:407-858) ------------------------------------------------
 * The per-SAMPLE work of RENI*FiLM.forward_with_frequencies_phase_shifts (RENI.py:665-676, FiLMLayer
 * :508-519) runs in the fused kernels; the per-IMAGE glue stays with the caller (reni_amd/film.py):
 *   - the mapping network (RENI.py:470-505) is evaluated once per image -- the reference evaluates it per
 *     pixel on repeated rows (RENI.py:413-447), same values -- giving freq = 15 f + 30 and phase (RENI.py:666);
 *   - the first FiLMLayer acts on Siren_Input = [|d_xz|, d_y, D_xz Z_xz^T] (SO2, RENI.py:441) or D Z^T (SO3,
 *     RENI.py:410), which is linear in (dx, dy, dz, r): theta_0 = A_b (dx, dy, dz, r, 1) with the per-image
 *     A [B,H,8] (columns dx, dy, dz, r, 1, 3 unused) = freq_0 . (W_0 x + b_0) + phase_0 folded by the caller.
 * Flat `params`: net.0.layer.weight [H,F0], net.0.layer.bias [H] (unused by the kernels), net.l.layer.weight
 * [H,H], .bias [H] for l = 1..L, final_layer.weight [3,H], .bias [3];  F0 = reni_in_features().
 * film [B,L,2,H]: film[b][l-1][0] = freq, [1] = phase of hidden FiLM layer l of image b.
 * Gradients: dA [B,H,8] (d loss / d A), dfilm [B,L,2,H], dparams (flat; the net.0 slot is zero-filled --
 * its gradient follows from dA in the caller's glue).  loss_terms = (mse + cosine, mse, 0, cosine): the
 * latent prior of RENITestLoss (alpha |Z|^2) is the caller's. */
int reni_film_forward(const reni_plan* plan, int64_t B, int64_t P, const float* D, int64_t d_batch_stride,
                      const float* A, const float* film, const float* params, float* out, void* ws,
                      size_t ws_bytes, void* stream);
int reni_film_forward_loss_backward(const reni_plan* plan, int64_t B, int64_t P, const float* D,
                                    int64_t d_batch_stride, const float* A, const float* film,
                                    const float* params, const float* target, const int64_t tgt_strides[3],
                                    const float* weight, const int64_t w_strides[3], int32_t loss_kind,
                                    float beta, uint32_t flags, float* out, float* loss_terms, float* dA,
                                    float* dfilm, float* dparams, void* ws, size_t ws_bytes, void* stream);
int reni_film_backward(const reni_plan* plan, int64_t B, int64_t P, const float* D, int64_t d_batch_stride,
                       const float* A, const float* film, const float* params, const float* dout,
                       uint32_t flags, float* dA, float* dfilm, float* dparams, void* ws, size_t ws_bytes,
                       void* stream);

/* FiLM, whole model: the per-image glue of the three calls above runs in HIP too (k_film_minput / _linear / _fold,
 * k_film_dout / _linear_t / _grads), so these take the latent codes and the mapping network's parameters and return
 * their gradients -- the drop-in for RENI*FiLM.forward (RENI.py:628-676) + criterion + loss.backward().
 *   map_params / dmap_params : mapping_network.network.{0,2,...}.{weight [N_i,K_i], bias [N_i]} flat, in state-dict
 *                              order (RENI.py:482-496); reni_film_map_param_count() elements;
 *   dparams                  : flat net.* / final_layer.* gradient INCLUDING the first layer's slot;
 *   loss_terms               : (loss, mse, prior, cosine) with prior = alpha |Z|^2 (RENI_LOSS_TEST). */
int64_t reni_film_map_param_count(const reni_plan* plan);
int reni_film_model_forward(const reni_plan* plan, int64_t B, int64_t P, const float* Z, const float* D,
                            int64_t d_batch_stride, const float* params, const float* map_params, float* out,
                            void* ws, size_t ws_bytes, void* stream);
int reni_film_model_forward_loss_backward(const reni_plan* plan, int64_t B, int64_t P, const float* Z, const float* D,
                                          int64_t d_batch_stride, const float* params, const float* map_params,
                                          const float* target, const int64_t tgt_strides[3], const float* weight,
                                          const int64_t w_strides[3], int32_t loss_kind, float alpha, float beta,
                                          uint32_t flags, float* out, float* loss_terms, float* dZ, float* dparams,
                                          float* dmap_params, void* ws, size_t ws_bytes, void* stream);
int reni_film_model_backward(const reni_plan* plan, int64_t B, int64_t P, const float* Z, const float* D,
                             int64_t d_batch_stride, const float* params, const float* map_params, const float* dout,
                             uint32_t flags, float* dZ, float* dparams, float* dmap_params, void* ws, size_t ws_bytes,
                             void* stream);

/* torch.optim.Adam(lr, betas=(b1,b2), eps) step on a flat buffer (RENI_module.py:192: the
 * reference always uses the default betas (0.9, 0.999), eps 1e-8).  `step` is the 1-based step
 * count; g is multiplied by grad_scale first (1/world_size after a sum all-reduce). */
int reni_adam_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, float b1,
                   float b2, float eps, int64_t step, float grad_scale, void* stream);

/* One iteration of the reference's FIT_LATENT loop -- the test-time optimisation of examples.ipynb cell 4: training_step with a fixed
 * decoder (RENI_module.py:92-103, 126-128), loss.backward(), Adam on the latent table (:178-192) -- in one call:
 * reni_forward_loss_backward_rows with flags = RENI_NEED_DZ | `flags`, THEN reni_adam_rows_step(Z_table, dZ, idx, ...) with
 * grad_scale 1 -- same arguments, same results.  `flags`: 0, RENI_WEIGHT_SPARSE or RENI_WEIGHT_COMPACT (a masked weight).
 * Z_table, m_lat, v_lat are updated in place; dZ [B,ND,3] receives the step's gradient; concat plans, one process. */
int reni_latent_step_rows(const reni_plan* plan, int64_t B, int64_t P, float* Z_table, int64_t n_rows, const int64_t* idx, const float* D,
                          int64_t d_batch_stride, const float* params, const float* target, const int64_t target_strides[3],
                          const float* weight, const int64_t weight_strides[3], int32_t loss_kind, float alpha, float beta,
                          uint32_t flags, float* m_lat, float* v_lat, float lr, float b1, float b2, float eps, int64_t step,
                          float* loss_terms, float* dZ, void* workspace, size_t workspace_bytes, void* stream);

/* The lists RENI_WEIGHT_SPARSE / RENI_WEIGHT_COMPACT work from -- which tiles carry weight, which images' cosine term is live, and
 * (COMPACT) the position -> pixel map -- depend on the loss weight alone, and the inpainting mask of a FIT_LATENT run is constant
 * over its 2 400 epochs (/root/reference/configs/experiment.yaml:47-48; RENI_module.py:92-94 multiplies the same mask into the sine
 * weight every step).  reni_latent_step_rows rebuilds them from the weight in every call (two or three dependent launches and an
 * 8 MB read in front of a 0.2 ms step); here the caller builds them ONCE into a buffer of its own and hands them to every step:
 *   reni_weight_lists_bytes(B, P)            size of the buffer (256-byte aligned device memory);
 *   reni_weight_lists_build(...)             fills it for this weight, these strides and `flags` (SPARSE or COMPACT), on `stream`;
 *                                            optionally reports what the lists hold (one synchronisation);
 *   reni_latent_step_rows_cached(..., lists) reni_latent_step_rows with the list-building launches left out: the same kernels on the
 *                                            same lists, results BIT-EQUAL to the rebuilt-every-call entry point.
 * The caller rebuilds the lists whenever the weight, B or P changes; `flags` of the step = `flags` of the build.
 * The library remembers, on the host, which buffers reni_weight_lists_build filled for which (B, P, mode) -- the last 64 builds --
 * and reni_latent_step_rows_cached returns RENI_EINVAL for a buffer it has no record of or that was built for another shape or mode
 * (the lists' layout depends on B and P: read at another shape they would be left).  What the record cannot see is a weight that
 * changed under unchanged lists. */
size_t reni_weight_lists_bytes(int64_t B, int64_t P);
/* summary_host (optional, 3 ints): after ONE stream synchronisation -- tiles the main pass will visit, tiles of the statistics pass,
 * images whose cosine term is live (0: the steps may carry RENI_WEIGHT_COS_CONSTANT). */
int reni_weight_lists_build(int64_t B, int64_t P, const float* weight, const int64_t weight_strides[3], uint32_t flags, void* lists,
                            size_t lists_bytes, int32_t* summary_host, void* stream);
int reni_latent_step_rows_cached(const reni_plan* plan, int64_t B, int64_t P, float* Z_table, int64_t n_rows, const int64_t* idx,
                                 const float* D, int64_t d_batch_stride, const float* params, const float* target,
                                 const int64_t target_strides[3], const float* weight, const int64_t weight_strides[3],
                                 int32_t loss_kind, float alpha, float beta, uint32_t flags, const void* weight_lists, float* m_lat,
                                 float* v_lat, float lr, float b1, float b2, float eps, int64_t step, float* loss_terms, float* dZ,
                                 void* workspace, size_t workspace_bytes, void* stream);

/* The same Adam step over a table p [n_rows][row_len] whose gradient is given for the B rows idx[0..B) only
 * (g_rows [B][row_len], idx int64 on the device; repeated indices accumulate).  All other rows have gradient zero and
 * still move by their momentum: the dense torch.optim.Adam over the whole latent table that the reference runs
 * (RENI_module.py:178-192 puts model.Z / model.mu, all N rows, into the optimiser; a batch touches B of them). */
int reni_adam_rows_step(float* p, const float* g_rows, const int64_t* idx, int64_t B, int64_t row_len, float* m, float* v,
                        int64_t n_rows, float lr, float b1, float b2, float eps, int64_t step, float grad_scale,
                        void* stream);

/* One launch for both updates of a training step: reni_adam_step on the flat decoder buffer (p, g, m, v, n) and
 * reni_adam_rows_step on the latent table (table, g_rows, idx, B, row_len, tm, tv, n_rows) with the same
 * hyper-parameters -- the reference has ONE torch.optim.Adam over the decoder and the latent table (RENI_module.py:178-192). */
int reni_adam_step2(float* p, const float* g, float* m, float* v, int64_t n, float* table, const float* g_rows,
                    const int64_t* idx, int64_t B, int64_t row_len, float* tm, float* tv, int64_t n_rows, float lr, float b1,
                    float b2, float eps, int64_t step, float grad_scale, void* stream);

/* Self-test of the MFMA fragment layouts the kernels rely on; out (host pointer) receives the
 * number of mismatching elements per probe (0 = layout as assumed). */
int reni_selftest_layouts(int32_t* out_host_mismatch, int32_t n_probes);

/* Optional timing of the fused main kernel (the dominant kernel of the path) with HIP events recorded
 * on the caller's stream around each launch.  reni_profile_read synchronises the recorded events and
 * returns their summed duration and count since the last reset.  Used by bench.py for `roofline`. */
int reni_profile_enable(int32_t on);
int reni_profile_read(double* total_ms, int64_t* launches, int32_t reset);
/* The same, restricted to one kind of launch: 0 fused forward+loss+backward (what reni_profile_read returns),
 * 1 the statistics pass of RENITestLoss's cosine term, 2 plain inference (reni_forward), 3 the kernel that finishes the
 * backward pass from the g_1 stream behind the persistent training kernel (k_reni_l0_ring / k_reni_dw1_ring / k_reni_dw1),
 * 4 the weight-gradient consumers of the fragment / operand streams (k_dw_frag, k_dw_frag32, k_dw_stream, k_wide_head_dw),
 * 5 the data-parallel exchange inside reni_train_step_rows_dp (the skip flag's and the gradient's all-reduce, on the caller's stream),
 * -1 all of them. */
int reni_profile_read_kind(int32_t kind, double* total_ms, int64_t* launches, int32_t reset);
/* Shortest and longest launch of one kind among those recorded since the last reset (bench.py prints them beside the average:
 * boxes of the pool differ by several per cent, a cross-round delta is read against that spread).  Does not reset. */
int reni_profile_minmax(int32_t kind, double* min_ms, double* max_ms);

/* Diagnostic probe of the LDS transpose-read instruction (ds_read_b64_tr_b16): LDS holds u16 element i = i;
 * lane l reads at byte address 8*l (mode 0) or lane_addr_host[l] (mode 1); out_host[4*l + e] = element e. */
int reni_probe_tr(const int32_t* lane_addr_host, int32_t mode, uint16_t* out_host);

/* ---- environment-map Blinn-Phong shading of a G-buffer (FIT_INVERSE task) ------------------------------------
 * Replaces the arithmetic of blinn_phong_shading_env_map (src/utils/pytorch3d_envmap_shader.py:46-116) behind the
 * two interpolate_face_attributes calls (:68-73): the caller hands over the per-pixel interpolated vertex normals
 * and positions (the rasteriser is pytorch3d's, outside this library).
 *   normals, positions : [NP][3] device, NP = render pixels; NOT normalised (the library applies F.normalize(eps=1e-6)
 *                        as :82,:93 do); rows of zeros where no face covers the pixel (pix_to_face < 0)
 *   cam_*              : cameras.get_camera_center() (:78)
 *   light_dirs         : [J][3] unit directions of the environment-map texels (EnvironmentMap.directions, :75) of image
 *                        b at light_dirs + b * dirs_batch_stride (floats; 0 = one grid shared by all images, which is
 *                        what the reference's directions.repeat(B,1,1) amounts to, RENI_module.py:376)
 *   light_colors       : [B][J][3] EnvironmentMap.environment_map = map * sineweight (:41,:77)
 *   shininess, kd, ks  : materials.shininess (:79, 500 in build_renderer :186), kd, ks = 1 - kd (:199)
 *   colors             : [B][NP][3] = kd * diffuse + (s+2)/(4(2-exp(-s/2))) * ks * specular   (:112-115)
 * reni_envmap_shade_backward returns d loss / d light_colors [B][J][3] for an upstream d loss / d colors [B][NP][3]
 * (the only tensor of the shader that carries a gradient in the reference: the mesh and the grid are constants).
 * ws: reni_envmap_shade_workspace_bytes(B, NP, J) bytes, 256-byte aligned, for the partial sums. */
size_t reni_envmap_shade_workspace_bytes(int64_t B, int64_t NP, int64_t J);
int reni_envmap_shade(int64_t B, int64_t NP, int64_t J, const float* normals, const float* positions, float cam_x,
                      float cam_y, float cam_z, const float* light_dirs, int64_t dirs_batch_stride,
                      const float* light_colors, float shininess, float kd, float ks, float* colors, void* ws,
                      size_t ws_bytes, void* stream);
int reni_envmap_shade_backward(int64_t B, int64_t NP, int64_t J, const float* normals, const float* positions, float cam_x,
                               float cam_y, float cam_z, const float* light_dirs, int64_t dirs_batch_stride,
                               const float* dcolors, float shininess, float kd, float ks, float* dlight_colors, void* ws,
                               size_t ws_bytes, void* stream);

/* ---- HDR image epilogue / prologue ------------------------------------------------
 * reni_unnormalise_srgb replaces, on the device and in one call, the reference's viewing chain
 *   UnMinMaxNormlise(minmax)   src/utils/custom_transforms.py:14-21   y = exp(0.5 (x + 1)(m1 - m0) + m0)
 *   sRGB                       src/utils/utils.py:30-42               y / q_b, clamp to [0,1], sRGB transfer curve, where
 *                              q_b = the nested 0.98-quantile over channels, then rows, then columns of image b
 *                              (torch.quantile semantics: rank q (n - 1) in fp32, ATen's lerp between the neighbours)
 * as the image callbacks apply it to a model output (src/lightning/callbacks.py; RENI_module.py:108 un-normalises in the
 * FIT_INVERSE step).
 *   img        : element (b, c, h, w) at img[b*strides[0] + c*strides[1] + h*strides[2] + w*strides[3]] (floats) -- a
 *                model output [B, H*W, 3] is read in place with strides {3HW, 1, 3W, 3}, a [B,3,H,W] batch with {3HW, HW, W, 1}
 *   unnormalise: 1 = apply UnMinMaxNormlise(minmax0, minmax1) first, 0 = img is linear already (plain sRGB())
 *   srgb       : 1 = produce out_srgb [B][3][H][W]; 0 = only the linear image
 *   out_linear : [B][3][H][W] linear HDR, or NULL when not wanted (then srgb must be 1)
 *   ws         : reni_image_workspace_bytes(B, H, W) bytes, 256-byte aligned (needed when srgb = 1); H, W <= 4096
 * reni_minmax_normalise is the forward transform MinMaxNormalise (custom_transforms.py:4-12) over the n elements of one
 * image: clip to [smallest positive, largest finite value of the image] -> log -> 2 (. - m0) / (m1 - m0) - 1. */
size_t reni_image_workspace_bytes(int64_t B, int64_t H, int64_t W);
int reni_unnormalise_srgb(int64_t B, int64_t H, int64_t W, const float* img, const int64_t strides[4], int32_t unnormalise,
                          double minmax0, double minmax1, int32_t srgb, float* out_srgb, float* out_linear, void* ws,
                          size_t ws_bytes, void* stream);
int reni_minmax_normalise(int64_t n, const float* img, double minmax0, double minmax1, float* out, void* ws, size_t ws_bytes,
                          void* stream);

/* ---- the data-parallel exchange step over RCCL -------------------------------
 * Replaces, for the flat decoder gradient, what Lightning's DDP wrapper does in the reference (run.py:97-110:
 * strategy="ddp" -> NCCL all-reduce of every parameter's gradient, mean over ranks): ONE in-place ncclAllReduce(sum) of
 * the n floats at `flat`, then flat *= scale (1 / world_size for DDP's mean), both on `stream`.  Latent rows are owned by
 * one rank each and are never exchanged.
 * `comm` is an ncclComm_t.  It may come from any RCCL in the process; the three helpers below make one without PyTorch:
 * rank 0 calls reni_rccl_unique_id and hands the 128 bytes to the other ranks (any side channel), then every rank calls
 * reni_rccl_comm_create (collective) with its device current.  librccl is loaded on first use: without it these four
 * return RENI_EUNSUPPORTED and the rest of the library is unaffected. */
typedef struct { char internal[128]; } reni_rccl_id; /* layout of ncclUniqueId */
int reni_rccl_unique_id(reni_rccl_id* id);
int reni_rccl_comm_create(const reni_rccl_id* id, int32_t nranks, int32_t rank, void** comm);
int reni_rccl_comm_destroy(void* comm);
int reni_allreduce_grads(void* comm, float* flat, size_t n, float scale, void* stream);

/* Launch geometry chosen for (B,P): workgroups, threads, dynamic LDS bytes (diagnostics). */
int reni_launch_info(const reni_plan* plan, int64_t B, int64_t P, int32_t* info4);

/* Which kernels a backward call of this shape will take (diagnostics; no reference counterpart -- the reference has one path).
 * info8 = { persistent kernels (k_reni_train_bf16) 0/1, dW_1 kernel 0 none / 1 k_reni_dw1_ring / 2 k_reni_dw1,
 *           side stream 0/1, images per chunk of the H = 256 training path (= B: one pass), operand stream 0/1,
 *           fragment stream 0 / 1 bf16 / 2 fp32, environment overrides in force (bit 0 RENI_NO_PERSIST, 1 RENI_NO_SIDE_STREAM,
 *           2 RENI_FRAG_WS_CAP_MB, 3 RENI_DW1_OLD; 0 in a clean environment), workgroups }.
 * The selectors are read from the environment ONCE, at reni_plan_create (a switch is on when its variable is set to anything but ""
 * or "0"); every alternative is a tested, correct path, and
 * bench.py prints this record so that a stray variable cannot silently change what is measured. */
int reni_path_info(const reni_plan* plan, int64_t B, int64_t P, uint32_t flags, int32_t* info8);

/* Data-parallel overlap hook (reference: DDP overlaps its bucketed all-reduce with the tail of backward, run.py:97).  While an
 * event is set (thread-local; NULL clears it), every reni_forward_loss_backward[_rows] call with RENI_NEED_DW records it on the
 * call's stream at the point where dparams[n_first + H*H + H ...) -- layers >= 2 and the head, 28 % of the gradient at config 2 -- is
 * final: on the persistent path that is before k_reni_dw1 runs (~0.13 ms before the call's work ends), on every other path at
 * the end of the call.  The caller makes its communication stream wait for the event and all-reduces that slice there, the
 * rest behind the call as before.  `hip_event` is a hipEvent_t created by the caller on the buffers' device. */
int reni_set_grad_ready_event(void* hip_event);

/* Kernel launches the library has issued in this process (all entry points, all streams); reset != 0 returns the count and
 * sets it to zero.  bench.py reports launches per step: at small problems a step costs its dependent launches. */
int64_t reni_launch_count(int32_t reset);

#ifdef __cplusplus
}
#endif
#endif /* RENI_HIP_H */
