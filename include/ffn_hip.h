/*
 * ffn_hip.h -- C ABI of libffn_hip.so: the MI355X (gfx950) kernels behind the
 * fourier_feature_nets volume-rendering hot path.
 *
 * The reference (matajoh/fourier_feature_nets) is pure Python on PyTorch and has no
 * FFI of its own; its "operator interface" for this path is the set of Python methods
 * cited next to each entry point below (paths relative to the reference root).  Each
 * entry point replaces the ATen op sequence those lines launch.
 *
 * Conventions
 *   - every function returns 0 on success or a hipError_t value; the message of the
 *     last failure on the calling thread is available from ffn_last_error_string();
 *     nothing throws across this boundary
 *   - the caller owns every buffer and passes raw DEVICE pointers; all arrays are
 *     contiguous row-major float32 unless the comment says otherwise
 *   - no hidden allocation, no hidden synchronisation; kernels are enqueued on
 *     `stream` (a hipStream_t passed as void*; NULL = the default stream)
 *   - functions are re-entrant and keep no mutable global state
 *   - one device per call: the device that is current on the calling thread
 */
#ifndef FFN_HIP_H
#define FFN_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FFN_ABI_VERSION 3

int ffn_abi_version(void);
const char* ffn_last_error_string(void);

/* ------------------------------------------------------------------------------------
 * K1  ray generation + AABB slab test.
 * Replaces CameraInfo.raycast (camera_info.py:99-109) and RaySampler._near_far
 * (ray_sampler.py:202-232) for every pixel of every camera.
 *   unproj      (C,16)  per-camera 4x4 pixel->world matrix (camera_info.py:66-70,
 *                       computed on the host exactly as the reference does)
 *   cam_pos     (C,3)   camera positions (extrinsics[:3,3])
 *   points      (W*H,2) explicit pixel coordinates shared by all cameras, or NULL for the
 *                       integer grid x = id % W, y = id / W (ray_sampler.py:133-136)
 *   box_lo/hi   (3)     HOST pointers: AABB corners (ray_sampler.py:101-104)
 *   starts      (C*W*H,3), directions (C*W*H,3), near_far (2,C*W*H)
 *   valid       (C*W*H) uint8: 1 where near < far (the complement of invalid_rays)
 * Ray id = cam*W*H + y*W + x, integer pixel coordinates, no half-pixel offset.
 */
int ffn_raygen_nearfar(const float* unproj, const float* cam_pos, const float* points,
                       int num_cameras, int width, int height, const float* box_lo,
                       const float* box_hi,
                       float* starts, float* directions, float* near_far, uint8_t* valid,
                       void* stream);

/* ------------------------------------------------------------------------------------
 * K2a  t-sampling: gather + anneal + linspace (+ stratified jitter).
 * Replaces RaySampler.sample up to t_values (ray_sampler.py:364-386) and utils.linspace
 * (utils.py:179-194).  Bit-exact w.r.t. the reference for identical inputs: every
 * multiply and add is rounded separately (no FMA contraction).
 *   near_far    (2,num_rays_total)
 *   ray_index   (R) int64 ray ids
 *   unit        (count) torch.linspace(0,1,count) computed on the host
 *   noise       (R,count) uniform [0,1) block or NULL (not stratified)
 *   anneal      factor already clamped to [anneal_start,1]; pass a negative value when
 *               no annealing applies (step is None or step >= num_anneal_steps)
 *   t_out       (R, t_stride) -- the first `count` entries of each row are written
 */
int ffn_sample_t(const float* near_far, int64_t num_rays_total, const int64_t* ray_index,
                 int num_rays, int count, const float* unit, const float* noise,
                 float anneal, float* t_out, int t_stride, void* stream);

/* K2b  positions = start + t * dir and view_directions = dir repeated
 * (ray_sampler.py:394-397).  positions/views are (R,S,3); views may be NULL. */
int ffn_materialise_samples(const float* starts, const float* directions,
                            const int64_t* ray_index, const float* t_values, int num_rays,
                            int num_samples, float* positions, float* views, void* stream);

/* K2a + K2b in one launch (ray_sampler.py:372-397 for a sampler WITHOUT an opacity model, where
 * nothing merges into t between the two): the arguments of ffn_sample_t with t_stride == count,
 * plus the ray starts / directions; t_out (R,count), positions / views (R,count,3), views may be
 * NULL.  Bit-identical to ffn_sample_t followed by ffn_materialise_samples.  The three outputs
 * must be 16-byte aligned (the kernel stores whole float4 lines of 1024-sample chunks); a pointer
 * that is not is refused with an error, never written through. */
int ffn_sample_materialise(const float* near_far, int64_t num_rays_total, const float* starts,
                           const float* directions, const int64_t* ray_index, int num_rays,
                           int count, const float* unit, const float* noise, float anneal,
                           float* t_out, float* positions, float* views, void* stream);

/* K2c  per-ray CDF from probe opacities (ray_sampler.py:59-67, _determine_cdf).
 *   t_probe (P,n), opacity (P,n) -> cdf (P,n-1) */
int ffn_cdf_build(const float* t_probe, const float* opacity, int64_t num_rays, int n,
                  float* cdf, void* stream);

/* K2c on the coarse model's raw outputs: logits (P,n,4); sigma = softplus(logits[...,3])
 * (ray_sampler.py:261-265) is applied inside.  Used by the live focus sampler, which builds
 * CDF rows per batch instead of a per-sampler table. */
int ffn_cdf_build_logits(const float* t_probe, const float* logits, int64_t num_rays, int n,
                         float* cdf, void* stream);

/* K2d  inverse-transform focus samples + merge with the uniform half + sort
 * (ray_sampler.py:301-357 and :388-392).
 *   cdfs (num_rays_total, n_focus-1) indexed by global ray id
 *   u    (R,n_focus) uniform block (torch.rand, or linspace(0,1,n) repeated)
 *   t_io (R,S): on entry the first S-n_focus entries of each row hold the uniform
 *        samples (from ffn_sample_t with t_stride=S); on exit the row holds all S
 *        samples sorted ascending.  S <= 256. */
int ffn_focus_sample_merge(const float* near_far, int64_t num_rays_total,
                           const float* cdfs, const int64_t* ray_index, const float* u,
                           const float* unit_focus, int num_rays, int num_samples,
                           int n_focus, float* t_io, void* stream);

/* K2d with one CDF row per BATCH ray: cdf_rows (R, n_focus-1), row r belongs to
 * ray_index[r].  Everything else as ffn_focus_sample_merge. */
int ffn_focus_sample_merge_rows(const float* near_far, int64_t num_rays_total,
                                const float* cdf_rows, const int64_t* ray_index, const float* u,
                                const float* unit_focus, int num_rays, int num_samples,
                                int n_focus, float* t_io, void* stream);

/* ------------------------------------------------------------------------------------
 * K3  standalone Fourier feature encoding.
 * Replaces fourier_feature_models.py:59-68 (scale = pi, optional a_values) and
 * nerf_model.py:97-102 (scale = 1, include_input appends x).  Output is (N, 2F[+3]),
 * cos block first.  b is (3,F); a is (F) or NULL.  F == 0 copies x (class MLP).
 * `out` must be 16-byte aligned (rows are assembled in LDS and leave as whole float4 lines);
 * F <= 2046.
 */
int ffn_fourier_encode(const float* x, int64_t n, const float* b, const float* a,
                       int num_freq, float scale, int include_input, float* out,
                       void* stream);

/* ------------------------------------------------------------------------------------
 * K5  activations + front-to-back alpha compositing, one wavefront per ray.
 * Replaces Raycaster.render after the model call (ray_caster.py:66-93) and
 * utils.calculate_blend_weights (utils.py:72-97).
 *   logits (R,S,4) raw [r,g,b,sigma]; t (R,S)
 *   color (R,3), alpha (R), depth (R) or NULL
 *   nan_flag: int32 on the device, OR-ed with 1 if a NaN is seen in sigmoid(rgb) or
 *             softplus(sigma) (the asserts at ray_caster.py:73-74); may be NULL
 */
int ffn_composite_fwd(const float* logits, const float* t, int num_rays, int num_samples,
                      float* color, float* alpha, float* depth, int32_t* nan_flag,
                      void* stream);

/* K5w  utils.calculate_blend_weights (utils.py:72-97) on its own: t (R,S), sigma (R,S)
 * already activated -> weights (R,S). */
int ffn_blend_weights(const float* t, const float* sigma, int num_rays, int num_samples,
                      float* weights, void* stream);

/* K5w backward: the autograd of utils.py:72-97 in closed form.  d_weights (R,S) ->
 * d_sigma (R,S) and, when d_t is not NULL, d_t (R,S).  S <= 256. */
int ffn_blend_weights_bwd(const float* t, const float* sigma, const float* d_weights,
                          int num_rays, int num_samples, float* d_sigma, float* d_t,
                          void* stream);

/* K5b backward of K5: d(loss)/d(logits) from d/d(color) (R,3) and d/d(alpha) (R). */
int ffn_composite_bwd(const float* logits, const float* t, const float* d_color,
                      const float* d_alpha, int num_rays, int num_samples,
                      float* d_logits, void* stream);

/* ------------------------------------------------------------------------------------
 * K6  ground-truth gather + MSE loss + its gradient.
 * Replaces ImageDataset.render / .loss (image_dataset.py:224-262).
 *   gt_colors (num_rays_total,3), gt_alphas (num_rays_total) or NULL
 *   sums      (2) device floats: sum((c-c_gt)^2), sum((a-a_gt)^2) over this call
 *   d_color = color_scale * 2 (c - c_gt), d_alpha = alpha_scale * 2 (a - a_gt);
 *   the caller picks color_scale = 1/(3R) and alpha_scale = alpha_weight/R (or the
 *   global counts under data parallelism).  d_color/d_alpha may be NULL (loss only).
 *   scratch   (2*ceil(R/256)) floats of workspace
 */
int ffn_mse_loss(const float* color, const float* alpha, const float* gt_colors,
                 const float* gt_alphas, const int64_t* ray_index, int num_rays,
                 float color_scale, float alpha_scale, float* sums, float* d_color,
                 float* d_alpha, float* scratch, void* stream);

/* ------------------------------------------------------------------------------------
 * K7  clip_grad_value_ -> clip_grad_norm_ -> Adam(L2 weight decay) on one flat buffer.
 * Replaces ray_caster.py:327-329 (+ torch.optim.Adam's single-tensor update).
 *   scratch (ceil(n/1024)) floats; step_size = lr/(1-beta1^t), inv_sqrt_bc2 =
 *   1/sqrt(1-beta2^t) are computed by the host in double precision.
 */
int ffn_clip_adam(float* params, float* grads, float* exp_avg, float* exp_avg_sq,
                  int64_t n, float clip_value, float max_norm, float step_size,
                  float inv_sqrt_bc2, float beta1, float beta2, float eps,
                  float weight_decay, float* scratch, float* grad_norm_out, void* stream);

/* K5t  The training step's K5 + K6 + K5b in one launch: Raycaster.render (ray_caster.py:66-93),
 * ImageDataset.render / .loss (image_dataset.py:224-262) and their autograd for one batch.
 *   logits (R,S,4), t (R,S); gt_colors / gt_alphas (or NULL) / ray_index / the two scales as in
 *   ffn_mse_loss; d_logits (R,S,4) out -- bit-identical to ffn_composite_fwd -> ffn_mse_loss ->
 *   ffn_composite_bwd; partials: 2 * ffn_composite_train_blocks(R) floats out, one
 *   (sum((c-c_gt)^2), sum((a-a_gt)^2)) pair per workgroup, for ffn_loss_from_partials.
 *   nan_flag as in ffn_composite_fwd (may be NULL).  S <= 512, R >= 1. */
int ffn_composite_train_blocks(int num_rays);
int ffn_composite_train(const float* logits, const float* t, int num_rays, int num_samples,
                        const float* gt_colors, const float* gt_alphas, const int64_t* ray_index,
                        float color_scale, float alpha_scale, float* d_logits, float* partials,
                        int32_t* nan_flag, void* stream);

/* Fixed-order sum of K5t's partials into sums (2 floats; may be NULL) and / or the scalar loss
 *   loss = sums[0] / colour_count + alpha_weight * (sums[1] / alpha_count)
 * into loss_out (may be NULL) (image_dataset.py:237-242). */
int ffn_loss_from_partials(const float* partials, int num_blocks, float colour_count,
                           float alpha_count, float alpha_weight, float* sums, float* loss_out,
                           void* stream);

/* The scalar loss from K6's two sums (possibly all-reduced over the ranks in between):
 *   loss = sums[0] / colour_count + alpha_weight * (sums[1] / alpha_count)
 * (image_dataset.py:237-242: colour_count = 3 * rays, alpha_count = rays); one launch instead
 * of four scalar tensor ops per optimisation step. */
int ffn_loss_value(const float* sums, float colour_count, float alpha_count, float alpha_weight,
                   float* loss_out, void* stream);

/* ------------------------------------------------------------------------------------
 * K8  image assembly (ray_sampler.py:191-196): zeros, scatter, (x*255) truncated to u8.
 *   colors (n,3); pixel_index (n) int64 pixel id inside the frame; image (H*W*3) u8
 */
int ffn_to_image(const float* colors, const int64_t* pixel_index, int64_t n, int width,
                 int height, uint8_t* image, void* stream);

/* ------------------------------------------------------------------------------------
 * K8b  8-bit YCrCb -> RGB, in place on a (pixels,3) u8 frame.  Replaces
 * cv2.cvtColor(pixels, cv2.COLOR_YCrCB2RGB) at ray_sampler.py:197-198 and
 * ray_dataset.py:180-181 (color_space == "YCrCb").  OpenCV's 8-bit fixed-point path
 * (coefficients x 2^14, rounded shift, saturating cast), restated from its documented
 * constants -- cv2 is not available where this library is built: parity unpinned.
 */
int ffn_ycrcb_to_rgb_u8(uint8_t* image, int64_t pixels, void* stream);

/* ------------------------------------------------------------------------------------
 * K4  fused Fourier-feature MLP (sigma + view-dependent RGB), exact-f32 MFMA.
 * Replaces FourierFeatureMLP.forward (fourier_feature_models.py:57-78), NeRF.forward
 * (nerf_model.py:86-124) and their autograd backward.
 *
 * The network is described to the kernels as a short chain of dense steps
 * (ffn_mlp_chain); the same interpreter runs the forward chain and the backward-data
 * chain.  A wavefront owns 32 consecutive samples; activations never leave the CU
 * between layers.  Weights are consumed in an MFMA-operand order produced by
 * ffn_mlp_pack from the natural nn.Linear (out,in) tensors.
 */
#define FFN_MAX_STEPS 16

typedef struct ffn_encoding {
    const float* b;        /* (3,max(F,1)) frequency matrix (device), never NULL      */
    const float* a;        /* (max(F,1)) amplitudes (device), never NULL              */
    int32_t num_freq;      /* F; 0 => features are the raw 3 inputs                  */
    int32_t include_input; /* append x after [cos,sin]                               */
    float scale;           /* pi (FourierFeatureMLP) or 1 (NeRF)                     */
    int32_t width;         /* internal feature count padded to a multiple of 32     */
} ffn_encoding;

/* One dense step.  Its K dimension is [act_groups groups of 8 channels read back from the
 * wave's activation slab (the previous step's output)] followed by [aux_groups generated
 * groups]: Fourier features of encoding enc_id in a forward chain (first layer = features
 * only, hidden layer = activations only, NeRF's skip and view layers = both,
 * nerf_model.py:112-121), or the d_logits columns [lg_col, lg_col+lg_n) in a backward
 * chain (the sigma / rgb heads; 4 groups, only the first is non-zero).  Both counts are
 * multiples of 4.                                                                      */
typedef struct ffn_step {
    int32_t act_groups;
    int32_t aux_groups;
    int32_t enc_id;        /* forward: 0 = position encoding, 1 = view encoding        */
    int32_t lg_col;        /* backward: first d_logits column feeding this step        */
    int32_t lg_n;
    int32_t out_tiles;     /* ceil(out/32): 8, 4, 2 or 1 (wide chains: 16, 8, 4 or 2)  */
    int32_t relu;          /* forward: ReLU on the output                              */
    int32_t dst;           /* forward: 0 = activation slab, 1 = logits [out_col,+out_n)*/
    int32_t out_col;
    int32_t out_n;
    int32_t save_in_slot;  /* slab that receives the act-part INPUT while it is being
                              consumed (forward: H for backward; backward: dZ), or -1.
                              A TRAINING forward chain of <= 256 channels sets it on every
                              step with act_groups > 0 (and save_enc_slot on every step
                              with aux_groups > 0): its K-loop trips save unconditionally */
    int32_t save_out_slot; /* slab that receives the step's output (backward: dZ of the
                              last step; forward: input of a fused head), or -1        */
    int32_t mask_slot;     /* ReLU sign-mask slot: a forward step writes the sign bits of
                              its output there, the backward step that differentiates that
                              layer reads them; -1 = no ReLU                           */
    int32_t save_enc_slot; /* forward: slab that receives the generated encoding features
                              of this step (so the weight gradients read them back instead
                              of regenerating them), or -1                             */
    int32_t head_off;      /* forward: >= 0 fuses a logits head into this step's epilogue:
                              float offset (inside the bias buffer) of 4 bias floats then
                              channels*4 weights [channel][logits column], zero in the
                              columns the head does not write; the step's output is then
                              also stored into slab save_out_slot when training; -1 none */
    int32_t out_slot;      /* split-bf16 kernels only (the f32 kernels ignore it): the slab of the
                              step's output -- its activations in a forward chain, its dZ in a
                              backward chain -- which those kernels save from registers in
                              every step's epilogue; -1 = none                            */
    int64_t w_off;         /* float offset of this step's packed operand weights       */
    int64_t b_off;         /* forward: float offset of the bias (padded to 32*tiles)   */
} ffn_step;

typedef struct ffn_mlp_chain {
    ffn_encoding enc[2];
    ffn_step step[FFN_MAX_STEPS];
    int32_t num_steps;
    int32_t num_slots;                     /* hidden-layer slabs (= sign-mask slots);
                                              entries num_slots.. of the two arrays below
                                              describe the encoding-feature slabs        */
    int32_t bias_floats;                   /* floats in the bias buffer: [fused-head blocks |
                                              per-step padded biases].  The kernels stage its
                                              first 4096 floats in LDS; every head block must lie
                                              inside that copy, a step bias beyond it is read
                                              from global memory                              */
    int32_t wide;                          /* waves per 32-sample block of the exact-f32 chain
                                              kernels.  0: one.  1: TWO waves share a block and a
                                              64 KiB slab (required when some layer is wider than
                                              256 channels: out_tiles may be 16 and must be even,
                                              act_groups <= 64, heads must be fused); the mask
                                              buffer holds 512 uint32 per slot and block.  2: FOUR
                                              waves share a block (one team per workgroup; narrow
                                              chains whose steps have 4 or 8 output tiles, heads
                                              fused; training / backward launches only): 1024
                                              uint32 of masks per slot and block.  The host runs
                                              the short last round of a training launch on the
                                              team kernels (same packs, same slabs)            */
    int32_t slot_channels[FFN_MAX_STEPS];  /* channels of each slab (multiple of 32)  */
    int64_t slot_offset[FFN_MAX_STEPS];    /* sum of channels of the slabs before it  */
} ffn_mlp_chain;

/* Slab ("block") layout, shared by forward, dgrad and wgrad: samples are grouped in blocks
 * of 32; slab `slot` starts at float offset slot_offset[slot]*num_blocks*32 and holds,
 * per block, channels*32 floats as float4s indexed [cq][pos], cq = channel/4,
 * pos = sample_in_block ^ (cq & 15), each float4 = 4 consecutive channels of one sample.
 * The XOR spreads the weight-gradient kernel's transposed reads over distinct sectors. */

/* Gather a natural (rows, cols) matrix into MFMA A-operand order:
 *   dst[((g*tiles + o)*64 + lane)*4 + p] = src[row(32*o + (lane&31)) * ld + col(8*g + 4*(lane>>5) + p)]
 * row_map / col_map (int32, device, may be NULL = identity) translate internal to
 * natural indices; a negative entry or an index outside [0,rows)x[0,cols) yields 0.
 * transpose != 0 swaps the roles (operand rows index src columns).                    */
int ffn_mlp_pack(const float* src, int rows, int cols, int ld, int transpose,
                 const int32_t* row_map, const int32_t* col_map, int groups, int tiles,
                 float* dst, void* stream);

/* The same gather for a whole model in one launch: a device array of jobs, one per operand
 * pack (kind 0: ffn_mlp_pack's formula with row_map = identity) or per bias / fused-head block
 * (kind 1: the strided copy dst[c*dst_cs + r*dst_rs] = src[r*ld + c], r < rows, c < cols).
 * The optimiser step rewrites the weights in place, so a training loop re-packs after every
 * step; this keeps that at one launch. */
typedef struct {
    const float* src;
    float* dst;
    const int32_t* col_map;   /* kind 0: internal K index -> natural column, or NULL */
    int32_t kind, rows, cols, ld, transpose, groups, tiles, dst_rs, dst_cs, reserved;
} ffn_pack_job;
int ffn_mlp_pack_jobs(const ffn_pack_job* jobs, int num_jobs, void* stream);

/* Forward chain.  positions (N,3), views (N,3) or NULL, logits out (N,4).  Training: when
 * `saved` and `masks` are non-NULL, every step writes what the backward pass needs into
 * `saved` (block layout): its input activations (save_in_slot), the encoding features it
 * generated (save_enc_slot), its output when a fused head reads it (save_out_slot); and every
 * ReLU step writes the sign bits of its output into `masks` (num_slots * num_blocks * W
 * uint32, W = 256, or 512 for a wide chain: [slot][block][half][lane][4], bit 31-(16*(tile&1)+r)
 * of word tile/2 = accumulator register r of that lane, tiles counted inside the wave's half;
 * an odd tile count leaves the last word's bits shifted down by 16)
 * for the backward-data chain.  `bias` = bias_floats floats: per step its padded bias, plus
 * the fused heads' blocks (head_off).
 *
 * A launch may cover a SUB-RANGE of a batch's 32-sample blocks (the host splits a batch whose
 * block count leaves a short last round for the persistent grid, and runs that tail on the
 * two-waves-per-block kernels): positions / views / logits / masks are then the sub-range's own
 * arrays (n samples), while the slabs in `saved` are addressed by the batch's block ids --
 * slab_block0 = the launch's first block, slab_blocks = the batch's block count (the slot stride).
 * slab_blocks == 0: the launch is the whole batch. */
int ffn_mlp_forward(const ffn_mlp_chain* chain, const float* packed_w, const float* bias,
                    const float* positions, const float* views, int64_t n, float* logits,
                    float* saved, uint32_t* masks, int64_t slab_block0, int64_t slab_blocks,
                    void* stream);

/* ------------------------------------------------------------------------------------
 * Fused inference render: K2 + K3 + K4 + K5 in one launch.
 * Replaces the body of Raycaster.render_image / batched_render (ray_caster.py:103-159:
 * sampler.sample -> model -> activations -> calculate_blend_weights -> sums, each a pass over
 * HBM-resident (R,S,3)/(R,S,4) arrays) for an eval-mode model: per ray, the t-samples, the
 * positions, the features, every hidden activation and the logits stay on the CU; only the
 * ray id + ray state are read and the composited pixel is written.
 *   rays:      resident sampler state (K1's outputs) + the ids of the rays to render.
 *              t_values == NULL: t_j = near + unit[j] * (far - near) (non-stratified uniform
 *              sampling, ray_sampler.py:380-381, rounded like ffn_sample_t); otherwise the
 *              caller's (R,S) t-values (stratified / opacity-guided samplers).
 *   occupancy: NULL or an occupancy grid (K9): samples in empty cells are not evaluated
 *              (sigma = 0 there: weight 0, transmittance factor 1) -- PSNR-level parity with
 *              the full render, exactly the K9 semantics.
 *   out:       any of color (R,3), alpha (R), depth (R) may be NULL; image != NULL also
 *              writes the u8 pixel (x*255 truncated, ray_sampler.py:193-196) at pixel
 *              ray_id - pixel_offset of an (H*W,3) frame the caller has zeroed.
 * Any forward chain (a wide one runs a pair of wavefronts per ray); num_samples <= 256.
 */
typedef struct ffn_render_rays {
    const float* starts;       /* (num_rays_total,3)                                   */
    const float* directions;   /* (num_rays_total,3), also the view directions          */
    const float* near_far;     /* (2,num_rays_total)                                   */
    int64_t num_rays_total;
    const int64_t* ray_index;  /* (num_rays) int64 ray ids, or NULL: ids ray_base + 0..R-1   */
    int64_t ray_base;
    const uint8_t* valid;      /* NULL, or K1's (num_rays_total) mask: rays with valid == 0
                                  are not traced (zeros in color/alpha/depth, no pixel):
                                  a whole camera renders without a filtered index list  */
    int32_t num_rays;
    int32_t num_samples;
    const float* unit;         /* (num_samples) torch.linspace(0,1,S), or NULL with t_values */
    const float* t_values;     /* (num_rays,num_samples) or NULL                       */
} ffn_render_rays;

typedef struct ffn_occupancy {
    const uint32_t* bits;      /* resolution^3 bits (ffn_occupancy_build)               */
    float box_min[3];
    float box_size[3];
    int32_t resolution;
} ffn_occupancy;

typedef struct ffn_render_out {
    float* color;
    float* alpha;
    float* depth;
    int32_t* nan_flag;         /* as ffn_composite_fwd; may be NULL                    */
    uint8_t* image;
    int64_t pixel_offset;
} ffn_render_out;

int ffn_render_fused_fwd(const ffn_mlp_chain* chain, const float* packed_w, const float* bias,
                         const ffn_render_rays* rays, const ffn_occupancy* occupancy,
                         const ffn_render_out* out, void* stream);

/* Fused live focus sampling: the coarse pass of ray_sampler.py:234-269 (probe the opacity model
 * on n_focus <= 64 points t = linspace(near, far, n_focus) per ray, sigma = softplus of its last
 * output), the CDF of :59-67 and the inverse-transform sampling + merge + sort of :301-357 /
 * :388-392 in ONE launch, per batch ray, with no per-sampler table.  `chain` describes the
 * opacity model (a narrow forward chain; view directions = the ray directions when it takes
 * them).  t_io (R,S): on entry the first S-n_focus entries of each row hold the uniform samples
 * (ffn_sample_t with t_stride = S); on exit all S samples sorted.  u (R,n_focus) as
 * ffn_focus_sample_merge.  Arithmetic identical to ffn_sample_t + ffn_cdf_build_logits +
 * ffn_focus_sample_merge_rows. */
int ffn_focus_fused(const ffn_mlp_chain* chain, const float* packed_w, const float* bias,
                    const float* starts, const float* directions, const float* near_far,
                    int64_t num_rays_total, const int64_t* ray_index, int num_rays,
                    int num_samples, int n_focus, const float* unit_focus, const float* u,
                    float* t_io, void* stream);

/* ------------------------------------------------------------------------------------
 * OPT-IN split-bf16 inference mode of K4 (separately labelled; the exact-f32 entry points above
 * are the parity mode).  Every f32 operand is split into two bf16 parts and every product into
 * three v_mfma_f32_32x32x16_bf16 instructions with f32 accumulation (~2^-16 relative error per
 * product).  The chain is an ffn_mlp_chain whose w_off are offsets (in bf16 elements) into the
 * operand buffer written by ffn_mlp_pack_bf16; bias / fused-head blocks are the f32 buffer of
 * ffn_mlp_forward.  Narrow chains (<= 256 channels), slab-destination steps with fused heads.
 *
 * ffn_mlp_pack_bf16: dst[(((G*tiles + o)*2 + part)*64 + lane)*8 + j] =
 *   part(src[32*o + (lane & 31)][col_map[16*G + 8*(lane >> 5) + j]]), part 0 = bf16(v) rounded to
 *   nearest even, part 1 = bf16(v - part 0); col_map (int32, device, 16*kblocks entries) maps the
 *   operand K order to natural columns (-1 = zero): for activation K blocks the hand-off order
 *   16G + {4h+j | 8+4h+(j-4)}, for encoding K blocks the internal feature order 16G + 8h + j.
 *   transpose = 1 packs src^T (backward data): tile rows walk src's first `cols` columns and
 *   col_map maps K to src's rows. */
int ffn_mlp_pack_bf16(const float* src, int rows, int cols, int ld, const int32_t* col_map,
                      int kblocks, int tiles, int transpose, uint16_t* dst, void* stream);
int ffn_mlp_forward_bf16x3(const ffn_mlp_chain* chain, const uint16_t* packed_w, const float* bias,
                           const float* positions, const float* views, int64_t n, float* logits,
                           void* stream);
/* The same forward pass, leaving what the backward kernels need in the formats of
 * ffn_mlp_forward's training mode (`saved` activation slabs, `masks`): every step saves the
 * encoding features it generates (save_enc_slot), its output (out_slot) and its ReLU sign mask
 * (mask_slot).  OPT-IN ("bf16x3" training precision): the saved values carry the ~1e-6 relative
 * error of the split products. */
int ffn_mlp_forward_bf16x3_train(const ffn_mlp_chain* chain, const uint16_t* packed_w,
                                 const float* bias, const float* positions, const float* views,
                                 int64_t n, float* logits, float* saved, uint32_t* masks,
                                 void* stream);

/* Split-bf16 backward-data chain (OPT-IN "bf16x3" training precision): the chain of
 * ffn_mlp_backward_data with w_off pointing into transposed ffn_mlp_pack_bf16 operands (hidden
 * consumer's K blocks, then -- if the producer feeds a logits head -- two K blocks whose K rows
 * 0..lg_n-1 are the head's rows) and step.out_slot = the slab slot of the step's dZ.
 * Reads the sign masks, writes every dZ slab in the f32 kernels' format. */
int ffn_mlp_backward_data_bf16x3(const ffn_mlp_chain* chain, const uint16_t* packed_wt,
                                 const float* d_logits, int64_t n, const uint32_t* masks,
                                 float* dz, void* stream);

/* ------------------------------------------------------------------------------------
 * OPT-IN f32-ACCURATE split mode of K4 ("bf16x6"; separately labelled, the exact-f32 entry points
 * stay the parity mode and the headline).  Every f32 operand is THREE bf16 parts (hi, mid, lo:
 * 8 + 8 + 8 significand bits = the f32 value exactly) and every f32 product SIX
 * v_mfma_f32_32x32x16_bf16 instructions -- every partial product down to 2^-16 of the leading
 * one (h.l, l.h, m.m, h.m, m.h, h.h, smallest first; the three dropped terms are each below 2^-24
 * of it), f32 accumulation: 12 matrix cycles per K where v_mfma_f32_32x32x2_f32 takes 32, at the
 * error of an f32 dot product (same networks, same tolerances as the exact mode in the tests).
 * Same organisation, chains, slab and mask formats as the bf16x3 entry points (csrc/mlp_bf16_ws.hip:
 * eight waves, one output tile each, two blocks of 32 samples per pass); operands come from
 * ffn_mlp_pack_bf16_parts(parts = 3): dst[(((G*tiles + o)*parts + part)*64 + lane)*8 + j], part p =
 * bf16(v - part 0 - .. - part p-1) (parts = 2 is ffn_mlp_pack_bf16), so every w_off of the bf16x3
 * chain scales by 3/2.  Narrow chains (<= 256 channels per layer) with fused heads; encoding
 * features are the exact-f32 kernels' (polynomial sin / cos, bit for bit).  The weight gradients of
 * this mode are TWO launches over one partial buffer and one reducer table: full units (four
 * 128 x 128 quadrants) and the logits-head units on ffn_mlp_wgrad_units_bf16x6 (three-part operands,
 * below), units with fewer quadrants on the exact-f32 ffn_mlp_wgrad_units, which folds narrow input
 * windows -- both read the slabs these kernels write (host: MlpProgram._plan_wgrad;
 * FFN_BF16X6_WGRAD=f32 keeps every unit on the exact-f32 kernel).
 * FFN_BF16X6_PRODUCTS=9 (environment, measurement only) multiplies out all nine partial products.
 * Two workgroup organisations sit behind these entry points (same packs, same slab / mask / dZ formats;
 * the forward's slabs and masks bit-identical, the backward's dZ within 2e-7 of its largest element: the
 * head term of its step 0 is f32 vector arithmetic in one, six bf16 products in the other): chains whose
 * first step is features-only (16 j K blocks) and whose
 * other steps are 256 -> 256 -- the tiny NeRF / Fourier MLP family -- run the MATRIX WAVES / VECTOR WAVES
 * kernels (csrc/mlp_bf16_mv.hip: four waves that only multiply, eight that only generate features and
 * run epilogues), everything else the two-waves-per-SIMD kernels (csrc/mlp_bf16_ws.hip);
 * FFN_BF16X6_ORG=ws (environment, read per launch) keeps the latter for every chain (A/B).
 * ffn_mlp_bf16x6_organisation answers which one a launch of `chain` takes: 1 = matrix / vector waves,
 * 0 = two waves per SIMD (backward != 0: the backward-data chain).
 * Reference arithmetic being matched: fourier_feature_models.py:57-78, nerf_model.py:86-124. */
int ffn_mlp_bf16x6_organisation(const ffn_mlp_chain* chain, int backward);
int ffn_mlp_pack_bf16_parts(const float* src, int rows, int cols, int ld, const int32_t* col_map,
                            int kblocks, int tiles, int transpose, int parts, uint16_t* dst,
                            void* stream);
int ffn_mlp_forward_bf16x6(const ffn_mlp_chain* chain, const uint16_t* packed_w, const float* bias,
                           const float* positions, const float* views, int64_t n, float* logits,
                           void* stream);
int ffn_mlp_forward_bf16x6_train(const ffn_mlp_chain* chain, const uint16_t* packed_w,
                                 const float* bias, const float* positions, const float* views,
                                 int64_t n, float* logits, float* saved, uint32_t* masks,
                                 void* stream);
int ffn_mlp_backward_data_bf16x6(const ffn_mlp_chain* chain, const uint16_t* packed_wt,
                                 const float* d_logits, int64_t n, const uint32_t* masks,
                                 float* dz, void* stream);

/* Backward-data chain: d_logits (N,4) + ReLU sign masks -> dZ slabs (same slab geometry as
 * `saved`).  packed_wt holds the transposed operand packs. */
int ffn_mlp_backward_data(const ffn_mlp_chain* chain, const float* packed_wt,
                          const float* d_logits, int64_t n, uint32_t* masks, float* dz,
                          int64_t slab_block0, int64_t slab_blocks, void* stream);

/* Weight gradients  dW_l = sum over samples of dZ_l (x) X_l  (autograd of the nn.Linear
 * layers, fourier_feature_models.py:70-78 / nerf_model.py:103-124).  Every operand is a
 * slab: dZ from ffn_mlp_backward_data, X = hidden activations and encoding features saved by
 * ffn_mlp_forward.  A unit is a (<=256 output channels) x (<=256 input channels) block of
 * some dW; segments assign contiguous ranges of 32-sample blocks of a unit to persistent
 * workgroups: workgroup g (256 threads) processes segments seg_start[g] .. seg_start[g+1]
 * (segment.job indexes `units`).  Its four waves own the 128x128 quadrants of the unit that
 * exist (quadrant id = 2*m_half + n_half when both sides are wider than 128 channels,
 * otherwise the half index of the wide side, otherwise 0); when fewer than four exist, wave
 * w takes quadrant w % Q and the (w / Q)-th share of every block's samples.  Each wave
 * writes one partial of ffn_mlp_wgrad_partial_floats() floats into slot segment.slot + w.
 * Segments may be planned for more blocks than ceil(n/32): the kernel clamps blk_end and
 * writes zero partials for segments that lie entirely past the end. */
typedef struct ffn_wgrad_segment {
    int32_t job;
    int32_t slot;
    int64_t blk_begin;
    int64_t blk_end;
} ffn_wgrad_segment;

typedef struct ffn_wgrad_unit {
    int32_t m_slot;    /* dZ slab                        (head unit: first d_logits column) */
    int32_t m_cq0;     /* first channel quad of the output window (head unit: #columns)    */
    int32_t m_quads;   /* valid quads (<= 64, multiple of 8)                           */
    int32_t want_bias; /* != 0: also produce the bias gradient (sum of dZ over samples)
                          in the bias strip of the n-half-0 partials               */
    int32_t n_slot;    /* slab index of the input window                               */
    int32_t n_cq0;     /* first channel quad of the window                             */
    int32_t n_quads;   /* valid quads (<= 64, multiple of 8)                           */
    int32_t kind;      /* 0 = dW block; 1 = logits-head rows: wave w owns channel quads
                          16 w .. 16 w + 15 of the window, partial slot segment.slot + w     */
} ffn_wgrad_unit;

int ffn_mlp_wgrad_units(const ffn_mlp_chain* chain, const ffn_wgrad_unit* units,
                        const ffn_wgrad_segment* segments, const int32_t* seg_start,
                        int num_groups, const float* saved, const float* dz,
                        const float* d_logits, int64_t n, float* partials, void* stream);
/* The same launch with every f32 product as three bf16 matrix products (OPT-IN "bf16x3" training
 * precision): same units, segments, slabs and partial format; ffn_mlp_wgrad_reduce is shared.
 * (Slabs are staged by LDS-DMA in half-block stages, operands converted one contraction step
 * ahead of the matrix instructions: csrc/wgrad_bf16.hip.) */
int ffn_mlp_wgrad_units_bf16x3(const ffn_mlp_chain* chain, const ffn_wgrad_unit* units,
                               const ffn_wgrad_segment* segments, const int32_t* seg_start,
                               int num_groups, const float* saved, const float* dz,
                               const float* d_logits, int64_t n, float* partials, void* stream);
/* Weight-gradient units of the f32-accurate "bf16x6" mode (csrc/wgrad_bf16x6.hip): ffn_mlp_wgrad_units with every f32
 * product as six bf16 matrix products on three-part operands; units, segments, partial format and
 * the reducer are shared with the exact-f32 and the bf16x3 kernels (the logits-head unit stays
 * exact f32).  The dZ window is double buffered; the input window is split just in time, hi parts
 * first, and the products are ordered by the part of it they need. */
int ffn_mlp_wgrad_units_bf16x6(const ffn_mlp_chain* chain, const ffn_wgrad_unit* units,
                               const ffn_wgrad_segment* segments, const int32_t* seg_start,
                               int num_groups, const float* saved, const float* dz,
                               const float* d_logits, int64_t n, float* partials, void* stream);

/* Fixed-order reduction of the partials of each job into the flat natural-layout
 * gradient buffer (nn.Linear weight (out,in) row-major, then bias). */
typedef struct ffn_reduce_job {
    int32_t kind;
    int32_t slot_begin, slot_end;  /* partial slots slot_begin, +stride, ... < slot_end      */
    int32_t m_ch0;                 /* kind 0: first output channel of the patch        */
    int32_t rows;                  /* output rows of the layer                         */
    int32_t n_quad0;               /* first input quad of the panel                    */
    int32_t n_quads;
    int32_t k_base;                /* index of the panel's first quad in col_map space */
    int32_t ld;                    /* leading dimension (= in_features) of dW          */
    int32_t has_bias;              /* this job also carries the bias gradient          */
    int32_t lg_n;
    int32_t slot_stride;           /* distance between consecutive slots of this job   */
    int32_t n_fold;                /* kind 0: 1, or the fold of a narrow input window: the
                                      exact-f32 unit kernel packs a window of <= 16 / <= 8
                                      quads into 2 / 1 column tiles (n_fold 2 / 4; column j of
                                      tile q = quad j % (32/n_fold), component
                                      (j / (32/n_fold)) * (4/n_fold) + q); the split-bf16 unit
                                      kernel does not fold (n_fold 1)                       */
    int32_t reserved;
    int64_t w_grad_off;            /* float offset of dW inside `grads`                */
    int64_t b_grad_off;            /* float offset of db inside `grads`                */
    const int32_t* col_map;        /* internal K index -> natural column, or -1        */
} ffn_reduce_job;

int ffn_mlp_wgrad_reduce(const ffn_reduce_job* jobs, int num_jobs, const float* partials,
                         float* grads, void* stream);

int64_t ffn_mlp_wgrad_partial_floats(void);

/* ---- K9: empty-space skipping for inference (opt-in).  The reference has no
 * counterpart on its render path (its octree ray walker, octree.py:418-501, serves only the
 * lecture visualisations), so parity here is PSNR-level, not sample-level.
 *
 * An occupancy grid is resolution^3 bits (cell (ix,iy,iz) = bit ((iz*G + iy)*G + ix), x fastest)
 * over the box [box_min, box_min + box_size); box_min / box_size are HOST pointers to 3 floats.
 *
 * ffn_occupancy_build: logits (G^3,4) = the model evaluated at the cell centres; a cell is
 *   occupied when softplus(sigma logit) > sigma_threshold; dilate != 0 also marks the 26
 *   neighbours (needs scratch_bits of the same size).
 * ffn_occupancy_count: block_offsets (ceil(n/256) int32) <- exclusive scan of the number of
 *   occupied samples per 256-sample block; *total (device int64) <- their sum.  A sample outside
 *   the box takes the occupancy of the nearest cell (indices clamped); NaN positions count as
 *   occupied.
 * ffn_occupancy_compact: packs the occupied samples (order preserved): out_positions /
 *   out_views (total,3), out_index (total) = their position in the input.
 * ffn_scatter_logits: out (n,4) <- (0,0,0,empty_sigma_logit) everywhere, then
 *   out[index[i]] = packed[i]. */
int ffn_occupancy_build(const float* logits, int resolution, float sigma_threshold, int dilate,
                        uint32_t* scratch_bits, uint32_t* bits, void* stream);
int ffn_occupancy_count(const float* positions, int64_t n, const float* box_min,
                        const float* box_size, int resolution, const uint32_t* bits,
                        int32_t* block_offsets, int64_t* total, void* stream);
int ffn_occupancy_compact(const float* positions, const float* views, int64_t n,
                          const float* box_min, const float* box_size, int resolution,
                          const uint32_t* bits, const int32_t* block_offsets,
                          float* out_positions, float* out_views, int32_t* out_index, void* stream);
int ffn_scatter_logits(const float* packed, const int32_t* index, int64_t m, int64_t n,
                       float empty_sigma_logit, float* out, void* stream);
/* K9g: packed[i] = full[index[i]] for (.,4) rows -- the backward of the scatter (training with
 * empty-space skipping: d_logits of the evaluated samples). */
int ffn_gather_logits(const float* full, const int32_t* index, int64_t m, float* packed,
                      void* stream);

/* ---- K10: dense voxel radiance field (voxels_model.py:35-45): trilinear lookup of a
 * (4,S,S,S) volume at positions/scale in [-1,1]^3 (grid_sample semantics: x = fastest axis,
 * border padding, align_corners = false) + bias (4) -> logits (N,4).  Inference only; it is
 * the opacity model the README workflows hand to the focus sampler. */
int ffn_voxels_forward(const float* volume, const float* bias, const float* positions, int64_t n,
                       int side, float scale, float* out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* FFN_HIP_H */
