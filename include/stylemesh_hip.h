/*
 * libstylemesh_hip.so - C ABI of the MI355X-native StyleMesh texture-optimisation hot path.
 *
 * The reference (lukasHoel/stylemesh) has no FFI: its hot path is Python calling PyTorch ATen operators.
 * Each entry point below replaces the ATen operator sequence reached from the cited reference lines
 * (paths relative to the reference repository). INTEGRATION.md shows the ctypes binding a maintainer of
 * the reference would add.
 *
 * Conventions
 *  - extern "C", plain pointers and sizes, no torch types. All pointers are DEVICE pointers unless a
 *    parameter is documented as host. Every tensor, sum, loss and optimizer state is fp32 (the reference computes
 *    in fp32). The matrix-core kernels come in two arithmetic flavours with fp32 accumulation: exact fp32 MFMA
 *    (sm_conv3x3*, sm_gram_masked, sm_gram_backward) and
 *    fp16x2-split operands scaled by recorded power-of-two bounds (*_split2: 3 partial products; the default of the
 *    Python host: |err| <= 2^-21 sum|x||w| + 2^-38 (max|x| sum|w| + max|w| sum|x|) elementwise).
 *  - Every call is asynchronous on the caller's hipStream_t (passed as void*), allocates nothing and
 *    keeps no state: the caller owns every buffer. Return value: 0 (hipSuccess) or a hipError_t.
 *  - Feature maps use the "padded planar" layout: [C][plane] floats; a plane holds (H+2) rows of Wp
 *    floats, Wp = sm_fmap_row_stride(W) (a multiple of 4); pixel (y,x) lives at q = (y+1)*Wp + (x+1);
 *    row 0, row H+1, column 0 and columns > W are zero and MUST stay zero (kernels that write a plane
 *    write zeros there). Buffers need SM_FMAP_GUARD floats of readable slack before and after.
 *  - Textures / gradients / Adam moments keep the reference layout [3][H][W] per layer
 *    (model/texture/texture.py:29-32); the layers of a hierarchical texture sit back to back in one arena.
 */
#ifndef STYLEMESH_HIP_H
#define STYLEMESH_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SM_MAX_TEX_LAYERS 8
#define SM_FMAP_GUARD 4096 /* floats of slack required before and after every feature-map buffer */

/* epilogue flags of sm_conv3x3 */
#define SM_EPI_BIAS_RELU 1 /* out = relu(acc + bias[co])                                  (forward)  */
#define SM_EPI_RELU_MASK 2 /* out = gate[co][q] > 0 ? v : 0                               (dgrad)    */
#define SM_EPI_ADD 4       /* v += out[co][q] (value already in the output buffer) before the gate   */
#define SM_EPI_GRAM 16     /* with SM_EPI_RELU_MASK and unpool_code, sm_conv3x3_grouped_split2 only, Cout = 64 / 128: v += the masked
                            * Gram backward of the output layer (sm_conv_problem::gram_*), computed in the epilogue   (dgrad)    */
#define SM_EPI_POOL 8      /* with SM_EPI_BIAS_RELU, sm_conv3x3_grouped_split2 only: store the 2x2 max-pooled map and the
                            * pool's argmax codes (sm_conv_problem::pool_out / pool_code) INSTEAD of out     (forward)  */
#define SM_LIST_QUADS 32   /* (ABI 10; not an epilogue) sm_conv3x3_grouped_split2 with Cout = 64, Cin % 64 == 0 and a tile list
                            * whose every four entries are a QUAD - the same 32 columns of four consecutive rows: q, q + Wp,
                            * q + 2 Wp, q + 3 Wp (sm_cover_problem::quad) - takes the resident-input kernel: the block stages
                            * its (4 + 2) x 34 input positions of 64 channels once. Same results, bit for bit.
                            * PRECONDITIONS (not checked - a list that breaks them gives wrong outputs, not an error;
                            * sm_cover_segments with sm_cover_problem::quad = 1 produces conforming lists): every four
                            * entries really are q, q + Wp, q + 2 Wp, q + 3 Wp of ONE problem (or padding behind a live
                            * first entry); the quad's first row is image row 4 Y; with unpool_code the first column is
                            * even. The kernel reads window rows of 36 floats with 16-byte loads: behind the last channel's
                            * plane it relies on the SM_FMAP_GUARD floats every feature-map buffer must have. (The Python host checks a list
                            * against these on request: ops.check_quad_list, STYLEMESH_VALIDATE_LISTS=1.) */

/* ---- layout helpers (host, pure functions) -------------------------------------------------------- */
int sm_fmap_row_stride(int W);        /* Wp */
int sm_fmap_plane(int H, int W);      /* floats per channel plane (multiple of 64) */
int sm_abi_version(void);
/* sizeof of the problem structs of this header as the library was compiled (which: 0 sm_conv_problem, 1
 * sm_plane_problem, 2 sm_gram_problem, 3 sm_style_problem, 4 sm_gram_bwd_problem, 5 sm_cover_problem, 6 sm_view_masks_desc,
 * 7 sm_view_layer_mask, 8 sm_view_resize, 9 sm_view_lists_desc, 10 sm_view_list, 11 sm_call; anything else: -1) - a binding checks
 * its own struct layouts against it. */
int sm_sizeof_problem(int which);

/* ---- texture: model/texture/texture.py ------------------------------------------------------------ */

/* K1. HierarchicalNeuralTexture.forward / NeuralTexture.forward (texture.py:46-54,96-100):
 * out[c][q] = sum_l bilinear_border_sample(layer_l, grid), align_corners=True, for a [h][w][2] grid in
 * [-1,1]. Layers must already be clamped (normalize(), texture.py:41-44; sm_adam_fused keeps them so).
 * out: padded planar fmap with out_channels >= 3 planes (planes >= 3 are left untouched = zero). */
int sm_tex_sample_fwd(const float* const* layers, const int* layer_w, const int* layer_h, int n_layers,
                      const float* grid, int h, int w, float* out, void* stream);
/* The same for up to 8 images (the UV levels of a view) in ONE launch: grids / hs / ws / outs are HOST arrays of n
 * entries (device grid [h][w][2], sizes, device output planes). */
int sm_tex_sample_fwd_grouped(const float* const* layers, const int* layer_w, const int* layer_h, int n_layers,
                              const float* const* grids, const int* hs, const int* ws, float* const* outs, int n,
                              void* stream);

/* K2. Backward of K1 (ATen grid_sampler_2d_backward + RepeatBackward + the zero-filled dense gradient,
 * texture.py:49-53) fused with the two tensor hooks of model/model.py:195-202,245-251:
 * grad_layer_l[c][tap] += pixel_weight[y][x] * grad_img[c][q] * bilinear_weight(tap)  (atomic).
 * pixel_weight: dense [h][w] or NULL (= 1). Gradients ACCUMULATE into grad_layers. */
int sm_tex_sample_bwd(float* const* grad_layers, const int* layer_w, const int* layer_h, int n_layers,
                      const float* grid, int h, int w, const float* grad_img, const float* pixel_weight,
                      void* stream);

/* Multi-GPU helper (SURVEY.md section 8 e; the reference is single-GPU): sets flags[k] = 1 for every chunk k of
 * 2^chunk_log2 floats of the flat gradient arena (starting at arena_base; grad_layers point into it) that
 * sm_tex_sample_bwd can write for this grid / pixel_weight - a superset of the chunks it does write. flags is NOT
 * cleared here. Evaluated once per view: the gradient exchange then covers only flagged chunks. */
int sm_tex_touch_flags(float* const* grad_layers, const int* layer_w, const int* layer_h, int n_layers,
                       const float* arena_base, const float* grid, int h, int w, const float* pixel_weight,
                       int32_t* flags, int chunk_log2, void* stream);

/* K2 as a sorted gather (stylemesh_amd/csrc/scatter_plan.hip). Which texels the pixels of a view hit, and with which
 * weights, depends only on the view - and the reference's RepeatingSampler (data/abstract_dataset.py:498-512) optimises a
 * view for 20-100 consecutive steps. sm_tex_scatter_plan lists every (pixel, layer, tap) contribution of ALL levels
 * of the view as key = arena offset of the texel (channel 0), value = (pixel, level, layer, tap weight x pixel weight),
 * radix-sorts the list by key (sm_radix_sort_pairs) and lists the runs of equal texels that cross a 64-entry chunk boundary (once per view).
 * sm_tex_scatter_planned then replaces the n_levels sm_tex_sample_bwd calls of a step: the planar image gradients
 * are packed into one float4 per pixel, one lane per sorted entry gathers its pixel, a wave-wide segmented sum adds
 * the runs of equal texels, a run inside a chunk writes its texel, and one thread per crossing run adds that run's
 * pieces in chunk order. No atomics: every texel has one writer, the result is bit-reproducible. Same result as
 * sm_tex_sample_bwd up to fp32 summation order.
 *   n_entries = 4 * n_layers * sum_k(level_h[k] * level_w[k]); key_bits = bits of the arena size + 1 (<= 32; the
 *   all-ones key marks zero-weight entries). Caller-owned DEVICE buffers: keys0/1 [n_entries] u32, vals0/1
 *   [n_entries] u64, temp (sm_tex_scatter_plan_temp_bytes), cross (sm_tex_scatter_plan_cross_bytes: crossing-run
 *   list + per-chunk partial sums; written by the plan, used by every planned call), packed_scratch (4 floats per
 *   position of every level's padded plane: 4 * sum_k sm_fmap_plane(level_h[k], level_w[k])). *sorted_in (HOST)
 *   receives 0 / 1 = which of the two key / value buffer pairs holds the sorted list. grids / pixel_weights /
 *   grad_imgs: HOST arrays of device pointers (pixel_weights or its entries may be NULL). accumulate = 0: the caller
 *   guarantees that the arena is ZERO where this view writes (sm_adam_fused zeroes it): texels are stored without
 *   being read; 1: add into the arena like sm_tex_sample_bwd. */
size_t sm_tex_scatter_plan_temp_bytes(size_t n_entries, int key_bits);
/* The sort of the plan as an entry point of its own (ABI 9; rounds 1-4 called rocPRIM's device radix sort here): STABLE LSD
 * radix sort of n (u32 key, u64 value) pairs by the low key_bits bits of the keys - equal keys keep their input order, so
 * everything downstream adds in a fixed order. Ping-pongs between (keys0, vals0) - the input - and (keys1, vals1);
 * *sorted_in (HOST) = 0 / 1 = which pair holds the result. temp: sm_tex_scatter_plan_temp_bytes(n, key_bits) bytes.
 * ceil(key_bits / 9) passes of three launches (per-tile digit histograms, per-bin scan over the tiles, stable scatter). */
int sm_radix_sort_pairs(uint32_t* keys0, uint32_t* keys1, uint64_t* vals0, uint64_t* vals1, size_t n, int key_bits,
                        void* temp, size_t temp_bytes, int* sorted_in, void* stream);
size_t sm_tex_scatter_plan_cross_bytes(size_t n_entries);
int sm_tex_scatter_plan(float* const* grad_layers, const int* layer_w, const int* layer_h, int n_layers,
                        const float* arena_base, const float* const* grids, const float* const* pixel_weights,
                        const int* level_h, const int* level_w, int n_levels, uint32_t* keys0, uint32_t* keys1,
                        uint64_t* vals0, uint64_t* vals1, void* temp, size_t temp_bytes, void* cross, int key_bits,
                        int* sorted_in, void* stream);
int sm_tex_scatter_planned(const uint32_t* keys, const uint64_t* vals, size_t n_entries, const float* const* grad_imgs,
                           const int* level_h, const int* level_w, int n_levels, float* const* grad_layers,
                           const int* layer_w, const int* layer_h, int n_layers, float* arena_base, int key_bits,
                           float* packed_scratch, void* cross, int accumulate, void* stream);

/* K7. torch.optim.Adam.step (model/model.py:387-395) fused with the analytic gradient of
 * HierarchicalNeuralTexture.regularizer (texture.py:102-108: g += reg_coef[seg] * p), the next forward's
 * normalize() clamp (texture.py:41-44), zeroing of the gradient for the next step, and the reduction
 * sum(p_new^2) per segment that the next step's tex_reg loss value needs (sumsq_out[seg], pre-zeroed).
 * The arena holds n_seg layers back to back: seg_end[i] = end offset (floats) of layer i.
 * grad_scale multiplies the data-term gradient first (1/R after an all-reduce over R ranks).
 * bias_corr1 = 1-beta1^t and bias_corr2 = 1-beta2^t are computed by the caller in double; the betas are
 * doubles because torch derives the fp32 constants 1-beta from Python doubles.
 * dev_hyper (optional, device, 2 floats): when given, the kernel reads {lr / bias_corr1, 1 / sqrt(bias_corr2)}
 * from it instead of the scalar arguments, so a captured launch (hipGraph) can be replayed across steps
 * (sm_adam_hyper_step maintains it on the device).
 * touched (optional, device, one int32 per 2^touched_chunk_log2 floats of the arena, log2 in [2, 24]): chunks whose
 * flag is 0 are skipped without being read. Exact ONLY while every skipped element has p = g = m = v = 0 (a
 * zero-initialised texture, texture.py:26-28, that no view has reached yet: its regulariser gradient
 * reg_coef * p and its Adam update are then exactly 0); the caller ORs sm_tex_touch_flags of every view it has
 * ever optimised into the flags and passes NULL for textures that start non-zero (random_init, from_tensor).
 * g may be NULL (ABI 7): the data-term gradient is then taken as zero wherever the launch walks and is neither read nor
 * zeroed (the host's split update: the chunks the current views cannot reach, updated beside the forward pass). */
int sm_adam_fused(float* p, float* g, float* m, float* v, size_t n, const size_t* seg_end,
                  const float* reg_coef, int n_seg, float lr, double beta1, double beta2, float eps,
                  double bias_corr1, double bias_corr2, float grad_scale, float clamp_lo, float clamp_hi,
                  int zero_grad, float* sumsq_out, const float* dev_hyper, const int32_t* touched,
                  int touched_chunk_log2, void* stream);

/* Device-side step counter of the fused update for hipGraph replay (torch.optim.Adam's state['step'] and the
 * bias corrections of model/model.py:387-395's optimizer): state = {lr, step} (device, 2 doubles). Adds 1 to the
 * step and writes dev_hyper = {lr / (1 - beta1^step), 1 / sqrt(1 - beta2^step)} (2 floats, computed in double as
 * torch does). Captured together with sm_adam_fused(dev_hyper=...), every replay advances one step with no
 * step-dependent value crossing from the host. (ABI 11: the `guard` argument and the `valid` word of ABI 9 left with
 * the pair images.) */
int sm_adam_hyper_step(double* state, double beta1, double beta2, float* dev_hyper, void* stream);

/* Head of a training step in ONE launch: *reg_out = sum_l coef[l] * sumsq[l] (the regulariser loss `tex_reg` of the
 * current texture from the per-layer sums of squares the previous sm_adam_fused left; model/model.py:387-395), and a
 * zero fill of the two float ranges a step accumulates into (loss values + operand bounds; Gram slabs). n_a, n_b:
 * multiples of 4 (16-byte aligned ranges); either may be 0; reg_out may be NULL. */
int sm_step_begin(const float* sumsq, const float* coef, int n_seg, float* reg_out, float* zero_a, size_t n_a,
                  float* zero_b, size_t n_b, void* stream);

/* dst[i] = 1 wherever src[i] != 0 (int32 flag arrays of n entries): accumulates a view's sm_tex_touch_flags (after
 * the ranks' max-all-reduce, if any) into the ever-touched flags sm_adam_fused takes. */
int sm_flags_or(int32_t* dst, const int32_t* src, size_t n, void* stream);

/* normalize() alone (texture.py:41-44) + per-segment sum of squares (for the first step's tex_reg). */
int sm_clamp_sumsq(float* p, size_t n, const size_t* seg_end, int n_seg, float clamp_lo, float clamp_hi,
                   float* sumsq_out, void* stream);

/* ---- VGG feature stack: model/losses/content_and_style_losses.py:7-70 ------------------------------ */

/* K3/K4. 3x3, pad 1, stride 1 convolution as an implicit-im2col GEMM on fp32 MFMA.
 * in [Cin_pad][plane], wt [9][Cin_pad][Cout] (tap-major, Cout fastest; see sm_pack_* in the Python host),
 * out [Cout][plane]. Forward (nn.Conv2d + F.relu, :49-68): flags = SM_EPI_BIAS_RELU.
 * Data gradient (ATen convolution_backward + threshold_backward): call with the flipped / transposed
 * weights and flags = SM_EPI_RELU_MASK [| SM_EPI_ADD]; gate = the conv INPUT's forward activation.
 * Cin_pad must be a multiple of 4 (8 when > 4), Cout a multiple of 64.
 * ws (optional, ws_floats floats of caller-owned scratch): when the natural grid of 128x128 output tiles cannot
 * fill the chip the K dimension is split; every split stores its partial tile into its own slab of ws and a
 * second kernel reduces the slabs and applies the epilogue (deterministic, no atomics). NULL = never split. */
int sm_conv3x3(const float* in, const float* wt, const float* bias, float* out, const float* gate,
               int Cin_pad, int Cout, int H, int W, int flags, float* ws, size_t ws_floats, void* stream);

/* The same convolution over up to 8 feature maps of different sizes in ONE launch (the UV pyramid levels of
 * a view share the VGG weights): the position tiles of all problems form one grid, which removes the
 * per-level tail rounds and launch boundaries. problems: HOST array. */
typedef struct {
    const float* in;
    float* out;
    const float* gate;
    int H, W;
    /* sm_conv3x3_grouped_split2 only (NULL elsewhere), flags SM_EPI_RELU_MASK [| SM_EPI_ADD], all problems of a launch
     * or none: the conv's input is the backward of a 2x2 max-pool (+ the ReLU before it), taken on the fly. `in` is
     * then the gradient of the POOLED map [Cin][plane(H/2, W/2)] and unpool_code the code image
     * sm_maxpool2x2_fwd_codes_tiles wrote beside the pooled map ([Cin / 8][plane(H/2, W/2)] dwords, nibble c of
     * [g][q] = code of channel 8 g + c: 0..3 = the window element dy * 2 + dx holding the first maximum, 4 = maximum
     * <= 0): operand(y, x) = in(y/2, x/2) if code(y/2, x/2) == (y & 1) * 2 + (x & 1) else 0. Replaces the
     * sm_maxpool2x2_bwd_relu pass (2.75 plane-sizes of traffic) and the re-read of its output. */
    const uint32_t* unpool_code;
    /* sm_conv3x3_grouped_split2 with flags SM_EPI_BIAS_RELU | SM_EPI_POOL only (ABI 7; NULL elsewhere): the forward conv
     * BELOW a 2x2 max-pool (F.max_pool2d of the VGG, content_and_style_losses.py:56-68) takes the maxima in its epilogue:
     * pool_out [Cout][plane(H/2, W/2)] receives the pooled map, pool_code [Cout / 8][plane(H/2, W/2)] the argmax codes
     * (the formats of sm_maxpool2x2_fwd_codes_tiles, bit-identical values), `out` is NOT written - the pre-pool map has
     * no other reader. Needs a tile_list whose segments come in vertical PAIRS (entries 2k, 2k + 1 = the same 32 columns
     * of image rows 2Y and 2Y + 1, first column even: sm_cover_segments with pair_w). Replaces the pool pass and 1.75
     * plane sizes of HBM traffic per pool. */
    float* pool_out;
    uint32_t* pool_code;
    /* sm_conv3x3_grouped_split2 with flags SM_EPI_RELU_MASK | SM_EPI_GRAM only (ABI 7; NULL elsewhere), Cout = 64 or 128
     * (the kernel's row tile then holds all channels of a position), all problems with unpool_code: the conv's OUTPUT layer
     * is a style layer (relu1_1 / relu2_1 for the data gradients of conv1_2 / conv2_2) whose masked
     * Gram backward (content_and_style_losses.py:301-340 through autograd: sum over the masks k of m_k(q) (D_k F)(q),
     * F = `gate`) is added in the epilogue - the bits of sm_gram_backward_split2_grouped's result - instead of being
     * written by that call and read back through SM_EPI_ADD. gram_ws: the operand images of D0 / D1 as
     * sm_gram_backward_split2_grouped leaves them in its `ws` when called with dfeat = NULL (pack only); gram_mask0 /
     * gram_mask1 (optional): the layer's mask planes [plane(H, W)]; gram_amax_feat / gram_amax_d: the amax arrays of F and
     * of the derivative matrices. Such a launch never splits K (no workspace use). */
    const void* gram_ws;
    const float* gram_mask0;
    const float* gram_mask1;
    const float* gram_amax_feat;
    const float* gram_amax_d;
} sm_conv_problem;
/* "amax" bounds. An amax argument is a DEVICE array of sm_amax_floats() floats (64 slots, 256 bytes apart), zeroed by
 * the caller before the first launch that records into it; its VALUE is the maximum over the slots. Writers atomically
 * max max |output| into one slot per block (bit pattern of a non-negative float; spreading the words keeps thousands
 * of concurrent same-address atomics from serialising), readers take the max of the slots.
 * amax_out (optional): the launch records max |output| - the operand bound of an sm_conv3x3_grouped_split2 /
 * sm_gram_*_split that consumes the output. */
int sm_amax_floats(void);
int sm_conv3x3_grouped(const sm_conv_problem* problems, int n_problems, const float* wt, const float* bias,
                       int Cin_pad, int Cout, int flags, const int32_t* tile_list, int n_list, float* ws,
                       size_t ws_floats, float* amax_out, void* stream);
/* The same grouped convolution on the fp16 matrix cores (stylemesh_amd/csrc/conv_split_kernel.h), Cin % 16 == 0,
 * Cout % 64 == 0: TWO fp16 parts per fp32 operand and THREE partial products (hh' + hl' + lh', each exact in
 * fp32, fp32 accumulate; v_mfma_f32_32x32x16_f16): x s = h + l carries 22 significand bits, the dropped ll' is below
 * 2^-22 of a product. Activations / outputs are the same fp32 planes. fp16 has 5 exponent bits, so both operands are
 * scaled by powers of two:
 *   wt2 = the weights times a power of two s_w that puts max |w| into [2^14, 2^15), split by the host into
 *         [9 taps][Cin/16][2 parts][2][Cout][8] fp16 (runtime/ops.py:pack_conv_split2); w_scale_inv = 1 / s_w;
 *   amax_in (amax array, required) = an upper bound of max |x| over the input planes of all problems, recorded by
 *         the kernel that produced them (amax_out of that launch; a max-pool passes its input's bound through): the
 *         kernel scales x by the power of two that maps amax_in into [2^14, 2^15) while it stages the operand;
 *   the epilogue multiplies the accumulators by the (exact) inverse scales before bias / add / gate.
 * Elements more than 2^18 below amax_in lose low bits of l: an absolute error <= 2^-40 amax_in per element.
 * tile_list (ABI 6): a list of 32-position SEGMENTS, not of whole tiles - entry = (problem << 24) | q, q = index of the
 * segment's first position in the padded plane (a multiple of 4, >= Wp; segments of one launch must not overlap),
 * consumed tile positions / 32 entries per tile: ANY live segments of ONE problem form a tile (the dead 32-position
 * ranges inside 128-position tiles were 8-17 % of all matrix instructions of a step); every problem's run is padded to a
 * whole number of tiles with (problem << 24) | 0xFFFFFF, n_list counts entries and must be a multiple of tile positions
 * / 32. sm_cover_segments builds such lists from need maps.
 * Same flags / ws / amax_out semantics as sm_conv3x3_grouped; a tile covers sm_conv_split2_tile_positions(Cout)
 * positions (128; 256 for the 64-channel layers, whose 64 x 256 tiles keep four 64 x 64 wave tiles busy). Error against
 * an fp64 convolution: same class as the fp32-MFMA kernel (tests/test_kernels_gpu.py). */
int sm_conv_split2_tile_positions(int Cout);
int sm_conv3x3_grouped_split2(const sm_conv_problem* problems, int n_problems, const uint16_t* wt2, float w_scale_inv,
                              const float* bias, int Cin, int Cout, int flags, const int32_t* tile_list, int n_list,
                              float* ws, size_t ws_floats, const float* amax_in, float* amax_out, void* stream);
/* max |x| over a feature map [C][plane(H,W)], max-ed into the amax array like the convolutions' amax_out: the operand
 * bound of sm_conv3x3_grouped_split2 for tensors no convolution produced (the deepest loss layer's gradient). */
int sm_fmap_amax(const float* planes, int C, int H, int W, float* amax_out, void* stream);
/* tile_list (optional, DEVICE array of n_list entries (problem << 24) | tile): compute only these position
 * tiles; a tile covers sm_conv_tile_positions(Cin_pad, Cout) consecutive positions q starting at row 1 of the
 * problem's plane. Positions of absent tiles are neither read nor written. NULL = every tile. The caller uses
 * it to skip tiles that cannot influence the loss (outside the receptive-field-dilated mask of a UV level);
 * gradient planes must then be zero outside the listed tiles (they are: zero-initialised and only ever
 * written inside listed tiles until the caller re-zeroes them for a new view). */
int sm_conv_tile_positions(int Cin_pad, int Cout);

/* Data gradient of the first conv (64 -> 3 channels; conv1_1, :11,49): out [3][plane] from
 * dz [64][plane], wd [9][64][4] (tap-major, 3 real output channels + 1 zero). */
int sm_conv3x3_dgrad_c3(const float* dz, const float* wd, float* out, int Cin, int H, int W, void* stream);

/* nn.MaxPool2d(2,2) (:27-32,51,54,59,64), floor output size: in [C][plane(H,W)] -> out [C][plane(H/2,W/2)]. */
int sm_maxpool2x2_fwd(const float* in, float* out, int C, int H, int W, void* stream);

/* max_pool2d backward fused with the ReLU gate of the layer below: for each 2x2 window the gradient goes
 * to the first maximum (row-major, ATen order) if that activation is > 0.
 * act [C][plane(H,W)] forward activation, pooled [C][plane(H/2,W/2)], dpooled likewise, dact out. */
int sm_maxpool2x2_bwd_relu(const float* act, const float* pooled, const float* dpooled, float* dact, int C,
                           int H, int W, void* stream);

/* The three per-level memory-bound VGG helpers over up to 8 feature maps in ONE launch (the UV levels of a view;
 * per-level launches are latency-bound on the small levels). problems: HOST array; a / b / c / out per kernel:
 * dgrad_c3: a = dz, out;  pool forward: a = in, out;  pool backward: a = act, b = pooled, c = dpooled, out = dact. */
typedef struct {
    const float* a;
    const float* b;
    const float* c;
    float* out;
    int H, W;
} sm_plane_problem;
int sm_conv3x3_dgrad_c3_grouped(const sm_plane_problem* problems, int n, const float* wd, int Cin, void* stream);
int sm_maxpool2x2_fwd_grouped(const sm_plane_problem* problems, int n, int C, void* stream);
int sm_maxpool2x2_bwd_relu_grouped(const sm_plane_problem* problems, int n, int C, void* stream);
/* The same three grouped kernels restricted to the blocks that can influence the loss (as the tile lists of the
 * convolutions): tile_list = DEVICE array of n_list entries (problem << 24) | block, a block = sm_plane_tile_positions
 * (kind) consecutive positions from row 1 of the plane the kernel is indexed by - kind 0: conv1_1's data gradient
 * (1024 positions of the image plane), kind 1 / 2: pool forward / backward (256 positions of the POOLED plane; the
 * same list serves both). NULL = all blocks. Positions outside the listed blocks are neither read nor written. */
int sm_plane_tile_positions(int kind);
int sm_conv3x3_dgrad_c3_tiles(const sm_plane_problem* problems, int n, const float* wd, int Cin,
                              const int32_t* tile_list, int n_list, void* stream);
int sm_maxpool2x2_fwd_tiles(const sm_plane_problem* problems, int n, int C, const int32_t* tile_list, int n_list,
                            void* stream);
int sm_maxpool2x2_bwd_relu_tiles(const sm_plane_problem* problems, int n, int C, const int32_t* tile_list, int n_list,
                                 void* stream);
/* Pool forward that also records, per pooled element, WHERE its maximum came from: codes[i] = DEVICE code image of
 * problem i, [C / 8][plane(H/2, W/2)] dwords (C % 8 == 0; codes: HOST array of n pointers, or NULL = plain forward):
 * nibble c of [g][q] belongs to channel 8 g + c: 0..3 = window element dy * 2 + dx holding the first maximum in
 * row-major order (the rule of sm_maxpool2x2_bwd_relu), 4 = maximum <= 0 (no gradient passes the ReLU) or padding.
 * sm_conv3x3_grouped_split2 takes the pool's backward from these codes on the fly (sm_conv_problem::unpool_code). */
int sm_maxpool2x2_fwd_codes_tiles(const sm_plane_problem* problems, uint32_t* const* codes, int n, int C,
                                  const int32_t* tile_list, int n_list, void* stream);

/* ---- Gram / style / content losses: content_and_style_losses.py:74-80,136-143,288-350 --------------- */

/* K5a. Masked Gram sums S_k = (m_k F)(m_k F)^T for up to two 0/1 masks (GramMatrix :74-80 on
 * masked_features :136-143, without the 1/N). feat [C][plane]; mask0/mask1 one plane each (mask1 may be
 * NULL). The positions are split over many blocks per tile; each writes its partial tile to its own slab
 * (plain stores, deterministic) and, when there are more than 32, a second pass sums groups of 32. S0/S1 must
 * hold sm_gram_workspace_slabs(C,H,W) slabs of [C][C]; afterwards S_k = sum of the FIRST
 * sm_gram_num_slabs(C,H,W) slabs. Only the upper-triangular 64x64 tiles are valid. No pre-zeroing needed. */
int sm_gram_num_slabs(int C, int H, int W);
int sm_gram_workspace_slabs(int C, int H, int W);
int sm_gram_masked(const float* feat, const float* mask0, const float* mask1, float* S0, float* S1, int C,
                   int H, int W, void* stream);

/* K5a on the fp16 matrix cores at fp32 accuracy (fp16x2 split, 3 partial products, fp32 accumulate - see
 * sm_conv3x3_grouped_split2). Same arguments and results class as sm_gram_masked, but the position ranges add into
 * ONE slab with fp32 atomics (no reduction pass): afterwards S_k = the first sm_gram_split_num_slabs() = 1 slab
 * (upper-triangular 64x64 tiles valid; S0 / S1 need room for one [C][C] slab only). Stages whose 16 mask values
 * are all zero are skipped before their data is loaded.
 * amax_feat (REQUIRED since ABI 11, amax array): bound of max |feat| recorded by the producing conv - the operand is
 * split into two fp16 parts scaled by a power of two from it (sm_conv3x3_grouped_split2). */
int sm_gram_split_num_slabs(void);
int sm_gram_masked_split(const float* feat, const float* mask0, const float* mask1, float* S0, float* S1, int C,
                         int H, int W, const float* amax_feat, void* stream);
/* Same kernel without the zero fill: the position ranges ADD into S0 / S1, which the caller has zeroed (one fill
 * over the slabs of every level and layer of a step instead of two per call) or which hold a partial sum to
 * continue. */
int sm_gram_masked_split_acc(const float* feat, const float* mask0, const float* mask1, float* S0, float* S1, int C,
                             int H, int W, const float* amax_feat, void* stream);
/* The same Gram forward for MANY (level, layer) problems in at most two launches (one per tile class: 64-channel tiles
 * for C % 128 != 0, 128-channel tiles otherwise) - the loss phase of a step is ~20 small problems whose separate
 * launches are latency-bound. fp16x2 operands (amax_feat required, see sm_conv3x3_grouped_split2). S0 / S1 must be
 * ZERO on entry (every position range adds into them atomically); mask1 / S1 may be NULL. problems: HOST array. */
typedef struct {
    const float* feat;      /* [C][plane(H,W)] */
    const float* mask0;     /* [plane(H,W)] 0/1 */
    const float* mask1;
    float* S0;              /* [C][C]; upper-triangular 64x64 tiles valid */
    float* S1;
    const float* amax_feat; /* amax array: bound of max |feat| */
    int C, H, W;
} sm_gram_problem;
int sm_gram_masked_split2_grouped(const sm_gram_problem* problems, int n_problems, void* stream);


/* K5b. Style-loss value and its derivative matrices for one (level, layer) (:301-340).
 * S0/S1: the n_slabs partial slabs written by sm_gram_masked (summed here).
 * For each mask k: G_k = S_k / max(N_k,1) (N from counts[k]; N_k == 0 -> G_k = 0, and with
 * skip_if_empty[k] the term is dropped, :332). Terms: for t in 0..n_terms-1: mask k = term_mask[t],
 * target Y = targets[t] ([C][C]), loss += coef * mean((Y - G_k)^2), D_k += coef' * (G_k - Y).
 * coef = style_weight * loss_weight * (*factor) [/ avg_n for gram_mode 'average'].
 * Writes D0/D1 [C][C] (full symmetric) such that dL/dF = m_0 * (D0 F) + m_1 * (D1 F), and atomically adds
 * the loss value to *loss_out. history (may be NULL): [9][C][C] ring of past normalised mask-0 Grams for
 * gram_mode 'average' (:319-323): the first hist_len entries are averaged with the current Gram, which is
 * then stored in entry hist_slot. targets / term_mask / skip_if_empty are HOST arrays (<= 4 terms).
 * amax_d_out (optional, amax array, caller-zeroed): receives max(|D0|, |D1|), the operand bound of the fp16x2
 * sm_gram_backward_split. */
int sm_style_loss(const float* S0, const float* S1, const float* counts, const float* factor,
                  const float* const* targets, const int* term_mask, int n_terms, const int* skip_if_empty,
                  float weight, int C, float* D0, float* D1, float* loss_out, float* history, int hist_len,
                  int hist_slot, int n_slabs, float* amax_d_out, void* stream);

/* K5c. dF[c][q] = m0[q] * (D0 F)[c][q] + m1[q] * (D1 F)[c][q], optionally gated by feat > 0
 * (for the top layer r51, whose ReLU gate no later dgrad applies). OVERWRITES dfeat. */
int sm_gram_backward(const float* feat, const float* mask0, const float* mask1, const float* D0,
                     const float* D1, float* dfeat, int C, int H, int W, int relu_gate, void* stream);

/* K5c on the fp16 matrix cores (same split). ws: DEVICE scratch of sm_gram_backward_split_ws_bytes(C) bytes that
 * receives the split image of D0 / D1 (a small pack kernel runs first on the same stream).
 * amax_feat / amax_d (REQUIRED since ABI 11; amax arrays): bounds of max |feat| and max(|D0|, |D1|) - the operands are
 * split into two fp16 parts scaled by powers of two from them (as sm_conv3x3_grouped_split2). */
size_t sm_gram_backward_split_ws_bytes(int C);
int sm_gram_backward_split(const float* feat, const float* mask0, const float* mask1, const float* D0,
                           const float* D1, float* dfeat, int C, int H, int W, int relu_gate, void* ws,
                           const float* amax_feat, const float* amax_d, void* stream);

/* The loss phase of a step in a handful of launches. sm_style_loss_grouped = sm_style_loss for many (level, layer)
 * problems (same field meanings as its arguments), all adding into ONE loss value; sm_gram_backward_split2_grouped =
 * sm_gram_backward_split (fp16x2: amax_feat / amax_d required) for many problems: one launch packs every derivative
 * matrix, then one launch per row-tile class (64 / 128 channels) runs the GEMMs. ws: per-problem DEVICE scratch of
 * sm_gram_backward_split_ws_bytes(C) bytes (distinct for every problem of a call). problems: HOST arrays. */
typedef struct {
    const float* S0;
    const float* S1;
    const float* counts;
    const float* factor;
    const float* targets[4];
    int term_mask[4];
    int n_terms;
    int skip_if_empty[2];
    float weight;
    int C;
    float* D0;
    float* D1;
    float* history;
    int hist_len, hist_slot, n_slabs;
    float* amax_d_out;
} sm_style_problem;
int sm_style_loss_grouped(const sm_style_problem* problems, int n_problems, float* loss_out, void* stream);
typedef struct {
    const float* feat;
    const float* mask0;
    const float* mask1;
    const float* D0;
    const float* D1;
    float* dfeat;           /* NULL (ABI 7): only the operand images of D0 / D1 are written into ws - a conv launch with
                             * SM_EPI_GRAM (sm_conv_problem::gram_ws) computes this problem's result in its epilogue */
    void* ws;
    const float* amax_feat;
    const float* amax_d;
    float* amax_out;        /* optional amax array: records max |dfeat| (see sm_conv3x3_grouped: amax_out) */
    int C, H, W, relu_gate;
} sm_gram_bwd_problem;
int sm_gram_backward_split2_grouped(const sm_gram_bwd_problem* problems, int n_problems, void* stream);

/* K6. Masked content MSE (:343-348): loss += coef * sum m (P-T)^2 / (C N); dP = coef * 2 m (P-T)/(C N),
 * coef = content_weight * loss_weight * (*factor), N = *count (0 -> nothing). OVERWRITES dpred.
 * relu_gate: additionally zero dP where pred <= 0 (when this is the deepest layer, see K5c). */
int sm_mse_masked(const float* pred, const float* target, const float* mask, const float* count,
                  const float* factor, float weight, float* dpred, float* loss_out, int C, int H, int W,
                  int relu_gate, void* stream);

/* ---- per-view constants: model/model.py:188-254, content_and_style_losses.py:146-217 ---------------- */

/* Level masks and depth-interpolation weights at the view resolution (mask_depth, mask_interpolation_weight,
 * erode: model/model.py:204-239). For level i: E[i] = erode(((rounded==i)|(other==i)) & mask),
 * Wt[i] = erode((rounded==i)&mask)*w + erode((other==i)&mask)*(1-w). E, Wt: [n_levels][h][w]. */
int sm_level_masks(const int64_t* rounded, const int64_t* other, const float* interp_w, const uint8_t* mask,
                   int h, int w, int n_levels, float* E, float* Wt, void* stream);

/* Per-level maps at the level's resolution (model/model.py:199,219,238; losses :161):
 * M = nearest(E) > 0; pixel_weight = [bilinear(angle_guidance)] * [nearest(Wt)]; passed = bilinear(angle_deg) < thr.
 * Any of Wt, angle_guidance, angle_deg, pixel_weight, passed may be NULL (factor 1 / passed = 1 / not written).
 * Also accumulates sum(M) into *m_sum (pre-zeroed). */
int sm_level_maps(const float* E, const float* Wt, const float* angle_guidance, const float* angle_deg,
                  float angle_threshold, int h, int w, int H, int W, float* M, float* pixel_weight,
                  uint8_t* passed, float* m_sum, void* stream);

/* Layer-resolution masks (calculate_pyramid :172-174,181): nearest-down of M, M*passed, M*!passed into
 * padded planes m_all / m_pass / m_fail (hl x wl), and their sums into counts[0..2] (pre-zeroed). */
int sm_layer_masks(const float* M, const uint8_t* passed, int H, int W, int hl, int wl, float* m_all,
                   float* m_pass, float* m_fail, float* counts, void* stream);

/* factor[i] = (counts_all[i]/size[i]) / sum_j (counts_all[j]/size[j]) over n levels (:181,200-204).
 * counts_all[i] points at the level's N_all; all device pointers given as a device array. */
int sm_level_factors(const float* const* counts_all, const float* sizes, int n, float* const* factors,
                     void* stream);

/* F.interpolate(..., mode='bilinear') between padded planar fmaps (:176, content target). */
int sm_fmap_resize_bilinear(const float* in, int C, int h, int w, float* out, int H, int W, void* stream);

/* F.interpolate(..., mode='bilinear') dense [C][h][w] -> padded planar [C][plane(H,W)] (style pyramid :94,120;
 * also the plain dense->padded copy when sizes match). */
int sm_image_to_fmap(const float* in, int C, int h, int w, float* out, int H, int W, void* stream);
int sm_fmap_to_image(const float* in, int C, int H, int W, float* out, void* stream);

/* ---- dead-tile analysis (which positions of a VGG layer can reach the loss) -------------- */

/* need_src = (mode 1: 3x3 dilation of need_out | mode 2: 2x2 up-sampling of need_out | mode 0: nothing)
 * OR (M != NULL: the level mask M [H][W] nearest-down-sampled to (hs,ws)). Dense [h][w] 0/1 float maps. */
int sm_need_step(const float* need_out, int ho, int wo, int mode, const float* M, int H, int W, float* need_src,
                 int hs, int ws, void* stream);
/* flags[t] = does position tile t (bn positions, sm_conv_tile_positions) of an (h,w) plane hold a needed pixel;
 * flags has ceil(h * sm_fmap_row_stride(w) / bn) entries. */
int sm_tile_flags(const float* need, int h, int w, int bn, uint8_t* flags, void* stream);

/* Active SEGMENT lists of the split conv kernels (sm_conv3x3_grouped_split / _split2: tile_list): for up to 64 need maps
 * in one call, cover the needed positions of each plane with disjoint 32-position segments that start at any multiple
 * of 4 (greedy over the flattened padded plane). starts receives (tag << 24) | q, q = first position of a segment as an
 * index into the padded plane (ascending; at most cap entries), *count their number (device). Once per view.
 * ABI 8: the cover is computed in parallel (bit images -> per-chunk tables of the greedy's 8 possible entry states ->
 * one scan per map -> emission: four launches, no limit on the plane size beyond the 24-bit position of a list entry;
 * the lists are those of the sequential greedy, bit for bit) and needs ws: sm_cover_segments_ws_bytes(problems, n)
 * bytes of 16-byte-aligned device scratch. */
typedef struct {
    const float* need;   /* [h][w] 0 / 1 */
    int32_t* starts;
    int32_t* count;
    int h, w, tag, cap;
    /* < 0 (ABI 8): TILE mode - aligned tiles of -pair_w positions (a divisor of 2048) instead of segments: starts
     * receives (tag << 24) | t for every tile t that holds a needed position (sm_tile_flags + compaction in one).
     * 0: as above. > 0 (ABI 7): PAIR mode for a conv with SM_EPI_POOL - need is the need map of the POOLED plane
     * [h][w], pair_w the width of the full-resolution plane the conv writes (w == pair_w / 2); every pooled row Y is
     * covered with runs of 16 windows starting at any window X0, and each run becomes TWO entries: the segment of
     * image row 2Y that starts at column 2 X0, then the one right below it (row 2Y + 1). */
    int pair_w;
    /* ABI 10. 1: QUAD mode, the lists of the resident-input conv kernel (SM_LIST_QUADS) - the rows of the need map are
     * walked in GROUPS whose bits are OR-ed: two pooled rows with pair_w > 0 (runs of 16 windows from any window X0,
     * columns 2 X0 ...), four rows with pair_w == 0 (runs of 32 positions from the first needed position, rounded down to even) - and every run
     * becomes FOUR entries: the same 32 columns of image rows 4 Y .. 4 Y + 3 (q, q + Wp, q + 2 Wp, q + 3 Wp; a row
     * behind the last image row: (tag << 24) | 0xFFFFFF). Runs of one group are disjoint. 0: as above. */
    int quad;
} sm_cover_problem;
size_t sm_cover_segments_ws_bytes(const sm_cover_problem* problems, int n);
int sm_cover_segments(const sm_cover_problem* problems, int n, void* ws, size_t ws_bytes, void* stream);

/* ---- the per-view constants of a whole view in TWO calls (ABI 8) ------------------------------------------------
 * Everything below depends on the view only; the reference recomputes it every step (model/model.py:204-254,
 * content_and_style_losses.py:146-217). Rounds 1-3 sequenced it from the Python host with one call per (level, layer):
 * 130 (one UV level) to 250 (four) launches per view change = 2.6 - 4.3 ms of HOST time, which bounds the schedules that
 * change the view every step (--index_repeat 1, scripts/train/optimize_texture_scannet_dip.sh:16). The two entry points
 * take a whole view's tables and issue grouped launches (one per kind over all levels / layers). Descriptors and the
 * tables they point to are HOST memory, read during the call; every other pointer is device memory. */
#define SM_VIEW_MAX_LEVELS 8
#define SM_VIEW_MAX_LAYERS 24      /* 'img' + the outputs of the VGG nodes up to the deepest loss layer */
#define SM_VIEW_MAX_LISTS 48
typedef struct {
    int H, W;                  /* resolution of the UV level */
    int has_maps;              /* 0: the level takes no part (without depth scaling all but the last, model.py:253-254) */
    float* M;                  /* out [H][W]: the level's mask (sm_level_maps) */
    float* pixel_weight;       /* out [H][W] or NULL */
    uint8_t* passed;           /* out [H][W] */
    float* m_sum;              /* out: sum of M (zeroed by the call) */
} sm_view_level;
typedef struct {
    int level;                 /* index into sm_view_masks_desc.levels */
    int loss_layer;            /* the level factors are normalised over the entries of one loss layer (:181,200-204) */
    int hl, wl;                /* the loss layer's resolution at that level */
    float* mask_planes;        /* out: three padded planes [plane(hl, wl)]: all, passed, failed (sm_layer_masks) */
    float* counts;             /* out [3] (zeroed by the call) */
    float* factor;             /* out [1] */
} sm_view_layer_mask;
typedef struct {
    const float* src; int C, h, w;   /* padded planar source */
    float* dst; int H, W;            /* padded planar destination (sm_fmap_resize_bilinear) */
} sm_view_resize;
typedef struct {
    const uint8_t* mask;             /* [h][w] */
    const float* angle_guidance;     /* [h][w] or NULL (no angle weighting) */
    const float* angle_degrees;      /* [h][w] */
    const int64_t* rounded;          /* depth scaling: [h][w] level indices + interpolation weight; else all NULL */
    const int64_t* other;
    const float* interp_w;
    int h, w;
    float angle_threshold;
    float* E;                        /* scratch [n_levels][h][w] with depth scaling, [h][w] without */
    float* Wt;                       /* scratch [n_levels][h][w] with depth scaling, NULL without */
    int n_levels;
    sm_view_level levels[SM_VIEW_MAX_LEVELS];
    int n_masks;                     /* <= 64 */
    const sm_view_layer_mask* masks;
    int n_resizes;                   /* <= 16: content targets (the content pass must have run on this stream) */
    const sm_view_resize* resizes;
} sm_view_masks_desc;
/* sm_level_masks + sm_level_maps of every level, sm_layer_masks of every (level, loss layer), sm_level_factors of every
 * loss layer, the content targets' resizes: five launches. */
int sm_view_masks(const sm_view_masks_desc* desc, void* stream);

typedef struct {
    int layer;                 /* index of the need map the list is built from (pair mode: the POOLED layer) */
    int mode;                  /* 0: free 32-position segments, 1: segment pairs (sm_cover_segments pair mode; the conv's
                                * full-resolution output is layer `pair_layer`), 2: aligned tiles of `bn` positions,
                                * entry = (level << 24) | tile (bn divides 2048); ABI 10: 3 = quads over the POOLED
                                * layer's need map (as mode 1, sm_cover_problem::quad), 4 = quads over the layer's own */
    int bn;
    int pair_layer;
    int group;                 /* every level's run is padded to a multiple of `group` entries with (level << 24) | 0xFFFFFF */
    int32_t* out;              /* the list: the levels' runs back to back */
    int cap;
    int32_t* staging;          /* scratch of the covers: n_levels x staging_cap entries */
    int staging_cap;
} sm_view_list;
typedef struct {
    int n_levels;                                   /* the ACTIVE levels; list entries carry the index into this table */
    const float* M[SM_VIEW_MAX_LEVELS];             /* level masks [H][W] */
    int H[SM_VIEW_MAX_LEVELS], W[SM_VIEW_MAX_LEVELS];
    int n_layers;                                   /* layer 0 = the image, layer j + 1 = output of VGG node j */
    int node_is_pool[SM_VIEW_MAX_LAYERS];           /* node j (produces layer j + 1 ...) */
    int node_src[SM_VIEW_MAX_LAYERS];               /* ... from this layer */
    int injected[SM_VIEW_MAX_LAYERS];               /* per layer: a loss reads it (the level mask is OR-ed into its need) */
    float* need[SM_VIEW_MAX_LEVELS][SM_VIEW_MAX_LAYERS];   /* out: need maps [h][w] of every (level, layer) */
    int lh[SM_VIEW_MAX_LEVELS][SM_VIEW_MAX_LAYERS];
    int lw[SM_VIEW_MAX_LEVELS][SM_VIEW_MAX_LAYERS];
    int n_lists;
    const sm_view_list* lists;
    int32_t* summary;          /* out, device: per list [n_list (padded entries), live entries of level 0 .. 7] = 9 ints */
    void* ws;                  /* scratch: sm_view_lists_ws_bytes(desc) bytes, 16-byte aligned */
    size_t ws_bytes;
} sm_view_lists_desc;
/* The dead-tile analysis of a view (runtime/sparsity.py): the need maps of every (level, layer) - one launch per VGG node
 * over all levels - then every active list of the step's conv / pool launches (covers of all lists in batches of 64 maps,
 * one concatenation launch). No host synchronisation: the caller reads `summary` back when it needs the grid sizes. */
size_t sm_view_lists_ws_bytes(const sm_view_lists_desc* desc);
int sm_view_lists(const sm_view_lists_desc* desc, void* stream);

/* ---- multi-GPU: SURVEY.md section 8 e --------------------------------------------------------------- */

/* The compact form of the gradient arena that travels between the ranks (the reference has no collective: this is the
 * exchange step of SURVEY.md section 8 e; runtime/distributed.py:SparseGradReducer sequences it).
 *   sm_flags_compact: ascending list of the indices k with flags[k] != 0 into idx_out (capacity n_flags) and their
 *     number into *count_out (device) - deterministic, so that ranks holding identical (max-all-reduced) flags derive
 *     identical lists; ws: sm_flags_compact_ws_ints(n_flags) int32 of scratch. No host synchronisation.
 *   sm_chunks_gather: compact[j][0 .. 2^chunk_log2) = arena[idx[j] << chunk_log2 ...] for j < n, where n = *n_idx_dev
 *     if that (device) pointer is given - n_idx is then only the capacity the launch is sized for - else n_idx.
 *   sm_chunks_scatter: the inverse copy, times `scale` (1 / world size if the caller wants the mean here).
 * chunk_log2 in [2, 24]; the arena pointer must be 16-byte aligned. */
size_t sm_flags_compact_ws_ints(size_t n_flags);
int sm_flags_compact(const int32_t* flags, size_t n_flags, int32_t* idx_out, int32_t* count_out, int32_t* ws,
                     void* stream);
int sm_chunks_gather(const float* arena, const int32_t* idx, const int32_t* n_idx_dev, size_t n_idx, int chunk_log2,
                     float* compact, void* stream);
int sm_chunks_scatter(float* arena, const int32_t* idx, const int32_t* n_idx_dev, size_t n_idx, int chunk_log2,
                      const float* compact, float scale, void* stream);

/* A HIP stream confined to n_cus (8 .. 256, a multiple of 8 is spread evenly: n_cus / 8 compute units of every XCD) of
 * the device's compute units - for work with slack that should share few CUs with the step's trunk (the Python host's
 * side streams, STYLEMESH_SIDE_CUS). Returns a hipError_t. */
int sm_stream_create_cu_subset(int n_cus, void** stream_out);
int sm_stream_destroy(void* stream);

/* RCCL over xGMI, one communicator per process (one process per GPU). The reference has no collective; this is the
 * exchange of the R-GPU step defined in SURVEY.md section 8 e. All return a ncclResult_t (0 = success).
 *   sm_comm_get_unique_id: rank 0 fills id_out (sm_comm_unique_id_bytes() bytes, HOST) and ships it to the other
 *     ranks by any side channel (the Python host broadcasts it through torch.distributed).
 *   sm_comm_init: collective over all ranks (ncclCommInitRank) on the CURRENT device; *comm_out = communicator handle.
 *   sm_allreduce_grad: in-place fp32 SUM all-reduce of n floats (the gradient arena or its compacted dirty chunks),
 *     enqueued on `stream` - ordered with the kernels that wrote g and the fused update that reads it.
 *   sm_allreduce_flags_max: in-place int32 MAX all-reduce (the per-view touch flags of sm_tex_touch_flags). */
size_t sm_comm_unique_id_bytes(void);
int sm_comm_get_unique_id(void* id_out);
int sm_comm_init(void** comm_out, int n_ranks, const void* unique_id, int rank);
int sm_comm_destroy(void* comm);
int sm_allreduce_grad(void* comm, float* g, size_t n, void* stream);
int sm_allreduce_flags_max(void* comm, int32_t* flags, size_t n, void* stream);
/* (ABI 8) What the communicator actually is, for the record a multi-GPU run leaves behind: info_out[0..3] (HOST ints) =
 * ranks in the communicator (ncclCommCount), this rank (ncclCommUserRank), its HIP device (ncclCommCuDevice), the RCCL
 * version (ncclGetVersion). */
int sm_comm_info(void* comm, int* info_out);
/* (ABI 8) The link between two HIP devices of this node as the runtime reports it (hipExtGetLinkTypeAndHopCount):
 * *link_type = HSA_AMD_LINK_INFO_TYPE_* (1 = PCIe, 4 = xGMI), *hops = hop count. Returns a hipError_t. */
int sm_device_link(int device_a, int device_b, int* link_type, int* hops);

/* ---- R1: launch tables (ABI 8) ------------------------------------------------------------------------------
 * A training step of the Python host is a fixed sequence of calls into this library; sm_call_replay issues a recorded
 * sequence with ONE call (csrc/replay.hip). An entry holds the id of an entry point whose last parameter is the stream
 * (sm_call_id(name); -1: not replayable) and its arguments as 64-bit words in declaration order: pointers and integers
 * by value, a float as its bit pattern in the low 32 bits, a double as its bit pattern. The recorded stream word is
 * ignored: every call is issued on `stream`. skip != 0: the entry is passed over (a launch whose active list is empty
 * for the current view). HOST arrays an argument points to (problem tables) must stay alive and are read at replay
 * time - the host may update them, and any argument word, between replays. Returns the first non-zero return code and
 * the index of the call that produced it. */
#define SM_CALL_MAX_ARGS 24
typedef struct {
    int fn;
    int n_args;
    int skip;
    int reserved;
    uint64_t args[SM_CALL_MAX_ARGS];
} sm_call;
int sm_call_id(const char* name);
int sm_call_n_args(int id);
int sm_call_replay(const sm_call* calls, int n, void* stream, int* failed_index);
/* dst[0 .. n) = src[0 .. n) / = 0 on the stream (device memory): the two torch operations of a step - the clone of the
 * loss pair, the fill of the sums of squares - as library calls, so that a recorded step has no gap in it. */
int sm_copy_floats(float* dst, const float* src, size_t n, void* stream);
int sm_zero_floats(float* dst, size_t n, void* stream);

/* ---- E1: multi-view consistency metric --------------------------------------- */

/* reproject() of data/utils.py:73-194 + the masked squared error of scripts/eval/eval_image_folders.py:286-305 for
 * one view pair: every source pixel is un-projected with depth_src, moved by src2tar (HOST, 4x4 row-major fp32 =
 * inverse(cam2world_tar) * cam2world_src), projected with intrinsics (HOST: fx, fy, cx, cy) and rejected when its
 * depth is 0, it lands outside [0,W-1) x [0,H-1), none of the 4 neighbouring target depths is within depth_tol
 * (reference: 0.1) of its target-space z, or the bilinearly warped mask_tar (float 0/1) is <= 0.99. Dense [H][W] /
 * [3][H][W] fp32 images (not padded planes). color_out = warped color_tar (0 where rejected), mask_out = uint8
 * validity; if styled_src is given, partial[2 b] / partial[2 b + 1] (DEVICE doubles, sm_reproject_blocks(H,W)
 * pairs) receive block b's sum of squared differences styled_src - color_out over valid pixels and the number
 * of summed elements (3 per valid pixel): MSE of the pair = sum / count. */
int sm_reproject_blocks(int H, int W);
int sm_reproject(const float* src2tar, const float* intrinsics, int H, int W, const float* depth_src,
                 const float* depth_tar, const float* color_tar, const float* mask_tar, const float* styled_src,
                 float* color_out, uint8_t* mask_out, double* partial, float depth_tol, void* stream);

/* ---- R1: UV / angle / depth rasteriser ----------------------------------------- */

/* Replaces the OpenGL renderer scripts/scannet/render_uv (renderer.cpp:165-224, scannet_renderer.cpp:19-84,
 * shader/{uvmap,angle,depth}.*, include/util.h:11-35) for one pose: per pixel the interpolated (u, v) of the
 * nearest surface of the UV-parameterised mesh, the cosine between the interpolated vertex normal and the
 * direction to the eye (clamped at 0) and the view-space depth; background = 0.
 * verts / normals [V][3], uvs [V][2], faces [F][3] int32: DEVICE. world2cam: HOST 4x4 row-major (camera x right,
 * y down, z forward = inverse of the ScanNet pose), intrinsics: HOST fx, fy, cx, cy at the render resolution
 * (pixel (i, j) is sampled at (i + 0.5, j + 0.5), u = fx X/Z + cx). zbuf: DEVICE scratch of H*W uint64.
 * big_scratch (optional): DEVICE scratch of 1 + 10 * big_cap floats - triangles whose screen box exceeds 256 pixels
 * are queued there and rasterised by one block each instead of one lane.
 * uv_out [H][W][3] (third channel 0), angle_out / depth_out [H][W]. Surfaces nearer than znear are clipped,
 * beyond zfar dropped (reference: 0.1 / 10 in the depth shader). */
int sm_raster_maps(const float* verts, const float* normals, const float* uvs, const int32_t* faces, int n_faces,
                   const float* world2cam, const float* intrinsics, int H, int W, float znear, float zfar,
                   uint64_t* zbuf, float* big_scratch, int big_cap, float* uv_out, float* angle_out,
                   float* depth_out, void* stream);

/* R2: textured re-render of a rasterised view (the rgb path of the same renderer: GL_LINEAR_MIPMAP_LINEAR lookup
 * of the optimised texture, renderer.cpp:110-139, shader/rgb.frag; its constant lighting factor and the anisotropic
 * extension are not reproduced). sm_mip_downsample: one 2x2 box-filter level (dst = [C][max(Hs/2,1)][max(Ws/2,1)]).
 * sm_tex_sample_mip: levels = HOST array of n_levels DEVICE images [3][h_l][w_l] (level 0 first), uv = the
 * [H][W][3] map of sm_raster_maps; out [3][H][W]; lod_out (optional) [H][W] = the level of detail used. */
int sm_mip_downsample(const float* src, float* dst, int C, int Hs, int Ws, void* stream);
int sm_tex_sample_mip(const float* const* levels, const int* level_w, const int* level_h, int n_levels,
                      const float* uv, int H, int W, float* out, float* lod_out, void* stream);

#ifdef __cplusplus
}
#endif
#endif
