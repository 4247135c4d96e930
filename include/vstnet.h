/*
 * vstnet.h — C ABI of libvstnet_hip.so: the MI355X (gfx950) implementation of CAP-VSTNet's
 * inference hot path (RevResNet forward/inverse + cWCT).
 *
 * The reference (delldu/VSTNet) is pure Python on torch ops and defines no native interface for
 * this path; its closest native analogue is ggml's
 *     GGMLNetwork::engine_forward(int argc, TENSOR* argv[])   project/ggml/include/ggml_engine.h:610
 * Each entry point below names the reference Python symbol (file:line, relative to the reference
 * root) whose device work it replaces.  INTEGRATION.md shows the ctypes binding a maintainer of the
 * reference would add in models/RevResNet.py / models/cWCT.py.
 *
 * Conventions
 *   - every function returns int: 0 = ok, <0 = VST_E_* (bad argument / unsupported shape),
 *     >0 = a hipError_t raised by a launch.  No exceptions cross the ABI.
 *   - all pointers are DEVICE pointers unless the name ends in _host; the library never allocates
 *     device memory: outputs and workspaces are caller-provided (vst_*_workspace_bytes tell sizes).
 *   - `stream` is a hipStream_t passed as void*; all work is stream-ordered and asynchronous.
 *     Entry points are re-entrant; use one stream per host thread / per GPU.
 *   - external tensors are NCHW fp32 contiguous (the reference's convention); H, W multiples of 4,
 *     H, W >= 8 (SURVEY.md 8(b) "Tensor conventions").
 *   - internal "state" buffers use the ZC layout described in DESIGN.md (quarter-resolution cells
 *     of 256 channels-last floats); they are opaque to callers of the whole-pass entry points.
 */
#ifndef VSTNET_H
#define VSTNET_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif
/* the library is built with hidden default visibility for its host code: exactly the declarations below are exported */
#pragma GCC visibility push(default)

#define VST_OK 0
#define VST_E_ARG (-1)      /* null pointer, non-positive size */
#define VST_E_SHAPE (-2)    /* H/W not multiple of 4, < 8, unsupported channel count */
#define VST_E_MODE (-3)     /* unknown precision / sp_steps / direction */
#define VST_E_WORKSPACE (-4)

/* conv arithmetic */
#define VST_PREC_BF16X3 0   /* bf16 MFMA, hi/lo split operands (3 products), fp32 accumulate: ~3e-6 rel */
#define VST_PREC_FP32 1     /* plain fp32 FMA direct convolution (diagnostic / cross-check, slow) */
#define VST_PREC_F16X2 2    /* fp16 MFMA, w_hi * (x_hi + x_lo): 2 products, activations split into fp16 hi + lo (22 bits),
                               weights rounded to fp16 once at pack time; in the 256-channel stride-1 blocks the
                               operands are pre-split in HBM and staged by LDS-DMA:
                               ~1.5e-4 rel on the code, ~4e-6 on a stylised frame */

#define VST_PREC_F16X2H 3   /* F16X2 with every conv input that crosses HBM taken as fp16 (11 bits) instead of fp16 hi + lo:
                               in the 256-channel blocks h1, h2 and the state as the first conv reads it (its hi plane;
                               the state itself keeps hi + lo) - one MFMA per product, half the bytes -, and h1 of the
                               16- and 64-channel blocks (2 bytes per value through HBM instead of 4):
                               ~1.75e-4 rel (1.94e-4 max) on the code, ~4e-5 (1.1e-4 max) on a stylised frame */

#define VST_NUM_BLOCKS 32   /* 30 stack blocks + 2 channel_reduction blocks */

int vst_version(void);
const char* vst_error_string(int code);

/* ---------------------------------------------------------------------------------------------
 * Weights.  A residual_block (models/RevResNet.py:68-94) has three 3x3 convs (conv.1, conv.4,
 * conv.7).  vst_conv_packed_bytes/vst_pack_conv turn one OIHW fp32 weight tensor (device) into the
 * packed form the kernels read: [fp32 taps-major copy | bf16 hi fragments | bf16 lo fragments
 * | for cin, cout >= 64: fp16 fragments in the K order of the LDS-DMA kernels].
 * ------------------------------------------------------------------------------------------- */
size_t vst_conv_packed_bytes(int cout, int cin);
int vst_pack_conv(const float* w_oihw, int cout, int cin, void* packed, void* stream);

/* Exponent normalisation of one residual_block's intermediates (models/RevResNet.py:79-88): ReLU is positively homogeneous, so
 *   h1 = ReLU(W1 x + b1), h2 = ReLU(W4 h1 + b4), F = W7 h2 + b7
 * is unchanged by (W1, b1) *= s1 per output channel, W4 /= s1 per input channel, (W4, b4) *= s2 per output channel, W7 /= s2 per
 * input channel.  With s = 2^-round(log2 ||row||_2) (powers of two: exact in fp32, bit-identical results in the fp32-class
 * modes) every intermediate channel's weight row has unit scale, so h1 / h2 have the scale of the state whatever per-channel
 * scales a checkpoint was trained into, which is what the fp16 operand range of VST_PREC_F16X2 / F16X2H wants.  In place on
 * DEVICE copies of the five tensors (OIHW fp32 / [cout]); call before vst_pack_conv.  scales (optional) = float[2 * c_mid]
 * {s1, s2}.  c_mid <= 64. */
int vst_normalize_block(float* w1, float* b1, float* w4, float* b4, float* w7, int c_in1, int c_mid, int c_out,
                        float* scales, void* stream);

/* fp16 range flags of the narrowed modes, accumulated on the device since the last reset: */
#define VST_RANGE_SATURATED 1u /* an activation beyond +-65504 was clamped when it was rounded to fp16 (F16X2 / F16X2H) */
#define VST_RANGE_WEIGHT 2u    /* vst_pack_conv met a weight beyond +-65504 (its fp16 copy is +-Inf; BF16X3 / FP32 unaffected) */
/* *flags_host = OR of the flags raised on the current device; reset != 0 clears them.  Synchronises the device (a calibration
 * / diagnostic call: RevResNet.check_range, bench.py, tests), never called by the passes themselves. */
int vst_range_flags(unsigned* flags_host, int reset);
/* the same without synchronising: four words (their OR = the flags) copied to DEVICE memory in stream order - a frame loop
 * appends them to the frame's own D2H copy and looks at them when it retires the frame (vstnet_amd/pipeline.py) */
int vst_range_flags_async(unsigned* flags4_dev, void* stream);

typedef struct vst_conv_weights {
    const void* packed;   /* from vst_pack_conv */
    const float* bias;    /* [cout] fp32 */
} vst_conv_weights;

typedef struct vst_block_weights {
    vst_conv_weights conv[3];
} vst_block_weights;

/* blocks[0..29] = stack.0..29, blocks[30..31] = channel_reduction.block_list.0..1 (host struct of
 * device pointers; models/RevResNet.py:190,192-201) */
typedef struct vst_net_weights {
    vst_block_weights blocks[VST_NUM_BLOCKS];
} vst_net_weights;

/* ---------------------------------------------------------------------------------------------
 * Layout glue (R-1..R-3 of SURVEY.md 8(a)): split/merge, injective_pad, squeeze/unsqueeze are
 * address arithmetic in the ZC state layout; only the NCHW boundary needs data movement.
 * state halves s1,s2: [B][H/4][W/4][256] fp32 each.
 * ------------------------------------------------------------------------------------------- */
/* x[B,C,H,W] (C<=16) -> s1 (channels C..15 zero), s2 = 0.   inj_pad.forward + split, RevResNet.py:212-214 */
int vst_pack_input(const float* x, float* s1, float* s2, int B, int C, int H, int W, void* stream);
/* s1 -> x[B,C,H,W]: merge + inj_pad.inverse, RevResNet.py:235-237 */
int vst_unpack_output(const float* s1, float* x, int B, int C, int H, int W, void* stream);
/* uint8 frame edge: frames_hwc[B,H,W,3] -> s1/s2 with ToTensor scaling (u8/255, image_transfer.py:167)
 * and back with mul(255).clamp(0,255).byte() truncation (image_transfer.py:217-218) */
int vst_pack_input_u8(const uint8_t* frames_hwc, float* s1, float* s2, int B, int H, int W, void* stream);
int vst_unpack_output_u8(const float* s1, uint8_t* frames_hwc, int B, int H, int W, void* stream);
/* Lab luminance-preserving post-process of the fork: out = lab2rgb(cat(L(content),
 * ab(clamp(stylized,0,1)))), all three [B,3,H,W] fp32 in [0,1]; any H, W >= 1 (no multiple-of-4 requirement).
 * project/image_style/vstnet.py:189-220, project/image_style/color.py:18-113.  out may alias stylized. */
int vst_lab_luminance(const float* content, const float* stylized, float* out, int B, int H, int W, void* stream);
/* merge + "spread" (unsqueeze x sp_steps), RevResNet.py:139-144 -> z[B,32,H,W] (sp=2) or [B,128,H/2,W/2] (sp=1) */
int vst_spread(const float* s1, const float* s2, float* z, int B, int H, int W, int sp_steps, void* stream);
/* inverse of vst_spread: squeeze x sp_steps + split, RevResNet.py:148-154 */
int vst_gather(const float* z, float* s1, float* s2, int B, int H, int W, int sp_steps, void* stream);

/* ---------------------------------------------------------------------------------------------
 * One coupling block, in place on the state (R-4/R-5; residual_block.forward RevResNet.py:96-104,
 * .inverse :106-116):   direction=+1:  dst += F(src)      direction=-1:  dst -= F(src)
 * `channel` in {16,64,256}, `stride` in {1,2} (stride 2: src is read at the finer resolution).
 * tmp must hold vst_block_tmp_bytes(B,H,W) bytes.
 * ------------------------------------------------------------------------------------------- */
size_t vst_block_tmp_bytes(int B, int H, int W);
int vst_block_apply(const vst_block_weights* w, int channel, int stride, int direction, int precision,
                    float* dst, const float* src, void* tmp, int B, int H, int W, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Whole passes (R-6/R-7): RevResNet._forward RevResNet.py:210-223 and ._inverse :225-239.
 * workspace: vst_pass_workspace_bytes(B,H,W).
 * ------------------------------------------------------------------------------------------- */
size_t vst_pass_workspace_bytes(int B, int H, int W);
/* images per internal sub-batch of vst_revnet_forward / _inverse for this shape (the passes keep a sub-batch's working set
 * inside the 256 MiB Infinity Cache); 1 = the passes run image by image.  < 0: VST_E_SHAPE. */
int vst_pass_sub_batch(int B, int H, int W);
int vst_revnet_forward(const vst_net_weights* w, const float* x, float* z, void* workspace,
                       int B, int C_in, int H, int W, int sp_steps, int precision, void* stream);
int vst_revnet_inverse(const vst_net_weights* w, const float* z, float* x, void* workspace,
                       int B, int C_out, int H, int W, int sp_steps, int precision, void* stream);

/* the same passes with the uint8 HWC frame edge fused into the boundary kernels (video_transfer.py:188,210-214) */
int vst_revnet_forward_u8(const vst_net_weights* w, const uint8_t* frames_hwc, float* z, void* workspace,
                          int B, int H, int W, int sp_steps, int precision, void* stream);
int vst_revnet_inverse_u8(const vst_net_weights* w, const float* z, uint8_t* frames_hwc, void* workspace,
                          int B, int H, int W, int sp_steps, int precision, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Packed code.  The reference's forward pass ends with merge + unsqueeze steps (RevResNet.py:139-144, :219-222) that only
 * permute (channel, pixel) pairs, and its inverse starts by undoing them (:148-154, :228-231); an unmasked cWCT
 * (cWCT.py:24-47, :206-262) is indifferent to the order of the pixels.  So the code can stay in the layout the coupling
 * blocks leave it in: `code` = float[B][2 halves][H/4][W/4][256] (the same number of floats as z), i.e. one row of N floats
 * per code pixel: sp_steps = 2 (photorealistic, N = 32): cell (h,w) of half i holds the rows of the pixels
 * (4h+2i+i', 4w+2j+j') in the order (j, i', j'); sp_steps = 1 (artistic, N = 128, z = [B,128,H/2,W/2]): the rows of the
 * pixels (2h+i, 2w+j), j = 0, 1.
 *   vst_revnet_encode[_u8] : forward pass without the spread; the state halves are written straight into `code`
 *                            (the same for both modes).
 *   vst_revnet_decode[_u8] : inverse pass without the gather.  `affines` (NULL, or float[B][N*N+N] as produced by
 *                            vst_cwct_factor): y = T x + t0 is applied to every row of image b first - the cWCT of
 *                            that frame - while the state is loaded (`code` itself is not modified).
 *   vst_code_to_z / vst_z_to_code : the permutation itself (z <-> code), for callers that want to look at z.
 *   vst_cwct_stats_code    : vst_cwct_stats (all pixels, no mask) of ONE image's code; same stats record.
 *   vst_cwct_apply_code    : y = T x + t0 on one image's code, out of place or in place (out may alias code).
 * workspace: vst_pass_workspace_bytes(1,H,W) for the passes (images are processed one at a time),
 * vst_cwct_stats_code_workspace_bytes(H,W,sp_steps) for the statistics.
 * ------------------------------------------------------------------------------------------- */
int vst_revnet_encode(const vst_net_weights* w, const float* x, float* code, void* workspace,
                      int B, int C_in, int H, int W, int precision, void* stream);
int vst_revnet_encode_u8(const vst_net_weights* w, const uint8_t* frames_hwc, float* code, void* workspace,
                         int B, int H, int W, int precision, void* stream);
int vst_revnet_decode(const vst_net_weights* w, const float* code, const float* affines, float* x, void* workspace,
                      int B, int C_out, int H, int W, int sp_steps, int precision, void* stream);
int vst_revnet_decode_u8(const vst_net_weights* w, const float* code, const float* affines, uint8_t* frames_hwc,
                         void* workspace, int B, int H, int W, int sp_steps, int precision, void* stream);
int vst_code_to_z(const float* code, float* z, int B, int H, int W, int sp_steps, void* stream);
int vst_z_to_code(const float* z, float* code, int B, int H, int W, int sp_steps, void* stream);
size_t vst_cwct_stats_code_workspace_bytes(int H, int W, int sp_steps);
int vst_cwct_stats_code(const float* code, int H, int W, int sp_steps, double* stats, void* workspace, void* stream);
int vst_cwct_apply_code(const float* code, float* out, int H, int W, int sp_steps, const float* affine, void* stream);
/* Masked transfer (cWCT.py:49-109) on ONE image's packed code (photorealistic codes, sp_steps = 2, only).  `mask_rows` = the label of every row, i.e. the [H][W] label
 * map in the code's pixel order (vst_mask_to_code; once per mask).  plan / max_slots / affines as in the vst_cwct_*_labels
 * calls below (vst_label_plan works on the label maps in any order); the apply and the decode take at most 8 slots
 * (VST_E_SHAPE otherwise: use the z route), the statistics any number.  Rows whose label has no slot keep their values. */
int vst_mask_to_code(const uint8_t* mask, uint8_t* mask_rows, int H, int W, void* stream);
size_t vst_cwct_stats_labels_code_workspace_bytes(int H, int W);
int vst_cwct_stats_labels_code(const float* code, int H, int W, const uint8_t* mask_rows, const void* plan, int max_slots,
                               double* stats, void* workspace, void* stream);
int vst_cwct_apply_labels_code(const float* code, float* out, int H, int W, const float* affines, const uint8_t* mask_rows,
                               const void* plan, int max_slots, void* stream);
int vst_revnet_decode_labels(const vst_net_weights* w, const float* code, const float* affines, const uint8_t* mask_rows,
                             const void* plan, int max_slots, float* x, void* workspace, int C_out, int H, int W,
                             int precision, void* stream);
int vst_revnet_decode_labels_u8(const vst_net_weights* w, const float* code, const float* affines, const uint8_t* mask_rows,
                                const void* plan, int max_slots, uint8_t* frame_hwc, void* workspace, int H, int W,
                                int precision, void* stream);

/* ---------------------------------------------------------------------------------------------
 * cWCT (C-1..C-6; models/cWCT.py).  Feature matrices are x[N][L] fp32 row-major (one NCHW image:
 * N channels, L = H*W).  `mask` (optional, may be NULL) is uint8[L]; with a mask only pixels whose
 * label == `label` take part.
 *
 * vst_cwct_stats  : mean and covariance  C = Xc Xc^T/(n-1)  (cWCT.py:138-144,153-157), two-level
 *                   shifted accumulation, final combine in fp64.  stats = double[1 + N + N*N]:
 *                   {n, mean[N], cov[N*N]}.
 * vst_cwct_factor : builds the affine map of one (content,style) pair from their stats:
 *                   Lc = chol(Cc), Ls_i = chol(Cs_i) with the cumulative-jitter retry of
 *                   cholesky_dec (cWCT.py:111-132), mixL = sum_i alpha_i Ls_i (+ alpha_c blend,
 *                   cWCT.py:231-254), T = mixL * Lc^-1, t0 = mix_mean - T*mean_c.
 *                   affine = float[N*N + N] {T, t0};  info = int[2 + n_styles], IN/OUT: on entry the minimum
 *                   number of retries to start from (zeros normally; the reference factors a whole
 *                   [B,N,N] batch at once, cWCT.py:122-128, so a failing sample jitters every sample: a
 *                   caller reproduces that by a second call with the batch maximum), on exit
 *                   {content retries, overflow flag, style retries...}.
 * vst_cwct_apply  : y[:,p] = T x[:,p] + t0  (cWCT.py:147,161-162 fused); with a mask only pixels
 *                   whose label matches are written (in place allowed: y may alias x).
 * ------------------------------------------------------------------------------------------- */
size_t vst_cwct_stats_workspace_bytes(int N, long L);
int vst_cwct_stats(const float* x, int N, long L, const uint8_t* mask, int label,
                   double* stats, void* workspace, void* stream);
int vst_cwct_factor(const double* content_stats, const double* const* style_stats_host_array,
                    const float* alphas_host, int n_styles, float alpha_c, float eps, int N,
                    float* affine, int* info, void* stream);
int vst_cwct_apply(const float* x, float* y, int N, long L, const float* affine,
                   const uint8_t* mask, int label, void* stream);
/* the same with the arithmetic chosen by the caller: VST_PREC_FP32 = exact-fp32 matrix-core / FMA kernels for every N,
 * shape and mask; otherwise (the default of vst_cwct_apply) unmasked N >= 64 applies with L % 64 == 0 run on bf16 MFMA
 * with split operands (3 products, ~1.5e-5 max-rel) and everything else is exact fp32 */
int vst_cwct_apply_prec(const float* x, float* y, int N, long L, const float* affine,
                        const uint8_t* mask, int label, int precision, void* stream);
/* ---- single-pass masked transfer (cWCT._transfer_seg, models/cWCT.py:49-109; compute_label_info :166-189) ----------------
 * vst_label_plan        : histograms both uint8 label maps ON THE DEVICE, applies the validity rule (count_c > 10, count_s > 10,
 *                         ratio < 100 both ways, cWCT.py:178) and gives the valid labels consecutive "slots" in increasing label
 *                         order (at most 32; more sets plan.overflow and keeps the content feature there).  plan = device buffer
 *                         of VST_LABEL_PLAN_BYTES: {int n_slots, overflow; int hist_c[256], hist_s[256]; u8 lut[256]; u8 slot_label[32]}.
 * vst_cwct_stats_labels : {n, mean, cov} of every slot in ONE pass over x (pixels of a tile are sorted by slot in LDS);
 *                         stats = double[32][1 + N + N*N].  N in {32, 64, 128}.  workspace: vst_cwct_labels_workspace_bytes.
 * vst_cwct_factor_labels: one workgroup per slot: affines[slot] = {T, t0} of (content slot, style slot); info = int[32][3].
 * vst_cwct_apply_labels : y[:,p] = T[slot(p)] x[:,p] + t0[slot(p)], y = x where the label has no slot; one pass (y may alias x).
 *                         precision VST_PREC_FP32: exact-fp32 MFMA (one sweep per slot present in a pixel group); otherwise,
 *                         when L % 64 == 0, bf16 MFMA with split operands (~1.5e-5): HBM-bound however the labels are mixed.
 * max_slots (1..32, 0 = 32) bounds the slots the launches cover when the host knows it (e.g. a plan reused over a clip);
 * nothing here synchronises with the host. */
#define VST_LABEL_PLAN_BYTES 2344
int vst_label_plan(const uint8_t* cmask, long Lc, const uint8_t* smask, long Ls, void* plan, void* stream);
size_t vst_cwct_labels_workspace_bytes(int N, long L);
int vst_cwct_stats_labels(const float* x, int N, long L, const uint8_t* mask, const void* plan, int max_slots,
                          double* stats, void* workspace, void* stream);
int vst_cwct_factor_labels(const double* content_stats, const double* style_stats, const void* plan, int max_slots,
                           float eps, int N, float* affines, int* info, void* stream);
int vst_cwct_apply_labels(const float* x, float* y, int N, long L, const float* affines, const uint8_t* mask,
                          const void* plan, int max_slots, int precision, void* stream);

/* Turns a statistics record into a "prefactored" one ({-(n+1), mean, chol(cov) with jitter retries}); a style that
 * is reused over many frames (video_transfer.py re-factors it per frame, :195-203) then costs no Cholesky in
 * vst_cwct_factor.  `out` may alias `stats`; info = int[1] retry count. */
int vst_cwct_prefactor(const double* stats, int N, float eps, double* out, int* info, void* stream);

/* ---- generic-architecture RevResNet ops -------------------------------------------------------------------------------------
 * models/RevResNet.py:166-201 accepts any nBlocks / nStrides / nChannels / mult / kernel / in_channel / hidden_dim / sp_steps; the
 * whole-pass entry points above implement the published architecture ([10,10,10] / [1,2,2] / [16,64,256], mult 4, kernel 3).
 * Any other architecture runs on these: plain NCHW fp32 tensors, exact fp32 FMA (the slow, complete path; csrc/generic.hip).
 *   vst_generic_conv : ReflectionPad2d((K-1)/2) + Conv2d(K, stride, bias=True) of residual_block.conv (:79-88), K odd <= 7,
 *                      stride 1 or 2; relu != 0 applies ReLU; old != NULL: out = old + sign * conv (the coupling of
 *                      residual_block.forward / .inverse, :96-116; out may alias old).  w = OIHW, out = [B,Cout,Ho,Wo],
 *                      Ho = (H + 2 pad - K) / stride + 1.
 *   vst_generic_squeeze / _unsqueeze : :34-43 (D, H, W = channels and size of the UNsqueezed tensor, H, W even).
 *   vst_generic_copy_channels : dst[b, d0 + k] = src[b, c0 + k], k < n  (split / merge / injective_pad, :8-31); vst_generic_zero. */
int vst_generic_conv(const float* x, const float* w, const float* bias, const float* old, float sign, int relu, float* out, int B,
                     int Cin, int Cout, int H, int W, int K, int stride, void* stream);
int vst_generic_squeeze(const float* x, float* y, int B, int D, int H, int W, void* stream);
int vst_generic_unsqueeze(const float* y, float* x, int B, int D, int H, int W, void* stream);
int vst_generic_copy_channels(const float* src, float* dst, int B, int C_src, int c0, int n, long HW, int C_dst, int d0,
                              void* stream);
int vst_generic_zero(float* dst, size_t n_floats, void* stream);

/* ---- fp64 cWCT: cWCT(use_double=True), models/cWCT.py:13-16,35-47,66,106,220,238,259 -------------------------------------
 * The reference converts the features to double and runs mean / covariance / Cholesky (with the same jitter schedule) / inverse
 * / both products in fp64, then converts back.  Same records as the fp32 calls (stats = double[1 + N + N*N], info as in
 * vst_cwct_factor) but: true fp64 two-pass statistics, an fp64 factorisation, affine = DOUBLE[N*N + N], and the apply
 * accumulates in fp64 (y = float(T double(x) + t0); y may alias x; with a mask only matching pixels are written).
 * A fidelity option, not a hot path (no script of the reference sets it): NCHW codes only, N in {16, 32, 64, 128}. */
size_t vst_cwct_stats_f64_workspace_bytes(int N, long L);
int vst_cwct_stats_f64(const float* x, int N, long L, const uint8_t* mask, int label, double* stats, void* workspace,
                       void* stream);
size_t vst_cwct_factor_f64_workspace_bytes(int N);
int vst_cwct_factor_f64(const double* content_stats, const double* const* style_stats_host_array, const float* alphas_host,
                        int n_styles, float alpha_c, float eps, int N, double* affine, int* info, void* workspace,
                        void* stream);
int vst_cwct_apply_f64(const float* x, float* y, int N, long L, const double* affine, const uint8_t* mask, int label,
                       void* stream);

/* ---------------------------------------------------------------------------------------------
 * Measurement hook (bench.py's live roofline figure): bracket every launch of one conv kernel class
 * with HIP events on the launch stream.  One profiling session at a time (begin/end and the launch
 * sites are serialised by a lock, so other host threads may keep launching while a session runs).
 *   vst_profile_begin(VST_KERNEL_ID(cin,cout,stride), max_records); ...run passes...;
 *   vst_profile_end(&total_ms, &launches)   (synchronises on the recorded events)
 * ------------------------------------------------------------------------------------------- */
#define VST_KERNEL_ID(cin, cout, stride) (((cin) << 16) | ((cout) << 4) | (stride))
#define VST_KERNEL_ALL (-1)        /* every hooked launch; read the session with vst_profile_end_table */
/* ids of the non-conv launches (one per C entry point) */
#define VST_KERNEL_PACK 1          /* vst_pack_input* (+ the folded block-0 constant) */
#define VST_KERNEL_UNPACK 2
#define VST_KERNEL_SPREAD 3
#define VST_KERNEL_GATHER 4
#define VST_KERNEL_CWCT_STATS 5
#define VST_KERNEL_CWCT_FACTOR 6
#define VST_KERNEL_CWCT_APPLY 7
#define VST_KERNEL_PRESPLIT 8      /* fp32 state -> split fp16 planes in front of the first 256-channel block (F16X2) */
int vst_profile_begin(int kernel_id, int max_records);
int vst_profile_end(double* total_ms, int* launches);
/* per-id totals of a VST_KERNEL_ALL (or single-id) session: ids[i], ms[i], launches[i] for i < *n_ids <= cap */
int vst_profile_end_table(int* ids, double* ms, int* launches, int cap, int* n_ids);

/* ---------------------------------------------------------------------------------------------
 * Process-wide tuning options (atomic; may be set at any time, a launch reads them when it is enqueued).
 *   VST_OPT_STAGE3_LEAN  1: the 256-channel convs of VST_PREC_BF16X3 (residual_block.conv of models/RevResNet.py:79-88 at
 *                        C = 256) run as half-CU workgroups (4 waves, 8 x 16 pixel tiles, 96 KB of LDS, <= 256 VGPRs) so that
 *                        the HBM-bound 16- / 64-channel kernels of ANOTHER frame on another stream share their CUs - for callers
 *                        that keep several frames in flight (video_transfer.py:160-214's loop run on HIP streams);
 *                        0: 8 waves on 16 x 16 tiles, one workgroup per CU (best for one frame at a time).  Results are
 *                        bit-identical.  Initial value: environment variable VST_LEAN (0 / 1), else VST_LEAN_DEFAULT.
 *   VST_OPT_STAGE3_PINGPONG  diagnostic builds only (-DVST_WITH_PINGPONG=1; VST_E_ARG in the shipped library): the same convs as
 *                        two wave groups that alternate between a matrix burst and a staging segment (csrc/conv.hip,
 *                        conv_pp_kernel: measured, bit-identical, not faster).
 * vst_set_option returns VST_E_ARG for an unknown option, vst_get_option the value (or VST_E_ARG).
 * ------------------------------------------------------------------------------------------- */
#define VST_OPT_STAGE3_LEAN 1
#define VST_OPT_STAGE3_PINGPONG 2
int vst_set_option(int option, int value);
int vst_get_option(int option);

#pragma GCC visibility pop
#ifdef __cplusplus
}
#endif
#endif /* VSTNET_H */
