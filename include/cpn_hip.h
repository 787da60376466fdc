/* libcpn_hip.so -- C ABI of the MI355X (gfx950) Contour Proposal Network inference path.
 *
 * The reference (FZJ-INM1-BDA/celldetection v0.4.9) is pure Python: the "FFI" this library replaces is the set of
 * PyTorch / torchvision operator calls on the CPN inference path.  Each entry point cites the reference call
 * site(s) it replaces (paths relative to the reference repository root).  The Python binding a maintainer would
 * add is a ctypes stub (see INTEGRATION.md and celldetection_amd/_lib.py).
 *
 * Conventions
 *   - plain pointers and sizes only; no torch types.  All device pointers are HIP device memory owned by the
 *     caller (e.g. PyTorch's caching allocator); the library allocates no device memory after plan creation
 *     except what the caller hands it as workspace.
 *   - every call enqueues on the caller's HIP stream (`stream` = hipStream_t cast to void*), no internal threads,
 *     no host synchronisation unless stated.
 *   - return value: 0 = ok, otherwise a negative CPN_E_* code or a positive hipError_t; cpn_last_error() returns a
 *     thread-local message.  No exceptions cross the ABI.
 */
#ifndef CPN_HIP_H
#define CPN_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CPN_ABI_VERSION 14

#define CPN_E_INVALID (-1)
#define CPN_E_UNSUPPORTED (-2)
#define CPN_E_WORKSPACE (-3)

const char *cpn_last_error(void);
int cpn_abi_version(void);

/* ------------------------------------------------------------------------------------------------------------
 * Conv-graph plan: the backbone + head convolution stack.
 * Replaces CPNCore.forward (celldetection/models/cpn.py:238-283): backbone(inputs) [models/unet.py:296-304,178-249;
 * models/fpn.py:180-185; models/resnet.py:265-297 + torchvision block forwards], the four ReadOut heads
 * (models/commons.py:461-511) and Normalize's range assert (models/commons.py:694-700).
 * ---------------------------------------------------------------------------------------------------------- */

/* activation tensors of the graph (NHWC bf16, channel count padded to a multiple of 32) */
typedef struct {
    int32_t channels;  /* padded channel count (multiple of 32; 64 for CPN_PRECISION_FP8 plans)   */
    int32_t down;      /* nominal down-sampling factor (1,2,4,...,32); actual sizes are propagated per input size */
    float scale;       /* CPN_PRECISION_FP8: value of one e4m3 code unit of this tensor (> 0); a NEGATIVE scale marks a tensor
                        * stored as bf16 values inside an fp8 plan -- the partial sums between the PHASE and the LATERAL op of
                        * a sub-pixel triple (ABI 11).  Other precisions: unused */
} cpn_tensor_desc;

/* CPN_OP_CONV_DEFERRED (score-gated heads, cpn_sparse_heads below): a fused ReadOut head conv that cpn_plan_run does NOT
 * execute -- its weights are packed and its output size is reported like a CPN_OP_CONV's, and its source tensor stays
 * intact in the workspace until the end of the run (cpn_plan_tensor_info locates it). */
/* CPN_OP_INPUT_STEM / CPN_OP_STEM7 (bf16 and fp8 plans; csrc/stem.hip): the ResNet stem `body.0` = Conv2d(in_channels <= 4 -> 32 | 64
 * output channels after padding, 7x7, stride 2, pad 3) + BN + ReLU (celldetection/models/resnet.py:274-284) on a dedicated
 * layout: CPN_OP_INPUT_STEM converts the input to bf16 [N][H + 6][W + 8][4] with a zero border inside the storage of its
 * dst tensor (which needs (H + 6) * (W + 8) * 4 <= H * W * channels elements), CPN_OP_STEM7 reads it with weights
 * [7][cout_b][32] bf16 at weight_offset (filter row, output channel, (kx 0..7, c 0..3); kx = 7 and c >= in_channels zero).
 * Both carry `alt` = 2 and stand next to the generic CPN_OP_INPUT / CPN_OP_CONV pair (`alt` = 1): the executor runs the
 * fast pair at every input size the layout fits into the tensor, the generic pair otherwise. */
/* CPN_OP_CONV_PAIR (bf16 plans; csrc/conv_pair.hip): the head of a grouped bottleneck block -- conv1 1x1 + BN + ReLU ->
 * conv2 3x3 (groups, stride 1 | 2, pad 1) + BN + ReLU, torchvision Bottleneck.forward as built by
 * celldetection/models/resnet.py:88-116,119-193 for the ResNeXt encoders -- as ONE kernel: conv1's output stays in LDS.
 * The op stands directly BEHIND the two CPN_OP_CONV ops it restates (conv1 at index i - 2, conv2 at i - 1) and adds no
 * weights: src0 = conv1's source, dst = conv2's destination, cin_b / cout_b = conv1's input / output channels,
 * weight_offset / bias_offset = conv1's, fuse_weight_offset / fuse_bias_offset = conv2's, bundles = conv2's bundles,
 * fuse_cout = conv2's channels per bundle (32 | 64).  The executor runs it INSTEAD of the two convs at every input size at
 * which the kernel applies -- feature maps exactly 16 | 32 | 64 pixels wide as full-width row strips (conv1 output channels
 * a multiple of 256; 128 at width 64), any other width > 32 as generic 16 x 32 tiles (multiple of 128, 32-channel bundles);
 * a stride-2 conv2 (`stride` = 2: the first block of a stage) on generic tiles at any width >= 32 -- and batch x tiles x
 * slabs give >= 192 workgroups; the two convs run otherwise.  Same operands and per-conv
 * rounding (bf16 activations between the two convs) as the unfused pair; conv1 is recomputed on one halo row above and
 * below each 8-row strip. */
/* CPN_OP_CONV_BRIDGE (bf16 plans, ABI 11; csrc/conv_igemm.hip MODE_BR): the bridge level of the ResNet-UNets -- TwoConvNormRelu
 * (bias-free convs, celldetection/models/unet.py:92-107; commons.py:120-149) over the x2 nearest-upsampled 64-channel map
 * (unet.py:213-217, `scale_factor=2`) -- as ONE kernel: the first conv runs as its CPN_SUBPIXEL_SCATTER form (four 2x2 phase convs
 * + bias + ReLU) on the halo tile of the second conv's workgroup and lands in LDS, the full-resolution tensor between the two
 * convs is neither written nor read.  The op stands directly BEHIND the two CPN_OP_CONV ops it restates (the SCATTER conv at
 * index i - 2, the 3x3 conv at i - 1) and adds no weights: src0 = the scatter conv's source (32 | 64 channels), dst / res /
 * res_up / act = the 3x3 conv's, cin_b = the scatter conv's input channels, cout_b = 64, weight_offset / bias_offset = the
 * scatter conv's, fuse_weight_offset / fuse_bias_offset = the 3x3 conv's.  The executor runs it INSTEAD of the two convs
 * wherever the output is at least 16 x 32 pixels and nothing else reads the tensor in between (CPN_BRIDGE=0 in the
 * environment: never); same operands, K order and rounding points as the two launches. */
enum { CPN_OP_INPUT = 0, CPN_OP_CONV = 1, CPN_OP_MAXPOOL = 2, CPN_OP_BILINEAR = 3, CPN_OP_CONV_DEFERRED = 4,
       CPN_OP_INPUT_STEM = 5, CPN_OP_STEM7 = 6, CPN_OP_CONV_PAIR = 7, CPN_OP_CONV_BRIDGE = 8, CPN_OP_ACT = 9 };
/* 4..12 (ABI 11): hidden activations of the ReadOut heads other than ReLU (`head_activation*`, celldetection/models/cpn.py:183-233), as
 * the torch.nn modules of those names compute them with default arguments.  They are ops of their own -- CPN_OP_ACT: dst = act(src0),
 * elementwise on an NHWC tensor of any precision -- between the head's k x k conv (act NONE, not fused) and its 1x1 conv; conv ops
 * take CPN_ACT_NONE .. CPN_ACT_TANH_SCALED only */
enum { CPN_ACT_NONE = 0, CPN_ACT_RELU = 1, CPN_ACT_SIGMOID = 2, CPN_ACT_TANH_SCALED = 3, CPN_ACT_LEAKY_RELU = 4, CPN_ACT_SILU = 5,
       CPN_ACT_GELU = 6, CPN_ACT_ELU = 7, CPN_ACT_TANH = 8, CPN_ACT_HARDSWISH = 9, CPN_ACT_MISH = 10, CPN_ACT_SELU = 11,
       CPN_ACT_SOFTPLUS = 12 };
/* Sub-pixel decomposition of a k = 3 conv over cat(lateral, nearest-x2-upsampled top-down map) -- the first conv of every
 * GeneralizedUNet decoder level (celldetection/models/unet.py:213-224).  Output pixel (2i+py, 2j+px) sees the upsampled
 * map through 2 x 2 distinct low-resolution pixels only, so that part of the conv is FOUR 2 x 2 convs on the
 * low-resolution map (4/9 of the multiply-accumulates, tap sums rounded to bf16 once).  In a plan such a conv is a
 * triple of consecutive ops:
 *   CPN_SUBPIXEL_HEAD     the conv as the reference states it (virtual concat + nearest resize in the loader)
 *   CPN_SUBPIXEL_PHASE    src0 = top-down map, kh = kw = 2, `bundles` = 4 output phases (py, px) that all read the SAME
 *                         cin_b input channels with padding (pad - py, pad - px); dst = [h/2, w/2, 4 * cout_b] partial
 *                         sums (no bias, no activation), phase-major channels
 *   CPN_SUBPIXEL_LATERAL  src0 = lateral, res = the phase tensor read pixel-shuffled (res_up = 2), bias + activation,
 *                         dst = the HEAD op's dst
 * The executor runs PHASE + LATERAL when the lateral is exactly twice the top-down map's size and HEAD otherwise
 * (any other ratio: PyTorch's nearest index does not decompose).
 *   CPN_SUBPIXEL_SCATTER  a k = 3 conv whose ONLY source is a x2-upsampled map (`scale_factor=2`: always exact; the
 *                         bridge levels of GeneralizedUNet, unet.py:100-107,213-217) as a single op: the four 2 x 2 phase
 *                         convs (kh = kw = 2, bundles = 4 sharing cin_b input channels and ONE bias of cout_b entries)
 *                         + bias + activation, each phase written to its pixels (2i+py, 2j+px) of the [2h, 2w, cout_b]
 *                         destination.  Replaces the conv it restates (no HEAD op).
 * Bilinear counterpart -- the k x k ReadOut conv over the x2 BILINEAR-resized feature map in front of the FPN models'
 * refinement head (celldetection/models/cpn.py:277-278 + commons.py:461-511; fused ReadOut heads with k = 3 (mod 4), i.e.
 * 3 or the default 7).  For an exact x2 the resized map is a fixed linear filter of the low-resolution map (up[2i] =
 * .25 x[i-1] + .75 x[i], up[2i+1] = .75 x[i] + .25 x[i+1] per axis), so the conv is FOUR k2 x k2 convs on the low-resolution
 * map, k2 = (k + 3) / 2 (25 instead of 49 taps per output pixel for k = 7; tap sums formed in float64, rounded to bf16
 * once) -- wherever the conv window does not reach beyond the resized image.  A triple of consecutive ops again:
 *   CPN_SUBPIXEL_BL_HEAD   the conv as the reference states it (bf16 plans: up0 == 2, bilinear resize in the loader; fp8
 *                          plans: up0 == 0, src0 = the output of a CPN_OP_BILINEAR op of its own)
 *   CPN_SUBPIXEL_BL_PHASE  src0 = the low-resolution map (no resize; fp8 plans: the map in FRONT of that resize op), kh = kw =
 *                          k2, pad = k2 / 2, `bundles` = 4 output
 *                          phases sharing cin_b input channels, ONE bias and the fused tail; phase (py, px) writes pixel
 *                          (2i + py, 2j + px) of the external output for the low-resolution pixels i in [F/2, h - F/2)
 *   CPN_SUBPIXEL_BL_FRAME  the HEAD op restricted to the frame of F full-resolution pixels along every edge (F = 4 for
 *                          k = 7, 2 for k = 3), where the conv's zero padding cuts the window
 *                          (launched on the tiles that reach into the frame only; per tile row that crosses the interior ONE
 *                          wrap tile = output columns W - 16 .. W - 1 and 0 .. 15).  A CPN_OP_BILINEAR op flagged
 *                          CPN_SUBPIXEL_BL_FRAME feeds such a triple: when PHASE + FRAME run it writes only the ring of its
 *                          output the frame's windows reach (F + k / 2 pixels from the border)
 * The executor runs PHASE + FRAME when the input is exactly twice the feature map's size and the tile-granular frame leaves
 * a gain (CPN_BLPHASE=0 / 2 in the environment: never / wherever exact), HEAD otherwise. */
enum { CPN_SUBPIXEL_NONE = 0, CPN_SUBPIXEL_HEAD = 1, CPN_SUBPIXEL_PHASE = 2, CPN_SUBPIXEL_LATERAL = 3,
       CPN_SUBPIXEL_SCATTER = 4, CPN_SUBPIXEL_BL_HEAD = 5, CPN_SUBPIXEL_BL_PHASE = 6, CPN_SUBPIXEL_BL_FRAME = 7 };
enum { CPN_OUT_SCORES = 0, CPN_OUT_LOCATIONS = 1, CPN_OUT_FOURIER = 2, CPN_OUT_REFINEMENT = 3, CPN_OUT_UNCERTAINTY = 4,
       CPN_NUM_OUTPUTS = 5 };

typedef struct {
    int32_t op;               /* CPN_OP_*                                                                   */
    int32_t src0, src1, res;  /* tensor ids (-1 = none). src1: second source of a virtual channel concat      */
    int32_t dst;              /* tensor id, or -1 when the op writes an external fp32 NCHW output             */
    int32_t up0, up1, res_up; /* 1: source / residual is nearest-resized (PyTorch 'nearest': floor(dst*in/out)) to the
                               * size of the other concat source / of the output; a lone up0 source: exact x2.
                               * res_up == 2: the residual is a CPN_SUBPIXEL_PHASE tensor [h/2, w/2, 4 * C] read
                               * pixel-shuffled: out(y, x, c) += res(y >> 1, x >> 1, ((y & 1) * 2 + (x & 1)) * C + c).
                               * up0 == 2: src0 is read through a BILINEAR resize (align_corners=False) to the input
                               * size H x W (cpn.py:277-278; k x k stride-1 single-source convs of bf16 / fp8 plans) */
    int32_t c0_used;          /* channels of the concat taken from src0 (multiple of 32)                      */
    int32_t kh, kw, stride, pad;
    int32_t bundles, cin_b, cout_b; /* grouped convs run as `bundles` dense convs of cin_b -> cout_b channels  */
    int64_t weight_offset;    /* byte offset into the packed weight blob: [bundle][(cin_b/32)*kh*kw items, + one
                               * all-zero item when that count is odd][cout_b][32] bf16 (chunk-major, tap-minor)  */
    int64_t bias_offset;      /* float offset into the bias blob, -1 = no bias                                 */
    int32_t act;              /* CPN_ACT_*                                                                    */
    float act_scale;
    int32_t out_index;        /* dst == -1: CPN_OUT_* index of the external output                            */
    int32_t cout_real;        /* dst == -1: real number of output channels                                    */
    int32_t dst_coff;         /* channel offset inside dst                                                    */
    int32_t in_channels;      /* CPN_OP_INPUT: real input channels                                            */
    /* fused ReadOut tail (dst == -1 and fuse_cout > 0): conv -> act -> bf16 -> 1x1 conv [32][cout_b] bf16 at
     * fuse_weight_offset (+ fuse_bias_offset, -1 = none) -> fuse_act -> fp32 NCHW output `out_index` with fuse_cout
     * channels; requires bundles == 1 and cout_b in {32, 64, 128, 256} */
    int64_t fuse_weight_offset, fuse_bias_offset;
    int32_t fuse_cout, fuse_act;
    float fuse_act_scale;
    int32_t mult_offset;      /* CPN_PRECISION_FP8: float offset into the bias blob of the per-output-channel
                               * multipliers (weight scales), -1 = none; unused by the other precisions        */
    int32_t subpixel;         /* CPN_SUBPIXEL_* (bf16 plans)                                                    */
    int32_t alt;              /* 0: always runs; 1 / 2: generic / fast member of the stem alternatives (see above)  */
} cpn_op_desc;

typedef struct cpn_plan cpn_plan;

/* precision of a plan: bf16 activations/weights on MFMA with fp32 accumulation (the performance path), or an fp32
 * verification path (fp32 activations/weights/FMA on the vector ALUs; weights packed [bundle][kh*kw][cin_b][cout_b]
 * fp32, weight_offset in bytes of that blob; no fused heads) used to check the whole path against the reference's
 * fp32 CPU forward at 1e-4 */
enum { CPN_PRECISION_BF16 = 0, CPN_PRECISION_F32 = 1, CPN_PRECISION_FP8 = 2 };
/* CPN_PRECISION_FP8 (groundwork for BASELINE.json configs[4], no reference counterpart): activations are OCP e4m3
 * codes with one static scale per tensor (cpn_tensor_desc.scale, taken from a bf16 run through cpn_plan_run_stats),
 * weights e4m3 codes with one scale per output channel and the input-tensor scale folded in (see cpn_conv2d_fp8),
 * accumulation in fp32 on v_mfma_scale_f32_32x32x64_f8f6f4 (2x the bf16 MFMA rate); the fused ReadOut tails stay
 * bf16, the head outputs fp32. */

/* Creates a plan (host-side object; copies the descriptors).  `weights` / `bias` are DEVICE pointers to the packed
 * blobs and must stay alive as long as the plan. */
int cpn_plan_create(cpn_plan **plan, const cpn_tensor_desc *tensors, int32_t n_tensors, const cpn_op_desc *ops,
                    int32_t n_ops, const void *weights, size_t weight_bytes, const float *bias, size_t bias_count,
                    int32_t precision);
void cpn_plan_destroy(cpn_plan *plan);
/* Workspace (activation arena, liveness-planned) needed for a batch of N inputs of H x W.  Any H, W the graph can
 * digest: tensor sizes are propagated op by op with the reference modules' rules (conv / max-pool floor((in+2p-k)/s)+1;
 * top-down maps nearest-resized to the lateral's size, models/unet.py:213-217 and torchvision FPN; features
 * bilinear-resized to the input size, models/cpn.py:277-278); too small an input returns CPN_E_INVALID. */
int64_t cpn_plan_workspace_bytes(cpn_plan *plan, int32_t N, int32_t H, int32_t W);
/* Spatial size (h, w) of the external output CPN_OUT_* for an H x W input (0 x 0: the plan has no such output), and
 * the element count per image of the largest activation tensor (callers split batches at 2^31 elements). */
int cpn_plan_output_dims(cpn_plan *plan, int32_t H, int32_t W, int32_t out_index, int32_t *h, int32_t *w);
int64_t cpn_plan_max_tensor_elements(cpn_plan *plan, int32_t H, int32_t W);
/* Location of activation tensor `tensor` inside the workspace of a (N, H, W) run: byte offset, spatial size and channel
 * stride (padded channel count; NHWC, bf16 / e4m3 / fp32 by plan precision).  Only tensors that are live at the end of the
 * run may be read afterwards -- the sources of CPN_OP_CONV_DEFERRED ops are. */
int cpn_plan_tensor_info(cpn_plan *plan, int32_t N, int32_t H, int32_t W, int32_t tensor, int64_t *byte_offset,
                         int32_t *h, int32_t *w, int32_t *channel_stride);
/* 2*MAC FLOPs executed by the MFMA loops for that shape (includes channel/tile padding). */
double cpn_plan_executed_flops(cpn_plan *plan, int32_t N, int32_t H, int32_t W);

/* Runs the conv graph.  input: fp32 (dtype 0) or uint8 (dtype 1, scaled by 1/255) NCHW [N,C,H,W].
 * outputs[CPN_OUT_*] (CPN_NUM_OUTPUTS pointers; unused ones may be NULL): fp32 NCHW device buffers: scores
 * [N,1,h,w] with the sigmoid applied (binary) or raw logits [N,classes,h,w] (multi-class, cpn.py:583-585), locations
 * [N,2,h,w], fourier [N,4*order,h,w], refinement [N,2*buckets,H,W] (tanh*margin applied), uncertainty [N,4,h,w]
 * (sigmoid applied; only for plans with an uncertainty head, cpn.py:209-221).
 * range_flag: device int32, zeroed by the caller; set to 1 if an input value lies outside [0,1]
 * (the caller raises the reference's AssertionError, models/commons.py:696-697). */
int cpn_plan_run(cpn_plan *plan, const void *input, int32_t in_dtype, int32_t N, int32_t H, int32_t W,
                 void *workspace, int64_t workspace_bytes, float *const *outputs, int32_t *range_flag, void *stream);

/* Calibration run of a CPN_PRECISION_BF16 plan: like cpn_plan_run, additionally writes max|x| of every activation
 * tensor to absmax[tensor id] (device float[n_tensors], zeroed by the caller). */
int cpn_plan_run_stats(cpn_plan *plan, const void *input, int32_t in_dtype, int32_t N, int32_t H, int32_t W,
                       void *workspace, int64_t workspace_bytes, float *const *outputs, int32_t *range_flag,
                       float *absmax, void *stream);

/* Profiling variant: brackets every op with HIP events on `stream`, synchronises, and returns the per-op duration
 * (ms, op_ms[cpn_plan_num_ops]) and the executed MFMA FLOPs per op (op_flops, may be NULL). */
int cpn_plan_num_ops(cpn_plan *plan);
int cpn_plan_run_timed(cpn_plan *plan, const void *input, int32_t in_dtype, int32_t N, int32_t H, int32_t W,
                       void *workspace, int64_t workspace_bytes, float *const *outputs, int32_t *range_flag,
                       void *stream, float *op_ms, double *op_flops);

/* Single convolution (testing / building blocks); same semantics as one CPN_OP_CONV. */
int cpn_conv2d(const cpn_op_desc *op, const void *src0, int32_t c0_stride, const void *src1, int32_t c1_stride,
               const void *res, int32_t res_stride, void *dst, int32_t dst_stride, int32_t N, int32_t Hin, int32_t Win,
               const void *weights, const float *bias, void *stream);
/* fp8 (OCP e4m3) variant of cpn_conv2d on v_mfma_scale_f32_32x32x64_f8f6f4 (BASELINE.json configs[4] groundwork; no
 * reference counterpart): activations and residual are e4m3 codes (NHWC, channel strides multiples of 64, value =
 * code * tensor scale), weights are e4m3 codes packed [bundle][cin/64][kh*kw (+1 zero slab when the item count is
 * odd)][cout][64] with the input-tensor scale folded in before quantisation; value = acc * mult[cout] + bias
 * (+ residual code * res_scale) -> act -> e4m3(value * out_inv_scale) (NHWC outputs) or the fp32 head outputs. */
int cpn_conv2d_fp8(const cpn_op_desc *op, const void *src0, int32_t c0_stride, const void *src1, int32_t c1_stride,
                   const void *res, int32_t res_stride, void *dst, int32_t dst_stride, int32_t N, int32_t Hin,
                   int32_t Win, const void *weights, const float *bias, const float *mult, float res_scale,
                   float out_inv_scale, void *stream);
/* ResNet stem fast path (see CPN_OP_INPUT_STEM / CPN_OP_STEM7): input conversion into the padded 4-channel layout
 * (dst: (H + 6) * (W + 8) * 4 bf16 per image) and the 7x7 stride-2 conv + bias + ReLU from it (`op`: a CPN_OP_STEM7
 * descriptor; dst NHWC bf16 [N][(H - 1) / 2 + 1][(W - 1) / 2 + 1][dst_stride], or -- out_inv_scale > 0, the output tensor
 * of an fp8 plan -- OCP e4m3 codes of value * out_inv_scale; the stem computes in bf16 on the bf16 input either way). */
int cpn_convert_input_stem(const void *src, int32_t in_dtype, void *dst, int32_t N, int32_t C, int32_t H, int32_t W,
                           int32_t *range_flag, void *stream);
int cpn_stem7(const cpn_op_desc *op, const void *src, void *dst, int32_t dst_stride, int32_t N, int32_t H, int32_t W,
              const void *weights, const float *bias, float out_inv_scale, void *stream);
/* Fused bottleneck head (see CPN_OP_CONV_PAIR; `op`: such a descriptor, `weights` / `bias`: the blobs its four offsets
 * index): src NHWC bf16 [N][H][W][c_stride] -> dst NHWC bf16 [N][Ho][Wo][dst_stride], channels [0, cout_b), Ho = (H - 1) /
 * op->stride + 1.  Returns
 * CPN_E_UNSUPPORTED when W < 16 or between 17 and 31, or cout_b is no multiple of the slab width (run the two convs). */
int cpn_conv_pair(const cpn_op_desc *op, const void *src, int32_t c_stride, void *dst, int32_t dst_stride, int32_t N,
                  int32_t H, int32_t W, const void *weights, const float *bias, void *stream);
/* Fused bridge level (see CPN_OP_CONV_BRIDGE; `op`: such a descriptor, `weights` / `bias`: the blobs its four offsets point
 * into): src = [N, H, W, c_stride] bf16 low-resolution map, dst = [N, 2H, 2W, dst_stride] bf16; res (optional) = a residual of
 * the 3x3 conv at the output's size.  CPN_E_UNSUPPORTED when the kernel's shape does not apply (output below 16 x 32 pixels). */
int cpn_conv_bridge(const cpn_op_desc *op, const void *src, int32_t c_stride, const void *res, int32_t res_stride, void *dst,
                    int32_t dst_stride, int32_t N, int32_t H, int32_t W, const void *weights, const float *bias, void *stream);
int cpn_maxpool2d(const void *src, void *dst, int32_t N, int32_t Hin, int32_t Win, int32_t C, int32_t k, int32_t stride,
                  int32_t pad, void *stream);
int cpn_resize_bilinear(const void *src, void *dst, int32_t N, int32_t Hin, int32_t Win, int32_t Hout, int32_t Wout,
                        int32_t C, void *stream);
int cpn_convert_input(const void *src, int32_t in_dtype, void *dst, int32_t N, int32_t C, int32_t H, int32_t W,
                      int32_t Cpad, int32_t *range_flag, void *stream);

/* ------------------------------------------------------------------------------------------------------------
 * Proposal extraction + contour decode.
 * ---------------------------------------------------------------------------------------------------------- */

/* Order-preserving stream compaction of (scores > thresh), replaces `torch.where(fg_mask)`
 * (celldetection/models/cpn.py:579,616-620): writes the linear indices (b*h*w + y*w + x, ascending = (b,y,x)
 * row-major order) of the selected pixels to `indices` (capacity N*h*w int32) and the per-image counts to
 * `counts` (N+1 int32: counts[b] = proposals of image b, counts[N] = total).  `workspace`: cpn_compact_workspace_bytes. */
int64_t cpn_compact_workspace_bytes(int32_t N, int32_t h, int32_t w);
int cpn_compact(const float *scores, int32_t N, int32_t h, int32_t w, float thresh, int32_t *indices, int32_t *counts,
                void *workspace, void *stream);

/* Fused gather + rel->abs location + Fourier-to-contour synthesis + rescale + local refinement + clamp + boxes
 * (+ per-image offsets).  Replaces celldetection/models/cpn.py:613-702 i.e. rel_location2abs_location
 * (ops/cpn.py:15-41), advanced-index gathers (cpn.py:621-628), fouriers2contours (ops/cpn.py:44-95), scale_contours /
 * scale_fourier (ops/cpn.py:106-165), local_refinement (cpn.py:63-85), clamp + min/max boxes (cpn.py:661-670) and the
 * offsets add (cpn.py:695-702).
 *   indices[P]           from cpn_compact
 *   scores [N,1,h,w], locations [N,2,h,w], fourier [N,4*order_total,h,w], refinement [N,2,H,W] or NULL (fp32 NCHW)
 *   order <= order_total (cpn.py:597-598 "changed order"), samples = S, iterations = refinement iterations
 *   cos_table/sin_table [order][samples] fp32 device (built by the host exactly like ops/cpn.py:69-78)
 *   offsets: float [N,2] (xy) device or NULL (the reference adds int64 offsets to fp32 tensors = fp32 add of the
 *   converted value); when no refinement runs, contours and contour_proposals are ONE tensor in the reference and
 *   receive the offset twice (cpn.py:655-656,697-699) -- reproduced
 *   buckets = refinement_buckets (cpn.py:72-82): 1 = plain map; > 1: refinement is [N,2*buckets,H,W] and
 *   bucket_index / bucket_weight are [3][samples] device tables (bucket number and blend weight of the three
 *   neighbouring buckets of every sample, built by the host like resolve_refinement_buckets, ops/cpn.py:238-255)
 * outputs (device, row-major): contours [P,S,2], proposals [P,S,2], boxes [P,4], out_scores [P], out_locations [P,2],
 *   out_fourier [P,order,4], batch_index [P] int32. */
int cpn_decode(const int32_t *indices, int32_t P, const float *scores, const float *locations, const float *fourier,
               const float *refinement, int32_t N, int32_t h, int32_t w, int32_t H, int32_t W, int32_t order_total,
               int32_t order, int32_t samples, int32_t iterations, const float *cos_table, const float *sin_table,
               const float *offsets, float *contours, float *proposals, float *boxes, float *out_scores,
               float *out_locations, float *out_fourier, int32_t *batch_index, int32_t buckets,
               const int32_t *bucket_index, const float *bucket_weight, void *stream);

/* cpn_decode on GATHERED head values: locations [P,2] and fourier [P,4*order_total] hold the head outputs of proposal p
 * (what cpn_sparse_heads writes) instead of dense maps; everything else as cpn_decode. */
int cpn_decode_gathered(const int32_t *indices, int32_t P, const float *scores, const float *locations,
                        const float *fourier, const float *refinement, int32_t N, int32_t h, int32_t w, int32_t H,
                        int32_t W, int32_t order_total, int32_t order, int32_t samples, int32_t iterations,
                        const float *cos_table, const float *sin_table, const float *offsets, float *contours,
                        float *proposals, float *boxes, float *out_scores, float *out_locations, float *out_fourier,
                        int32_t *batch_index, int32_t buckets, const int32_t *bucket_index, const float *bucket_weight,
                        void *stream);

/* Score-gated ReadOut heads (bf16 plans): evaluates two fused ReadOut heads (cpn_op_desc with fuse_cout > 0: k x k
 * stride-1 'same' conv + BN + ReLU + 1x1 conv, celldetection/models/commons.py:461-511) that read the same NHWC bf16
 * feature tensor [N,h,w,channel_stride] ONLY at the P pixels `indices` (cpn_compact's output) and writes out_a [P,
 * op_a->fuse_cout] and out_b [P, op_b->fuse_cout] (fp32): bit-identical to the values the dense heads produce at those
 * pixels.  CPN.forward reads the location / Fourier maps at the proposals only (celldetection/models/cpn.py:613-637); the
 * dense maps are (N h w) / P times more work.  `weights` / `bias`: the plan's packed blobs (the ops' offsets index them).
 * Both heads must share source, kernel size and hidden width (128 or 256). */
int cpn_sparse_heads(const cpn_op_desc *op_a, const cpn_op_desc *op_b, const void *features, int32_t channel_stride,
                     int32_t N, int32_t h, int32_t w, const int32_t *indices, int32_t P, const void *weights,
                     const float *bias, float *out_a, float *out_b, void *stream);

/* Standalone pieces of the decode (parity tests, reference ops API celldetection/ops/cpn.py). */
int cpn_fouriers2contours(const float *fourier, const float *locations, int32_t P, int32_t order, int32_t samples,
                          const float *cos_table, const float *sin_table, float *contours, void *stream);
int cpn_local_refinement(float *contours /* in/out [P,S,2] */, const int32_t *batch_index, int32_t P, int32_t samples,
                         const float *refinement, int32_t N, int32_t H, int32_t W, int32_t iterations,
                         int32_t buckets, const int32_t *bucket_index, const float *bucket_weight, void *stream);

/* Score variants of CPN.forward.
 * cpn_class_scores (multi-class CPNs, celldetection/models/cpn.py:583-585,631-632): softmax over the C logit planes
 * [N,C,h,w], optional score bounds lower/upper [N,1,h,w] applied to every class plane (_apply_score_bounds,
 * cpn.py:118-123), classes = argmax (first maximum), selected = probability of that class, foreground = 1.0 where
 * classes > 0 else 0.0 (feeds cpn_compact with thresh 0.5).  probs [N,C,h,w] may be NULL.
 * cpn_certainty_mask (cpn.py:617-618): out = scores where mean_c(uncertainty[N,C,h,w]) < limit, else -1
 * (limit = 1 - certainty_thresh), so that the thresholding in cpn_compact applies `fg_mask &= ...`.
 * cpn_gather_channels (cpn.py:634-636): out[p][c] = map[b][c][y][x] for the pixel index indices[p] of cpn_compact. */
int cpn_class_scores(const float *logits, int32_t N, int32_t C, int32_t h, int32_t w, const float *lower,
                     const float *upper, float *probs, float *selected, int32_t *classes, float *foreground,
                     void *stream);
int cpn_certainty_mask(const float *scores, const float *uncertainty, int32_t N, int32_t C, int32_t h, int32_t w,
                       float limit, float *out, void *stream);
int cpn_gather_channels(const float *map, const int32_t *indices, int64_t P, int32_t C, int32_t h, int32_t w,
                        float *out, void *stream);

/* ------------------------------------------------------------------------------------------------------------
 * Box NMS.  Replaces torch.ops.torchvision.nms as called from batched_box_nmsi (celldetection/ops/cpn.py:189-227)
 * and the slide-level NMS (celldetection_scripts/cpn_inference.py:405-408,426): greedy, stable descending-score
 * order, suppress iff inter/(area_i+area_j-inter) > thresh (NaN never suppresses), output = kept indices in that
 * order.  Segmented: boxes/scores hold `nseg` consecutive segments (images) [seg_offsets[s], seg_offsets[s+1]);
 * segments are independent.  seg_offsets_host: host int64[nseg+1]; seg_offsets_dev: same values on the device.
 * keep (int64 [P], device): for each segment the kept indices (GLOBAL indices into boxes/scores, i.e. segment start
 * + index within the segment; descending score) are written from position seg_offsets[s]; keep_counts
 * (int32 [nseg], device) receives the number kept per segment.
 * ---------------------------------------------------------------------------------------------------------- */
int64_t cpn_nms_workspace_bytes(int64_t P, int64_t max_segment, int32_t nseg);
int cpn_nms(const float *boxes, const float *scores, int64_t P, const int64_t *seg_offsets_host,
            const int64_t *seg_offsets_dev, int32_t nseg, float thresh, int64_t *keep, int32_t *keep_counts,
            void *workspace, int64_t workspace_bytes, void *stream);

/* Box voting of the multi-model ensemble path (get_iou_voting / filter_by_box_voting, celldetection/ops/boxes.py:52-83,
 * called from celldetection_scripts/cpn_inference.py:419-423): votes[i] = sum_j iou(i,j) * (iou(i,j) > thresh), IoU
 * as torchvision.ops.box_iou; every box votes for itself (smallest vote 1).  boxes [P,4] fp32 (16-byte aligned). */
int cpn_box_votes(const float *boxes, int64_t P, float thresh, float *votes, void *stream);

/* remove_border_contours (celldetection/ops/cpn.py:258-290): keep[i] = 1 iff all points of contour i (+offset)
 * satisfy y>pad (top), x<w-pad (right), y<h-pad (bottom), x>pad (left) on the enabled sides.
 * sides: bit0 top, bit1 right, bit2 bottom, bit3 left. */
int cpn_border_keep(const float *contours, int64_t P, int32_t samples, float off_x, float off_y, float h, float w,
                    float pad, int32_t sides, uint8_t *keep, void *stream);

/* F.interpolate(x, size, mode='bilinear', align_corners=False) on fp32 NCHW planes: `_equal_size` of the score-bound
 * masks and head maps (celldetection/models/cpn.py:109-123,279).  src [planes, Hin, Win] -> dst [planes, Hout, Wout]. */
int cpn_resize_bilinear_f32(const float *src, float *dst, int64_t planes, int32_t Hin, int32_t Win, int32_t Hout,
                            int32_t Wout, void *stream);
/* The same with the mode of CPNCore's `refinement_interpolation` (celldetection/models/cpn.py:109-115,277-279): mode 0 =
 * bilinear (= cpn_resize_bilinear_f32), 1 = bicubic (PyTorch's cubic convolution, A = -0.75, align_corners=False).  The
 * non-interpolating modes cannot run in the reference (torch rejects align_corners for them) and are no modes here.
 * Inside a plan the resize op CPN_OP_BILINEAR selects bicubic with act = 1 (bf16 / fp32 plans). */
int cpn_resize_f32(const float *src, float *dst, int64_t planes, int32_t Hin, int32_t Win, int32_t Hout, int32_t Wout,
                   int32_t mode, void *stream);

/* Measurement aid (no reference counterpart): shader clock of the bf16 implicit-GEMM conv kernels, measured inside the kernels.
 * Only libcpn_hip_clock.so (csrc/conv_igemm.hip compiled with -DCPN_EXP_CLOCK=2; built next to libcpn_hip.so, selected with
 * CPN_HIP_LIB) carries the probe: one workgroup of every conv launch adds the s_memtime ticks (shader clock) and the 100-MHz
 * s_memrealtime ticks its main loop took, its K steps, 1, and 4 x the matrix-pipe cycles its MFMAs occupy per SIMD (all waves of the
 * CU counted) to out15[5 * bin + 0..4], bin 0 = 7x7 convs, 1 = 3x3, 2 = other tap counts.  out15 (host, may be NULL) receives the
 * sums, reset != 0 clears them afterwards.  The product library returns 1 ("built without the clock probe").  bench.py reports
 * shader MHz = 100 * ticks / ref ticks and matrix-pipe duty = pipe cycles / ticks next to the roofline. */
int cpn_debug_clock_probe(unsigned long long *out15, int reset);

/* The same rule for ALL detections of a forwarded batch of tiles in one launch (the per-tile loop of
 * celldetection_scripts/cpn_inference.py:370-380): contour p belongs to image image_index[p] (int32, device);
 * sides[n] (int32, device) is the bit mask above of tile n, offsets[n] = (x, y) floats ADDED to the coordinates
 * (the reference passes -tile_offset); h, w, pad as above. */
int cpn_border_keep_batched(const float *contours, int64_t P, int32_t samples, const int32_t *image_index,
                            const int32_t *sides, const float *offsets, int32_t n_images, float h, float w, float pad,
                            uint8_t *keep, void *stream);

/* Slide-scale NMS: same semantics and the SAME keep list as cpn_nms with one segment (torchvision nms: greedy, stable
 * descending-score order), computed with a spatial grid + sparse suppressor lists + a monotone fixed point instead of
 * the dense P x P/64 bit mask -- O(P + E) memory, E = number of (box, higher-ranked box with IoU > thresh) pairs.
 * Replaces the global NMS over all detections of a slide (celldetection_scripts/cpn_inference.py:405-408,426).
 * Needs thresh >= 0.  max_edges: capacity of the caller's edge buffer; when E exceeds it the call returns
 * CPN_E_WORKSPACE with *edges_needed = E (retry with a larger workspace).  keep: int64 [P] device (first
 * *keep_count_host entries valid); keep_count_dev (optional) receives the count on the device as well.
 * sweeps (optional, host): number of fixed-point sweeps executed.  This call SYNCHRONISES the stream (it reads E, the
 * convergence counter and the keep count back). */
int64_t cpn_nms_binned_workspace_bytes(int64_t P, int64_t max_edges);
int cpn_nms_binned(const float *boxes, const float *scores, int64_t P, float thresh, int64_t max_edges, int64_t *keep,
                   int64_t *keep_count_dev, int64_t *keep_count_host, int64_t *edges_needed, int32_t *sweeps,
                   void *workspace, int64_t workspace_bytes, void *stream);

/* Slide preprocessing before tiling: `preprocess` -> cd.data.normalize_percentile
 * (celldetection_scripts/cpn_inference.py:196-222, celldetection/data/misc.py:156-161).
 * cpn_histogram: value histogram of an 8-bit (dtype 1, 256 bins) or 16-bit (dtype 2, 65536 bins) image into the
 * pre-zeroed uint32 array `hist` (the caller derives np.percentile's interpolated order statistics from it).
 * cpn_rescale_to_uint8: out = uint8(rint(((clip(x, low, high) - low) / (high - low)) * 255)) in float64
 * (normalize_percentile + skimage.img_as_ubyte); x: dtype 0 = f32, 1 = u8, 2 = u16. */
int cpn_histogram(const void *x, int32_t dtype, int64_t n, uint32_t *hist, void *stream);
int cpn_rescale_to_uint8(const void *x, int32_t dtype, int64_t n, double low, double high, uint8_t *out, void *stream);

/* Tile pre-filter of the slide loop.  Replaces the per-tile `mask[slices].any()` of TileLoader
 * (celldetection_scripts/cpn_inference.py:88-100: tiles whose mask / point-mask crop is empty are skipped): for every
 * window i = (y0, y1, x0, x1) of `windows` (int32 [n][4], device; bounds inside the H x W map -- the caller's tiling
 * table) out[i] |= any(mask[y0:y1, x0:x1] != 0).  mask: [H][W] row-major, dtype 0 = f32, 1 = u8 / bool; out: int32 [n]
 * device, pre-zeroed by the caller.  ONE launch and no host synchronisation for the whole table. */
int cpn_window_any(const void *mask, int32_t dtype, int32_t H, int32_t W, const int32_t *windows, int32_t n, int32_t *out,
                   void *stream);

/* ------------------------------------------------------------------------------------------------------------
 * Label rasterisation.  Replaces celldetection.data.contours2labels / render_contour
 * (celldetection/data/cpn.py:245-255,292-358; called from celldetection_scripts/cpn_inference.py:811):
 * contour k is rounded (half to even), clipped to the image, filled as a polygon (OpenCV
 * drawContours(thickness=-1) rule for integer vertices, restated) and added with value k + 1 to the FIRST channel whose
 * region [bbox expanded by `gap`] holds no label yet.  The sequential loop of the reference is resolved in rounds over
 * mutually independent contours (see csrc/labels.hip); the caller drives the rounds:
 *   cpn_labels_prepare:     contours fp32 [K,S,2] -> integer points [K,S,2], boxes [K,4] = (xmin, ymin, xmax, ymax)
 *   cpn_labels_bin:         cell id of every box centre for a grid_w x grid_h grid of `cell`-pixel cells (cell >= largest
 *                           box extent + gap + 1) and the identity index; the caller sorts (stably) by cell id
 *   cpn_labels_cell_bounds: [begin, end) of every non-empty cell in the sorted order (arrays pre-zeroed)
 *   cpn_labels_round:       ONE round: marks the contours whose predecessors are all painted, chooses their channel on
 *                           the planar int32 canvas [channels][H][W] and paints them.  counters_host[0] = painted in
 *                           this round, [1] = contours that found all `channels` occupied (grow the canvas, call
 *                           again), [2] = ready contours.  Synchronises the stream.  use_ioa != 0 (`ioa_thresh`,
 *                           data/cpn.py:341-350): a ready contour whose filled area is covered by labels of any channel
 *                           to more than ioa_thresh (int / int in float64, like numpy) is resolved WITHOUT painting:
 *                           state = 2, counted in counters_host[0].  Painted contours carry the provisional value k + 1;
 *                           the caller renumbers to the reference's running label (k + 1 - skipped contours before k).
 * ---------------------------------------------------------------------------------------------------------- */
int cpn_labels_prepare(const float *contours, int64_t K, int32_t S, int32_t H, int32_t W, int32_t rounded, int32_t clip,
                       int32_t *points, int32_t *boxes, void *stream);
int cpn_labels_bin(const int32_t *boxes, int64_t K, int32_t grid_w, int32_t grid_h, int32_t cell, uint32_t *cell_id,
                   uint32_t *index, void *stream);
int cpn_labels_cell_bounds(const uint32_t *sorted_cell_id, int64_t K, uint32_t *cell_begin, uint32_t *cell_end,
                           void *stream);
int cpn_labels_round(const int32_t *points, const int32_t *boxes, int64_t K, int32_t S, int32_t H, int32_t W,
                     int32_t gap, int32_t grid_w, int32_t grid_h, int32_t cell, const uint32_t *sorted_index,
                     const uint32_t *cell_begin, const uint32_t *cell_end, int32_t *canvas, int32_t channels,
                     uint8_t *state, uint8_t *ready, uint32_t *ready_list, int32_t *channel, int32_t *counters,
                     int32_t *counters_host, int32_t use_ioa, double ioa_thresh, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* CPN_HIP_H */
