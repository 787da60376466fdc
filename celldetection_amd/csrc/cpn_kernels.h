// Internal (non-ABI) declarations shared by the HIP translation units of libcpn_hip.so.
// Target: gfx950 (MI355X, CDNA4) only -- wave64, MFMA 32x32x16 bf16, 160 KiB LDS.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace cpn {

// ---------------------------------------------------------------------------------------------------------
// Implicit-GEMM convolution (NHWC bf16 activations, packed bf16 weights, fp32 accumulate on MFMA)
// ---------------------------------------------------------------------------------------------------------
enum OutMode : int { OUT_BF16_NHWC = 0, OUT_F32_NCHW = 1, OUT_FUSED_HEAD = 2 };
enum Act : int { ACT_NONE = 0, ACT_RELU = 1, ACT_SIGMOID = 2, ACT_TANH_SCALED = 3,
                 // hidden activations of the ReadOut heads beyond ReLU (round 5; `head_activation*` of models/cpn.py:183-233 ->
                 // lookup_nn(name): torch.nn modules with their default arguments)
                 ACT_LEAKY_RELU = 4, ACT_SILU = 5, ACT_GELU = 6, ACT_ELU = 7, ACT_TANH = 8, ACT_HARDSWISH = 9, ACT_MISH = 10,
                 ACT_SELU = 11, ACT_SOFTPLUS = 12 };
// the activations above ACT_TANH_SCALED, as torch.nn computes them in fp32 (LeakyReLU slope 0.01, GELU exact (erf), ELU alpha 1,
// Softplus beta 1 / threshold 20); the first four keep their own code at the call sites
__host__ __device__ inline float act_apply(float x, int act) {
    switch (act) {
        case ACT_RELU: return fmaxf(x, 0.f);
        case ACT_SIGMOID: return 1.f / (1.f + expf(-x));
        case ACT_LEAKY_RELU: return x > 0.f ? x : 0.01f * x;
        case ACT_SILU: return x / (1.f + expf(-x));
        case ACT_GELU: return 0.5f * x * (1.f + erff(x * 0.70710678118654752440f));
        case ACT_ELU: return x > 0.f ? x : expm1f(x);
        case ACT_TANH: return tanhf(x);
        case ACT_HARDSWISH: return x * fminf(fmaxf(x + 3.f, 0.f), 6.f) / 6.f;
        case ACT_MISH: return x * tanhf(x > 20.f ? x : log1pf(expf(x)));
        case ACT_SELU: return 1.0507009873554804934193349852946f * (x > 0.f ? x : 1.6732632423543772848170429916717f * expm1f(x));
        case ACT_SOFTPLUS: return x > 20.f ? x : log1pf(expf(x));
        default: return x;
    }
}

struct ConvArgs {
    // sources (NHWC bf16).  The conv input is the virtual channel concat [src0 | src1]; a source flagged "up"
    // is stored at its own resolution Hs x Ws and read through a nearest-neighbour resize to Hin x Win with
    // PyTorch's 'nearest' index rule: src = min(floor(dst * (float(Hs) / Hin)), Hs - 1)  (== dst >> 1 for an exact x2)
    const void *src0, *src1;
    int c0_stride, c1_stride;   // channel stride (= padded channel count) of the source buffers
    int c0_used;                // channels of the virtual concat that come from src0 (multiple of 32)
    int up0, up1;               // 0 none | 1 nearest | (up0 only) 2 bilinear, align_corners=False (KxK stride-1 convs)
    int Hs0, Ws0, Hs1, Ws1;     // stored sizes of the sources (== Hin, Win when not resized)
    float sy0, sx0, sy1, sx1;   // float(Hs) / Hin, float(Ws) / Win of the resized sources
    int N, Hin, Win;            // virtual (post-resize) input size
    int Hout, Wout;
    int KH, KW, stride, pad;
    int bundles;                // grid.z: independent channel bundles (grouped conv); 1 for dense
    int cin_b, cout_b;          // channels per bundle (cin_b multiple of 32; cout_b multiple of 32)
    const void *weights;        // [bundle][cin_b/32 x KH*KW items (+1 zero item if odd)][cout_b][32] bf16
    const float *bias;          // [bundles*cout_b] fp32 (BN folded) or nullptr
    // epilogue
    int phase;                  // sub-pixel phase conv (CPN_SUBPIXEL_PHASE): the `bundles` = 4 output phases (py, px) =
                                // (g >> 1, g & 1) read the SAME cin_b input channels with padding (pad - py, pad - px);
                                // 2 (CPN_SUBPIXEL_SCATTER): additionally phase g writes channels [0, cout_b) of pixel
                                // (2 oy + py, 2 ox + px) of a [2 Hout][2 Wout] destination and all phases share one bias
                                // 3 (CPN_SUBPIXEL_BL_PHASE): the four phases of a k x k conv over a x2 BILINEAR-upsampled map as
                                // k2 x k2 convs on the low-resolution map (k2 = (k + 3) / 2, the SAME padding for every phase,
                                // one shared bias); OUT_FUSED_HEAD only: phase g writes pixel (2 oy + py, 2 ox + px) of the
                                // [2 Hout][2 Wout] planes
    int region, region_margin;  // region 1: store only outputs with oy in [m, Hout - m) and ox in [m, Wout - m) (m = margin);
                                // region 2: store only outputs OUTSIDE that box, workgroups whose tile lies inside it exit at
                                // once (the frame the bilinear phase convs leave to the conv over the resized map)
    int narrow;                 // set by launch_conv (MODE_N): Hout / Wout hold the virtual [H/2][32] view of a 16-column output
    const void *res;            // residual NHWC bf16 (added before activation) or nullptr
    int res_stride, res_up;     // res_up 1: stored at Hr x Wr, nearest-resized to Hout x Wout (FPN top-down path)
                                // res_up 2: phase tensor [Hout/2][Wout/2][4 * res_cph], read pixel-shuffled
    int res_cph;                // res_up 2: channels per phase
    int Hr, Wr;
    float ry, rx;               // float(Hr) / Hout, float(Wr) / Wout
    int act;
    float act_scale;
    int out_mode;
    void *dst;
    int dst_stride, dst_coff;   // OUT_BF16_NHWC: channel stride / channel offset of the destination buffer
    int cout_real;              // OUT_F32_NCHW: number of real output channels (planes written)
    // OUT_FUSED_HEAD (ReadOut head: kxk conv -> BN -> ReLU -> 1x1 conv -> activation in ONE kernel): the block's
    // act(conv + bias) tile is rounded to bf16 in LDS and multiplied by the [32][cout_b] bf16 matrix fuse_w
    // (rows >= fuse_cout are zero); result + fuse_b -> fuse_act -> fp32 NCHW planes in dst (cout_real = fuse_cout)
    const void *fuse_w;
    const float *fuse_b;
    int fuse_cout, fuse_act;
    float fuse_scale;
    // fp8 (e4m3) variant only (csrc/conv_fp8.hip): activations / weights are e4m3 codes, 64 channels per record;
    // value = acc * mult[cout] + bias (+ residual code * res_scale); NHWC outputs are stored as e4m3(value * out_inv_scale)
    const float *mult;
    float res_scale, out_inv_scale;
    // fp8 plans, sub-pixel triples (round 5): the partial sums between the phase convs and the lateral conv travel as bf16
    // (one e4m3 rounding of a partial sum would outweigh the accumulated e4m3 noise of its inputs): dst_wide = the NHWC
    // destination holds bf16 VALUES (acc * mult + bias, no output scale); res_wide = the residual holds bf16 values
    int dst_wide, res_wide;
    // Bridge fusion (bf16, round 5; csrc/conv_igemm.hip MODE_BR): this 3x3 conv's input -- the output of a CPN_SUBPIXEL_SCATTER
    // conv (four 2x2 phase convs + one bias + ReLU over a x2 nearest-upsampled map, models/unet.py:92-107,213-217) -- is never
    // stored: the workgroup computes its 18 x 34 x 64-channel halo tile from the LOW-resolution map pre_src [N][pre_H][pre_W]
    // [pre_stride] (pre_cin = 32 | 64 channels) straight into the main loop's halo buffers.  Hin x Win = 2 pre_H x 2 pre_W,
    // cin_b = cout_b = 64.  pre_w: the scatter op's packed weights [phase][chunk][2 x 2 taps][64][32], pre_b: its 64 biases
    const void *pre_src, *pre_w;
    const float *pre_b;
    int pre_stride, pre_cin, pre_H, pre_W;
};

// PyTorch 'nearest' source index (upsample_nearest2d, legacy 'nearest' mode, size= given)
__host__ __device__ inline int nearest_src(int dst, float scale, int in_size) {
    const int s = (int) floorf((float) dst * scale);
    return s < in_size - 1 ? s : in_size - 1;
}

// tiles (th x tw) of an H x W output that are NOT entirely inside the box [m, H - m) x [m, W - m) (region 2 launches)
// kw > 0 (stride-1 k x k convs, k <= 9, m <= 16): the two sides of a row that crosses the box are ONE wrap tile -- output
// columns W - 16 .. W - 1 and 0 .. 15 (two halo segments of 16 + kw - 1 columns in the 48-column halo pitch) -- instead of a
// whole 32-column tile per side for a frame that is m columns wide
struct FrameTiles {
    int ty0, ty1, tx0, tx1;  // tile rows / columns [ty0, ty1) x [tx0, tx1) lie entirely inside the box
    int top, side, mid, total, wrap;
};
constexpr int WRAP_HALF = 16;
__host__ __device__ inline FrameTiles frame_tiles(int H, int W, int m, int th, int tw, int kw) {
    FrameTiles f;
    const int tiles_x = (W + tw - 1) / tw, tiles_y = (H + th - 1) / th;
    f.ty0 = (m + th - 1) / th; f.tx0 = (m + tw - 1) / tw;
    f.ty1 = (H - m) / th; f.tx1 = (W - m) / tw;
    if (f.ty1 < f.ty0) f.ty1 = f.ty0;
    if (f.tx1 < f.tx0) f.tx1 = f.tx0;
    if (f.ty0 > tiles_y) f.ty0 = f.ty1 = tiles_y;
    if (f.tx1 == f.tx0) f.ty1 = f.ty0;  // no inner column: every row is a full row
    f.top = f.ty0 * tiles_x;
    f.wrap = kw > 1 && kw <= 9 && m <= WRAP_HALF && W >= 2 * WRAP_HALF && f.tx1 > f.tx0;
    f.side = f.wrap ? 1 : f.tx0 + (tiles_x - f.tx1);
    f.mid = (f.ty1 - f.ty0) * f.side;
    f.total = f.top + f.mid + (tiles_y - f.ty1) * tiles_x;
    return f;
}

// picks a tile configuration and launches; returns hipError_t as int
int launch_conv(const ConvArgs &a, hipStream_t stream);
int launch_conv_fp8(const ConvArgs &a, hipStream_t stream);
bool conv_bridge_supported(const ConvArgs &a);  // ConvArgs.pre_*: the shape the bridge kernel (MODE_BR) runs  // e4m3 operands, v_mfma_scale_f32_32x32x64_f8f6f4
// algorithmic FLOPs actually executed by the MFMA loop of that launch (for utilisation reports)
double conv_executed_flops(const ConvArgs &a);

// Fused head of a grouped bottleneck block (csrc/conv_pair.hip): conv1 1x1 (cin -> cmid) + bias + ReLU -> conv2 3x3 stride 1
// pad 1 in bundles of cb2 channels (block-diagonal packed like any grouped CPN_OP_CONV) + bias + ReLU, NHWC bf16 in / out
struct PairArgs {
    const void *src;       // [N][H][W][c_stride] bf16
    int c_stride;
    int N, H, W;           // W = 16 | 32 | 64: full-width row strips; any W > 32: generic 16 x 32 tiles
    int cin, cmid;         // multiples of 32 / of the slab width (256 at W = 16 | 32; 128 otherwise)
    const void *w1;        // [cin/32 items (+1 zero item if odd)][cmid][32] bf16
    const float *b1;       // [cmid] or nullptr
    const void *w2;        // [cmid/cb2 bundles][(cb2/32)*9 items (+1 zero item if odd)][cb2][32] bf16
    const float *b2;       // [cmid] or nullptr
    int cb2;               // channels per conv2 bundle: 32 | 64
    void *dst;             // [N][Ho][Wo][dst_stride] bf16, Ho = (H - 1) / stride + 1 (3x3, pad 1)
    int dst_stride;
    int stride;            // conv2's stride: 1 | 2 (2: generic tiles, W >= 32)
};
bool conv_pair_supported(const PairArgs &a);
int launch_conv_pair(const PairArgs &a, hipStream_t stream);
long conv_pair_blocks(const PairArgs &a);  // workgroups of that launch
double conv_pair_executed_flops(const PairArgs &a);

struct PoolArgs {
    const void *src; void *dst;
    int N, Hin, Win, Hout, Wout, C;   // C = channel stride (multiple of 8)
    int k, stride, pad;
};
int launch_maxpool(const PoolArgs &a, hipStream_t stream);

// elementwise activation of an NHWC tensor (CPN_OP_ACT: the hidden activation of a ReadOut head other than ReLU; the kernels that
// carry the libm code are these three alone -- inlined into the conv epilogues it multiplied their compile time)
struct ActArgs {
    const void *src; void *dst;
    long count;              // elements (multiple of 8)
    int act;                 // ACT_* (act_apply)
    float in_scale, out_inv_scale;   // fp8: value = code * in_scale; code' = e4m3(act(value) * out_inv_scale)
};
int launch_act(const ActArgs &a, hipStream_t stream);       // bf16
int launch_act_f32(const ActArgs &a, hipStream_t stream);
int launch_act_fp8(const ActArgs &a, hipStream_t stream);

struct ResizeArgs {
    const void *src; void *dst;
    int N, Hin, Win, Hout, Wout, C;
    int ring;  // fp8 kernel only: > 0 -> write just the pixels within `ring` of the output's border (0: the whole map)
    int mode;  // 0 bilinear | 1 bicubic (bf16 / fp32 kernels; F.interpolate(mode=..., align_corners=False))
};
// PyTorch upsample_bicubic2d, align_corners=False (ATen UpSample.h: cubic convolution, A = -0.75): source position
// scale * (dst + 0.5) - 0.5 (NOT clamped at 0, unlike bilinear), index = min(floor(pos), size - 1), lambda = clamp(pos - index, 0, 1),
// taps index - 1 .. index + 2 clamped to the image, weights w[0..3] below; out = sum_i wy[i] * (sum_j wx[j] * v[i][j])
__host__ __device__ inline void bicubic_taps(float scale, int dst, int size, int (&idx)[4], float (&w)[4]) {
    const float pos = scale * ((float) dst + 0.5f) - 0.5f;
    int i0 = (int) floorf(pos);
    if (i0 > size - 1) i0 = size - 1;
    const float t = fminf(fmaxf(pos - (float) i0, 0.f), 1.f);
    const float A = -0.75f;
    const float x0 = t + 1.f, x3 = 2.f - t, x2 = 1.f - t;
    w[0] = ((A * x0 - 5.f * A) * x0 + 8.f * A) * x0 - 4.f * A;
    w[1] = ((A + 2.f) * t - (A + 3.f)) * t * t + 1.f;
    w[2] = ((A + 2.f) * x2 - (A + 3.f)) * x2 * x2 + 1.f;
    w[3] = ((A * x3 - 5.f * A) * x3 + 8.f * A) * x3 - 4.f * A;
    for (int j = 0; j < 4; ++j) {
        const int v = i0 - 1 + j;
        idx[j] = v < 0 ? 0 : (v > size - 1 ? size - 1 : v);
    }
}
int launch_bilinear(const ResizeArgs &a, hipStream_t stream);

struct InputArgs {
    const void *src;   // f32 NCHW (dtype 0) or u8 NCHW (dtype 1, scaled by 1/255)
    void *dst;         // bf16 NHWC, channel stride Cpad (zero padded)
    int N, C, H, W, Cpad, dtype;
    int *range_flag;   // set to 1 when a value is outside [0,1] (reference: models/commons.py:694-697)
};
int launch_input(const InputArgs &a, hipStream_t stream);

// ResNet stem fast path (csrc/stem.hip): the input as bf16 [N][H + STEM_PAD_ROWS][W + STEM_PAD_COLS][4] with a zero border
// (3 rows / columns in front), and conv 7x7 stride 2 pad 3 (+ folded BN + ReLU) straight from that layout
constexpr int STEM_PAD_ROWS = 6, STEM_PAD_COLS = 8;
struct StemArgs {
    const void *src;       // padded 4-channel input (launch_input_stem)
    void *dst;             // bf16 NHWC [N][Hout][Wout][dst_stride]
    const void *weights;   // [7][coutp][32] bf16: filter row ky, output channel, (kx 0..7, c 0..3) -- kx = 7 and c >= C zero
    const float *bias;     // [coutp] or nullptr
    int N, H, W, Hout, Wout, coutp, dst_stride;
    float out_inv_scale;   // > 0: dst holds e4m3 codes of value * out_inv_scale (fp8 plans; dst_stride in bytes = channels)
};
int launch_input_stem(const InputArgs &a, hipStream_t stream);
int launch_stem7(const StemArgs &a, hipStream_t stream);

// fp32 verification path (csrc/conv_f32.hip): same argument structs, fp32 NHWC activations, weights
// [bundle][kh*kw][cin_b][cout_b] fp32
int launch_conv_f32(const ConvArgs &a, hipStream_t stream);
int launch_input_f32(const InputArgs &a, hipStream_t stream);
int launch_maxpool_f32(const PoolArgs &a, hipStream_t stream);
int launch_bilinear_f32(const ResizeArgs &a, hipStream_t stream);

// slide preprocessing: value histogram (dtype 1 = u8: 256 bins, 2 = u16: 65536 bins) and percentile rescale to uint8
int launch_histogram(const void *x, int dtype, long n, unsigned int *hist, hipStream_t stream);
// tile pre-filter: out[i] |= any(mask[y0:y1, x0:x1] != 0) for windows[i] = (y0, y1, x0, x1); dtype 0 = f32, 1 = u8 / bool
int launch_window_any(const void *mask, int dtype, int W, const int *windows, int n, int *out, hipStream_t stream);
int launch_rescale_u8(const void *x, int dtype, long n, double low, double high, unsigned char *out, hipStream_t stream);

// fp8 (e4m3) helper kernels (csrc/misc_fp8.hip) and the calibration reduction
int launch_input_fp8(const InputArgs &a, float inv_scale, hipStream_t stream);
int launch_maxpool_fp8(const PoolArgs &a, hipStream_t stream);
int launch_bilinear_fp8(const ResizeArgs &a, hipStream_t stream);
int launch_absmax_bf16(const void *x, long count, float *out, hipStream_t stream);

}  // namespace cpn
