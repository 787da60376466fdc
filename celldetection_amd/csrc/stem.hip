// ResNet stem for gfx950 (MI355X / CDNA4): Conv2d(in_channels <= 4 -> 32 | 64, 7x7, stride 2, pad 3) + folded BN + ReLU.
//
// Reference: celldetection/models/resnet.py:274-284 (`body.0` = conv7x7 s2 + BN + ReLU of every ResNet / ResNeXt encoder).
// In the generic implicit-GEMM kernel (conv_igemm.hip) the 3 input channels are padded to a 32-channel record, i.e. the
// MFMA loop runs 49 taps x 32 channels = 1568 K-values for 147 real ones, and the converted input tensor is 32 channels
// wide (268 MB for 16 x 512^2 instead of 25 MB).  This kernel uses the fact that with a FOUR-channel NHWC input the 7 taps
// of one filter row are 28 CONTIGUOUS bf16 values in memory:
//   * cpn::launch_input_stem writes the input as bf16 [N][H + 6][W + 8][4] with a zero border (3 rows / columns in front,
//     >= 3 behind), so that every filter window lies inside the buffer and needs no bounds logic;
//   * K = 7 filter rows x 32 values (8 pixels x 4 channels: 7 taps + one pixel that meets zero weights) = 224: 7x fewer
//     MFMAs, and a lane's MFMA B-operand (8 consecutive K-values of one output pixel) is ONE 16-byte global load -- the
//     windows of neighbouring output pixels overlap, L1/L2 absorb it; there is no LDS staging of the input at all;
//   * weights: [7][cout][32] bf16 (BN folded), each wave keeps its A-fragments in registers (28 KB for 64 outputs) and
//     walks over 32-pixel row fragments (persistent waves, grid-stride);
//   * epilogue: bias + ReLU in the MFMA layout, v_permlane32_swap pairs the two lane halves so that every lane owns 8
//     consecutive channels of one pixel and stores 16 B (cdna guide T21).
#include <hip/hip_runtime.h>

#include "cpn_kernels.h"

namespace cpn {
namespace stem {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
typedef __attribute__((ext_vector_type(2))) float f32x2_t;

__device__ __forceinline__ unsigned int pack_bf16x2(float lo, float hi) {
    const f32x2_t v = {lo, hi};
    return __builtin_bit_cast(unsigned int, __builtin_convertvector(v, bf16x2_t));
}

// ---- input conversion: f32 / u8 NCHW -> bf16 [N][H + 6][W + 8][4], zero border, channels >= C zero ------------------
__global__ __launch_bounds__(256) void input_stem_kernel(const InputArgs a) {
    const int Hp = a.H + STEM_PAD_ROWS, Wp = a.W + STEM_PAD_COLS;
    const long total = (long) a.N * Hp * Wp;
    const long HW = (long) a.H * a.W;
    int bad = 0;
    for (long i = blockIdx.x * (long) blockDim.x + threadIdx.x; i < total; i += (long) gridDim.x * blockDim.x) {
        const int xp = (int) (i % Wp);
        const long r = i / Wp;
        const int yp = (int) (r % Hp);
        const long n = r / Hp;
        const int y = yp - 3, x = xp - 3;
        float v[4] = {0.f, 0.f, 0.f, 0.f};
        if (y >= 0 && y < a.H && x >= 0 && x < a.W) {
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                if (c >= a.C) break;
                const long si = (n * a.C + c) * HW + (long) y * a.W + x;
                v[c] = a.dtype == 0 ? ((const float *) a.src)[si] : (float) ((const unsigned char *) a.src)[si] / 255.f;
                if (!(v[c] >= 0.f && v[c] <= 1.f)) bad = 1;
            }
        }
        u32x2 o;
        o.x = pack_bf16x2(v[0], v[1]);
        o.y = pack_bf16x2(v[2], v[3]);
        *(u32x2 *) ((unsigned char *) a.dst + i * 8) = o;
    }
    if (bad && a.range_flag) atomicOr(a.range_flag, 1);
}

// ---- the conv --------------------------------------------------------------------------------------------------------
// NF = 32-channel output fragments (1 | 2).  One wave = one 32-pixel row fragment at a time.
// F8: the output tensor of an fp8 plan -- OCP e4m3 codes of value * out_inv_scale (the stem itself computes in bf16 on the
// bf16 input in both cases: more exact than the e4m3 graph's generic stem, and it skips the 64-channel padded input tensor).
template <int NF, bool F8>
__global__ __launch_bounds__(256, 2) void stem7_kernel(const StemArgs a) {
    const int lane = threadIdx.x & 63;
    const int l31 = lane & 31, lhi = lane >> 5;
    const int wave = (int) ((blockIdx.x * (unsigned) blockDim.x + threadIdx.x) >> 6);
    const int nwaves = (int) ((gridDim.x * (unsigned) blockDim.x) >> 6);
    const int Wp = a.W + STEM_PAD_COLS, Hp = a.H + STEM_PAD_ROWS;
    const int fx = (a.Wout + 31) >> 5;                 // fragments per output row
    const long nfrag = (long) a.N * a.Hout * fx;
    const int coutp = NF * 32;

    // A-fragments: weights [7][coutp][32] bf16; lane -> output channel j * 32 + l31, K-octet (half, lhi) of filter row ky
    bf16x8 wf[NF][14];
#pragma unroll
    for (int j = 0; j < NF; ++j)
#pragma unroll
        for (int ks = 0; ks < 14; ++ks)
            wf[j][ks] = *(const bf16x8 *) ((const unsigned char *) a.weights +
                                           ((size_t) ((ks >> 1) * coutp + j * 32 + l31) * 32 + (ks & 1) * 16 + lhi * 8) * 2);
    float bias[NF][4][4];
#pragma unroll
    for (int j = 0; j < NF; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int e = 0; e < 4; ++e) bias[j][q][e] = a.bias ? a.bias[j * 32 + 8 * q + 4 * lhi + e] : 0.f;

    for (long f = wave; f < nfrag; f += nwaves) {
        const int tx = (int) (f % fx);
        const long r = f / fx;
        const int oy = (int) (r % a.Hout);
        const int n = (int) (r / a.Hout);
        const int ox = tx * 32 + l31;
        const int oxc = ox < a.Wout ? ox : a.Wout - 1;  // lanes past the row end recompute its last pixel (never stored)
        // B-fragments: this lane's 8 K-values of filter row ky are 16 contiguous bytes of the padded input
        const unsigned char *p = (const unsigned char *) a.src +
                                 (((size_t) n * Hp + 2 * oy) * Wp + 2 * oxc) * 8 + lhi * 16;
        bf16x8 bx[14];
#pragma unroll
        for (int ks = 0; ks < 14; ++ks) bx[ks] = *(const bf16x8 *) (p + (size_t) (ks >> 1) * Wp * 8 + (ks & 1) * 32);
        f32x16 acc[NF];
#pragma unroll
        for (int j = 0; j < NF; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
#pragma unroll
        for (int ks = 0; ks < 14; ++ks)
#pragma unroll
            for (int j = 0; j < NF; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[j][ks], bx[ks], acc[j], 0, 0, 0);
        // epilogue: lane (l31, lhi) holds channels j*32 + 8q + 4 lhi + e of pixel l31.  Swapping the q-odd half of the lower
        // lanes with the q-even half of the upper lanes gives every lane 8 consecutive channels: 16-byte stores
        unsigned char *drow = (unsigned char *) a.dst +
                              (((size_t) n * a.Hout + oy) * a.Wout + ox) * (size_t) a.dst_stride * (F8 ? 1 : 2);
#pragma unroll
        for (int j = 0; j < NF; ++j)
#pragma unroll
            for (int qp = 0; qp < 2; ++qp) {
                const int q0 = 2 * qp, q1 = q0 + 1;
                float v0[4], v1[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    v0[e] = fmaxf(acc[j][q0 * 4 + e] + bias[j][q0][e], 0.f);
                    v1[e] = fmaxf(acc[j][q1 * 4 + e] + bias[j][q1][e], 0.f);
                }
                const int ch = j * 32 + 8 * (lhi ? q1 : q0);
                if constexpr (F8) {  // four e4m3 codes per dword: one swap gives every lane 8 consecutive channels = 8 bytes
                    int c0 = 0, c1 = 0;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        v0[e] = __builtin_amdgcn_fmed3f(v0[e] * a.out_inv_scale, -448.f, 448.f);
                        v1[e] = __builtin_amdgcn_fmed3f(v1[e] * a.out_inv_scale, -448.f, 448.f);
                    }
                    c0 = __builtin_amdgcn_cvt_pk_fp8_f32(v0[0], v0[1], c0, false);
                    c0 = __builtin_amdgcn_cvt_pk_fp8_f32(v0[2], v0[3], c0, true);
                    c1 = __builtin_amdgcn_cvt_pk_fp8_f32(v1[0], v1[1], c1, false);
                    c1 = __builtin_amdgcn_cvt_pk_fp8_f32(v1[2], v1[3], c1, true);
                    const u32x2 sw = __builtin_amdgcn_permlane32_swap((unsigned) c0, (unsigned) c1, false, false);
                    if (ox < a.Wout) *(u32x2 *) (drow + (size_t) ch) = sw;
                    continue;
                }
                unsigned a0 = pack_bf16x2(v0[0], v0[1]), a1 = pack_bf16x2(v0[2], v0[3]);  // channels 8 q0 + 4 lhi ..+3
                unsigned b0 = pack_bf16x2(v1[0], v1[1]), b1 = pack_bf16x2(v1[2], v1[3]);  // channels 8 q1 + 4 lhi ..+3
                // v_permlane32_swap(x, y): x of lanes 32..63 <-> y of lanes 0..31
                const u32x2 s0 = __builtin_amdgcn_permlane32_swap(a0, b0, false, false);
                const u32x2 s1 = __builtin_amdgcn_permlane32_swap(a1, b1, false, false);
                // lower lanes: (a = own q0 channels 0..3, b = upper lanes' q0 channels 4..7)  -> channels 8 q0 .. +7
                // upper lanes: (a = lower lanes' q1 channels 0..3, b = own q1 channels 4..7)  -> channels 8 q1 .. +7
                u32x4 o;
                o.x = s0.x; o.y = s1.x; o.z = s0.y; o.w = s1.y;
                if (ox < a.Wout) *(u32x4 *) (drow + (size_t) ch * 2) = o;
            }
    }
}

}  // namespace stem

int launch_input_stem(const InputArgs &a, hipStream_t stream) {
    const long total = (long) a.N * (a.H + STEM_PAD_ROWS) * (a.W + STEM_PAD_COLS);
    int blocks = (int) ((total + 255) / 256);
    if (blocks > 256 * 16) blocks = 256 * 16;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(stem::input_stem_kernel, dim3(blocks), dim3(256), 0, stream, a);
    return (int) hipGetLastError();
}

int launch_stem7(const StemArgs &a, hipStream_t stream) {
    if (a.coutp != 32 && a.coutp != 64) return (int) hipErrorInvalidValue;
    const long nfrag = (long) a.N * a.Hout * ((a.Wout + 31) / 32);
    long blocks = (nfrag + 3) / 4;        // 4 waves per block, one fragment per wave and pass
    if (blocks > 256 * 2) blocks = 256 * 2;  // persistent: two blocks per CU (the A-fragments are loaded once per wave)
    if (blocks < 1) blocks = 1;
    const bool f8 = a.out_inv_scale > 0.f;  // e4m3 output (fp8 plans)
    if (a.coutp == 64) {
        if (f8) hipLaunchKernelGGL((stem::stem7_kernel<2, true>), dim3((unsigned) blocks), dim3(256), 0, stream, a);
        else hipLaunchKernelGGL((stem::stem7_kernel<2, false>), dim3((unsigned) blocks), dim3(256), 0, stream, a);
    } else {
        if (f8) hipLaunchKernelGGL((stem::stem7_kernel<1, true>), dim3((unsigned) blocks), dim3(256), 0, stream, a);
        else hipLaunchKernelGGL((stem::stem7_kernel<1, false>), dim3((unsigned) blocks), dim3(256), 0, stream, a);
    }
    return (int) hipGetLastError();
}

}  // namespace cpn
