// Implicit-GEMM convolution for gfx950 (MI355X / CDNA4): NHWC bf16 activations, pre-packed bf16 weights,
// fp32 accumulation on v_mfma_f32_32x32x16_bf16, fused bias(BN-folded)/residual/activation epilogue.
//
// Replaces on the CPN inference path (reference = PyTorch ATen/cuDNN calls, no native code of its own):
//   nn.Conv2d + BatchNorm2d + ReLU of celldetection/models/resnet.py:28-116,274-284 (stem, 1x1, grouped 3x3, residual),
//   TwoConvNormRelu celldetection/models/commons.py:120-149 (UNet encoder/decoder), ConvNorm :68-92 (FPN),
//   ReadOut :461-511 (7x7 head conv + final 1x1 with sigmoid / ScaledTanh :175-187),
//   F.interpolate(nearest) + torch.cat of GeneralizedUNet.forward celldetection/models/unet.py:207-230
//   (virtual concat + index>>1 upsample in the halo loader) and the FPN top-down add (torchvision FPN.forward).
//
// Design (MI355X-first, not a port):
//   * one workgroup = TH x 32 output pixels x BN output channels; wave64 tiles of (WM*32 px) x (WN*32 cout);
//     D[cout][pixel] orientation so that every lane owns 4 consecutive output channels of one pixel
//     (8-byte NHWC bf16 stores, 128-byte coalesced NCHW fp32 plane stores for the head outputs);
//   * K loop = (32-channel chunk) x (filter tap).  The input halo tile of a chunk is staged ONCE into LDS and
//     re-read for every tap (49x reuse for the 7x7 heads); weights stream through a double-buffered LDS slab;
//   * all staging is LDS-DMA (global_load_lds_dwordx4: no VGPR round trip, no ds_write pass).  One pipeline step
//     covers TWO K items (two taps, or two chunks of a 1x1 conv): the DMA of step s+1 is issued right after the
//     barrier that opens step s and is waited for (vmcnt(0)) only at the top of step s+1, i.e. it has a whole
//     32-MFMA-per-wave step to land;
//   * LDS records are un-padded 64 B (32 bf16); the 16-byte part index is XOR-swizzled with (record>>2)&3 --
//     applied on the DMA *source* address and on the ds_read address (LDS-DMA destinations are lane-linear) --
//     which removes the bank conflicts among the lanes of a ds_read_b128 fragment read for stride-1 convs at any tap
//     offset (round-3 PMC of the flagship 8x256 tile: SQ_LDS_BANK_CONFLICT 2.5 M cycles per dispatch, ~2 % of its
//     LDS-array cycles -- what is left comes from the epilogue staging tile and the stride-2 / narrow variants);
//   * the kernel is issue-bound long before it is LDS- or HBM-bound (a 32x32x16 MFMA hides only ~5-7 other
//     instructions per wave), so the halo tile has a compile-time row pitch (multiple of 16 pixels): every
//     fragment address of an item is ONE VGPR (computed with ~6 VALU per 16 MFMAs) + immediate offsets, the
//     k-half is an XOR 32 on it, and the pipeline state is tracked incrementally (no divisions in the loop);
//   * grouped convs (ResNeXt cardinality 32) run as independent dense "bundles" (grid.z) of >=32 channels with
//     block-diagonal packed weights.
#include <algorithm>
#include <atomic>
#include <cstdio>
#include <cstdlib>

#include "cpn_error.h"
#include "cpn_kernels.h"
#include "lds_dma.h"

// This source is compiled twice: as is (bf16 operands, v_mfma_f32_32x32x16_bf16) and through csrc/conv_fp8.hip with
// CPN_FP8 = 1 (OCP e4m3 operands, v_mfma_scale_f32_32x32x64_f8f6f4: a 64-byte LDS record holds 64 channels instead
// of 32, ONE K=64 MFMA consumes both 16-byte parts of a lane's record half, i.e. twice the MACs per staged byte,
// LDS read and MFMA cycle).  Everything that is not marked CPN_FP8 is shared.
#ifndef CPN_FP8
#define CPN_FP8 0
#endif
#if CPN_FP8
#define CPN_NS cpn_fp8
#else
#define CPN_NS cpn
#endif

namespace CPN_NS {
using namespace cpn;


#if CPN_FP8
typedef unsigned char elem_t;    // e4m3
#else
typedef unsigned short elem_t;   // bf16
#endif
constexpr int ES = (int) sizeof(elem_t);  // bytes per channel
constexpr int CH = 64 / ES;               // channels per 64-byte LDS record (= K extent of one pipeline item)
constexpr int EPP = 16 / ES;              // channels per 16-byte part

#if CPN_FP8
typedef i32x4 frag_t;   // 16 e4m3 (one 16-byte part)
#else
typedef bf16x8 frag_t;  // 8 bf16
#endif

constexpr int REC = 64;  // LDS bytes per 32-channel record (pixel or weight row)
constexpr int TW = 32;   // output tile width in pixels (= one MFMA column fragment)


// 8 consecutive channels of one pixel <-> memory (16 B of bf16 | 8 B of e4m3 scaled by 1/out_scale)
#if CPN_FP8
typedef u32x2 store8_t;
__device__ __forceinline__ store8_t pack8(const float (&v)[8], float inv_scale) {
    float q[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) q[e] = __builtin_amdgcn_fmed3f(v[e] * inv_scale, -448.f, 448.f);  // saturate (e4m3fn)
    int lo = 0, hi = 0;
    lo = __builtin_amdgcn_cvt_pk_fp8_f32(q[0], q[1], lo, false);
    lo = __builtin_amdgcn_cvt_pk_fp8_f32(q[2], q[3], lo, true);
    hi = __builtin_amdgcn_cvt_pk_fp8_f32(q[4], q[5], hi, false);
    hi = __builtin_amdgcn_cvt_pk_fp8_f32(q[6], q[7], hi, true);
    store8_t o;
    o.x = (unsigned) lo; o.y = (unsigned) hi;
    return o;
}
__device__ __forceinline__ void add_res8(float (&v)[8], const store8_t r, float scale) {
    v[0] += __builtin_amdgcn_cvt_f32_fp8((int) r.x, 0) * scale; v[1] += __builtin_amdgcn_cvt_f32_fp8((int) r.x, 1) * scale;
    v[2] += __builtin_amdgcn_cvt_f32_fp8((int) r.x, 2) * scale; v[3] += __builtin_amdgcn_cvt_f32_fp8((int) r.x, 3) * scale;
    v[4] += __builtin_amdgcn_cvt_f32_fp8((int) r.y, 0) * scale; v[5] += __builtin_amdgcn_cvt_f32_fp8((int) r.y, 1) * scale;
    v[6] += __builtin_amdgcn_cvt_f32_fp8((int) r.y, 2) * scale; v[7] += __builtin_amdgcn_cvt_f32_fp8((int) r.y, 3) * scale;
}
// bf16 side of an fp8 plan (partial sums of a sub-pixel triple: ConvArgs.dst_wide / res_wide)
__device__ __forceinline__ u32x4 pack8_wide(const float (&v)[8]) {
    u32x4 o;
    o.x = pack_bf16x2(v[0], v[1]); o.y = pack_bf16x2(v[2], v[3]);
    o.z = pack_bf16x2(v[4], v[5]); o.w = pack_bf16x2(v[6], v[7]);
    return o;
}
__device__ __forceinline__ void add_res8_wide(float (&v)[8], const u32x4 r) {
    const unsigned r4[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        v[2 * e] += bf16_bits_to_f32(r4[e] & 0xffffu);
        v[2 * e + 1] += __uint_as_float(r4[e] & 0xffff0000u);
    }
}
#else
typedef u32x4 store8_t;
__device__ __forceinline__ store8_t pack8(const float (&v)[8], float) {
    store8_t o;
    o.x = pack_bf16x2(v[0], v[1]); o.y = pack_bf16x2(v[2], v[3]);
    o.z = pack_bf16x2(v[4], v[5]); o.w = pack_bf16x2(v[6], v[7]);
    return o;
}
__device__ __forceinline__ void add_res8(float (&v)[8], const store8_t r, float) {
    const unsigned r4[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        v[2 * e] += bf16_bits_to_f32(r4[e] & 0xffffu);
        v[2 * e + 1] += __uint_as_float(r4[e] & 0xffff0000u);
    }
}
#endif

// pointwise stride 1 / KxK stride 1 / KxK stride 2 / KxK stride 1 whose source is read through a bilinear resize
// MODE_S1R = MODE_S1 with the weight operand loaded from L2 straight into registers (dense KxK, >= 9 taps, 8x256 tile)
// MODE_PWR = MODE_PW with the register-weight loop of MODE_S1R (1x1 convs stream BOTH operands once per output tile: the
// LDS-DMA issue path is their bottleneck, and the weight half of it moves to plain buffer loads)
// MODE_N = MODE_S1 for outputs that are exactly 16 pixels wide (the deepest encoder / decoder level of a 512^2 tile): an
// MFMA pixel fragment is TWO output rows x 16 columns instead of one row x 32, so no half of the 32-pixel tile is empty.
// [N][H][16] IS [N][H/2][32] in memory, hence the launcher passes the output (and a same-size residual) with those
// virtual dimensions and the epilogue is unchanged; only the halo geometry and the fragment addresses know about it.
#ifndef CPN_RW_DEFAULT
#define CPN_RW_DEFAULT 0  // see conv_mode
#endif
#ifndef CPN_S1F_DEFAULT
#define CPN_S1F_DEFAULT 1  // see flat_ok
#endif
#ifndef CPN_WAVE_PRIO
#define CPN_WAVE_PRIO 1  // alternating wave priority in the e4m3 main loop of the 8-wave tiles (bf16: measured negative)
#endif
// MODE_S1F (round 5) = MODE_S1 for TWO co-resident 4-wave workgroups per CU (tile 8 x 32 px x 128 cout, the flagship's 128 px x
// 64 cout wave tile): a tile's head (first DMA round trip) and tail (LDS-staged epilogue + 64 KiB of stores, MFMA idle) overlap
// the OTHER workgroup's main loop instead of nothing.  What makes two of them fit the 160 KiB: a FLAT halo tile with a row pitch
// of 36 records (34 used by a 3x3 conv) instead of 48 -- a DMA instruction covers 16 consecutive records across row ends, its
// per-lane source offsets are six loop-invariant registers per wave (no column table) -- and 128-row weight slabs: 2 x 23 + 32
// = 78 KiB.  Single-source stride-1 k x k convs, 2 <= k <= 5, NHWC output.
// MODE_S1Q (round 5) = MODE_S1 on the <16,64,2,2> tile with FOUR K items per pipeline step: the 64-channel 7x7 ReadOut head runs 32
// MFMAs per SIMD between two step boundaries (barrier + DMA landing + drained fragment pipeline: ~950 cycles against 1024 of MFMA
// work), twice as many halve that share.  Four items x two buffers of 4-KiB slabs fit because the halo tiles are FLAT at pitch 40
// (38 used by a 7x7 conv): 2 x 55 + 32 = 142 KiB.  Same K order and MFMA sequence per output element as the two-item loop; items
// past the packed count read zeros through the weight descriptor's bounds.
// MODE_BR (round 5) = MODE_S1 on the <16,64,2,2> tile whose two halo tiles (cin = 64 = both chunks) are COMPUTED in the kernel:
// the bridge level of the ResNet-UNets -- conv 3x3 (64 -> 64, bias-free, BN, ReLU) over the x2-upsampled 64-channel map, run as
// four 2x2 phase convs (CPN_SUBPIXEL_SCATTER), followed by the second conv 3x3 of that TwoConvNormRelu block -- as ONE launch:
// the 537 MB full-resolution intermediate of a 16 x 512^2 batch is neither written nor read (ConvArgs.pre_*).
enum Mode : int { MODE_PW = 0, MODE_S1 = 1, MODE_S2 = 2, MODE_BL = 3, MODE_S1R = 4, MODE_PWR = 5, MODE_N = 6, MODE_S1F = 7, MODE_BR = 8, MODE_BRF = 9, MODE_S1Q = 10 };

template <int MODE>
struct ModeCfg {
    static constexpr int S = MODE == MODE_S2 ? 2 : 1;                              // conv stride
    static constexpr int PITCH = (MODE == MODE_PW || MODE == MODE_PWR || MODE == MODE_N) ? 32 : (MODE == MODE_S2 ? 80 : (MODE == MODE_S1F ? 36 : (MODE == MODE_BRF ? 34 : (MODE == MODE_S1Q ? 40 : 48))));  // halo row pitch (pixels)
};

template <int TH, int BN, int WM, int WN>
struct Cfg {
    static constexpr int WAVES_M = TH / WM;
    static constexpr int WAVES_N = BN / (32 * WN);
    static constexpr int NWAVES = WAVES_M * WAVES_N;
    static constexpr int THREADS = 64 * NWAVES;
    static constexpr int W_INSTR_ITEM = BN / 16;                   // 1-KiB DMA instructions per weight item slab
    static constexpr int W_INSTR_WAVE = 2 * W_INSTR_ITEM / NWAVES;  // per wave per step (two items)
    static_assert(TH % WM == 0 && BN % (32 * WN) == 0, "bad tile");
    static_assert((2 * W_INSTR_ITEM) % NWAVES == 0, "weight DMA must divide evenly over the waves");
};

struct HaloGeo {
    int n, iy0, ix0, sub, Hin, Win, Hs0, Ws0, Hs1, Ws1, up0, up1, c0_stride, c1_stride, HH, HWreal;
    int wseg, wx0, wx1;  // wrap tile (frame launches): halo columns [0, wseg) start at input column wx0, [wseg, 2 wseg) at wx1
    float sy0, sx0, sy1, sx1;
};
// input column of halo column hx
__device__ __forceinline__ int halo_ix(const HaloGeo &G, int hx) {
    if (G.wseg) return hx >= G.wseg ? G.wx1 + (hx - G.wseg) : G.wx0 + hx;
    return G.ix0 + hx * G.sub;
}

// source element offsets (src0 / src1 variants; -1 = zero padding) of this lane's 16 B of halo DMA instruction q:
// lane -> (pixel, 16-B slot); the slot holds channel part (slot ^ ((pixel>>2)&3)) of the pixel record
template <int PITCH>
__device__ __forceinline__ void halo_src_offsets(const HaloGeo &G, int q, int lane, int &o0, int &o1) {
    const int idx = (q << 6) + lane;
    const int pix = idx >> 2;
    const int part = (idx & 3) ^ ((pix >> 2) & 3);
    const int hy = pix / PITCH, hx = pix - hy * PITCH;
    const int iy = G.iy0 + hy * G.sub, ix = G.ix0 + hx * G.sub;  // sub > 1: strided 1x1 conv (tile = outputs)
    const bool valid = hy < G.HH && hx < G.HWreal && iy >= 0 && iy < G.Hin && ix >= 0 && ix < G.Win;
    o0 = -1;
    o1 = -1;
    if (valid) {
        const int y0 = G.up0 ? nearest_src(iy, G.sy0, G.Hs0) : iy, x0 = G.up0 ? nearest_src(ix, G.sx0, G.Ws0) : ix;
        const int y1 = G.up1 ? nearest_src(iy, G.sy1, G.Hs1) : iy, x1 = G.up1 ? nearest_src(ix, G.sx1, G.Ws1) : ix;
        o0 = ((G.n * G.Hs0 + y0) * G.Ws0 + x0) * G.c0_stride + part * EPP;
        o1 = ((G.n * G.Hs1 + y1) * G.Ws1 + x1) * G.c1_stride + part * EPP;
    }
}

// MODE_BL: the conv reads src0 through F.interpolate(mode='bilinear', align_corners=False) to Hin x Win (the FPN
// refinement head, celldetection/models/cpn.py:277-278: 256 channels at full resolution are never materialised).
// This lane's 16 B of halo instruction q = EPP channels of one halo pixel: four source pixels are loaded, blended in fp32
// with the arithmetic of bilinear_kernel (csrc/misc_kernels.hip) / bilinear_fp8_kernel, rounded to the activation type
// and written to the lane-linear LDS slot the DMA would have filled (same swizzled part order).
template <int PITCH>
__device__ __forceinline__ void halo_bilinear_store(const HaloGeo &G, int q, int lane, const unsigned char *src_chunk,
                                                    unsigned char *dst_instr) {
    typedef __attribute__((address_space(3))) unsigned char lds_u8;  // explicit LDS address space: through a generic
    typedef __attribute__((address_space(3))) u32x4 lds_u32x4;       // pointer hipcc emitted FLAT stores for the e4m3
    typedef __attribute__((address_space(3))) unsigned int lds_u32;  // instantiation (8 M VMEM writes per launch)
    lds_u8 *const dst = (lds_u8 *) dst_instr + lane * 16;
    const int idx = (q << 6) + lane;
    const int pix = idx >> 2;
    const int part = (idx & 3) ^ ((pix >> 2) & 3);
    const int hy_ = pix / PITCH, hx_ = pix - hy_ * PITCH;
    const int iy = G.iy0 + hy_, ix = halo_ix(G, hx_);
    if (!(hy_ < G.HH && hx_ < G.HWreal && iy >= 0 && iy < G.Hin && ix >= 0 && ix < G.Win)) {
        const u32x4 z = {0u, 0u, 0u, 0u};
        *(lds_u32x4 *) dst = z;
        return;
    }
    const float fy = fmaxf(G.sy0 * ((float) iy + 0.5f) - 0.5f, 0.f);
    const float fx = fmaxf(G.sx0 * ((float) ix + 0.5f) - 0.5f, 0.f);
    const int y0 = (int) fy, x0 = (int) fx;
    const int y1 = y0 + (y0 < G.Hs0 - 1 ? 1 : 0), x1 = x0 + (x0 < G.Ws0 - 1 ? 1 : 0);
    const float ly = fy - (float) y0, lx = fx - (float) x0;
    const float hy = 1.f - ly, hx = 1.f - lx;
    const unsigned char *b = src_chunk + ((size_t) G.n * G.Hs0 * G.Ws0 * G.c0_stride + (size_t) part * EPP) * ES;
    const unsigned ps = (unsigned) G.c0_stride * ES;
    const unsigned char *p00 = b + (size_t) (y0 * G.Ws0 + x0) * ps, *p01 = b + (size_t) (y0 * G.Ws0 + x1) * ps;
    const unsigned char *p10 = b + (size_t) (y1 * G.Ws0 + x0) * ps, *p11 = b + (size_t) (y1 * G.Ws0 + x1) * ps;
#if CPN_FP8
    // one dword (4 e4m3 codes) at a time: the e4m3 kernel has ~10 free VGPRs next to its 128 accumulators and four
    // fragment sets -- the 16-channel version of the bf16 code below spilled 138..257 of them
    _Pragma("nounroll") for (int w = 0; w < 4; ++w) {
        const unsigned a00 = ((const unsigned *) p00)[w], a01 = ((const unsigned *) p01)[w];
        const unsigned a10 = ((const unsigned *) p10)[w], a11 = ((const unsigned *) p11)[w];
        float r[4];
#define CPN_BL_E(E)                                                                                                \
    r[E] = __builtin_amdgcn_fmed3f(hy * (hx * __builtin_amdgcn_cvt_f32_fp8((int) a00, E) +                        \
                                         lx * __builtin_amdgcn_cvt_f32_fp8((int) a01, E)) +                       \
                                   ly * (hx * __builtin_amdgcn_cvt_f32_fp8((int) a10, E) +                        \
                                         lx * __builtin_amdgcn_cvt_f32_fp8((int) a11, E)), -448.f, 448.f)
        CPN_BL_E(0); CPN_BL_E(1); CPN_BL_E(2); CPN_BL_E(3);
#undef CPN_BL_E
        int c = 0;
        c = __builtin_amdgcn_cvt_pk_fp8_f32(r[0], r[1], c, false);
        c = __builtin_amdgcn_cvt_pk_fp8_f32(r[2], r[3], c, true);
        ((lds_u32 *) dst)[w] = (unsigned) c;  // the tensor scale is unchanged by the resize
    }
#else
    const u32x4 v00 = *(const u32x4 *) p00, v01 = *(const u32x4 *) p01, v10 = *(const u32x4 *) p10, v11 = *(const u32x4 *) p11;
    const unsigned a00[4] = {v00.x, v00.y, v00.z, v00.w}, a01[4] = {v01.x, v01.y, v01.z, v01.w};
    const unsigned a10[4] = {v10.x, v10.y, v10.z, v10.w}, a11[4] = {v11.x, v11.y, v11.z, v11.w};
    unsigned o[4];
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        const float lo = hy * (hx * bf16_bits_to_f32(a00[w] & 0xffffu) + lx * bf16_bits_to_f32(a01[w] & 0xffffu)) +
                         ly * (hx * bf16_bits_to_f32(a10[w] & 0xffffu) + lx * bf16_bits_to_f32(a11[w] & 0xffffu));
        const float hi = hy * (hx * bf16_bits_to_f32(a00[w] >> 16) + lx * bf16_bits_to_f32(a01[w] >> 16)) +
                         ly * (hx * bf16_bits_to_f32(a10[w] >> 16) + lx * bf16_bits_to_f32(a11[w] >> 16));
        o[w] = pack_bf16x2(lo, hi);
    }
    const u32x4 out = {o[0], o[1], o[2], o[3]};
    *(lds_u32x4 *) dst = out;
#endif
}

// ---- hand-counted LDS fragment reads (ds_read16: lds_dma.h) and their counted waits
template <int N, int WN, int WM>
__device__ __forceinline__ void wait_frags(frag_t (&w)[WN], frag_t (&p)[WM]) {
    static_assert((WN == 2 && (WM == 4 || WM == 2 || WM == 1)) || (WN == 1 && (WM == 2 || WM == 1)), "frag shape");
    if constexpr (WN == 2 && WM == 4)
        asm volatile("s_waitcnt lgkmcnt(%6)" : "+v"(w[0]), "+v"(w[1]), "+v"(p[0]), "+v"(p[1]), "+v"(p[2]), "+v"(p[3]) : "n"(N));
    else if constexpr (WN == 2 && WM == 2)
        asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(w[0]), "+v"(w[1]), "+v"(p[0]), "+v"(p[1]) : "n"(N));
    else if constexpr (WN == 2 && WM == 1)
        asm volatile("s_waitcnt lgkmcnt(%3)" : "+v"(w[0]), "+v"(w[1]), "+v"(p[0]) : "n"(N));
    else if constexpr (WN == 1 && WM == 2)
        asm volatile("s_waitcnt lgkmcnt(%3)" : "+v"(w[0]), "+v"(p[0]), "+v"(p[1]) : "n"(N));
    else
        asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(w[0]), "+v"(p[0]) : "n"(N));
}
#ifdef CPN_EXP_NOVMWAIT  // tuning ablation (races by construction): the step boundary does not wait for the DMA
#define CPN_WAIT_ALL_ASM "s_waitcnt lgkmcnt(0)"
#else
#define CPN_WAIT_ALL_ASM "s_waitcnt vmcnt(0) lgkmcnt(0)"
#endif
// step boundary: all my DMA landed + all my LDS reads returned (fragment set named "+v" as above)
template <int WN, int WM>
__device__ __forceinline__ void wait_all(frag_t (&w)[WN], frag_t (&p)[WM]) {
    if constexpr (WN == 2 && WM == 4)
        asm volatile(CPN_WAIT_ALL_ASM : "+v"(w[0]), "+v"(w[1]), "+v"(p[0]), "+v"(p[1]), "+v"(p[2]), "+v"(p[3]) :: "memory");
    else if constexpr (WN == 2 && WM == 2)
        asm volatile(CPN_WAIT_ALL_ASM : "+v"(w[0]), "+v"(w[1]), "+v"(p[0]), "+v"(p[1]) :: "memory");
    else if constexpr (WN == 2 && WM == 1)
        asm volatile(CPN_WAIT_ALL_ASM : "+v"(w[0]), "+v"(w[1]), "+v"(p[0]) :: "memory");
    else if constexpr (WN == 1 && WM == 2)
        asm volatile(CPN_WAIT_ALL_ASM : "+v"(w[0]), "+v"(p[0]), "+v"(p[1]) :: "memory");
    else
        asm volatile(CPN_WAIT_ALL_ASM : "+v"(w[0]), "+v"(p[0]) :: "memory");
}
template <int WN, int WM, int FRAG_STRIDE>
__device__ __forceinline__ void load_frags(frag_t (&w)[WN], frag_t (&p)[WM], unsigned paddr, unsigned waddr) {
    ds_read16<0>(w[0], waddr);
    if constexpr (WN > 1) ds_read16<32 * REC>(w[1], waddr);
    ds_read16<0>(p[0], paddr);
    if constexpr (WM > 1) ds_read16<FRAG_STRIDE>(p[1], paddr);
    if constexpr (WM > 2) {
        ds_read16<2 * FRAG_STRIDE>(p[2], paddr);
        ds_read16<3 * FRAG_STRIDE>(p[3], paddr);
    }
}

// pixel fragments only (MODE_S1R / MODE_PWR: the weight fragments come from global memory)
template <int WM, int FRAG_STRIDE>
__device__ __forceinline__ void load_pfrags(frag_t (&p)[WM], unsigned paddr) {
    static_assert(WM == 4 || WM == 2, "register-weight loop: 128- or 64-pixel wave tile");
    ds_read16<0>(p[0], paddr);
    ds_read16<FRAG_STRIDE>(p[1], paddr);
    if constexpr (WM > 2) {
        ds_read16<2 * FRAG_STRIDE>(p[2], paddr);
        ds_read16<3 * FRAG_STRIDE>(p[3], paddr);
    }
}
template <int N, int WM>
__device__ __forceinline__ void wait_pfrags(frag_t (&p)[WM]) {
    if constexpr (WM == 4) asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(p[0]), "+v"(p[1]), "+v"(p[2]), "+v"(p[3]) : "n"(N));
    else asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(p[0]), "+v"(p[1]) : "n"(N));
}

// (FrameTiles / frame_tiles: cpn_kernels.h -- the executor's heuristic enumerates the same tiles)
// wrap tiles apply to the stride-1 k x k modes (MODE_S1 / S1R / BL): the kernel and the launcher must agree
__host__ __device__ inline int frame_kw(const ConvArgs &a) { return (a.stride == 1 && !a.narrow && a.KW > 1) ? a.KW : 0; }

struct ItemState {  // one K item = (32-channel chunk c, filter tap (ky, kx)); wave-uniform scalars
    int c, ky, kx;
};
// the item that follows I in the flattened (chunk-major, tap-minor) K order; no divisions
__device__ __forceinline__ ItemState next_item(const ItemState I, int KH, int KW) {
    ItemState N = I;
    N.kx = I.kx + 1;
    if (N.kx == KW) {
        N.kx = 0;
        N.ky = I.ky + 1;
        if (N.ky == KH) {
            N.ky = 0;
            N.c = I.c + 1;
        }
    }
    return N;
}

#if defined(CPN_EXP_CLOCK) && CPN_EXP_CLOCK == 2 && !CPN_FP8
// [bin][core ticks, 100-MHz ticks, steps, launches, 4 x matrix-pipe cycles per SIMD] of the probed workgroup of every launch;
// bins: 7x7 | 3x3 | other tap counts
__device__ unsigned long long g_clock_probe[15];
#endif

template <int TH, int BN, int WM, int WN, int MODE>
__global__ __launch_bounds__((64 * (TH / WM) * (BN / (32 * WN))), ((MODE == MODE_S1F || MODE == MODE_BRF) ? 2 : 1)) void conv_igemm_kernel(const ConvArgs a) {
#if defined(CPN_EXP_CLOCK) && CPN_EXP_CLOCK == 1
    const unsigned long long clk_entry = __builtin_readcyclecounter();
#endif
    using C = Cfg<TH, BN, WM, WN>;
    constexpr int S = ModeCfg<MODE>::S;
    constexpr int PITCH = ModeCfg<MODE>::PITCH;
    constexpr bool PW = MODE == MODE_PW || MODE == MODE_PWR;
    constexpr bool BL = MODE == MODE_BL;
    constexpr bool RW = MODE == MODE_S1R || MODE == MODE_PWR;  // weights: global -> registers (no weight tiles in LDS)
    constexpr bool FL = MODE == MODE_S1F || MODE == MODE_BRF || MODE == MODE_S1Q;  // flat halo tile (pitch 36 | 34 | 40)
    constexpr int IPS = MODE == MODE_S1Q ? 4 : 2;  // K items per pipeline step
    constexpr bool BR = MODE == MODE_BR || MODE == MODE_BRF;   // halo tiles computed in the kernel by the scattered phase conv in front (bridge fusion)
    constexpr bool NR = MODE == MODE_N;  // narrow output: fragment = 2 rows x 16 px; a.Hout / a.Wout are the virtual [H/2][32]
    constexpr int RPF = NR ? 2 : 1;      // output rows per pixel fragment
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wave_m = wave / C::WAVES_N;
    const int wave_n = wave % C::WAVES_N;
    constexpr bool prio_wave = C::NWAVES == 8;  // (two waves per SIMD: alternating priority in the e4m3 main loop)

    // ---- block coordinates
    const int tiles_x = (a.Wout + TW - 1) / TW;
    const int tiles_y = (a.Hout + TH - 1) / TH;
    int bid = blockIdx.x;
    int tx, ty, n;
    int yblk = blockIdx.y;
    bool wrap = false;  // (block-uniform) this block is a wrap tile of a frame launch
    if (a.region == 2) {
        // frame-only launch: the grid holds ONLY the tiles that reach outside the box [m, H - m) x [m, W - m), enumerated rows
        // above the box | the two sides of the rows that cross it | rows below.  (A dense grid whose inner workgroups exit at
        // once put every left- and right-edge tile -- block ids = 0 / tiles_x - 1 mod tiles_x -- on TWO of the eight XCDs: 8.6
        // ms instead of 1.6 for the frame of 8 x 512^2.)
        const FrameTiles F = frame_tiles(a.Hout, a.Wout, a.region_margin, TH, TW, frame_kw(a));
        n = bid / F.total;
        int r = bid - n * F.total;
        if (r < F.top) {
            ty = r / tiles_x; tx = r - ty * tiles_x;
        } else if (r < F.top + F.mid) {
            r -= F.top;
            const int q = r / F.side, c = r - q * F.side;
            ty = F.ty0 + q; tx = c < F.tx0 ? c : F.tx1 + (c - F.tx0);
            wrap = F.wrap != 0;
        } else {
            r -= F.top + F.mid;
            const int q = r / tiles_x;
            ty = F.ty1 + q; tx = r - q * tiles_x;
        }
    } else {
        if constexpr (FL) {
            // the cout blocks of a pixel tile are folded into blockIdx.x so that they are CONSECUTIVE workgroups of one XCD
            // (block id mod 8 = XCD): bid = (group of 8 tiles, cout block, tile within the group) -- they read the same halo
            // tiles through one L2 (ideally as the two residents of one CU) instead of a whole grid sweep apart
            const int nh = (a.cout_b + BN - 1) / BN;
#ifdef CPN_EXP_S1F_MAP  // tuning ablation: 1 = cout blocks on ADJACENT block ids (different XCDs), 2 = cout-block-major (a grid sweep apart)
            const int nt8 = ((tiles_x * tiles_y * a.N + 7) / 8) * 8;
            if (CPN_EXP_S1F_MAP == 1) { yblk = bid % nh; bid = bid / nh; }
            else { yblk = bid / nt8; bid = bid % nt8; }
#else
            const int j = bid & 7, r = bid >> 3;
            yblk = r % nh;
            bid = (r / nh) * 8 + j;
#endif
            if (bid >= tiles_x * tiles_y * a.N) return;  // (grid rounded up to whole groups of 8 tiles)
        }
        tx = bid % tiles_x;
        bid /= tiles_x;
        ty = bid % tiles_y;
        n = bid / tiles_y;
    }
    const int oy0 = ty * TH, ox0 = tx * TW;
    const int n0 = yblk * BN;  // first output channel (within the bundle) of this block
    const int g = blockIdx.z;        // bundle

    const int KW = PW ? 1 : a.KW;
    const int KH = PW ? 1 : a.KH;
    const int HH = (RPF * TH - 1) * S + KH;           // halo rows
    const int hinstr = (HH * PITCH * 4 + 63) >> 6;    // 1-KiB DMA instructions per halo tile
    const int halo_buf = hinstr << 10;
    const int nchunks = a.cin_b / CH;
    const int ntaps = KH * KW;
    const bool pw = ntaps == 1;                        // one item per chunk (1x1, any stride): PW-style halo schedule
    const int nhb = pw ? 4 : (nchunks > 1 ? 2 : 1);   // halo ring size (chunk c lives in buffer c & (nhb-1))
    const int nhb_mask = nhb - 1;
    const int nreal = nchunks * ntaps;                // flattened K items; a pipeline step covers two of them
    const int nitems_packed = nreal + (nreal & 1);    // + one all-zero weight slab: the packer pads an odd item count
    const int nitems = (nreal + IPS - 1) / IPS * IPS; // every step holds IPS items: the K loop has no conditional tail (IPS = 4:
    const int nsteps = nitems / IPS;                  // items past the packed count read zeros through the descriptor's bounds)
    const int cout_b = a.cout_b;
    const int cin0 = a.phase ? 0 : g * a.cin_b;  // (sub-pixel phase conv: the four phases read the same channels)
    const bool shift_pad = a.phase == 1 || a.phase == 2;  // (phase 3, bilinear phases: one symmetric support for all four)
    const int pad_y = a.pad - (shift_pad ? (g >> 1) : 0), pad_x = a.pad - (shift_pad ? (g & 1) : 0);
    const int c0_used = a.c0_used;
    constexpr int WITEM = BN * REC;   // one item's weight slab tile
    constexpr int WBUF = IPS * WITEM; // one step's weights
    const int ldsW_off = nhb * halo_buf;
    constexpr int IPR = FL ? 1 : PITCH / 16;  // halo DMA instructions per halo row (16 pixel records of 64 B each)
    static_assert(FL || PITCH % 16 == 0, "a halo row must be a whole number of DMA instructions");
    typedef __attribute__((address_space(3))) unsigned lds_u32_t;
    lds_u32_t *const coltab = (lds_u32_t *) (smem + ldsW_off + 2 * WBUF);  // column offset table (see below)

    HaloGeo G;
    G.sub = PW ? a.stride : 1;
    G.n = n; G.iy0 = RPF * oy0 * (PW ? a.stride : S) - pad_y; G.ix0 = ox0 * (PW ? a.stride : S) - pad_x;
#ifdef CPN_EXP_HALO_SAMETILE  // (tuning ablation, wrong results: every workgroup stages the input tile of workgroup 0 -> the halo DMA hits the L2)
    G.n = 0; G.iy0 = -pad_y; G.ix0 = -pad_x;
#endif
    G.Hin = a.Hin; G.Win = a.Win;
    G.up0 = a.up0; G.up1 = a.up1;
    G.Hs0 = a.Hs0; G.Ws0 = a.Ws0; G.Hs1 = a.Hs1; G.Ws1 = a.Ws1;
    G.sy0 = a.sy0; G.sx0 = a.sx0; G.sy1 = a.sy1; G.sx1 = a.sx1;
    G.c0_stride = a.c0_stride; G.c1_stride = a.c1_stride; G.HH = HH; G.HWreal = ((NR ? 16 : TW) - 1) * S + KW;
    G.wseg = 0; G.wx0 = G.wx1 = 0;
    if (wrap) {  // fragment lanes 0..15 = output columns Wout - 16 .., lanes 16..31 = output columns 0 ..
        G.wseg = WRAP_HALF + KW - 1;
        G.wx0 = a.Wout - WRAP_HALF - pad_x;
        G.wx1 = -pad_x;
        G.HWreal = 2 * G.wseg;
    }

    // halo DMA instruction q (0..hinstr) is issued by wave q % NWAVES; the source offsets are recomputed per issue
    // (a few VALU per 1-KiB DMA; keeping them in registers cost 10 VGPRs of a kernel that sits at the 256 limit)
    const rsrc_t rs0 = make_rsrc(a.src0, (unsigned) ((size_t) a.N * a.Hs0 * a.Ws0 * a.c0_stride * ES));
    const rsrc_t rs1 = make_rsrc(a.src1, a.src1 ? (unsigned) ((size_t) a.N * a.Hs1 * a.Ws1 * a.c1_stride * ES) : 0u);

    // weight DMA: instruction q = wave + it*NWAVES of a step covers item k = q / W_INSTR_ITEM, rows qi*16..+15 of
    // the BN tile; lane -> (row, swizzled 16-B part).  Everything that does not change from step to step is
    // computed once (the loop is instruction-issue bound): item selector, LDS destination, per-lane source offset
    // (rows past cout_b read row 0 -- their outputs are never stored).
    const unsigned w_dma_lane = (unsigned) ((lane >> 2) * REC + (((lane & 3) ^ ((lane >> 4) & 3)) << 4));
    constexpr int W_IW = IPS * C::W_INSTR_ITEM / C::NWAVES;  // weight DMA instructions per wave per step
    static_assert((IPS * C::W_INSTR_ITEM) % C::NWAVES == 0, "weight DMA must divide evenly over the waves");
    int w_k[W_IW];
    int w_m0[W_IW];
    unsigned w_voff[W_IW];
#pragma unroll
    for (int it = 0; it < W_IW; ++it) {
        const int q = wave + it * C::NWAVES;
        const int k = q / C::W_INSTR_ITEM, qi = q % C::W_INSTR_ITEM;
        w_k[it] = k;
        w_m0[it] = ldsW_off + k * WITEM + (qi << 10);
        w_voff[it] = (n0 + qi * 16 + (lane >> 2) < cout_b) ? (unsigned) (qi << 10) + w_dma_lane
                                                           : (w_dma_lane & 63u);
    }
    const unsigned item_bytes = (unsigned) cout_b * REC;
    const rsrc_t rsw = make_rsrc(a.weights, (unsigned) ((size_t) a.bundles * nitems_packed * cout_b * REC));
    // byte offset of the slab of the first item of the NEXT step to be staged (steps are staged in order, two items each)
    unsigned wsoff = (unsigned) (((size_t) g * nitems_packed * cout_b + n0) * REC);

    // 1x1 fast path of the activation-tile DMA: single full-resolution source -> per-lane offsets are loop constants
    constexpr int A_INSTR_WAVE = PW ? (TH * 2 + C::NWAVES - 1) / C::NWAVES : 1;
    const bool pw_fast = PW && a.src1 == nullptr && !a.up0;
    unsigned a_voff[A_INSTR_WAVE];
    bool a_ok[A_INSTR_WAVE];
#pragma unroll
    for (int it = 0; it < A_INSTR_WAVE; ++it) {
        int o0 = -1, o1 = -1;
        if (PW) halo_src_offsets<PITCH>(G, wave + it * C::NWAVES, lane, o0, o1);
        a_ok[it] = (wave + it * C::NWAVES) < hinstr;  // (wave-uniform) the instruction exists
        a_voff[it] = o0 < 0 ? OOB_LANE : (unsigned) o0 * (unsigned) ES;  // out-of-image lanes read zeros
    }

    // MODE_S1F: DMA instruction q covers records 16 q .. 16 q + 15 of the FLAT halo tile (record r = halo row r / 36, column
    // r % 36): lane -> (record, 16-byte slot), the slot holds channel part slot ^ ((column >> 2) & 3) -- the same column-only
    // swizzle the fragment reads undo.  Per-lane byte offsets within the image are loop constants of the tile (one register per
    // instruction of this wave); image + channel chunk travel in the scalar offset.  Single source, no resize.
    constexpr int FQ = FL ? (((TH - 1 + (MODE == MODE_S1Q ? 7 : 5)) * PITCH * 4 + 63) / 64 + C::NWAVES - 1) / C::NWAVES : 1;
    unsigned f_voff[FQ];
    unsigned f_soff = 0;
    if constexpr (FL && !BR) {
        f_soff = (unsigned) (((size_t) G.n * a.Hs0 * a.Ws0 * a.c0_stride + cin0) * ES);
#pragma unroll
        for (int it = 0; it < FQ; ++it) {
            const int idx = ((wave + it * C::NWAVES) << 6) + lane;
            const int rec = idx >> 2;
            const int hy = rec / PITCH, hx = rec - hy * PITCH;
            const int iy = G.iy0 + hy, ix = G.ix0 + hx;
            const bool valid = hy < HH && hx < G.HWreal && iy >= 0 && iy < G.Hin && ix >= 0 && ix < G.Win;
            f_voff[it] = valid ? (unsigned) (((iy * a.Ws0 + ix) * a.c0_stride + ((idx & 3) ^ ((hx >> 2) & 3)) * EPP) * ES) : OOB_LANE;
        }
    }

#define HALO_DMA(CHUNK) HALO_DMA_RANGE(CHUNK, 0, hinstr)
    // cache policy of the activation / weight DMA (kernel A/B: -DCPN_HALO_AUX=2 = nt ...) and the traffic-free ablation of the halo DMA
#ifndef CPN_HALO_AUX
#define CPN_HALO_AUX 0
#endif
#ifndef CPN_W_AUX
#define CPN_W_AUX 0
#endif
#ifndef CPN_EXP_HALO_OOB
#define CPN_EXP_HALO_OOB 0
#endif
#ifndef CPN_EXP_HALO_NOISSUE  // (tuning ablation, wrong results: the halo path computes its addresses but issues no DMA instruction)
#define CPN_EXP_HALO_NOISSUE 0
#endif

    // instructions [Q0, Q1) of the halo tile of chunk CHUNK
#define HALO_DMA_RANGE(CHUNK, Q0, Q1)                                                                          \
    if constexpr (BR) {  /* both chunks' halo tiles are computed before the main loop (bridge stage below) */  \
    } else if constexpr (FL) {                                                                                        \
        const int c_ = (CHUNK);                                                                                \
        const unsigned s_ = f_soff + (unsigned) (c_ * CH * ES);                                                \
        unsigned char *dstb_ = smem + (c_ & nhb_mask) * halo_buf;                                              \
        _Pragma("unroll") for (int it_ = 0; it_ < FQ; ++it_) {                                                 \
            const int q_ = wave + it_ * C::NWAVES;                                                             \
            if (q_ < hinstr) bdma16<CPN_HALO_AUX>(rs0, f_voff[it_], s_, dstb_ + (q_ << 10));                   \
        }                                                                                                      \
    } else HALO_DMA_RANGE_ROWS(CHUNK, Q0, Q1)

#define HALO_DMA_RANGE_ROWS(CHUNK, Q0, Q1)                                                                          \
    {                                                                                                          \
        const int c_ = (CHUNK);                                                                                \
        const int cin_ = cin0 + c_ * CH;                                                                       \
        const bool from0_ = cin_ < c0_used;                                                                    \
        const unsigned soff_ = (unsigned) (from0_ ? cin_ : cin_ - c0_used) * (unsigned) ES;                    \
        unsigned char *dstb_ = smem + (c_ & nhb_mask) * halo_buf;                                              \
        const int q1_ = (Q1);                                                                                  \
        const int up_ = from0_ ? G.up0 : G.up1, Hs_ = from0_ ? G.Hs0 : G.Hs1;                                  \
        const float sy_ = from0_ ? G.sy0 : G.sy1;                                                              \
        const unsigned rowb_ = (unsigned) ((from0_ ? G.Ws0 * G.c0_stride : G.Ws1 * G.c1_stride) * ES);          \
        const int img_ = G.n * Hs_;                                                                            \
        unsigned col_[IPR];                                                                                    \
        if constexpr (!BL) {                                                                                   \
            _Pragma("unroll") for (int j_ = 0; j_ < IPR; ++j_)                                                 \
                col_[j_] = coltab[(from0_ ? 0 : IPR * 64) + j_ * 64 + lane];                                   \
        }                                                                                                      \
        _Pragma("nounroll") for (int q_ = (Q0) + wave; q_ < q1_; q_ += C::NWAVES) {                            \
            if constexpr (BL) {                                                                                \
                halo_bilinear_store<PITCH>(G, q_, lane, (const unsigned char *) a.src0 + soff_, dstb_ + (q_ << 10)); \
            } else {                                                                                           \
                const int row_ = q_ / IPR, seg_ = q_ - row_ * IPR;          /* wave-uniform */                 \
                const int iy_ = G.iy0 + row_ * G.sub;                                                          \
                const bool ok_ = iy_ >= 0 && iy_ < G.Hin;                                                      \
                /* (wave-uniform, but the float path of nearest_src lives in VGPRs: named uniform, else hipcc wraps */ \
                /*  every DMA instruction in a waterfall loop over its scalar offset) */                        \
                const int ys_ = __builtin_amdgcn_readfirstlane(up_ ? nearest_src(iy_, sy_, Hs_) : iy_);        \
                unsigned v_ = col_[0];                                                                         \
                _Pragma("unroll") for (int j_ = 1; j_ < IPR; ++j_) v_ = seg_ == j_ ? col_[j_] : v_;           \
                v_ = (ok_ && !CPN_EXP_HALO_OOB) ? v_ : OOB_LANE;  /* rows above / below the image: every lane reads zeros */ \
                const unsigned s_ = ok_ ? (unsigned) (img_ + ys_) * rowb_ + soff_ : 0u;                        \
                if (CPN_EXP_HALO_NOISSUE) asm volatile("" :: "v"(v_), "s"(s_), "s"((unsigned) (size_t) (dstb_ + (q_ << 10)))); \
                else if (from0_) bdma16<CPN_HALO_AUX>(rs0, v_, s_, dstb_ + (q_ << 10));                        \
                else bdma16<CPN_HALO_AUX>(rs1, v_, s_, dstb_ + (q_ << 10));                                    \
            }                                                                                                  \
        }                                                                                                      \
    }

#define PW_HALO_DMA(CHUNK)                                                                                     \
    {                                                                                                          \
        const int c_ = (CHUNK);                                                                                \
        const unsigned soff_ = (unsigned) (cin0 + c_ * CH) * (unsigned) ES;                                    \
        unsigned char *dstb_ = smem + (c_ & nhb_mask) * halo_buf;                                              \
        _Pragma("unroll") for (int it = 0; it < A_INSTR_WAVE; ++it)                                            \
            if (a_ok[it]) bdma16<CPN_HALO_AUX>(rs0, a_voff[it], soff_, dstb_ + ((wave + it * C::NWAVES) << 10)); \
    }

    // stages the (two) weight slabs of the next step in order into weight buffer BUF
#define W_DMA(BUF)                                                                                             \
    if constexpr (!RW) {                                                                                       \
        _Pragma("unroll") for (int it = 0; it < W_IW; ++it)                                                    \
            bdma16<CPN_W_AUX>(rsw, w_voff[it], wsoff + (unsigned) w_k[it] * item_bytes, smem + w_m0[it] + (BUF) * WBUF); \
        wsoff += IPS * item_bytes;                                                                             \
    }

    // ---- accumulators
    f32x16 acc[WN][WM];
#pragma unroll
    for (int j = 0; j < WN; ++j)
#pragma unroll
        for (int f = 0; f < WM; ++f)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][f][r] = 0.f;

    const int l31 = lane & 31, lhi = lane >> 5;
    // weight fragment: row = wave_n*WN*32 + j*32 + l31 -> (row>>2)&3 == (l31>>2)&3 is lane-constant; the k-half is
    // an XOR 32 on the byte address (records are 64-B aligned, the swizzled part index lives in bits 4-5)
    constexpr int PSEL = CPN_FP8 ? 2 : 1;  // first 16-byte part of lane half lhi: lhi (bf16 k-half 0) | 2*lhi (fp8)
    const unsigned w_lane = (unsigned) ((wave_n * WN * 32 + l31) * REC + (((PSEL * lhi) ^ ((l31 >> 2) & 3)) << 4));
    // halo column of this lane's pixel for tap column 0 (wrap tile: lanes 16..31 live in the second halo segment)
    const int x_lane = (NR ? (l31 & 15) : l31) * S + (wrap ? (l31 >> 4) * (KW - 1) : 0);
    const int row_wave = wave_m * WM * RPF * S;            // halo row of fragment 0 for tap row 0
    constexpr int FRAG_STRIDE = RPF * S * PITCH * REC;     // bytes between the halo rows of consecutive fragments
    const unsigned nr_lane = NR ? (unsigned) ((l31 >> 4) * S * PITCH * REC) : 0u;  // narrow: lanes 16..31 = the fragment's 2nd row

    // ---- software-pipelined main loop -------------------------------------------------------------------------
#ifndef CPN_BACKEDGE_WAIT  // 1 (default): the bf16 loops close every iteration with lgkmcnt(0); 0: the first group's reads stay in flight
#define CPN_BACKEDGE_WAIT 1  // across the back-edge; 2: the wait pinned behind the last MFMA group -- all three within +-1 % (r06 experiments #17)
#endif
    // A step has up to four MFMA groups (item x k-half), each WN + WM fragments and WN*WM MFMAs.  Two fragment
    // register sets alternate (A: groups 0,2; B: groups 1,3): the ds_reads of group g+1 are issued BEFORE the MFMAs
    // of group g, so LDS latency hides behind the matrix pipe (ablation: the non-MFMA skeleton of the un-pipelined
    // loop cost 2.97 ms of the 5.33 ms of a 7x7 head conv and did not overlap with the MFMAs).  The step boundary
    // (vmcnt(0) + barrier + DMA issue for step s+2 + first fragment reads of step s+1) sits between the loads and
    // the MFMAs of the LAST group of step s, whose operands are already in registers.
    // tuning experiments (profiles/): -DCPN_EXP_NOWDMA / -DCPN_EXP_NOHDMA skip the steady-state weight / halo DMA,
    // -DCPN_EXP_NOMFMA keeps the fragment reads alive but issues no MFMA (results are wrong in all three)
#ifdef CPN_EXP_NOWDMA
#define CPN_EXP_W(x)
#else
#define CPN_EXP_W(x) x
#endif
#ifdef CPN_EXP_NOHDMA
#define CPN_EXP_H(x)
#else
#define CPN_EXP_H(x) x
#endif
#ifdef CPN_EXP_NOMFMA
#define CPN_EXP_MMA(ACC, A, B) asm volatile("" ::"v"(A), "v"(B))
#else
#define CPN_EXP_MMA(ACC, A, B) ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A, B, ACC, 0, 0, 0)
#endif

    // byte address (within smem) of this lane's k-half-0 pixel / weight fragment of an item
#define ITEM_PADDR(C_, KY_, KX_)                                                                               \
    ((unsigned) ((min((C_), nchunks - 1) & nhb_mask) * halo_buf + (row_wave + (KY_)) * (PITCH * REC)) +        \
     (unsigned) ((x_lane + (KX_)) * REC) + (unsigned) (((PSEL * lhi) ^ (((x_lane + (KX_)) >> 2) & 3)) << 4) + nr_lane)
    const unsigned lds0 = (unsigned) (size_t) (__attribute__((address_space(3))) unsigned char *) smem;
#define LOAD_GROUP(WF, PF, PADDR, WADDR) load_frags<WN, WM, FRAG_STRIDE>(WF, PF, lds0 + (PADDR), lds0 + (WADDR))
    // MFMAs of a group whose reads were followed by PENDING younger ds_reads (the next group's prefetch)
#define MMA_GROUP(WF, PF, PENDING)                                                                             \
    {                                                                                                          \
        wait_frags<PENDING, WN, WM>(WF, PF);                                                                   \
        _Pragma("unroll") for (int j = 0; j < WN; ++j)                                                         \
            _Pragma("unroll") for (int f = 0; f < WM; ++f) CPN_EXP_MMA(acc[j][f], WF[j], PF[f]);               \
    }
    // Branches of the step boundary that are taken once per chunk (or never: the 1x1 schedules of this instantiation): named unlikely,
    // hipcc places their blocks behind the loop and the common path falls through -- bf16 +0.5 ... +2.8 % (two taken long-distance
    // branches less per step), e4m3 neutral to negative (profiles/r06_kernel_experiments.txt #15): bf16 build only.
#if CPN_FP8
#define CPN_RARE(X) (X)
#else
#define CPN_RARE(X) __builtin_expect((X), 0)
#endif
    // DMA issued at the transition INTO step ST1 (first item IA, flattened index 2*ST1): the weights of step
    // ST1+1 and the halo tiles that step ST1+1 (1x1) / the next chunk (KxK) will need
#define ISSUE_AT_TRANSITION(IA, ST1, CHUNK_CHANGED)                                                            \
    {                                                                                                          \
        const int idx2_ = IPS * (ST1) + IPS; /* first item of step ST1+1 */                                    \
        if (idx2_ < nitems) {                                                                                  \
            if (CPN_RARE(pw_fast)) {                                                                           \
                if (idx2_ < nreal) { CPN_EXP_H(PW_HALO_DMA(idx2_)); }                                          \
                if (idx2_ + 1 < nreal) { CPN_EXP_H(PW_HALO_DMA(idx2_ + 1)); }                                  \
            } else if (CPN_RARE(pw)) {                                                                         \
                if (idx2_ < nreal) { CPN_EXP_H(HALO_DMA(idx2_)); }                                             \
                if (idx2_ + 1 < nreal) { CPN_EXP_H(HALO_DMA(idx2_ + 1)); }                                     \
            }                                                                                                  \
            CPN_EXP_W(W_DMA(((ST1) + 1) & 1));                                                                 \
        }                                                                                                      \
        /* KxK: every chunk before IA.c is completely consumed -> its ring buffer can take chunk IA.c+1 */     \
        CPN_HALO_AT_TRANSITION(IA, CHUNK_CHANGED)                                                              \
    }

    // MODE_BL: the next chunk's tile is blended in registers (global loads + VALU + ds_write), which needs ~50 VGPRs:
    // it is deferred to the END of the loop iteration, where only one fragment set is live (done at the transition,
    // between the loads and the MFMAs of the last group, the e4m3 instantiation spilled 138 VGPRs).  Any point of the
    // step that follows the chunk change is early enough: the tile is first read >= floor(ntaps / 2) steps later
    int bl_pending = -1;
#define CPN_HALO_AT_TRANSITION(IA, CHUNK_CHANGED)                                                              \
    if (CPN_RARE(!pw && (CHUNK_CHANGED) && (IA).c + 1 < nchunks)) {                                            \
        if constexpr (BL) bl_pending = (IA).c + 1;                                                             \
        else { CPN_EXP_H(HALO_DMA((IA).c + 1)); }                                                              \
    }
#define CPN_BL_FLUSH()                                                                                         \
    if constexpr (BL) {                                                                                        \
        if (bl_pending >= 0) {                                                                                 \
            HALO_DMA(bl_pending);                                                                              \
            bl_pending = -1;                                                                                   \
        }                                                                                                      \
    }

    ItemState i0{0, 0, 0};                 // first item of the current step
    ItemState i1 = next_item(i0, KH, KW);  // second item of the current step

    // ---- halo DMA source offsets are separable: the per-lane part depends only on the halo COLUMN (instruction q covers
    // halo row q / IPR, columns (q % IPR) * 16 ..+15: the source column incl. nearest upsampling, the swizzled 16-byte
    // part, out-of-image -> OOB), the row part is wave-uniform and travels in the scalar offset of the buffer
    // instruction.  The column parts of both sources live in a small LDS table (2 x IPR x 64 dwords) filled once:
    // recomputing them per instruction (div / float nearest / bounds, ~50 VALU) cost 6 % of a 3x3 conv and 17 % of
    // the 64-channel layers (profiles/r02_dma_ablation.txt, HCONTIG vs HCONTIGC).
    if (!BL && !FL && !BR) {  // (the 1x1 fast path stages its first tiles through the generic path, too)
        for (int e = tid; e < 2 * IPR * 64; e += C::THREADS) {
            const int s = e >= IPR * 64, r = e - s * IPR * 64, ln = r & 63;
            const int hx = (r >> 6) * 16 + (ln >> 2);
            const int ix = halo_ix(G, hx);
            unsigned v = OOB_LANE;
            if (hx < G.HWreal && ix >= 0 && ix < G.Win) {
                const int xs = (s ? G.up1 : G.up0) ? nearest_src(ix, s ? G.sx1 : G.sx0, s ? G.Ws1 : G.Ws0) : ix;
                v = (unsigned) ((xs * (s ? G.c1_stride : G.c0_stride) + ((ln & 3) ^ ((hx >> 2) & 3)) * EPP) * ES);
            }
            coltab[e] = v;
        }
        __syncthreads();
    }

    // ---- MODE_BR stage 1: the scattered phase conv (CPN_SUBPIXEL_SCATTER) on the low-resolution map, straight into the two halo
    // buffers of the main loop.  Halo row r / column c = full-resolution pixel (Y, X) = (oy0 - 1 + r, ox0 - 1 + c) = phase (py, px)
    // = (Y & 1, X & 1) of low-resolution pixel (Y >> 1, X >> 1); out(i, j; py, px) = sum over the 2 x 2 taps (ty, tx) of
    // W[py, px][ty, tx] . in(i - 1 + py + ty, j - 1 + px + tx).  The 18 x 34 tile holds 9 x 17 = 153 pixels of every phase = 5 MFMA
    // pixel fragments; wave w owns phase w >> 1 and output channels (w & 1) * 32 .. + 31 = main-loop chunk w & 1: 5 accumulators,
    // weights from L2 straight into registers (the scatter op's packed slabs ARE the A-fragment rows), pixel operand from a 12 x 20
    // low-resolution input tile staged by LDS-DMA.  Same K order (chunk-major, tap-minor, k-half-minor), bias, ReLU and bf16
    // rounding as the stand-alone op -> the main loop reads the bits the intermediate tensor would have held.
    if constexpr (BR) {
        // TH = 16 (MODE_BR, 8 waves): 9 rows of every phase = 5 fragments, wave w = phase w >> 1 x channel block w & 1;
        // TH = 8 (MODE_BRF, 4 waves, two workgroups per CU): 5 rows = 3 fragments, wave w = phase w x BOTH channel blocks
        static_assert(BN == 64 && WM == 2 && WN == 2 && (TH == 16 || TH == 8), "bridge stage: the <16,64,2,2> / <8,64,2,2> tile");
        constexpr int NR1 = TH / 2 + 1, NPX1 = NR1 * 17, NF1 = (NPX1 + 31) / 32;   // rows, pixels, fragments of a phase
        constexpr int NJ = TH == 16 ? 1 : 2;                                        // channel blocks per wave
        constexpr int PR = NR1 + 2, PC = 20, PREC = PR * PC, PINSTR = (PREC + 15) / 16;   // input tile (records of 64 B)
        constexpr int PBUF = PINSTR * 1024;
        unsigned char *const pin = smem + ldsW_off + 2 * WBUF + (FL ? 0 : 2 * IPR * 256);
        const int pch = a.pre_cin >> 5;
        const int ly0 = (oy0 >> 1) - 2, lx0 = (ox0 >> 1) - 2;     // low-resolution pixel of input-tile record (0, 0)
        const rsrc_t rsp = make_rsrc(a.pre_src, (unsigned) ((size_t) a.N * a.pre_H * a.pre_W * a.pre_stride * ES));
        for (int c = 0; c < pch; ++c) {
            const unsigned so = (unsigned) (((size_t) n * a.pre_H * a.pre_W * a.pre_stride + c * 32) * ES);
            for (int q = wave; q < PINSTR; q += C::NWAVES) {
                const int idx = (q << 6) + lane, rec = idx >> 2;
                const int r = rec / PC, cc = rec - r * PC;
                const int ly = ly0 + r, lx = lx0 + cc;
                const bool valid = rec < PREC && ly >= 0 && ly < a.pre_H && lx >= 0 && lx < a.pre_W;
                const unsigned vo = valid ? (unsigned) (((ly * a.pre_W + lx) * a.pre_stride + ((idx & 3) ^ ((rec >> 2) & 3)) * EPP) * ES) : OOB_LANE;
                bdma16(rsp, vo, so, pin + c * PBUF + (q << 10));
            }
        }
        // this wave's stage-1 weights: its phase, its channel block(s), per (chunk, tap) item two k-halves from L2 straight into
        // registers (the packed slab rows ARE the A-fragment rows), requested step by step (all sixteen up front measured 3.5 %
        // slower: 214 instead of 171 registers and one L2 burst per tile); the first weight slabs of the MAIN loop are requested
        // here, in front of the stage, instead of behind it
        const int ph = TH == 16 ? wave >> 1 : wave, jb0 = TH == 16 ? (wave & 1) : 0, py = ph >> 1, px = ph & 1;
        const int l31b = lane & 31, lhib = lane >> 5;
        W_DMA(0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        f32x16 acc1[NJ][NF1];
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int f = 0; f < NF1; ++f)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc1[j][f][r] = 0.f;
        // output pixel o = 32 f + lane of this phase = (ri, rj) of its NR1 x 17 grid = low-resolution pixel ((oy0 >> 1) - py + ri,
        // (ox0 >> 1) - px + rj); its tap (ty, tx) reads input row i - 1 + py + ty = input-tile row ri + 1 + ty (likewise the columns)
        int ri[NF1], rj[NF1];
        unsigned base[NF1];
#pragma unroll
        for (int f = 0; f < NF1; ++f) {
            const int o = f * 32 + l31b;
            ri[f] = o < NPX1 ? o / 17 : 0;
            rj[f] = o < NPX1 ? o - ri[f] * 17 : 0;
            base[f] = (unsigned) ((ri[f] + 1) * PC + rj[f] + 1);
        }
        const int nit1 = 4 * pch;  // items of a phase bundle (even)
        const unsigned lds_pin = (unsigned) (size_t) (__attribute__((address_space(3))) unsigned char *) pin;
        const unsigned char *const w1 = (const unsigned char *) a.pre_w + ((size_t) ph * nit1 * 64 + jb0 * 32 + l31b) * REC + (lhib << 4);
        // (one fragment set: the two waves of a SIMD belong to different phases / workgroups and cover each other's read latency)
        frag_t pa0[NF1], pa1[NF1];
#pragma unroll
        for (int s1 = 0; s1 < 8; ++s1) {   // K step s1 = (chunk s1 >> 2, tap s1 & 3): k-half 0 and 1
            if (s1 < nit1) {
                frag_t wA[NJ][2];
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    const unsigned char *wp = w1 + ((size_t) s1 * 64 + j * 32) * REC;
                    wA[j][0] = *(const frag_t *) wp;
                    wA[j][1] = *(const frag_t *) (wp + 32);
                }
#ifndef CPN_BR_NOSTAGE1  // (ablation: no stage-1 reads / MFMAs -- wrong results by construction)
                const int toff = ((s1 >> 1) & 1) * PC + (s1 & 1);
#pragma unroll
                for (int f = 0; f < NF1; ++f) {
                    const unsigned rec = base[f] + toff;
                    const unsigned ad = lds_pin + (unsigned) ((s1 >> 2) * PBUF) + rec * REC + (((unsigned) lhib ^ ((rec >> 2) & 3u)) << 4);
                    ds_read16<0>(pa0[f], ad);
                    ds_read16<0>(pa1[f], ad ^ 32u);
                }
                if constexpr (NF1 == 5)
                    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(pa0[0]), "+v"(pa0[1]), "+v"(pa0[2]), "+v"(pa0[3]), "+v"(pa0[4]),
                                 "+v"(pa1[0]), "+v"(pa1[1]), "+v"(pa1[2]), "+v"(pa1[3]), "+v"(pa1[4]));
                else
                    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(pa0[0]), "+v"(pa0[1]), "+v"(pa0[2]), "+v"(pa1[0]), "+v"(pa1[1]), "+v"(pa1[2]));
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
#pragma unroll
                    for (int f = 0; f < NF1; ++f) acc1[j][f] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wA[j][0], pa0[f], acc1[j][f], 0, 0, 0);
#pragma unroll
                    for (int f = 0; f < NF1; ++f) acc1[j][f] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wA[j][1], pa1[f], acc1[j][f], 0, 0, 0);
                }
#endif
            }
        }
        // bias + ReLU -> bf16 -> the main loop's halo record of (row 1 - py + 2 ri, column 1 - px + 2 rj), chunk = channel block;
        // pixels outside the image are the 3x3 conv's zero padding
        typedef __attribute__((address_space(3))) u32x2 lds_u32x2_t;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int jb = jb0 + j;
            float b1[4][4];
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int e = 0; e < 4; ++e) b1[q][e] = a.pre_b ? a.pre_b[jb * 32 + 8 * q + 4 * lhib + e] : 0.f;
#pragma unroll
            for (int f = 0; f < NF1; ++f) {
                const int o = f * 32 + l31b;
                if (o >= NPX1) continue;
                const int r = 1 - py + 2 * ri[f], cc = 1 - px + 2 * rj[f];
                const int Y = oy0 - 1 + r, X = ox0 - 1 + cc;
                const bool in_img = Y >= 0 && Y < a.Hin && X >= 0 && X < a.Win;
                unsigned char *rec = smem + jb * halo_buf + (r * PITCH + cc) * REC + lhib * 8;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    float v[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        v[e] = acc1[j][f][q * 4 + e] + b1[q][e];
                        v[e] = in_img ? __int_as_float(max(__float_as_int(v[e]), 0)) : 0.f;   // (the stand-alone epilogue's ReLU)
                    }
                    u32x2 w2;
                    w2.x = pack_bf16x2(v[0], v[1]);
                    w2.y = pack_bf16x2(v[2], v[3]);
                    *(lds_u32x2_t *) (rec + ((q ^ ((cc >> 2) & 3)) << 4)) = w2;
                }
            }
        }
    }

#ifdef CPN_EXP_CLOCK  // (probe: shader clock of the main loop = s_memtime ticks per 100-MHz s_memrealtime tick, one workgroup per launch)
    const unsigned long long clk_c0 = __builtin_readcyclecounter(), clk_r0 = __builtin_amdgcn_s_memrealtime();
#endif
    // ---- prologue: stage step 0, open it, stage step 1, first fragment reads
    HALO_DMA(0);
    if (pw && nchunks > 1) { HALO_DMA(1); }
    if constexpr (!BR) W_DMA(0);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");  // (lgkmcnt: the MODE_BL halo is written with ds_write)
    __builtin_amdgcn_s_barrier();
    ISSUE_AT_TRANSITION(i0, 0, true);
    CPN_BL_FLUSH();

#if CPN_FP8
    // fp8: an item = 64 channels x 1 tap.  A lane's operand is the 32 contiguous bytes of its record half = two
    // 16-byte parts X (part 2*lhi) and Y (the next part = byte address ^ 16); ONE K=64 MFMA consumes X|Y.  Two
    // operand sets (item 0 / item 1 of the step) alternate so that the reads of item i+1 are in flight while the
    // MFMAs of item i issue; the step boundary sits in front of the MFMAs of item 1, whose operands are in registers.
    frag_t wX0[WN], pX0[WM], wY0[WN], pY0[WM], wX1[WN], pX1[WM], wY1[WN], pY1[WM];
    constexpr int NF = WN + WM;  // ds_reads per part set
#define CAT8(LO, HI) __builtin_shufflevector(LO, HI, 0, 1, 2, 3, 4, 5, 6, 7)
#define MMA8(WX, WY, PX, PY, PENDING)                                                                          \
    {                                                                                                          \
        wait_frags<PENDING, WN, WM>(WY, PY);                                                                   \
        _Pragma("unroll") for (int j = 0; j < WN; ++j)                                                         \
            _Pragma("unroll") for (int f = 0; f < WM; ++f)                                                     \
                acc[j][f] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(                                   \
                    CAT8(WX[j], WY[j]), CAT8(PX[f], PY[f]), acc[j][f], 0, 0, 0, 127, 0, 127);                  \
    }
    unsigned pa = ITEM_PADDR(i0.c, i0.ky, i0.kx);
    unsigned wa = (unsigned) ldsW_off + w_lane;
#ifndef CPN_FP8_HALFSETS
    // Round 6: WHOLE operand sets alternate (set 0 = X|Y of a step's first item, set 1 = of its second): the twelve reads of the
    // next item are issued in front of the eight MFMAs of the current one, so no MFMA group waits for a read that was issued right
    // in front of it.  (Rounds 3-5 alternated half sets -- Y of item i and X of item i+1 in flight per group: the MFMAs of item 0
    // waited for Y0 issued two instructions earlier, the step boundary for Y1 issued right in front of it: two exposed LDS
    // latencies per step with BOTH waves of a SIMD in lockstep behind the workgroup barrier, MFMA busy 0.64 of the CU cycles at
    // all-zero operands, profiles/r05_kernel_experiments.txt #8.)  Same registers, same K order per accumulator: bit-identical.
    // The step boundary's DMA issue (four 1-KiB LDS-DMA instructions per wave + the chunk's halo rows) and the reads of the next
    // step's first item sit BETWEEN the two halves of item 1's MFMA group, whose operands are in registers.
#define MMA4(WX, WY, PX, PY, J)                                                                                \
    _Pragma("unroll") for (int f = 0; f < WM; ++f)                                                             \
        acc[J][f] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(                                           \
            CAT8(WX[J], WY[J]), CAT8(PX[f], PY[f]), acc[J][f], 0, 0, 0, 127, 0, 127);
#define PIN_ACC(J) asm volatile("" : "+v"(acc[J][0]), "+v"(acc[J][1]), "+v"(acc[J][2]), "+v"(acc[J][3]))
    static_assert(WN == 2 || WN == 1, "fp8 main loop: one or two weight fragments per wave");
    LOAD_GROUP(wX0, pX0, pa, wa);
    LOAD_GROUP(wY0, pY0, pa ^ 16u, wa ^ 16u);
    for (int st = 0; st + 1 < nsteps; ++st) {
#if CPN_WAVE_PRIO
        // Alternating wave priority (round 6): the two waves of a SIMD (w and w + NWAVES / 2) share its matrix pipe, arbitrated by
        // priority, then AGE -- at equal priority the older wave's MFMAs always go first, it reaches the barrier ~900 cycles before
        // its partner (s_memtime stamps, profiles/r06_kernel_experiments.txt) and idles there while the partner runs its remaining
        // MFMA groups and gaps alone.  From the barrier to the middle of a step the younger half has priority, from there to the
        // barrier the older half: each wave gets its MFMA groups through while the other is in its DMA-issue / fragment-read gaps.
        if (prio_wave) { if (wave >= C::NWAVES / 2) __builtin_amdgcn_s_setprio(0); else __builtin_amdgcn_s_setprio(1); }
#endif
        const unsigned pa1 = ITEM_PADDR(i1.c, i1.ky, i1.kx), wa1 = wa + WITEM;
        LOAD_GROUP(wX1, pX1, pa1, wa1);               // item 1 (both parts) in flight behind ...
        LOAD_GROUP(wY1, pY1, pa1 ^ 16u, wa1 ^ 16u);
        wait_frags<2 * NF, WN, WM>(wX0, pX0);         // ... item 0, whose reads have returned
        MMA8(wX0, wY0, pX0, pY0, 2 * NF);             // item 0
        if constexpr (WM == 4 && WN == 2) {           // (issued in FRONT of the boundary: 8 x 64 cycles of queued matrix work
            PIN_ACC(0);                               // cover the wait for the DMA and the barrier; left to itself hipcc sinks
            PIN_ACC(1);                               // seven of the eight behind the barrier)
        }
        const ItemState n0i = next_item(i1, KH, KW);  // first item of step st+1
        wait_all<WN, WM>(wY1, pY1);                   // my DMA for step st+1 landed, all my LDS reads returned
        wait_frags<0, WN, WM>(wX1, pX1);              // (names the X set of item 1 as complete, too)
#ifndef CPN_EXP_NOBAR  // (tuning ablation: wrong results)
        __builtin_amdgcn_s_barrier();
#endif
#if CPN_WAVE_PRIO
        if (prio_wave) { if (wave >= C::NWAVES / 2) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0); }
#endif
        if constexpr (WM == 4 && WN == 2) {
            MMA4(wX1, wY1, pX1, pY1, 0);              // item 1, first weight fragment (operands in registers)
            PIN_ACC(0);
        }
        ISSUE_AT_TRANSITION(n0i, st + 1, n0i.c != i0.c);
        pa = ITEM_PADDR(n0i.c, n0i.ky, n0i.kx);
        wa = (unsigned) (ldsW_off + ((st + 1) & 1) * WBUF) + w_lane;
        LOAD_GROUP(wX0, pX0, pa, wa);                 // item 0 of step st+1 (both parts)
        LOAD_GROUP(wY0, pY0, pa ^ 16u, wa ^ 16u);
        if constexpr (WM == 4 && WN == 2) {
            asm volatile("" : "+v"(wX1[1]), "+v"(wY1[1]));  // (keeps the MFMAs below behind the reads above)
            MMA4(wX1, wY1, pX1, pY1, 1);              // item 1, second weight fragment
        } else {
            MMA8(wX1, wY1, pX1, pY1, 2 * NF);
        }
        CPN_BL_FLUSH();
        i0 = n0i;
        i1 = next_item(n0i, KH, KW);
    }
    {   // last step (two items: an odd item count was padded with a zero slab)
        const unsigned pa1 = ITEM_PADDR(i1.c, i1.ky, i1.kx), wa1 = wa + WITEM;
        LOAD_GROUP(wX1, pX1, pa1, wa1);
        LOAD_GROUP(wY1, pY1, pa1 ^ 16u, wa1 ^ 16u);
        wait_frags<2 * NF, WN, WM>(wX0, pX0);
        MMA8(wX0, wY0, pX0, pY0, 2 * NF);
        wait_frags<0, WN, WM>(wX1, pX1);
        MMA8(wX1, wY1, pX1, pY1, 0);
    }
#undef MMA4
#undef PIN_ACC
#else
    LOAD_GROUP(wX0, pX0, pa, wa);
    wait_frags<0, WN, WM>(wX0, pX0);
    for (int st = 0; st + 1 < nsteps; ++st) {
        LOAD_GROUP(wY0, pY0, pa ^ 16u, wa ^ 16u);     // item 0, second part
        const unsigned pa1 = ITEM_PADDR(i1.c, i1.ky, i1.kx), wa1 = wa + WITEM;
        LOAD_GROUP(wX1, pX1, pa1, wa1);               // item 1, first part
        MMA8(wX0, wY0, pX0, pY0, NF);                 // item 0
        LOAD_GROUP(wY1, pY1, pa1 ^ 16u, wa1 ^ 16u);   // item 1, second part
        const ItemState n0i = next_item(i1, KH, KW);  // first item of step st+1
        wait_all<WN, WM>(wY1, pY1);                   // my DMA for step st+1 landed, all my LDS reads returned
        wait_frags<0, WN, WM>(wX1, pX1);              // (names the X set of item 1 as complete, too)
        __builtin_amdgcn_s_barrier();
        ISSUE_AT_TRANSITION(n0i, st + 1, n0i.c != i0.c);
        pa = ITEM_PADDR(n0i.c, n0i.ky, n0i.kx);
        wa = (unsigned) (ldsW_off + ((st + 1) & 1) * WBUF) + w_lane;
        LOAD_GROUP(wX0, pX0, pa, wa);                 // first part of step st+1
        MMA8(wX1, wY1, pX1, pY1, NF);                 // item 1 (operands already in registers)
        wait_frags<0, WN, WM>(wX0, pX0);
        CPN_BL_FLUSH();
        i0 = n0i;
        i1 = next_item(n0i, KH, KW);
    }
    {   // last step (two items: an odd item count was padded with a zero slab)
        LOAD_GROUP(wY0, pY0, pa ^ 16u, wa ^ 16u);
        const unsigned pa1 = ITEM_PADDR(i1.c, i1.ky, i1.kx), wa1 = wa + WITEM;
        LOAD_GROUP(wX1, pX1, pa1, wa1);
        MMA8(wX0, wY0, pX0, pY0, NF);
        LOAD_GROUP(wY1, pY1, pa1 ^ 16u, wa1 ^ 16u);
        wait_frags<NF, WN, WM>(wX1, pX1);
        MMA8(wX1, wY1, pX1, pY1, 0);
    }
#endif
#undef MMA8
#undef CAT8
#else
    if constexpr (RW) {
        // ---- MODE_S1R: weight fragments straight from L2 into registers in MFMA layout (the packed slab rows are the
        // fragment rows; lane -> row l31 of the 32-row fragment, 16-byte part 2*khalf + lhi, same swizzle as the LDS
        // tiles).  Four register sets (item 0/1 x k-half 0/1) are refilled for the step after next as soon as their
        // MFMAs have issued: ~1.75 items of prefetch distance, vmcnt tracked by the compiler.  The LDS holds only
        // the halo tiles: one third fewer LDS reads, no weight DMA, and the workgroup barrier is needed only twice
        // per 32-channel chunk (buffer hand-over + tile landed) instead of once per step.
        frag_t W0a[WN], W0b[WN], W1a[WN], W1b[WN], pA[WM], pB[WM];
        // (the packed slabs are not swizzled: part p of a row lives at byte 16 p; the XOR swizzle is applied by the DMA path only)
        const unsigned wl0 = (unsigned) ((wave_n * WN * 32 + l31) * REC + (lhi << 4)), wl1 = wl0 + 32u;
#define VLOADW(DST, WL, SOFF)                                                                                  \
    _Pragma("unroll") for (int j = 0; j < WN; ++j)                                                             \
        DST[j] = __builtin_bit_cast(frag_t, __builtin_amdgcn_raw_buffer_load_b128(rsw, (int) ((WL) + j * 32 * REC), (int) (SOFF), 0))
#define LOADP(PF, PADDR) load_pfrags<WM, FRAG_STRIDE>(PF, lds0 + (PADDR))
#define MMAP(WF, PF, PENDING)                                                                                  \
    {                                                                                                          \
        wait_pfrags<PENDING, WM>(PF);                                                                          \
        _Pragma("unroll") for (int j = 0; j < WN; ++j)                                                         \
            _Pragma("unroll") for (int f = 0; f < WM; ++f) CPN_EXP_MMA(acc[j][f], WF[j], PF[f]);               \
    }
        unsigned so = wsoff;  // slab of the first item of the current step
        VLOADW(W0a, wl0, so); VLOADW(W0b, wl1, so);
        VLOADW(W1a, wl0, so + item_bytes); VLOADW(W1b, wl1, so + item_bytes);
        unsigned pa = ITEM_PADDR(i0.c, i0.ky, i0.kx);
        LOADP(pA, pa);
        bool land_due = nchunks > 1;  // chunk 1's tile was issued in the prologue
        for (int st = 0; st + 1 < nsteps; ++st) {
            so += 2 * item_bytes;
            LOADP(pB, pa ^ 32u);
            MMAP(W0a, pA, WM);
            VLOADW(W0a, wl0, so);
            pa = ITEM_PADDR(i1.c, i1.ky, i1.kx);
            LOADP(pA, pa);
            MMAP(W0b, pB, WM);
            VLOADW(W0b, wl1, so);
            LOADP(pB, pa ^ 32u);
            MMAP(W1a, pA, WM);
            VLOADW(W1a, wl0, so + item_bytes);
            const ItemState n0i = next_item(i1, KH, KW);
            const bool changed = n0i.c != i0.c;
            if (changed || land_due) {
                // all my reads of the finished chunk returned; my halo DMA of the previous hand-over (>= 8 weight loads
                // older than now: VMEM returns in order) landed; after the barrier that holds for every wave
                wait_pfrags<0, WM>(pB);
                asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                land_due = false;
                if constexpr (PW) {
                    // pointwise: every item is a chunk of its own -- stage the two chunks of step st + 2 into the ring
                    // slots step st occupied (everybody's reads of step st returned: barrier above)
                    const int idx2_ = 2 * (st + 1) + 2;
                    if (idx2_ < nreal) {
                        if (pw_fast) { PW_HALO_DMA(idx2_); } else { HALO_DMA(idx2_); }
                        land_due = true;
                    }
                    if (idx2_ + 1 < nreal) {
                        if (pw_fast) { PW_HALO_DMA(idx2_ + 1); } else { HALO_DMA(idx2_ + 1); }
                    }
                } else if (changed && n0i.c + 1 < nchunks) {
                    HALO_DMA(n0i.c + 1);
                    land_due = true;
                }
            }
            pa = ITEM_PADDR(n0i.c, n0i.ky, n0i.kx);
            LOADP(pA, pa);
            MMAP(W1b, pB, WM);
            VLOADW(W1b, wl1, so + item_bytes);
            wait_pfrags<0, WM>(pA);
            i0 = n0i;
            i1 = next_item(n0i, KH, KW);
        }
        LOADP(pB, pa ^ 32u);
        MMAP(W0a, pA, WM);
        pa = ITEM_PADDR(i1.c, i1.ky, i1.kx);
        LOADP(pA, pa);
        MMAP(W0b, pB, WM);
        LOADP(pB, pa ^ 32u);
        MMAP(W1a, pA, WM);
        MMAP(W1b, pB, 0);
#undef VLOADW
#undef LOADP
#undef MMAP
    } else {
    frag_t wA[WN], pA[WM], wB[WN], pB[WM];
    constexpr int NF = WN + WM;  // ds_reads per group
    unsigned pa = ITEM_PADDR(i0.c, i0.ky, i0.kx);
    unsigned wa = (unsigned) ldsW_off + w_lane;
    LOAD_GROUP(wA, pA, pa, wa);

    if constexpr (IPS == 4) {
    // ---- MODE_S1Q: four items per step -- the two-item body below with two more (item, k-half) group pairs in front of the
    // boundary; the fragment sets still alternate A / B and the boundary still sits between the loads and the MFMAs of the LAST group
    // (ONE running item state: four live ItemStates next to the kernel's other scalars cost 151 SGPR spills)
    ItemState cur = i0;
#define QSTEP_NEXT() { cur = next_item(cur, KH, KW); pa = ITEM_PADDR(cur.c, cur.ky, cur.kx); wa += WITEM; }
#define QSTEP_HEAD()                                                                                           \
        LOAD_GROUP(wB, pB, pa ^ 32u, wa ^ 32u);      /* item 0, k-half 1 */                                     \
        MMA_GROUP(wA, pA, NF);                        /* item 0, k-half 0 */                                     \
        QSTEP_NEXT();                                                                                           \
        LOAD_GROUP(wA, pA, pa, wa);                   /* item 1, k-half 0 */                                     \
        MMA_GROUP(wB, pB, NF);                        /* item 0, k-half 1 */                                     \
        LOAD_GROUP(wB, pB, pa ^ 32u, wa ^ 32u);      /* item 1, k-half 1 */                                     \
        MMA_GROUP(wA, pA, NF);                        /* item 1, k-half 0 */                                     \
        QSTEP_NEXT();                                                                                           \
        LOAD_GROUP(wA, pA, pa, wa);                   /* item 2, k-half 0 */                                     \
        MMA_GROUP(wB, pB, NF);                        /* item 1, k-half 1 */                                     \
        LOAD_GROUP(wB, pB, pa ^ 32u, wa ^ 32u);      /* item 2, k-half 1 */                                     \
        MMA_GROUP(wA, pA, NF);                        /* item 2, k-half 0 */                                     \
        QSTEP_NEXT();                                                                                           \
        LOAD_GROUP(wA, pA, pa, wa);                   /* item 3, k-half 0 */                                     \
        MMA_GROUP(wB, pB, NF);                        /* item 2, k-half 1 */                                     \
        LOAD_GROUP(wB, pB, pa ^ 32u, wa ^ 32u);      /* item 3, k-half 1 */                                     \
        MMA_GROUP(wA, pA, NF);                        /* item 3, k-half 0 */
    int c_first = 0;                                  // chunk of the current step's first item
    for (int st = 0; st + 1 < nsteps; ++st) {
        QSTEP_HEAD();
        cur = next_item(cur, KH, KW);                 // first item of step st+1
        wait_all<WN, WM>(wB, pB);                     // my DMA for step st+1 landed, all my LDS reads of step st returned
        __builtin_amdgcn_s_barrier();
        ISSUE_AT_TRANSITION(cur, st + 1, cur.c != c_first);
        c_first = cur.c;
        pa = ITEM_PADDR(cur.c, cur.ky, cur.kx);
        wa = (unsigned) (ldsW_off + ((st + 1) & 1) * WBUF) + w_lane;
        LOAD_GROUP(wA, pA, pa, wa);                   // first group of step st+1
        MMA_GROUP(wB, pB, NF);                        // item 3, k-half 1 (operands already in registers)
#if CPN_BACKEDGE_WAIT == 1                            // (see the two-item loop below)
        wait_frags<0, WN, WM>(wA, pA);
#endif
    }
    QSTEP_HEAD();                                     // last step
    MMA_GROUP(wB, pB, 0);
#undef QSTEP_NEXT
#undef QSTEP_HEAD
    } else {
    // every step that is followed by another step holds two items: the loop body is branch-free with respect to
    // the accumulators and fragment sets (conditional MFMA groups made hipcc rename/spill accumulators)
#define PIN_ACC_ALL()                                                                                          \
    _Pragma("unroll") for (int j_ = 0; j_ < WN; ++j_)                                                          \
        _Pragma("unroll") for (int f_ = 0; f_ < WM; ++f_) asm volatile("" : "+v"(acc[j_][f_]))
    for (int st = 0; st + 1 < nsteps; ++st) {
        LOAD_GROUP(wB, pB, pa ^ 32u, wa ^ 32u);      // item 0, k-half 1
        MMA_GROUP(wA, pA, NF);                        // item 0, k-half 0
        pa = ITEM_PADDR(i1.c, i1.ky, i1.kx);
        wa += WITEM;
        LOAD_GROUP(wA, pA, pa, wa);                   // item 1, k-half 0
        MMA_GROUP(wB, pB, NF);                        // item 0, k-half 1
        LOAD_GROUP(wB, pB, pa ^ 32u, wa ^ 32u);      // item 1, k-half 1
        MMA_GROUP(wA, pA, NF);                        // item 1, k-half 0
        const ItemState n0i = next_item(i1, KH, KW);  // first item of step st+1
        // step boundary: my DMA for step st+1 has landed and all my LDS reads of step st are complete ...
        wait_all<WN, WM>(wB, pB);
        __builtin_amdgcn_s_barrier();                 // ... and so have everybody else's
        ISSUE_AT_TRANSITION(n0i, st + 1, n0i.c != i0.c);  // stage step st+2 (its buffers were last read in step st)
        pa = ITEM_PADDR(n0i.c, n0i.ky, n0i.kx);
        wa = (unsigned) (ldsW_off + ((st + 1) & 1) * WBUF) + w_lane;
        LOAD_GROUP(wA, pA, pa, wa);                   // first group of step st+1
        MMA_GROUP(wB, pB, NF);                        // last group of step st (operands already in registers)
#if CPN_BACKEDGE_WAIT == 1
        wait_frags<0, WN, WM>(wA, pA);                // nothing is in flight across the loop back-edge
#elif CPN_BACKEDGE_WAIT == 2
        PIN_ACC_ALL();                                // the same wait, held behind the eight MFMAs of the group
        wait_frags<0, WN, WM>(wA, pA);
#endif
        CPN_BL_FLUSH();
        i0 = n0i;
        i1 = next_item(n0i, KH, KW);
    }
    // last step (two items: an odd item count was padded with a zero slab)
    LOAD_GROUP(wB, pB, pa ^ 32u, wa ^ 32u);
    MMA_GROUP(wA, pA, NF);
    pa = ITEM_PADDR(i1.c, i1.ky, i1.kx);
    wa += WITEM;
    LOAD_GROUP(wA, pA, pa, wa);
    MMA_GROUP(wB, pB, NF);
    LOAD_GROUP(wB, pB, pa ^ 32u, wa ^ 32u);
    MMA_GROUP(wA, pA, NF);
    MMA_GROUP(wB, pB, 0);
    }
    }
#endif
#undef HALO_DMA
#undef HALO_DMA_RANGE
#undef HALO_DMA_RANGE_ROWS
#undef CPN_HALO_AT_TRANSITION
#undef CPN_BL_FLUSH
#undef PW_HALO_DMA
#undef W_DMA
#undef LOAD_GROUP
#undef MMA_GROUP
#undef ISSUE_AT_TRANSITION
#undef ITEM_PADDR

#if defined(CPN_EXP_CLOCK) && CPN_EXP_CLOCK == 1
    unsigned long long clk_epi[3] = {0, 0, 0};
#define CPN_EPI_STAMP(I) clk_epi[I] = __builtin_readcyclecounter()
#else
#define CPN_EPI_STAMP(I)
#endif
#ifdef CPN_EXP_CLOCK
    const unsigned long long clk_loop_end = __builtin_readcyclecounter();
    unsigned long long clk_print = 0;  // (the printf below is a host call: its cycles are taken out of the epilogue phase)
    if (tid == 0 && blockIdx.x == gridDim.x / 2 && blockIdx.y == gridDim.y / 2 && blockIdx.z == 0) {
        const unsigned long long dc = clk_loop_end - clk_c0, dr = __builtin_amdgcn_s_memrealtime() - clk_r0;
#if CPN_EXP_CLOCK == 2 && !CPN_FP8  // the probe library (libcpn_hip_clock.so): sums per tap count, read by cpn_debug_clock_probe
        unsigned long long *g = g_clock_probe + 5 * (ntaps == 49 ? 0 : (ntaps == 9 ? 1 : 2));
        // matrix-pipe cycles the loop's MFMAs occupy on one SIMD x 4: a 32x32x16 bf16 MFMA = 32 cycles, 2 k-halves x WN x WM per item
        // and wave, NWAVES / 4 waves per SIMD, two co-resident workgroups in the flat modes
        const unsigned long long pipe4 = (unsigned long long) nsteps * IPS * (2 * WN * WM * 32) * C::NWAVES *
                                         ((MODE == MODE_S1F || MODE == MODE_BRF) ? 2 : 1);
        atomicAdd(g, dc); atomicAdd(g + 1, dr); atomicAdd(g + 2, (unsigned long long) nsteps); atomicAdd(g + 3, 1ull);
        atomicAdd(g + 4, pipe4);
#else
        printf("CLK steps %d core_ticks %llu ref_ticks %llu -> %.0f MHz, %.0f core cycles / step\n", nsteps, dc, dr, 100. * dc / dr, (double) dc / nsteps);
        clk_print = __builtin_readcyclecounter() - clk_loop_end;
#endif
    }
#endif
    // ---- epilogue
#ifdef CPN_EXP_NOEPI
    if (a.N < 0)
#endif
    if (a.out_mode == OUT_BF16_NHWC) {
        // NHWC bf16 output through a per-wave fp32 LDS staging tile: the MFMA D layout gives every lane 4 channels of
        // 16 different pixels (8-byte pieces of sixteen 128-B lines: measured 20-70 % of the run time of the 1x1 /
        // 3x3 layers); after the transpose a lane owns 8 consecutive channels of one pixel, a wave instruction
        // writes whole 64/128-B channel runs with 16-B stores, and the residual is read the same way.
        constexpr int CW = WN * 32;                  // channels per wave
        constexpr int SPITCH = CW * 4 + 16;          // staging row pitch in bytes (fp32 + pad: conflict-free b128)
        constexpr int LPP = CW / 8;                  // lanes per pixel in the store pass (8 channels each)
        constexpr int PPI = 64 / LPP;                // pixels per store instruction
        __syncthreads();                             // every wave is done reading the main-loop LDS buffers
        unsigned char *stg = smem + wave * (32 * SPITCH);
        const int part = lane % LPP, prow = lane / LPP;
        const int cob = n0 + wave_n * CW + part * 8;  // first of this lane's 8 channels within the bundle
        const bool ch_ok = cob < cout_b;
        const bool scat = a.phase == 2;  // phase g scatters to pixels (2 oy + py, 2 ox + px); channels / bias are shared
        const int co = (scat ? 0 : g * cout_b) + cob;
        const int spy = scat ? (g >> 1) : 0, spx = scat ? (g & 1) : 0, ssh = scat ? 1 : 0;
        float bias8[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) bias8[e] = 0.f;
        if (a.bias && ch_ok) {
            const float4 b0 = *(const float4 *) (a.bias + co), b1 = *(const float4 *) (a.bias + co + 4);
            bias8[0] = b0.x; bias8[1] = b0.y; bias8[2] = b0.z; bias8[3] = b0.w;
            bias8[4] = b1.x; bias8[5] = b1.y; bias8[6] = b1.z; bias8[7] = b1.w;
        }
#if CPN_FP8
        // fp8: acc = sum of e4m3 code products; mult = per-output-channel weight scale (the input scale is folded into
        // the weights before they are quantised); outputs are stored as e4m3 codes of value * out_inv_scale
        float mult8[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) mult8[e] = 1.f;
        if (a.mult && ch_ok) {
            const float4 m0 = *(const float4 *) (a.mult + co), m1 = *(const float4 *) (a.mult + co + 4);
            mult8[0] = m0.x; mult8[1] = m0.y; mult8[2] = m0.z; mult8[3] = m0.w;
            mult8[4] = m1.x; mult8[5] = m1.y; mult8[6] = m1.z; mult8[7] = m1.w;
        }
#define CPN_V8(V0, V1)                                                                                              \
    {V0.x * mult8[0] + bias8[0], V0.y * mult8[1] + bias8[1], V0.z * mult8[2] + bias8[2], V0.w * mult8[3] + bias8[3],    \
     V1.x * mult8[4] + bias8[4], V1.y * mult8[5] + bias8[5], V1.z * mult8[6] + bias8[6], V1.w * mult8[7] + bias8[7]}
#else
#define CPN_V8(V0, V1)                                                                                              \
    {V0.x + bias8[0], V0.y + bias8[1], V0.z + bias8[2], V0.w + bias8[3],                                            \
     V1.x + bias8[4], V1.y + bias8[5], V1.z + bias8[6], V1.w + bias8[7]}
#endif
        // fast path (whole tile inside the image and the channel range, residual at full resolution): the 1x1 /
        // grouped layers are HBM-bound and their epilogue was VALU/issue-bound (~80 instructions per 16-B store: 64-bit
        // address math, per-lane bounds, software bf16 rounding, a vmcnt(0) stall per residual load).  Here the row
        // bases are wave-uniform, lane offsets are 32-bit, ALL residual loads of the wave are issued before the first
        // staging pass, and a store costs ~25 instructions.
        constexpr int NIT = 32 / PPI;
        // bytes per element of the destination / the residual (fp8 plans: bf16 partial sums of a sub-pixel triple are "wide")
        const bool dwide = CPN_FP8 && a.dst_wide, rwide = CPN_FP8 && a.res_wide;
        const int DES = dwide ? 2 : ES, RES = rwide ? 2 : ES;
        const bool full_tile = oy0 + TH <= a.Hout && ox0 + TW <= a.Wout && n0 + BN <= cout_b && a.res_up != 1;
        if (full_tile) {
            const unsigned lane_off = (unsigned) ((((prow << ssh) + spx) * a.dst_stride + a.dst_coff + co) * DES);
            const unsigned step = (unsigned) ((PPI << ssh) * a.dst_stride * DES);
            const bool has_res = a.res != nullptr;
#if CPN_FP8
            u32x4 rr[WM][NIT];  // (16 bytes per entry: a wide residual is bf16; e4m3 codes use .x / .y)
#else
            store8_t rr[WM][NIT];
#endif
            if (has_res) {
                // res_up 2: phase tensor [Hout/2][Wout/2][4 * res_cph] read pixel-shuffled -- pixel p of the row sits in
                // low-resolution pixel p >> 1, phase column p & 1 (ox0 and PPI are even); the row phase is wave-uniform
                static_assert(PPI % 2 == 0, "pixel-shuffled residual: even pixel count per store instruction");
                const bool shuf = a.res_up == 2;
                const unsigned rlane_off = shuf ? (unsigned) (((prow >> 1) * a.res_stride + (prow & 1) * a.res_cph + co) * RES)
                                                : (unsigned) ((prow * a.res_stride + co) * RES);
                const unsigned rstep = (unsigned) ((shuf ? PPI / 2 : PPI) * a.res_stride * RES);
#pragma unroll
                for (int f = 0; f < WM; ++f) {
                    const int oy = oy0 + wave_m * WM + f;
                    const unsigned char *rrow = (const unsigned char *) a.res +
                        (shuf ? (((size_t) n * a.Hr + (oy >> 1)) * a.Wr + (ox0 >> 1)) * (size_t) a.res_stride * RES +
                                    (size_t) ((oy & 1) * 2 * a.res_cph) * RES
                              : (((size_t) n * a.Hout + oy) * a.Wout + ox0) * (size_t) a.res_stride * RES);
#if CPN_FP8
                    if (rwide) {
#pragma unroll
                        for (int it = 0; it < NIT; ++it) rr[f][it] = *(const u32x4 *) (rrow + rlane_off + it * rstep);
                    } else {
#pragma unroll
                        for (int it = 0; it < NIT; ++it) {
                            const store8_t r8 = *(const store8_t *) (rrow + rlane_off + it * rstep);
                            rr[f][it].x = r8.x; rr[f][it].y = r8.y;
                        }
                    }
#else
#pragma unroll
                    for (int it = 0; it < NIT; ++it) rr[f][it] = *(const store8_t *) (rrow + rlane_off + it * rstep);
#endif
                }
            }
#pragma unroll
            for (int f = 0; f < WM; ++f) {
                const int oy = oy0 + wave_m * WM + f;
                unsigned char *drow = (unsigned char *) a.dst +
                                      (((size_t) n * (a.Hout << ssh) + ((oy << ssh) + spy)) * (a.Wout << ssh) + (ox0 << ssh)) *
                                          (size_t) a.dst_stride * DES;
#pragma unroll
                for (int j = 0; j < WN; ++j)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        float4 v;
                        v.x = acc[j][f][q * 4 + 0]; v.y = acc[j][f][q * 4 + 1];
                        v.z = acc[j][f][q * 4 + 2]; v.w = acc[j][f][q * 4 + 3];
                        *(float4 *) (stg + l31 * SPITCH + (j * 32 + 8 * q + 4 * lhi) * 4) = v;
                    }
#pragma unroll
                for (int it = 0; it < NIT; ++it) {
                    const int p = it * PPI + prow;
                    const float4 v0 = *(const float4 *) (stg + p * SPITCH + part * 32);
                    const float4 v1 = *(const float4 *) (stg + p * SPITCH + part * 32 + 16);
                    float v[8] = CPN_V8(v0, v1);
#if CPN_FP8
                    if (has_res) {
                        if (rwide) add_res8_wide(v, rr[f][it]);
                        else { store8_t r8; r8.x = rr[f][it].x; r8.y = rr[f][it].y; add_res8(v, r8, a.res_scale); }
                    }
#else
                    if (has_res) add_res8(v, rr[f][it], a.res_scale);
#endif
                    if (a.act == ACT_RELU) {
#pragma unroll
                        for (int e = 0; e < 8; ++e)  // relu as one integer max (sign bit set -> 0; -0.0 -> +0.0)
                            v[e] = __int_as_float(max(__float_as_int(v[e]), 0));
                    }
#if CPN_FP8
                    if (dwide) *(u32x4 *) (drow + lane_off + it * step) = pack8_wide(v);
                    else
#endif
                    *(store8_t *) (drow + lane_off + it * step) = pack8(v, a.out_inv_scale);
                }
            }
        } else
#pragma unroll
        for (int f = 0; f < WM; ++f) {
            const int oy = oy0 + wave_m * WM + f;
#pragma unroll
            for (int j = 0; j < WN; ++j)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    float4 v;
                    v.x = acc[j][f][q * 4 + 0]; v.y = acc[j][f][q * 4 + 1];
                    v.z = acc[j][f][q * 4 + 2]; v.w = acc[j][f][q * 4 + 3];
                    *(float4 *) (stg + l31 * SPITCH + (j * 32 + 8 * q + 4 * lhi) * 4) = v;
                }
#pragma unroll
            for (int it = 0; it < 32 / PPI; ++it) {
                const int p = it * PPI + prow;
                const int ox = ox0 + p;
                const float4 v0 = *(const float4 *) (stg + p * SPITCH + part * 32);
                const float4 v1 = *(const float4 *) (stg + p * SPITCH + part * 32 + 16);
                if (!(ch_ok && oy < a.Hout && ox < a.Wout)) continue;
                float v[8] = CPN_V8(v0, v1);
                const size_t pix = ((size_t) n * a.Hout + oy) * a.Wout + ox;
                const size_t dpix = scat ? ((size_t) n * 2 * a.Hout + 2 * oy + spy) * (2 * a.Wout) + 2 * ox + spx : pix;
                if (a.res) {
                    size_t rpix = pix;
                    int rco = co;
                    if (a.res_up == 2) {  // pixel-shuffled phase tensor
                        rpix = ((size_t) n * a.Hr + (oy >> 1)) * a.Wr + (ox >> 1);
                        rco += ((oy & 1) * 2 + (ox & 1)) * a.res_cph;
                    } else if (a.res_up)
                        rpix = ((size_t) n * a.Hr + nearest_src(oy, a.ry, a.Hr)) * a.Wr + nearest_src(ox, a.rx, a.Wr);
#if CPN_FP8
                    if (rwide) add_res8_wide(v, *(const u32x4 *) ((const unsigned short *) a.res + rpix * a.res_stride + rco));
                    else
#endif
                    add_res8(v, *(const store8_t *) ((const elem_t *) a.res + rpix * a.res_stride + rco), a.res_scale);
                }
                if (a.act == ACT_RELU) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
                }
#if CPN_FP8
                if (dwide) *(u32x4 *) ((unsigned short *) a.dst + dpix * a.dst_stride + a.dst_coff + co) = pack8_wide(v);
                else
#endif
                *(store8_t *) ((elem_t *) a.dst + dpix * a.dst_stride + a.dst_coff + co) = pack8(v, a.out_inv_scale);
            }
        }
    } else if (a.out_mode == OUT_FUSED_HEAD) {
      if constexpr (TH % C::NWAVES == 0) {
        // fused ReadOut tail (needs BN == cout_b): stage relu(conv + bias) as bf16 [pixel][channel] in LDS (16-B slot
        // index XOR f(pixel): conflict-free MFMA B-operand reads), then D2[32][32 px] = W2[32][BN] x X[BN][32 px]
        constexpr int SPR = BN / 8;                        // 16-B slots per pixel row
        constexpr int SW = SPR >= 16 ? 16 : SPR;           // swizzle period
        constexpr int PSH = SPR >= 16 ? 0 : (SPR == 8 ? 1 : 2);
        constexpr int RPW = TH / C::NWAVES;                // pixel rows per wave in the second GEMM
        const bool hscat = a.phase == 3;                   // bilinear phases: shared bias / multipliers, scattered planes
        const int gb = hscat ? 0 : g * cout_b;
        // Per-channel bias (and e4m3 dequantisation multiplier) of the block's BN channels through an LDS table behind the staging
        // area: ONE coalesced global load per thread.  (Up to round 6 every (fragment, channel quad) of a lane loaded its four biases
        // / multipliers with single-dword global loads behind null-pointer branches and waited for them: 32 exposed L2 round trips
        // per lane, ~25 k of the 26 - 32 k cycles of this epilogue = 5 % of a 7x7 head workgroup in bf16, 11 - 19 % in e4m3 --
        // profiles/r06_phases_in_graph.txt, r06_phases_configs4_fp8.txt.)  Same expression per element: bit-identical.
        // The tail's weights W2[32][BN] (bf16, 16 KiB at BN = 256) take the same route into an LDS tile behind the table (rows XOR-
        // swizzled like the staging area): every wave needs all of W2 as the A operand of its pixel row, and eight waves reading it
        // from L2 -- five loads at a time inside the dependent MFMA chain -- were 9 k cycles of the epilogue; from LDS the chain runs
        // at MFMA latency.
        float *const ep_tab = (float *) (smem + (size_t) TH * 32 * BN * 2);
        unsigned char *const ep_w2 = (unsigned char *) (ep_tab + 2 * BN);
        constexpr int W2CH = 32 * SPR;                               // 16-byte chunks of W2
        constexpr int W2PT = (W2CH + C::THREADS - 1) / C::THREADS;   // ... per thread
        float ep_b = 0.f, ep_m = 1.f;
        u32x4 w2c[W2PT];
        if (tid < BN) {
            if (a.bias) ep_b = a.bias[gb + tid];
            if (CPN_FP8 && a.mult) ep_m = a.mult[gb + tid];
        }
#pragma unroll
        for (int i = 0; i < W2PT; ++i)
            if (tid + i * C::THREADS < W2CH) w2c[i] = *(const u32x4 *) ((const unsigned char *) a.fuse_w + (size_t) (tid + i * C::THREADS) * 16);
        __syncthreads();
        if (tid < BN) {
            ep_tab[tid] = ep_b;
            ep_tab[BN + tid] = ep_m;
        }
#pragma unroll
        for (int i = 0; i < W2PT; ++i) {
            const int c = tid + i * C::THREADS, row = c / SPR, slot = c % SPR;
            if (c < W2CH) *(u32x4 *) (ep_w2 + row * (BN * 2) + ((slot ^ (row & (SW - 1))) << 4)) = w2c[i];
        }
        __syncthreads();
        CPN_EPI_STAMP(0);
#pragma unroll
        for (int f = 0; f < WM; ++f) {
            const int p = (wave_m * WM + f) * 32 + l31;
            const int fp = (p >> PSH) & (SW - 1);
#pragma unroll
            for (int j = 0; j < WN; ++j)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int cb = wave_n * WN * 32 + j * 32 + 8 * q + 4 * lhi;  // channel within the block (n0 == 0)
                    const float4 b4 = *(const float4 *) (ep_tab + cb);
                    const float bq[4] = {b4.x, b4.y, b4.z, b4.w};
#if CPN_FP8
                    const float4 m4 = *(const float4 *) (ep_tab + BN + cb);
                    const float mq[4] = {m4.x, m4.y, m4.z, m4.w};
#endif
                    float v[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
#if CPN_FP8
                        v[e] = acc[j][f][q * 4 + e] * mq[e] + bq[e];
#else
                        v[e] = acc[j][f][q * 4 + e] + bq[e];
#endif
                        if (a.act == ACT_RELU) v[e] = fmaxf(v[e], 0.f);
                    }
                    u32x2 o;
                    o.x = pack_bf16x2(v[0], v[1]);
                    o.y = pack_bf16x2(v[2], v[3]);
                    const int slot = cb >> 3;
                    *(u32x2 *) (smem + p * (BN * 2) + ((slot ^ fp) << 4) + lhi * 8) = o;
                }
        }
        __syncthreads();
        CPN_EPI_STAMP(1);
        const unsigned char *w2 = ep_w2 + l31 * (BN * 2);
        const int w2x = l31 & (SW - 1);
#pragma unroll
        for (int r = 0; r < RPW; ++r) {
            const int row = wave * RPW + r;
            const int p = row * 32 + l31;
            const int fp = (p >> PSH) & (SW - 1);
            f32x16 acc2;
            float fb2[16];  // the tail's biases of this lane's 16 output channels: in flight behind the MFMA chain below
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                acc2[e] = 0.f;
                const int c2 = (e & 3) + 8 * (e >> 2) + 4 * lhi;
                fb2[e] = (a.fuse_b && c2 < a.fuse_cout) ? a.fuse_b[c2] : 0.f;
            }
#pragma unroll
            for (int ks = 0; ks < BN / 16; ++ks) {
                const bf16x8 xv = *(const bf16x8 *) (smem + p * (BN * 2) + (((ks * 2 + lhi) ^ fp) << 4));
                const bf16x8 wv = *(const bf16x8 *) (w2 + (((ks * 2 + lhi) ^ w2x) << 4));
                acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wv, xv, acc2, 0, 0, 0);
            }
            CPN_EPI_STAMP(2);
            const int oy = oy0 + row, ox = wrap ? (l31 < WRAP_HALF ? a.Wout - WRAP_HALF + l31 : l31 - WRAP_HALF) : ox0 + l31;
            if (oy >= a.Hout || ox >= a.Wout) continue;
            if (a.region) {  // 1: only inside the box [m, H - m) x [m, W - m); 2: only outside it
                const int m = a.region_margin;
                const bool inside = oy >= m && oy < a.Hout - m && ox >= m && ox < a.Wout - m;
                if (inside != (a.region == 1)) continue;
            }
            // (bilinear phases: phase g = (py, px) owns pixel (2 oy + py, 2 ox + px) of the [2 Hout][2 Wout] planes)
            const int ph = hscat ? 2 * a.Hout : a.Hout, pw = hscat ? 2 * a.Wout : a.Wout;
            const int py_ = hscat ? 2 * oy + (g >> 1) : oy, px_ = hscat ? 2 * ox + (g & 1) : ox;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                if (8 * (e >> 2) >= a.fuse_cout) continue;  // (wave-uniform: a 1- or 2-channel head leaves three of the four quads out)
                const int c2 = (e & 3) + 8 * (e >> 2) + 4 * lhi;
                if (c2 >= a.fuse_cout) continue;
                float x = acc2[e] + fb2[e];
                if (a.fuse_act == ACT_RELU) x = fmaxf(x, 0.f);
                else if (a.fuse_act == ACT_SIGMOID) x = 1.f / (1.f + expf(-x));
                else if (a.fuse_act == ACT_TANH_SCALED) x = tanhf(x) * a.fuse_scale;
                ((float *) a.dst)[(((size_t) n * a.fuse_cout + c2) * ph + py_) * pw + px_] = x;
            }
        }
      }
    } else {
        // fp32 NCHW planes (head outputs): lanes 0..31 are 32 consecutive x -> 128-B coalesced plane stores
        const int ox = ox0 + l31;
#pragma unroll
        for (int f = 0; f < WM; ++f) {
            const int oy = oy0 + wave_m * WM + f;
            if (oy >= a.Hout || ox >= a.Wout) continue;
#pragma unroll
            for (int j = 0; j < WN; ++j) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int co_b = n0 + wave_n * WN * 32 + j * 32 + 8 * q + 4 * lhi;  // channel within bundle
                    if (co_b >= cout_b) continue;
                    const int co = g * cout_b + co_b;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int ce = co + e;
                        if (ce >= a.cout_real) continue;
                        float x = acc[j][f][q * 4 + e] * (CPN_FP8 && a.mult ? a.mult[ce] : 1.f) + (a.bias ? a.bias[ce] : 0.f);
                        if (a.act == ACT_RELU) x = fmaxf(x, 0.f);
                        else if (a.act == ACT_SIGMOID) x = 1.f / (1.f + expf(-x));
                        else if (a.act == ACT_TANH_SCALED) x = tanhf(x) * a.act_scale;
                        ((float *) a.dst)[(((size_t) n * a.cout_real + ce) * a.Hout + oy) * a.Wout + ox] = x;
                    }
                }
            }
        }
    }
#if defined(CPN_EXP_CLOCK) && CPN_EXP_CLOCK == 1  // phases of the probed workgroup: set-up (coordinates, tables), prologue + main loop, epilogue
    __syncthreads();
    if (tid == 0 && blockIdx.x == gridDim.x / 2 && blockIdx.y == gridDim.y / 2 && blockIdx.z == 0)
        printf("PHASES out_mode %d setup %llu loop %llu epilogue %llu cycles (fused head: table %llu, stage 1 %llu, tail GEMM %llu)\n", (int) a.out_mode,
               clk_c0 - clk_entry, clk_loop_end - clk_c0, (unsigned long long) __builtin_readcyclecounter() - clk_loop_end - clk_print,
               clk_epi[0] ? clk_epi[0] - clk_loop_end - clk_print : 0ull, clk_epi[1] - clk_epi[0], clk_epi[2] - clk_epi[1]);
#endif
}

// ---------------------------------------------------------------------------------------------------------
// host side: tile selection + launch
// ---------------------------------------------------------------------------------------------------------
struct TileChoice {
    int TH, BN;
};

// narrow outputs (see MODE_N): exactly 16 columns, an even number of rows, stride-1 k x k conv whose epilogue addresses nothing
// but the output and a same-size residual
static bool narrow_ok(const ConvArgs &a) {
    return !CPN_FP8 && a.Wout == 16 && a.Hout % 2 == 0 && a.stride == 1 && !(a.KH == 1 && a.KW == 1 && a.pad == 0) &&
           a.up0 != 2 && a.res_up == 0 && a.phase != 2 && a.phase != 3 && a.region == 0 && a.KW <= 17;
}

}  // namespace CPN_NS
#if !CPN_FP8
namespace cpn {
// the bridge kernel's shape: 3x3 stride-1 pad-1 conv 64 -> 64 (one bundle, NHWC output, no resize / second source / residual
// resize) over the x2 map of a 32 | 64-channel low-resolution source
bool conv_bridge_supported(const ConvArgs &a) {
    return a.pre_src && a.pre_w && (a.pre_cin == 32 || a.pre_cin == 64) && a.pre_stride >= a.pre_cin && a.KH == 3 && a.KW == 3 &&
           a.stride == 1 && a.pad == 1 && a.bundles == 1 && a.cin_b == 64 && a.cout_b == 64 && !a.src1 && !a.up0 && !a.up1 &&
           a.res_up != 1 && a.phase == 0 && a.region == 0 && a.out_mode == OUT_BF16_NHWC && a.Hin == 2 * a.pre_H &&
           a.Win == 2 * a.pre_W && a.Hout == a.Hin && a.Wout == a.Win && a.Hout >= 16 && a.Wout >= 32;
}
}  // namespace cpn
#endif
namespace CPN_NS {
// MODE_S1F: single-source stride-1 k x k convs (2 <= k <= 5) with NHWC output and >= 128 output channels per bundle whose
// 8 x 32 x 128 tiles fill the chip twice over (two resident workgroups per CU); CPN_S1F=0 / 2 (read per call): never / wherever
// the kernel applies -- kernel A/B and tests
static bool flat_ok(const ConvArgs &a) {
    if (CPN_FP8) return false;
    const char *e = getenv("CPN_S1F");
    const int mode = e ? atoi(e) : CPN_S1F_DEFAULT;
    if (mode == 0) return false;
    const bool shape = a.stride == 1 && a.KH >= 2 && a.KH <= 5 && a.KW >= 2 && a.KW <= 5 && !a.src1 && !a.up0 && !a.up1 &&
                       a.region == 0 && !a.narrow && a.out_mode == OUT_BF16_NHWC && a.cout_b % 128 == 0 && a.Hout >= 8;
    if (!shape) return false;
    // two workgroups per CU is the point of the mode: halo ring + weight slabs of one must fit half the LDS (k <= 3 at 8 rows)
    const size_t halo_buf = (size_t) (((8 - 1 + a.KH) * 36 * 4 + 63) / 64) * 1024;
    if ((a.cin_b / 32 > 1 ? 2 : 1) * halo_buf + 2 * 2 * (size_t) 128 * REC > 160 * 1024 / 2) return false;
    const long blocks = (long) ((a.Wout + TW - 1) / TW) * ((a.Hout + 7) / 8) * a.N * (a.cout_b / 128) * a.bundles;
    return mode == 2 || blocks >= 1024;
}

// MODE_S1Q: single-source stride-1 5x5 / 7x7 convs into exactly 64 output channels (the refinement ReadOut head of the UNet models
// and the 64-channel fused heads in general) on the <16,64,2,2> tile; CPN_S1Q=0 (read per call): the two-item loop -- kernel A/B
static bool quad_ok(const ConvArgs &a) {
    if (CPN_FP8) return false;
    const char *e = getenv("CPN_S1Q");
    if (e && atoi(e) == 0) return false;
    return a.stride == 1 && a.KH == a.KW && (a.KH == 5 || a.KH == 7) && !a.src1 && !a.up0 && !a.up1 && a.region == 0 && !a.narrow &&
           a.phase == 0 && a.bundles == 1 && a.cout_b == 64 && a.Hout >= 16 &&
           (a.out_mode == OUT_BF16_NHWC || a.out_mode == OUT_FUSED_HEAD) &&
           (long) ((a.Wout + TW - 1) / TW) * ((a.Hout + 15) / 16) * a.N >= 448;
}

static int conv_mode(const ConvArgs &a) {
    if (a.pre_src) return MODE_BR;
    if (quad_ok(a)) return MODE_S1Q;
    if (a.narrow) return MODE_N;
    if (flat_ok(a)) return MODE_S1F;
    if (a.KH == 1 && a.KW == 1 && a.pad == 0) {  // incl. strided 1x1: the tile gathers only its outputs
        const char *e = getenv("CPN_PWR");  // opt-in experiment (read per call): register-weight loop, 8x256 tile only
        return (!CPN_FP8 && e && atoi(e) != 0) ? MODE_PWR : MODE_PW;
    }
    if (a.up0 == 2) return MODE_BL;
    if (a.stride == 2) return MODE_S2;
    // MODE_S1R (register-weight loop; CPN_RW read per call so that tests can toggle it), bit 0: the 8x256 tile -- opt-in: on
    // random operands it runs within 1 % of the LDS-weight loop (both sit on the same power wall), on all-zero operands 7x7
    // +8 % / 3x3 -4 %; bit 1: the 64-channel tiles (2 weight + 2 pixel fragment reads per 4 MFMAs in the LDS-weight
    // loop -> 2 pixel reads; every wave streams the 8 KiB of a 64 -> 64 tap from L2 itself)
    const char *e = getenv("CPN_RW");
    const int rw = e ? atoi(e) : CPN_RW_DEFAULT;
    if (CPN_FP8 || a.KH * a.KW < 9) return MODE_S1;
    if (a.cout_b == 64 && a.bundles == 1) return (rw & 2) ? MODE_S1R : MODE_S1;
    return (rw & 1) ? MODE_S1R : MODE_S1;  // (taken by the 8x256 tile only)
}

static size_t lds_bytes(const ConvArgs &a, int TH, int BN) {
    const int mode = conv_mode(a);
    const int S = mode == MODE_S2 ? 2 : 1;
    if (mode == MODE_BR) {  // two halo tiles of a 3x3 conv + slabs + the low-resolution input tile of the bridge stage (two chunks)
        const size_t pin = 2 * (size_t) (((TH / 2 + 3) * 20 + 15) / 16) * 1024;
        if (TH == 8)  // MODE_BRF: flat pitch-34 halo tiles, no column table: 22 + 22 + 16 + 18 = 78 KiB -> two workgroups per CU
            return 2 * (size_t) (((TH - 1 + 3) * 34 * 4 + 63) / 64) * 1024 + 2 * 2 * (size_t) BN * REC + pin;
        return 2 * (size_t) (((TH - 1 + 3) * 48 * 4 + 63) / 64) * 1024 + 2 * 2 * (size_t) BN * REC + 2 * 3 * 256 + pin;
    }
    if (mode == MODE_S1Q)  // flat pitch-40 halo tiles (two chunks in flight), four items x two buffers of slabs, no column table
        return (a.cin_b / 32 > 1 ? 2 : 1) * (size_t) (((TH - 1 + a.KH) * 40 * 4 + 63) / 64) * 1024 + 2 * 4 * (size_t) BN * REC;
    if (mode == MODE_S1F) {  // flat pitch-36 halo tiles, no column table
        const size_t halo_buf = (size_t) (((TH - 1 + a.KH) * 36 * 4 + 63) / 64) * 1024;
        return (a.cin_b / 32 > 1 ? 2 : 1) * halo_buf + 2 * 2 * (size_t) BN * REC;
    }
    const int pitch = (mode == MODE_PW || mode == MODE_PWR || mode == MODE_N) ? 32 : (mode == MODE_S2 ? 80 : 48);
    const int HH = ((mode == MODE_N ? 2 : 1) * TH - 1) * S + a.KH;
    const int nchunks = a.cin_b / 32;
    const size_t halo_buf = (size_t) ((HH * pitch * 4 + 63) / 64) * 1024;
    const int nhb = (a.KH * a.KW == 1) ? 4 : (nchunks > 1 ? 2 : 1);
    return nhb * halo_buf + 2 * 2 * (size_t) BN * REC + 2 * (size_t) (pitch / 16) * 256;  // + column offset table
}

// the epilogue's per-wave fp32 staging tiles (32 pixels x (WN*32 channels + pad)) reuse the main-loop LDS
static size_t staging_bytes(int nwaves, int WN) { return (size_t) nwaves * 32 * (WN * 32 * 4 + 16); }

constexpr size_t LDS_MAX = 160 * 1024;
constexpr int MAX_DEVICES = 64;

template <int TH, int BN, int WM, int WN, int MODE>
static int launch_mode(const ConvArgs &a, hipStream_t stream) {
    using C = Cfg<TH, BN, WM, WN>;
    size_t lds = std::max(lds_bytes(a, TH, BN), staging_bytes(C::NWAVES, WN));
    if (a.out_mode == OUT_FUSED_HEAD) lds = std::max(lds, (size_t) TH * 32 * BN * 2 + 2 * (size_t) BN * 4 + (size_t) 32 * BN * 2);  // staging + bias / multiplier table + W2
    // the dynamic-LDS limit is a per-device function attribute: remember it per device ordinal (atomic flags: plans of
    // different devices / host threads may launch the same instantiation concurrently)
    static std::atomic<bool> attr_set[MAX_DEVICES];
    auto kern = conv_igemm_kernel<TH, BN, WM, WN, MODE>;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAX_DEVICES) return (int) hipErrorInvalidDevice;
    if (!attr_set[dev].load(std::memory_order_acquire)) {
        hipError_t e = hipFuncSetAttribute((const void *) kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int) LDS_MAX);
        if (e != hipSuccess) return (int) e;
        attr_set[dev].store(true, std::memory_order_release);
    }
    const int tiles_x = (a.Wout + TW - 1) / TW, tiles_y = (a.Hout + TH - 1) / TH;
    const int ntiles = a.region == 2 ? frame_tiles(a.Hout, a.Wout, a.region_margin, TH, TW, frame_kw(a)).total : tiles_x * tiles_y;
    dim3 grid((unsigned) (ntiles * a.N), (unsigned) ((a.cout_b + BN - 1) / BN), (unsigned) a.bundles);
    if constexpr (MODE == MODE_S1F || MODE == MODE_BRF || MODE == MODE_S1Q) {  // cout blocks folded into x (see the kernel's block coordinates)
        if (MODE != MODE_S1Q && lds > LDS_MAX / 2) return (int) hipErrorInvalidValue;  // (two workgroups per CU is the point of those modes)
        grid = dim3((unsigned) (((ntiles * a.N + 7) / 8) * 8 * ((a.cout_b + BN - 1) / BN)), 1u, (unsigned) a.bundles);
    }
    hipLaunchKernelGGL(kern, grid, dim3(C::THREADS), lds, stream, a);
    return (int) hipGetLastError();
}

template <int TH, int BN, int WM, int WN>
static int launch_cfg(const ConvArgs &a, hipStream_t stream) {
    switch (conv_mode(a)) {
        case MODE_PWR:
#if !CPN_FP8
            if constexpr (TH == 8 && BN == 256 && WM == 4 && WN == 2) return launch_mode<TH, BN, WM, WN, MODE_PWR>(a, stream);
#endif
            return launch_mode<TH, BN, WM, WN, MODE_PW>(a, stream);
        case MODE_PW: return launch_mode<TH, BN, WM, WN, MODE_PW>(a, stream);
#if !CPN_FP8
        case MODE_N: return launch_mode<TH, BN, WM, WN, MODE_N>(a, stream);
#endif
        case MODE_S1: return launch_mode<TH, BN, WM, WN, MODE_S1>(a, stream);
        case MODE_S1R:
#if !CPN_FP8
            if constexpr ((TH == 8 && BN == 256 && WM == 4 && WN == 2) || (BN == 64 && WM == 2 && WN == 2))
                return launch_mode<TH, BN, WM, WN, MODE_S1R>(a, stream);
#endif
            return launch_mode<TH, BN, WM, WN, MODE_S1>(a, stream);
#if !CPN_FP8  // bf16 only: the e4m3 kernel has no registers to spare for the in-register blend (it spilled, and ran at
              // 0.16 of its plain rate); fp8 plans keep the separate resize op, which moves half the bytes of a bf16 one
        case MODE_BL: return launch_mode<TH, BN, WM, WN, MODE_BL>(a, stream);
#endif
        default: return launch_mode<TH, BN, WM, WN, MODE_S2>(a, stream);
    }
}

static TileChoice choose_tile(const ConvArgs &a) {
    int BN = a.cout_b >= 256 ? 256 : a.cout_b >= 128 ? 128 : a.cout_b >= 64 ? 64 : 32;
    int TH = 8;
    auto blocks = [&](int th, int bn) {
        return (long) ((a.Wout + TW - 1) / TW) * ((a.Hout + th - 1) / th) * a.N * ((a.cout_b + bn - 1) / bn) * a.bundles;
    };
    if (lds_bytes(a, 8, BN) > LDS_MAX || a.Hout < 8) TH = 4;
    // the big tiles have the best FLOP per L2 byte; shrink only while the grid cannot fill the 256 CUs once
    constexpr long MIN_BLOCKS = 224;
    if (TH == 8 && blocks(8, BN) < MIN_BLOCKS) TH = 4;
    if (blocks(TH, BN) < MIN_BLOCKS && BN > 128) BN = 128;
    if (blocks(TH, BN) < MIN_BLOCKS && BN > 64) BN = 64;
    while (lds_bytes(a, TH, BN) > LDS_MAX && BN > 32) BN >>= 1;
    // narrow-channel layers at high resolution (64 -> 64 @ 512^2): 16-row tiles keep 8 waves per CU busy
    // (CPN_TH64=8: kernel A/B switch, read per call -- 8-row tiles, two workgroups per CU where their LDS fits twice)
    const char *e64 = getenv("CPN_TH64");
    if (BN == 64 && TH == 8 && a.Hout >= 16 && lds_bytes(a, 16, 64) <= LDS_MAX && blocks(16, 64) >= 2 * MIN_BLOCKS &&
        !(e64 && atoi(e64) == 8))
        TH = 16;
    // CPN_PW_TILE="TH,BN" (read per call): tile of the 1x1 convs -- kernel A/B (r06 experiments #20)
    if (const char *ep = getenv("CPN_PW_TILE"); ep && a.KH == 1 && a.KW == 1 && a.pad == 0 && a.out_mode == OUT_BF16_NHWC) {
        int th = 0, bn = 0;
        if (sscanf(ep, "%d,%d", &th, &bn) == 2 && (th == 4 || th == 8) && (bn == 64 || bn == 128 || bn == 256) && bn <= a.cout_b &&
            a.Hout >= th && lds_bytes(a, th, bn) <= LDS_MAX)
            return TileChoice{th, bn};
    }
    return TileChoice{TH, BN};
}

// fused ReadOut heads: the block owns all output channels; tile rows = a multiple of the wave count
static int fused_head_rows(const ConvArgs &a, const TileChoice &c) { return (a.cout_b == 64 && c.TH == 16) ? 16 : 8; }

int launch_conv(const ConvArgs &a_in, hipStream_t stream) {
    ConvArgs a = a_in;
    if (narrow_ok(a)) {  // [N][H][16] viewed as [N][H/2][32]: same memory, full 32-pixel fragments (MODE_N)
        a.narrow = 1;
        a.Hout /= 2;
        a.Wout = 32;
    }
    if (a.cin_b % CH || a.cout_b % 32 || a.c0_used % CH) return (int) hipErrorInvalidValue;
    if (a.phase == 3 && (a.out_mode != OUT_FUSED_HEAD || a.bundles != 4 || a.stride != 1)) return (int) hipErrorInvalidValue;
    if ((a.region == 1 || a.region == 2) ? (a.out_mode != OUT_FUSED_HEAD || a.region_margin < 0) : a.region != 0)
        return (int) hipErrorInvalidValue;  // (region masks live in the fused-head epilogue)
    // (k x k: stride 1 | 2; a 1x1 conv's tile gathers only its outputs: any stride -- the 1x1 tail of a ReadOut head at stride 4 / 8)
    if (a.stride != 1 && a.stride != 2 && !(a.KH == 1 && a.KW == 1 && a.pad == 0 && a.stride >= 1)) return (int) hipErrorInvalidValue;
    if (a.up0 == 2 && (CPN_FP8 || a.stride != 1 || a.src1 || a.up1 || (a.KH == 1 && a.KW == 1)))
        return (int) hipErrorInvalidValue;  // bilinear source: single-source KxK stride-1 convs of the bf16 path only
    if ((a.stride == 1 && a.KW > 17) || (a.stride == 2 && a.KW > 18)) return (int) hipErrorInvalidValue;
#if !CPN_FP8
    if (a.pre_src) {  // bridge fusion (see ConvArgs.pre_*): checked by conv_bridge_supported
        if (!conv_bridge_supported(a)) return (int) hipErrorInvalidValue;
        // MODE_BRF (opt-in: CPN_BRF=1, read per call -- kernel A/B and tests): 8-row tiles on 4 waves, flat pitch-34 halo tiles,
        // 78 KiB = two co-resident workgroups per CU.  Bit-identical, measured NEUTRAL against the <16,64,2,2> tile (580 vs 575 us,
        // profiles/r05_kernel_experiments.txt #6): unlike the 256-channel decoder convs this kernel is not waiting on a tile's head
        // and tail -- the default stays the one-workgroup tile
        const char *e = getenv("CPN_BRF");
        if (e && atoi(e) == 1) return launch_mode<8, 64, 2, 2, MODE_BRF>(a, stream);
        return launch_mode<16, 64, 2, 2, MODE_BR>(a, stream);
    }
    if (conv_mode(a) == MODE_S1F) return launch_mode<8, 128, 4, 2, MODE_S1F>(a, stream);
    if (conv_mode(a) == MODE_S1Q) {
        if (lds_bytes(a, 16, 64) > LDS_MAX) return (int) hipErrorInvalidValue;
        return launch_mode<16, 64, 2, 2, MODE_S1Q>(a, stream);
    }
#else
    if (a.pre_src) return (int) hipErrorInvalidValue;
#endif
    TileChoice c = choose_tile(a);
    if (a.out_mode == OUT_FUSED_HEAD) {  // the block must own all output channels; TH multiple of the wave count
        if (a.bundles != (a.phase == 3 ? 4 : 1) || (a.cout_b != 256 && a.cout_b != 128 && a.cout_b != 64 && a.cout_b != 32))
            return (int) hipErrorInvalidValue;
        c.BN = a.cout_b;
        c.TH = fused_head_rows(a, c);
    }
    if (lds_bytes(a, c.TH, c.BN) > LDS_MAX) return (int) hipErrorInvalidValue;
    if (c.TH == 16) return launch_cfg<16, 64, 2, 2>(a, stream);
    if (c.TH == 8) {
        switch (c.BN) {
            case 256: return launch_cfg<8, 256, 4, 2>(a, stream);
            case 128: return launch_cfg<8, 128, 2, 2>(a, stream);
            case 64: return launch_cfg<8, 64, 2, 2>(a, stream);
            default: return launch_cfg<8, 32, 2, 1>(a, stream);
        }
    } else {
        switch (c.BN) {
            case 256: return launch_cfg<4, 256, 2, 2>(a, stream);
            case 128: return launch_cfg<4, 128, 2, 2>(a, stream);
            case 64: return launch_cfg<4, 64, 1, 2>(a, stream);
            default: return launch_cfg<4, 32, 1, 1>(a, stream);
        }
    }
}

double conv_executed_flops(const ConvArgs &a) {
    double px = (double) a.Hout * a.Wout;
    if (a.region == 2) {  // frame-only launch: only the tiles that reach outside the box run (the fused-head kernels' tiles)
        const int th = fused_head_rows(a, choose_tile(a));
        px = (double) frame_tiles(a.Hout, a.Wout, a.region_margin, th, TW, frame_kw(a)).total * th * TW;
    }
    return 2.0 * a.N * px * (double) a.bundles * a.cout_b * a.cin_b * a.KH * a.KW;
}

}  // namespace CPN_NS

#if CPN_FP8
namespace cpn {
int launch_conv_fp8(const ConvArgs &a, hipStream_t stream) { return cpn_fp8::launch_conv(a, stream); }
}  // namespace cpn
#else
// include/cpn_hip.h: shader-clock probe of the bf16 conv kernels (compiled in with -DCPN_EXP_CLOCK=2 only: libcpn_hip_clock.so)
extern "C" int cpn_debug_clock_probe(unsigned long long *out15, int reset) {
#if defined(CPN_EXP_CLOCK) && CPN_EXP_CLOCK == 2
    if (out15) {
        const int rc = cpn::check_hip(hipMemcpyFromSymbol(out15, HIP_SYMBOL(cpn::g_clock_probe), 15 * sizeof(unsigned long long)), "cpn_debug_clock_probe");
        if (rc) return rc;
    }
    if (reset) {
        const unsigned long long zero[15] = {};
        return cpn::check_hip(hipMemcpyToSymbol(HIP_SYMBOL(cpn::g_clock_probe), zero, sizeof(zero)), "cpn_debug_clock_probe (reset)");
    }
    return 0;
#else
    (void) out15; (void) reset;
    return cpn::fail(1, "cpn_debug_clock_probe: this library was built without the clock probe (-DCPN_EXP_CLOCK=2: libcpn_hip_clock.so)");
#endif
}
#endif
