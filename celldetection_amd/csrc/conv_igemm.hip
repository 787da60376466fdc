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
//     which makes the ds_read_b128 fragment reads bank-conflict free for stride-1 convs at any tap offset
//     (SQ_LDS_BANK_CONFLICT = 0 measured);
//   * the kernel is issue-bound long before it is LDS- or HBM-bound (a 32x32x16 MFMA hides only ~5-7 other
//     instructions per wave), so the halo tile has a compile-time row pitch (multiple of 16 pixels): every
//     fragment address of an item is ONE VGPR (computed with ~6 VALU per 16 MFMAs) + immediate offsets, the
//     k-half is an XOR 32 on it, and the pipeline state is tracked incrementally (no divisions in the loop);
//   * grouped convs (ResNeXt cardinality 32) run as independent dense "bundles" (grid.z) of >=32 channels with
//     block-diagonal packed weights.
#include "cpn_kernels.h"

namespace cpn {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;

constexpr int REC = 64;  // LDS bytes per 32-channel record (pixel or weight row)
constexpr int TW = 32;   // output tile width in pixels (= one MFMA column fragment)
constexpr int HREG = 5;  // halo DMA instructions per wave whose source offsets are kept in registers

__device__ __attribute__((aligned(16))) unsigned int g_zero16[4];  // source of zero padding for the halo DMA

__device__ __forceinline__ unsigned int f32_to_bf16_bits(float f) {
    unsigned int u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (u >> 16) | 0x40u;  // NaN
    u += 0x7fffu + ((u >> 16) & 1u);                                 // round to nearest even
    return u >> 16;
}
__device__ __forceinline__ float bf16_bits_to_f32(unsigned int b) { return __uint_as_float(b << 16); }

__device__ __forceinline__ void dma16(const void *gsrc, unsigned char *lds_wave_base) {
    // 64 lanes x 16 B -> LDS [lds_wave_base + lane*16]; the LDS base must be wave-uniform
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *) gsrc,
                                     (__attribute__((address_space(3))) void *) lds_wave_base, 16, 0, 0);
}

enum Mode : int { MODE_PW = 0, MODE_S1 = 1, MODE_S2 = 2 };  // pointwise stride 1 / KxK stride 1 / stride 2

template <int MODE>
struct ModeCfg {
    static constexpr int S = MODE == MODE_S2 ? 2 : 1;                              // conv stride
    static constexpr int PITCH = MODE == MODE_PW ? 32 : (MODE == MODE_S1 ? 48 : 80);  // halo row pitch (pixels)
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
    int n, iy0, ix0, Hin, Win, Hs0, Ws0, Hs1, Ws1, up0, up1, c0_stride, c1_stride, HH, HWreal;
};

// source element offsets (src0 / src1 variants; -1 = zero padding) of this lane's 16 B of halo DMA instruction q:
// lane -> (pixel, 16-B slot); the slot holds channel part (slot ^ ((pixel>>2)&3)) of the pixel record
template <int PITCH>
__device__ __forceinline__ void halo_src_offsets(const HaloGeo &G, int q, int lane, int &o0, int &o1) {
    const int idx = (q << 6) + lane;
    const int pix = idx >> 2;
    const int part = (idx & 3) ^ ((pix >> 2) & 3);
    const int hy = pix / PITCH, hx = pix - hy * PITCH;
    const int iy = G.iy0 + hy, ix = G.ix0 + hx;
    const bool valid = hy < G.HH && hx < G.HWreal && iy >= 0 && iy < G.Hin && ix >= 0 && ix < G.Win;
    o0 = -1;
    o1 = -1;
    if (valid) {
        const int y0 = G.up0 ? (iy >> 1) : iy, x0 = G.up0 ? (ix >> 1) : ix;
        const int y1 = G.up1 ? (iy >> 1) : iy, x1 = G.up1 ? (ix >> 1) : ix;
        o0 = ((G.n * G.Hs0 + y0) * G.Ws0 + x0) * G.c0_stride + part * 8;
        o1 = ((G.n * G.Hs1 + y1) * G.Ws1 + x1) * G.c1_stride + part * 8;
    }
}

template <int TH, int BN, int WM, int WN, int MODE>
__global__ __launch_bounds__((64 * (TH / WM) * (BN / (32 * WN)))) void conv_igemm_kernel(const ConvArgs a) {
    using C = Cfg<TH, BN, WM, WN>;
    constexpr int S = ModeCfg<MODE>::S;
    constexpr int PITCH = ModeCfg<MODE>::PITCH;
    constexpr bool PW = MODE == MODE_PW;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wave_m = wave / C::WAVES_N;
    const int wave_n = wave % C::WAVES_N;

    // ---- block coordinates
    const int tiles_x = (a.Wout + TW - 1) / TW;
    const int tiles_y = (a.Hout + TH - 1) / TH;
    int bid = blockIdx.x;
    const int tx = bid % tiles_x;
    bid /= tiles_x;
    const int ty = bid % tiles_y;
    const int n = bid / tiles_y;
    const int oy0 = ty * TH, ox0 = tx * TW;
    const int n0 = blockIdx.y * BN;  // first output channel (within the bundle) of this block
    const int g = blockIdx.z;        // bundle

    const int KW = PW ? 1 : a.KW;
    const int KH = PW ? 1 : a.KH;
    const int HH = (TH - 1) * S + KH;                 // halo rows
    const int hinstr = (HH * PITCH * 4 + 63) >> 6;    // 1-KiB DMA instructions per halo tile
    const int halo_buf = hinstr << 10;
    const int nchunks = a.cin_b >> 5;
    const int ntaps = KH * KW;
    const int nhb = PW ? 4 : (nchunks > 1 ? 2 : 1);   // halo ring size (chunk c lives in buffer c & (nhb-1))
    const int nhb_mask = nhb - 1;
    const int spc = (ntaps + 1) >> 1;                 // steps per chunk (KxK)
    const int nsteps = PW ? ((nchunks + 1) >> 1) : nchunks * spc;
    const int cout_b = a.cout_b;
    const int cin0 = g * a.cin_b;
    const int c0_used = a.c0_used;
    constexpr int WITEM = BN * REC;   // one item's weight slab tile
    constexpr int WBUF = 2 * WITEM;   // one step's weights
    const int ldsW_off = nhb * halo_buf;

    HaloGeo G;
    G.n = n; G.iy0 = oy0 * S - a.pad; G.ix0 = ox0 * S - a.pad; G.Hin = a.Hin; G.Win = a.Win;
    G.up0 = a.up0; G.up1 = a.up1;
    G.Hs0 = a.up0 ? (a.Hin >> 1) : a.Hin; G.Ws0 = a.up0 ? (a.Win >> 1) : a.Win;
    G.Hs1 = a.up1 ? (a.Hin >> 1) : a.Hin; G.Ws1 = a.up1 ? (a.Win >> 1) : a.Win;
    G.c0_stride = a.c0_stride; G.c1_stride = a.c1_stride; G.HH = HH; G.HWreal = (TW - 1) * S + KW;

    // halo DMA instruction q (0..hinstr) is issued by wave q % NWAVES; offsets of the first HREG kept in registers
    int h_o0[HREG], h_o1[HREG];
#pragma unroll
    for (int it = 0; it < HREG; ++it) halo_src_offsets<PITCH>(G, wave + it * C::NWAVES, lane, h_o0[it], h_o1[it]);

    const unsigned short *const src0 = (const unsigned short *) a.src0;
    const unsigned short *const src1 = (const unsigned short *) a.src1;
    const unsigned char *const zero_src = (const unsigned char *) g_zero16;

    // weight DMA: instruction q = wave + it*NWAVES of a step covers item k = q / W_INSTR_ITEM, rows qi*16..+15 of
    // the BN tile; lane -> (row, swizzled 16-B part); rows past cout_b read row 0 (their outputs are never stored)
    unsigned w_lane_off[C::W_INSTR_WAVE];
#pragma unroll
    for (int it = 0; it < C::W_INSTR_WAVE; ++it) {
        const int q = wave + it * C::NWAVES;
        const int qi = q % C::W_INSTR_ITEM;
        const int r = qi * 16 + (lane >> 2);
        const int part = (lane & 3) ^ ((r >> 2) & 3);
        w_lane_off[it] = (unsigned) (((n0 + r < cout_b) ? r : 0) * REC + part * 16);
    }
    const size_t item_bytes = (size_t) cout_b * REC;
    const unsigned char *const wbase_n0 =
            (const unsigned char *) a.weights + ((size_t) g * nchunks * ntaps * cout_b + n0) * REC;

#define HALO_DMA(CHUNK)                                                                                        \
    {                                                                                                          \
        const int c_ = (CHUNK);                                                                                \
        const int cin_ = cin0 + c_ * 32;                                                                       \
        const bool from0_ = cin_ < c0_used;                                                                    \
        const unsigned short *base_ = from0_ ? src0 + cin_ : src1 + (cin_ - c0_used);                          \
        unsigned char *dstb_ = smem + (c_ & nhb_mask) * halo_buf;                                              \
        _Pragma("unroll") for (int it = 0; it < HREG; ++it) {                                                  \
            const int q_ = wave + it * C::NWAVES;                                                              \
            if (q_ < hinstr) {                                                                                 \
                const int off_ = from0_ ? h_o0[it] : h_o1[it];                                                 \
                const void *gsrc_ = off_ >= 0 ? (const void *) (base_ + off_) : (const void *) zero_src;       \
                dma16(gsrc_, dstb_ + (q_ << 10));                                                              \
            }                                                                                                  \
        }                                                                                                      \
        for (int q_ = wave + HREG * C::NWAVES; q_ < hinstr; q_ += C::NWAVES) {                                 \
            int o0_, o1_;                                                                                      \
            halo_src_offsets<PITCH>(G, q_, lane, o0_, o1_);                                                    \
            const int off_ = from0_ ? o0_ : o1_;                                                               \
            const void *gsrc_ = off_ >= 0 ? (const void *) (base_ + off_) : (const void *) zero_src;           \
            dma16(gsrc_, dstb_ + (q_ << 10));                                                                  \
        }                                                                                                      \
    }

    // weights of a step whose first item has flattened index IDX0 (TWO: the step has a second item) -> buffer BUF
#define W_DMA(IDX0, TWO, BUF)                                                                                  \
    {                                                                                                          \
        const unsigned char *slab0_ = wbase_n0 + (size_t) (IDX0) * item_bytes;                                 \
        const unsigned char *slab1_ = slab0_ + item_bytes;                                                     \
        unsigned char *dstb_ = smem + ldsW_off + (BUF) * WBUF;                                                 \
        _Pragma("unroll") for (int it = 0; it < C::W_INSTR_WAVE; ++it) {                                       \
            const int q_ = wave + it * C::NWAVES;                                                              \
            const int k_ = q_ / C::W_INSTR_ITEM;                                                               \
            const int qi_ = q_ % C::W_INSTR_ITEM;                                                              \
            if (k_ == 0 || (TWO))                                                                              \
                dma16((k_ ? slab1_ : slab0_) + w_lane_off[it], dstb_ + k_ * WITEM + (qi_ << 10));              \
        }                                                                                                      \
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
    const unsigned w_lane = (unsigned) ((wave_n * WN * 32 + l31) * REC + ((lhi ^ ((l31 >> 2) & 3)) << 4));
    const int x_lane = l31 * S;                            // halo column of this lane's pixel for tap column 0
    const int row_wave = wave_m * WM * S;                  // halo row of fragment 0 for tap row 0
    constexpr int FRAG_STRIDE = S * PITCH * REC;           // bytes between the halo rows of consecutive fragments

    // one K item: halo buffer byte offset ABUF, tap (KY, KX), weight tile byte offset WOFF.
    // (A variant that issued the fragment reads of both items of a step up front measured 15 % slower: the
    // 24 x 8 waves ds_read_b128 burst right after the barrier delays every wave's first MFMA.)
#ifdef CPN_SETPRIO
#define CPN_PRIO(x) __builtin_amdgcn_s_setprio(x)
#else
#define CPN_PRIO(x)
#endif
#define COMPUTE_ITEM(ABUF, KY, KX, WOFF)                                                                       \
    {                                                                                                          \
        const int vx_ = x_lane + (KX);                                                                         \
        const unsigned pa0_ = (unsigned) ((ABUF) + (row_wave + (KY)) * (PITCH * REC)) + (unsigned) (vx_ * REC) + \
                              (unsigned) ((lhi ^ ((vx_ >> 2) & 3)) << 4);                                      \
        const unsigned wa0_ = (unsigned) (WOFF) + w_lane;                                                      \
        _Pragma("unroll") for (int kh = 0; kh < 2; ++kh) {                                                     \
            const unsigned pa_ = kh ? (pa0_ ^ 32u) : pa0_;                                                     \
            const unsigned wa_ = kh ? (wa0_ ^ 32u) : wa0_;                                                     \
            bf16x8 wf[WN], pf[WM];                                                                             \
            _Pragma("unroll") for (int j = 0; j < WN; ++j) wf[j] = *(const bf16x8 *) (smem + wa_ + j * 32 * REC); \
            _Pragma("unroll") for (int f = 0; f < WM; ++f) pf[f] = *(const bf16x8 *) (smem + pa_ + f * FRAG_STRIDE); \
            CPN_PRIO(1);                                                                                       \
            _Pragma("unroll") for (int j = 0; j < WN; ++j)                                                     \
                _Pragma("unroll") for (int f = 0; f < WM; ++f)                                                 \
                    acc[j][f] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[j], pf[f], acc[j][f], 0, 0, 0);     \
            CPN_PRIO(0);                                                                                       \
        }                                                                                                      \
    }

    // ---- pipeline state (wave-uniform scalars; no divisions inside the loop)
    int c = 0;                     // chunk of item 0 of the current step
    int t0 = 0, ky0 = 0, kx0 = 0;  // tap of item 0 (KxK)
    int idx0 = 0;                  // flattened (chunk*ntaps + tap) index of item 0
    // the two waves that share a SIMD (w, w + NWAVES/2) take complementary orders: one issues the next step's DMA
    // before its MFMAs, the other between its two items -> the matrix pipe is fed while the partner issues
    const bool issue_first = wave >= (C::NWAVES / 2);

    // ---- prologue: stage step 0 (and the halo tiles it needs)
    HALO_DMA(0);
    if (PW && nchunks > 1) HALO_DMA(1);
    {
        const bool two0 = PW ? (nchunks > 1) : (ntaps > 1);
        W_DMA(0, two0, 0);
    }

    for (int st = 0; st < nsteps; ++st) {
        // (1) my DMA for this step has landed; (2) everybody's has, and everybody finished reading step st-1
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        const bool two = PW ? (c + 1 < nchunks) : (t0 + 1 < ntaps);
        // item 1 of this step / item 0 of the next step
        int c1 = c, ky1 = ky0, kx1 = kx0 + 1;
        int cn = c, tn = t0 + 2, kyn = 0, kxn = 0;
        if (PW) {
            c1 = c + 1; kx1 = 0; cn = c + 2; tn = 0;
        } else {
            if (kx1 == KW) { kx1 = 0; ky1 = ky0 + 1; }
            kyn = ky1; kxn = kx1 + 1;
            if (kxn == KW) { kxn = 0; kyn = ky1 + 1; }
            if (tn >= ntaps) { tn = 0; kyn = 0; kxn = 0; cn = c + 1; }
        }
        const int idxn = idx0 + (two ? 2 : 1);
        const bool has_next = st + 1 < nsteps;
        const bool two_n = PW ? (cn + 1 < nchunks) : (tn + 1 < ntaps);
        const bool halo_ahead = !PW && t0 == 0 && c + 1 < nchunks;

#define ISSUE_NEXT()                                                                                           \
    {                                                                                                          \
        if (has_next) {                                                                                        \
            if (PW) {                                                                                          \
                HALO_DMA(cn);                                                                                  \
                if (two_n) HALO_DMA(cn + 1);                                                                   \
            }                                                                                                  \
            W_DMA(idxn, two_n, (st + 1) & 1);                                                                  \
        }                                                                                                      \
        if (halo_ahead) HALO_DMA(c + 1); /* next chunk's halo tile, a whole chunk ahead */                     \
    }

        const int wb = ldsW_off + (st & 1) * WBUF;
        if (issue_first) ISSUE_NEXT();
        COMPUTE_ITEM((c & nhb_mask) * halo_buf, ky0, kx0, wb);
        if (!issue_first) ISSUE_NEXT();
        if (two) COMPUTE_ITEM((c1 & nhb_mask) * halo_buf, ky1, kx1, wb + WITEM);
#undef ISSUE_NEXT
        c = cn; t0 = tn; ky0 = kyn; kx0 = kxn; idx0 = idxn;
    }
#undef HALO_DMA
#undef W_DMA
#undef COMPUTE_ITEM
#undef CPN_PRIO

    // ---- epilogue
    const int ox = ox0 + l31;
#pragma unroll
    for (int f = 0; f < WM; ++f) {
        const int oy = oy0 + wave_m * WM + f;
        if (oy >= a.Hout || ox >= a.Wout) continue;
        const size_t pix = ((size_t) n * a.Hout + oy) * a.Wout + ox;
        size_t rpix = pix;
        if (a.res_up) rpix = ((size_t) n * (a.Hout >> 1) + (oy >> 1)) * (a.Wout >> 1) + (ox >> 1);
#pragma unroll
        for (int j = 0; j < WN; ++j) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int co_b = n0 + wave_n * WN * 32 + j * 32 + 8 * q + 4 * lhi;  // channel within bundle
                if (co_b >= a.cout_b) continue;
                const int co = g * a.cout_b + co_b;  // global output channel
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = acc[j][f][q * 4 + e];
                if (a.bias) {
                    const float4 b = *(const float4 *) (a.bias + co);
                    v[0] += b.x; v[1] += b.y; v[2] += b.z; v[3] += b.w;
                }
                if (a.out_mode == OUT_BF16_NHWC) {
                    if (a.res) {
                        const u32x2 r = *(const u32x2 *) ((const unsigned short *) a.res + rpix * a.res_stride + co);
                        v[0] += bf16_bits_to_f32(r.x & 0xffffu);
                        v[1] += bf16_bits_to_f32(r.x >> 16);
                        v[2] += bf16_bits_to_f32(r.y & 0xffffu);
                        v[3] += bf16_bits_to_f32(r.y >> 16);
                    }
                    if (a.act == ACT_RELU) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
                    }
                    u32x2 o;
                    o.x = f32_to_bf16_bits(v[0]) | (f32_to_bf16_bits(v[1]) << 16);
                    o.y = f32_to_bf16_bits(v[2]) | (f32_to_bf16_bits(v[3]) << 16);
                    *(u32x2 *) ((unsigned short *) a.dst + pix * a.dst_stride + a.dst_coff + co) = o;
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int ce = co + e;
                        if (ce >= a.cout_real) continue;
                        float x = v[e];
                        if (a.act == ACT_RELU) x = fmaxf(x, 0.f);
                        else if (a.act == ACT_SIGMOID) x = 1.f / (1.f + expf(-x));
                        else if (a.act == ACT_TANH_SCALED) x = tanhf(x) * a.act_scale;
                        ((float *) a.dst)[(((size_t) n * a.cout_real + ce) * a.Hout + oy) * a.Wout + ox] = x;
                    }
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// host side: tile selection + launch
// ---------------------------------------------------------------------------------------------------------
struct TileChoice {
    int TH, BN;
};

static int conv_mode(const ConvArgs &a) {
    if (a.stride == 2) return MODE_S2;
    return (a.KH == 1 && a.KW == 1) ? MODE_PW : MODE_S1;
}

static size_t lds_bytes(const ConvArgs &a, int TH, int BN) {
    const int mode = conv_mode(a);
    const int S = mode == MODE_S2 ? 2 : 1;
    const int pitch = mode == MODE_PW ? 32 : (mode == MODE_S1 ? 48 : 80);
    const int HH = (TH - 1) * S + a.KH;
    const int nchunks = a.cin_b / 32;
    const size_t halo_buf = (size_t) ((HH * pitch * 4 + 63) / 64) * 1024;
    const int nhb = mode == MODE_PW ? 4 : (nchunks > 1 ? 2 : 1);
    return nhb * halo_buf + 2 * 2 * (size_t) BN * REC;
}

constexpr size_t LDS_MAX = 160 * 1024;

template <int TH, int BN, int WM, int WN, int MODE>
static int launch_mode(const ConvArgs &a, hipStream_t stream) {
    using C = Cfg<TH, BN, WM, WN>;
    const size_t lds = lds_bytes(a, TH, BN);
    static bool attr_set = false;
    auto kern = conv_igemm_kernel<TH, BN, WM, WN, MODE>;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void *) kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int) LDS_MAX);
        if (e != hipSuccess) return (int) e;
        attr_set = true;
    }
    const int tiles_x = (a.Wout + TW - 1) / TW, tiles_y = (a.Hout + TH - 1) / TH;
    dim3 grid((unsigned) (tiles_x * tiles_y * a.N), (unsigned) ((a.cout_b + BN - 1) / BN), (unsigned) a.bundles);
    hipLaunchKernelGGL(kern, grid, dim3(C::THREADS), lds, stream, a);
    return (int) hipGetLastError();
}

template <int TH, int BN, int WM, int WN>
static int launch_cfg(const ConvArgs &a, hipStream_t stream) {
    switch (conv_mode(a)) {
        case MODE_PW: return launch_mode<TH, BN, WM, WN, MODE_PW>(a, stream);
        case MODE_S1: return launch_mode<TH, BN, WM, WN, MODE_S1>(a, stream);
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
    if (BN == 64 && TH == 8 && a.Hout >= 16 && lds_bytes(a, 16, 64) <= LDS_MAX && blocks(16, 64) >= 2 * MIN_BLOCKS) TH = 16;
    return TileChoice{TH, BN};
}

int launch_conv(const ConvArgs &a, hipStream_t stream) {
    if (a.cin_b % 32 || a.cout_b % 32 || a.c0_used % 32) return (int) hipErrorInvalidValue;
    if (a.stride != 1 && a.stride != 2) return (int) hipErrorInvalidValue;
    if ((a.stride == 1 && a.KW > 17) || (a.stride == 2 && a.KW > 18)) return (int) hipErrorInvalidValue;
    const TileChoice c = choose_tile(a);
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
    return 2.0 * a.N * a.Hout * a.Wout * (double) a.bundles * a.cout_b * a.cin_b * a.KH * a.KW;
}

}  // namespace cpn
