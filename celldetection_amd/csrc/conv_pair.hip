// Fused head of a ResNeXt bottleneck block for gfx950 (MI355X / CDNA4):
//     conv1 (1x1, Cin -> Cmid) + BN + ReLU  ->  conv2 (3x3, groups = 32, stride 1, pad 1) + BN + ReLU
// in ONE kernel: conv1's output never travels to HBM.  Replaces, on the CPN inference path, the first two thirds of
// torchvision's Bottleneck.forward as instantiated by celldetection/models/resnet.py:88-116,119-193 (ResNeXt 32x4d / 32x8d
// stages), i.e. two cpn::conv_igemm launches + one [N, H, W, Cmid] tensor written and re-read per block.
//
// Why this shape (MI355X-first, not a translation of anything):
//   * a grouped conv needs, for an output channel, only the Cmid / 32 input channels of its own group -- so a workgroup that
//     owns a SLAB of SL conv1 output channels (= whole groups) can run conv2 for those groups without any other slab;
//     what it needs from conv1 is the slab on its output tile + a one-pixel halo.  Images exactly 16 | 32 | 64 pixels wide are
//     cut into FULL-WIDTH ROW STRIPS: the left / right halo is the conv's zero padding, only the two halo ROWS are recomputed
//     (10 rows for 8: 1.25 x conv1's MACs); every other width takes GENERIC tiles of 16 x 32 output pixels with a real halo
//     on all four sides (18 x 34 of 16 x 32: 1.2 x), 128-channel slabs;
//   * stage 1 = conv1 as an implicit GEMM  D1[SL][halo px] += W1[SL][Cin] * X[Cin][halo px]  on v_mfma_f32_32x32x16_bf16:
//     halo pixels flattened (a pixel fragment is ANY 32 consecutive pixels of the strip -- rows are contiguous in NHWC), wave
//     tile 64 channels x 160 pixels (10 MFMAs per 7 fragment reads), operands staged by LDS-DMA through a 3-deep ring of
//     32-channel chunks (two chunks of prefetch distance, one workgroup barrier per chunk);
//   * the activated slab is rounded to bf16 into LDS as [32-channel chunk][halo pixel][32] records (same XOR-swizzled 64-byte
//     records as the conv_igemm halo tiles; pixels outside the image are ZERO = conv2's padding) and OVERWRITES the staging
//     ring: 81920 values x 2 B = the whole 160 KiB;
//   * stage 2 = conv2 from that tile: a wave owns 64 output channels (two 32-channel bundles, or one 64-channel bundle) on a
//     quarter of the output fragments, reads the pixel operand from LDS (tap offsets are address immediates; the columns the
//     3x3 window reaches beyond the image edge are masked to zero in registers) and the block-diagonal packed weights of
//     ITS bundles straight from L2 into registers -- no barrier, no weight staging, nothing shared between waves;
//   * epilogue: bias + ReLU, bf16, per-wave LDS transpose, 16-byte NHWC stores of 128-byte channel runs.
#include <atomic>
#include <cstdlib>

#include "cpn_kernels.h"
#include "lds_dma.h"

namespace cpn {
namespace {

typedef bf16x8 frag_t;

// WLOG > 0: full-width row strips of an image exactly 2^WLOG pixels wide.  WLOG == 0: GENERIC tiles of 16 x 32 output pixels
// with a real one-pixel halo on all four sides (18 x 34 = 612 halo pixels, flattened with pitch 34) for any image size.
template <int WLOG, int SL>
struct PairCfg {
    static constexpr bool GEN = WLOG == 0;
    static constexpr int W = GEN ? 34 : (1 << WLOG);         // row pitch of the halo tile (strips: = image width)
    static constexpr int WC = SL / 64;                       // stage 1: waves along the channels (64 each)
    static constexpr int WP = 8 / WC;                        //          waves along the pixels (5 fragments each)
    static constexpr int NF1 = 5 * WP;                       // pixel fragments of the halo tile
    static constexpr int NPXP = NF1 * 32;                    // pixels staged (incl. padding up to whole fragments)
    static constexpr int ROWS = (GEN || W == 16) ? 18 : NPXP / W;  // halo rows
    static constexpr int NPX = ROWS * W;                     // real halo pixels
    static constexpr int THO = ROWS - 2;                     // output rows per tile
    static constexpr int TWO = GEN ? 32 : W;                 // output columns per tile
    static constexpr int OUTF = THO * TWO / 32;              // output fragments per tile
    static constexpr int PQ = OUTF / 4;                      // stage 2: waves along the pixels (4 fragments each)
    static constexpr int XBUF = NPXP * 64;                   // one 32-channel chunk of the tile
    static constexpr int WBUF = SL * 64;                     // one 32-channel chunk of the slab's weights
    static constexpr int XI = NPXP / 16;                     // 1-KiB DMA instructions per activation chunk
    static constexpr int XIW = (XI + 7) / 8;                 //   per wave (the last round may be partial)
    static constexpr int WIW = SL / 16 / 8;                  // weight DMA instructions per wave and chunk
    static constexpr int LDS = (SL / 32) * XBUF;             // the bf16 slab tile
    static_assert(SL == 128 || SL == 256, "slab");
    static_assert(!GEN || SL == 128, "generic tiles: 128-channel slabs");
    static_assert(NPX <= NPXP && OUTF % 4 == 0 && (SL / 64) * PQ == 8, "eight stage-2 jobs");
    static_assert(LDS == 160 * 1024, "LDS budget");
};

template <int N>
__device__ __forceinline__ void wait_vm() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// stage-1 fragment set: 2 weight fragments (64 channels) + 5 pixel fragments (160 pixels) of one k-half
struct Set1 {
    frag_t w[2], p[5];
};
__device__ __forceinline__ void load_set1(Set1 &s, unsigned waddr, unsigned paddr) {
    ds_read16<0>(s.w[0], waddr);
    ds_read16<2048>(s.w[1], waddr);
    ds_read16<0>(s.p[0], paddr);
    ds_read16<2048>(s.p[1], paddr);
    ds_read16<4096>(s.p[2], paddr);
    ds_read16<6144>(s.p[3], paddr);
    ds_read16<8192>(s.p[4], paddr);
}
template <int N>
__device__ __forceinline__ void wait_set1(Set1 &s) {
    asm volatile("s_waitcnt lgkmcnt(%7)"
                 : "+v"(s.w[0]), "+v"(s.w[1]), "+v"(s.p[0]), "+v"(s.p[1]), "+v"(s.p[2]), "+v"(s.p[3]), "+v"(s.p[4])
                 : "n"(N));
}
template <int N>
__device__ __forceinline__ void wait_p4(frag_t (&p)[4]) {
    asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(p[0]), "+v"(p[1]), "+v"(p[2]), "+v"(p[3]) : "n"(N));
}

// CPS = 32-channel chunks per stage-1 pipeline step: 1 = ring of three chunks (two chunks of prefetch distance), 2 = ring of
// two steps x two chunks like conv_igemm's 1x1 loop (half the barriers; needs an even chunk count and 4 ring slots in LDS)
// STRIDE = conv2's stride.  2 (generic tiles only): the 18 x 34 halo tile of conv1's output is the input of an 8 x 16 OUTPUT
// tile (output (oy, ox) reads halo pixels (2 oy + ky, 2 ox + kx)); stage 1 is unchanged, stage 2 has one fragment per wave
template <int WLOG, int SL, int CB, int CPS, int STRIDE = 1>
__global__ __launch_bounds__(512) void conv_pair_kernel(const PairArgs a) {
    using C = PairCfg<WLOG, SL>;
    static_assert(STRIDE == 1 || (STRIDE == 2 && C::GEN), "stride-2 conv2: generic tiles");
    constexpr int NR = CPS == 2 ? 4 : 3;  // ring slots
    static_assert(NR * (C::XBUF + C::WBUF) <= C::LDS, "staging ring exceeds the LDS");
    constexpr int W = C::W;
    constexpr int XB = C::XBUF, WB = C::WBUF;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    typedef __attribute__((address_space(3))) unsigned char lds_u8;
    typedef __attribute__((address_space(3))) u32x2 lds_u32x2;
    typedef __attribute__((address_space(3))) u32x4 lds_u32x4;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, lhi = lane >> 5;
    constexpr bool GEN = C::GEN;
    const int Wimg = GEN ? a.W : W;   // image width (strips: a template constant)
    // (stride 2: tiles are counted on the OUTPUT, 8 x 16 pixels each = 16 x 32 input pixels)
    const int Ho = STRIDE == 2 ? (a.H - 1) / 2 + 1 : a.H, Wo = STRIDE == 2 ? (a.W - 1) / 2 + 1 : a.W;
    const int tiles_y = (Ho + C::THO / STRIDE - 1) / (C::THO / STRIDE);
    const int tiles_x = GEN ? (Wo + C::TWO / STRIDE - 1) / (C::TWO / STRIDE) : 1;
    const int tx = (int) (blockIdx.x % (unsigned) tiles_x), ty = (int) ((blockIdx.x / (unsigned) tiles_x) % (unsigned) tiles_y);
    const int n = (int) (blockIdx.x / (unsigned) (tiles_x * tiles_y));
    const int oy0 = ty * C::THO;      // first output row of the tile; halo row r is image row oy0 - 1 + r
    const int ox0 = tx * C::TWO;      // generic tiles: halo column c is image column ox0 - 1 + c
    const int n0 = blockIdx.y * SL;   // first conv1 output channel of the slab
    // halo pixel p of the flattened tile -> image coordinates; false outside the image (= conv2's zero padding)
    auto halo_px = [&](int p, int &iy, int &ix) -> bool {
        if constexpr (GEN) {
            const int r = p / 34;
            iy = oy0 - 1 + r;
            ix = ox0 - 1 + (p - r * 34);
            return p < C::NPX && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
        } else {
            iy = oy0 - 1 + (p >> WLOG);
            ix = p & (W - 1);
            return p < C::NPX && iy >= 0 && iy < a.H;
        }
    };
    const int nchunks = a.cin >> 5;
    const unsigned lds0 = (unsigned) (size_t) (lds_u8 *) smem;

    // ------------------------------------------------------------------------------------------------------------
    // stage 1: conv1 on the halo strip
    // ------------------------------------------------------------------------------------------------------------
    // activation DMA: instruction q covers halo pixels q*16 .. +15 (lane -> pixel q*16 + lane/4, 16-byte slot lane%4 which
    // holds channel part slot ^ ((pixel >> 2) & 3)); per-lane offsets are loop constants, the chunk travels in the scalar
    // offset.  Pixels of rows outside the image (and the padding behind the last real pixel) read zeros.
    const rsrc_t rsx = make_rsrc(a.src, (unsigned) ((size_t) a.N * a.H * Wimg * a.c_stride * 2));
    unsigned x_voff[C::XIW];
#pragma unroll
    for (int it = 0; it < C::XIW; ++it) {
        const int q = wave + it * 8;
        const int p = q * 16 + (lane >> 2);
        const int part = (lane & 3) ^ ((p >> 2) & 3);
        int iy, ix;
        const bool ok = halo_px(p, iy, ix);
        x_voff[it] = ok ? (unsigned) ((((n * a.H + iy) * Wimg + ix) * a.c_stride + part * 8) * 2) : OOB_LANE;
    }
    const bool x_extra = wave + (C::XIW - 1) * 8 < C::XI;  // (wave-uniform) this wave issues the last, partial round
    // weight DMA: instruction q covers rows q*16 .. +15 of the slab's chunk slab [SL][32] (same swizzle)
    const int nitems1 = nchunks + (nchunks & 1);  // (the packer pads an odd item count with an all-zero slab)
    const rsrc_t rsw = make_rsrc(a.w1, (unsigned) ((size_t) nitems1 * a.cmid * 64));
    const unsigned w_dma_lane = (unsigned) ((lane >> 2) * 64 + (((lane & 3) ^ ((lane >> 4) & 3)) << 4));
    unsigned char *const xring = smem;
    unsigned char *const wring = smem + NR * XB;
#define PAIR_DMA(CHUNK, BUF)                                                                                   \
    {                                                                                                          \
        const unsigned xs_ = (unsigned) (CHUNK) * 64u;                                                         \
        _Pragma("unroll") for (int it = 0; it < C::XIW; ++it)                                                  \
            if (it + 1 < C::XIW || x_extra) bdma16(rsx, x_voff[it], xs_, xring + (BUF) * XB + ((wave + it * 8) << 10)); \
        const unsigned ws_ = (unsigned) (((CHUNK) * a.cmid + n0) * 64);                                        \
        _Pragma("unroll") for (int it = 0; it < C::WIW; ++it)                                                  \
            bdma16(rsw, (unsigned) ((wave + it * 8) << 10) + w_dma_lane, ws_, wring + (BUF) * WB + ((wave + it * 8) << 10)); \
    }

    const int wc = wave / C::WP, wp = wave % C::WP;  // this wave: channels wc*64 .. +63, pixel fragments wp*5 .. +4
    const unsigned swz = (unsigned) ((lhi ^ ((l31 >> 2) & 3)) << 4);
    const unsigned p_lane = lds0 + (unsigned) ((wp * 5 * 32 + l31) * 64) + swz;
    const unsigned w_lane = lds0 + (unsigned) (NR * XB + (wc * 64 + l31) * 64) + swz;

    f32x16 acc[2][5];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int f = 0; f < 5; ++f)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][f][r] = 0.f;

#ifdef PAIR_EXP_NOMFMA1
#define PAIR_MMA(ACC, A_, B_) asm volatile("" ::"v"(A_), "v"(B_))
#else
#define PAIR_MMA(ACC, A_, B_) ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A_, B_, ACC, 0, 0, 0)
#endif
#define MMA_SET1(S, PENDING)                                                                                   \
    {                                                                                                          \
        wait_set1<PENDING>(S);                                                                                 \
        _Pragma("unroll") for (int f = 0; f < 5; ++f) {                                                        \
            PAIR_MMA(acc[0][f], S.w[0], S.p[f]);                                                               \
            PAIR_MMA(acc[1][f], S.w[1], S.p[f]);                                                               \
        }                                                                                                      \
    }

    // tuning builds (tools/build_variant.sh; results are wrong by construction): -DPAIR_EXP_NOS1 skips stage 1's DMA + MFMA
    // loop, -DPAIR_EXP_NOS2 stage 2's groups, -DPAIR_EXP_NOEPI the output stores
#ifndef PAIR_EXP_NOS1
  if constexpr (CPS == 1) {
    // prologue: three chunks in flight, the first one landed
    PAIR_DMA(0, 0);
    if (nchunks > 1) PAIR_DMA(1, 1);
    if (nchunks > 2) PAIR_DMA(2, 2);
    wait_vm<0>();
    __builtin_amdgcn_s_barrier();
    Set1 A, B;
    load_set1(A, w_lane, p_lane);
    int buf = 0;
    // every chunk that is followed by another one: the loop body is branch-free with respect to the accumulators and the
    // fragment sets (conditional MFMA groups make hipcc rename / spill accumulators); the last chunk is peeled
    for (int c = 0; c + 1 < nchunks; ++c) {
        load_set1(B, (w_lane + (unsigned) (buf * WB)) ^ 32u, (p_lane + (unsigned) (buf * XB)) ^ 32u);  // k-half 1 of chunk c
        MMA_SET1(A, 7);                                                                                // k-half 0
        // step boundary: my DMA of chunk c+1 landed (chunk c+2's may still fly), all my reads of chunk c returned ...
        if (c + 2 < nchunks) {
            if (x_extra) wait_vm<C::XIW + C::WIW>();
            else wait_vm<C::XIW - 1 + C::WIW>();
        } else {
            wait_vm<0>();
        }
        wait_set1<0>(B);
#ifndef PAIR_EXP_NOBAR
        __builtin_amdgcn_s_barrier();  // ... and so have everybody else's: ring slot `buf` is free
#endif
#ifndef PAIR_EXP_NODMA
        if (c + 3 < nchunks) PAIR_DMA(c + 3, buf);
#endif
        buf = buf == 2 ? 0 : buf + 1;
        load_set1(A, w_lane + (unsigned) (buf * WB), p_lane + (unsigned) (buf * XB));  // k-half 0 of chunk c+1
        MMA_SET1(B, 7);
    }
    load_set1(B, (w_lane + (unsigned) (buf * WB)) ^ 32u, (p_lane + (unsigned) (buf * XB)) ^ 32u);
    MMA_SET1(A, 7);
    MMA_SET1(B, 0);
  } else {
    // two chunks per step: step s = chunks 2s, 2s+1 in ring slots 2(s & 1), 2(s & 1) + 1; the DMA of step s+2 is issued at the
    // boundary that closes step s and has the whole of step s+1 (40 MFMAs per wave) to land
    const int nsteps = nchunks >> 1;
    PAIR_DMA(0, 0);
    PAIR_DMA(1, 1);
    wait_vm<0>();
    __builtin_amdgcn_s_barrier();
    if (nsteps > 1) {
        PAIR_DMA(2, 2);
        PAIR_DMA(3, 3);
    }
    Set1 A, B;
    load_set1(A, w_lane, p_lane);
    unsigned xo = 0, wo = 0;  // byte offsets of the current step's first ring slot
    for (int st = 0; st + 1 < nsteps; ++st) {
        load_set1(B, (w_lane + wo) ^ 32u, (p_lane + xo) ^ 32u);             // chunk 0, k-half 1
        MMA_SET1(A, 7);                                                      // chunk 0, k-half 0
        load_set1(A, w_lane + wo + WB, p_lane + xo + XB);                    // chunk 1, k-half 0
        MMA_SET1(B, 7);
        load_set1(B, (w_lane + wo + WB) ^ 32u, (p_lane + xo + XB) ^ 32u);    // chunk 1, k-half 1
        MMA_SET1(A, 7);
        // step boundary: my DMA of step st+1 landed, all my reads of step st returned ...
        wait_vm<0>();
        wait_set1<0>(B);
#ifndef PAIR_EXP_NOBAR
        __builtin_amdgcn_s_barrier();  // ... and so have everybody else's: the two slots of step st are free
#endif
#ifndef PAIR_EXP_NODMA
        if (st + 2 < nsteps) {
            const int sl_ = (st & 1) * 2;
            PAIR_DMA(2 * st + 4, sl_);
            PAIR_DMA(2 * st + 5, sl_ + 1);
        }
#endif
        xo = xo ? 0u : (unsigned) (2 * XB);
        wo = wo ? 0u : (unsigned) (2 * WB);
        load_set1(A, w_lane + wo, p_lane + xo);                              // first group of step st+1
        MMA_SET1(B, 7);                                                      // last group of step st (operands in registers)
    }
    load_set1(B, (w_lane + wo) ^ 32u, (p_lane + xo) ^ 32u);
    MMA_SET1(A, 7);
    load_set1(A, w_lane + wo + WB, p_lane + xo + XB);
    MMA_SET1(B, 7);
    load_set1(B, (w_lane + wo + WB) ^ 32u, (p_lane + xo + XB) ^ 32u);
    MMA_SET1(A, 7);
    MMA_SET1(B, 0);
  }
#endif
#undef MMA_SET1
#undef PAIR_MMA
#undef PAIR_DMA

    // stage 2's job of this wave (see below): its first weight fragments are requested from L2 NOW, ahead of the slab-tile
    // write and its two barriers
    const int jp = wave / C::PQ, pq = wave % C::PQ;
    // weights: packed [bundle][item = chunk-in-bundle * 9 + tap (+ 1 zero item if odd)][cout_b][32] bf16, rows are the MFMA
    // fragment rows (not swizzled): lane -> row l31 of row block j, 16-byte part 2 * khalf + lhi
    constexpr int COB = 32 * CB;                  // channels per bundle
    constexpr int NI2 = 9 * CB + ((9 * CB) & 1);  // items per bundle incl. padding
    const unsigned char *wrow[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int bundle = CB == 1 ? (n0 >> 5) + jp * 2 + j : (n0 >> 6) + jp;
        const int row = CB == 1 ? l31 : j * 32 + l31;
        wrow[j] = (const unsigned char *) a.w2 + ((size_t) bundle * NI2 * COB + row) * 64 + lhi * 16;
    }
    frag_t Wr[3][2];
#define PAIR_G_S(G) ((G) / 18)
#define PAIR_G_TAP(G) (((G) % 18) >> 1)
#define PAIR_G_KH(G) ((G) & 1)
#define PAIR_LOADW(G)                                                                                          \
    {                                                                                                          \
        constexpr int s_ = PAIR_G_S(G), t_ = PAIR_G_TAP(G), kh_ = PAIR_G_KH(G);                                \
        if constexpr (CB == 1) {                                                                               \
            Wr[(G) % 3][0] = *(const frag_t *) (wrow[s_] + t_ * (COB * 64) + kh_ * 32);                        \
        } else {                                                                                               \
            Wr[(G) % 3][0] = *(const frag_t *) (wrow[0] + (s_ * 9 + t_) * (COB * 64) + kh_ * 32);              \
            Wr[(G) % 3][1] = *(const frag_t *) (wrow[1] + (s_ * 9 + t_) * (COB * 64) + kh_ * 32);              \
        }                                                                                                      \
    }
#ifndef PAIR_EXP_NOS2
    PAIR_LOADW(0);
    PAIR_LOADW(1);
#endif

    // ---- the slab tile: relu(conv1 + bias) as bf16 records [chunk][halo pixel][32], zero outside the image
    __syncthreads();  // every wave is done with the staging ring
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int chunk = wc * 2 + j;
        float4 b4[4];
#pragma unroll
        for (int q = 0; q < 4; ++q)
            b4[q] = a.b1 ? *(const float4 *) (a.b1 + n0 + chunk * 32 + 8 * q + 4 * lhi) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int f = 0; f < 5; ++f) {
            const int p = (wp * 5 + f) * 32 + l31;
            int iy_, ix_;
            const bool ok = halo_px(p, iy_, ix_);
            lds_u8 *rec = (lds_u8 *) smem + chunk * XB + p * 64 + lhi * 8;
            const int sw = (p >> 2) & 3;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float v0 = acc[j][f][q * 4 + 0] + b4[q].x, v1 = acc[j][f][q * 4 + 1] + b4[q].y;
                const float v2 = acc[j][f][q * 4 + 2] + b4[q].z, v3 = acc[j][f][q * 4 + 3] + b4[q].w;
                u32x2 o;
                o.x = ok ? pack_bf16x2(fmaxf(v0, 0.f), fmaxf(v1, 0.f)) : 0u;
                o.y = ok ? pack_bf16x2(fmaxf(v2, 0.f), fmaxf(v3, 0.f)) : 0u;
                *(lds_u32x2 *) (rec + ((q ^ sw) << 4)) = o;
            }
        }
    }
    __syncthreads();

    // ------------------------------------------------------------------------------------------------------------
    // stage 2: grouped 3x3 conv from the slab tile.  Job of this wave: 64 output channels n0 + jp*64 .. +63 (row blocks
    // j = 0, 1 of 32) on output fragments pq*4 .. +3.  CB = 1: row block j is the 32-channel bundle 2 jp + j and reads slab
    // chunk 2 jp + j; CB = 2: both row blocks belong to the 64-channel bundle jp and read chunks 2 jp, 2 jp + 1.
    // ------------------------------------------------------------------------------------------------------------
    if constexpr (STRIDE == 2) {
        // ---- stride-2 conv2: this wave = 64 output channels x ONE output fragment (output rows 2 pq, 2 pq + 1 of the 8 x 16
        // tile, 16 columns each): lane -> (row 2 pq + (l31 >> 4), column l31 & 15), operand = halo pixel (2 row + ky) * 34 +
        // 2 column + kx.  36 groups of 1 | 2 MFMAs -- a tenth of stage 1's work
        f32x16 a2[2];
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) a2[j][r] = 0.f;
        const unsigned u2 = (unsigned) ((4 * pq + 2 * (l31 >> 4)) * 34 + 2 * (l31 & 15));
        const unsigned base2 = lds0 + (unsigned) (jp * 2 * XB);
        frag_t P2[2];
        auto p2_addr = [&](int g) -> unsigned {
            const int s_ = g / 18, t_ = (g % 18) >> 1, kh_ = g & 1;
            const unsigned tt = u2 + (unsigned) ((t_ / 3) * 34 + t_ % 3);
            return base2 + (unsigned) (s_ * XB) + tt * 64u + ((((unsigned) lhi ^ ((tt >> 2) & 3u)) << 4) ^ (unsigned) (kh_ * 32));
        };
#ifndef PAIR_EXP_NOS2
        ds_read16<0>(P2[0], p2_addr(0));
#pragma unroll
        for (int g = 0; g < 36; ++g) {
            const int s_ = g / 18, t_ = (g % 18) >> 1, kh_ = g & 1;
            if (g + 2 < 36) {
                const int s2_ = (g + 2) / 18, t2_ = ((g + 2) % 18) >> 1, k2_ = (g + 2) & 1;
                if constexpr (CB == 1) {
                    Wr[(g + 2) % 3][0] = *(const frag_t *) (wrow[s2_] + t2_ * (COB * 64) + k2_ * 32);
                } else {
                    Wr[(g + 2) % 3][0] = *(const frag_t *) (wrow[0] + (s2_ * 9 + t2_) * (COB * 64) + k2_ * 32);
                    Wr[(g + 2) % 3][1] = *(const frag_t *) (wrow[1] + (s2_ * 9 + t2_) * (COB * 64) + k2_ * 32);
                }
            }
            if (g + 1 < 36) {
                ds_read16<0>(P2[(g + 1) & 1], p2_addr(g + 1));
                asm volatile("s_waitcnt lgkmcnt(1)" : "+v"(P2[g & 1]));
            } else {
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(P2[g & 1]));
            }
            (void) t_; (void) kh_;
            if constexpr (CB == 1) {
                a2[s_] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Wr[g % 3][0], P2[g & 1], a2[s_], 0, 0, 0);
            } else {
                a2[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Wr[g % 3][0], P2[g & 1], a2[0], 0, 0, 0);
                a2[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Wr[g % 3][1], P2[g & 1], a2[1], 0, 0, 0);
            }
        }
#endif
        __syncthreads();  // every wave is done reading the slab tile
        constexpr int SP2 = 144;
        lds_u8 *const stg2 = (lds_u8 *) smem + wave * (32 * SP2);
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 b = a.b2 ? *(const float4 *) (a.b2 + n0 + jp * 64 + j * 32 + 8 * q + 4 * lhi) : make_float4(0.f, 0.f, 0.f, 0.f);
                u32x2 o;
                o.x = pack_bf16x2(fmaxf(a2[j][q * 4 + 0] + b.x, 0.f), fmaxf(a2[j][q * 4 + 1] + b.y, 0.f));
                o.y = pack_bf16x2(fmaxf(a2[j][q * 4 + 2] + b.z, 0.f), fmaxf(a2[j][q * 4 + 3] + b.w, 0.f));
                *(lds_u32x2 *) (stg2 + l31 * SP2 + j * 64 + q * 16 + lhi * 8) = o;
            }
        const int px_l2 = lane >> 3, part_l2 = lane & 7;
        unsigned char *const dst2 = (unsigned char *) a.dst + ((size_t) n * Ho * Wo * a.dst_stride + n0 + jp * 64 + part_l2 * 8) * 2;
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int px = it * 8 + px_l2;
            const u32x4 v = *(const lds_u32x4 *) (stg2 + px * SP2 + part_l2 * 16);
            const int oy = ty * 8 + 2 * pq + (px >> 4), ox = tx * 16 + (px & 15);
#ifdef PAIR_EXP_NOEPI
            if (a.N < 0)
#endif
            if (oy < Ho && ox < Wo) *(u32x4 *) (dst2 + ((size_t) oy * Wo + ox) * a.dst_stride * 2) = v;
        }
        return;
    }
    f32x16 acc2[2][4];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int f = 0; f < 4; ++f)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc2[j][f][r] = 0.f;

    // pixel operand of output pixel o = (pq*4 + f)*32 + l31 at tap (ky, kx): slab pixel o + ky*W + kx - 1
    unsigned a_kx[3];
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
        const int xo = l31 + kx - 1;  // (-1 for lane 0 at kx = 0: the address is garbage and the value masked)
        a_kx[kx] = lds0 + (unsigned) (jp * 2 * XB + pq * 4 * 2048) + (unsigned) (xo * 64) + (unsigned) ((lhi ^ ((xo >> 2) & 3)) << 4);
    }
    // group g = (s, tap, khalf), s = row block (CB = 1) | chunk of the bundle (CB = 2): 4 pixel fragments, 4 | 8 MFMAs
    constexpr int NG = 36;
    frag_t P[2][4];
    // generic tiles: output fragment f' = pq*4 + f is output ROW f' (32 columns), lane = column; its operand at tap (ky, kx)
    // is halo pixel t = (f' + ky) * 34 + lane + kx -- the pitch is no multiple of 4, so the record's swizzle term (t >> 2) & 3 is
    // recomputed per read (a handful of VALU next to 4 | 8 MFMAs)
    const unsigned u_gen = (unsigned) (l31 + pq * 4 * 34);
    const unsigned gen_base = lds0 + (unsigned) (jp * 2 * XB);
#define PAIR_LOADP(G)                                                                                          \
    {                                                                                                          \
        constexpr int s_ = PAIR_G_S(G), t_ = PAIR_G_TAP(G), kh_ = PAIR_G_KH(G);                                \
        constexpr int ky_ = t_ / 3, kx_ = t_ % 3;                                                              \
        if constexpr (GEN) {                                                                                   \
            _Pragma("unroll") for (int f = 0; f < 4; ++f) {                                                    \
                const unsigned tt_ = u_gen + (unsigned) ((f + ky_) * 34 + kx_);                                \
                const unsigned ad_ = gen_base + (unsigned) (s_ * XB) + tt_ * 64u +                             \
                                     ((((unsigned) lhi ^ ((tt_ >> 2) & 3u)) << 4) ^ (unsigned) (kh_ * 32));    \
                ds_read16<0>(P[(G) & 1][f], ad_);                                                              \
            }                                                                                                  \
        } else {                                                                                               \
            const unsigned ad_ = (a_kx[kx_] + (unsigned) (s_ * XB)) ^ (unsigned) (kh_ * 32);                   \
            ds_read16<0 * 2048 + ky_ * W * 64>(P[(G) & 1][0], ad_);                                            \
            ds_read16<1 * 2048 + ky_ * W * 64>(P[(G) & 1][1], ad_);                                            \
            ds_read16<2 * 2048 + ky_ * W * 64>(P[(G) & 1][2], ad_);                                            \
            ds_read16<3 * 2048 + ky_ * W * 64>(P[(G) & 1][3], ad_);                                            \
        }                                                                                                      \
    }
    // columns beyond the left / right image edge (the conv's zero padding): output x = (fragment * 32 + l31) mod W
    const bool edge_l = (l31 & (W - 1)) == 0, edge_r = (l31 & (W - 1)) == ((W - 1) & 31);
    const frag_t zero_frag = {};
#ifndef PAIR_EXP_NOS2
    PAIR_LOADP(0);
#define PAIR_GROUP(G)                                                                                          \
    {                                                                                                          \
        if constexpr ((G) + 2 < NG) PAIR_LOADW((G) + 2);                                                       \
        if constexpr ((G) + 1 < NG) {                                                                          \
            PAIR_LOADP((G) + 1);                                                                               \
            wait_p4<4>(P[(G) & 1]);                                                                            \
        } else {                                                                                               \
            wait_p4<0>(P[(G) & 1]);                                                                            \
        }                                                                                                      \
        constexpr int kx_ = PAIR_G_TAP(G) % 3;                                                                 \
        _Pragma("unroll") for (int f = 0; f < 4; ++f) {                                                        \
            frag_t pv = P[(G) & 1][f];                                                                         \
            /* W = 64: a fragment is half a row -- the left edge lies in even, the right edge in odd fragments */ \
            /* (generic tiles carry a real halo column on both sides: nothing to mask) */                     \
            if (!GEN && kx_ == 0 && (W < 64 || (f & 1) == 0)) pv = edge_l ? zero_frag : pv;                    \
            if (!GEN && kx_ == 2 && (W < 64 || (f & 1) == 1)) pv = edge_r ? zero_frag : pv;                    \
            if constexpr (CB == 1) {                                                                           \
                acc2[PAIR_G_S(G)][f] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Wr[(G) % 3][0], pv, acc2[PAIR_G_S(G)][f], 0, 0, 0); \
            } else {                                                                                           \
                acc2[0][f] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Wr[(G) % 3][0], pv, acc2[0][f], 0, 0, 0); \
                acc2[1][f] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Wr[(G) % 3][1], pv, acc2[1][f], 0, 0, 0); \
            }                                                                                                  \
        }                                                                                                      \
    }
#define PAIR_G4(G) PAIR_GROUP(G) PAIR_GROUP((G) + 1) PAIR_GROUP((G) + 2) PAIR_GROUP((G) + 3)
    PAIR_G4(0) PAIR_G4(4) PAIR_G4(8) PAIR_G4(12) PAIR_G4(16) PAIR_G4(20) PAIR_G4(24) PAIR_G4(28) PAIR_G4(32)
#endif
#undef PAIR_G4
#undef PAIR_GROUP
#undef PAIR_LOADW
#undef PAIR_LOADP
#undef PAIR_G_S
#undef PAIR_G_TAP
#undef PAIR_G_KH

    // ---- epilogue: bias + ReLU -> bf16 -> per-wave LDS transpose -> 16-byte NHWC stores (128-byte channel runs)
    __syncthreads();  // every wave is done reading the slab tile
    constexpr int SP = 144;  // staging pitch per pixel: 64 channels x 2 B + 16 B pad
    lds_u8 *const stg = (lds_u8 *) smem + wave * (32 * SP);
    float4 b2[2][4];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q)
            b2[j][q] = a.b2 ? *(const float4 *) (a.b2 + n0 + jp * 64 + j * 32 + 8 * q + 4 * lhi) : make_float4(0.f, 0.f, 0.f, 0.f);
    const int px_l = lane >> 3, part_l = lane & 7;
    const long img_px = (long) a.H * Wimg;
    unsigned char *const dst_n = (unsigned char *) a.dst + ((size_t) n * img_px * a.dst_stride + n0 + jp * 64 + part_l * 8) * 2;
#pragma unroll
    for (int f = 0; f < 4; ++f) {
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                u32x2 o;
                o.x = pack_bf16x2(fmaxf(acc2[j][f][q * 4 + 0] + b2[j][q].x, 0.f), fmaxf(acc2[j][f][q * 4 + 1] + b2[j][q].y, 0.f));
                o.y = pack_bf16x2(fmaxf(acc2[j][f][q * 4 + 2] + b2[j][q].z, 0.f), fmaxf(acc2[j][f][q * 4 + 3] + b2[j][q].w, 0.f));
                *(lds_u32x2 *) (stg + l31 * SP + j * 64 + q * 16 + lhi * 8) = o;
            }
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int px = it * 8 + px_l;
            const u32x4 v = *(const lds_u32x4 *) (stg + px * SP + part_l * 16);
            long lin;        // pixel index within the image
            bool inside;
            if constexpr (GEN) {  // fragment = output row oy0 + pq*4 + f, px = column
                const int oy = oy0 + pq * 4 + f, ox = ox0 + px;
                lin = (long) oy * Wimg + ox;
                inside = oy < a.H && ox < a.W;
            } else {
                lin = (long) oy0 * W + (pq * 4 + f) * 32 + px;
                inside = lin < img_px;
            }
#ifdef PAIR_EXP_NOEPI
            if (a.N < 0)
#endif
            if (inside) *(u32x4 *) (dst_n + (size_t) lin * a.dst_stride * 2) = v;
        }
    }
}

constexpr size_t PAIR_LDS = 160 * 1024;
constexpr int MAX_DEVICES = 64;

// which tiling serves an image of width W with cmid conv1 output channels: 4 / 5 / 6 = full-width strips (W = 16 / 32 / 64),
// 0 = generic 16 x 32 tiles (any larger width), -1 = none
int pair_wlog(const PairArgs &a) {
    if (a.stride == 2) return (a.W >= 32 && a.cmid % 128 == 0) ? 0 : -1;  // stride-2 conv2: generic tiles only
    if (a.W == 32 && a.cmid % 256 == 0) return 5;
    if (a.W == 64 && a.cmid % 128 == 0) return 6;
    if (a.W == 16 && a.cmid % 256 == 0) return 4;
    if (a.W > 32 && a.cmid % 128 == 0 && a.cb2 == 32) return 0;  // (64-channel bundles on generic tiles spill registers)
    return -1;
}
int pair_slab(int wlog) { return (wlog == 6 || wlog == 0) ? 128 : 256; }
int pair_rows(int wlog) { return (wlog == 4 || wlog == 0) ? 16 : 8; }  // output rows per tile
int pair_cols(int wlog, int W) { return wlog == 0 ? 32 : W; }         // output columns per tile

template <int WLOG, int SL, int CB, int CPS, int STRIDE = 1>
int launch_pair_cps(const PairArgs &a, hipStream_t stream) {
    using C = PairCfg<WLOG, SL>;
    static std::atomic<bool> attr_set[MAX_DEVICES];
    auto kern = conv_pair_kernel<WLOG, SL, CB, CPS, STRIDE>;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAX_DEVICES) return (int) hipErrorInvalidDevice;
    if (!attr_set[dev].load(std::memory_order_acquire)) {
        hipError_t e = hipFuncSetAttribute((const void *) kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int) PAIR_LDS);
        if (e != hipSuccess) return (int) e;
        attr_set[dev].store(true, std::memory_order_release);
    }
    const int Ho = STRIDE == 2 ? (a.H - 1) / 2 + 1 : a.H, Wo = STRIDE == 2 ? (a.W - 1) / 2 + 1 : a.W;
    const int tiles_y = (Ho + C::THO / STRIDE - 1) / (C::THO / STRIDE);
    const int tiles_x = C::GEN ? (Wo + C::TWO / STRIDE - 1) / (C::TWO / STRIDE) : 1;
    dim3 grid((unsigned) (a.N * tiles_y * tiles_x), (unsigned) (a.cmid / SL), 1);
    hipLaunchKernelGGL(kern, grid, dim3(512), PAIR_LDS, stream, a);
    return (int) hipGetLastError();
}

template <int WLOG, int SL, int CB, int STRIDE = 1>
int launch_pair_cfg(const PairArgs &a, hipStream_t stream) {
    // two chunks per pipeline step wherever four ring slots fit the LDS (the 256-channel slabs) and the chunk count is even
    if constexpr (4 * (PairCfg<WLOG, SL>::XBUF + PairCfg<WLOG, SL>::WBUF) <= PairCfg<WLOG, SL>::LDS) {
        const char *e = getenv("CPN_PAIR_CPS");  // kernel A/B switch
        if ((a.cin & 63) == 0 && !(e && atoi(e) == 1)) return launch_pair_cps<WLOG, SL, CB, 2, STRIDE>(a, stream);
    }
    return launch_pair_cps<WLOG, SL, CB, 1, STRIDE>(a, stream);
}

}  // namespace

bool conv_pair_supported(const PairArgs &a) {
    if (a.stride != 1 && a.stride != 2) return false;
    if (a.N <= 0 || a.H <= 0 || a.W <= 0 || a.cin <= 0 || a.cin % 32 || a.cmid <= 0 || pair_wlog(a) < 0) return false;
    if (a.cb2 != 32 && a.cb2 != 64) return false;
    if (a.c_stride < a.cin || a.dst_stride < a.cmid || a.c_stride % 8 || a.dst_stride % 8) return false;
    // sources are read through raw buffer descriptors (2^31-byte limit), destinations with 32-bit element offsets
    if ((int64_t) a.N * a.H * a.W * a.c_stride * 2 >= (1ll << 31) || (int64_t) a.N * a.H * a.W * a.dst_stride >= (1ll << 31)) return false;
    return true;
}

// workgroups of the launch (tiles x slabs): below ~3/4 of the 256 CUs the two plain convs, whose tiles shrink with the
// problem, are as fast or faster (profiles/r04_pair_microbench.txt)
long conv_pair_blocks(const PairArgs &a) {
    const int wl = pair_wlog(a);
    if (wl < 0) return 0;
    const int tho = pair_rows(wl), two = pair_cols(wl, a.W);  // (stride 2: 16 x 32 input pixels = 8 x 16 outputs per tile)
    return (long) a.N * ((a.H + tho - 1) / tho) * ((a.W + two - 1) / two) * (a.cmid / pair_slab(wl));
}

int launch_conv_pair(const PairArgs &a, hipStream_t stream) {
    if (!conv_pair_supported(a) || !a.src || !a.dst || !a.w1 || !a.w2) return (int) hipErrorInvalidValue;
    const bool cb2 = a.cb2 == 64;
    switch (pair_wlog(a)) {
        case 4: return cb2 ? launch_pair_cfg<4, 256, 2>(a, stream) : launch_pair_cfg<4, 256, 1>(a, stream);
        case 5: return cb2 ? launch_pair_cfg<5, 256, 2>(a, stream) : launch_pair_cfg<5, 256, 1>(a, stream);
        case 6: return cb2 ? launch_pair_cfg<6, 128, 2>(a, stream) : launch_pair_cfg<6, 128, 1>(a, stream);
        default:
            if (a.stride == 2) return cb2 ? launch_pair_cfg<0, 128, 2, 2>(a, stream) : launch_pair_cfg<0, 128, 1, 2>(a, stream);
            return launch_pair_cfg<0, 128, 1>(a, stream);
    }
}

// MFMA FLOPs the launch executes: conv1 on every staged halo pixel (incl. the recomputed halo rows and fragment padding) +
// the block-diagonal conv2 bundles
double conv_pair_executed_flops(const PairArgs &a) {
    const int wl = pair_wlog(a);
    if (wl < 0) return 0.;
    const int tho = pair_rows(wl), two = pair_cols(wl, a.W);
    const double tiles = (double) a.N * ((a.H + tho - 1) / tho) * ((a.W + two - 1) / two);
    const double staged = pair_slab(wl) == 128 ? 640. : 320.;
    const double s1 = 2.0 * tiles * staged * (double) a.cmid * a.cin;
    const double s2 = 2.0 * tiles * (double) (tho * two / (a.stride * a.stride)) * (double) a.cmid * a.cb2 * 9.;
    return s1 + s2;
}

}  // namespace cpn
