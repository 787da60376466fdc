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
//     re-read for every tap (49x reuse for the 7x7 heads), weights stream through a double-buffered LDS slab;
//     global->register prefetch of step s+1 is issued before the MFMAs of step s (one barrier per step);
//   * LDS records are 64 B of data + 16 B pad (80 B): ds_read_b128 fragment reads are bank-conflict free for
//     stride-1 convs (4 LDS cycles per wave instruction, checked against the per-instruction lane groups);
//   * grouped convs (ResNeXt cardinality 32) run as independent dense "bundles" (grid.z) of >=32 channels with
//     block-diagonal packed weights.
#include "cpn_kernels.h"

namespace cpn {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;

constexpr int PS = 80;   // LDS bytes per 32-channel record (pixel or weight row): 64 B data + 16 B pad
constexpr int TW = 32;   // output tile width in pixels (= one MFMA column fragment)
constexpr int HREG = 4;  // halo prefetch iterations held in registers

__device__ __forceinline__ unsigned int f32_to_bf16_bits(float f) {
    unsigned int u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (u >> 16) | 0x40u;  // NaN
    u += 0x7fffu + ((u >> 16) & 1u);                                 // round to nearest even
    return u >> 16;
}
__device__ __forceinline__ float bf16_bits_to_f32(unsigned int b) { return __uint_as_float(b << 16); }

template <int TH, int BN, int WM, int WN>
struct Cfg {
    static constexpr int WAVES_M = TH / WM;
    static constexpr int WAVES_N = BN / (32 * WN);
    static constexpr int NWAVES = WAVES_M * WAVES_N;
    static constexpr int THREADS = 64 * NWAVES;
    static constexpr int W_PARTS = BN * 4;  // 16-byte parts of one weight slab tile
    static constexpr int W_ITERS = (W_PARTS + THREADS - 1) / THREADS;
    static_assert(TH % WM == 0 && BN % (32 * WN) == 0, "bad tile");
};

struct HaloGeom {
    int HH, HWp, hpix, parts;
};

template <int TH, int BN, int WM, int WN>
__global__ __launch_bounds__((64 * (TH / WM) * (BN / (32 * WN)))) void conv_igemm_kernel(const ConvArgs a) {
    using C = Cfg<TH, BN, WM, WN>;
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

    const int s = a.stride;
    const int HH = (TH - 1) * s + a.KH;
    const int HWp = (TW - 1) * s + a.KW;
    const int hpix = HH * HWp;
    const int hparts = hpix * 4;
    const int halo_bytes = hpix * PS;
    const int nchunks = a.cin_b >> 5;
    const int ntaps = a.KH * a.KW;
    const int nsteps = nchunks * ntaps;
    const int nhalo_buf = nchunks > 1 ? 2 : 1;
    unsigned char *const ldsA = smem;
    unsigned char *const ldsW = smem + nhalo_buf * halo_bytes;
    constexpr int WBUF = BN * PS;

    const int iy0 = oy0 * s - a.pad, ix0 = ox0 * s - a.pad;

    // ---- per-thread halo geometry (source pixel offsets in elements, -1 = zero padding), fixed over chunks
    const int Hs0 = a.up0 ? (a.Hin >> 1) : a.Hin, Ws0 = a.up0 ? (a.Win >> 1) : a.Win;
    const int Hs1 = a.up1 ? (a.Hin >> 1) : a.Hin, Ws1 = a.up1 ? (a.Win >> 1) : a.Win;
    auto src_offsets = [&](int idx, int &o0, int &o1, int &lds_off) {
        const int pix = idx >> 2, part = idx & 3;
        const int hy = pix / HWp, hx = pix - hy * HWp;
        const int iy = iy0 + hy, ix = ix0 + hx;
        const bool valid = (idx < hparts) && iy >= 0 && iy < a.Hin && ix >= 0 && ix < a.Win;
        lds_off = pix * PS + part * 16;
        if (valid) {
            const int y0 = a.up0 ? (iy >> 1) : iy, x0 = a.up0 ? (ix >> 1) : ix;
            const int y1 = a.up1 ? (iy >> 1) : iy, x1 = a.up1 ? (ix >> 1) : ix;
            o0 = ((n * Hs0 + y0) * Ws0 + x0) * a.c0_stride + part * 8;
            o1 = ((n * Hs1 + y1) * Ws1 + x1) * a.c1_stride + part * 8;
        } else {
            o0 = -1;
            o1 = -1;
        }
    };
    int h_o0[HREG], h_o1[HREG], h_lds[HREG];
#pragma unroll
    for (int it = 0; it < HREG; ++it) src_offsets(tid + it * C::THREADS, h_o0[it], h_o1[it], h_lds[it]);

    const unsigned short *const src0 = (const unsigned short *) a.src0;
    const unsigned short *const src1 = (const unsigned short *) a.src1;
    const unsigned char *const wbase_g =
            (const unsigned char *) a.weights + (size_t) g * nchunks * ntaps * a.cout_b * 64;

    u32x4 hreg[HREG];
    u32x4 wreg[C::W_ITERS];

    // issue the global loads of a chunk's halo (register part)
    auto halo_issue = [&](int c) {
        const int cin = g * a.cin_b + c * 32;
        const bool from0 = cin < a.c0_used;
        const unsigned short *base = from0 ? src0 + cin : src1 + (cin - a.c0_used);
#pragma unroll
        for (int it = 0; it < HREG; ++it) {
            const int off = from0 ? h_o0[it] : h_o1[it];
            u32x4 v = {0u, 0u, 0u, 0u};
            if (off >= 0) v = *(const u32x4 *) (base + off);
            hreg[it] = v;
        }
    };
    auto halo_commit = [&](int c) {
        unsigned char *dstb = ldsA + (nhalo_buf == 2 ? (c & 1) * halo_bytes : 0);
#pragma unroll
        for (int it = 0; it < HREG; ++it)
            if (tid + it * C::THREADS < hparts) *(u32x4 *) (dstb + h_lds[it]) = hreg[it];
        // remainder (halo larger than HREG*THREADS parts): synchronous load+store
        const int cin = g * a.cin_b + c * 32;
        const bool from0 = cin < a.c0_used;
        const unsigned short *base = from0 ? src0 + cin : src1 + (cin - a.c0_used);
        for (int idx = tid + HREG * C::THREADS; idx < hparts; idx += C::THREADS) {
            int o0, o1, l;
            src_offsets(idx, o0, o1, l);
            const int off = from0 ? o0 : o1;
            u32x4 v = {0u, 0u, 0u, 0u};
            if (off >= 0) v = *(const u32x4 *) (base + off);
            *(u32x4 *) (dstb + l) = v;
        }
    };
    auto w_issue = [&](int step) {
        const unsigned char *slab = wbase_g + ((size_t) step * a.cout_b + n0) * 64;
#pragma unroll
        for (int it = 0; it < C::W_ITERS; ++it) {
            const int idx = tid + it * C::THREADS;
            u32x4 v = {0u, 0u, 0u, 0u};
            if (idx < C::W_PARTS && n0 + (idx >> 2) < a.cout_b) v = *(const u32x4 *) (slab + (size_t) idx * 16);
            wreg[it] = v;
        }
    };
    auto w_commit = [&](int step) {
        unsigned char *dstb = ldsW + (step & 1) * WBUF;
#pragma unroll
        for (int it = 0; it < C::W_ITERS; ++it) {
            const int idx = tid + it * C::THREADS;
            if (idx < C::W_PARTS) *(u32x4 *) (dstb + (idx >> 2) * PS + (idx & 3) * 16) = wreg[it];
        }
    };

    // ---- accumulators
    f32x16 acc[WN][WM];
#pragma unroll
    for (int j = 0; j < WN; ++j)
#pragma unroll
        for (int f = 0; f < WM; ++f)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][f][r] = 0.f;

    const int l31 = lane & 31, lhi = lane >> 5;
    const int w_lane = (wave_n * WN * 32 + l31) * PS + lhi * 16;
    const int p_lane = (wave_m * WM * s * HWp + l31 * s) * PS + lhi * 16;
    const int p_frag_stride = s * HWp * PS;

    // ---- prologue
    halo_issue(0);
    w_issue(0);
    halo_commit(0);
    w_commit(0);
    __syncthreads();

    int c = 0, t = 0, ky = 0, kx = 0;
    for (int step = 0; step < nsteps; ++step) {
        const int nstep = step + 1;
        const bool has_next = nstep < nsteps;
        const bool new_chunk = has_next && (t + 1 == ntaps);
        if (has_next) w_issue(nstep);
        if (new_chunk) halo_issue(c + 1);

        const unsigned char *A = ldsA + (nhalo_buf == 2 ? (c & 1) * halo_bytes : 0) + (ky * HWp + kx) * PS + p_lane;
        const unsigned char *Wb = ldsW + (step & 1) * WBUF + w_lane;
#pragma unroll
        for (int kh = 0; kh < 2; ++kh) {
            bf16x8 wf[WN], pf[WM];
#pragma unroll
            for (int j = 0; j < WN; ++j) wf[j] = *(const bf16x8 *) (Wb + j * 32 * PS + kh * 32);
#pragma unroll
            for (int f = 0; f < WM; ++f) pf[f] = *(const bf16x8 *) (A + f * p_frag_stride + kh * 32);
#pragma unroll
            for (int j = 0; j < WN; ++j)
#pragma unroll
                for (int f = 0; f < WM; ++f)
                    acc[j][f] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[j], pf[f], acc[j][f], 0, 0, 0);
        }

        if (has_next) w_commit(nstep);
        if (new_chunk) halo_commit(c + 1);
        __syncthreads();
        // advance (c, t)
        ++t;
        ++kx;
        if (kx == a.KW) { kx = 0; ++ky; }
        if (t == ntaps) { t = 0; ky = 0; kx = 0; ++c; }
    }

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
    int TH, BN, WM, WN;
};

static size_t lds_bytes(const ConvArgs &a, int TH, int BN) {
    const int HH = (TH - 1) * a.stride + a.KH, HWp = (TW - 1) * a.stride + a.KW;
    const int nchunks = a.cin_b / 32;
    return (size_t) (nchunks > 1 ? 2 : 1) * HH * HWp * PS + 2 * (size_t) BN * PS;
}

template <int TH, int BN, int WM, int WN>
static int launch_cfg(const ConvArgs &a, hipStream_t stream) {
    using C = Cfg<TH, BN, WM, WN>;
    const size_t lds = lds_bytes(a, TH, BN);
    static size_t max_set = 0;
    auto kern = conv_igemm_kernel<TH, BN, WM, WN>;
    if (lds > max_set) {
        hipError_t e = hipFuncSetAttribute((const void *) kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return (int) e;
        max_set = 160 * 1024;
    }
    const int tiles_x = (a.Wout + TW - 1) / TW, tiles_y = (a.Hout + TH - 1) / TH;
    dim3 grid((unsigned) (tiles_x * tiles_y * a.N), (unsigned) ((a.cout_b + BN - 1) / BN), (unsigned) a.bundles);
    hipLaunchKernelGGL(kern, grid, dim3(C::THREADS), lds, stream, a);
    return (int) hipGetLastError();
}

static TileChoice choose_tile(const ConvArgs &a) {
    int BN = a.cout_b >= 256 ? 256 : a.cout_b >= 128 ? 128 : a.cout_b >= 64 ? 64 : 32;
    int TH = 8;
    auto blocks = [&](int th, int bn) {
        return (long) ((a.Wout + TW - 1) / TW) * ((a.Hout + th - 1) / th) * a.N * ((a.cout_b + bn - 1) / bn) * a.bundles;
    };
    const size_t LDS_MAX = 160 * 1024;
    if (lds_bytes(a, 8, BN) > LDS_MAX || a.Hout < 8) TH = 4;
    // prefer >= 2 workgroups per CU worth of blocks: shrink the tile while the grid is small
    if (TH == 8 && blocks(8, BN) < 512) TH = 4;
    if (blocks(TH, BN) < 512 && BN > 128) BN = 128;
    if (blocks(TH, BN) < 512 && BN > 64) BN = 64;
    while (lds_bytes(a, TH, BN) > LDS_MAX && BN > 32) BN >>= 1;
    TileChoice c{TH, BN, 0, 0};
    return c;
}

int launch_conv(const ConvArgs &a, hipStream_t stream) {
    if (a.cin_b % 32 || a.cout_b % 32 || a.c0_used % 32) return (int) hipErrorInvalidValue;
    const TileChoice c = choose_tile(a);
    if (lds_bytes(a, c.TH, c.BN) > 160 * 1024) return (int) hipErrorInvalidValue;
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
