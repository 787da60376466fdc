// HBM-bound helper kernels of the conv graph (NHWC bf16, 16-byte vector accesses, gfx950 only):
//   * input conversion  f32/u8 NCHW -> bf16 NHWC (zero-padded channels) + [0,1] range check
//     (reference: Normalize assert, celldetection/models/commons.py:694-700; LitBase.prepare_inputs u8->f32/255,
//      celldetection/models/lightning_base.py:774-780)
//   * MaxPool2d(3,2,1) / MaxPool2d(2,2)   (celldetection/models/resnet.py:279, unet.py:56)
//   * bilinear resize, align_corners=False (celldetection/models/cpn.py:109-115,277-279 `_equal_size`)
#include "cpn_kernels.h"

namespace cpn {

typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

__device__ __forceinline__ unsigned int to_bf16(float f) {
    unsigned int u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (u >> 16) | 0x40u;
    u += 0x7fffu + ((u >> 16) & 1u);
    return u >> 16;
}
__device__ __forceinline__ float from_bf16(unsigned int b) { return __uint_as_float(b << 16); }

// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void input_kernel(const InputArgs a) {
    // one thread per (pixel, 8-channel group); Cpad is a multiple of 8
    const int groups = a.Cpad >> 3;
    const long total = (long) a.N * a.H * a.W * groups;
    const long HW = (long) a.H * a.W;
    int bad = 0;
    for (long i = blockIdx.x * (long) blockDim.x + threadIdx.x; i < total; i += (long) gridDim.x * blockDim.x) {
        const int gidx = (int) (i % groups);
        const long pix = i / groups;
        const long n = pix / HW, p = pix - n * HW;
        unsigned int h[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int c = gidx * 8 + e;
            float v = 0.f;
            if (c < a.C) {
                const long si = (n * a.C + c) * HW + p;
                if (a.dtype == 0) v = ((const float *) a.src)[si];
                else v = (float) ((const unsigned char *) a.src)[si] / 255.f;
                if (!(v >= 0.f && v <= 1.f)) bad = 1;
            }
            h[e] = to_bf16(v);
        }
        u32x4 o;
        o.x = h[0] | (h[1] << 16);
        o.y = h[2] | (h[3] << 16);
        o.z = h[4] | (h[5] << 16);
        o.w = h[6] | (h[7] << 16);
        *(u32x4 *) ((unsigned short *) a.dst + pix * a.Cpad + gidx * 8) = o;
    }
    if (bad && a.range_flag) atomicOr(a.range_flag, 1);
}

int launch_input(const InputArgs &a, hipStream_t stream) {
    const long total = (long) a.N * a.H * a.W * (a.Cpad >> 3);
    int blocks = (int) ((total + 255) / 256);
    if (blocks > 256 * 16) blocks = 256 * 16;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(input_kernel, dim3(blocks), dim3(256), 0, stream, a);
    return (int) hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void maxpool_kernel(const PoolArgs a) {
    const int groups = a.C >> 3;
    const long total = (long) a.N * a.Hout * a.Wout * groups;
    for (long i = blockIdx.x * (long) blockDim.x + threadIdx.x; i < total; i += (long) gridDim.x * blockDim.x) {
        const int gidx = (int) (i % groups);
        long pix = i / groups;
        const int ox = (int) (pix % a.Wout);
        pix /= a.Wout;
        const int oy = (int) (pix % a.Hout);
        const int n = (int) (pix / a.Hout);
        float m[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) m[e] = -__builtin_inff();
        for (int ky = 0; ky < a.k; ++ky) {
            const int iy = oy * a.stride - a.pad + ky;
            if (iy < 0 || iy >= a.Hin) continue;
            for (int kx = 0; kx < a.k; ++kx) {
                const int ix = ox * a.stride - a.pad + kx;
                if (ix < 0 || ix >= a.Win) continue;
                const u32x4 v = *(const u32x4 *) ((const unsigned short *) a.src +
                                                 (((long) n * a.Hin + iy) * a.Win + ix) * a.C + gidx * 8);
                const unsigned int w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    m[2 * e] = fmaxf(m[2 * e], from_bf16(w[e] & 0xffffu));
                    m[2 * e + 1] = fmaxf(m[2 * e + 1], from_bf16(w[e] >> 16));
                }
            }
        }
        u32x4 o;
        o.x = to_bf16(m[0]) | (to_bf16(m[1]) << 16);
        o.y = to_bf16(m[2]) | (to_bf16(m[3]) << 16);
        o.z = to_bf16(m[4]) | (to_bf16(m[5]) << 16);
        o.w = to_bf16(m[6]) | (to_bf16(m[7]) << 16);
        *(u32x4 *) ((unsigned short *) a.dst + (((long) n * a.Hout + oy) * a.Wout + ox) * a.C + gidx * 8) = o;
    }
}

int launch_maxpool(const PoolArgs &a, hipStream_t stream) {
    const long total = (long) a.N * a.Hout * a.Wout * (a.C >> 3);
    int blocks = (int) ((total + 255) / 256);
    if (blocks > 256 * 16) blocks = 256 * 16;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(maxpool_kernel, dim3(blocks), dim3(256), 0, stream, a);
    return (int) hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void bilinear_kernel(const ResizeArgs a) {
    // PyTorch upsample_bilinear2d, align_corners=False: src = max(scale*(dst+0.5)-0.5, 0), scale = in/out
    const int groups = a.C >> 3;
    const long total = (long) a.N * a.Hout * a.Wout * groups;
    const float sy = (float) a.Hin / (float) a.Hout, sx = (float) a.Win / (float) a.Wout;
    for (long i = blockIdx.x * (long) blockDim.x + threadIdx.x; i < total; i += (long) gridDim.x * blockDim.x) {
        const int gidx = (int) (i % groups);
        long pix = i / groups;
        const int ox = (int) (pix % a.Wout);
        pix /= a.Wout;
        const int oy = (int) (pix % a.Hout);
        const int n = (int) (pix / a.Hout);
        if (a.mode == 1) {  // bicubic (refinement_interpolation='bicubic', models/cpn.py:109-115,277-279): 4 x 4 taps, fp32 math
            int iy[4], ix[4];
            float wy[4], wx[4];
            bicubic_taps(sy, oy, a.Hin, iy, wy);
            bicubic_taps(sx, ox, a.Win, ix, wx);
            const unsigned short *base = (const unsigned short *) a.src + (long) n * a.Hin * a.Win * a.C + gidx * 8;
            float acc[8];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float row[8];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const u32x4 v = *(const u32x4 *) (base + ((long) iy[i] * a.Win + ix[j]) * a.C);
                    const unsigned int q[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float lo = from_bf16(q[e] & 0xffffu), hi = from_bf16(q[e] >> 16);
                        row[2 * e] = j == 0 ? lo * wx[0] : row[2 * e] + lo * wx[j];
                        row[2 * e + 1] = j == 0 ? hi * wx[0] : row[2 * e + 1] + hi * wx[j];
                    }
                }
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[e] = i == 0 ? row[e] * wy[0] : acc[e] + row[e] * wy[i];
            }
            u32x4 ov;
            ov.x = to_bf16(acc[0]) | (to_bf16(acc[1]) << 16); ov.y = to_bf16(acc[2]) | (to_bf16(acc[3]) << 16);
            ov.z = to_bf16(acc[4]) | (to_bf16(acc[5]) << 16); ov.w = to_bf16(acc[6]) | (to_bf16(acc[7]) << 16);
            *(u32x4 *) ((unsigned short *) a.dst + (((long) n * a.Hout + oy) * a.Wout + ox) * a.C + gidx * 8) = ov;
            continue;
        }
        const float fy = fmaxf(sy * ((float) oy + 0.5f) - 0.5f, 0.f);
        const float fx = fmaxf(sx * ((float) ox + 0.5f) - 0.5f, 0.f);
        const int y0 = (int) fy, x0 = (int) fx;
        const int y1 = y0 + (y0 < a.Hin - 1 ? 1 : 0), x1 = x0 + (x0 < a.Win - 1 ? 1 : 0);
        const float ly = fy - (float) y0, lx = fx - (float) x0;
        const float hy = 1.f - ly, hx = 1.f - lx;
        const unsigned short *base = (const unsigned short *) a.src + (long) n * a.Hin * a.Win * a.C + gidx * 8;
        const u32x4 v00 = *(const u32x4 *) (base + ((long) y0 * a.Win + x0) * a.C);
        const u32x4 v01 = *(const u32x4 *) (base + ((long) y0 * a.Win + x1) * a.C);
        const u32x4 v10 = *(const u32x4 *) (base + ((long) y1 * a.Win + x0) * a.C);
        const u32x4 v11 = *(const u32x4 *) (base + ((long) y1 * a.Win + x1) * a.C);
        const unsigned int a00[4] = {v00.x, v00.y, v00.z, v00.w}, a01[4] = {v01.x, v01.y, v01.z, v01.w};
        const unsigned int a10[4] = {v10.x, v10.y, v10.z, v10.w}, a11[4] = {v11.x, v11.y, v11.z, v11.w};
        unsigned int o[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float lo = hy * (hx * from_bf16(a00[e] & 0xffffu) + lx * from_bf16(a01[e] & 0xffffu)) +
                             ly * (hx * from_bf16(a10[e] & 0xffffu) + lx * from_bf16(a11[e] & 0xffffu));
            const float hi = hy * (hx * from_bf16(a00[e] >> 16) + lx * from_bf16(a01[e] >> 16)) +
                             ly * (hx * from_bf16(a10[e] >> 16) + lx * from_bf16(a11[e] >> 16));
            o[e] = to_bf16(lo) | (to_bf16(hi) << 16);
        }
        u32x4 ov;
        ov.x = o[0]; ov.y = o[1]; ov.z = o[2]; ov.w = o[3];
        *(u32x4 *) ((unsigned short *) a.dst + (((long) n * a.Hout + oy) * a.Wout + ox) * a.C + gidx * 8) = ov;
    }
}

int launch_bilinear(const ResizeArgs &a, hipStream_t stream) {
    const long total = (long) a.N * a.Hout * a.Wout * (a.C >> 3);
    int blocks = (int) ((total + 255) / 256);
    if (blocks > 256 * 16) blocks = 256 * 16;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(bilinear_kernel, dim3(blocks), dim3(256), 0, stream, a);
    return (int) hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------
// percentile normalisation of a whole slide before tiling (celldetection_scripts/cpn_inference.py:196-222 `preprocess`
// -> cd.data.normalize_percentile, celldetection/data/misc.py:156-161): value histogram of 8/16-bit integer images
// (exact order statistics without sorting 10^9 values) and the clip/rescale/round-to-uint8 pass
template <typename T>
__global__ __launch_bounds__(256) void histogram_kernel(const T *__restrict__ x, long n, unsigned int *__restrict__ hist) {
    // 8-bit: per-block LDS histogram; 16-bit: 65536 bins do not fit the per-block budget -> global atomics (values of
    // natural images spread over thousands of bins: little contention)
    __shared__ unsigned int lh[256];
    constexpr bool small = sizeof(T) == 1;
    if (small) {
        lh[threadIdx.x] = 0;
        __syncthreads();
    }
    for (long i = blockIdx.x * 256l + threadIdx.x; i < n; i += (long) gridDim.x * 256l) {
        const unsigned int v = (unsigned int) x[i];
        if (small) atomicAdd(&lh[v], 1u);
        else atomicAdd(&hist[v], 1u);
    }
    if (small) {
        __syncthreads();
        if (lh[threadIdx.x]) atomicAdd(&hist[threadIdx.x], lh[threadIdx.x]);
    }
}

template <typename T>
__global__ __launch_bounds__(256) void rescale_u8_kernel(const T *__restrict__ x, long n, double low, double high,
                                                        unsigned char *__restrict__ out) {
    // img = (clip(image, low, high) - low) / (high - low)  (float64, data/misc.py:160); img_as_ubyte: rint(img * 255)
    const double inv = high - low;
    for (long i = blockIdx.x * 256l + threadIdx.x; i < n; i += (long) gridDim.x * 256l) {
        double v = (double) x[i];
        v = fmin(fmax(v, low), high);
        v = (v - low) / inv;
        v = rint(v * 255.0);
        out[i] = (unsigned char) fmin(fmax(v, 0.0), 255.0);
    }
}

int launch_histogram(const void *x, int dtype, long n, unsigned int *hist, hipStream_t stream) {
    int blocks = (int) ((n + 256 * 16 - 1) / (256 * 16));
    blocks = blocks < 1 ? 1 : (blocks > 256 * 32 ? 256 * 32 : blocks);
    if (dtype == 1) hipLaunchKernelGGL(histogram_kernel<unsigned char>, dim3(blocks), dim3(256), 0, stream, (const unsigned char *) x, n, hist);
    else if (dtype == 2) hipLaunchKernelGGL(histogram_kernel<unsigned short>, dim3(blocks), dim3(256), 0, stream, (const unsigned short *) x, n, hist);
    else return (int) hipErrorInvalidValue;
    return (int) hipGetLastError();
}

int launch_rescale_u8(const void *x, int dtype, long n, double low, double high, unsigned char *out, hipStream_t stream) {
    int blocks = (int) ((n + 256 * 8 - 1) / (256 * 8));
    blocks = blocks < 1 ? 1 : (blocks > 256 * 64 ? 256 * 64 : blocks);
    if (dtype == 0) hipLaunchKernelGGL(rescale_u8_kernel<float>, dim3(blocks), dim3(256), 0, stream, (const float *) x, n, low, high, out);
    else if (dtype == 1) hipLaunchKernelGGL(rescale_u8_kernel<unsigned char>, dim3(blocks), dim3(256), 0, stream, (const unsigned char *) x, n, low, high, out);
    else if (dtype == 2) hipLaunchKernelGGL(rescale_u8_kernel<unsigned short>, dim3(blocks), dim3(256), 0, stream, (const unsigned short *) x, n, low, high, out);
    else return (int) hipErrorInvalidValue;
    return (int) hipGetLastError();
}

// ---- elementwise activation (CPN_OP_ACT): 8 bf16 per thread, fp32 math as torch.nn computes it, one rounding
__global__ __launch_bounds__(256) void act_bf16_kernel(const ActArgs a) {
    const long groups = a.count >> 3;
    for (long i = blockIdx.x * (long) blockDim.x + threadIdx.x; i < groups; i += (long) gridDim.x * blockDim.x) {
        const uint4 r = ((const uint4 *) a.src)[i];
        const unsigned in[4] = {r.x, r.y, r.z, r.w};
        unsigned out[4];
#pragma unroll 1
        for (int e = 0; e < 4; ++e) {
            const float lo = act_apply(__uint_as_float(in[e] << 16), a.act), hi = act_apply(__uint_as_float(in[e] & 0xffff0000u), a.act);
            typedef __attribute__((ext_vector_type(2))) __bf16 bf2;
            typedef __attribute__((ext_vector_type(2))) float f2;
            const f2 v = {lo, hi};
            out[e] = __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf2));
        }
        ((uint4 *) a.dst)[i] = make_uint4(out[0], out[1], out[2], out[3]);
    }
}
int launch_act(const ActArgs &a, hipStream_t stream) {
    if (a.count % 8) return (int) hipErrorInvalidValue;
    long blocks = ((a.count >> 3) + 255) / 256;
    if (blocks > 256 * 16) blocks = 256 * 16;
    hipLaunchKernelGGL(act_bf16_kernel, dim3((unsigned) (blocks < 1 ? 1 : blocks)), dim3(256), 0, stream, a);
    return (int) hipGetLastError();
}



// ---------------------------------------------------------------------------------------------------------------------
// Tile pre-filter of the slide loop (TileLoader.__init__ skips tiles whose mask crop is empty,
// celldetection_scripts/cpn_inference.py:88-100): any(mask[y0:y1, x0:x1] != 0) for every window of the tiling table in ONE
// launch -- one workgroup per window, rows strided over the waves, 16 bytes per lane and step where the row allows it.
// HBM-bound: a window is read once; overlapping windows (stride < crop) re-read the overlap from L2.
template <typename T>
__global__ __launch_bounds__(256) void window_any_kernel(const T *__restrict__ mask, int W, const int *__restrict__ win,
                                                         int *__restrict__ out) {
    const int *wv = win + 4 * blockIdx.x;
    const int y0 = wv[0], y1 = wv[1], x0 = wv[2], x1 = wv[3];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    bool any = false;
    for (int y = y0 + wave; y < y1 && !any; y += 4) {
        const T *row = mask + (long) y * W;
        for (int x = x0 + lane; x < x1; x += 64) any |= row[x] != (T) 0;
    }
    if (__ballot(any) != 0ull && lane == 0) atomicOr(out + blockIdx.x, 1);
}

int launch_window_any(const void *mask, int dtype, int W, const int *windows, int n, int *out, hipStream_t stream) {
    if (n <= 0) return 0;
    if (dtype == 0) hipLaunchKernelGGL(window_any_kernel<float>, dim3(n), dim3(256), 0, stream, (const float *) mask, W, windows, out);
    else hipLaunchKernelGGL(window_any_kernel<unsigned char>, dim3(n), dim3(256), 0, stream, (const unsigned char *) mask, W, windows, out);
    return (int) hipGetLastError();
}

}  // namespace cpn
