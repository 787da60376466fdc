// fp32 verification path of the conv graph (precision = CPN_PRECISION_F32): NHWC fp32 activations, fp32 weights,
// accumulation in fp64 (round 5; `CPN_F32_ACC=32` restores the plain fp32 FMA chain): a product of two fp32 values is
// exact in fp64, so an output is the correctly rounded dot product and what remains between this path and the reference's
// fp32 CPU forward is the reference's OWN summation-order noise (oneDNN blocks its sums differently from any chain we
// could pick) instead of the sum of two such noises.  It exists so that the WHOLE path can be compared with the reference's fp32 CPU forward at
// the north-star tolerance (contour coordinates within 1e-4, threshold / NMS index sets equal) -- the bf16 MFMA path
// cannot be bit-compatible end-to-end because `scores > thresh` and round-half-even are discontinuous.
// Throughput is not a goal here (a few TFLOP/s on the vector ALUs); every fused feature of the bf16 kernel is
// supported with identical semantics (virtual concat, nearest-resized sources, resized residual, bundles, activations,
// fp32 NCHW head outputs).  Weights: [bundle][kh*kw][cin_b][cout_b] fp32.
#include <cstdlib>

#include "cpn_kernels.h"

namespace cpn {

template <typename ACC>
__global__ __launch_bounds__(256) void conv_f32_kernel(const ConvArgs a) {
    // one thread = one output pixel x 4 consecutive output channels of one bundle
    const int cq_per_b = a.cout_b >> 2;
    const long total = (long) a.N * a.Hout * a.Wout * a.bundles * cq_per_b;
    const long idx = blockIdx.x * 256l + threadIdx.x;
    if (idx >= total) return;
    const int cq = (int) (idx % cq_per_b);
    long t = idx / cq_per_b;
    const int g = (int) (t % a.bundles);
    t /= a.bundles;
    const int ox = (int) (t % a.Wout);
    t /= a.Wout;
    const int oy = (int) (t % a.Hout);
    const int n = (int) (t / a.Hout);
    const int Hs0 = a.Hs0, Ws0 = a.Ws0, Hs1 = a.Hs1, Ws1 = a.Ws1;
    const float *src0 = (const float *) a.src0, *src1 = (const float *) a.src1;
    const float *W = (const float *) a.weights + (size_t) g * a.KH * a.KW * a.cin_b * a.cout_b + cq * 4;
    ACC acc0 = 0, acc1 = 0, acc2 = 0, acc3 = 0;
    for (int ky = 0; ky < a.KH; ++ky) {
        const int iy = oy * a.stride - a.pad + ky;
        if (iy < 0 || iy >= a.Hin) continue;
        for (int kx = 0; kx < a.KW; ++kx) {
            const int ix = ox * a.stride - a.pad + kx;
            if (ix < 0 || ix >= a.Win) continue;
            const int y0 = a.up0 ? nearest_src(iy, a.sy0, Hs0) : iy, x0 = a.up0 ? nearest_src(ix, a.sx0, Ws0) : ix;
            const int y1 = a.up1 ? nearest_src(iy, a.sy1, Hs1) : iy, x1 = a.up1 ? nearest_src(ix, a.sx1, Ws1) : ix;
            const size_t p0 = ((size_t) (n * Hs0 + y0) * Ws0 + x0) * a.c0_stride;
            const size_t p1 = ((size_t) (n * Hs1 + y1) * Ws1 + x1) * a.c1_stride;
            const float *wt = W + (size_t) (ky * a.KW + kx) * a.cin_b * a.cout_b;
            for (int c = 0; c < a.cin_b; c += 4) {
                const int cin = g * a.cin_b + c;
                const float4 x = cin < a.c0_used ? *(const float4 *) (src0 + p0 + cin)
                                                  : *(const float4 *) (src1 + p1 + (cin - a.c0_used));
                const float4 w0 = *(const float4 *) (wt + (size_t) (c + 0) * a.cout_b);
                const float4 w1 = *(const float4 *) (wt + (size_t) (c + 1) * a.cout_b);
                const float4 w2 = *(const float4 *) (wt + (size_t) (c + 2) * a.cout_b);
                const float4 w3 = *(const float4 *) (wt + (size_t) (c + 3) * a.cout_b);
                const ACC xx = x.x, xy = x.y, xz = x.z, xw = x.w;
                acc0 += xx * (ACC) w0.x + xy * (ACC) w1.x + xz * (ACC) w2.x + xw * (ACC) w3.x;
                acc1 += xx * (ACC) w0.y + xy * (ACC) w1.y + xz * (ACC) w2.y + xw * (ACC) w3.y;
                acc2 += xx * (ACC) w0.z + xy * (ACC) w1.z + xz * (ACC) w2.z + xw * (ACC) w3.z;
                acc3 += xx * (ACC) w0.w + xy * (ACC) w1.w + xz * (ACC) w2.w + xw * (ACC) w3.w;
            }
        }
    }
    const int co = g * a.cout_b + cq * 4;
    if (a.bias) {  // (added before the one rounding to fp32)
        const float4 b = *(const float4 *) (a.bias + co);
        acc0 += (ACC) b.x; acc1 += (ACC) b.y; acc2 += (ACC) b.z; acc3 += (ACC) b.w;
    }
    float v[4] = {(float) acc0, (float) acc1, (float) acc2, (float) acc3};
    const size_t pix = ((size_t) n * a.Hout + oy) * a.Wout + ox;
    if (a.out_mode == OUT_BF16_NHWC) {  // (fp32 NHWC in this precision)
        if (a.res) {
            size_t rpix = pix;
            if (a.res_up) rpix = ((size_t) n * a.Hr + nearest_src(oy, a.ry, a.Hr)) * a.Wr + nearest_src(ox, a.rx, a.Wr);
            const float4 r = *(const float4 *) ((const float *) a.res + rpix * a.res_stride + co);
            v[0] += r.x; v[1] += r.y; v[2] += r.z; v[3] += r.w;
        }
        if (a.act == ACT_RELU)
            for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
        *(float4 *) ((float *) a.dst + pix * a.dst_stride + a.dst_coff + co) = make_float4(v[0], v[1], v[2], v[3]);
    } else {
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

int launch_conv_f32(const ConvArgs &a, hipStream_t stream) {
    if (a.cin_b % 4 || a.cout_b % 4 || a.out_mode == OUT_FUSED_HEAD) return (int) hipErrorInvalidValue;
    const long total = (long) a.N * a.Hout * a.Wout * a.bundles * (a.cout_b >> 2);
    static const bool acc32 = [] { const char *e = getenv("CPN_F32_ACC"); return e && atoi(e) == 32; }();
    if (acc32) hipLaunchKernelGGL(conv_f32_kernel<float>, dim3((unsigned) ((total + 255) / 256)), dim3(256), 0, stream, a);
    else hipLaunchKernelGGL(conv_f32_kernel<double>, dim3((unsigned) ((total + 255) / 256)), dim3(256), 0, stream, a);
    return (int) hipGetLastError();
}

// ---- fp32 helpers (one thread per 4 channels) -------------------------------------------------------------------
__global__ __launch_bounds__(256) void input_f32_kernel(const InputArgs a) {
    const int groups = a.Cpad >> 2;
    const long total = (long) a.N * a.H * a.W * groups;
    const long HW = (long) a.H * a.W;
    const long i = blockIdx.x * 256l + threadIdx.x;
    if (i >= total) return;
    const int gidx = (int) (i % groups);
    const long pix = i / groups;
    const long n = pix / HW, p = pix - n * HW;
    float v[4];
    int bad = 0;
    for (int e = 0; e < 4; ++e) {
        const int c = gidx * 4 + e;
        v[e] = 0.f;
        if (c < a.C) {
            const long si = (n * a.C + c) * HW + p;
            v[e] = a.dtype == 0 ? ((const float *) a.src)[si] : (float) ((const unsigned char *) a.src)[si] / 255.f;
            if (!(v[e] >= 0.f && v[e] <= 1.f)) bad = 1;
        }
    }
    *(float4 *) ((float *) a.dst + pix * a.Cpad + gidx * 4) = make_float4(v[0], v[1], v[2], v[3]);
    if (bad && a.range_flag) atomicOr(a.range_flag, 1);
}

__global__ __launch_bounds__(256) void maxpool_f32_kernel(const PoolArgs a) {
    const int groups = a.C >> 2;
    const long total = (long) a.N * a.Hout * a.Wout * groups;
    const long i = blockIdx.x * 256l + threadIdx.x;
    if (i >= total) return;
    const int gidx = (int) (i % groups);
    long pix = i / groups;
    const int ox = (int) (pix % a.Wout);
    pix /= a.Wout;
    const int oy = (int) (pix % a.Hout);
    const int n = (int) (pix / a.Hout);
    float4 m = make_float4(-__builtin_inff(), -__builtin_inff(), -__builtin_inff(), -__builtin_inff());
    for (int ky = 0; ky < a.k; ++ky) {
        const int iy = oy * a.stride - a.pad + ky;
        if (iy < 0 || iy >= a.Hin) continue;
        for (int kx = 0; kx < a.k; ++kx) {
            const int ix = ox * a.stride - a.pad + kx;
            if (ix < 0 || ix >= a.Win) continue;
            const float4 v = *(const float4 *) ((const float *) a.src + (((long) n * a.Hin + iy) * a.Win + ix) * a.C + gidx * 4);
            m.x = fmaxf(m.x, v.x); m.y = fmaxf(m.y, v.y); m.z = fmaxf(m.z, v.z); m.w = fmaxf(m.w, v.w);
        }
    }
    *(float4 *) ((float *) a.dst + (((long) n * a.Hout + oy) * a.Wout + ox) * a.C + gidx * 4) = m;
}

__global__ __launch_bounds__(256) void bilinear_f32_kernel(const ResizeArgs a) {
    // PyTorch upsample_bilinear2d (align_corners=False) including its operation order:
    // w0*(h0*v00 + h1*v01) ... evaluated as hy*(hx*v00 + lx*v01) + ly*(hx*v10 + lx*v11)
    const int groups = a.C >> 2;
    const long total = (long) a.N * a.Hout * a.Wout * groups;
    const long i = blockIdx.x * 256l + threadIdx.x;
    if (i >= total) return;
    const int gidx = (int) (i % groups);
    long pix = i / groups;
    const int ox = (int) (pix % a.Wout);
    pix /= a.Wout;
    const int oy = (int) (pix % a.Hout);
    const int n = (int) (pix / a.Hout);
    const float sy = (float) a.Hin / (float) a.Hout, sx = (float) a.Win / (float) a.Wout;
    if (a.mode == 1) {  // bicubic: see bicubic_taps (cpn_kernels.h)
        int iy[4], ix[4];
        float wy[4], wx[4];
        bicubic_taps(sy, oy, a.Hin, iy, wy);
        bicubic_taps(sx, ox, a.Win, ix, wx);
        const float *base = (const float *) a.src + (long) n * a.Hin * a.Win * a.C + gidx * 4;
        float4 o = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float4 r = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float4 v = *(const float4 *) (base + ((long) iy[i] * a.Win + ix[j]) * a.C);
                r.x = j == 0 ? v.x * wx[0] : r.x + v.x * wx[j]; r.y = j == 0 ? v.y * wx[0] : r.y + v.y * wx[j];
                r.z = j == 0 ? v.z * wx[0] : r.z + v.z * wx[j]; r.w = j == 0 ? v.w * wx[0] : r.w + v.w * wx[j];
            }
            o.x = i == 0 ? r.x * wy[0] : o.x + r.x * wy[i]; o.y = i == 0 ? r.y * wy[0] : o.y + r.y * wy[i];
            o.z = i == 0 ? r.z * wy[0] : o.z + r.z * wy[i]; o.w = i == 0 ? r.w * wy[0] : o.w + r.w * wy[i];
        }
        *(float4 *) ((float *) a.dst + (((long) n * a.Hout + oy) * a.Wout + ox) * a.C + gidx * 4) = o;
        return;
    }
    const float fy = fmaxf(sy * ((float) oy + 0.5f) - 0.5f, 0.f), fx = fmaxf(sx * ((float) ox + 0.5f) - 0.5f, 0.f);
    const int y0 = (int) fy, x0 = (int) fx;
    const int y1 = y0 + (y0 < a.Hin - 1 ? 1 : 0), x1 = x0 + (x0 < a.Win - 1 ? 1 : 0);
    const float ly = fy - (float) y0, lx = fx - (float) x0, hy = 1.f - ly, hx = 1.f - lx;
    const float *base = (const float *) a.src + (long) n * a.Hin * a.Win * a.C + gidx * 4;
    const float4 v00 = *(const float4 *) (base + ((long) y0 * a.Win + x0) * a.C);
    const float4 v01 = *(const float4 *) (base + ((long) y0 * a.Win + x1) * a.C);
    const float4 v10 = *(const float4 *) (base + ((long) y1 * a.Win + x0) * a.C);
    const float4 v11 = *(const float4 *) (base + ((long) y1 * a.Win + x1) * a.C);
    float4 o;
    o.x = hy * (hx * v00.x + lx * v01.x) + ly * (hx * v10.x + lx * v11.x);
    o.y = hy * (hx * v00.y + lx * v01.y) + ly * (hx * v10.y + lx * v11.y);
    o.z = hy * (hx * v00.z + lx * v01.z) + ly * (hx * v10.z + lx * v11.z);
    o.w = hy * (hx * v00.w + lx * v01.w) + ly * (hx * v10.w + lx * v11.w);
    *(float4 *) ((float *) a.dst + (((long) n * a.Hout + oy) * a.Wout + ox) * a.C + gidx * 4) = o;
}

int launch_input_f32(const InputArgs &a, hipStream_t stream) {
    const long total = (long) a.N * a.H * a.W * (a.Cpad >> 2);
    hipLaunchKernelGGL(input_f32_kernel, dim3((unsigned) ((total + 255) / 256)), dim3(256), 0, stream, a);
    return (int) hipGetLastError();
}
int launch_maxpool_f32(const PoolArgs &a, hipStream_t stream) {
    const long total = (long) a.N * a.Hout * a.Wout * (a.C >> 2);
    hipLaunchKernelGGL(maxpool_f32_kernel, dim3((unsigned) ((total + 255) / 256)), dim3(256), 0, stream, a);
    return (int) hipGetLastError();
}
int launch_bilinear_f32(const ResizeArgs &a, hipStream_t stream) {
    const long total = (long) a.N * a.Hout * a.Wout * (a.C >> 2);
    hipLaunchKernelGGL(bilinear_f32_kernel, dim3((unsigned) ((total + 255) / 256)), dim3(256), 0, stream, a);
    return (int) hipGetLastError();
}

__global__ __launch_bounds__(256) void act_f32_kernel(const ActArgs a) {
    for (long i = blockIdx.x * (long) blockDim.x + threadIdx.x; i < a.count; i += (long) gridDim.x * blockDim.x)
        ((float *) a.dst)[i] = act_apply(((const float *) a.src)[i], a.act);
}
int launch_act_f32(const ActArgs &a, hipStream_t stream) {
    long blocks = (a.count + 255) / 256;
    if (blocks > 256 * 32) blocks = 256 * 32;
    hipLaunchKernelGGL(act_f32_kernel, dim3((unsigned) (blocks < 1 ? 1 : blocks)), dim3(256), 0, stream, a);
    return (int) hipGetLastError();
}

}  // namespace cpn
