// fp8 (OCP e4m3) variants of the HBM-bound helper kernels of the conv graph + the calibration reduction
// (groundwork for BASELINE.json configs[4]; no reference counterpart -- the reference has no fp8 path):
//   * input conversion f32/u8 NCHW -> e4m3 NHWC codes of value / scale (+ [0,1] range check)
//   * MaxPool2d, bilinear x2 resize on e4m3 codes (decode -> fp32 -> encode with the SAME tensor scale)
//   * max |x| of a bf16 NHWC tensor (static activation scales are taken from one bf16 run of the graph)
#include "cpn_kernels.h"

namespace cpn {

typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;

__device__ __forceinline__ float sat448(float v) { return __builtin_amdgcn_fmed3f(v, -448.f, 448.f); }

__device__ __forceinline__ u32x2 encode8(const float (&v)[8]) {
    int lo = 0, hi = 0;
    lo = __builtin_amdgcn_cvt_pk_fp8_f32(sat448(v[0]), sat448(v[1]), lo, false);
    lo = __builtin_amdgcn_cvt_pk_fp8_f32(sat448(v[2]), sat448(v[3]), lo, true);
    hi = __builtin_amdgcn_cvt_pk_fp8_f32(sat448(v[4]), sat448(v[5]), hi, false);
    hi = __builtin_amdgcn_cvt_pk_fp8_f32(sat448(v[6]), sat448(v[7]), hi, true);
    u32x2 o;
    o.x = (unsigned) lo; o.y = (unsigned) hi;
    return o;
}
__device__ __forceinline__ void decode8(const u32x2 r, float (&v)[8]) {
    v[0] = __builtin_amdgcn_cvt_f32_fp8((int) r.x, 0); v[1] = __builtin_amdgcn_cvt_f32_fp8((int) r.x, 1);
    v[2] = __builtin_amdgcn_cvt_f32_fp8((int) r.x, 2); v[3] = __builtin_amdgcn_cvt_f32_fp8((int) r.x, 3);
    v[4] = __builtin_amdgcn_cvt_f32_fp8((int) r.y, 0); v[5] = __builtin_amdgcn_cvt_f32_fp8((int) r.y, 1);
    v[6] = __builtin_amdgcn_cvt_f32_fp8((int) r.y, 2); v[7] = __builtin_amdgcn_cvt_f32_fp8((int) r.y, 3);
}

static int grid_for(long total) {
    long blocks = (total + 255) / 256;
    if (blocks > 256 * 16) blocks = 256 * 16;
    return (int) (blocks < 1 ? 1 : blocks);
}

// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void input_fp8_kernel(const InputArgs a, float inv_scale) {
    const int groups = a.Cpad >> 3;
    const long total = (long) a.N * a.H * a.W * groups;
    const long HW = (long) a.H * a.W;
    int bad = 0;
    for (long i = blockIdx.x * (long) blockDim.x + threadIdx.x; i < total; i += (long) gridDim.x * blockDim.x) {
        const int gidx = (int) (i % groups);
        const long pix = i / groups;
        const long n = pix / HW, p = pix - n * HW;
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int c = gidx * 8 + e;
            float x = 0.f;
            if (c < a.C) {
                const long si = (n * a.C + c) * HW + p;
                if (a.dtype == 0) x = ((const float *) a.src)[si];
                else x = (float) ((const unsigned char *) a.src)[si] / 255.f;
                if (!(x >= 0.f && x <= 1.f)) bad = 1;
            }
            v[e] = x * inv_scale;
        }
        *(u32x2 *) ((unsigned char *) a.dst + pix * a.Cpad + gidx * 8) = encode8(v);
    }
    if (bad && a.range_flag) atomicOr(a.range_flag, 1);
}

int launch_input_fp8(const InputArgs &a, float inv_scale, hipStream_t stream) {
    const long total = (long) a.N * a.H * a.W * (a.Cpad >> 3);
    hipLaunchKernelGGL(input_fp8_kernel, dim3(grid_for(total)), dim3(256), 0, stream, a, inv_scale);
    return (int) hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void maxpool_fp8_kernel(const PoolArgs a) {
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
                float v[8];
                decode8(*(const u32x2 *) ((const unsigned char *) a.src +
                                          (((long) n * a.Hin + iy) * a.Win + ix) * a.C + gidx * 8), v);
#pragma unroll
                for (int e = 0; e < 8; ++e) m[e] = fmaxf(m[e], v[e]);
            }
        }
        *(u32x2 *) ((unsigned char *) a.dst + (((long) n * a.Hout + oy) * a.Wout + ox) * a.C + gidx * 8) = encode8(m);
    }
}

int launch_maxpool_fp8(const PoolArgs &a, hipStream_t stream) {
    const long total = (long) a.N * a.Hout * a.Wout * (a.C >> 3);
    hipLaunchKernelGGL(maxpool_fp8_kernel, dim3(grid_for(total)), dim3(256), 0, stream, a);
    return (int) hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------
// one workgroup per output row (n, oy): the vertical taps and weights are wave-uniform, a thread owns 16 channels (16 bytes)
// of one output pixel -- 16-byte loads / stores and 32-bit index math (the round-1 kernel did 64-bit div / mod per 8 bytes
// in a grid-stride loop and ran at 1.5 TB/s: 885 us for the 4 x 1024^2 x 256 map in front of the FPN refinement head)
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
__device__ __forceinline__ void decode4(unsigned r, float (&v)[4]) {
    v[0] = __builtin_amdgcn_cvt_f32_fp8((int) r, 0); v[1] = __builtin_amdgcn_cvt_f32_fp8((int) r, 1);
    v[2] = __builtin_amdgcn_cvt_f32_fp8((int) r, 2); v[3] = __builtin_amdgcn_cvt_f32_fp8((int) r, 3);
}
__global__ __launch_bounds__(256) void bilinear_fp8_kernel(const ResizeArgs a) {
    const int n = (int) (blockIdx.x / (unsigned) a.Hout), oy = (int) (blockIdx.x % (unsigned) a.Hout);
    const float sy = (float) a.Hin / (float) a.Hout, sx = (float) a.Win / (float) a.Wout;
    const float fy = fmaxf(sy * ((float) oy + 0.5f) - 0.5f, 0.f);
    const int y0 = (int) fy, y1 = y0 + (y0 < a.Hin - 1 ? 1 : 0);
    const float ly = fy - (float) y0, hy = 1.f - ly;
    const unsigned g16 = (unsigned) a.C >> 4;
    // ring mode: rows within `ring` of the top / bottom border are written whole, the others at their first and last `ring` pixels
    const bool whole = a.ring <= 0 || 2 * a.ring >= a.Wout || oy < a.ring || oy >= a.Hout - a.ring;
    const unsigned per_row = (unsigned) (whole ? a.Wout : 2 * a.ring) * g16;
    const unsigned skip = whole ? 0u : (unsigned) (a.Wout - 2 * a.ring);
    const unsigned char *r0 = (const unsigned char *) a.src + ((size_t) n * a.Hin + y0) * a.Win * (size_t) a.C;
    const unsigned char *r1 = (const unsigned char *) a.src + ((size_t) n * a.Hin + y1) * a.Win * (size_t) a.C;
    unsigned char *dr = (unsigned char *) a.dst + ((size_t) n * a.Hout + oy) * a.Wout * (size_t) a.C;
    for (unsigned i = threadIdx.x; i < per_row; i += 256u) {
        const unsigned oi = i / g16, c = (i - oi * g16) << 4;
        const unsigned ox = oi + (oi >= (unsigned) a.ring ? skip : 0u);
        const float fx = fmaxf(sx * ((float) ox + 0.5f) - 0.5f, 0.f);
        const int x0 = (int) fx, x1 = x0 + (x0 < a.Win - 1 ? 1 : 0);
        const float lx = fx - (float) x0, hx = 1.f - lx;
        const u32x4 q00 = *(const u32x4 *) (r0 + (size_t) x0 * a.C + c), q01 = *(const u32x4 *) (r0 + (size_t) x1 * a.C + c);
        const u32x4 q10 = *(const u32x4 *) (r1 + (size_t) x0 * a.C + c), q11 = *(const u32x4 *) (r1 + (size_t) x1 * a.C + c);
        u32x4 out;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            float v00[4], v01[4], v10[4], v11[4];
            decode4(q00[w], v00); decode4(q01[w], v01); decode4(q10[w], v10); decode4(q11[w], v11);
            float o[4];
#pragma unroll
            for (int e = 0; e < 4; ++e)  // (same expression as before: the rounding is unchanged)
                o[e] = sat448(hy * (hx * v00[e] + lx * v01[e]) + ly * (hx * v10[e] + lx * v11[e]));
            int p = 0;
            p = __builtin_amdgcn_cvt_pk_fp8_f32(o[0], o[1], p, false);
            p = __builtin_amdgcn_cvt_pk_fp8_f32(o[2], o[3], p, true);
            out[w] = (unsigned) p;
        }
        *(u32x4 *) (dr + (size_t) ox * a.C + c) = out;
    }
}

int launch_bilinear_fp8(const ResizeArgs &a, hipStream_t stream) {
    if (a.C % 16 || a.N <= 0 || a.Hout <= 0 || a.Wout <= 0) return (int) hipErrorInvalidValue;
    hipLaunchKernelGGL(bilinear_fp8_kernel, dim3((unsigned) a.N * (unsigned) a.Hout), dim3(256), 0, stream, a);
    return (int) hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------
// max |x| over a bf16 tensor -> atomicMax on the float bits of *out (values are >= 0: integer order == float order)
__global__ __launch_bounds__(256) void absmax_bf16_kernel(const unsigned short *__restrict__ x, long count,
                                                         unsigned int *__restrict__ out) {
    float m = 0.f;
    for (long i = blockIdx.x * (long) blockDim.x + threadIdx.x; i < count; i += (long) gridDim.x * blockDim.x) {
        const float v = fabsf(__uint_as_float((unsigned int) x[i] << 16));
        m = v > m ? v : m;  // NaN never wins
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) m = fmaxf(m, __shfl_xor(m, d, 64));
    if ((threadIdx.x & 63) == 0 && m > 0.f) atomicMax(out, __float_as_uint(m));
}

int launch_absmax_bf16(const void *x, long count, float *out, hipStream_t stream) {
    hipLaunchKernelGGL(absmax_bf16_kernel, dim3(grid_for(count)), dim3(256), 0, stream, (const unsigned short *) x, count,
                       (unsigned int *) out);
    return (int) hipGetLastError();
}

// ---- elementwise activation on e4m3 codes (CPN_OP_ACT): decode with the source tensor's scale, encode with the destination's
__global__ __launch_bounds__(256) void act_fp8_kernel(const ActArgs a) {
    const long groups = a.count >> 3;
    for (long i = blockIdx.x * (long) blockDim.x + threadIdx.x; i < groups; i += (long) gridDim.x * blockDim.x) {
        float v[8];
        decode8(((const u32x2 *) a.src)[i], v);
#pragma unroll 1
        for (int e = 0; e < 8; ++e) v[e] = act_apply(v[e] * a.in_scale, a.act) * a.out_inv_scale;
        ((u32x2 *) a.dst)[i] = encode8(v);
    }
}
int launch_act_fp8(const ActArgs &a, hipStream_t stream) {
    if (a.count % 8) return (int) hipErrorInvalidValue;
    hipLaunchKernelGGL(act_fp8_kernel, dim3(grid_for(a.count >> 3)), dim3(256), 0, stream, a);
    return (int) hipGetLastError();
}

}  // namespace cpn
