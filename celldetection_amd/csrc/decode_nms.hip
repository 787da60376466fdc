// Proposal extraction, Fourier-to-contour decode, local refinement, boxes and greedy box NMS for gfx950 (wave64).
// Compiled with -ffp-contract=off: the arithmetic below must reproduce the reference's fp32 operation order
// bit-for-bit (no FMA contraction), because `scores > thresh`, round-half-even and `iou > thresh` are discontinuous.
//
// Reference call sites replaced: see include/cpn_hip.h (each entry point cites file:line).
#include <cstring>
#include <hip/hip_runtime.h>
#include <rocprim/device/device_radix_sort.hpp>
#include <stdint.h>

#include "../../include/cpn_hip.h"
#include "cpn_error.h"
#include "cpn_kernels.h"

namespace {

constexpr int WAVE = 64;

// =========================================================================================================
// 1. order-preserving compaction of scores > thresh
// =========================================================================================================
constexpr int CBLK = 256;

__global__ __launch_bounds__(CBLK) void compact_count_kernel(const float *__restrict__ scores, int hw, float thresh,
                                                            int *__restrict__ block_counts) {
    const int b = blockIdx.y, i = blockIdx.x * CBLK + threadIdx.x;
    const bool flag = (i < hw) && (scores[(size_t) b * hw + i] > thresh);
    const unsigned long long m = __ballot(flag);
    __shared__ int wsum[CBLK / WAVE];
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = __popcll(m);
    __syncthreads();
    if (threadIdx.x == 0) {
        int s = 0;
        for (int k = 0; k < CBLK / WAVE; ++k) s += wsum[k];
        block_counts[b * gridDim.x + blockIdx.x] = s;
    }
}

// single block: exclusive scan of block_counts -> block_offsets, per-image counts, total
__global__ __launch_bounds__(1024) void compact_scan_kernel(const int *__restrict__ block_counts, int nb, int bpi, int N,
                                                           int *__restrict__ block_offsets, int *__restrict__ counts) {
    __shared__ int wsum[16];
    __shared__ int carry_s;
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int base = 0; base < nb; base += 1024) {
        const int i = base + threadIdx.x;
        const int v = i < nb ? block_counts[i] : 0;
        // inclusive scan within wave
        int x = v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int y = __shfl_up(x, d, 64);
            if (lane >= d) x += y;
        }
        if (lane == 63) wsum[wave] = x;
        __syncthreads();
        int woff = 0;
        for (int k = 0; k < wave; ++k) woff += wsum[k];
        const int carry = carry_s;
        if (i < nb) block_offsets[i] = carry + woff + x - v;
        __syncthreads();
        if (threadIdx.x == 1023) carry_s = carry + woff + x;
        __syncthreads();
    }
    // per-image counts from offsets (blocks never straddle images)
    for (int b = threadIdx.x; b < N; b += 1024) {
        const int start = block_offsets[b * bpi];
        const int end = (b + 1 < N) ? block_offsets[(b + 1) * bpi] : carry_s;
        counts[b] = end - start;
    }
    if (threadIdx.x == 0) counts[N] = carry_s;
}

__global__ __launch_bounds__(CBLK) void compact_write_kernel(const float *__restrict__ scores, int hw, float thresh,
                                                            const int *__restrict__ block_offsets,
                                                            int *__restrict__ indices) {
    const int b = blockIdx.y, i = blockIdx.x * CBLK + threadIdx.x;
    const bool flag = (i < hw) && (scores[(size_t) b * hw + i] > thresh);
    const unsigned long long m = __ballot(flag);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __shared__ int wsum[CBLK / WAVE];
    if (lane == 0) wsum[wave] = __popcll(m);
    __syncthreads();
    int off = block_offsets[b * gridDim.x + blockIdx.x];
    for (int k = 0; k < wave; ++k) off += wsum[k];
    if (flag) indices[off + __popcll(m & ((1ull << lane) - 1ull))] = b * hw + i;
}

// =========================================================================================================
// 2. decode
// =========================================================================================================
constexpr int DWAVES = 4;       // proposals per block
constexpr int MAX_COEF = 256;   // order*4 <= 256  (order <= 64)

__device__ __forceinline__ float wave_min(float v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v = fminf(v, __shfl_xor(v, d, 64));
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v = fmaxf(v, __shfl_xor(v, d, 64));
    return v;
}

// x/y of one contour sample: ((loc + sum_k f[k][sincol]*sin[k][s]) + sum_k f[k][coscol]*cos[k][s]), ops/cpn.py:91-94
__device__ __forceinline__ float synth(const float *coef, int order, int samples, int s, int sincol, int coscol,
                                       const float *__restrict__ cos_t, const float *__restrict__ sin_t, float loc) {
    float a = __fmul_rn(coef[sincol], sin_t[s]);
    for (int k = 1; k < order; ++k) a = __fadd_rn(a, __fmul_rn(coef[k * 4 + sincol], sin_t[k * samples + s]));
    float v = __fadd_rn(loc, a);
    float c = __fmul_rn(coef[coscol], cos_t[s]);
    for (int k = 1; k < order; ++k) c = __fadd_rn(c, __fmul_rn(coef[k * 4 + coscol], cos_t[k * samples + s]));
    return __fadd_rn(v, c);
}

// bucketed refinement (models/cpn.py:72-82, ops/cpn.py:238-255): sample s blends the channel pairs of three
// neighbouring buckets; idx/w are host-built [3][samples] tables (bucket index, weight) in the reference's order a,b,c
struct Buckets {
    int n;               // refinement_buckets (1 = plain two-channel map)
    const int32_t *idx;  // [3][samples]
    const float *w;      // [3][samples]
    int samples;
};

__device__ __forceinline__ void refine_point(float &cx, float &cy, const float *__restrict__ ref_b, int H, int W,
                                             int iterations, const Buckets &B, int s) {
    // models/cpn.py:63-85: round (half to even) -> clamp -> gather -> add
    const size_t plane = (size_t) H * W;
    for (int it = 0; it < iterations; ++it) {
        cx = fminf(fmaxf(rintf(cx), 0.f), (float) (W - 1));
        cy = fminf(fmaxf(rintf(cy), 0.f), (float) (H - 1));
        const int ix = (int) cx, iy = (int) cy;
        const size_t o = (size_t) iy * W + ix;
        if (B.n <= 1) {
            cx = __fadd_rn(cx, ref_b[o]);
            cy = __fadd_rn(cy, ref_b[plane + o]);
        } else {  // responses = (r_a*w_a + r_b*w_b) + r_c*w_c, every product and sum rounded (cpn.py:76-81)
            float rx = 0.f, ry = 0.f;
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const int bi = B.idx[k * B.samples + s];
                const float wk = B.w[k * B.samples + s];
                const float tx = __fmul_rn(ref_b[(size_t) (2 * bi) * plane + o], wk);
                const float ty = __fmul_rn(ref_b[(size_t) (2 * bi + 1) * plane + o], wk);
                rx = k == 0 ? tx : __fadd_rn(rx, tx);
                ry = k == 0 ? ty : __fadd_rn(ry, ty);
            }
            cx = __fadd_rn(cx, rx);
            cy = __fadd_rn(cy, ry);
        }
    }
}

struct DecodeArgs {
    const int32_t *indices;
    int32_t P;
    const float *scores, *locations, *fourier, *refinement;
    int32_t N, h, w, H, W, order_total, order, samples, iterations;
    const float *cos_t, *sin_t;
    const float *offsets;  // [N][2] xy; the reference adds int64 offsets to fp32 tensors = fp32 add of (float) offset
    float *contours, *proposals, *boxes, *out_scores, *out_locations, *out_fourier;
    int32_t *batch_index;
    Buckets bk;
};

// GATHERED: locations [P,2] / fourier [P,4*order_total] hold the head values of proposal p (cpn_sparse_heads) instead of
// dense NCHW maps
template <bool GATHERED>
__global__ __launch_bounds__(DWAVES *WAVE) void decode_kernel(const DecodeArgs a) {
    __shared__ float coef_s[DWAVES][MAX_COEF];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int p = blockIdx.x * DWAVES + wave;
    const bool active = p < a.P;
    float *coef = coef_s[wave];
    const int hw = a.h * a.w;
    int b = 0, y = 0, x = 0, lin = 0;
    if (active) {
        lin = a.indices[p];
        b = lin / hw;
        const int rem = lin - b * hw;
        y = rem / a.w;
        x = rem - y * a.w;
        for (int i = lane; i < a.order * 4; i += 64)
            coef[i] = GATHERED ? a.fourier[(size_t) p * a.order_total * 4 + i]
                               : a.fourier[((size_t) b * a.order_total * 4 + i) * hw + (size_t) y * a.w + x];
    }
    __syncthreads();
    if (!active) return;
    const size_t pos = (size_t) y * a.w + x;
    // rel_location2abs_location, ops/cpn.py:15-41
    const float lx = __fadd_rn(GATHERED ? a.locations[(size_t) p * 2] : a.locations[((size_t) b * 2 + 0) * hw + pos], (float) x);
    const float ly = __fadd_rn(GATHERED ? a.locations[(size_t) p * 2 + 1] : a.locations[((size_t) b * 2 + 1) * hw + pos], (float) y);
    const float sx = (float) a.W / (float) a.w, sy = (float) a.H / (float) a.h;  // get_scale, ops/cpn.py:98-103
    float offx = 0.f, offy = 0.f;
    if (a.offsets) {
        offx = a.offsets[b * 2 + 0];
        offy = a.offsets[b * 2 + 1];
    }
    const bool do_refine = a.refinement != nullptr && a.iterations > 0;
    const float *ref_b = do_refine ? a.refinement + (size_t) b * 2 * (a.bk.n < 1 ? 1 : a.bk.n) * a.H * a.W : nullptr;
    float mnx = __builtin_inff(), mny = __builtin_inff(), mxx = -__builtin_inff(), mxy = -__builtin_inff();
    for (int s = lane; s < a.samples; s += 64) {
        float px = synth(coef, a.order, a.samples, s, 1, 0, a.cos_t, a.sin_t, lx);
        float py = synth(coef, a.order, a.samples, s, 3, 2, a.cos_t, a.sin_t, ly);
        px = __fmul_rn(px, sx);  // scale_contours, ops/cpn.py:106-130
        py = __fmul_rn(py, sy);
        float cx = px, cy = py;
        if (do_refine) refine_point(cx, cy, ref_b, a.H, a.W, a.iterations, a.bk, s);
        cx = fminf(fmaxf(cx, 0.f), (float) (a.W - 1));  // models/cpn.py:661-663
        cy = fminf(fmaxf(cy, 0.f), (float) (a.H - 1));
        if (!do_refine) { px = cx; py = cy; }  // proposals alias the contours tensor in the reference
        mnx = fminf(mnx, cx); mny = fminf(mny, cy);
        mxx = fmaxf(mxx, cx); mxy = fmaxf(mxy, cy);
        const size_t o = ((size_t) p * a.samples + s) * 2;
        float ox = __fadd_rn(cx, offx), oy = __fadd_rn(cy, offy);
        float qx = __fadd_rn(px, offx), qy = __fadd_rn(py, offy);
        if (!do_refine && a.offsets) {
            // without refinement `selected_contours` IS `selected_contour_proposals` in the reference (cpn.py:655-656):
            // the two in-place `+= offsets` (cpn.py:697-699) hit the one tensor twice; boxes/locations get it once
            ox = __fadd_rn(ox, offx); oy = __fadd_rn(oy, offy);
            qx = ox; qy = oy;
        }
        a.contours[o] = ox;
        a.contours[o + 1] = oy;
        a.proposals[o] = qx;
        a.proposals[o + 1] = qy;
    }
    mnx = wave_min(mnx); mny = wave_min(mny);
    mxx = wave_max(mxx); mxy = wave_max(mxy);
    if (lane == 0) {
        float *bx = a.boxes + (size_t) p * 4;
        bx[0] = __fadd_rn(mnx, offx); bx[1] = __fadd_rn(mny, offy);
        bx[2] = __fadd_rn(mxx, offx); bx[3] = __fadd_rn(mxy, offy);
        a.out_scores[p] = a.scores[lin];
        a.out_locations[(size_t) p * 2] = __fadd_rn(__fmul_rn(lx, sx), offx);  // scale_fourier, ops/cpn.py:133-165
        a.out_locations[(size_t) p * 2 + 1] = __fadd_rn(__fmul_rn(ly, sy), offy);
        a.batch_index[p] = b;
    }
    for (int i = lane; i < a.order * 4; i += 64)
        a.out_fourier[(size_t) p * a.order * 4 + i] = __fmul_rn(coef[i], (i & 3) < 2 ? sx : sy);
}

__global__ __launch_bounds__(DWAVES *WAVE) void f2c_kernel(const float *__restrict__ fourier,
                                                          const float *__restrict__ locations, int P, int order,
                                                          int samples, const float *__restrict__ cos_t,
                                                          const float *__restrict__ sin_t, float *__restrict__ contours) {
    __shared__ float coef_s[DWAVES][MAX_COEF];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int p = blockIdx.x * DWAVES + wave;
    float *coef = coef_s[wave];
    if (p < P)
        for (int i = lane; i < order * 4; i += 64) coef[i] = fourier[(size_t) p * order * 4 + i];
    __syncthreads();
    if (p >= P) return;
    const float lx = locations[(size_t) p * 2], ly = locations[(size_t) p * 2 + 1];
    for (int s = lane; s < samples; s += 64) {
        contours[((size_t) p * samples + s) * 2] = synth(coef, order, samples, s, 1, 0, cos_t, sin_t, lx);
        contours[((size_t) p * samples + s) * 2 + 1] = synth(coef, order, samples, s, 3, 2, cos_t, sin_t, ly);
    }
}

__global__ __launch_bounds__(256) void refine_kernel(float *__restrict__ contours, const int32_t *__restrict__ bidx,
                                                    long total, int samples, const float *__restrict__ refinement,
                                                    int H, int W, int iterations, const Buckets bk) {
    const long i = blockIdx.x * 256l + threadIdx.x;
    if (i >= total) return;
    const int b = bidx[i / samples];
    float cx = contours[i * 2], cy = contours[i * 2 + 1];
    refine_point(cx, cy, refinement + (size_t) b * 2 * (bk.n < 1 ? 1 : bk.n) * H * W, H, W, iterations, bk,
                 (int) (i % samples));
    contours[i * 2] = cx;
    contours[i * 2 + 1] = cy;
}

__global__ __launch_bounds__(256) void border_kernel(const float *__restrict__ contours, long P, int samples, float offx,
                                                    float offy, float h, float w, float pad, int sides,
                                                    uint8_t *__restrict__ keep) {
    // one wave per contour (ops/cpn.py:258-290)
    const int lane = threadIdx.x & 63;
    const long p = blockIdx.x * 4l + (threadIdx.x >> 6);
    if (p >= P) return;
    bool ok = true;
    for (int s = lane; s < samples; s += 64) {
        const float x = __fadd_rn(contours[(p * samples + s) * 2], offx);
        const float y = __fadd_rn(contours[(p * samples + s) * 2 + 1], offy);
        if (sides & 1) ok = ok && (y > pad);
        if (sides & 2) ok = ok && (x < __fsub_rn(w, pad));
        if (sides & 4) ok = ok && (y < __fsub_rn(h, pad));
        if (sides & 8) ok = ok && (x > pad);
    }
    const bool all_ok = __all(ok);
    if (lane == 0) keep[p] = all_ok ? 1 : 0;
}

// all detections of a forwarded batch at once: contour p belongs to image img[p]; per-image tile parameters
// (neighbour-side bit mask, xy offsets that are ADDED to the coordinates) come from device tables, so the slide loop
// needs one launch per batch instead of one launch + boolean indexing per tile
__global__ __launch_bounds__(256) void border_batched_kernel(const float *__restrict__ contours, long P, int samples,
                                                            const int32_t *__restrict__ img,
                                                            const int32_t *__restrict__ sides_tab,
                                                            const float *__restrict__ off_tab, float h, float w,
                                                            float pad, uint8_t *__restrict__ keep) {
    const int lane = threadIdx.x & 63;
    const long p = blockIdx.x * 4l + (threadIdx.x >> 6);
    if (p >= P) return;
    const int b = img[p];
    const int sides = sides_tab[b];
    const float offx = off_tab[2 * b], offy = off_tab[2 * b + 1];
    bool ok = true;
    for (int s = lane; s < samples; s += 64) {
        const float x = __fadd_rn(contours[(p * samples + s) * 2], offx);
        const float y = __fadd_rn(contours[(p * samples + s) * 2 + 1], offy);
        if (sides & 1) ok = ok && (y > pad);
        if (sides & 2) ok = ok && (x < __fsub_rn(w, pad));
        if (sides & 4) ok = ok && (y < __fsub_rn(h, pad));
        if (sides & 8) ok = ok && (x > pad);
    }
    const bool all_ok = __all(ok);
    if (lane == 0) keep[p] = all_ok ? 1 : 0;
}

// fp32 NCHW bilinear resize, align_corners=False, of score-bound masks / head maps (`_equal_size`,
// models/cpn.py:109-123,279): torch CPU's separable form out = (v00*wx0 + v01*wx1)*wy0 + (v10*wx0 + v11*wx1)*wy1 with
// src = max(scale*(dst+0.5)-0.5, 0), scale = in/out, w1 = src - floor(src), w0 = 1 - w1; every product/sum rounded
__global__ __launch_bounds__(256) void resize_f32_kernel(const float *__restrict__ src, float *__restrict__ dst,
                                                        long planes, int Hin, int Win, int Hout, int Wout, int mode) {
    const long i = blockIdx.x * 256l + threadIdx.x;
    const long total = planes * Hout * Wout;
    if (i >= total) return;
    const int ox = (int) (i % Wout);
    const long t = i / Wout;
    const int oy = (int) (t % Hout);
    const long pl = t / Hout;
    const float sy = (float) Hin / (float) Hout, sx = (float) Win / (float) Wout;
    const float *p = src + pl * Hin * Win;
    if (mode == 1) {  // bicubic, align_corners=False (cpn_kernels.h bicubic_taps): sum_i wy[i] * (sum_j wx[j] * v[i][j])
        int iy[4], ix[4];
        float wy[4], wx[4];
        cpn::bicubic_taps(sy, oy, Hin, iy, wy);
        cpn::bicubic_taps(sx, ox, Win, ix, wx);
        float o = 0.f;
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const float *row = p + (long) iy[a] * Win;
            float r = __fmul_rn(row[ix[0]], wx[0]);
#pragma unroll
            for (int b = 1; b < 4; ++b) r = __fadd_rn(r, __fmul_rn(row[ix[b]], wx[b]));
            o = a == 0 ? __fmul_rn(r, wy[0]) : __fadd_rn(o, __fmul_rn(r, wy[a]));
        }
        dst[i] = o;
        return;
    }
    int y0 = oy, y1 = oy, x0 = ox, x1 = ox;
    float wy0 = 1.f, wy1 = 0.f, wx0 = 1.f, wx1 = 0.f;
    if (Hin != Hout) {
        const float fy = fmaxf(__fsub_rn(__fmul_rn(sy, __fadd_rn((float) oy, 0.5f)), 0.5f), 0.f);
        y0 = min((int) fy, Hin - 1);
        y1 = y0 + (y0 < Hin - 1 ? 1 : 0);
        wy1 = fminf(fmaxf(__fsub_rn(fy, (float) y0), 0.f), 1.f);
        wy0 = __fsub_rn(1.f, wy1);
    }
    if (Win != Wout) {
        const float fx = fmaxf(__fsub_rn(__fmul_rn(sx, __fadd_rn((float) ox, 0.5f)), 0.5f), 0.f);
        x0 = min((int) fx, Win - 1);
        x1 = x0 + (x0 < Win - 1 ? 1 : 0);
        wx1 = fminf(fmaxf(__fsub_rn(fx, (float) x0), 0.f), 1.f);
        wx0 = __fsub_rn(1.f, wx1);
    }
    const float ra = __fadd_rn(__fmul_rn(p[(long) y0 * Win + x0], wx0), __fmul_rn(p[(long) y0 * Win + x1], wx1));
    const float rb = __fadd_rn(__fmul_rn(p[(long) y1 * Win + x0], wx0), __fmul_rn(p[(long) y1 * Win + x1], wx1));
    dst[i] = __fadd_rn(__fmul_rn(ra, wy0), __fmul_rn(rb, wy1));
}

// =========================================================================================================
// 2b. score variants: multi-class softmax/argmax (models/cpn.py:583-585,631-632), certainty filter (cpn.py:617-618),
//     per-proposal channel gather (cpn.py:634-636)
// =========================================================================================================
// one thread per pixel: probs = softmax(logits) over C channels, bounds (min with upper, max with lower) applied to
// every channel, cls = first argmax; sel = probs[cls]; fg = cls > 0
__global__ __launch_bounds__(256) void class_scores_kernel(const float *__restrict__ logits, int N, int C, int hw,
                                                          const float *__restrict__ lower,
                                                          const float *__restrict__ upper, float *__restrict__ probs,
                                                          float *__restrict__ sel, int32_t *__restrict__ cls,
                                                          float *__restrict__ fg) {
    const long i = blockIdx.x * 256l + threadIdx.x;
    if (i >= (long) N * hw) return;
    const int b = (int) (i / hw);
    const long pos = i - (long) b * hw;
    const float *lg = logits + (size_t) b * C * hw + pos;
    float mx = lg[0];
    for (int c = 1; c < C; ++c) mx = fmaxf(mx, lg[(size_t) c * hw]);
    float sum = 0.f;
    for (int c = 0; c < C; ++c) sum = __fadd_rn(sum, expf(__fsub_rn(lg[(size_t) c * hw], mx)));
    const float ub = upper ? upper[i] : __builtin_inff(), lb = lower ? lower[i] : -__builtin_inff();
    float best = -__builtin_inff();
    int arg = 0;
    for (int c = 0; c < C; ++c) {
        float p = __fdiv_rn(expf(__fsub_rn(lg[(size_t) c * hw], mx)), sum);
        p = fmaxf(fminf(p, ub), lb);
        if (probs) probs[(size_t) b * C * hw + (size_t) c * hw + pos] = p;
        if (p > best) { best = p; arg = c; }
    }
    sel[i] = best;
    cls[i] = arg;
    fg[i] = arg > 0 ? 1.f : 0.f;
}

// out = mean_c(uncertainty) < limit ? scores : -1   (fg_mask &= uncertainty.mean(1) < 1 - certainty_thresh)
__global__ __launch_bounds__(256) void certainty_kernel(const float *__restrict__ scores,
                                                       const float *__restrict__ unc, int N, int C, int hw, float limit,
                                                       float *__restrict__ out) {
    const long i = blockIdx.x * 256l + threadIdx.x;
    if (i >= (long) N * hw) return;
    const int b = (int) (i / hw);
    const long pos = i - (long) b * hw;
    float s = unc[(size_t) b * C * hw + pos];
    for (int c = 1; c < C; ++c) s = __fadd_rn(s, unc[(size_t) b * C * hw + (size_t) c * hw + pos]);
    const float m = __fdiv_rn(s, (float) C);
    out[i] = m < limit ? scores[i] : -1.f;
}

// out[p][c] = map[b][c][pos] for the linear pixel index lin = b*hw + pos of proposal p
__global__ __launch_bounds__(256) void gather_channels_kernel(const float *__restrict__ map,
                                                             const int32_t *__restrict__ indices, long P, int C, int hw,
                                                             float *__restrict__ out) {
    const long i = blockIdx.x * 256l + threadIdx.x;
    if (i >= P * C) return;
    const long p = i / C;
    const int c = (int) (i - p * C);
    const int lin = indices[p];
    const int b = lin / hw;
    out[i] = map[(size_t) b * C * hw + (size_t) c * hw + (lin - b * hw)];
}

// votes[i] = sum_j iou(i, j) * (iou(i, j) > thresh)  (get_iou_voting, celldetection/ops/boxes.py:52-58; IoU as
// torchvision.ops.box_iou: inter / (area_i + area_j - inter), NaN propagates like `iou *= iou > thresh`)
__global__ __launch_bounds__(256) void box_votes_kernel(const float *__restrict__ boxes, long P, float thresh,
                                                       float *__restrict__ votes) {
    const int lane = threadIdx.x & 63;
    const long i = blockIdx.x * 4l + (threadIdx.x >> 6);
    if (i >= P) return;
    const float4 a = *(const float4 *) (boxes + i * 4);
    const float area_a = __fmul_rn(__fsub_rn(a.z, a.x), __fsub_rn(a.w, a.y));
    float acc = 0.f;
    for (long j = lane; j < P; j += 64) {
        const float4 b = *(const float4 *) (boxes + j * 4);
        const float area_b = __fmul_rn(__fsub_rn(b.z, b.x), __fsub_rn(b.w, b.y));
        const float w = fmaxf(__fsub_rn(fminf(a.z, b.z), fmaxf(a.x, b.x)), 0.f);
        const float h = fmaxf(__fsub_rn(fminf(a.w, b.w), fmaxf(a.y, b.y)), 0.f);
        const float inter = __fmul_rn(w, h);
        const float iou = __fdiv_rn(inter, __fsub_rn(__fadd_rn(area_a, area_b), inter));
        acc = __fadd_rn(acc, __fmul_rn(iou, iou > thresh ? 1.f : 0.f));
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) acc = __fadd_rn(acc, __shfl_xor(acc, d, 64));
    if (lane == 0) votes[i] = acc;
}

// =========================================================================================================
// 3. NMS
// =========================================================================================================
__global__ __launch_bounds__(256) void nms_keys_kernel(const float *__restrict__ scores, long P,
                                                      const int64_t *__restrict__ seg_off, int nseg,
                                                      unsigned long long *__restrict__ keys,
                                                      unsigned int *__restrict__ vals) {
    const long i = blockIdx.x * 256l + threadIdx.x;
    if (i >= P) return;
    int seg = 0;
    {  // last segment with seg_off[seg] <= i
        int lo = 0, hi = nseg - 1;
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (seg_off[mid] <= i) lo = mid; else hi = mid - 1;
        }
        seg = lo;
    }
    float s = scores[i];
    unsigned int u = __float_as_uint(s);
    if ((u & 0x7fffffffu) > 0x7f800000u) u = 0x7fc00000u;  // NaN sorts first in a descending torch.sort
    else if (u == 0x80000000u) u = 0u;                       // -0 == +0
    const unsigned int asc = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
    keys[i] = ((unsigned long long) seg << 32) | (unsigned long long) (~asc);
    vals[i] = (unsigned int) i;
}

__global__ __launch_bounds__(256) void nms_gather_kernel(const float *__restrict__ boxes,
                                                        const unsigned int *__restrict__ vals, long P,
                                                        float4 *__restrict__ sboxes) {
    const long i = blockIdx.x * 256l + threadIdx.x;
    if (i >= P) return;
    sboxes[i] = ((const float4 *) boxes)[vals[i]];
}

__global__ __launch_bounds__(64) void nms_mask_kernel(const float4 *__restrict__ sboxes,
                                                     const int64_t *__restrict__ seg_off, float thr, long CW,
                                                     unsigned long long *__restrict__ mask) {
    const int seg = blockIdx.z;
    const long s0 = seg_off[seg];
    const long cnt = seg_off[seg + 1] - s0;
    const long rb = blockIdx.y, cb = blockIdx.x;
    if (rb * 64 >= cnt || cb * 64 >= cnt || cb < rb) return;
    __shared__ float4 cbox[64];
    const int t = threadIdx.x;
    const long col0 = cb * 64;
    if (col0 + t < cnt) cbox[t] = sboxes[s0 + col0 + t];
    __syncthreads();
    const long row = rb * 64 + t;
    if (row >= cnt) return;
    const float4 me = sboxes[s0 + row];
    const float iarea = __fmul_rn(__fsub_rn(me.z, me.x), __fsub_rn(me.w, me.y));
    unsigned long long bits = 0ull;
    const int ncol = (int) ((cnt - col0) < 64 ? (cnt - col0) : 64);
    for (int j = 0; j < ncol; ++j) {
        if (col0 + j <= row) continue;
        const float4 o = cbox[j];
        const float xx1 = fmaxf(me.x, o.x), yy1 = fmaxf(me.y, o.y);
        const float xx2 = fminf(me.z, o.z), yy2 = fminf(me.w, o.w);
        const float w = fmaxf(0.f, __fsub_rn(xx2, xx1)), h = fmaxf(0.f, __fsub_rn(yy2, yy1));
        const float inter = __fmul_rn(w, h);
        const float jarea = __fmul_rn(__fsub_rn(o.z, o.x), __fsub_rn(o.w, o.y));
        const float ovr = __fdiv_rn(inter, __fsub_rn(__fadd_rn(iarea, jarea), inter));
        if (ovr > thr) bits |= (1ull << j);
    }
    mask[(size_t) (s0 + row) * CW + cb] = bits;
}

__global__ __launch_bounds__(256) void nms_scan_kernel(const unsigned long long *__restrict__ mask,
                                                      const unsigned int *__restrict__ vals,
                                                      const int64_t *__restrict__ seg_off, long CW,
                                                      int64_t *__restrict__ keep, int32_t *__restrict__ keep_counts) {
    extern __shared__ __attribute__((aligned(16))) unsigned long long remv[];
    __shared__ unsigned long long sh_keep;
    __shared__ int sh_kept[64];  // rows (0..63) of the current row block that were kept
    const int seg = blockIdx.x;
    const long s0 = seg_off[seg];
    const long cnt = seg_off[seg + 1] - s0;
    const long nrb = (cnt + 63) / 64;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (long w = tid; w < nrb; w += 256) remv[w] = 0ull;
    __syncthreads();
    long nkeep = 0;  // maintained identically by all threads of wave 0
    for (long rb = 0; rb < nrb; ++rb) {
        if (wave == 0) {
            const long row = rb * 64 + lane;
            const bool valid = row < cnt;
            const unsigned long long d = valid ? mask[(size_t) (s0 + row) * CW + rb] : 0ull;
            unsigned long long R = remv[rb];
            unsigned long long kb = 0ull;
            const int nrow = (int) ((cnt - rb * 64) < 64 ? (cnt - rb * 64) : 64);
            for (int t = 0; t < nrow; ++t) {
                const unsigned long long dt = __shfl(d, t, 64);
                if (!((R >> t) & 1ull)) {
                    kb |= (1ull << t);
                    R |= dt;
                }
            }
            if (valid && ((kb >> lane) & 1ull)) {
                const long posn = nkeep + __popcll(kb & ((1ull << lane) - 1ull));
                keep[s0 + posn] = (int64_t) vals[s0 + row];  // index into the concatenated input
            }
            nkeep += __popcll(kb);
            if ((kb >> lane) & 1ull) sh_kept[__popcll(kb & ((1ull << lane) - 1ull))] = lane;
            if (lane == 0) sh_keep = kb;
        }
        __syncthreads();
        // OR the suppression rows of the kept boxes into the pending words: (kept row, word) pairs are spread over
        // the block so that all mask loads are independent (a per-thread loop over the kept rows serialised up to 64
        // HBM round trips per row block: 276 us per call at 1500 boxes per image)
        const int nk = __popcll(sh_keep);
        const long nw = nrb - rb - 1;
        for (long idx = tid; idx < nk * nw; idx += 256) {
            const int t = sh_kept[idx / nw];
            const long w = rb + 1 + idx % nw;
            const unsigned long long m = mask[(size_t) (s0 + rb * 64 + t) * CW + w];
            if (m) atomicOr(&remv[w], m);
        }
        __syncthreads();
    }
    if (tid == 0) keep_counts[seg] = (int32_t) nkeep;
}

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

struct NmsLayout {
    size_t keys_in, keys_out, vals_in, vals_out, sboxes, mask, sort_tmp, sort_tmp_bytes, total;
    long CW;
};

NmsLayout nms_layout(int64_t P, int64_t max_seg, int nseg) {
    NmsLayout L{};
    L.CW = (max_seg + 63) / 64;
    if (L.CW < 1) L.CW = 1;
    size_t o = 0;
    const size_t Pn = P > 0 ? (size_t) P : 1;
    L.keys_in = o; o = align_up(o + Pn * 8, 256);
    L.keys_out = o; o = align_up(o + Pn * 8, 256);
    L.vals_in = o; o = align_up(o + Pn * 4, 256);
    L.vals_out = o; o = align_up(o + Pn * 4, 256);
    L.sboxes = o; o = align_up(o + Pn * 16, 256);
    L.mask = o; o = align_up(o + Pn * (size_t) L.CW * 8, 256);
    size_t tmp = 0;
    (void) rocprim::radix_sort_pairs(nullptr, tmp, (unsigned long long *) nullptr, (unsigned long long *) nullptr,
                                     (unsigned int *) nullptr, (unsigned int *) nullptr, Pn, 0, 64, (hipStream_t) 0);
    L.sort_tmp = o; L.sort_tmp_bytes = tmp; o = align_up(o + tmp, 256);
    L.total = o;
    return L;
}

}  // namespace

// =========================================================================================================
// C ABI
// =========================================================================================================
extern "C" {

int64_t cpn_compact_workspace_bytes(int32_t N, int32_t h, int32_t w) {
    const long bpi = ((long) h * w + CBLK - 1) / CBLK;
    return (int64_t) (2 * N * bpi + 16) * 4;
}

int cpn_compact(const float *scores, int32_t N, int32_t h, int32_t w, float thresh, int32_t *indices, int32_t *counts,
                void *workspace, void *stream) {
    if (N <= 0 || h <= 0 || w <= 0) return cpn::fail(CPN_E_INVALID, "cpn_compact: bad shape");
    if ((int64_t) N * h * w >= (1ll << 31))
        return cpn::fail(CPN_E_UNSUPPORTED, "cpn_compact: N*h*w must stay below 2^31 (int32 proposal indices)");
    hipStream_t st = (hipStream_t) stream;
    const int hw = h * w;
    const int bpi = (hw + CBLK - 1) / CBLK;
    int *block_counts = (int *) workspace;
    int *block_offsets = block_counts + (size_t) N * bpi;
    hipLaunchKernelGGL(compact_count_kernel, dim3(bpi, N), dim3(CBLK), 0, st, scores, hw, thresh, block_counts);
    hipLaunchKernelGGL(compact_scan_kernel, dim3(1), dim3(1024), 0, st, block_counts, N * bpi, bpi, N, block_offsets,
                       counts);
    hipLaunchKernelGGL(compact_write_kernel, dim3(bpi, N), dim3(CBLK), 0, st, scores, hw, thresh, block_offsets,
                       indices);
    return cpn::check_hip(hipGetLastError(), "cpn_compact");
}

static int decode_impl(bool gathered, const int32_t *indices, int32_t P, const float *scores, const float *locations,
                       const float *fourier, const float *refinement, int32_t N, int32_t h, int32_t w, int32_t H, int32_t W,
                       int32_t order_total, int32_t order, int32_t samples, int32_t iterations, const float *cos_table,
                       const float *sin_table, const float *offsets, float *contours, float *proposals, float *boxes,
                       float *out_scores, float *out_locations, float *out_fourier, int32_t *batch_index, int32_t buckets,
                       const int32_t *bucket_index, const float *bucket_weight, void *stream) {
    if (P < 0 || order < 1 || order > order_total || order * 4 > MAX_COEF || samples < 1)
        return cpn::fail(CPN_E_INVALID, "cpn_decode: bad arguments (need 1 <= order <= min(order_total, 64))");
    if (buckets > 1 && refinement && iterations > 0 && (!bucket_index || !bucket_weight))
        return cpn::fail(CPN_E_INVALID, "cpn_decode: refinement_buckets > 1 needs the bucket tables");
    if (N <= 0 || h <= 0 || w <= 0 || (int64_t) N * h * w >= (1ll << 31))
        return cpn::fail(CPN_E_UNSUPPORTED, "cpn_decode: N*h*w must be positive and below 2^31 (int32 proposal indices)");
    if (P == 0) return 0;
    DecodeArgs a{indices, P, scores, locations, fourier, refinement, N, h, w, H, W, order_total, order, samples,
                 iterations, cos_table, sin_table, offsets, contours, proposals, boxes, out_scores, out_locations,
                 out_fourier, batch_index, Buckets{buckets, bucket_index, bucket_weight, samples}};
    if (gathered)
        hipLaunchKernelGGL(decode_kernel<true>, dim3((P + DWAVES - 1) / DWAVES), dim3(DWAVES * WAVE), 0, (hipStream_t) stream, a);
    else
        hipLaunchKernelGGL(decode_kernel<false>, dim3((P + DWAVES - 1) / DWAVES), dim3(DWAVES * WAVE), 0, (hipStream_t) stream, a);
    return cpn::check_hip(hipGetLastError(), "cpn_decode");
}

int cpn_decode(const int32_t *indices, int32_t P, const float *scores, const float *locations, const float *fourier,
               const float *refinement, int32_t N, int32_t h, int32_t w, int32_t H, int32_t W, int32_t order_total,
               int32_t order, int32_t samples, int32_t iterations, const float *cos_table, const float *sin_table,
               const float *offsets, float *contours, float *proposals, float *boxes, float *out_scores,
               float *out_locations, float *out_fourier, int32_t *batch_index, int32_t buckets,
               const int32_t *bucket_index, const float *bucket_weight, void *stream) {
    return decode_impl(false, indices, P, scores, locations, fourier, refinement, N, h, w, H, W, order_total, order, samples,
                       iterations, cos_table, sin_table, offsets, contours, proposals, boxes, out_scores, out_locations,
                       out_fourier, batch_index, buckets, bucket_index, bucket_weight, stream);
}

int cpn_decode_gathered(const int32_t *indices, int32_t P, const float *scores, const float *locations,
                        const float *fourier, const float *refinement, int32_t N, int32_t h, int32_t w, int32_t H,
                        int32_t W, int32_t order_total, int32_t order, int32_t samples, int32_t iterations,
                        const float *cos_table, const float *sin_table, const float *offsets, float *contours,
                        float *proposals, float *boxes, float *out_scores, float *out_locations, float *out_fourier,
                        int32_t *batch_index, int32_t buckets, const int32_t *bucket_index, const float *bucket_weight,
                        void *stream) {
    return decode_impl(true, indices, P, scores, locations, fourier, refinement, N, h, w, H, W, order_total, order, samples,
                       iterations, cos_table, sin_table, offsets, contours, proposals, boxes, out_scores, out_locations,
                       out_fourier, batch_index, buckets, bucket_index, bucket_weight, stream);
}

int cpn_fouriers2contours(const float *fourier, const float *locations, int32_t P, int32_t order, int32_t samples,
                          const float *cos_table, const float *sin_table, float *contours, void *stream) {
    if (P < 0 || order < 1 || order * 4 > MAX_COEF || samples < 1)
        return cpn::fail(CPN_E_INVALID, "cpn_fouriers2contours: bad arguments");
    if (P == 0) return 0;
    hipLaunchKernelGGL(f2c_kernel, dim3((P + DWAVES - 1) / DWAVES), dim3(DWAVES * WAVE), 0, (hipStream_t) stream,
                       fourier, locations, P, order, samples, cos_table, sin_table, contours);
    return cpn::check_hip(hipGetLastError(), "cpn_fouriers2contours");
}

int cpn_local_refinement(float *contours, const int32_t *batch_index, int32_t P, int32_t samples,
                         const float *refinement, int32_t N, int32_t H, int32_t W, int32_t iterations,
                         int32_t buckets, const int32_t *bucket_index, const float *bucket_weight, void *stream) {
    (void) N;
    if (P <= 0 || iterations <= 0) return 0;
    if (buckets > 1 && (!bucket_index || !bucket_weight))
        return cpn::fail(CPN_E_INVALID, "cpn_local_refinement: refinement_buckets > 1 needs the bucket tables");
    const long total = (long) P * samples;
    hipLaunchKernelGGL(refine_kernel, dim3((unsigned) ((total + 255) / 256)), dim3(256), 0, (hipStream_t) stream,
                       contours, batch_index, total, samples, refinement, H, W, iterations,
                       Buckets{buckets, bucket_index, bucket_weight, samples});
    return cpn::check_hip(hipGetLastError(), "cpn_local_refinement");
}

int cpn_class_scores(const float *logits, int32_t N, int32_t C, int32_t h, int32_t w, const float *lower,
                     const float *upper, float *probs, float *selected, int32_t *classes, float *foreground,
                     void *stream) {
    if (N < 0 || C < 1 || !logits || !selected || !classes || !foreground)
        return cpn::fail(CPN_E_INVALID, "cpn_class_scores: bad arguments");
    const long total = (long) N * h * w;
    if (total == 0) return 0;
    hipLaunchKernelGGL(class_scores_kernel, dim3((unsigned) ((total + 255) / 256)), dim3(256), 0, (hipStream_t) stream,
                       logits, N, C, h * w, lower, upper, probs, selected, classes, foreground);
    return cpn::check_hip(hipGetLastError(), "cpn_class_scores");
}

int cpn_certainty_mask(const float *scores, const float *uncertainty, int32_t N, int32_t C, int32_t h, int32_t w,
                       float limit, float *out, void *stream) {
    if (N < 0 || C < 1 || !scores || !uncertainty || !out)
        return cpn::fail(CPN_E_INVALID, "cpn_certainty_mask: bad arguments");
    const long total = (long) N * h * w;
    if (total == 0) return 0;
    hipLaunchKernelGGL(certainty_kernel, dim3((unsigned) ((total + 255) / 256)), dim3(256), 0, (hipStream_t) stream,
                       scores, uncertainty, N, C, h * w, limit, out);
    return cpn::check_hip(hipGetLastError(), "cpn_certainty_mask");
}

int cpn_box_votes(const float *boxes, int64_t P, float thresh, float *votes, void *stream) {
    if (P < 0) return cpn::fail(CPN_E_INVALID, "cpn_box_votes: bad arguments");
    if (P == 0) return 0;
    hipLaunchKernelGGL(box_votes_kernel, dim3((unsigned) ((P + 3) / 4)), dim3(256), 0, (hipStream_t) stream, boxes,
                       (long) P, thresh, votes);
    return cpn::check_hip(hipGetLastError(), "cpn_box_votes");
}

int cpn_gather_channels(const float *map, const int32_t *indices, int64_t P, int32_t C, int32_t h, int32_t w,
                        float *out, void *stream) {
    if (P < 0 || C < 1) return cpn::fail(CPN_E_INVALID, "cpn_gather_channels: bad arguments");
    if (P == 0) return 0;
    const long total = (long) P * C;
    hipLaunchKernelGGL(gather_channels_kernel, dim3((unsigned) ((total + 255) / 256)), dim3(256), 0,
                       (hipStream_t) stream, map, indices, (long) P, C, h * w, out);
    return cpn::check_hip(hipGetLastError(), "cpn_gather_channels");
}

int cpn_border_keep(const float *contours, int64_t P, int32_t samples, float off_x, float off_y, float h, float w,
                    float pad, int32_t sides, uint8_t *keep, void *stream) {
    if (P <= 0) return 0;
    hipLaunchKernelGGL(border_kernel, dim3((unsigned) ((P + 3) / 4)), dim3(256), 0, (hipStream_t) stream, contours,
                       (long) P, samples, off_x, off_y, h, w, pad, sides, keep);
    return cpn::check_hip(hipGetLastError(), "cpn_border_keep");
}

int cpn_resize_f32(const float *src, float *dst, int64_t planes, int32_t Hin, int32_t Win, int32_t Hout, int32_t Wout,
                   int32_t mode, void *stream) {
    if (!src || !dst || planes < 0 || Hin <= 0 || Win <= 0 || Hout <= 0 || Wout <= 0 || (mode != 0 && mode != 1))
        return cpn::fail(CPN_E_INVALID, "cpn_resize_f32: bad arguments (mode 0 bilinear | 1 bicubic)");
    const long total = (long) planes * Hout * Wout;
    if (total == 0) return 0;
    hipLaunchKernelGGL(resize_f32_kernel, dim3((unsigned) ((total + 255) / 256)), dim3(256), 0, (hipStream_t) stream,
                       src, dst, (long) planes, Hin, Win, Hout, Wout, mode);
    return cpn::check_hip(hipGetLastError(), "cpn_resize_f32");
}

int cpn_resize_bilinear_f32(const float *src, float *dst, int64_t planes, int32_t Hin, int32_t Win, int32_t Hout,
                            int32_t Wout, void *stream) {
    return cpn_resize_f32(src, dst, planes, Hin, Win, Hout, Wout, 0, stream);
}

int cpn_border_keep_batched(const float *contours, int64_t P, int32_t samples, const int32_t *image_index,
                            const int32_t *sides, const float *offsets, int32_t n_images, float h, float w, float pad,
                            uint8_t *keep, void *stream) {
    if (P < 0 || samples < 1 || n_images < 1) return cpn::fail(CPN_E_INVALID, "cpn_border_keep_batched: bad arguments");
    if (P == 0) return 0;
    if (!contours || !image_index || !sides || !offsets || !keep)
        return cpn::fail(CPN_E_INVALID, "cpn_border_keep_batched: null pointer");
    hipLaunchKernelGGL(border_batched_kernel, dim3((unsigned) ((P + 3) / 4)), dim3(256), 0, (hipStream_t) stream,
                       contours, (long) P, samples, image_index, sides, offsets, h, w, pad, keep);
    return cpn::check_hip(hipGetLastError(), "cpn_border_keep_batched");
}

int64_t cpn_nms_workspace_bytes(int64_t P, int64_t max_segment, int32_t nseg) {
    return (int64_t) nms_layout(P, max_segment, nseg).total;
}

int cpn_nms(const float *boxes, const float *scores, int64_t P, const int64_t *seg_offsets_host,
            const int64_t *seg_offsets_dev, int32_t nseg, float thresh, int64_t *keep, int32_t *keep_counts,
            void *workspace, int64_t workspace_bytes, void *stream) {
    if (nseg < 1 || P < 0) return cpn::fail(CPN_E_INVALID, "cpn_nms: bad arguments");
    hipStream_t st = (hipStream_t) stream;
    int64_t max_seg = 0;
    for (int s = 0; s < nseg; ++s) {
        const int64_t c = seg_offsets_host[s + 1] - seg_offsets_host[s];
        if (c < 0) return cpn::fail(CPN_E_INVALID, "cpn_nms: segment offsets must be non-decreasing");
        if (c > max_seg) max_seg = c;
    }
    if (seg_offsets_host[0] != 0 || seg_offsets_host[nseg] != P)
        return cpn::fail(CPN_E_INVALID, "cpn_nms: segment offsets must span [0, P]");
    if (P == 0) return cpn::check_hip(hipMemsetAsync(keep_counts, 0, sizeof(int32_t) * nseg, st), "cpn_nms");
    const NmsLayout L = nms_layout(P, max_seg, nseg);
    if ((int64_t) L.total > workspace_bytes) return cpn::fail(CPN_E_WORKSPACE, "cpn_nms: workspace too small");
    char *ws = (char *) workspace;
    auto *keys_in = (unsigned long long *) (ws + L.keys_in);
    auto *keys_out = (unsigned long long *) (ws + L.keys_out);
    auto *vals_in = (unsigned int *) (ws + L.vals_in);
    auto *vals_out = (unsigned int *) (ws + L.vals_out);
    auto *sboxes = (float4 *) (ws + L.sboxes);
    auto *mask = (unsigned long long *) (ws + L.mask);
    const unsigned blocks = (unsigned) ((P + 255) / 256);
    hipLaunchKernelGGL(nms_keys_kernel, dim3(blocks), dim3(256), 0, st, scores, (long) P, seg_offsets_dev, nseg,
                       keys_in, vals_in);
    int seg_bits = 0;
    while ((1 << seg_bits) < nseg) ++seg_bits;
    size_t tmp = L.sort_tmp_bytes;
    hipError_t e = rocprim::radix_sort_pairs(ws + L.sort_tmp, tmp, keys_in, keys_out, vals_in, vals_out, (size_t) P, 0,
                                             (unsigned) (32 + seg_bits), st);
    if (e != hipSuccess) return cpn::check_hip(e, "cpn_nms: radix sort");
    hipLaunchKernelGGL(nms_gather_kernel, dim3(blocks), dim3(256), 0, st, boxes, vals_out, (long) P, sboxes);
    const unsigned nb = (unsigned) ((max_seg + 63) / 64);
    hipLaunchKernelGGL(nms_mask_kernel, dim3(nb, nb, nseg), dim3(64), 0, st, sboxes, seg_offsets_dev, thresh, L.CW,
                       mask);
    const size_t lds = (size_t) nb * 8;
    if (lds > 150 * 1024) return cpn::fail(CPN_E_UNSUPPORTED, "cpn_nms: segment larger than 1.2M boxes");
    if (lds > 48 * 1024) {
        e = hipFuncSetAttribute((const void *) nms_scan_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
        if (e != hipSuccess) return cpn::check_hip(e, "cpn_nms: lds attribute");
    }
    hipLaunchKernelGGL(nms_scan_kernel, dim3(nseg), dim3(256), lds, st, mask, vals_out, seg_offsets_dev, L.CW, keep,
                       keep_counts);
    return cpn::check_hip(hipGetLastError(), "cpn_nms");
}

}  // extern "C"
