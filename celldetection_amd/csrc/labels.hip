// contours -> label image on the GPU (gfx950).
//
// Replaces celldetection.data.contours2labels (celldetection/data/cpn.py:292-358, called from
// celldetection_scripts/cpn_inference.py:811) (rounded, clip, gap, int32; optional ioa_thresh, data/cpn.py:341-350):
// every contour is rasterised as a filled polygon (render_contour -> cv2.drawContours(thickness=-1),
// data/cpn.py:245-255) and written with value (index + 1) into the FIRST channel whose gap-expanded bounding-box
// region holds no label yet (data/cpn.py:346-351); channels grow on demand.
//
// The reference is sequential over the contours.  Here the same result is produced in parallel ROUNDS: contour k
// only depends on the earlier contours j < k whose bounding box intersects k's gap-expanded box E_k (only they can
// have painted pixels inside E_k).  A contour is *ready* when all those predecessors are painted; ready contours of
// one round do not interact (if bbox_j meets E_k then j is a predecessor of k and k is not ready), so they read the
// canvas and paint concurrently.  Rounds = longest chain of the predecessor relation (~10-20 for cells in NMS order).
//
// Polygon fill rule = OpenCV's FillEdgeCollection + boundary lines for integer vertices, restated (cv2 is not
// available in the build image -- see oracle/labels_oracle.py, "parity unpinned"):
//   boundary: every edge drawn with the 8-connected LineIterator (left-to-right, err0 = dmaj - 2*dmin, diagonal step
//             iff err < 0):  minor(t) = ceil((2*dmin*t - dmaj) / (2*dmaj));
//   interior: scanline y takes the edges with y0 <= y < y1 (horizontal edges skipped), crossing x in 16.16 fixed point
//             x = x_top * 65536 + (y - y_top) * trunc(dx * 65536 / dy), rounded (x + 32768) >> 16; sorted crossings are
//             paired and the pixels between a pair (inclusive) are filled.
#include <hip/hip_runtime.h>

#include <stdint.h>

#include "../../include/cpn_hip.h"
#include "cpn_error.h"

namespace {

constexpr int MAX_S = 512;  // contour points supported per contour (LDS staging)

struct Box { int x0, y0, x1, y1; };  // inclusive pixel bounds

// 1. round (half to even) / clip / integer points + bounding boxes ------------------------------------------------------
__global__ __launch_bounds__(256) void lb_prepare_kernel(const float *__restrict__ contours, long K, int S, int H, int W,
                                                        int rounded, int clip, int32_t *__restrict__ pts,
                                                        int32_t *__restrict__ boxes) {
    const int lane = threadIdx.x & 63;
    const long k = blockIdx.x * 4l + (threadIdx.x >> 6);
    if (k >= K) return;
    float mnx = __builtin_inff(), mny = __builtin_inff(), mxx = -__builtin_inff(), mxy = -__builtin_inff();
    for (int s = lane; s < S; s += 64) {
        float x = contours[(k * S + s) * 2], y = contours[(k * S + s) * 2 + 1];
        if (rounded) { x = rintf(x); y = rintf(y); }                     // np.round: half to even
        if (clip) { x = fminf(fmaxf(x, 0.f), (float) (W - 1)); y = fminf(fmaxf(y, 0.f), (float) (H - 1)); }
        mnx = fminf(mnx, x); mny = fminf(mny, y); mxx = fmaxf(mxx, x); mxy = fmaxf(mxy, y);
        pts[(k * S + s) * 2] = (int32_t) x;                              // np.array(contour, dtype=np.int32): truncation
        pts[(k * S + s) * 2 + 1] = (int32_t) y;
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        mnx = fminf(mnx, __shfl_xor(mnx, d, 64)); mny = fminf(mny, __shfl_xor(mny, d, 64));
        mxx = fmaxf(mxx, __shfl_xor(mxx, d, 64)); mxy = fmaxf(mxy, __shfl_xor(mxy, d, 64));
    }
    if (lane == 0) {  // render_contour: floor(min) .. ceil(max) of the reference points
        boxes[k * 4 + 0] = (int32_t) floorf(mnx); boxes[k * 4 + 1] = (int32_t) floorf(mny);
        boxes[k * 4 + 2] = (int32_t) ceilf(mxx);  boxes[k * 4 + 3] = (int32_t) ceilf(mxy);
    }
}

// 2. ready test: no unpainted predecessor j < k with bbox_j meeting the gap-expanded box of k -------------------------
// contours are binned by the cell of their box centre (cell edge >= largest box extent + gap), so predecessors sit in
// the 3x3 neighbourhood; cell lists hold contour indices in ascending order
struct Grid { int gw, gh, cell; };

__device__ __forceinline__ void cell_of(const int32_t *b, const Grid &g, int &cx, int &cy) {
    cx = min(max(((b[0] + b[2]) >> 1) / g.cell, 0), g.gw - 1);
    cy = min(max(((b[1] + b[3]) >> 1) / g.cell, 0), g.gh - 1);
}

__global__ __launch_bounds__(256) void lb_cell_kernel(const int32_t *__restrict__ boxes, long K, Grid g,
                                                     unsigned int *__restrict__ cid, unsigned int *__restrict__ idx) {
    const long k = blockIdx.x * 256l + threadIdx.x;
    if (k >= K) return;
    int cx, cy;
    cell_of(boxes + k * 4, g, cx, cy);
    cid[k] = (unsigned int) (cy * g.gw + cx);
    idx[k] = (unsigned int) k;
}

__global__ __launch_bounds__(256) void lb_bounds_kernel(const unsigned int *__restrict__ scid, long K,
                                                       unsigned int *__restrict__ cbegin, unsigned int *__restrict__ cend) {
    const long i = blockIdx.x * 256l + threadIdx.x;
    if (i >= K) return;
    const unsigned int c = scid[i];
    if (i == 0 || scid[i - 1] != c) cbegin[c] = (unsigned int) i;
    if (i == K - 1 || scid[i + 1] != c) cend[c] = (unsigned int) (i + 1);
}

__global__ __launch_bounds__(256) void lb_ready_kernel(const int32_t *__restrict__ boxes, long K, Grid g, int gap,
                                                      const unsigned int *__restrict__ sidx,
                                                      const unsigned int *__restrict__ cbegin,
                                                      const unsigned int *__restrict__ cend,
                                                      const unsigned char *__restrict__ state,
                                                      unsigned char *__restrict__ ready) {
    const long k = blockIdx.x * 256l + threadIdx.x;
    if (k >= K) return;
    ready[k] = 0;
    if (state[k]) return;
    const int32_t *b = boxes + k * 4;
    const int ex0 = b[0] - gap, ey0 = b[1] - gap, ex1 = b[2] + gap, ey1 = b[3] + gap;
    int cx, cy;
    cell_of(b, g, cx, cy);
    for (int dy = -1; dy <= 1; ++dy) {
        const int y = cy + dy;
        if (y < 0 || y >= g.gh) continue;
        for (int dx = -1; dx <= 1; ++dx) {
            const int x = cx + dx;
            if (x < 0 || x >= g.gw) continue;
            const unsigned int c = (unsigned int) (y * g.gw + x);
            const unsigned int e = cend[c];
            for (unsigned int i = cbegin[c]; i < e; ++i) {
                const unsigned int j = sidx[i];
                if (j >= (unsigned int) k) break;  // indices ascend within a cell
                if (state[j]) continue;            // already painted
                const int32_t *o = boxes + (long) j * 4;
                if (o[0] <= ex1 && o[2] >= ex0 && o[1] <= ey1 && o[3] >= ey0) return;  // unpainted predecessor
            }
        }
    }
    ready[k] = 1;
}

// 3. paint: one workgroup (4 waves) per ready contour ------------------------------------------------------------------
__device__ __forceinline__ bool on_line(int px, int py, int ax, int ay, int bx, int by) {
    // 8-connected LineIterator from the left end point (left_to_right)
    if (bx < ax) { int t = ax; ax = bx; bx = t; t = ay; ay = by; by = t; }
    const int dx = bx - ax, dyv = by - ay, dy = dyv < 0 ? -dyv : dyv, sy = dyv < 0 ? -1 : 1;
    if (dy <= dx) {  // x major
        const int t = px - ax;
        if (t < 0 || t > dx) return false;
        const int m = dx == 0 ? 0 : (2 * dy * t - dx + 2 * dx - 1) / (2 * dx);  // ceil((2 dy t - dx) / (2 dx)), numerator > -2dx
        return py == ay + sy * m;
    }
    const int t = (py - ay) * sy;  // y major (x is the minor axis and grows: left to right)
    if (t < 0 || t > dy) return false;
    const int m = (2 * dx * t - dy + 2 * dy - 1) / (2 * dy);
    return px == ax + m;
}

// pixel (xx, yy) belongs to the filled polygon (boundary lines + scanline interior, see the header)
__device__ __forceinline__ bool lb_filled(int xx, int yy, const int *px, const int *py, int S) {
    bool set = false;
    int n_lt = 0, n_le = 0;
    for (int s = 0; s < S && !set; ++s) {
        const int ax = px[s], ay = py[s], bx = px[s + 1 == S ? 0 : s + 1], by = py[s + 1 == S ? 0 : s + 1];
        set = on_line(xx, yy, ax, ay, bx, by);
        if (ay == by) continue;  // horizontal edges take no part in the scanline fill
        const int ty = ay < by ? ay : by, tx = ay < by ? ax : bx, byy = ay < by ? by : ay;
        if (yy < ty || yy >= byy) continue;
        const long long ddx = ((long long) (bx - ax) * 65536ll) / (long long) (by - ay);  // C division: truncation
        const long long xf = (long long) tx * 65536ll + (long long) (yy - ty) * ddx;
        const int xr = (int) ((xf + 32768ll) >> 16);
        n_lt += xr < xx;
        n_le += xr <= xx;
    }
    return set || (n_lt & 1) || n_le > n_lt;
}

__global__ __launch_bounds__(256) void lb_paint_kernel(const int32_t *__restrict__ pts, const int32_t *__restrict__ boxes,
                                                      long K, int S, int H, int W, int gap,
                                                      const unsigned char *__restrict__ ready,
                                                      const unsigned int *__restrict__ ready_list, long n_ready,
                                                      int32_t *__restrict__ canvas, int C, unsigned char *__restrict__ state,
                                                      int32_t *__restrict__ channel, int32_t *__restrict__ counters,
                                                      int use_ioa, double ioa_thresh) {
    __shared__ int px[MAX_S], py[MAX_S];
    __shared__ int occupied;
    __shared__ int chosen;
    __shared__ unsigned int n_mask, n_covered;
    const long r = blockIdx.x;
    if (r >= n_ready) return;
    const long k = ready_list[r];
    (void) ready; (void) K;
    const int tid = threadIdx.x;
    for (int s = tid; s < S; s += 256) { px[s] = pts[(k * S + s) * 2]; py[s] = pts[(k * S + s) * 2 + 1]; }
    const int x0 = boxes[k * 4], y0 = boxes[k * 4 + 1], x1 = boxes[k * 4 + 2], y1 = boxes[k * 4 + 3];
    // region labels[max(0, ymin-gap) : gap+ymin+h, max(0, xmin-gap) : gap+xmin+w] (numpy slicing clips at the far end)
    const int ex0 = max(x0 - gap, 0), ey0 = max(y0 - gap, 0), ex1 = min(x1 + gap, W - 1), ey1 = min(y1 + gap, H - 1);
    const int ew = ex1 - ex0 + 1, eh = ey1 - ey0 + 1;
    if (tid == 0) { chosen = -1; n_mask = 0u; n_covered = 0u; }
    __syncthreads();
    if (use_ioa) {
        // data/cpn.py:341-350: skip the contour when more than ioa_thresh of its own filled area already carries a label of
        // ANY channel (the predecessors that can have painted there are resolved: that is what made this contour ready)
        const int w_ = x1 - x0 + 1, h_ = y1 - y0 + 1;
        unsigned int m = 0, cov = 0;
        for (int i = tid; i < w_ * h_; i += 256) {
            const int yy = y0 + i / w_, xx = x0 + i % w_;
            if (xx < 0 || xx >= W || yy < 0 || yy >= H || !lb_filled(xx, yy, px, py, S)) continue;
            ++m;
            bool any = false;
            for (int c = 0; c < C && !any; ++c) any = canvas[((size_t) c * H + yy) * W + xx] != 0;
            cov += any;
        }
        if (m) atomicAdd(&n_mask, m);
        if (cov) atomicAdd(&n_covered, cov);
        __syncthreads();
        // numpy: int / int -> float64 true division; 0 / 0 = nan compares False (the contour is kept)
        if ((double) n_covered / (double) n_mask > ioa_thresh) {
            if (tid == 0) {
                state[k] = 2;  // resolved without painting
                atomicAdd(&counters[0], 1);
            }
            return;
        }
    }
    for (int c = 0; c < C; ++c) {
        if (tid == 0) occupied = 0;
        __syncthreads();
        const int32_t *plane = canvas + (size_t) c * H * W;
        int any = 0;
        for (int i = tid; i < ew * eh && !any; i += 256) {
            const int yy = ey0 + i / ew, xx = ex0 + i % ew;
            if (plane[(size_t) yy * W + xx] != 0) any = 1;
        }
        if (any) occupied = 1;
        __syncthreads();
        if (!occupied) {
            if (tid == 0) chosen = c;
            __syncthreads();
            break;
        }
        __syncthreads();
    }
    const int c = chosen;
    if (c < 0) {  // every allocated channel is taken: the host grows the canvas and retries this contour
        if (tid == 0) atomicAdd(&counters[1], 1);
        return;
    }
    int32_t *plane = canvas + (size_t) c * H * W;
    const int w = x1 - x0 + 1, h = y1 - y0 + 1;
    const int32_t val = (int32_t) (k + 1);
    for (int i = tid; i < w * h; i += 256) {
        const int yy = y0 + i / w, xx = x0 + i % w;
        if (xx < 0 || xx >= W || yy < 0 || yy >= H) continue;
        if (lb_filled(xx, yy, px, py, S)) plane[(size_t) yy * W + xx] += val;  // labels[...] += a
    }
    if (tid == 0) {
        channel[k] = c;
        state[k] = 1;
        atomicAdd(&counters[0], 1);
    }
}

__global__ __launch_bounds__(256) void lb_compact_ready_kernel(const unsigned char *__restrict__ ready, long K,
                                                              unsigned int *__restrict__ list,
                                                              int32_t *__restrict__ counters) {
    const long k = blockIdx.x * 256l + threadIdx.x;
    if (k >= K || !ready[k]) return;
    list[atomicAdd(&counters[2], 1)] = (unsigned int) k;  // order within a round is irrelevant (no interaction)
}

}  // namespace

extern "C" {

int cpn_labels_prepare(const float *contours, int64_t K, int32_t S, int32_t H, int32_t W, int32_t rounded, int32_t clip,
                       int32_t *points, int32_t *boxes, void *stream) {
    if (K < 0 || S < 1 || S > MAX_S || H < 1 || W < 1)
        return cpn::fail(CPN_E_INVALID, "cpn_labels_prepare: bad arguments (1 <= points per contour <= 512)");
    if (K == 0) return 0;
    hipLaunchKernelGGL(lb_prepare_kernel, dim3((unsigned) ((K + 3) / 4)), dim3(256), 0, (hipStream_t) stream, contours,
                       (long) K, S, H, W, rounded, clip, points, boxes);
    return cpn::check_hip(hipGetLastError(), "cpn_labels_prepare");
}

int cpn_labels_bin(const int32_t *boxes, int64_t K, int32_t grid_w, int32_t grid_h, int32_t cell, uint32_t *cell_id,
                   uint32_t *index, void *stream) {
    if (K <= 0) return 0;
    hipLaunchKernelGGL(lb_cell_kernel, dim3((unsigned) ((K + 255) / 256)), dim3(256), 0, (hipStream_t) stream, boxes,
                       (long) K, Grid{grid_w, grid_h, cell}, cell_id, index);
    return cpn::check_hip(hipGetLastError(), "cpn_labels_bin");
}

int cpn_labels_cell_bounds(const uint32_t *sorted_cell_id, int64_t K, uint32_t *cell_begin, uint32_t *cell_end,
                           void *stream) {
    if (K <= 0) return 0;
    hipLaunchKernelGGL(lb_bounds_kernel, dim3((unsigned) ((K + 255) / 256)), dim3(256), 0, (hipStream_t) stream,
                       sorted_cell_id, (long) K, cell_begin, cell_end);
    return cpn::check_hip(hipGetLastError(), "cpn_labels_cell_bounds");
}

int cpn_labels_round(const int32_t *points, const int32_t *boxes, int64_t K, int32_t S, int32_t H, int32_t W,
                     int32_t gap, int32_t grid_w, int32_t grid_h, int32_t cell, const uint32_t *sorted_index,
                     const uint32_t *cell_begin, const uint32_t *cell_end, int32_t *canvas, int32_t channels,
                     uint8_t *state, uint8_t *ready, uint32_t *ready_list, int32_t *channel, int32_t *counters,
                     int32_t *counters_host, int32_t use_ioa, double ioa_thresh, void *stream) {
    if (K <= 0) return 0;
    if (S < 1 || S > MAX_S || channels < 1 || !counters_host)
        return cpn::fail(CPN_E_INVALID, "cpn_labels_round: bad arguments");
    hipStream_t st = (hipStream_t) stream;
    const Grid g{grid_w, grid_h, cell};
    const unsigned blocks = (unsigned) ((K + 255) / 256);
    hipError_t e = hipMemsetAsync(counters, 0, 3 * sizeof(int32_t), st);
    if (e != hipSuccess) return cpn::check_hip(e, "cpn_labels_round: memset");
    hipLaunchKernelGGL(lb_ready_kernel, dim3(blocks), dim3(256), 0, st, boxes, (long) K, g, gap, sorted_index, cell_begin,
                       cell_end, state, ready);
    hipLaunchKernelGGL(lb_compact_ready_kernel, dim3(blocks), dim3(256), 0, st, ready, (long) K, ready_list, counters);
    int32_t n_ready = 0;
    e = hipMemcpyAsync(&n_ready, counters + 2, 4, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e != hipSuccess) return cpn::check_hip(e, "cpn_labels_round: ready count");
    if (n_ready > 0)
        hipLaunchKernelGGL(lb_paint_kernel, dim3((unsigned) n_ready), dim3(256), 0, st, points, boxes, (long) K, S, H, W,
                           gap, ready, ready_list, (long) n_ready, canvas, channels, state, channel, counters, (int) use_ioa,
                           ioa_thresh);
    e = hipMemcpyAsync(counters_host, counters, 3 * sizeof(int32_t), hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e != hipSuccess) return cpn::check_hip(e, "cpn_labels_round: counters");
    return cpn::check_hip(hipGetLastError(), "cpn_labels_round");
}

}  // extern "C"
