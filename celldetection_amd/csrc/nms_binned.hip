// Spatially binned greedy NMS for slide-scale detection sets (10^5 .. 10^7 boxes), gfx950.
//
// Replaces the ONE global `torchvision.ops.nms` over all detections of a slide in the reference's tile loop
// (celldetection_scripts/cpn_inference.py:405-408; multi-model final NMS :426) where the dense 64x64 bit-mask
// formulation of csrc/decode_nms.hip needs P * ceil(P/64) * 8 bytes (77 k boxes -> 742 MB, 10^6 -> 125 GB).
//
// Result: the SAME keep list (indices in stable descending-score order) as the dense path / torchvision, because
// greedy NMS only ever relates boxes with IoU > thr >= 0, i.e. boxes that intersect:
//   1. rank the boxes by (score desc, index asc) with the same keys as the dense path (stable radix sort);
//   2. bin the ranked boxes by the grid cell of their centre; the cell edge is the largest box extent, or -- when
//      that is more than 4x the extent below which all but at most MAX_BIG boxes lie (an extent histogram in 1/8-octave
//      buckets) -- that smaller extent, so two intersecting NORMAL boxes always sit
//      in the same or in adjacent cells (3x3 neighbourhood).  The few OVERSIZED boxes (a tile-sized outlier on a slide
//      of cells would otherwise collapse the grid to a handful of cells and make step 3 quadratic) are kept in a list:
//      every box tests that list directly, and an oversized box scans every cell its extent reaches;
//   3. per box: the list of its *suppressor candidates* = higher-ranked boxes of the neighbourhood with IoU > thr
//      (same fp32 expression, same operation order as the dense mask kernel: count pass -> scan -> fill pass);
//   4. resolve the greedy recurrence keep[r] = !any(keep[q], q in suppressors(r)) as a monotone fixed point:
//      a box is decided as soon as one suppressor is known kept (-> removed) or all are known removed (-> kept);
//      rank 0 has no suppressors, every sweep decides at least the lowest undecided rank, and the fixed point IS the
//      greedy solution (unique: the recurrence is well-founded on rank);
//   5. compact the kept ranks in rank order.
// Memory: O(P + E) with E = number of (box, suppressor candidate) pairs -- ~10 per box for cell detections.
#include <cstring>
#include <hip/hip_runtime.h>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>

#include <stdint.h>

#include "../../include/cpn_hip.h"
#include "cpn_error.h"

namespace {

typedef unsigned long long u64;

struct GridParams {       // device-resident, written by grid_setup_kernel
    float minx, miny, inv_cell, cell;
    float big_thr;        // boxes whose larger extent exceeds this are "oversized" (kept in the big list)
    int gw, gh;
};

struct Stats {            // order-preserving int encodings of floats (atomicMin / atomicMax)
    int min_cx, min_cy, max_cx, max_cy, max_w, max_h;
};

__device__ __forceinline__ int f2ord(float f) {
    const int i = __float_as_int(f);
    return i >= 0 ? i : i ^ 0x7fffffff;
}
__device__ __forceinline__ float ord2f(int i) { return __int_as_float(i >= 0 ? i : i ^ 0x7fffffff); }

constexpr int MAX_CELLS = 1 << 22;
constexpr int MAX_BIG = 256;         // oversized boxes handled through the list (every box tests all of them)
constexpr int EXT_BUCKETS = 2048;    // extent histogram: float bits >> 20 (8 exponent bits + 3 mantissa bits)

__device__ __forceinline__ float extent_of(const float4 b) {  // larger box extent; NaN / inf -> 0 (never oversized)
    const float w = fabsf(b.z - b.x), h = fabsf(b.w - b.y);
    const float e = fmaxf(w - w == 0.f ? w : 0.f, h - h == 0.f ? h : 0.f);
    return e;
}

__global__ __launch_bounds__(256) void bn_keys_kernel(const float *__restrict__ scores, long P,
                                                     unsigned int *__restrict__ keys, unsigned int *__restrict__ vals) {
    const long i = blockIdx.x * 256l + threadIdx.x;
    if (i >= P) return;
    unsigned int u = __float_as_uint(scores[i]);
    if ((u & 0x7fffffffu) > 0x7f800000u) u = 0x7fc00000u;  // NaN sorts first in a descending torch.sort
    else if (u == 0x80000000u) u = 0u;                       // -0 == +0
    const unsigned int asc = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
    keys[i] = ~asc;
    vals[i] = (unsigned int) i;
}

__global__ __launch_bounds__(256) void bn_gather_stats_kernel(const float *__restrict__ boxes,
                                                             const unsigned int *__restrict__ vals, long P,
                                                             float4 *__restrict__ sboxes, Stats *__restrict__ st,
                                                             unsigned int *__restrict__ ext_hist) {
    const long i = blockIdx.x * 256l + threadIdx.x;
    if (i >= P) return;
    const float4 b = ((const float4 *) boxes)[vals[i]];
    sboxes[i] = b;
    atomicAdd(ext_hist + (__float_as_uint(extent_of(b)) >> 20), 1u);
    const float cx = 0.5f * (b.x + b.z), cy = 0.5f * (b.y + b.w), w = b.z - b.x, h = b.w - b.y;
    if (cx - cx == 0.f) { atomicMin(&st->min_cx, f2ord(cx)); atomicMax(&st->max_cx, f2ord(cx)); }  // finite only
    if (cy - cy == 0.f) { atomicMin(&st->min_cy, f2ord(cy)); atomicMax(&st->max_cy, f2ord(cy)); }
    if (w - w == 0.f) atomicMax(&st->max_w, f2ord(fabsf(w)));
    if (h - h == 0.f) atomicMax(&st->max_h, f2ord(fabsf(h)));
}

__global__ void bn_stats_init_kernel(Stats *st) {
    st->min_cx = st->min_cy = 0x7fffffff;
    st->max_cx = st->max_cy = (int) 0x80000000;
    st->max_w = st->max_h = f2ord(0.f);
}

__global__ void bn_grid_setup_kernel(const Stats *__restrict__ st, const unsigned int *__restrict__ ext_hist,
                                     GridParams *__restrict__ g) {
    float minx = ord2f(st->min_cx), miny = ord2f(st->min_cy), maxx = ord2f(st->max_cx), maxy = ord2f(st->max_cy);
    if (!(minx <= maxx)) { minx = 0.f; maxx = 0.f; }  // no finite box at all
    if (!(miny <= maxy)) { miny = 0.f; maxy = 0.f; }
    // cell edge >= the extent of all but at most MAX_BIG boxes: the upper edge of the first histogram bucket (from the
    // top) at which the number of larger boxes would exceed MAX_BIG; (+0.1 %: rounding of the centre / division never
    // splits intersecting normal boxes over non-adjacent cells); doubled until the grid fits MAX_CELLS
    unsigned int above = 0;
    int bstar = EXT_BUCKETS - 1;
    for (; bstar > 0; --bstar) {
        if (above + ext_hist[bstar] > (unsigned int) MAX_BIG) break;
        above += ext_hist[bstar];
    }
    // every box in buckets > bstar is oversized; bucket bstar's upper edge bounds the normal ones
    const float max_ext = fmaxf(ord2f(st->max_w), ord2f(st->max_h));
    float thr = __uint_as_float((unsigned int) (bstar + 1) << 20);
    // the list only pays when the largest boxes are real outliers: a grid four times coarser per axis still has 16x fewer
    // boxes per cell than no grid, while every listed box costs one IoU test per box of the set
    const bool use_list = max_ext > 4.f * thr;
    if (!use_list) thr = max_ext;
    float cell = fmaxf(thr * 1.001f, 1e-3f);
    long gw, gh;
    for (;;) {
        gw = (long) floorf((maxx - minx) / cell) + 1;
        gh = (long) floorf((maxy - miny) / cell) + 1;
        if (gw > 0 && gh > 0 && gw * gh <= MAX_CELLS) break;
        cell *= 2.f;
    }
    g->minx = minx; g->miny = miny; g->inv_cell = 1.f / cell; g->cell = cell;
    // oversized = larger extent > big_thr.  With the list: exactly the boxes of the histogram buckets above bstar (<= MAX_BIG
    // by construction; a grid that had to be coarsened raises the bound: fewer oversized boxes).  Without it NO box is
    // oversized: scaling the largest extent up and down again can land one ulp below it, which would have sent
    // every box that shares the maximum extent to the list
    g->big_thr = use_list ? fmaxf(thr, cell / 1.001f) : __uint_as_float(0x7f800000u);
    g->gw = (int) gw; g->gh = (int) gh;
}

__device__ __forceinline__ void cell_of(const float4 b, const GridParams &g, int &cx, int &cy) {
    const float fx = (0.5f * (b.x + b.z) - g.minx) * g.inv_cell, fy = (0.5f * (b.y + b.w) - g.miny) * g.inv_cell;
    cx = (fx >= 0.f) ? (int) fminf(fx, (float) (g.gw - 1)) : 0;  // NaN / -inf -> 0 (such a box overlaps nothing)
    cy = (fy >= 0.f) ? (int) fminf(fy, (float) (g.gh - 1)) : 0;
}

__global__ __launch_bounds__(256) void bn_cell_kernel(const float4 *__restrict__ sboxes, long P,
                                                     const GridParams *__restrict__ gp, unsigned int *__restrict__ cid,
                                                     unsigned int *__restrict__ rank, unsigned int *__restrict__ big,
                                                     unsigned int *__restrict__ nbig) {
    const long i = blockIdx.x * 256l + threadIdx.x;
    if (i >= P) return;
    const GridParams g = *gp;
    int cx, cy;
    const float4 b = sboxes[i];
    cell_of(b, g, cx, cy);
    cid[i] = (unsigned int) (cy * g.gw + cx);
    rank[i] = (unsigned int) i;
    if (extent_of(b) > g.big_thr) {  // at most MAX_BIG by construction of the threshold
        const unsigned int k = atomicAdd(nbig, 1u);
        if (k < (unsigned int) MAX_BIG) big[k] = (unsigned int) i;
    }
}

__global__ __launch_bounds__(256) void bn_cell_bounds_kernel(const unsigned int *__restrict__ scid, long P,
                                                            unsigned int *__restrict__ cbegin,
                                                            unsigned int *__restrict__ cend) {
    const long i = blockIdx.x * 256l + threadIdx.x;
    if (i >= P) return;
    const unsigned int c = scid[i];
    if (i == 0 || scid[i - 1] != c) cbegin[c] = (unsigned int) i;
    if (i == P - 1 || scid[i + 1] != c) cend[c] = (unsigned int) (i + 1);
}

// same fp32 expression and operation order as nms_mask_kernel (csrc/decode_nms.hip): `hi` is the higher-ranked box
__device__ __forceinline__ bool suppresses(const float4 hi, const float4 lo, float thr) {
    const float iarea = __fmul_rn(__fsub_rn(hi.z, hi.x), __fsub_rn(hi.w, hi.y));
    const float xx1 = fmaxf(hi.x, lo.x), yy1 = fmaxf(hi.y, lo.y);
    const float xx2 = fminf(hi.z, lo.z), yy2 = fminf(hi.w, lo.w);
    const float w = fmaxf(0.f, __fsub_rn(xx2, xx1)), h = fmaxf(0.f, __fsub_rn(yy2, yy1));
    const float inter = __fmul_rn(w, h);
    const float jarea = __fmul_rn(__fsub_rn(lo.z, lo.x), __fsub_rn(lo.w, lo.y));
    const float ovr = __fdiv_rn(inter, __fsub_rn(__fadd_rn(iarea, jarea), inter));
    return ovr > thr;
}

// FILL == false: deg[r] = number of suppressor candidates of rank r; FILL == true: write them to edges[off[r] ...]
template <bool FILL>
__global__ __launch_bounds__(256) void bn_edges_kernel(const float4 *__restrict__ sboxes, long P,
                                                      const GridParams *__restrict__ gp,
                                                      const unsigned int *__restrict__ srank,
                                                      const unsigned int *__restrict__ cbegin,
                                                      const unsigned int *__restrict__ cend, float thr,
                                                      u64 *__restrict__ deg, const u64 *__restrict__ off,
                                                      unsigned int *__restrict__ edges, u64 max_edges,
                                                      const unsigned int *__restrict__ big,
                                                      const unsigned int *__restrict__ nbig) {
    const long r = blockIdx.x * 256l + threadIdx.x;
    if (r >= P) return;
    const GridParams g = *gp;
    const float4 me = sboxes[r];
    int cx, cy;
    cell_of(me, g, cx, cy);
    u64 n = 0;
    const u64 base = FILL ? off[r] : 0;
    // normal box: the 3 x 3 neighbourhood of its centre cell; oversized box: every cell its extent (+ one cell: the other
    // box's half extent and the rounding margin) reaches.  Oversized candidates are skipped here and taken from the list.
    int x0 = cx - 1, x1 = cx + 1, y0 = cy - 1, y1 = cy + 1;
    if (extent_of(me) > g.big_thr) {
        int ax, ay, bx, by;
        cell_of(make_float4(me.x, me.y, me.x, me.y), g, ax, ay);
        cell_of(make_float4(me.z, me.w, me.z, me.w), g, bx, by);
        x0 = min(ax, bx) - 1; x1 = max(ax, bx) + 1; y0 = min(ay, by) - 1; y1 = max(ay, by) + 1;
    }
    for (int y = max(y0, 0); y <= min(y1, g.gh - 1); ++y)
        for (int x = max(x0, 0); x <= min(x1, g.gw - 1); ++x) {
            const unsigned int c = (unsigned int) (y * g.gw + x);
            const unsigned int e = cend[c];
            for (unsigned int k = cbegin[c]; k < e; ++k) {
                const unsigned int q = srank[k];
                if (q >= (unsigned int) r) break;  // ranks ascend within a cell
                const float4 hi = sboxes[q];
                if (extent_of(hi) > g.big_thr) continue;
                if (suppresses(hi, me, thr)) {
                    if (FILL && base + n < max_edges) edges[base + n] = q;
                    ++n;
                }
            }
        }
    const unsigned int nb = min(*nbig, (unsigned int) MAX_BIG);
    for (unsigned int k = 0; k < nb; ++k) {
        const unsigned int q = big[k];
        if (q >= (unsigned int) r) continue;
        if (suppresses(sboxes[q], me, thr)) {
            if (FILL && base + n < max_edges) edges[base + n] = q;
            ++n;
        }
    }
    if (!FILL) deg[r] = n;
}

// one sweep of the fixed point; state: 0 undecided, 1 kept, 2 removed.  undecided[0] counts what is left.
__global__ __launch_bounds__(256) void bn_resolve_kernel(long P, const u64 *__restrict__ off, const u64 *__restrict__ deg,
                                                        const unsigned int *__restrict__ edges,
                                                        unsigned char *__restrict__ state,
                                                        unsigned int *__restrict__ undecided) {
    const long r = blockIdx.x * 256l + threadIdx.x;
    if (r >= P) return;
    if (state[r] != 0) return;
    const u64 b = off[r], n = deg[r];
    bool all_removed = true, any_kept = false;
    for (u64 k = 0; k < n; ++k) {
        const unsigned char s = __builtin_nontemporal_load(state + edges[b + k]);
        if (s == 1) { any_kept = true; break; }
        if (s == 0) all_removed = false;
    }
    if (any_kept) state[r] = 2;
    else if (all_removed) state[r] = 1;
    else atomicAdd(undecided, 1u);
}

__global__ __launch_bounds__(256) void bn_flags_kernel(const unsigned char *__restrict__ state, long P,
                                                      u64 *__restrict__ flags) {
    const long r = blockIdx.x * 256l + threadIdx.x;
    if (r < P) flags[r] = state[r] == 1 ? 1ull : 0ull;
}

__global__ __launch_bounds__(256) void bn_emit_kernel(const unsigned char *__restrict__ state, const u64 *__restrict__ pos,
                                                     const unsigned int *__restrict__ vals, long P,
                                                     int64_t *__restrict__ keep, int64_t *__restrict__ keep_count) {
    const long r = blockIdx.x * 256l + threadIdx.x;
    if (r >= P) return;
    if (state[r] == 1) keep[pos[r]] = (int64_t) vals[r];
    if (r == P - 1) *keep_count = (int64_t) (pos[r] + (state[r] == 1 ? 1 : 0));
}

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

struct Layout {
    size_t keys_in, keys_out, vals_in, vals_out, sboxes, cid_in, cid_out, rank_in, rank_out, cbegin, cend, deg, off,
        state, stats, grid, counter, ext_hist, big, edges, tmp, tmp_bytes, total;
};

Layout layout(int64_t P, int64_t max_edges) {
    Layout L{};
    const size_t n = P > 0 ? (size_t) P : 1;
    size_t o = 0;
    auto take = [&](size_t bytes) { const size_t at = o; o = align_up(o + bytes, 256); return at; };
    L.keys_in = take(n * 4); L.keys_out = take(n * 4); L.vals_in = take(n * 4); L.vals_out = take(n * 4);
    L.sboxes = take(n * 16);
    L.cid_in = take(n * 4); L.cid_out = take(n * 4); L.rank_in = take(n * 4); L.rank_out = take(n * 4);
    L.cbegin = take((size_t) MAX_CELLS * 4); L.cend = take((size_t) MAX_CELLS * 4);
    L.deg = take(n * 8); L.off = take(n * 8);
    L.state = take(n);
    L.stats = take(sizeof(Stats)); L.grid = take(sizeof(GridParams)); L.counter = take(256);
    L.ext_hist = take((size_t) EXT_BUCKETS * 4); L.big = take((size_t) MAX_BIG * 4);
    L.edges = take((size_t) (max_edges > 0 ? max_edges : 1) * 4);
    size_t t1 = 0, t2 = 0;
    (void) rocprim::radix_sort_pairs(nullptr, t1, (unsigned int *) nullptr, (unsigned int *) nullptr,
                                     (unsigned int *) nullptr, (unsigned int *) nullptr, n, 0, 32, (hipStream_t) 0);
    (void) rocprim::exclusive_scan(nullptr, t2, (u64 *) nullptr, (u64 *) nullptr, (u64) 0, n, rocprim::plus<u64>(),
                                   (hipStream_t) 0);
    L.tmp_bytes = t1 > t2 ? t1 : t2;
    L.tmp = take(L.tmp_bytes);
    L.total = o;
    return L;
}

}  // namespace

extern "C" {

int64_t cpn_nms_binned_workspace_bytes(int64_t P, int64_t max_edges) { return (int64_t) layout(P, max_edges).total; }

int cpn_nms_binned(const float *boxes, const float *scores, int64_t P, float thresh, int64_t max_edges, int64_t *keep,
                   int64_t *keep_count_dev, int64_t *keep_count_host, int64_t *edges_needed, int32_t *sweeps,
                   void *workspace, int64_t workspace_bytes, void *stream) {
    if (P < 0 || max_edges < 0 || !keep_count_host) return cpn::fail(CPN_E_INVALID, "cpn_nms_binned: bad arguments");
    if (!(thresh >= 0.f))
        return cpn::fail(CPN_E_UNSUPPORTED, "cpn_nms_binned: needs iou_threshold >= 0 (a negative threshold lets "
                                            "disjoint boxes suppress each other: use the dense cpn_nms)");
    if (P >= (1ll << 32) - 1) return cpn::fail(CPN_E_UNSUPPORTED, "cpn_nms_binned: more than 2^32 - 2 boxes");
    hipStream_t st = (hipStream_t) stream;
    *keep_count_host = 0;
    if (edges_needed) *edges_needed = 0;
    if (sweeps) *sweeps = 0;
    if (P == 0) return keep_count_dev ? cpn::check_hip(hipMemsetAsync(keep_count_dev, 0, 8, st), "cpn_nms_binned") : 0;
    if (!boxes || !scores || !keep || !workspace) return cpn::fail(CPN_E_INVALID, "cpn_nms_binned: null pointer");
    const Layout L = layout(P, max_edges);
    if ((int64_t) L.total > workspace_bytes) return cpn::fail(CPN_E_WORKSPACE, "cpn_nms_binned: workspace too small");
    char *ws = (char *) workspace;
    auto *keys_in = (unsigned int *) (ws + L.keys_in), *keys_out = (unsigned int *) (ws + L.keys_out);
    auto *vals_in = (unsigned int *) (ws + L.vals_in), *vals_out = (unsigned int *) (ws + L.vals_out);
    auto *sboxes = (float4 *) (ws + L.sboxes);
    auto *cid_in = (unsigned int *) (ws + L.cid_in), *cid_out = (unsigned int *) (ws + L.cid_out);
    auto *rank_in = (unsigned int *) (ws + L.rank_in), *rank_out = (unsigned int *) (ws + L.rank_out);
    auto *cbegin = (unsigned int *) (ws + L.cbegin), *cend = (unsigned int *) (ws + L.cend);
    auto *deg = (u64 *) (ws + L.deg), *off = (u64 *) (ws + L.off);
    auto *state = (unsigned char *) (ws + L.state);
    auto *stats = (Stats *) (ws + L.stats);
    auto *grid = (GridParams *) (ws + L.grid);
    auto *counter = (unsigned int *) (ws + L.counter);  // [0] undecided boxes of a sweep, [2..3] keep count, [8] big-list length
    auto *nbig = counter + 8;
    auto *ext_hist = (unsigned int *) (ws + L.ext_hist), *big = (unsigned int *) (ws + L.big);
    auto *edges = (unsigned int *) (ws + L.edges);
    const unsigned blocks = (unsigned) ((P + 255) / 256);
    size_t tmp = L.tmp_bytes;
    hipError_t e;
    // 1. rank by (score desc, index asc)
    hipLaunchKernelGGL(bn_keys_kernel, dim3(blocks), dim3(256), 0, st, scores, (long) P, keys_in, vals_in);
    e = rocprim::radix_sort_pairs(ws + L.tmp, tmp, keys_in, keys_out, vals_in, vals_out, (size_t) P, 0, 32, st);
    if (e != hipSuccess) return cpn::check_hip(e, "cpn_nms_binned: score sort");
    // 2. ranked boxes + extent statistics -> grid
    hipLaunchKernelGGL(bn_stats_init_kernel, dim3(1), dim3(1), 0, st, stats);
    e = hipMemsetAsync(ext_hist, 0, (size_t) EXT_BUCKETS * 4, st);
    if (e == hipSuccess) e = hipMemsetAsync(counter, 0, 256, st);
    if (e != hipSuccess) return cpn::check_hip(e, "cpn_nms_binned: memset");
    hipLaunchKernelGGL(bn_gather_stats_kernel, dim3(blocks), dim3(256), 0, st, boxes, vals_out, (long) P, sboxes, stats,
                       ext_hist);
    hipLaunchKernelGGL(bn_grid_setup_kernel, dim3(1), dim3(1), 0, st, stats, ext_hist, grid);
    hipLaunchKernelGGL(bn_cell_kernel, dim3(blocks), dim3(256), 0, st, sboxes, (long) P, grid, cid_in, rank_in, big, nbig);
    tmp = L.tmp_bytes;
    e = rocprim::radix_sort_pairs(ws + L.tmp, tmp, cid_in, cid_out, rank_in, rank_out, (size_t) P, 0, 22, st);
    if (e != hipSuccess) return cpn::check_hip(e, "cpn_nms_binned: cell sort");
    e = hipMemsetAsync(cbegin, 0, (size_t) MAX_CELLS * 4, st);
    if (e == hipSuccess) e = hipMemsetAsync(cend, 0, (size_t) MAX_CELLS * 4, st);
    if (e != hipSuccess) return cpn::check_hip(e, "cpn_nms_binned: memset");
    hipLaunchKernelGGL(bn_cell_bounds_kernel, dim3(blocks), dim3(256), 0, st, cid_out, (long) P, cbegin, cend);
    // 3. suppressor candidates: count -> scan -> fill
    hipLaunchKernelGGL(bn_edges_kernel<false>, dim3(blocks), dim3(256), 0, st, sboxes, (long) P, grid, rank_out, cbegin,
                       cend, thresh, deg, (const u64 *) nullptr, (unsigned int *) nullptr, (u64) 0, big, nbig);
    tmp = L.tmp_bytes;
    e = rocprim::exclusive_scan(ws + L.tmp, tmp, deg, off, (u64) 0, (size_t) P, rocprim::plus<u64>(), st);
    if (e != hipSuccess) return cpn::check_hip(e, "cpn_nms_binned: scan");
    u64 last[2] = {0, 0};
    unsigned int nbig_host = 0;
    e = hipMemcpyAsync(&last[0], off + (P - 1), 8, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipMemcpyAsync(&last[1], deg + (P - 1), 8, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipMemcpyAsync(&nbig_host, nbig, 4, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e != hipSuccess) return cpn::check_hip(e, "cpn_nms_binned: edge count");
    if (nbig_host > (unsigned int) MAX_BIG)  // (cannot happen by construction of big_thr; a dropped box would be a wrong result)
        return cpn::fail(CPN_E_UNSUPPORTED, "cpn_nms_binned: more oversized boxes than the list holds");
    const u64 E = last[0] + last[1];
    if (edges_needed) *edges_needed = (int64_t) E;
    if (E > (u64) max_edges)
        return cpn::fail(CPN_E_WORKSPACE, "cpn_nms_binned: more suppressor candidates than max_edges (retry with the "
                                          "returned edges_needed)");
    hipLaunchKernelGGL(bn_edges_kernel<true>, dim3(blocks), dim3(256), 0, st, sboxes, (long) P, grid, rank_out, cbegin,
                       cend, thresh, deg, off, edges, (u64) max_edges, big, nbig);
    // 4. fixed point (every sweep decides at least the lowest undecided rank; the host looks at the counter every
    //    few sweeps)
    e = hipMemsetAsync(state, 0, (size_t) P, st);
    if (e != hipSuccess) return cpn::check_hip(e, "cpn_nms_binned: memset");
    int nsweeps = 0;
    // the host looks at the counter after 4, 8, 16, ... 64 sweeps (a sweep over decided boxes is cheap; a long suppression
    // chain needs one sweep per link and must not cost one host round trip per four of them)
    for (int CHUNK = 4;; CHUNK = CHUNK < 64 ? 2 * CHUNK : 64) {
        for (int k = 0; k < CHUNK; ++k) {
            e = hipMemsetAsync(counter, 0, 4, st);
            if (e != hipSuccess) return cpn::check_hip(e, "cpn_nms_binned: memset");
            hipLaunchKernelGGL(bn_resolve_kernel, dim3(blocks), dim3(256), 0, st, (long) P, off, deg, edges, state,
                               counter);
        }
        nsweeps += CHUNK;
        unsigned int left = 0;
        e = hipMemcpyAsync(&left, counter, 4, hipMemcpyDeviceToHost, st);
        if (e == hipSuccess) e = hipStreamSynchronize(st);
        if (e != hipSuccess) return cpn::check_hip(e, "cpn_nms_binned: resolve");
        if (left == 0) break;
        if (nsweeps > P + 128) return cpn::fail(CPN_E_INVALID, "cpn_nms_binned: fixed point did not converge");
    }
    if (sweeps) *sweeps = nsweeps;
    // 5. kept ranks, in rank order (deg / off are reused as flag / position arrays)
    hipLaunchKernelGGL(bn_flags_kernel, dim3(blocks), dim3(256), 0, st, state, (long) P, deg);
    tmp = L.tmp_bytes;
    e = rocprim::exclusive_scan(ws + L.tmp, tmp, deg, off, (u64) 0, (size_t) P, rocprim::plus<u64>(), st);
    if (e != hipSuccess) return cpn::check_hip(e, "cpn_nms_binned: scan");
    int64_t *kc = keep_count_dev ? keep_count_dev : (int64_t *) counter + 1;  // counter block is 256 B
    hipLaunchKernelGGL(bn_emit_kernel, dim3(blocks), dim3(256), 0, st, state, off, vals_out, (long) P, keep, kc);
    e = hipMemcpyAsync(keep_count_host, kc, 8, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e != hipSuccess) return cpn::check_hip(e, "cpn_nms_binned: result");
    return cpn::check_hip(hipGetLastError(), "cpn_nms_binned");
}

}  // extern "C"
