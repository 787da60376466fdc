// C ABI + native graph executor of the CPN conv stack (see include/cpn_hip.h).
// The executor owns no device memory: activations live in a caller-provided arena whose layout is planned once
// per input shape with a liveness-based first-fit allocator (static workspace planning instead of a caching
// allocator; 288 GB of HBM3E make arena reuse a locality optimisation, not a necessity).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <tuple>
#include <vector>

#include "../../include/cpn_hip.h"
#include "cpn_error.h"
#include "cpn_kernels.h"

namespace cpn {

static thread_local std::string g_last_error;

int fail(int code, const char *msg) {
    g_last_error = msg;
    return code;
}
int check_hip(hipError_t e, const char *where) {
    if (e == hipSuccess) return 0;
    g_last_error = std::string(where) + ": " + hipGetErrorString(e);
    return (int) e;
}

struct ShapePlan {
    std::vector<int64_t> offsets;  // per tensor (arena byte offsets)
    std::vector<int> th, tw;       // per tensor spatial size for this input size (propagated op by op: any H x W)
    std::vector<char> skip;        // per op: not executed at this input size (sub-pixel triples: HEAD or PHASE + LATERAL)
    std::vector<int> ring;         // per op: bilinear resize ops that write only a border ring of their output (0: whole map)
    int out_h[CPN_NUM_OUTPUTS], out_w[CPN_NUM_OUTPUTS];  // sizes of the external fp32 outputs (0 = absent)
    int64_t total = 0;
    int64_t max_elems = 0;         // largest tensor of the graph, elements per image
    int error = 0;                 // CPN_E_* when the graph cannot run at this input size
    std::string message;
};

}  // namespace cpn

struct cpn_plan {
    std::vector<cpn_tensor_desc> tensors;
    std::vector<cpn_op_desc> ops;
    const unsigned char *weights = nullptr;
    size_t weight_bytes = 0;
    const float *bias = nullptr;
    size_t bias_count = 0;
    int precision = 0;  // CPN_PRECISION_BF16 / CPN_PRECISION_F32 / CPN_PRECISION_FP8
    std::map<std::tuple<int, int, int, int, int>, cpn::ShapePlan> shape_plans;  // guarded by shape_mutex (std::map nodes are
    std::mutex shape_mutex;                                            // stable: returned references stay valid)
};

namespace cpn {

static int64_t tensor_bytes(const cpn_tensor_desc &t, int N, int h, int w, int elem) {
    const int64_t b = (int64_t) N * h * w * t.channels * elem;
    return (b + 255) / 256 * 256;
}

// Spatial sizes of every tensor for an H x W input, following the reference's modules: conv / max-pool output
// size = floor((in + 2p - k) / s) + 1; a nearest-resized source takes the size of the other concat source
// (F.interpolate(size=lateral.shape), models/unet.py:213-217, torchvision FPN) or, without one, twice its own size
// (scale_factor=2, bridge levels); CPN_OP_BILINEAR resizes to the INPUT size (_equal_size(features, inputs),
// models/cpn.py:277-278) and is a no-op alias when the sizes already agree.
// argument struct of a CPN_OP_CONV_PAIR op (tensor pointers / strides filled by the caller)
static PairArgs pair_args(const cpn_plan *p, const cpn_op_desc &o, int N, int H, int W) {
    PairArgs a{};
    a.N = N; a.H = H; a.W = W;
    a.cin = o.cin_b; a.cmid = o.cout_b; a.cb2 = o.fuse_cout; a.stride = o.stride;
    a.c_stride = o.src0 >= 0 && p ? p->tensors[o.src0].channels : o.cin_b;
    a.dst_stride = o.dst >= 0 && p ? p->tensors[o.dst].channels : o.cout_b;
    if (p) {
        a.w1 = p->weights + o.weight_offset;
        a.b1 = o.bias_offset >= 0 ? p->bias + o.bias_offset : nullptr;
        a.w2 = p->weights + o.fuse_weight_offset;
        a.b2 = o.fuse_bias_offset >= 0 ? p->bias + o.fuse_bias_offset : nullptr;
    }
    return a;
}

// ConvArgs of a CPN_OP_CONV_BRIDGE op `o` (behind the scatter conv c1 and the 3x3 conv c2 it restates); pointers filled by the caller
static int build_conv_args(const cpn_plan *p, const cpn_op_desc &o, int N, ConvArgs &a, const void *s0, int c0s, const void *s1,
                           int c1s, const void *res, int rs, void *dst, int ds, int Hin, int Win, const int *src_dims);
static int bridge_args(const cpn_plan *p, const cpn_op_desc &o, const cpn_op_desc &c2, int N, int Hp, int Wp, ConvArgs &a,
                       const void *src, int c_stride, const void *res, int rs, void *dst, int ds, const void *weights,
                       const float *bias) {
    static const char dummy = 0;
    const int sdims[6] = {2 * Hp, 2 * Wp, 0, 0, 2 * Hp, 2 * Wp};
    int rc = build_conv_args(p, c2, N, a, &dummy, c2.cin_b, nullptr, 0, res, rs, dst, ds, 2 * Hp, 2 * Wp, sdims);
    if (rc) return rc;
    a.src0 = src;  // (unused by the kernel: its halo tiles are computed from pre_src)
    a.pre_src = src; a.pre_stride = c_stride; a.pre_cin = o.cin_b; a.pre_H = Hp; a.pre_W = Wp;
    a.pre_w = (const unsigned char *) weights + o.weight_offset;
    a.pre_b = (bias && o.bias_offset >= 0) ? bias + o.bias_offset : nullptr;
    a.weights = (const unsigned char *) weights + o.fuse_weight_offset;
    a.bias = (bias && o.fuse_bias_offset >= 0) ? bias + o.fuse_bias_offset : nullptr;
    return 0;
}
// FLOPs the bridge kernel's MFMA loops execute: the 3x3 conv + the scatter conv on every tile's 18 x 34 halo (20 fragments)
static double bridge_executed_flops(const ConvArgs &a) {
    const double tiles = (double) a.N * ((a.Hout + 15) / 16) * ((a.Wout + 31) / 32);
    return conv_executed_flops(a) + tiles * 20. * 32. * 64. * a.pre_cin * 4. * 2.;
}

static void propagate_dims(const cpn_plan *p, int N, int H, int W, ShapePlan &sp, int blphase_mode, int pair_mode_) {
    const int pair_mode = pair_mode_ & 7, bridge_mode = (pair_mode_ >> 3) & 1;
    const int nt = (int) p->tensors.size();
    sp.th.assign(nt, 0);
    sp.tw.assign(nt, 0);
    for (int i = 0; i < CPN_NUM_OUTPUTS; ++i) sp.out_h[i] = sp.out_w[i] = 0;
    auto bad = [&](const char *m) { sp.error = CPN_E_INVALID; sp.message = m; };
    sp.skip.assign(p->ops.size(), 0);
    sp.ring.assign(p->ops.size(), 0);
    for (size_t oi = 0; oi < p->ops.size(); ++oi) {
        const cpn_op_desc &o = p->ops[oi];
        if (sp.error) return;
        if (o.op == CPN_OP_CONV && o.subpixel == CPN_SUBPIXEL_HEAD) {
            // the decomposition holds for the exact x2 case only (PyTorch's nearest index at any other ratio does not
            // split into phases): decided per input size
            const bool exact = p->precision != CPN_PRECISION_F32 && o.src1 >= 0 && o.up1 &&
                               sp.th[o.src0] == 2 * sp.th[o.src1] && sp.tw[o.src0] == 2 * sp.tw[o.src1];
            sp.skip[oi] = exact;
            sp.skip[oi + 1] = sp.skip[oi + 2] = !exact;
        }
        if (o.op == CPN_OP_CONV && o.subpixel == CPN_SUBPIXEL_BL_HEAD) {
            // bilinear phases + frame instead of the conv over the resized map wherever the resize is an exact x2
            // (CPN_BLPHASE=0: kernel A/B switch, read when a shape is planned)
            // ... and the decomposition executes fewer MACs than the conv it replaces: the frame is whole 8 x 32 tiles of the
            // k x k conv, most of a small image (CPN_BLPHASE=0 / 2: never / wherever exact -- kernel A/B and tests)
            const int mode = blphase_mode;
            // bf16 plans: head and frame conv resize their source in the halo loader (up0 == 2, all three ops read the
            // low-resolution map); fp8 plans: the resize is an op of its own, head and frame conv read its output
            const int lo = p->ops[oi + 1].src0;
            const bool hi_ok = o.up0 == 2 ? lo == o.src0 : (sp.th[o.src0] == H && sp.tw[o.src0] == W);
            bool exact = mode != 0 && p->precision != CPN_PRECISION_F32 && hi_ok && 2 * sp.th[lo] == H && 2 * sp.tw[lo] == W &&
                         sp.th[lo] >= o.kh && sp.tw[lo] >= o.kw;
            if (exact && mode != 2) {
                const int k2 = (o.kh + 3) / 2, m = 2 * ((o.kh / 2 + 1) / 2);
                auto tiles = [](int h, int w) { return (double) ((h + 7) / 8) * ((w + 31) / 32); };
                // frame tiles exactly as the frame launch enumerates them (cpn_kernels.h frame_tiles, 8 x 32 tiles of a stride-1
                // k x k conv: whole tile rows above / below the box, per row that crosses it one wrap tile or its side tiles)
                const double frame = (double) frame_tiles(H, W, m, 8, 32, o.kw > 1 ? o.kw : 0).total;
                const double head = tiles(H, W) * o.kh * o.kh;
                const double parts = 4. * tiles(H / 2, W / 2) * k2 * k2 + frame * o.kh * o.kh;
                exact = parts <= 0.85 * head;
            }
            if (exact && o.up0 != 2) {
                // the materialised resized map is read by the frame conv alone: its resize op writes only the pixels the
                // frame's outputs reach (frame width + conv padding from the border)
                int producer = -1;
                bool shared = false;
                for (size_t j = 0; j < p->ops.size(); ++j) {
                    const cpn_op_desc &q = p->ops[j];
                    if (j < oi && q.op == CPN_OP_BILINEAR && q.dst == o.src0 && q.subpixel == CPN_SUBPIXEL_BL_FRAME) producer = (int) j;
                    if (j != oi && j != oi + 2 && (q.src0 == o.src0 || q.src1 == o.src0 || q.res == o.src0)) shared = true;
                }
                if (producer >= 0 && !shared) sp.ring[producer] = 2 * ((o.kh / 2 + 1) / 2) + o.kh / 2;
            }
            sp.skip[oi] = exact;
            sp.skip[oi + 1] = sp.skip[oi + 2] = !exact;
        }
        if (o.alt == 1 || o.alt == 2) {
            // stem alternatives: the fast pair (padded 4-channel input layout inside the input tensor's storage + the
            // dedicated 7x7 stride-2 kernel) wherever that layout fits, the generic pair otherwise
            int tin = -1;
            for (const cpn_op_desc &q : p->ops)
                if (q.op == CPN_OP_INPUT) { tin = q.dst; break; }
            // (the padded layout is bf16 [H + 6][W + 8][4] = 8 bytes per pixel in bf16 AND fp8 plans; the input tensor
            // offers channels * 2 | 1 bytes per pixel)
            const int elem = p->precision == CPN_PRECISION_FP8 ? 1 : 2;
            const bool fast = p->precision != CPN_PRECISION_F32 && tin >= 0 &&
                              (int64_t) (H + STEM_PAD_ROWS) * (W + STEM_PAD_COLS) * 8 <=
                                  (int64_t) H * W * p->tensors[tin].channels * elem;
            sp.skip[oi] = (o.alt == 2) != fast;
        }
        switch (o.op) {
            case CPN_OP_CONV_BRIDGE: {
                // runs instead of the scatter conv + 3x3 conv in front of it wherever the kernel's 16 x 32 tiles fit the output
                // (CPN_BRIDGE=0: never -- kernel A/B and tests)
                const int Hp = sp.th[o.src0], Wp = sp.tw[o.src0];
                const bool fused = bridge_mode != 0 && p->precision == CPN_PRECISION_BF16 && 2 * Hp >= 16 && 2 * Wp >= 32 &&
                                   sp.th[o.dst] == 2 * Hp && sp.tw[o.dst] == 2 * Wp;
                sp.skip[oi] = !fused;
                sp.skip[oi - 1] = sp.skip[oi - 2] = fused;
                break;
            }
            case CPN_OP_CONV_PAIR: {
                // runs instead of the two convs in front of it wherever the kernel's full-width strips fit the feature map
                // and its strips x slabs fill the chip (CPN_PAIR=0 / 2: never / wherever supported -- kernel A/B and tests)
                const int mode = pair_mode;
                const int mid = p->ops[oi - 2].dst;  // conv1's output: the kernel's H x W (conv2 may stride it down)
                const PairArgs pa = pair_args(p, o, N, sp.th[mid], sp.tw[mid]);
                const bool fused = mode != 0 && p->precision == CPN_PRECISION_BF16 && conv_pair_supported(pa) &&
                                   (mode == 2 || conv_pair_blocks(pa) >= 192);
                sp.skip[oi] = !fused;
                sp.skip[oi - 1] = sp.skip[oi - 2] = fused;
                break;
            }
            case CPN_OP_INPUT:
            case CPN_OP_INPUT_STEM: sp.th[o.dst] = H; sp.tw[o.dst] = W; break;
            case CPN_OP_STEM7:
                if (o.dst < 0 || o.src0 < 0) { bad("stem conv: missing tensors"); break; }
                sp.th[o.dst] = (sp.th[o.src0] - 1) / 2 + 1;  // floor((in + 6 - 7) / 2) + 1
                sp.tw[o.dst] = (sp.tw[o.src0] - 1) / 2 + 1;
                break;
            case CPN_OP_MAXPOOL:
                sp.th[o.dst] = (sp.th[o.src0] + 2 * o.pad - o.kh) / o.stride + 1;
                sp.tw[o.dst] = (sp.tw[o.src0] + 2 * o.pad - o.kw) / o.stride + 1;
                if (sp.th[o.src0] + 2 * o.pad < o.kh || sp.tw[o.src0] + 2 * o.pad < o.kw) bad("input too small for the max-pool");
                break;
            case CPN_OP_BILINEAR: sp.th[o.dst] = H; sp.tw[o.dst] = W; break;
            case CPN_OP_ACT:
                if (o.dst < 0 || o.src0 < 0) { bad("activation op: missing tensors"); break; }
                sp.th[o.dst] = sp.th[o.src0]; sp.tw[o.dst] = sp.tw[o.src0];
                break;
            case CPN_OP_CONV:
            case CPN_OP_CONV_DEFERRED: {
                int hv, wv;
                if (o.up0 && o.up1) { bad("conv: both sources resized"); break; }
                if (o.subpixel == CPN_SUBPIXEL_PHASE) {  // 2 x 2 taps per output phase: the output keeps the source's size
                    sp.th[o.dst] = sp.th[o.src0]; sp.tw[o.dst] = sp.tw[o.src0];
                    break;
                }
                if (o.subpixel == CPN_SUBPIXEL_BL_PHASE) break;  // (writes the BL_HEAD op's external output: sized there)
                if (o.subpixel == CPN_SUBPIXEL_SCATTER) {  // conv over the x2-upsampled source (scale_factor = 2)
                    if (o.dst < 0) { bad("conv: a sub-pixel scatter conv needs a tensor destination"); break; }
                    sp.th[o.dst] = 2 * sp.th[o.src0]; sp.tw[o.dst] = 2 * sp.tw[o.src0];
                    break;
                }
                if (o.up0 == 2) { hv = H; wv = W; }  // bilinear resize of the source to the INPUT size (cpn.py:277-278)
                else if (o.up1) { hv = sp.th[o.src0]; wv = sp.tw[o.src0]; }
                else if (o.up0 && o.src1 >= 0) { hv = sp.th[o.src1]; wv = sp.tw[o.src1]; }
                else if (o.up0) { hv = 2 * sp.th[o.src0]; wv = 2 * sp.tw[o.src0]; }
                else {
                    hv = sp.th[o.src0]; wv = sp.tw[o.src0];
                    if (o.src1 >= 0 && (sp.th[o.src1] != hv || sp.tw[o.src1] != wv)) { bad("conv: concat sources differ in size"); break; }
                }
                if (hv + 2 * o.pad < o.kh || wv + 2 * o.pad < o.kw) { bad("input too small for a convolution of the graph"); break; }
                const int ho = (hv + 2 * o.pad - o.kh) / o.stride + 1, wo = (wv + 2 * o.pad - o.kw) / o.stride + 1;
                if (o.res >= 0 && !o.res_up && (sp.th[o.res] != ho || sp.tw[o.res] != wo)) { bad("conv: residual size mismatch"); break; }
                if (o.res >= 0 && o.res_up == 2 && !sp.skip[oi] && (2 * sp.th[o.res] != ho || 2 * sp.tw[o.res] != wo)) { bad("conv: phase tensor size mismatch"); break; }
                if (o.dst >= 0) { sp.th[o.dst] = ho; sp.tw[o.dst] = wo; }
                else if (o.out_index >= 0 && o.out_index < CPN_NUM_OUTPUTS) { sp.out_h[o.out_index] = ho; sp.out_w[o.out_index] = wo; }
                break;
            }
            default: bad("unknown op");
        }
    }
    for (int t = 0; t < nt && !sp.error; ++t) {
        if (sp.th[t] < 0 || sp.tw[t] < 0) bad("negative tensor size");
        sp.max_elems = std::max(sp.max_elems, (int64_t) sp.th[t] * sp.tw[t] * p->tensors[t].channels);
    }
}

static const ShapePlan &get_shape_plan(cpn_plan *p, int N, int H, int W) {
    std::lock_guard<std::mutex> lock(p->shape_mutex);
    // the executor's A/B switches are part of the key: toggling CPN_BLPHASE / CPN_PAIR on a live plan re-plans the shape
    // (the Python engine's hipGraph key carries them as well)
    const char *eb = getenv("CPN_BLPHASE"), *ep = getenv("CPN_PAIR"), *er = getenv("CPN_BRIDGE");
    const int blphase_mode = eb ? atoi(eb) : 1, pair_mode = (ep ? atoi(ep) : 1) + 8 * (er ? (atoi(er) != 0) : 1);  // (bit 3: bridge fusion)
    auto key = std::make_tuple(N, H, W, blphase_mode, pair_mode);
    auto it = p->shape_plans.find(key);
    if (it != p->shape_plans.end()) return it->second;
    ShapePlan sp;
    propagate_dims(p, N, H, W, sp, blphase_mode, pair_mode);
    const int nt = (int) p->tensors.size();
    sp.offsets.assign(nt, -1);
    if (sp.error) return p->shape_plans.emplace(key, std::move(sp)).first->second;
    // a bilinear op whose source already has the input size is an alias (no kernel, shared storage)
    std::vector<int> root(nt);
    for (int t = 0; t < nt; ++t) root[t] = t;
    for (const cpn_op_desc &o : p->ops)
        if (o.op == CPN_OP_BILINEAR && sp.th[o.src0] == sp.th[o.dst] && sp.tw[o.src0] == sp.tw[o.dst]) root[o.dst] = root[o.src0];
    std::vector<int> def(nt, -1), last(nt, -1);
    for (int i = 0; i < (int) p->ops.size(); ++i) {
        const cpn_op_desc &o = p->ops[i];
        if (sp.skip[i]) continue;  // (the alternative of a sub-pixel triple that does not run at this size)
        if (o.dst >= 0 && def[root[o.dst]] < 0) def[root[o.dst]] = i;
        // (the sources of a deferred conv are read after the run, cpn_sparse_heads: they stay live to the end)
        const int use = o.op == CPN_OP_CONV_DEFERRED ? (int) p->ops.size() : i;
        for (int s_ : {o.src0, o.src1, o.res})
            if (s_ >= 0) last[root[s_]] = std::max(last[root[s_]], use);
        if (o.dst >= 0) last[root[o.dst]] = std::max(last[root[o.dst]], i);
    }
    std::vector<int> order;
    for (int t = 0; t < nt; ++t)
        if (root[t] == t && def[t] >= 0) order.push_back(t);
    std::sort(order.begin(), order.end(), [&](int a, int b) { return def[a] < def[b]; });
    // bytes per element: fp32 4 | bf16 2 | e4m3 1 -- except the bf16 partial-sum tensors of an fp8 plan (scale < 0)
    auto elem_of = [&](int t) {
        return p->precision == CPN_PRECISION_F32 ? 4 : (p->precision == CPN_PRECISION_FP8 ? (p->tensors[t].scale < 0.f ? 2 : 1) : 2);
    };
    std::vector<int> placed;
    for (int t : order) {
        const int elem = elem_of(t);
        const int64_t sz = tensor_bytes(p->tensors[t], N, sp.th[t], sp.tw[t], elem);
        // candidate offsets: 0 and the end of every conflicting placed tensor; take the lowest that fits
        std::vector<std::pair<int64_t, int64_t>> busy;  // [begin, end) of live-overlapping tensors
        for (int q : placed)
            if (!(last[q] < def[t] || last[t] < def[q]))
                busy.emplace_back(sp.offsets[q], sp.offsets[q] + tensor_bytes(p->tensors[q], N, sp.th[q], sp.tw[q], elem_of(q)));
        std::sort(busy.begin(), busy.end());
        int64_t off = 0;
        for (auto &b : busy) {
            if (off + sz <= b.first) break;
            off = std::max(off, b.second);
        }
        sp.offsets[t] = off;
        sp.total = std::max(sp.total, off + sz);
        placed.push_back(t);
    }
    for (int t = 0; t < nt; ++t)
        if (root[t] != t) sp.offsets[t] = sp.offsets[root[t]];
    return p->shape_plans.emplace(key, std::move(sp)).first->second;
}

struct Dims {
    int h, w;
};

// Hin x Win: virtual (post-resize) input size.  src_dims (optional): stored sizes {Hs0, Ws0, Hs1, Ws1, Hr, Wr} of the
// two sources and the residual; without it a resized source / residual is an exact x2 (the stand-alone cpn_conv2d).
static int build_conv_args(const cpn_plan *p, const cpn_op_desc &o, int N, ConvArgs &a, const void *s0,
                           int c0s, const void *s1, int c1s, const void *res, int rs, void *dst, int ds, int Hin,
                           int Win, const int *src_dims = nullptr) {
    a = ConvArgs{};
    a.src0 = s0; a.src1 = s1; a.c0_stride = c0s; a.c1_stride = c1s;
    a.c0_used = o.c0_used;
    a.up0 = o.up0; a.up1 = o.up1;
    a.N = N; a.Hin = Hin; a.Win = Win;
    a.Hs0 = src_dims ? src_dims[0] : (o.up0 ? Hin >> 1 : Hin); a.Ws0 = src_dims ? src_dims[1] : (o.up0 ? Win >> 1 : Win);
    a.Hs1 = src_dims ? src_dims[2] : (o.up1 ? Hin >> 1 : Hin); a.Ws1 = src_dims ? src_dims[3] : (o.up1 ? Win >> 1 : Win);
    if (!o.up0) { a.Hs0 = Hin; a.Ws0 = Win; }
    if (!o.up1) { a.Hs1 = Hin; a.Ws1 = Win; }
    if (a.Hs0 == Hin && a.Ws0 == Win) a.up0 = 0;  // same size: the resize (nearest or bilinear) is the identity
    if (a.Hs1 == Hin && a.Ws1 == Win) a.up1 = 0;
    if (Hin <= 0 || Win <= 0 || a.Hs0 <= 0 || a.Ws0 <= 0 || (s1 && (a.Hs1 <= 0 || a.Ws1 <= 0)))
        return fail(CPN_E_INVALID, "conv: empty input");
    a.sy0 = (float) a.Hs0 / (float) Hin; a.sx0 = (float) a.Ws0 / (float) Win;
    a.sy1 = (float) a.Hs1 / (float) Hin; a.sx1 = (float) a.Ws1 / (float) Win;
    a.KH = o.kh; a.KW = o.kw; a.stride = o.stride; a.pad = o.pad;
    a.Hout = (Hin + 2 * o.pad - o.kh) / o.stride + 1;
    a.Wout = (Win + 2 * o.pad - o.kw) / o.stride + 1;
    a.phase = o.subpixel == CPN_SUBPIXEL_PHASE ? 1 : (o.subpixel == CPN_SUBPIXEL_SCATTER ? 2 : 0);
    if (o.subpixel == CPN_SUBPIXEL_BL_PHASE) {
        // four k2 x k2 convs on the low-resolution map, one symmetric support (pad k2 / 2) and one bias for all phases, fused
        // ReadOut tail scattered to the [2 Hin][2 Win] planes; the frame of k2 / 2 low-resolution pixels belongs to BL_FRAME
        if (o.kh != o.kw || o.kh % 2 == 0 || o.pad != o.kh / 2 || o.stride != 1 || o.bundles != 4 || s1 || o.up0 || res ||
            o.fuse_cout <= 0 || o.dst >= 0)
            return fail(CPN_E_INVALID, "conv: a bilinear phase conv is k2 x k2, pad k2 / 2, stride 1, 4 bundles, one plain source, "
                                       "fused ReadOut tail");
        a.phase = 3;
        a.region = 1;
        a.region_margin = o.kh / 2;
        a.Hout = Hin; a.Wout = Win;
    }
    if (o.subpixel == CPN_SUBPIXEL_BL_FRAME) {  // the conv over the resized map, frame only: k = 2 k2 - 3 -> F = 2 (k2 / 2)
        if (o.fuse_cout <= 0 || o.dst >= 0) return fail(CPN_E_INVALID, "conv: a bilinear frame conv is a fused ReadOut head over a bilinear-resized source");
        a.region = 2;
        a.region_margin = 2 * ((o.kh / 2 + 1) / 2);
    }
    if (a.phase == 1 || a.phase == 2) {  // four 2 x 2 convs (one per output phase, padding (1 - py, 1 - px)) on the low-resolution map
        if (o.kh != 2 || o.kw != 2 || o.pad != 1 || o.stride != 1 || o.bundles != 4 || s1 || o.up0 || res)
            return fail(CPN_E_INVALID, "conv: a sub-pixel phase conv is 2x2, pad 1, stride 1, 4 bundles, one plain source");
        a.Hout = Hin; a.Wout = Win;
    }
    a.bundles = o.bundles; a.cin_b = o.cin_b; a.cout_b = o.cout_b;
    a.weights = p ? p->weights + o.weight_offset : nullptr;
    a.bias = (p && o.bias_offset >= 0) ? p->bias + o.bias_offset : nullptr;
    a.res = res; a.res_stride = rs; a.res_up = o.res_up;
    a.Hr = (src_dims && o.res_up) ? src_dims[4] : (o.res_up ? a.Hout >> 1 : a.Hout);
    a.Wr = (src_dims && o.res_up) ? src_dims[5] : (o.res_up ? a.Wout >> 1 : a.Wout);
    if (o.res_up == 2) {
        if (!res || rs % 4 || 2 * a.Hr != a.Hout || 2 * a.Wr != a.Wout)
            return fail(CPN_E_INVALID, "conv: a pixel-shuffled residual is a [H/2, W/2, 4 * C] phase tensor");
        a.res_cph = rs / 4;
    }
    if (res && (a.Hr <= 0 || a.Wr <= 0)) return fail(CPN_E_INVALID, "conv: empty residual");
    a.ry = (float) a.Hr / (float) a.Hout; a.rx = (float) a.Wr / (float) a.Wout;
    a.act = o.act; a.act_scale = o.act_scale;
    a.out_mode = o.dst >= 0 ? OUT_BF16_NHWC : (o.fuse_cout > 0 ? OUT_FUSED_HEAD : OUT_F32_NCHW);
    if (o.fuse_cout > 0) {
        a.fuse_w = p ? p->weights + o.fuse_weight_offset : nullptr;
        a.fuse_b = (p && o.fuse_bias_offset >= 0) ? p->bias + o.fuse_bias_offset : nullptr;
        a.fuse_cout = o.fuse_cout; a.fuse_act = o.fuse_act; a.fuse_scale = o.fuse_act_scale;
    }
    a.dst = dst; a.dst_stride = ds; a.dst_coff = o.dst_coff;
    a.cout_real = o.cout_real;
    const int kc = (p && p->precision == CPN_PRECISION_FP8) ? 64 : 32;  // input channels per packed weight record
    if (o.cin_b <= 0 || o.cin_b % kc || o.cout_b <= 0 || o.cout_b % 32 || o.c0_used % kc || o.bundles < 1)
        return fail(CPN_E_INVALID, "conv: channel counts must be positive multiples of 32 (64 input channels for fp8)");
    if (p && p->precision == CPN_PRECISION_FP8) {
        a.mult = o.mult_offset >= 0 ? p->bias + o.mult_offset : nullptr;
        a.res_wide = o.res >= 0 && p->tensors[o.res].scale < 0.f;   // (bf16 partial sums of a sub-pixel triple)
        a.dst_wide = o.dst >= 0 && p->tensors[o.dst].scale < 0.f;
        a.res_scale = o.res >= 0 ? (a.res_wide ? 1.f : p->tensors[o.res].scale) : 0.f;
        a.out_inv_scale = o.dst >= 0 ? (a.dst_wide ? 1.f : 1.f / p->tensors[o.dst].scale) : 0.f;
    }
    if (o.bundles > 1 && s1) return fail(CPN_E_INVALID, "conv: grouped conv with two sources");
    if (!s1 && o.c0_used < (a.phase ? 1 : o.bundles) * o.cin_b) return fail(CPN_E_INVALID, "conv: c0_used smaller than input channels");
    // sources are read through raw buffer descriptors whose out-of-range sentinel is byte offset 2^31 (conv_igemm.hip):
    // a source tensor may hold at most 2^31 BYTES (fp32 verification path: 2^31 elements); destinations are addressed
    // with 32-bit element offsets
    const int64_t src_limit = (p && p->precision == CPN_PRECISION_F32) ? (1ll << 31) : (1ll << 31) / (kc == 64 ? 1 : 2);
    if ((int64_t) N * a.Hs0 * a.Ws0 * c0s >= src_limit || (s1 && (int64_t) N * a.Hs1 * a.Ws1 * c1s >= src_limit) ||
        (int64_t) N * a.Hout * a.Wout * (a.phase == 2 ? 4 : 1) * std::max(ds, 1) >= (1ll << 31))
        return fail(CPN_E_UNSUPPORTED, "conv: tensor too large for one launch (sources: 2^31 bytes, destination: 2^31 "
                                       "elements); split the batch");
    // plain 1x1 convs are GEMMs over the flattened pixel axis: re-tile as [1, M/32, 32] so that narrow images
    // (16x16 at stride 32) still fill the 32-pixel MFMA column fragments
    if (o.kh == 1 && o.kw == 1 && o.stride == 1 && o.pad == 0 && !o.up0 && !o.up1 && !o.res_up &&
        a.out_mode == OUT_BF16_NHWC) {
        const int64_t M = (int64_t) N * Hin * Win;
        if (M % 32 == 0) {
            a.N = 1; a.Hin = a.Hout = a.Hs0 = a.Hs1 = a.Hr = (int) (M / 32); a.Win = a.Wout = a.Ws0 = a.Ws1 = a.Wr = 32;
        }
    }
    return 0;
}

}  // namespace cpn

using namespace cpn;

extern "C" {

const char *cpn_last_error(void) { return g_last_error.c_str(); }
int cpn_abi_version(void) { return CPN_ABI_VERSION; }

int cpn_plan_create(cpn_plan **plan, const cpn_tensor_desc *tensors, int32_t n_tensors, const cpn_op_desc *ops,
                    int32_t n_ops, const void *weights, size_t weight_bytes, const float *bias, size_t bias_count,
                    int32_t precision) {
    if (precision != CPN_PRECISION_BF16 && precision != CPN_PRECISION_F32 && precision != CPN_PRECISION_FP8)
        return fail(CPN_E_INVALID, "cpn_plan_create: unknown precision");
    if (!plan || !tensors || !ops || n_tensors <= 0 || n_ops <= 0) return fail(CPN_E_INVALID, "cpn_plan_create: null/empty");
    cpn_plan *p = new cpn_plan();
    p->tensors.assign(tensors, tensors + n_tensors);
    p->ops.assign(ops, ops + n_ops);
    p->weights = (const unsigned char *) weights;
    p->weight_bytes = weight_bytes;
    p->bias = bias;
    p->bias_count = bias_count;
    p->precision = precision;
    for (const auto &t : p->tensors)
        if (t.channels <= 0 || t.channels % (precision == CPN_PRECISION_FP8 ? 64 : 32) || t.down < 1 ||
            (t.down & (t.down - 1)) || t.down > 32 || (precision == CPN_PRECISION_FP8 && !(t.scale > 0.f || t.scale < 0.f))) {
            delete p;
            return fail(CPN_E_INVALID, "cpn_plan_create: tensor channels must be multiples of 32 (fp8: 64, with a "
                                       "positive scale, or a negative one for a bf16 partial-sum tensor), down a power of two <= 32");
        }
    for (const auto &o : p->ops) {
        for (int s : {o.src0, o.src1, o.res, o.dst})
            if (s >= n_tensors) {
                delete p;
                return fail(CPN_E_INVALID, "cpn_plan_create: tensor id out of range");
            }
        const size_t oi_ = (size_t) (&o - p->ops.data());
        if (precision == CPN_PRECISION_FP8) {
            // bf16 partial-sum tensors (negative scale) exist between the PHASE and the LATERAL op of a sub-pixel triple only
            auto wide = [&](int t) { return t >= 0 && p->tensors[t].scale < 0.f; };
            if (wide(o.src0) || wide(o.src1) || (wide(o.dst) != (o.op == CPN_OP_CONV && o.subpixel == CPN_SUBPIXEL_PHASE && o.dst >= 0)) ||
                (wide(o.res) && !(o.op == CPN_OP_CONV && o.subpixel == CPN_SUBPIXEL_LATERAL && o.res_up == 2))) {
                delete p;
                return fail(CPN_E_INVALID, "cpn_plan_create: a bf16 tensor of an fp8 plan (negative scale) is the destination of a "
                                           "sub-pixel PHASE op and the residual of its LATERAL op, nothing else");
            }
        }
        if (o.subpixel == CPN_SUBPIXEL_HEAD &&
            (o.op != CPN_OP_CONV || precision == CPN_PRECISION_F32 || oi_ + 2 >= p->ops.size() || p->ops[oi_ + 1].subpixel != CPN_SUBPIXEL_PHASE ||
             p->ops[oi_ + 2].subpixel != CPN_SUBPIXEL_LATERAL || p->ops[oi_ + 1].op != CPN_OP_CONV ||
             p->ops[oi_ + 2].op != CPN_OP_CONV || p->ops[oi_ + 2].dst != o.dst || p->ops[oi_ + 2].res != p->ops[oi_ + 1].dst ||
             p->ops[oi_ + 2].res_up != 2 || p->ops[oi_ + 1].src0 != o.src1 || p->ops[oi_ + 2].src0 != o.src0 || !o.up1 ||
             o.src1 < 0 || o.dst < 0 || p->ops[oi_ + 1].dst < 0)) {
            delete p;
            return fail(CPN_E_INVALID, "cpn_plan_create: malformed sub-pixel triple (HEAD, PHASE, LATERAL)");
        }
        if (o.subpixel == CPN_SUBPIXEL_BL_HEAD &&
            (o.op != CPN_OP_CONV || precision == CPN_PRECISION_F32 || oi_ + 2 >= p->ops.size() || o.dst >= 0 ||
             (o.up0 != 2 && o.up0 != 0) || (o.up0 == 2 && precision == CPN_PRECISION_FP8) ||  // (fp8: the resize is its own op)
             o.fuse_cout <= 0 || o.kh != o.kw || o.kh % 4 != 3 || p->ops[oi_ + 1].subpixel != CPN_SUBPIXEL_BL_PHASE ||
             p->ops[oi_ + 2].subpixel != CPN_SUBPIXEL_BL_FRAME || p->ops[oi_ + 1].op != CPN_OP_CONV || p->ops[oi_ + 2].op != CPN_OP_CONV ||
             (o.up0 == 2 && p->ops[oi_ + 1].src0 != o.src0) || p->ops[oi_ + 1].src0 < 0 ||
             p->tensors[p->ops[oi_ + 1].src0].channels != p->tensors[o.src0].channels ||
             p->ops[oi_ + 2].src0 != o.src0 || p->ops[oi_ + 1].out_index != o.out_index ||
             p->ops[oi_ + 2].out_index != o.out_index || p->ops[oi_ + 1].kh != (o.kh + 3) / 2 || p->ops[oi_ + 2].kh != o.kh ||
             p->ops[oi_ + 2].up0 != o.up0 || p->ops[oi_ + 1].fuse_cout != o.fuse_cout || p->ops[oi_ + 2].fuse_cout != o.fuse_cout)) {
            delete p;
            return fail(CPN_E_INVALID, "cpn_plan_create: malformed bilinear sub-pixel triple (BL_HEAD, BL_PHASE, BL_FRAME)");
        }
        // (a resize op flagged BL_FRAME feeds the frame conv of a triple: propagate_dims)
        if ((o.subpixel == CPN_SUBPIXEL_BL_PHASE && (oi_ < 1 || p->ops[oi_ - 1].subpixel != CPN_SUBPIXEL_BL_HEAD)) ||
            (o.subpixel == CPN_SUBPIXEL_BL_FRAME && o.op != CPN_OP_BILINEAR && (oi_ < 2 || p->ops[oi_ - 2].subpixel != CPN_SUBPIXEL_BL_HEAD))) {
            delete p;
            return fail(CPN_E_INVALID, "cpn_plan_create: bilinear PHASE / FRAME ops must follow their BL_HEAD op");
        }
        if ((o.subpixel == CPN_SUBPIXEL_PHASE && (oi_ < 1 || p->ops[oi_ - 1].subpixel != CPN_SUBPIXEL_HEAD)) ||
            (o.subpixel == CPN_SUBPIXEL_LATERAL && (oi_ < 2 || p->ops[oi_ - 2].subpixel != CPN_SUBPIXEL_HEAD))) {
            delete p;
            return fail(CPN_E_INVALID, "cpn_plan_create: sub-pixel PHASE / LATERAL ops must follow their HEAD op");
        }
        if ((o.op == CPN_OP_INPUT_STEM || o.op == CPN_OP_STEM7) &&
            (precision == CPN_PRECISION_F32 || o.alt != 2 || o.dst < 0 ||
             (o.op == CPN_OP_INPUT_STEM && (o.in_channels < 1 || o.in_channels > 4)) ||
             (o.op == CPN_OP_STEM7 && ((o.cout_b != 32 && o.cout_b != 64) || o.src0 < 0 || o.weight_offset < 0 ||
                                       (size_t) o.weight_offset + (size_t) 7 * o.cout_b * 64 > weight_bytes ||
                                       (o.bias_offset >= 0 && (size_t) o.bias_offset + o.cout_b > bias_count))))) {
            delete p;
            return fail(CPN_E_INVALID, "cpn_plan_create: malformed stem fast-path op (bf16 / fp8 plans, alt = 2, <= 4 input "
                                       "channels, 32 | 64 output channels)");
        }
        if (o.op == CPN_OP_CONV_PAIR) {
            const cpn_op_desc *c1 = oi_ >= 2 ? &p->ops[oi_ - 2] : nullptr, *c2 = oi_ >= 2 ? &p->ops[oi_ - 1] : nullptr;
            const size_t it1 = (size_t) (o.cin_b / 32), it2 = (size_t) (o.fuse_cout / 32) * 9;
            if (precision != CPN_PRECISION_BF16 || !c1 || c1->op != CPN_OP_CONV || c2->op != CPN_OP_CONV || c1->kh != 1 ||
                c1->kw != 1 || c1->stride != 1 || c1->pad != 0 || c1->bundles != 1 || c1->src1 >= 0 || c1->res >= 0 ||
                c1->up0 || c1->act != CPN_ACT_RELU || c1->subpixel || c1->alt || c1->dst < 0 || c2->src0 != c1->dst ||
                c2->src1 >= 0 || c2->res >= 0 || c2->up0 || c2->kh != 3 || c2->kw != 3 || (c2->stride != 1 && c2->stride != 2) ||
                c2->pad != 1 || o.stride != c2->stride ||
                c2->act != CPN_ACT_RELU || c2->subpixel || c2->alt || c2->dst < 0 || c2->cin_b != c2->cout_b ||
                (c2->cout_b != 32 && c2->cout_b != 64) || c2->bundles * c2->cout_b != c1->cout_b || o.src0 != c1->src0 ||
                o.dst != c2->dst || o.cin_b != c1->cin_b || o.cout_b != c1->cout_b || o.fuse_cout != c2->cout_b ||
                o.bundles != c2->bundles || o.weight_offset != c1->weight_offset || o.bias_offset != c1->bias_offset ||
                o.fuse_weight_offset != c2->weight_offset || o.fuse_bias_offset != c2->bias_offset || o.cin_b % 32 ||
                p->tensors[o.dst].channels != o.cout_b ||
                (size_t) o.weight_offset + (it1 + (it1 & 1)) * o.cout_b * 64 > weight_bytes ||
                (size_t) o.fuse_weight_offset + (size_t) o.bundles * (it2 + (it2 & 1)) * o.fuse_cout * 64 > weight_bytes) {
                delete p;
                return fail(CPN_E_INVALID, "cpn_plan_create: a CPN_OP_CONV_PAIR op must follow the 1x1 conv + ReLU and the grouped "
                                           "3x3 conv + ReLU (stride 1 | 2, bundles of 32 | 64 channels) it restates and share their offsets");
            }
        }
        if (o.op == CPN_OP_CONV_BRIDGE) {
            const cpn_op_desc *c1 = oi_ >= 2 ? &p->ops[oi_ - 2] : nullptr, *c2 = oi_ >= 2 ? &p->ops[oi_ - 1] : nullptr;
            bool ok = precision == CPN_PRECISION_BF16 && c1 && c1->op == CPN_OP_CONV && c2->op == CPN_OP_CONV &&
                      c1->subpixel == CPN_SUBPIXEL_SCATTER && c1->dst >= 0 && c1->act == CPN_ACT_RELU && c1->cout_b == 64 &&
                      (c1->cin_b == 32 || c1->cin_b == 64) && c1->bundles == 4 && c1->bias_offset >= 0 &&
                      c2->src0 == c1->dst && c2->src1 < 0 && !c2->up0 && c2->kh == 3 && c2->kw == 3 && c2->stride == 1 &&
                      c2->pad == 1 && c2->bundles == 1 && c2->cin_b == 64 && c2->cout_b == 64 && c2->subpixel == 0 && !c2->alt &&
                      c2->dst >= 0 && c2->res_up != 1 && c2->fuse_cout == 0 && o.src0 == c1->src0 && o.dst == c2->dst &&
                      o.res == c2->res && o.res_up == c2->res_up && o.act == c2->act && o.cin_b == c1->cin_b && o.cout_b == 64 &&
                      o.kh == 3 && o.kw == 3 && o.weight_offset == c1->weight_offset && o.bias_offset == c1->bias_offset &&
                      o.fuse_weight_offset == c2->weight_offset && o.fuse_bias_offset == c2->bias_offset &&
                      p->tensors[o.src0].channels >= o.cin_b;
            for (size_t j = 0; ok && j < p->ops.size(); ++j) {  // nothing else may read the tensor that is no longer stored
                const cpn_op_desc &q = p->ops[j];
                if (j != oi_ - 1 && (q.src0 == c1->dst || q.src1 == c1->dst || q.res == c1->dst)) ok = false;
            }
            if (!ok) {
                delete p;
                return fail(CPN_E_INVALID, "cpn_plan_create: a CPN_OP_CONV_BRIDGE op must follow the scatter conv (32 | 64 -> 64 "
                                           "channels, ReLU) and the 3x3 conv (64 -> 64) it restates, share their offsets, and "
                                           "the tensor between them must have no other reader");
            }
        }
        if (o.op == CPN_OP_ACT && (o.src0 < 0 || o.dst < 0 || o.act < CPN_ACT_RELU || o.act > CPN_ACT_SOFTPLUS || o.act == CPN_ACT_TANH_SCALED ||
                                   p->tensors[o.src0].channels != p->tensors[o.dst].channels)) {
            delete p;
            return fail(CPN_E_INVALID, "cpn_plan_create: an activation op needs source and destination tensors of equal channel count "
                                       "and one of the elementwise activations");
        }
        if (o.op == CPN_OP_BILINEAR && (o.act < 0 || o.act > 1 || (o.act == 1 && p->precision == CPN_PRECISION_FP8))) {
            delete p;
            return fail(o.act == 1 ? CPN_E_UNSUPPORTED : CPN_E_INVALID,
                        "cpn_plan_create: a resize op takes act = 0 (bilinear) or 1 (bicubic; bf16 / fp32 plans only: bicubic weights "
                        "are negative in places, the result leaves the e4m3 range of its source's scale)");
        }
        if ((o.op == CPN_OP_CONV || o.op == CPN_OP_CONV_DEFERRED) && (o.act > CPN_ACT_TANH_SCALED || o.fuse_act > CPN_ACT_TANH_SCALED)) {
            delete p;
            return fail(CPN_E_INVALID, "cpn_plan_create: conv ops take CPN_ACT_NONE .. CPN_ACT_TANH_SCALED (other activations are CPN_OP_ACT ops)");
        }
        if (o.alt < 0 || o.alt > 2) {
            delete p;
            return fail(CPN_E_INVALID, "cpn_plan_create: alt must be 0, 1 or 2");
        }
        if (o.op == CPN_OP_CONV_DEFERRED && (precision != CPN_PRECISION_BF16 || o.fuse_cout <= 0 || o.dst >= 0)) {
            delete p;
            return fail(CPN_E_INVALID, "cpn_plan_create: a deferred conv must be a fused ReadOut head of a bf16 plan");
        }
        if (o.op == CPN_OP_CONV || o.op == CPN_OP_CONV_DEFERRED) {
            size_t wbytes = (size_t) o.bundles * o.cin_b * o.kh * o.kw * o.cout_b * 4;  // fp32 verification layout
            if (precision == CPN_PRECISION_BF16) {  // [bundle][items (+1 zero slab if odd)][cout_b][32] bf16
                const size_t items = (size_t) (o.cin_b / 32) * o.kh * o.kw;
                wbytes = (size_t) o.bundles * (items + (items & 1)) * o.cout_b * 64;
            }
            if (precision == CPN_PRECISION_FP8) {  // [bundle][items (+1 zero slab if odd)][cout_b][64] bytes
                const size_t items = (size_t) (o.cin_b / 64) * o.kh * o.kw;
                wbytes = (size_t) o.bundles * (items + (items & 1)) * o.cout_b * 64;
                if (o.cin_b % 64 || (o.mult_offset >= 0 && (size_t) o.mult_offset + (size_t) (o.subpixel == CPN_SUBPIXEL_BL_PHASE ? 1 : o.bundles) * o.cout_b > bias_count)) {
                    delete p;
                    return fail(CPN_E_INVALID, "cpn_plan_create: fp8 conv needs cin_b % 64 == 0 and a valid mult_offset");
                }
            }
            if (precision == CPN_PRECISION_F32 && o.fuse_cout > 0) {
                delete p;
                return fail(CPN_E_INVALID, "cpn_plan_create: fused heads are a bf16-only feature");
            }
            if (o.weight_offset < 0 || (size_t) o.weight_offset + wbytes > weight_bytes ||
                (o.bias_offset >= 0 && (size_t) o.bias_offset + (size_t) ((o.subpixel == CPN_SUBPIXEL_SCATTER ||
                                                                              o.subpixel == CPN_SUBPIXEL_BL_PHASE) ? 1 : o.bundles) *
                                                                     o.cout_b > bias_count)) {  // (the four phases share one bias)
                delete p;
                return fail(CPN_E_INVALID, "cpn_plan_create: weight/bias offset out of range");
            }
        }
    }
    *plan = p;
    return 0;
}

void cpn_plan_destroy(cpn_plan *plan) { delete plan; }

int64_t cpn_plan_workspace_bytes(cpn_plan *plan, int32_t N, int32_t H, int32_t W) {
    if (!plan || N <= 0 || H <= 0 || W <= 0) {
        fail(CPN_E_INVALID, "cpn_plan_workspace_bytes: N, H and W must be positive");
        return CPN_E_INVALID;
    }
    const ShapePlan &sp = get_shape_plan(plan, N, H, W);
    if (sp.error) return fail(sp.error, sp.message.c_str());
    return sp.total;
}

int cpn_plan_output_dims(cpn_plan *plan, int32_t H, int32_t W, int32_t out_index, int32_t *h, int32_t *w) {
    if (!plan || H <= 0 || W <= 0 || out_index < 0 || out_index >= CPN_NUM_OUTPUTS || !h || !w)
        return fail(CPN_E_INVALID, "cpn_plan_output_dims: bad arguments");
    const ShapePlan &sp = get_shape_plan(plan, 1, H, W);
    if (sp.error) return fail(sp.error, sp.message.c_str());
    *h = sp.out_h[out_index];
    *w = sp.out_w[out_index];
    return 0;
}

int cpn_plan_tensor_info(cpn_plan *plan, int32_t N, int32_t H, int32_t W, int32_t tensor, int64_t *byte_offset,
                         int32_t *h, int32_t *w, int32_t *channel_stride) {
    if (!plan || N <= 0 || H <= 0 || W <= 0 || tensor < 0 || tensor >= (int) plan->tensors.size() || !byte_offset || !h ||
        !w || !channel_stride)
        return fail(CPN_E_INVALID, "cpn_plan_tensor_info: bad arguments");
    const ShapePlan &sp = get_shape_plan(plan, N, H, W);
    if (sp.error) return fail(sp.error, sp.message.c_str());
    if (sp.offsets[tensor] < 0) return fail(CPN_E_INVALID, "cpn_plan_tensor_info: the tensor is never written");
    *byte_offset = sp.offsets[tensor];
    *h = sp.th[tensor];
    *w = sp.tw[tensor];
    *channel_stride = plan->tensors[tensor].channels;
    return 0;
}

int64_t cpn_plan_max_tensor_elements(cpn_plan *plan, int32_t H, int32_t W) {
    if (!plan || H <= 0 || W <= 0) {
        fail(CPN_E_INVALID, "cpn_plan_max_tensor_elements: bad arguments");
        return CPN_E_INVALID;
    }
    const ShapePlan &sp = get_shape_plan(plan, 1, H, W);
    if (sp.error) return fail(sp.error, sp.message.c_str());
    return sp.max_elems;
}

static int run_or_count(cpn_plan *plan, const void *input, int32_t in_dtype, int32_t N, int32_t H, int32_t W,
                        void *workspace, int64_t workspace_bytes, float *const *outputs, int32_t *range_flag,
                        hipStream_t st, double *flops, hipEvent_t *events = nullptr, double *op_flops = nullptr,
                        float *absmax = nullptr) {
    if (!plan || N <= 0 || H <= 0 || W <= 0) return fail(CPN_E_INVALID, "cpn_plan_run: N, H and W must be positive");
    const ShapePlan &sp = get_shape_plan(plan, N, H, W);
    if (sp.error) return fail(sp.error, sp.message.c_str());
    if (!flops && sp.total > workspace_bytes) return fail(CPN_E_WORKSPACE, "cpn_plan_run: workspace too small");
    const bool f32 = plan->precision == CPN_PRECISION_F32, fp8 = plan->precision == CPN_PRECISION_FP8;
    if (absmax && plan->precision != CPN_PRECISION_BF16) return fail(CPN_E_INVALID, "cpn_plan_run_stats: bf16 plans only");
    // (FLOP-count mode runs without a workspace: a non-null dummy base keeps "tensor at offset 0" distinguishable from "no tensor"
    //  in the argument checks -- nothing is launched in that mode)
    char *ws = (flops && !workspace) ? (char *) 256 : (char *) workspace;
    auto tptr = [&](int t) -> void * { return t >= 0 ? (void *) (ws + sp.offsets[t]) : nullptr; };
    auto tch = [&](int t) -> int { return t >= 0 ? plan->tensors[t].channels : 0; };
    for (size_t i = 0; i < plan->ops.size(); ++i) {
        const cpn_op_desc &o = plan->ops[i];
        int rc = 0;
        if (events) (void) hipEventRecord(events[i], st);
        if (sp.skip[i]) continue;
        switch (o.op) {
            case CPN_OP_INPUT: {
                if (flops) break;
                InputArgs a{input, tptr(o.dst), N, o.in_channels, H, W, tch(o.dst), in_dtype, range_flag};
                rc = check_hip((hipError_t) (f32 ? launch_input_f32(a, st)
                                                 : fp8 ? launch_input_fp8(a, 1.f / plan->tensors[o.dst].scale, st)
                                                       : launch_input(a, st)), "input kernel");
                break;
            }
            case CPN_OP_INPUT_STEM: {
                if (flops) break;
                InputArgs a{input, tptr(o.dst), N, o.in_channels, H, W, 4, in_dtype, range_flag};
                rc = check_hip((hipError_t) launch_input_stem(a, st), "stem input kernel");
                break;
            }
            case CPN_OP_STEM7: {
                StemArgs a{tptr(o.src0), tptr(o.dst), plan->weights + o.weight_offset,
                           o.bias_offset >= 0 ? plan->bias + o.bias_offset : nullptr, N, sp.th[o.src0], sp.tw[o.src0],
                           sp.th[o.dst], sp.tw[o.dst], o.cout_b, tch(o.dst),
                           fp8 ? 1.f / plan->tensors[o.dst].scale : 0.f};  // fp8 plans: e4m3 output codes
                const double fl = 2.0 * N * a.Hout * a.Wout * (double) o.cout_b * 7 * 32;
                if (op_flops) op_flops[i] = fl;
                if (flops) { *flops += fl; break; }
                rc = check_hip((hipError_t) launch_stem7(a, st), "stem conv kernel");
                break;
            }
            case CPN_OP_MAXPOOL: {
                if (flops) break;
                PoolArgs a{tptr(o.src0), tptr(o.dst), N, sp.th[o.src0], sp.tw[o.src0], sp.th[o.dst], sp.tw[o.dst],
                           tch(o.src0), o.kh, o.stride, o.pad};
                rc = check_hip((hipError_t) (f32 ? launch_maxpool_f32(a, st) : fp8 ? launch_maxpool_fp8(a, st)
                                                                                  : launch_maxpool(a, st)), "maxpool kernel");
                break;
            }
            case CPN_OP_ACT: {
                if (flops) break;
                ActArgs a{tptr(o.src0), tptr(o.dst), (long) N * sp.th[o.src0] * sp.tw[o.src0] * tch(o.src0), o.act,
                          fp8 ? plan->tensors[o.src0].scale : 1.f, fp8 ? 1.f / plan->tensors[o.dst].scale : 1.f};
                rc = check_hip((hipError_t) (f32 ? launch_act_f32(a, st) : fp8 ? launch_act_fp8(a, st) : launch_act(a, st)),
                               "activation kernel");
                break;
            }
            case CPN_OP_BILINEAR: {
                if (flops) break;
                if (sp.offsets[o.src0] == sp.offsets[o.dst]) break;  // same size: the planner aliased dst to src
                ResizeArgs a{tptr(o.src0), tptr(o.dst), N, sp.th[o.src0], sp.tw[o.src0], sp.th[o.dst], sp.tw[o.dst],
                             tch(o.src0), fp8 ? sp.ring[i] : 0,  // (ring: see propagate_dims, bilinear sub-pixel triple)
                             o.act == 1 ? 1 : 0};                // (CPN_OP_BILINEAR: act = 1 selects bicubic)
                rc = check_hip((hipError_t) (f32 ? launch_bilinear_f32(a, st) : fp8 ? launch_bilinear_fp8(a, st)
                                                                                   : launch_bilinear(a, st)), "bilinear kernel");
                break;
            }
            case CPN_OP_CONV_PAIR: {
                const int mid = plan->ops[i - 2].dst;
                PairArgs a = pair_args(plan, o, N, sp.th[mid], sp.tw[mid]);
                a.src = tptr(o.src0);
                a.dst = tptr(o.dst);
                const double fl = conv_pair_executed_flops(a);
                if (op_flops) op_flops[i] = fl;
                if (flops) { *flops += fl; break; }
                rc = check_hip((hipError_t) launch_conv_pair(a, st), "conv pair kernel");
                break;
            }
            case CPN_OP_CONV_BRIDGE: {
                ConvArgs a;
                rc = bridge_args(plan, o, plan->ops[i - 1], N, sp.th[o.src0], sp.tw[o.src0], a, tptr(o.src0), tch(o.src0), tptr(o.res),
                                 tch(o.res), tptr(o.dst), tch(o.dst), plan->weights, plan->bias);
                if (rc) return rc;
                if (!conv_bridge_supported(a)) return fail(CPN_E_INVALID, "cpn_plan_run: bridge op at an unsupported size");
                const double fl = bridge_executed_flops(a);
                if (op_flops) op_flops[i] = fl;
                if (flops) { *flops += fl; break; }
                rc = check_hip((hipError_t) launch_conv(a, st), "conv bridge kernel");
                break;
            }
            case CPN_OP_CONV_DEFERRED: break;  // evaluated at the proposal pixels only (cpn_sparse_heads)
            case CPN_OP_CONV: {
                int Hin, Win;  // virtual input size (see propagate_dims)
                if (o.up0 == 2) { Hin = H; Win = W; }
                else if (o.up1) { Hin = sp.th[o.src0]; Win = sp.tw[o.src0]; }
                else if (o.up0 && o.src1 >= 0) { Hin = sp.th[o.src1]; Win = sp.tw[o.src1]; }
                else if (o.up0) { Hin = 2 * sp.th[o.src0]; Win = 2 * sp.tw[o.src0]; }
                else { Hin = sp.th[o.src0]; Win = sp.tw[o.src0]; }
                const int sdims[6] = {sp.th[o.src0], sp.tw[o.src0], o.src1 >= 0 ? sp.th[o.src1] : 0,
                                      o.src1 >= 0 ? sp.tw[o.src1] : 0, o.res >= 0 ? sp.th[o.res] : 0,
                                      o.res >= 0 ? sp.tw[o.res] : 0};
                void *dst = o.dst >= 0 ? tptr(o.dst) : (outputs ? (void *) outputs[o.out_index] : nullptr);
                ConvArgs a;
                rc = build_conv_args(plan, o, N, a, tptr(o.src0), tch(o.src0), tptr(o.src1), tch(o.src1),
                                     tptr(o.res), tch(o.res), dst, o.dst >= 0 ? tch(o.dst) : 0, Hin, Win, sdims);
                if (rc) return rc;
                if (o.dst >= 0) {
                    const int64_t M = (int64_t) N * sp.th[o.dst] * sp.tw[o.dst];
                    if ((int64_t) a.N * a.Hout * a.Wout * (a.phase == 2 ? 4 : 1) != M)
                        return fail(CPN_E_INVALID, "cpn_plan_run: conv output size mismatch");
                }
                if (op_flops) op_flops[i] = conv_executed_flops(a);
                if (flops) { *flops += conv_executed_flops(a); break; }
                if (!dst) return fail(CPN_E_INVALID, "cpn_plan_run: missing external output buffer");
                rc = check_hip((hipError_t) (f32 ? launch_conv_f32(a, st) : fp8 ? launch_conv_fp8(a, st) : launch_conv(a, st)),
                               "conv kernel");
                break;
            }
            default: return fail(CPN_E_INVALID, "cpn_plan_run: unknown op");
        }
        if (rc) return rc;
        if (absmax && o.dst >= 0 && o.op != CPN_OP_INPUT_STEM) {  // calibration: max |x| of the tensor this op produced
            // (the padded stem layout does not fill the input tensor's storage: its scale is set by the caller -- inputs
            // lie in [0, 1])
            const cpn_tensor_desc &t = plan->tensors[o.dst];
            const long count = (long) N * sp.th[o.dst] * sp.tw[o.dst] * t.channels;
            rc = check_hip((hipError_t) launch_absmax_bf16(tptr(o.dst), count, absmax + o.dst, st), "absmax kernel");
            if (rc) return rc;
        }
    }
    if (events) (void) hipEventRecord(events[plan->ops.size()], st);
    return 0;
}

int cpn_plan_num_ops(cpn_plan *plan) { return plan ? (int) plan->ops.size() : 0; }

int cpn_plan_run_timed(cpn_plan *plan, const void *input, int32_t in_dtype, int32_t N, int32_t H, int32_t W,
                       void *workspace, int64_t workspace_bytes, float *const *outputs, int32_t *range_flag,
                       void *stream, float *op_ms, double *op_flops) {
    if (!plan || !op_ms) return fail(CPN_E_INVALID, "cpn_plan_run_timed: null");
    const size_t n = plan->ops.size();
    std::vector<hipEvent_t> ev(n + 1);
    for (auto &e : ev)
        if (hipEventCreate(&e) != hipSuccess) return fail(CPN_E_INVALID, "cpn_plan_run_timed: event create failed");
    if (op_flops) std::fill(op_flops, op_flops + n, 0.);
    int rc = run_or_count(plan, input, in_dtype, N, H, W, workspace, workspace_bytes, outputs, range_flag,
                          (hipStream_t) stream, nullptr, ev.data(), op_flops);
    if (!rc) rc = check_hip(hipEventSynchronize(ev[n]), "cpn_plan_run_timed: sync");
    for (size_t i = 0; i < n && !rc; ++i) (void) hipEventElapsedTime(&op_ms[i], ev[i], ev[i + 1]);
    for (auto &e : ev) (void) hipEventDestroy(e);
    return rc;
}

int cpn_plan_run(cpn_plan *plan, const void *input, int32_t in_dtype, int32_t N, int32_t H, int32_t W,
                 void *workspace, int64_t workspace_bytes, float *const *outputs, int32_t *range_flag, void *stream) {
    return run_or_count(plan, input, in_dtype, N, H, W, workspace, workspace_bytes, outputs, range_flag,
                        (hipStream_t) stream, nullptr);
}

int cpn_plan_run_stats(cpn_plan *plan, const void *input, int32_t in_dtype, int32_t N, int32_t H, int32_t W,
                       void *workspace, int64_t workspace_bytes, float *const *outputs, int32_t *range_flag,
                       float *absmax, void *stream) {
    if (!absmax) return fail(CPN_E_INVALID, "cpn_plan_run_stats: null absmax");
    return run_or_count(plan, input, in_dtype, N, H, W, workspace, workspace_bytes, outputs, range_flag,
                        (hipStream_t) stream, nullptr, nullptr, nullptr, absmax);
}

double cpn_plan_executed_flops(cpn_plan *plan, int32_t N, int32_t H, int32_t W) {
    double f = 0.;
    if (run_or_count(plan, nullptr, 0, N, H, W, nullptr, 0, nullptr, nullptr, nullptr, &f)) return -1.;
    return f;
}

int cpn_conv2d(const cpn_op_desc *op, const void *src0, int32_t c0_stride, const void *src1, int32_t c1_stride,
               const void *res, int32_t res_stride, void *dst, int32_t dst_stride, int32_t N, int32_t Hin, int32_t Win,
               const void *weights, const float *bias, void *stream) {
    if (!op || !src0 || !dst || !weights) return fail(CPN_E_INVALID, "cpn_conv2d: null pointer");
    ConvArgs a;
    int rc = build_conv_args(nullptr, *op, N, a, src0, c0_stride, src1, c1_stride, res, res_stride, dst,
                             dst_stride, Hin, Win);
    if (rc) return rc;
    a.weights = (const unsigned char *) weights + op->weight_offset;
    a.bias = (bias && op->bias_offset >= 0) ? bias + op->bias_offset : nullptr;
    if (op->fuse_cout > 0) {
        a.fuse_w = (const unsigned char *) weights + op->fuse_weight_offset;
        a.fuse_b = (bias && op->fuse_bias_offset >= 0) ? bias + op->fuse_bias_offset : nullptr;
    }
    return check_hip((hipError_t) launch_conv(a, (hipStream_t) stream), "cpn_conv2d");
}

int cpn_conv2d_fp8(const cpn_op_desc *op, const void *src0, int32_t c0_stride, const void *src1, int32_t c1_stride,
                   const void *res, int32_t res_stride, void *dst, int32_t dst_stride, int32_t N, int32_t Hin,
                   int32_t Win, const void *weights, const float *bias, const float *mult, float res_scale,
                   float out_inv_scale, void *stream) {
    if (!op || !src0 || !dst || !weights) return fail(CPN_E_INVALID, "cpn_conv2d_fp8: null pointer");
    if (op->cin_b % 64 || op->c0_used % 64 || c0_stride % 64 || (src1 && c1_stride % 64))
        return fail(CPN_E_INVALID, "cpn_conv2d_fp8: input channel counts / strides must be multiples of 64");
    ConvArgs a;
    int rc = build_conv_args(nullptr, *op, N, a, src0, c0_stride, src1, c1_stride, res, res_stride, dst,
                             dst_stride, Hin, Win);
    if (rc) return rc;
    a.weights = (const unsigned char *) weights + op->weight_offset;
    a.bias = (bias && op->bias_offset >= 0) ? bias + op->bias_offset : nullptr;
    a.mult = (mult && op->bias_offset >= 0) ? mult + op->bias_offset : mult;
    a.res_scale = res_scale;
    a.out_inv_scale = out_inv_scale;
    if (op->fuse_cout > 0) {
        a.fuse_w = (const unsigned char *) weights + op->fuse_weight_offset;
        a.fuse_b = (bias && op->fuse_bias_offset >= 0) ? bias + op->fuse_bias_offset : nullptr;
    }
    return check_hip((hipError_t) launch_conv_fp8(a, (hipStream_t) stream), "cpn_conv2d_fp8");
}

int cpn_conv_pair(const cpn_op_desc *op, const void *src, int32_t c_stride, void *dst, int32_t dst_stride, int32_t N,
                  int32_t H, int32_t W, const void *weights, const float *bias, void *stream) {
    if (!op || !src || !dst || !weights || op->op != CPN_OP_CONV_PAIR || N <= 0 || H <= 0 || W <= 0)
        return fail(CPN_E_INVALID, "cpn_conv_pair: needs a CPN_OP_CONV_PAIR descriptor and non-null buffers");
    PairArgs a = pair_args(nullptr, *op, N, H, W);
    a.src = src; a.c_stride = c_stride; a.dst = dst; a.dst_stride = dst_stride;
    a.w1 = (const unsigned char *) weights + op->weight_offset;
    a.w2 = (const unsigned char *) weights + op->fuse_weight_offset;
    a.b1 = (bias && op->bias_offset >= 0) ? bias + op->bias_offset : nullptr;
    a.b2 = (bias && op->fuse_bias_offset >= 0) ? bias + op->fuse_bias_offset : nullptr;
    if (!conv_pair_supported(a))
        return fail(CPN_E_UNSUPPORTED, "cpn_conv_pair: needs W = 16 or W >= 32, conv1 output channels a multiple of 256 (128 at "
                                       "W > 32 and for a stride-2 conv2), conv2 bundles of 32 | 64 channels (32 on stride-1 generic tiles)");
    return check_hip((hipError_t) launch_conv_pair(a, (hipStream_t) stream), "cpn_conv_pair");
}

int cpn_conv_bridge(const cpn_op_desc *op, const void *src, int32_t c_stride, const void *res, int32_t res_stride, void *dst,
                    int32_t dst_stride, int32_t N, int32_t H, int32_t W, const void *weights, const float *bias, void *stream) {
    if (!op || !src || !dst || !weights || op->op != CPN_OP_CONV_BRIDGE || N <= 0 || H <= 0 || W <= 0)
        return fail(CPN_E_INVALID, "cpn_conv_bridge: needs a CPN_OP_CONV_BRIDGE descriptor and non-null buffers");
    cpn_op_desc c2 = *op;  // the 3x3 conv the op restates: one plain 64-channel source, its own weights behind fuse_*_offset
    c2.op = CPN_OP_CONV; c2.src1 = -1; c2.up0 = c2.up1 = 0; c2.c0_used = 64; c2.cin_b = 64; c2.cout_b = 64; c2.bundles = 1;
    c2.kh = c2.kw = 3; c2.stride = 1; c2.pad = 1; c2.subpixel = 0; c2.fuse_cout = 0; c2.dst = 0; c2.dst_coff = 0;
    ConvArgs a;
    int rc = bridge_args(nullptr, *op, c2, N, H, W, a, src, c_stride, res, res_stride, dst, dst_stride, weights, bias);
    if (rc) return rc;
    if (!conv_bridge_supported(a))
        return fail(CPN_E_UNSUPPORTED, "cpn_conv_bridge: needs 32 | 64 input channels, 64 output channels and an output of at "
                                       "least 16 x 32 pixels (run the two convs)");
    return check_hip((hipError_t) launch_conv(a, (hipStream_t) stream), "cpn_conv_bridge");
}

int cpn_convert_input_stem(const void *src, int32_t in_dtype, void *dst, int32_t N, int32_t C, int32_t H, int32_t W,
                           int32_t *range_flag, void *stream) {
    if (!src || !dst || N <= 0 || H <= 0 || W <= 0 || C < 1 || C > 4 || (in_dtype != 0 && in_dtype != 1))
        return fail(CPN_E_INVALID, "cpn_convert_input_stem: bad arguments (1..4 channels, dtype 0 = f32 | 1 = u8)");
    InputArgs a{src, dst, N, C, H, W, 4, in_dtype, range_flag};
    return check_hip((hipError_t) launch_input_stem(a, (hipStream_t) stream), "cpn_convert_input_stem");
}

int cpn_stem7(const cpn_op_desc *op, const void *src, void *dst, int32_t dst_stride, int32_t N, int32_t H, int32_t W,
              const void *weights, const float *bias, float out_inv_scale, void *stream) {
    if (!op || !src || !dst || !weights || N <= 0 || H <= 0 || W <= 0) return fail(CPN_E_INVALID, "cpn_stem7: bad arguments");
    if (op->op != CPN_OP_STEM7 || (op->cout_b != 32 && op->cout_b != 64) || dst_stride < op->cout_b || dst_stride % 8)
        return fail(CPN_E_INVALID, "cpn_stem7: needs a CPN_OP_STEM7 descriptor with 32 | 64 output channels");
    StemArgs a{src, dst, (const unsigned char *) weights + op->weight_offset,
               (bias && op->bias_offset >= 0) ? bias + op->bias_offset : nullptr, N, H, W, (H - 1) / 2 + 1, (W - 1) / 2 + 1,
               op->cout_b, dst_stride, out_inv_scale > 0.f ? out_inv_scale : 0.f};
    return check_hip((hipError_t) launch_stem7(a, (hipStream_t) stream), "cpn_stem7");
}

int cpn_maxpool2d(const void *src, void *dst, int32_t N, int32_t Hin, int32_t Win, int32_t C, int32_t k, int32_t stride,
                  int32_t pad, void *stream) {
    if (C % 8) return fail(CPN_E_INVALID, "cpn_maxpool2d: C must be a multiple of 8");
    PoolArgs a{src, dst, N, Hin, Win, (Hin + 2 * pad - k) / stride + 1, (Win + 2 * pad - k) / stride + 1, C, k, stride, pad};
    return check_hip((hipError_t) launch_maxpool(a, (hipStream_t) stream), "cpn_maxpool2d");
}

int cpn_resize_bilinear(const void *src, void *dst, int32_t N, int32_t Hin, int32_t Win, int32_t Hout, int32_t Wout,
                        int32_t C, void *stream) {
    if (C % 8) return fail(CPN_E_INVALID, "cpn_resize_bilinear: C must be a multiple of 8");
    ResizeArgs a{src, dst, N, Hin, Win, Hout, Wout, C};
    return check_hip((hipError_t) launch_bilinear(a, (hipStream_t) stream), "cpn_resize_bilinear");
}

int cpn_convert_input(const void *src, int32_t in_dtype, void *dst, int32_t N, int32_t C, int32_t H, int32_t W,
                      int32_t Cpad, int32_t *range_flag, void *stream) {
    if (Cpad % 8 || Cpad < C) return fail(CPN_E_INVALID, "cpn_convert_input: bad Cpad");
    InputArgs a{src, dst, N, C, H, W, Cpad, in_dtype, range_flag};
    return check_hip((hipError_t) launch_input(a, (hipStream_t) stream), "cpn_convert_input");
}

int cpn_histogram(const void *x, int32_t dtype, int64_t n, uint32_t *hist, void *stream) {
    if (!x || !hist || n < 0 || (dtype != 1 && dtype != 2)) return fail(CPN_E_INVALID, "cpn_histogram: dtype 1 (u8) or 2 (u16)");
    if (n == 0) return 0;
    return check_hip((hipError_t) launch_histogram(x, dtype, (long) n, hist, (hipStream_t) stream), "cpn_histogram");
}

int cpn_window_any(const void *mask, int32_t dtype, int32_t H, int32_t W, const int32_t *windows, int32_t n, int32_t *out,
                   void *stream) {
    if (!mask || (n > 0 && (!windows || !out)) || n < 0 || H <= 0 || W <= 0 || (dtype != 0 && dtype != 1))
        return fail(CPN_E_INVALID, "cpn_window_any: dtype 0 (f32) or 1 (u8 / bool), H, W > 0");
    return check_hip((hipError_t) launch_window_any(mask, dtype, W, windows, n, out, (hipStream_t) stream), "cpn_window_any");
}

int cpn_rescale_to_uint8(const void *x, int32_t dtype, int64_t n, double low, double high, uint8_t *out, void *stream) {
    if (!x || !out || n < 0 || dtype < 0 || dtype > 2) return fail(CPN_E_INVALID, "cpn_rescale_to_uint8: dtype 0 (f32), 1 (u8), 2 (u16)");
    if (!(high > low)) return fail(CPN_E_INVALID, "cpn_rescale_to_uint8: needs high > low");
    if (n == 0) return 0;
    return check_hip((hipError_t) launch_rescale_u8(x, dtype, (long) n, low, high, out, (hipStream_t) stream), "cpn_rescale_to_uint8");
}

}  // extern "C"
