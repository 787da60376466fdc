// Score-gated ReadOut heads for gfx950 (MI355X / CDNA4): the location and the Fourier head evaluated ONLY at the
// proposal pixels.
//
// Reference: CPNCore.forward runs every ReadOut head (celldetection/models/commons.py:461-511: conv k x k -> BN ->
// ReLU -> conv 1 x 1) densely over the head grid (celldetection/models/cpn.py:238-283), and CPN.forward then reads the
// location / Fourier maps at the pixels whose score passed the threshold only (`fg_mask`, cpn.py:613-637).  With P
// proposals on an N x h x w grid the dense maps are (N h w) / P times more work than the output dict needs: 20x for the
// BASELINE configs[2] batch (51 k proposals on 16 x 256^2 pixels), 38 % of all FLOPs of the graph.
//
// This kernel computes, for the proposal pixels `indices` (the output of cpn_compact, row-major order), exactly the
// values the dense fused heads (conv_igemm.hip, OUT_FUSED_HEAD) write at those pixels: same bf16 operands, same K order
// (32-channel chunk major, tap minor, k-half 0 then 1) on v_mfma_f32_32x32x16_bf16 with fp32 accumulation, same
// bias / ReLU / bf16 rounding of the hidden activation, same second MFMA GEMM and final activation -- the results are
// bit-identical to gathering the dense maps (tests/test_gpu_sparse_heads.py).
//
// Design: a "gathered implicit GEMM".  One workgroup = 128 proposals x (2 heads x HID hidden units), 8 waves; wave w owns
// all 128 proposals x (HID/4) hidden units of head w >> 2 (128 fp32 accumulator registers, like the dense kernel).
//   * pixel operand: per K item (chunk, tap) every wave issues ONE LDS-DMA gather instruction (16 proposals x 64 B through
//     a raw buffer descriptor: per-lane offset = proposal centre + tap delta, out-of-image taps are out-of-range lanes =
//     zeros = the conv's zero padding) into a ring of four 8-KiB slots; two items per pipeline step, the gathers of step
//     s+2 are issued when step s has been consumed;
//   * weight operand: straight from L2 into registers in MFMA layout (the packed conv weights of the two heads are used
//     as they are; four register sets = item 0/1 x k-half 0/1, refilled as soon as their MFMAs have issued; vmcnt
//     tracked by the compiler) -- no weight tiles in LDS, every wave reads only its own 64 rows;
//   * epilogue: relu(acc + bias) -> bf16 -> LDS [128][HID] per head (XOR-swizzled 16-byte slots) -> D2[32][32 proposals]
//     = W2[32][HID] x X (one (head, 32-proposal fragment) per wave) -> + bias2 -> activation -> out[head][proposal][c].
#include <hip/hip_runtime.h>

#include <atomic>

#include "../../include/cpn_hip.h"
#include "cpn_error.h"
#include "cpn_kernels.h"

namespace cpn {
namespace sparse {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
typedef __attribute__((ext_vector_type(2))) float f32x2_t;
typedef bf16x8 frag_t;

constexpr int REC = 64;        // bytes per 32-channel record (pixel or weight row)
constexpr int MT = 128;        // proposals per workgroup
constexpr int WM = MT / 32;    // pixel fragments per wave
constexpr int NWAVES = 8;
constexpr int SLOT = MT * REC; // one K item's pixel tile
constexpr unsigned OOB_LANE = 0x80000000u;

typedef __amdgpu_buffer_rsrc_t rsrc_t;
__device__ __forceinline__ rsrc_t make_rsrc(const void *base, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc((void *) base, 0, (int) bytes, 0x00020000);
}
__device__ __forceinline__ void bdma16(rsrc_t rsrc, unsigned lane_off, unsigned scalar_off, unsigned char *lds_wave_base) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void *) lds_wave_base, 16,
                                             (int) lane_off, (int) scalar_off, 0, 0);
}
__device__ __forceinline__ unsigned int pack_bf16x2(float lo, float hi) {
    const f32x2_t v = {lo, hi};
    return __builtin_bit_cast(unsigned int, __builtin_convertvector(v, bf16x2_t));
}
template <int IMM>
__device__ __forceinline__ void ds_read16(frag_t &d, unsigned addr) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "n"(IMM));
}
__device__ __forceinline__ void load_pfrags(frag_t (&p)[WM], unsigned paddr) {
    ds_read16<0>(p[0], paddr);
    ds_read16<32 * REC>(p[1], paddr);
    ds_read16<64 * REC>(p[2], paddr);
    ds_read16<96 * REC>(p[3], paddr);
}
template <int N>
__device__ __forceinline__ void wait_pfrags(frag_t (&p)[WM]) {
    asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(p[0]), "+v"(p[1]), "+v"(p[2]), "+v"(p[3]) : "n"(N));
}

struct Head {
    const void *w1;   // packed conv weights [items (+1 zero item if odd)][HID][32] bf16 (chunk-major, tap-minor)
    const float *b1;  // [HID] folded BN bias
    const void *w2;   // [32][HID] bf16 (rows >= cout are zero)
    const float *b2;  // [32] or nullptr
    float *out;       // [P][cout] fp32
    int cout, act;
    float act_scale;
};
struct Args {
    const void *feat;  // NHWC bf16 [N][h][w][cs]
    int N, h, w, cs, cin, K, pad;
    const int *indices;
    int P;
    Head head[2];
};

template <int HID>
__global__ __launch_bounds__(64 * NWAVES) void sparse_heads_kernel(const Args a) {
    constexpr int WN = HID / 128;  // 32-row weight fragments per wave: a head's HID hidden units over four waves
    static_assert(HID == 256 || HID == 128, "hidden width");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int hd = wave >> 2, wq = wave & 3;  // head, quarter of its hidden units
    const int l31 = lane & 31, lhi = lane >> 5;
    const int p0 = blockIdx.x * MT;
    const Head &H = a.head[hd];

    const int ntaps = a.K * a.K;
    const int nchunks = a.cin / 32;
    const int nreal = nchunks * ntaps;
    const int nitems = nreal + (nreal & 1);
    const int nsteps = nitems >> 1;
    const unsigned item_bytes = (unsigned) HID * REC;

    // ---- gather geometry: this lane's 16 B of a gather instruction = proposal wave*16 + (lane >> 2), 16-byte part
    // (lane & 3) ^ ((lane >> 4) & 3) of its 64-byte record (the XOR swizzle that makes the fragment reads conflict-free)
    const int gp = p0 + wave * 16 + (lane >> 2);
    int gy = -0x40000000, gx = -0x40000000;  // proposals past the end: every tap out of range
    unsigned gcentre = 0;
    if (gp < a.P) {
        const int lin = a.indices[gp];
        const int hw = a.h * a.w;
        const int b = lin / hw, rem = lin - b * hw;
        gy = rem / a.w;
        gx = rem - gy * a.w;
        gcentre = (unsigned) (((b * a.h + gy) * a.w + gx) * a.cs) * 2u + (unsigned) (((lane & 3) ^ ((lane >> 4) & 3)) << 4);
    }
    // (descriptor words must be SGPRs: readfirstlane keeps the compiler from computing the sizes on the vector ALU next to
    // the per-lane index arithmetic, which turns every DMA into a waterfall loop)
    const rsrc_t rsf = make_rsrc(a.feat, (unsigned) __builtin_amdgcn_readfirstlane(a.N * a.h * a.w * a.cs * 2));
    const rsrc_t rsw = make_rsrc(H.w1, (unsigned) __builtin_amdgcn_readfirstlane((int) (nitems * item_bytes)));
    unsigned char *const gdst = smem + (wave << 10);  // this wave's 1 KiB of a slot
    // item I (chunk c, tap (ky, kx)) -> ring slot I & 3; items past the real ones (the zero weight slab): zeros
#define GATHER(I, C_, KY_, KX_)                                                                                \
    {                                                                                                          \
        const int dy_ = (KY_) - a.pad, dx_ = (KX_) - a.pad;                                                    \
        const bool ok_ = (I) < nreal && (unsigned) (gy + dy_) < (unsigned) a.h && (unsigned) (gx + dx_) < (unsigned) a.w; \
        const unsigned v_ = ok_ ? gcentre + (unsigned) ((dy_ * a.w + dx_) * a.cs * 2) : OOB_LANE;               \
        bdma16(rsf, v_, (unsigned) __builtin_amdgcn_readfirstlane((C_) * 64), gdst + ((I) & 3) * SLOT);         \
    }
    struct It { int c, ky, kx; };
    auto next = [&](It i) {
        if (++i.kx == a.K) { i.kx = 0; if (++i.ky == a.K) { i.ky = 0; ++i.c; } }
        return i;
    };

    // ---- accumulators
    f32x16 acc[WN][WM];
#pragma unroll
    for (int j = 0; j < WN; ++j)
#pragma unroll
        for (int f = 0; f < WM; ++f)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][f][r] = 0.f;

    // weight fragment (A operand): row wq*WN*32 + j*32 + l31 of the head's slab, 16-byte part 2*khalf + lhi (unswizzled in
    // global memory); pixel fragment (B operand): record f*32 + l31 of the slot, part slot-swizzled like the gather
    const unsigned wl0 = (unsigned) ((wq * WN * 32 + l31) * REC + (lhi << 4)), wl1 = wl0 + 32u;
    const unsigned lds0 = (unsigned) (size_t) (__attribute__((address_space(3))) unsigned char *) smem;
    const unsigned pl = lds0 + (unsigned) (l31 * REC + ((lhi ^ ((l31 >> 2) & 3)) << 4));
#define VLOADW(DST, WL, SOFF)                                                                                  \
    _Pragma("unroll") for (int j = 0; j < WN; ++j)                                                             \
        DST[j] = __builtin_bit_cast(frag_t, __builtin_amdgcn_raw_buffer_load_b128(rsw, (int) ((WL) + j * 32 * REC), (int) (SOFF), 0))
#define MMAP(WF, PF, PENDING)                                                                                  \
    {                                                                                                          \
        wait_pfrags<PENDING>(PF);                                                                              \
        _Pragma("unroll") for (int j = 0; j < WN; ++j)                                                         \
            _Pragma("unroll") for (int f = 0; f < WM; ++f)                                                     \
                acc[j][f] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(WF[j], PF[f], acc[j][f], 0, 0, 0);         \
    }

    // ---- prologue: items 0, 1 gathered and visible; items 2, 3 and the first four weight sets in flight
    It g0{0, 0, 0};            // next item to gather
    It g1 = next(g0);
    GATHER(0, g0.c, g0.ky, g0.kx);
    GATHER(1, g1.c, g1.ky, g1.kx);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    g0 = next(g1); g1 = next(g0);
    GATHER(2, g0.c, g0.ky, g0.kx);
    GATHER(3, g1.c, g1.ky, g1.kx);
    asm volatile("" ::: "memory");  // (compiler barrier: the weight loads below stay YOUNGER than the gathers, see the boundary)
    frag_t W0a[WN], W0b[WN], W1a[WN], W1b[WN], pA[WM], pB[WM];
    unsigned so = 0;  // slab of the first item of the current step
    VLOADW(W0a, wl0, so); VLOADW(W0b, wl1, so);
    VLOADW(W1a, wl0, so + item_bytes); VLOADW(W1b, wl1, so + item_bytes);
    load_pfrags(pA, pl);  // item 0, k-half 0
    for (int st = 0; st + 1 < nsteps; ++st) {
        const unsigned sA = (unsigned) (((2 * st) & 3) * SLOT), sB = (unsigned) (((2 * st + 1) & 3) * SLOT);
        so += 2 * item_bytes;
        load_pfrags(pB, (pl + sA) ^ 32u);        // item 2st, k-half 1
        MMAP(W0a, pA, WM);
        VLOADW(W0a, wl0, so);
        load_pfrags(pA, pl + sB);                 // item 2st+1, k-half 0
        MMAP(W0b, pB, WM);
        VLOADW(W0b, wl1, so);
        load_pfrags(pB, (pl + sB) ^ 32u);        // item 2st+1, k-half 1
        MMAP(W1a, pA, WM);
        VLOADW(W1a, wl0, so + item_bytes);
        // step boundary: all my reads of step st returned; my gathers of step st+1 (issued one step ago = 4 * WN weight
        // loads older than now: VMEM returns in order) landed; after the barrier that holds for every wave
        wait_pfrags<0>(pB);
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * WN) : "memory");
        __builtin_amdgcn_s_barrier();
        if (st + 2 < nsteps) {  // step st+2 into the two slots step st occupied
            g0 = next(g1); g1 = next(g0);
            GATHER(2 * st + 4, g0.c, g0.ky, g0.kx);
            GATHER(2 * st + 5, g1.c, g1.ky, g1.kx);
        }
        asm volatile("" ::: "memory");  // the four weight-load groups up to the next boundary are issued after the gathers
        load_pfrags(pA, pl + (unsigned) (((2 * st + 2) & 3) * SLOT));  // item 2st+2, k-half 0
        MMAP(W1b, pB, WM);
        VLOADW(W1b, wl1, so + item_bytes);
        wait_pfrags<0>(pA);  // nothing of the LDS reads is in flight across the loop back-edge
    }
    {
        const unsigned sA = (unsigned) (((2 * (nsteps - 1)) & 3) * SLOT), sB = (unsigned) (((2 * (nsteps - 1) + 1) & 3) * SLOT);
        load_pfrags(pB, (pl + sA) ^ 32u);
        MMAP(W0a, pA, WM);
        load_pfrags(pA, pl + sB);
        MMAP(W0b, pB, WM);
        load_pfrags(pB, (pl + sB) ^ 32u);
        MMAP(W1a, pA, WM);
        MMAP(W1b, pB, 0);
    }
#undef GATHER
#undef VLOADW
#undef MMAP

    // ---- epilogue (the arithmetic of the dense fused ReadOut tail, conv_igemm.hip OUT_FUSED_HEAD)
    constexpr int SPR = HID / 8;                       // 16-B slots per proposal row
    constexpr int SW = SPR >= 16 ? 16 : SPR;           // swizzle period
    unsigned char *const stage = smem + hd * (MT * HID * 2);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (weight prefetch past the end: register writes only)
    __syncthreads();                                   // every wave is done reading the gather ring
#pragma unroll
    for (int f = 0; f < WM; ++f) {
        const int p = f * 32 + l31;
        const int fp = p & (SW - 1);
#pragma unroll
        for (int j = 0; j < WN; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int cb = wq * WN * 32 + j * 32 + 8 * q + 4 * lhi;  // hidden unit of this head
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = fmaxf(acc[j][f][q * 4 + e] + H.b1[cb + e], 0.f);
                u32x2 o;
                o.x = pack_bf16x2(v[0], v[1]);
                o.y = pack_bf16x2(v[2], v[3]);
                *(u32x2 *) (stage + p * (HID * 2) + (((cb >> 3) ^ fp) << 4) + lhi * 8) = o;
            }
    }
    __syncthreads();
    {   // second GEMM: wave -> (head hd, proposal fragment wq)
        const int p = wq * 32 + l31;
        const int fp = p & (SW - 1);
        const unsigned char *w2 = (const unsigned char *) H.w2 + l31 * (HID * 2) + lhi * 16;
        f32x16 acc2;
#pragma unroll
        for (int e = 0; e < 16; ++e) acc2[e] = 0.f;
#pragma unroll
        for (int ks = 0; ks < HID / 16; ++ks) {
            const bf16x8 wv = *(const bf16x8 *) (w2 + ks * 32);
            const bf16x8 xv = *(const bf16x8 *) (stage + p * (HID * 2) + (((ks * 2 + lhi) ^ fp) << 4));
            acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wv, xv, acc2, 0, 0, 0);
        }
        const int gpo = p0 + p;
        if (gpo < a.P) {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int c2 = (e & 3) + 8 * (e >> 2) + 4 * lhi;
                if (c2 >= H.cout) continue;
                float x = acc2[e] + (H.b2 ? H.b2[c2] : 0.f);
                if (H.act == ACT_RELU) x = fmaxf(x, 0.f);
                else if (H.act == ACT_SIGMOID) x = 1.f / (1.f + expf(-x));
                else if (H.act == ACT_TANH_SCALED) x = tanhf(x) * H.act_scale;
                H.out[(size_t) gpo * H.cout + c2] = x;
            }
        }
    }
}

template <int HID>
static int launch(const Args &a, hipStream_t stream) {
    static std::atomic<bool> attr_set[64];  // per device ordinal (the dynamic-LDS limit is a per-device function attribute)
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return (int) hipErrorInvalidDevice;
    const size_t lds = (size_t) 2 * MT * HID * 2 > (size_t) 4 * SLOT ? (size_t) 2 * MT * HID * 2 : (size_t) 4 * SLOT;
    auto kern = sparse_heads_kernel<HID>;
    if (!attr_set[dev].load(std::memory_order_acquire)) {
        hipError_t e = hipFuncSetAttribute((const void *) kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return (int) e;
        attr_set[dev].store(true, std::memory_order_release);
    }
    hipLaunchKernelGGL(kern, dim3((unsigned) ((a.P + MT - 1) / MT)), dim3(64 * NWAVES), lds, stream, a);
    return (int) hipGetLastError();
}

}  // namespace sparse
}  // namespace cpn

extern "C" int cpn_sparse_heads(const cpn_op_desc *op_a, const cpn_op_desc *op_b, const void *features,
                                int32_t channel_stride, int32_t N, int32_t h, int32_t w, const int32_t *indices,
                                int32_t P, const void *weights, const float *bias, float *out_a, float *out_b,
                                void *stream) {
    using namespace cpn;
    if (!op_a || !op_b || !features || !weights || !bias || !out_a || !out_b || (P > 0 && !indices))
        return fail(CPN_E_INVALID, "cpn_sparse_heads: null pointer");
    if (P < 0 || N <= 0 || h <= 0 || w <= 0) return fail(CPN_E_INVALID, "cpn_sparse_heads: bad sizes");
    if (P == 0) return 0;
    const cpn_op_desc *ops[2] = {op_a, op_b};
    for (const cpn_op_desc *o : ops) {
        if (o->op != CPN_OP_CONV && o->op != CPN_OP_CONV_DEFERRED) return fail(CPN_E_INVALID, "cpn_sparse_heads: not a conv op");
        if (o->fuse_cout <= 0 || o->fuse_cout > 32 || o->fuse_weight_offset < 0 || o->bundles != 1 || o->src1 >= 0 ||
            o->res >= 0 || o->up0 || o->stride != 1 || o->kh != o->kw || o->pad != o->kh / 2 || o->act != CPN_ACT_RELU ||
            o->bias_offset < 0)
            return fail(CPN_E_UNSUPPORTED, "cpn_sparse_heads: the heads must be fused ReadOut convs (k x k, stride 1, 'same' "
                                           "padding, single plain source, ReLU)");
    }
    if (op_a->cout_b != op_b->cout_b || op_a->cin_b != op_b->cin_b || op_a->kh != op_b->kh || op_a->src0 != op_b->src0)
        return fail(CPN_E_UNSUPPORTED, "cpn_sparse_heads: the two heads must share source, kernel size and hidden width");
    if (op_a->cout_b != 256 && op_a->cout_b != 128)
        return fail(CPN_E_UNSUPPORTED, "cpn_sparse_heads: hidden width must be 128 or 256");
    if (op_a->cin_b % 32 || op_a->cin_b > channel_stride || op_a->kh > 15)
        return fail(CPN_E_INVALID, "cpn_sparse_heads: bad channel counts / kernel size");
    if ((int64_t) N * h * w * channel_stride * 2 > ((int64_t) 1 << 31))
        return fail(CPN_E_UNSUPPORTED, "cpn_sparse_heads: feature tensor beyond 2^31 bytes");
    sparse::Args a{};
    a.feat = features; a.N = N; a.h = h; a.w = w; a.cs = channel_stride; a.cin = op_a->cin_b; a.K = op_a->kh;
    a.pad = op_a->pad; a.indices = indices; a.P = P;
    float *outs[2] = {out_a, out_b};
    for (int i = 0; i < 2; ++i) {
        const cpn_op_desc *o = ops[i];
        sparse::Head &H = a.head[i];
        H.w1 = (const unsigned char *) weights + o->weight_offset;
        H.b1 = bias + o->bias_offset;
        H.w2 = (const unsigned char *) weights + o->fuse_weight_offset;
        H.b2 = o->fuse_bias_offset >= 0 ? bias + o->fuse_bias_offset : nullptr;
        H.out = outs[i]; H.cout = o->fuse_cout; H.act = o->fuse_act; H.act_scale = o->fuse_act_scale;
    }
    const int rc = op_a->cout_b == 256 ? sparse::launch<256>(a, (hipStream_t) stream)
                                       : sparse::launch<128>(a, (hipStream_t) stream);
    return check_hip((hipError_t) rc, "cpn_sparse_heads");
}
