// probe (round 6): does a LARGER wave tile -- fewer LDS fragment reads per MFMA -- buy throughput on random operands, where the
// chip clocks to its power budget?  bf16 32x32x16 MFMA loops whose operand fragments are re-read from LDS every K step, software
// pipelined over two fragment sets (the conv kernels' scheme), no DMA, no barrier:
//   T0  2 x 4 accumulators, operands in registers (no reads)                      : the register-only ceiling
//   T1  2 x 4 accumulators (the conv kernels' 128 px x 64 cout wave tile), 6 reads per 8 MFMAs, two waves per SIMD
//   T2  4 x 4 accumulators (128 px x 128 cout, 256 accumulator registers), 8 reads per 16 MFMAs, ONE wave per SIMD
// build: hipcc --offload-arch=gfx950 -O3 tools/probes/mfma_lds_ratio.hip -o tools/probes/mfma_lds_ratio.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef __attribute__((ext_vector_type(8))) __bf16 bf8_t;
typedef __attribute__((ext_vector_type(16))) float f16_t;

__device__ __forceinline__ void ldsr(bf8_t &d, unsigned addr) { asm volatile("ds_read_b128 %0, %1" : "=v"(d) : "v"(addr)); }

template <int WN, int WM, int READ, int MINB>
__global__ __launch_bounds__(256, MINB) void loop(const int *__restrict__ src, float *__restrict__ out, int iters) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    for (int i = threadIdx.x; i < 65536 / 4; i += 256) ((int *) smem)[i] = src[i & 16383];
    __syncthreads();
    const unsigned lds0 = (unsigned) (size_t) (__attribute__((address_space(3))) unsigned char *) smem;
    const unsigned lane_off = lds0 + (threadIdx.x & 63) * 16 + (threadIdx.x >> 6) * 2048;
    bf8_t wA[WN], pA[WM], wB[WN], pB[WM];
    f16_t acc[WN][WM];
    for (int j = 0; j < WN; ++j) for (int f = 0; f < WM; ++f) for (int e = 0; e < 16; ++e) acc[j][f][e] = 0.f;
#define LOADSET(W, P, IT)                                                                        \
    {                                                                                            \
        const unsigned b_ = lane_off + (((IT) & 3) << 13);                                       \
        _Pragma("unroll") for (int j = 0; j < WN; ++j) ldsr(W[j], b_ + j * 1024);                \
        _Pragma("unroll") for (int f = 0; f < WM; ++f) ldsr(P[f], b_ + 4096 + f * 1024 - ((IT) & 1) * 512); \
    }
#define WAITSET(W, P, N) { if constexpr (WN == 2) asm volatile("s_waitcnt lgkmcnt(%6)" : "+v"(W[0]), "+v"(W[1]), "+v"(P[0]), "+v"(P[1]), "+v"(P[2]), "+v"(P[3]) : "n"(N)); \
                           else asm volatile("s_waitcnt lgkmcnt(%8)" : "+v"(W[0]), "+v"(W[1]), "+v"(W[2]), "+v"(W[3]), "+v"(P[0]), "+v"(P[1]), "+v"(P[2]), "+v"(P[3]) : "n"(N)); }
#define MMASET(W, P)                                                                             \
    _Pragma("unroll") for (int j = 0; j < WN; ++j)                                               \
        _Pragma("unroll") for (int f = 0; f < WM; ++f) acc[j][f] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(W[j], P[f], acc[j][f], 0, 0, 0);
    LOADSET(wA, pA, 0);
    if constexpr (!READ) { LOADSET(wB, pB, 1); WAITSET(wA, pA, 0); WAITSET(wB, pB, 0); }
    for (int it = 0; it < iters; it += 2) {
        if constexpr (READ) {
            LOADSET(wB, pB, it + 1);
            WAITSET(wA, pA, WN + WM);
            MMASET(wA, pA);
            LOADSET(wA, pA, it + 2);
            WAITSET(wB, pB, WN + WM);
            MMASET(wB, pB);
        } else {
            MMASET(wA, pA);
            MMASET(wB, pB);
            asm volatile("" ::: "memory");
        }
    }
    if constexpr (READ) WAITSET(wA, pA, 0);
    float s = 0.f;
    for (int j = 0; j < WN; ++j) for (int f = 0; f < WM; ++f) for (int e = 0; e < 16; ++e) s += acc[j][f][e];
    if (s == 12345.678f) out[0] = s;
}

// e4m3 (scaled 32x32x64): an operand is 32 B per lane = two 16-byte reads; 2 x 4 accumulators, whole operand sets alternate
typedef __attribute__((ext_vector_type(4))) int i4_t;
typedef __attribute__((ext_vector_type(8))) int i8_t;
__device__ __forceinline__ void ldsr4(i4_t &d, unsigned addr) { asm volatile("ds_read_b128 %0, %1" : "=v"(d) : "v"(addr)); }
template <int READ>
__global__ __launch_bounds__(256, 2) void loop8(const int *__restrict__ src, float *__restrict__ out, int iters) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    for (int i = threadIdx.x; i < 65536 / 4; i += 256) ((int *) smem)[i] = src[i & 16383];
    __syncthreads();
    const unsigned lds0 = (unsigned) (size_t) (__attribute__((address_space(3))) unsigned char *) smem;
    const unsigned lane_off = lds0 + (threadIdx.x & 63) * 16 + (threadIdx.x >> 6) * 2048;
    i4_t x0[6], y0[6], x1[6], y1[6];  // [0..1] weights, [2..5] pixels; X | Y = the two 16-byte parts
    f16_t acc[2][4];
    for (int j = 0; j < 2; ++j) for (int f = 0; f < 4; ++f) for (int e = 0; e < 16; ++e) acc[j][f][e] = 0.f;
#define LOAD8(X, Y, IT) { const unsigned b_ = lane_off + (((IT) & 3) << 13); \
        _Pragma("unroll") for (int q = 0; q < 6; ++q) { ldsr4(X[q], b_ + q * 1024); } \
        _Pragma("unroll") for (int q = 0; q < 6; ++q) { ldsr4(Y[q], (b_ + q * 1024) ^ 16u); } }
#define WAIT8(X, Y, N) asm volatile("s_waitcnt lgkmcnt(%12)" : "+v"(X[0]), "+v"(X[1]), "+v"(X[2]), "+v"(X[3]), "+v"(X[4]), "+v"(X[5]), \
        "+v"(Y[0]), "+v"(Y[1]), "+v"(Y[2]), "+v"(Y[3]), "+v"(Y[4]), "+v"(Y[5]) : "n"(N))
#define MMA8(X, Y) _Pragma("unroll") for (int j = 0; j < 2; ++j) _Pragma("unroll") for (int f = 0; f < 4; ++f) \
        acc[j][f] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(__builtin_shufflevector(X[j], Y[j], 0, 1, 2, 3, 4, 5, 6, 7), \
            __builtin_shufflevector(X[2 + f], Y[2 + f], 0, 1, 2, 3, 4, 5, 6, 7), acc[j][f], 0, 0, 0, 127, 0, 127);
    LOAD8(x0, y0, 0);
    if constexpr (!READ) { LOAD8(x1, y1, 1); WAIT8(x0, y0, 0); WAIT8(x1, y1, 0); }
    for (int it = 0; it < iters; it += 2) {
        if constexpr (READ) {
            LOAD8(x1, y1, it + 1); WAIT8(x0, y0, 12); MMA8(x0, y0);
            LOAD8(x0, y0, it + 2); WAIT8(x1, y1, 12); MMA8(x1, y1);
        } else { MMA8(x0, y0); MMA8(x1, y1); asm volatile("" ::: "memory"); }
    }
    if constexpr (READ) WAIT8(x0, y0, 0);
    float s = 0.f;
    for (int j = 0; j < 2; ++j) for (int f = 0; f < 4; ++f) for (int e = 0; e < 16; ++e) s += acc[j][f][e];
    if (s == 12345.678f) out[0] = s;
}
template <int READ>
static double run8(const int *d_src, float *d_out, int blocks, int iters, int reps) {
    auto k = loop8<READ>;
    hipFuncSetAttribute((const void *) k, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k<<<blocks, 256, 65536>>>(d_src, d_out, iters);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < reps; ++r) k<<<blocks, 256, 65536>>>(d_src, d_out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    return 8. * 2. * 32 * 32 * 64 * iters * (blocks * 4.) * reps / (ms * 1e-3) * 1e-12;
}

template <int WN, int WM, int READ, int MINB>
static double run(const int *d_src, float *d_out, int blocks, int iters, int reps) {
    auto k = loop<WN, WM, READ, MINB>;
    hipFuncSetAttribute((const void *) k, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k<<<blocks, 256, 65536>>>(d_src, d_out, iters);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < reps; ++r) k<<<blocks, 256, 65536>>>(d_src, d_out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    return (double) WN * WM * 2. * 32 * 32 * 16 * iters * (blocks * 4.) * reps / (ms * 1e-3) * 1e-12;
}

int main(int argc, char **argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 20000, reps = 10;
    std::vector<int> h(16384);
    int *d_src; float *d_out;
    hipMalloc(&d_src, h.size() * 4); hipMalloc(&d_out, 64);
    for (int mode = 0; mode < 2; ++mode) {
        srand(1);
        for (auto &v : h) { unsigned r = ((unsigned) rand() << 16) ^ (unsigned) rand(); v = mode == 0 ? 0 : (int) (r & 0xBFBFBFBFu); }
        hipMemcpy(d_src, h.data(), h.size() * 4, hipMemcpyHostToDevice);
        const double t0 = run<2, 4, 0, 2>(d_src, d_out, 512, iters, reps);
        const double t1 = run<2, 4, 1, 2>(d_src, d_out, 512, iters, reps);
        const double t2 = run<4, 4, 1, 1>(d_src, d_out, 256, iters / 2, reps);
        const double t2r = run<4, 4, 0, 1>(d_src, d_out, 256, iters / 2, reps);
        const double f0 = run8<0>(d_src, d_out, 512, iters / 2, reps), f1 = run8<1>(d_src, d_out, 512, iters / 2, reps);
        printf("%-8s e4m3 32x32x64, 2x4 accumulators, 2 waves/SIMD: registers %7.1f TF/s (%.3f of 5 PF) | 12 LDS reads / 8 MFMAs %7.1f (%.3f)\n",
               mode == 0 ? "zeros" : "random", f0, f0 / 5000., f1, f1 / 5000.);
        printf("%-8s T0 2x4 registers, 2 waves/SIMD %7.1f TF/s | T1 2x4 + 6 LDS reads / 8 MFMAs, 2 waves/SIMD %7.1f | T2 4x4 + 8 reads / 16 MFMAs, "
               "1 wave/SIMD %7.1f | 4x4 registers, 1 wave/SIMD %7.1f\n", mode == 0 ? "zeros" : "random", t0, t1, t2, t2r);
    }
    return 0;
}
