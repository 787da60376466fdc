// Does the power (= sustained rate on random operands) of a register-only bf16 MFMA loop depend on WHICH operand changes between
// consecutive instructions?  The conv kernel's wave tile: 2 weight fragments x 4 pixel fragments, for j (weights) for f (pixels)
// MFMA(A = w[j], B = p[f]).  V0 = that; V1 = same order, operand roles swapped; V2 = f outer, j inner; V3 = V2 with swapped roles.
// hipcc --offload-arch=gfx950 -O3 mfma_operand_order.hip -o mfma_operand_order.bin   (profiles/r05_kernel_experiments.txt #12)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef __attribute__((ext_vector_type(8))) __bf16 bf8_t;
typedef __attribute__((ext_vector_type(4))) int i4_t;
typedef __attribute__((ext_vector_type(16))) float f16_t;
#define MMA(A, B, C) C = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf8_t, A), __builtin_bit_cast(bf8_t, B), C, 0, 0, 0)

template <int V>
__global__ __launch_bounds__(256, 2) void loop(const int* __restrict__ src, float* __restrict__ out, int iters) {
    i4_t w[2][2], p[4][2];  // [fragment][k-half]
    const int lane = threadIdx.x;
    for (int j = 0; j < 2; ++j) for (int h = 0; h < 2; ++h) for (int e = 0; e < 4; ++e) w[j][h][e] = src[(((j * 2 + h) * 4 + e) * 256 + lane) & 16383];
    for (int f = 0; f < 4; ++f) for (int h = 0; h < 2; ++h) for (int e = 0; e < 4; ++e) p[f][h][e] = src[((16 + (f * 2 + h) * 4 + e) * 256 + lane) & 16383];
    f16_t acc[2][4];
    for (int j = 0; j < 2; ++j) for (int f = 0; f < 4; ++f) for (int e = 0; e < 16; ++e) acc[j][f][e] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            if constexpr (V < 2) {
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int f = 0; f < 4; ++f) {
                        if constexpr (V == 0) MMA(w[j][h], p[f][h], acc[j][f]); else MMA(p[f][h], w[j][h], acc[j][f]);
                    }
            } else {
#pragma unroll
                for (int f = 0; f < 4; ++f)
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        if constexpr (V == 2) MMA(w[j][h], p[f][h], acc[j][f]); else MMA(p[f][h], w[j][h], acc[j][f]);
                    }
            }
        }
        asm volatile("" ::: "memory");
    }
    float s = 0.f;
    for (int j = 0; j < 2; ++j) for (int f = 0; f < 4; ++f) for (int e = 0; e < 16; ++e) s += acc[j][f][e];
    if (s == 12345.678f) out[0] = s;
}

template <int V>
static double run(const int* d_src, float* d_out, int blocks, int iters, int reps) {
    hipEvent_t e0, e1;
    (void) hipEventCreate(&e0); (void) hipEventCreate(&e1);
    loop<V><<<blocks, 256>>>(d_src, d_out, iters);
    (void) hipDeviceSynchronize();
    (void) hipEventRecord(e0);
    for (int r = 0; r < reps; ++r) loop<V><<<blocks, 256>>>(d_src, d_out, iters);
    (void) hipEventRecord(e1);
    (void) hipEventSynchronize(e1);
    float ms = 0.f;
    (void) hipEventElapsedTime(&ms, e0, e1);
    return 16. * 2 * 32 * 32 * 16 * iters * (blocks * 4.) * reps / (ms * 1e-3) * 1e-12;
}

int main() {
    std::vector<int> h(16384);
    int* d_src; float* d_out;
    (void) hipMalloc(&d_src, h.size() * 4); (void) hipMalloc(&d_out, 64);
    srand(1);
    for (auto& v : h) v = (int) ((((unsigned) rand() << 16) ^ (unsigned) rand()) & 0xBFBFBFBFu);
    (void) hipMemcpy(d_src, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    for (int rep = 0; rep < 2; ++rep) {
        const double v0 = run<0>(d_src, d_out, 512, 10000, 20), v1 = run<1>(d_src, d_out, 512, 10000, 20);
        const double v2 = run<2>(d_src, d_out, 512, 10000, 20), v3 = run<3>(d_src, d_out, 512, 10000, 20);
        printf("random operands, TF/s: V0 (A = w outer, B = p inner: the conv kernel) %.1f | V1 (roles swapped) %.1f | V2 (p outer, w inner) %.1f | V3 (V2 swapped) %.1f\n", v0, v1, v2, v3);
    }
    return 0;
}
