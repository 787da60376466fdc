// Probe (next-round groundwork, not product code): operand layout of v_mfma_scale_f32_32x32x64_f8f6f4 with fp8 e4m3
// A/B on gfx950.  Tests lane/byte -> (row|col, k) hypotheses against a CPU reference:
//   H0: lane l, byte j (0..31)  ->  row/col = l & 31,  k = 32*(l>>5) + j
//   H1: lane l, byte j          ->  row/col = l & 31,  k = 16*(l>>5) + (j & 15) + 32*(j >> 4)
// build: hipcc --offload-arch=gfx950 -O2 tools/probes/f8f6f4_layout.hip -o tools/probes/f8probe.bin ; run on the MI355X.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>

typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

__global__ void k(const uint8_t *a, const uint8_t *b, float *d) {
    const int l = threadIdx.x;
    i32x8 va = *(const i32x8 *) (a + l * 32), vb = *(const i32x8 *) (b + l * 32);
    f32x16 acc;
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    // cbsz = 0 / blgp = 0: fp8 e4m3 for A and B; scales: E8M0 exponent 127 = 1.0
    acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(va, vb, acc, 0, 0, 0, 127, 0, 127);
    for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5), col = l & 31;
        d[row * 32 + col] = acc[r];
    }
}

static float e4m3(uint8_t v) {  // OCP e4m3fn
    const int s = v >> 7, e = (v >> 3) & 15, m = v & 7;
    float x = e == 0 ? std::ldexp((float) m, -9) : std::ldexp(1.f + m / 8.f, e - 7);
    return s ? -x : x;
}

int main() {
    std::vector<uint8_t> A(32 * 64), B(64 * 32);  // A[row][k], B[k][col]
    srand(1);
    for (auto &v : A) { v = rand() & 0xff; if ((v & 0x7f) == 0x7f) v = 0x38; }   // avoid NaN
    for (auto &v : B) { v = rand() & 0xff; if ((v & 0x7f) == 0x7f) v = 0x38; }
    std::vector<float> ref(32 * 32, 0.f);
    for (int i = 0; i < 32; ++i)
        for (int j = 0; j < 32; ++j) {
            double s = 0;
            for (int kk = 0; kk < 64; ++kk) s += (double) e4m3(A[i * 64 + kk]) * e4m3(B[kk * 32 + j]);
            ref[i * 32 + j] = (float) s;
        }
    uint8_t *da, *db;
    float *dd;
    hipMalloc(&da, 2048); hipMalloc(&db, 2048); hipMalloc(&dd, 4096);
    for (int hyp = 0; hyp < 2; ++hyp) {
        std::vector<uint8_t> la(2048), lb(2048);
        for (int l = 0; l < 64; ++l)
            for (int j = 0; j < 32; ++j) {
                const int kk = hyp == 0 ? 32 * (l >> 5) + j : 16 * (l >> 5) + (j & 15) + 32 * (j >> 4);
                la[l * 32 + j] = A[(l & 31) * 64 + kk];
                lb[l * 32 + j] = B[kk * 32 + (l & 31)];
            }
        hipMemcpy(da, la.data(), 2048, hipMemcpyHostToDevice);
        hipMemcpy(db, lb.data(), 2048, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, da, db, dd);
        std::vector<float> got(1024);
        hipMemcpy(got.data(), dd, 4096, hipMemcpyDeviceToHost);
        double err = 0, mx = 0;
        for (int i = 0; i < 1024; ++i) { err = fmax(err, fabs(got[i] - ref[i])); mx = fmax(mx, fabs(ref[i])); }
        printf("hypothesis H%d: max abs err %.4g (ref max %.4g) -> %s\n", hyp, err, mx, err <= 1e-3 * mx ? "MATCH" : "no");
    }
    return 0;
}
