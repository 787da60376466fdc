// probe (round 6): what rocprofv3's FETCH_SIZE reports for the access patterns of the conv kernels, against a KNOWN byte count.
// MI355X_MICROARCH.md: "FETCH_SIZE reports exactly 1/2 of the bytes of a wide coalesced streaming read (16 B / lane) ... other access
// widths are uncalibrated: calibrate on a known byte count in your own access pattern".  The conv graph mixes two patterns:
//   weights      contiguous 1 KiB per LDS-DMA instruction (64 lanes x 16 B)                        -> kernel dma_contiguous
//   halo tiles   64-B pixel records at a stride of C x 2 B (C = 256 channels: 512 B), 4 lanes x 16 B each -> kernel dma_records
// plus, for reference, the guide's own case: global_load_dwordx4 streaming                          -> kernel vec_stream
// Every kernel reads each byte it touches exactly once from a 2 GiB buffer (> the 256 MiB Infinity Cache) and touches a known
// number of bytes; run under `rocprofv3 --pmc FETCH_SIZE` and compare (FETCH_SIZE is in KiB).
// build: hipcc --offload-arch=gfx950 -O2 tools/probes/fetch_calibration.hip -o tools/probes/fetch_calibration.bin
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

__global__ __launch_bounds__(256) void vec_stream(const u32x4 *p, size_t n16, unsigned *sink) {
    u32x4 acc = {0, 0, 0, 0};
    for (size_t i = blockIdx.x * (size_t) blockDim.x + threadIdx.x; i < n16; i += (size_t) gridDim.x * blockDim.x) {
        const u32x4 v = p[i];
        acc.x ^= v.x; acc.y ^= v.y; acc.z ^= v.z; acc.w ^= v.w;
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345u) sink[0] = 1;
}

// one wave-instruction = 1 KiB: lane -> 16 B at byte offset (instr * 1024 + lane * 16) * STRIDE_NUM / STRIDE_DEN ... two layouts:
// RECORDS = 0: contiguous (lane * 16);  RECORDS = 1: record r = lane >> 2 lives at r * 512 B, part (lane & 3) * 16 B within it
template <int RECORDS>
__global__ __launch_bounds__(256) void dma_kernel(const unsigned char *p, unsigned bytes_per_block, int iters, unsigned *sink) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[4 * 4096];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned char *base = p + (size_t) blockIdx.x * bytes_per_block;
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void *) base, 0, (int) bytes_per_block, 0x00020000);
    const unsigned voff = RECORDS ? (unsigned) ((lane >> 2) * 512 + (lane & 3) * 16) : (unsigned) (lane * 16);
    const unsigned span = RECORDS ? 16u * 512u : 1024u;  // bytes of address space one instruction walks over
    for (int i = 0; i < iters; ++i) {
        const unsigned soff = (unsigned) (i * 4 + wave) * span;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void *) (smem + wave * 4096 + (i & 3) * 1024), 16,
                                                 voff, soff, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (((unsigned *) smem)[threadIdx.x] == 0x12345u) sink[0] = 1;
}

int main() {
    const size_t total = (size_t) 2 << 30;  // 2 GiB
    unsigned char *d;
    unsigned *sink;
    hipMalloc(&d, total);
    hipMalloc(&sink, 64);
    hipMemset(d, 1, total);
    hipDeviceSynchronize();
    // (a) streaming vector loads over the whole buffer
    hipLaunchKernelGGL(vec_stream, dim3(256 * 16), dim3(256), 0, 0, (const u32x4 *) d, total / 16, sink);
    // (b) LDS-DMA, contiguous: every block walks its own 1 MiB
    const unsigned bpb = 1u << 20;
    const int blocks = (int) (total / bpb);
    hipLaunchKernelGGL(dma_kernel<0>, dim3(blocks), dim3(256), 0, 0, d, bpb, (int) (bpb / 4096), sink);
    // (c) LDS-DMA, 64-B records at 512-B stride: a block's 1 MiB holds 2048 records = 128 KiB touched
    hipLaunchKernelGGL(dma_kernel<1>, dim3(blocks), dim3(256), 0, 0, d, bpb, (int) (bpb / (4 * 16 * 512)), sink);
    hipDeviceSynchronize();
    printf("bytes touched: vec_stream %zu  dma_contiguous %zu  dma_records %zu (KiB: %zu %zu %zu)\n", total, total, total / 8,
           total >> 10, total >> 10, (total / 8) >> 10);
    return 0;
}
