// probe: semantics of buffer_load_dwordx4 ... offen lds (raw buffer, stride 0) on gfx950:
//   (a) in-range lanes copy 16 B each to LDS [base + lane*16]; (b) a lane whose voffset is 0x80000000 gets zeros;
//   (c) is the SGPR offset part of the range check?  (voffset in range, voffset + soffset beyond num_records)
// build: hipcc --offload-arch=gfx950 -O2 tools/probes/buffer_lds_probe.hip -o tools/probes/buffer_lds_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(const unsigned *p, unsigned nbytes, unsigned soff, unsigned *out) {
    __shared__ __attribute__((aligned(16))) unsigned smem[256 + 4];
    for (int i = threadIdx.x; i < 260; i += 64) smem[i] = 0xdeadbeefu;
    __syncthreads();
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void *) p, 0, nbytes, 0x00020000);
    const int lane = threadIdx.x & 63;
    unsigned voff = lane * 16u;
    if (lane == 7) voff = 0x80000000u;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void *) smem, 16, voff, soff, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < 256; i += 64) out[i] = smem[i];
}
int main() {
    const unsigned n = 4096;  // dwords
    std::vector<unsigned> h(n);
    for (unsigned i = 0; i < n; ++i) h[i] = 0x10000u + i;
    unsigned *d, *o;
    hipMalloc(&d, n * 4); hipMalloc(&o, 256 * 4);
    hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice);
    std::vector<unsigned> r(256);
    // case 1: num_records = 1024 B + 128 soffset window, soffset 128 -> lanes 0..63 read bytes 128 .. 1152
    for (int c = 0; c < 3; ++c) {
        const unsigned nbytes = c == 0 ? 4096u : (c == 1 ? 1024u : 1024u), soff = c == 2 ? 512u : 128u;
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, nbytes, soff, o);
        hipMemcpy(r.data(), o, 256 * 4, hipMemcpyDeviceToHost);
        printf("case %d: num_records %u soffset %u\n", c, nbytes, soff);
        for (int lane : {0, 1, 6, 7, 8, 31, 55, 56, 57, 63})
            printf("  lane %2d: %08x %08x %08x %08x (expect first dword %08x if in range)\n", lane, r[lane * 4], r[lane * 4 + 1],
                   r[lane * 4 + 2], r[lane * 4 + 3], 0x10000u + (lane * 16 + soff) / 4);
    }
    return 0;
}
