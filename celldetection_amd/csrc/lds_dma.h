// Device helpers shared by the MFMA conv kernels of libcpn_hip.so (conv_igemm.hip, conv_pair.hip): vector types, bf16
// packing, LDS-DMA through raw buffer descriptors, inline-asm LDS fragment reads.  gfx950 only.
#pragma once
#include <hip/hip_runtime.h>

namespace cpn {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;
typedef __attribute__((ext_vector_type(4))) int i32x4;
typedef __attribute__((ext_vector_type(8))) int i32x8;

// two fp32 -> packed bf16 pair (lo, hi), round to nearest even: one v_cvt_pk_bf16_f32 on gfx950 (the software
// sequence cost ~7 VALU per element and made the epilogue of the memory-bound layers VALU-bound)
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
typedef __attribute__((ext_vector_type(2))) float f32x2_t;
__device__ __forceinline__ unsigned int pack_bf16x2(float lo, float hi) {
    const f32x2_t v = {lo, hi};
    return __builtin_bit_cast(unsigned int, __builtin_convertvector(v, bf16x2_t));
}
__device__ __forceinline__ float bf16_bits_to_f32(unsigned int b) { return __uint_as_float(b << 16); }

// LDS-DMA through a raw buffer descriptor (buffer_load_dwordx4 ... offen lds): 64 lanes x 16 B -> LDS
// [lds_wave_base + lane*16] (wave-uniform base).  Address = descriptor base + per-lane 32-bit byte offset + scalar byte
// offset: no 64-bit address arithmetic per instruction, and a lane whose offset lies beyond the descriptor's size
// receives ZEROS (tools/probes/buffer_lds_probe.hip: voffset + soffset + 16 > num_records -> 0) -- that is the conv's
// zero padding; OOB_LANE is the offset used for such lanes (tensors are limited to 2^31 bytes, see cpn_abi.hip)
constexpr unsigned OOB_LANE = 0x80000000u;
typedef __amdgpu_buffer_rsrc_t rsrc_t;
__device__ __forceinline__ rsrc_t make_rsrc(const void *base, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc((void *) base, 0, (int) bytes, 0x00020000);
}
// AUX = cache policy bits of the instruction (gfx94x / gfx950: 1 = sc0, 2 = nt, 16 = sc1)
template <int AUX = 0>
__device__ __forceinline__ void bdma16(rsrc_t rsrc, unsigned lane_off, unsigned scalar_off, unsigned char *lds_wave_base) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void *) lds_wave_base, 16,
                                             (int) lane_off, (int) scalar_off, 0, AUX);
}

// ---- hand-counted LDS fragment reads ----------------------------------------------------------------------------
// hipcc (ROCm 7.2) emits `s_waitcnt lgkmcnt(0)` in front of every MFMA group of a kernel that also issues LDS-DMA (it
// stops counting DS returns once LDS-DMA is in the function), which serialises "prefetch next fragments -> MFMA current
// fragments".  The fragment reads are therefore inline asm (invisible to the compiler's counters) and every MFMA group is
// preceded by a counted wait that names the fragment registers as "+v" so that no use can be scheduled above it.
template <int IMM, typename T>
__device__ __forceinline__ void ds_read16(T &d, unsigned addr) {
    static_assert(sizeof(T) == 16, "one 16-byte fragment");
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "n"(IMM));
}

}  // namespace cpn
