// Device helpers shared by the one-launch legs (fused.hip, fused2.hip).
#pragma once
#include "common.h"

namespace simrank {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#ifndef SIMRANK_HOST_ONLY
__device__ __forceinline__ void split3f(float x0, float x1, uint32_t& hi, uint32_t& mid, uint32_t& lo) {
    const uint32_t u0 = __float_as_uint(x0), u1 = __float_as_uint(x1);
    hi = __builtin_amdgcn_perm(u1, u0, 0x07060302u);
    const float r0 = x0 - __uint_as_float(u0 & 0xFFFF0000u);
    const float r1 = x1 - __uint_as_float(u1 & 0xFFFF0000u);
    const uint32_t v0 = __float_as_uint(r0), v1 = __float_as_uint(r1);
    mid = __builtin_amdgcn_perm(v1, v0, 0x07060302u);
    const float q0 = r0 - __uint_as_float(v0 & 0xFFFF0000u);
    const float q1 = r1 - __uint_as_float(v1 & 0xFFFF0000u);
    lo = __builtin_amdgcn_perm(__float_as_uint(q1), __float_as_uint(q0), 0x07060302u);
}

__device__ __forceinline__ bf16x8 frag(uint32_t a, uint32_t b, uint32_t c, uint32_t d) {
    const uint4 v = make_uint4(a, b, c, d);
    return __builtin_bit_cast(bf16x8, v);
}

__device__ __forceinline__ float4 ld_seg(__amdgpu_buffer_rsrc_t srd, int id, uint32_t qoff) {
    typedef unsigned v4u __attribute__((ext_vector_type(4)));
    const v4u v = __builtin_amdgcn_raw_buffer_load_b128(srd, int(__umul24(uint32_t(id), 128u) + qoff), 0, 0);
    return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}

// the same with the row pitch of the operand in bytes (panel-blocked: 128; row-major shards: 4 x ld, below 2^24)
__device__ __forceinline__ float4 ld_seg_p(__amdgpu_buffer_rsrc_t srd, int id, uint32_t pitch, uint32_t qoff) {
    typedef unsigned v4u __attribute__((ext_vector_type(4)));
    const v4u v = __builtin_amdgcn_raw_buffer_load_b128(srd, int(__umul24(uint32_t(id), pitch) + qoff), 0, 0);
    return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}

// the compiler may not move LDS accesses of this wave across this point (no instruction is emitted)
__device__ __forceinline__ void wave_lds_order() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

#endif  // SIMRANK_HOST_ONLY

}  // namespace simrank
