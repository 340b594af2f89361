// Lane-level helpers shared by the plane product (plane.hip) and the grid product (grid.hip): a lane owns two adjacent rows of a
// grid line as one 16-byte pair; the +-1 neighbours of its rows are one DPP wave shift away, and positions of the stencil
// without an entry are masked so that whatever x holds there is multiplied away.
#pragma once
#include <hip/hip_runtime.h>

namespace vexhip {
namespace {

typedef double d2 __attribute__((ext_vector_type(2)));
typedef unsigned u4 __attribute__((ext_vector_type(4)));
typedef unsigned u2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ double shift_from_lower_lane(double v, double edge) {       // lane i <- lane i - 1, lane 0 <- edge
    const int lo = __builtin_amdgcn_update_dpp(__double2loint(edge), __double2loint(v), 0x138, 0xf, 0xf, false);   // wave_shr:1
    const int hi = __builtin_amdgcn_update_dpp(__double2hiint(edge), __double2hiint(v), 0x138, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double shift_from_upper_lane(double v, double edge) {       // lane i <- lane i + 1, lane 63 <- edge
    const int lo = __builtin_amdgcn_update_dpp(__double2loint(edge), __double2loint(v), 0x130, 0xf, 0xf, false);   // wave_shl:1
    const int hi = __builtin_amdgcn_update_dpp(__double2hiint(edge), __double2hiint(v), 0x130, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
// v for the lanes of `lanes`, elsewhere a number whose exponent field is 0: (+0.0) * that == +0.0 whatever x holds there
// (the matrix value of a position without an entry is +0.0; x may hold Inf / NaN where CSR would never look)
__device__ __forceinline__ double keep_lanes(double v, unsigned long long lanes) {
    unsigned rhi;
    asm("v_cndmask_b32_e64 %0, 0, %1, %2" : "=v"(rhi) : "v"((unsigned)__double2hiint(v)), "s"(lanes));
    return __hiloint2double((int)rhi, __double2loint(v));
}
__device__ __forceinline__ double keep_bit(double v, unsigned bits, int pos) {
    const int m = (int)(bits << (31 - pos)) >> 31;                // -1 where the bit is set
    return __hiloint2double(__double2hiint(v) & m, __double2loint(v));
}

// fp32 (plane32.hip): a lane owns FOUR adjacent rows as one 16-byte quad
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));    // a quad at any element (16-byte requests at 4-byte addresses)
__device__ __forceinline__ float shift_from_lower_lane(float v, float edge) {          // lane i <- lane i - 1, lane 0 <- edge
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(edge), __float_as_int(v), 0x138, 0xf, 0xf, false));
}
__device__ __forceinline__ float shift_from_upper_lane(float v, float edge) {          // lane i <- lane i + 1, lane 63 <- edge
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(edge), __float_as_int(v), 0x130, 0xf, 0xf, false));
}
__device__ __forceinline__ float keep_lanes(float v, unsigned long long lanes) {       // +0.0 outside `lanes`
    unsigned r;
    asm("v_cndmask_b32_e64 %0, 0, %1, %2" : "=v"(r) : "v"(__float_as_uint(v)), "s"(lanes));
    return __uint_as_float(r);
}
__device__ __forceinline__ float keep_bit(float v, unsigned bits, int pos) {
    const int m = (int)(bits << (31 - pos)) >> 31;                // -1 where the bit is set
    return __int_as_float(__float_as_int(v) & m);
}

} // namespace
} // namespace vexhip
