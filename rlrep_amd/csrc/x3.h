// bf16x3 helpers shared by the LDS-tiled GEMM engine (gemm_lds.hip) and the noise critic (noisecritic.hip): the exact
// three-way bf16 split of an fp32 value.  x = x1 + x2 + x3 with x1 = bf16(x), x2 = bf16(x - x1), x3 = bf16(x - x1 - x2)
// (3 x 8 significand bits; both residuals are exact in fp32), two values per call, packed as the MFMA operands want them.
#pragma once
#include "common.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2v __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2v __attribute__((ext_vector_type(2)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned x3_pk(float a, float b) {
    const f32x2v v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2v));
}
__device__ __forceinline__ void x3_split2(float x0, float x1, unsigned& hi, unsigned& mid, unsigned& lo) {
    hi = x3_pk(x0, x1);
    const float r0 = x0 - __builtin_bit_cast(float, hi << 16), r1 = x1 - __builtin_bit_cast(float, hi & 0xffff0000u);
    mid = x3_pk(r0, r1);
    const float s0 = r0 - __builtin_bit_cast(float, mid << 16), s1 = r1 - __builtin_bit_cast(float, mid & 0xffff0000u);
    lo = x3_pk(s0, s1);
}


// ---- bf16x3 weight shadows (ShadowEnt kind 1): [F / 32 steps][3 images][rows][4 groups of 8 k][8 bf16] --------------------------------
__device__ __forceinline__ size_t x3_shadow_off(int rows, int img, int r, int c) {        // byte offset of element (r, c) in image img
    return ((((size_t)(c >> 5) * 3 + img) * rows + r) * 4 + ((c >> 3) & 3)) * 16 + (size_t)(c & 7) * 2;
}
// four consecutive k (c % 4 == 0) of row r
__device__ __forceinline__ void x3_shadow_store(unsigned char* dst, int rows, int r, int c, const f32x4& v) {
    unsigned h0, m0, l0, h1, m1, l1;
    x3_split2(v[0], v[1], h0, m0, l0);
    x3_split2(v[2], v[3], h1, m1, l1);
    const u32x2 h = {h0, h1}, m = {m0, m1}, l = {l0, l1};
    *reinterpret_cast<u32x2*>(dst + x3_shadow_off(rows, 0, r, c)) = h;
    *reinterpret_cast<u32x2*>(dst + x3_shadow_off(rows, 1, r, c)) = m;
    *reinterpret_cast<u32x2*>(dst + x3_shadow_off(rows, 2, r, c)) = l;
}
__device__ __forceinline__ void x3_shadow_store1(unsigned char* dst, int rows, int r, int c, float v) {
    unsigned h, m, l;
    x3_split2(v, 0.f, h, m, l);
    *reinterpret_cast<unsigned short*>(dst + x3_shadow_off(rows, 0, r, c)) = (unsigned short)(h & 0xffffu);
    *reinterpret_cast<unsigned short*>(dst + x3_shadow_off(rows, 1, r, c)) = (unsigned short)(m & 0xffffu);
    *reinterpret_cast<unsigned short*>(dst + x3_shadow_off(rows, 2, r, c)) = (unsigned short)(l & 0xffffu);
}
