// Split-bf16 ("bf16x3") operand helpers shared by the contraction engine (gemm_f32.hip) and the fused attention kernels (attn.hip):
// x = hi + lo with hi = bf16(x), lo = bf16(x - hi); a.b ~= ah.bh + ah.bl + al.bh on v_mfma_f32_32x32x16_bf16, fp32 accumulate.
#pragma once
#include "common.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short bf16raw;
union Frag8 { uint4 u; bf16x8 v; };

// hi/lo split of 4 consecutive floats into 4+4 bf16, converting PAIRS (one v_cvt_pk_bf16_f32 per two values; the scalar casts
// compile to one cvt per value): 3 VALU per element instead of 4 in the staging path, which is what the bf16x3 loop is bound by.
typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void split4(const float4& v, uint2& hi, uint2& lo) {
  union { bf16x2_t b; unsigned u; } h0, h1, l0, l1;
  const f32x2_t a = {v.x, v.y}, c = {v.z, v.w};
  h0.b = __builtin_convertvector(a, bf16x2_t);
  h1.b = __builtin_convertvector(c, bf16x2_t);
  const f32x2_t ra = {v.x - __uint_as_float(h0.u << 16), v.y - __uint_as_float(h0.u & 0xffff0000u)};
  const f32x2_t rc = {v.z - __uint_as_float(h1.u << 16), v.w - __uint_as_float(h1.u & 0xffff0000u)};
  l0.b = __builtin_convertvector(ra, bf16x2_t);
  l1.b = __builtin_convertvector(rc, bf16x2_t);
  hi = make_uint2(h0.u, h1.u);
  lo = make_uint2(l0.u, l1.u);
}
// two floats -> one dword of (hi, hi) and one of (lo, lo)
__device__ __forceinline__ void split2(float x, float y, unsigned& hi, unsigned& lo) {
  union { bf16x2_t b; unsigned u; } h, l;
  const f32x2_t a = {x, y};
  h.b = __builtin_convertvector(a, bf16x2_t);
  const f32x2_t r = {x - __uint_as_float(h.u << 16), y - __uint_as_float(h.u & 0xffff0000u)};
  l.b = __builtin_convertvector(r, bf16x2_t);
  hi = h.u; lo = l.u;
}

// gfx950 LDS transpose read (ds_read_b64_tr_b16): see gemm_f32.hip::read_frag_presplit_mc for the lane <-> element map
typedef __attribute__((__vector_size__(4 * sizeof(__bf16)))) __bf16 bf16x4_t;
#define LDS_AS __attribute__((address_space(3)))
