// Shared device helpers for the AdvMIL gfx950 kernels (wave64, CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define ADVMIL_OK 0
#define ADVMIL_EINVAL -1     // bad shape / alignment / null pointer
#define ADVMIL_EWORKSPACE -2 // workspace too small

#define ADVMIL_LAUNCH_CHECK()                      \
  do {                                             \
    hipError_t _e = hipGetLastError();             \
    if (_e != hipSuccess) return (int)_e;          \
  } while (0)

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// ---- counter RNG: key = splitmix64 mix of (seed, stream), element = rng_hash32(key, idx); restated on the host in advmil_amd/synth.py ----
__device__ __forceinline__ uint64_t splitmix64(uint64_t x) {
  uint64_t z = x + 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
__device__ __forceinline__ uint64_t rng_key(uint64_t seed, uint64_t stream) {
  return splitmix64(seed ^ splitmix64(stream));
}
// Per-element hash of (stream key, element index) -> 32 bits. The key of a call site is still a full splitmix64 mix of (seed,
// stream) -- once per kernel --, but the per-ELEMENT function is 32-bit: a splitmix64 per element is two 64-bit multiplies = eight
// quarter-rate 32-bit multiplies, which made gate_score (two draws per gate element, 10^8 per launch) and gate_bwd instruction-bound
// instead of HBM-bound. Two rounds of a multiply-xorshift mixer (constants of the "lowbias32" family), the key's low word added
// before the first round and its high word (and the index's high word, for tensors beyond 2^32 elements) folded in between the
// rounds, so two streams are not index-shifted copies of each other. Restated on the host in advmil_amd/synth.py::kernel_hash32.
__device__ __forceinline__ uint32_t rng_hash32(uint64_t key, uint64_t idx) {
  uint32_t x = (uint32_t)idx + (uint32_t)key;
  x ^= x >> 16; x *= 0x21f0aaadu;
  x ^= (uint32_t)(key >> 32) ^ ((uint32_t)(idx >> 32) * 0x85ebca77u);
  x ^= x >> 15; x *= 0x735a2d97u;
  x ^= x >> 15;
  return x;
}
// U[0,1) with 24 random bits
__device__ __forceinline__ float rng_uniform(uint64_t key, uint64_t idx) {
  return (float)(rng_hash32(key, idx) >> 8) * 5.9604644775390625e-8f;
}
// multiplicative dropout factor: 0 or 1/(1-p)
__device__ __forceinline__ float rng_keep(uint64_t key, uint64_t idx, float p, float inv_keep) {
  return rng_uniform(key, idx) >= p ? inv_keep : 0.0f;
}

// Four consecutive elements idx .. idx + 3 with idx % 4 == 0 (a lane's float4): the index's high word -- and with it the term folded in
// between the two rounds -- is the same for all four (the low word cannot wrap inside an aligned group), so it is formed once instead of
// four times: one quarter-rate multiply in nine saved per element. Bit-identical to four rng_keep calls.
__device__ __forceinline__ void rng_keep4(uint64_t key, uint64_t idx, float p, float inv_keep, float (&f)[4]) {
  const uint32_t lo = (uint32_t)idx + (uint32_t)key;
  const uint32_t mid = (uint32_t)(key >> 32) ^ ((uint32_t)(idx >> 32) * 0x85ebca77u);
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    uint32_t x = lo + (uint32_t)q;
    x ^= x >> 16; x *= 0x21f0aaadu;
    x ^= mid;
    x ^= x >> 15; x *= 0x735a2d97u;
    x ^= x >> 15;
    f[q] = ((float)(x >> 8) * 5.9604644775390625e-8f) >= p ? inv_keep : 0.0f;
  }
}

// ---- hardware transcendentals with the result made safe to consume (every kernel of this library takes exp / log / rcp / rsq /
// sqrt through these wrappers; each is ~1 ulp, i.e. ~1e-7 relative, far inside the path's tolerances).
// v_exp_f32 / v_rcp_f32 run on the quarter-rate transcendental pipe (four 16-lane passes). In the fused gate-score epilogue of the
// bf16x3 contraction kernels the instruction that consumed such a result directly behind it read STALE values in lanes 48-63 (the
// last pass) in ~0.1 % of the wavefronts -- timing dependent, only with two waves per SIMD, invisible to small tests
// (tools/gate_stress.py reproduces it at 131072 rows). hipcc (ROCm 7.2) does not pad this trans -> VALU use on gfx950, so the
// padding is explicit: the op and two wait states travel together in one asm statement.
__device__ __forceinline__ float hw_exp2(float x) {
  float r;
  asm("v_exp_f32 %0, %1\n\ts_nop 1" : "=v"(r) : "v"(x));
  return r;
}
__device__ __forceinline__ float hw_rcp(float x) {
  float r;
  asm("v_rcp_f32 %0, %1\n\ts_nop 1" : "=v"(r) : "v"(x));
  return r;
}
__device__ __forceinline__ float hw_exp(float x) { return hw_exp2(x * 1.44269504088896340736f); }
__device__ __forceinline__ float hw_log2(float x) {
  float r;
  asm("v_log_f32 %0, %1\n\ts_nop 1" : "=v"(r) : "v"(x));
  return r;
}
__device__ __forceinline__ float hw_log(float x) { return hw_log2(x) * 0.693147180559945309f; }
__device__ __forceinline__ float hw_rsq(float x) {
  float r;
  asm("v_rsq_f32 %0, %1\n\ts_nop 1" : "=v"(r) : "v"(x));
  return r;
}
__device__ __forceinline__ float hw_sqrt(float x) {
  float r;
  asm("v_sqrt_f32 %0, %1\n\ts_nop 1" : "=v"(r) : "v"(x));
  return r;
}

// ---- activations ----
enum { ACT_NONE = 0, ACT_RELU = 1, ACT_TANH = 2, ACT_SIGMOID = 3 };

__device__ __forceinline__ float act_apply(int act, float v) {
  switch (act) {
    case ACT_RELU: return v > 0.0f ? v : 0.0f;
    // hardware exp2/rcp forms (v_exp_f32, v_rcp_f32: ~1 ulp each): abs error <= ~2e-7 against libm, a handful of instructions
    // instead of ~40 (tanhf) + a full-precision divide -- the gate contraction's epilogue evaluates 100 M of these per launch
    case ACT_TANH: {
      const float t = hw_exp(-2.0f * fabsf(v));
      return copysignf((1.0f - t) * hw_rcp(1.0f + t), v);
    }
    case ACT_SIGMOID: return hw_rcp(1.0f + hw_exp(-v));
    default: return v;
  }
}
// derivative expressed through the activation OUTPUT y
__device__ __forceinline__ float act_grad_from_out(int act, float y) {
  switch (act) {
    case ACT_RELU: return y > 0.0f ? 1.0f : 0.0f;
    case ACT_TANH: return 1.0f - y * y;
    case ACT_SIGMOID: return y * (1.0f - y);
    default: return 1.0f;
  }
}

// ---- wave64 reductions ----
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
