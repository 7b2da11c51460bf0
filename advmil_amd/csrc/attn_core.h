// Shared device helpers of the fused attention kernels (attn.hip: forward + two-launch backward; attn_bwd1.hip: single-pass backward).
// See the header comment of attn.hip for the structure (LDS image of a plane tile, swizzle, transposed fragments, k-slot map).
#pragma once
#include "common.h"
#include "bf16split.h"
#include "../../include/advmil_hip.h"

#define AT_KT 64                       // streamed rows per LDS tile
#define AT_NS 4                        // ring slots
#define AT_PLANE_B 8192                // bytes of one plane tile: 64 rows x 128 B
#define AT_SLOT_B (4 * AT_PLANE_B)     // [K hi | K lo | V hi | V lo]
#define AT_QB 256                      // stationary rows per workgroup (8 waves x 32)
#define AT_PPW 4                       // DMA pieces per wave and tile
#define AT_GOLD 0x9E3779B9u
#define GLB_AS __attribute__((address_space(1)))

struct AttnArgs {
  const bf16raw* qkv_hi;  // planes of the packed in-projection output [Ltot, 3*H*HD] (q | k | v), row pitch ldq halfwords
  const bf16raw* qkv_lo;
  int64_t ldq;
  const bf16raw* do_hi;   // bwd: planes of dO [Ltot, H*HD], row pitch H*HD
  const bf16raw* do_lo;
  float* out;             // fwd: O [Ltot, H*HD]
  float* lse;             // [Ltot, H] log2-domain log-sum-exp of the scaled scores
  const float* dsum;      // bwd: D [Ltot, H] = sum_d dO * O
  float* dqkv;            // bwd: [Ltot, 3*H*HD], row pitch ldq
  float* dq_part;         // single-pass bwd: per-key-block partial slabs of dQ, [ntile][Ltot, H*HD] (unscaled)
  const int64_t* ptr;     // [nseg+1] first region row of every bag, or NULL (one bag of Ltot rows)
  const int64_t* rng_rowoff;  // [nseg] added to a bag's local rows to form the dropout stream's row id, or NULL
  int64_t Ltot;
  int nseg, ntile, H;
  float scale_log2e;      // log2(e) / sqrt(head_dim)
  float scale;            // 1 / sqrt(head_dim)
  uint32_t drop_thr;      // floor(p * 256)
  float inv_keep;         // 256 / (256 - drop_thr)
  const uint64_t* seed;
  uint64_t stream_id;
};

// Dev probe (tools/probe/stamp_attn.sh builds a copy of the library with -DAT_STAMP): shader-clock stamps of waves 0 and 4 of
// workgroup 0 at the section boundaries of the forward's tile loop; never compiled into the product library.
#ifdef AT_STAMP
__device__ unsigned long long g_at_stamps[2 * 1024];
extern "C" int advmil_debug_attn_stamps(unsigned long long* dst, int n) { return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_at_stamps), (size_t)n * 8); }
#define AT_ST(tag)                                                                                          \
  do {                                                                                                      \
    if (blockIdx.x == 0 && (wave & 3) == 0 && lane == 0 && st_i < 1024)                                     \
      g_at_stamps[(wave >> 2) * 1024 + st_i++] = (__builtin_amdgcn_s_memtime() << 4) | (unsigned)(tag);     \
  } while (0)
#else
#define AT_ST(tag) do { } while (0)
#endif

template <int N>
__device__ __forceinline__ void at_wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ void at_barrier() {
  __builtin_amdgcn_s_barrier();          // raw: __syncthreads would drain vmcnt (an LDS-DMA is a pending LDS write)
  asm volatile("" ::: "memory");
}

__device__ __forceinline__ uint32_t attn_row_key(uint64_t key, uint64_t row_id) { return (uint32_t)(splitmix64(key + row_id) >> 32); }
// the per-(query, key group) hash: 4 keys, one byte each
__device__ __forceinline__ uint32_t attn_mix(uint32_t x) {
  x ^= x >> 15; x *= 0x7feb352du;
  x ^= x >> 15;
  return x;
}

// swizzle of the 16-byte units of plane-tile row r
__device__ __forceinline__ int at_sw(int r) { return (((r >> 1) & 1) << 2) | ((r >> 2) & 3); }

// row fragment (8 consecutive head dims of row `row`, unit u): lane (i = lane & 31, half) <- plane[row][8u .. 8u + 7]
__device__ __forceinline__ bf16x8 frag_rows(const unsigned char* __restrict__ plane, int row, int u) {
  Frag8 f;
  f.u = *reinterpret_cast<const uint4*>(plane + row * 128 + ((u ^ at_sw(row)) << 4));
  return f.v;
}
// transposed fragment for MFMA row block `dt` (head dims 32 dt ...): lane (i = lane & 31 -> head dim 32 dt + i, half) <- rows
// r0 + {0..3} (k slots 0-3) and r0 + 8 + {0..3} (slots 4-7); r0 already holds the half's offset (16 s + 4 half), a multiple of 4
__device__ __forceinline__ bf16x8 frag_tr(const unsigned char* __restrict__ plane, int r0, int dt, int lane) {
  const int q4 = (lane & 15) >> 2, b = (lane >> 4) & 1, e = lane & 3;
  const int u = 4 * dt + 2 * b + (e >> 1);
  const int ra = r0 + q4, rb = ra + 8;
  const unsigned char* pa = plane + ra * 128 + ((u ^ at_sw(ra)) << 4) + (e & 1) * 8;
  const unsigned char* pb = plane + rb * 128 + ((u ^ at_sw(rb)) << 4) + (e & 1) * 8;
  union { bf16x4_t q[2]; bf16x8 v; } a;
  a.q[0] = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((LDS_AS bf16x4_t*)(pa));
  a.q[1] = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((LDS_AS bf16x4_t*)(pb));
  return a.v;
}
// The same transposed fragment from per-lane offsets computed ONCE: for r0 = a multiple of 16 plus 4 half, the swizzle terms of
// the two rows r0 + q4 and r0 + q4 + 8 depend on the lane only ((ra >> 1) & 1 = (q4 >> 1) & 1, (ra >> 2) & 3 = half, (rb >> 2) & 3 =
// (half + 2) & 3), so a read is `plane + 128 * (16-row block) + off`: the per-read v_add3 / v_subrev address arithmetic of frag_tr
// (two VALU per ds_read_b64_tr_b16: 68 per tile in the dQ kernel, 136 in the dK/dV kernel) becomes an immediate offset.
struct TrAddr {
  unsigned a[2], b[2];                 // [dt]: byte offsets of the rows 4 half + q4 and 4 half + q4 + 8 inside a 16-row block
  __device__ __forceinline__ void init(int lane) {
    const int half = lane >> 5, q4 = (lane & 15) >> 2, bb = (lane >> 4) & 1, e = lane & 3;
    const int ra = 4 * half + q4, rb = ra + 8;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt) {
      const int u = 4 * dt + 2 * bb + (e >> 1);
      a[dt] = (unsigned)(ra * 128 + ((u ^ at_sw(ra)) << 4) + (e & 1) * 8);
      b[dt] = (unsigned)(rb * 128 + ((u ^ at_sw(rb)) << 4) + (e & 1) * 8);
    }
  }
};
// A tile slot's four read bases (slot address + lane offsets), formed once per tile and made opaque to the optimiser, which otherwise
// re-derives "(t % 3) * slot bytes" per read as an induction variable minus a scalar: one VALU per ds_read_b64_tr_b16. Every read
// of the tile is then base + immediate (16-row block, plane, K / V half of the slot: all < 32 KB).
__device__ __forceinline__ unsigned lds_addr(const void* p) { return (unsigned)(size_t)(LDS_AS const unsigned char*)p; }
struct TrBase {
  unsigned a[2], b[2];
  __device__ __forceinline__ void set(unsigned slot, const TrAddr& ta) {
    a[0] = slot + ta.a[0]; a[1] = slot + ta.a[1]; b[0] = slot + ta.b[0]; b[1] = slot + ta.b[1];
    asm volatile("" : "+v"(a[0]), "+v"(a[1]), "+v"(b[0]), "+v"(b[1]));
  }
};
// off: byte offset of the plane inside the slot; rows16: first row of the 16-row block (a multiple of 16)
__device__ __forceinline__ bf16x8 frag_tr_pre(const TrBase& tb, int off, int rows16, int dt) {
  union { bf16x4_t q[2]; bf16x8 v; } r;
  r.q[0] = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((LDS_AS bf16x4_t*)(size_t)(tb.a[dt] + (unsigned)(off + rows16 * 128)));
  r.q[1] = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((LDS_AS bf16x4_t*)(size_t)(tb.b[dt] + (unsigned)(off + rows16 * 128)));
  return r.v;
}
// 8 fp32 accumulator values -> hi / lo B fragments
__device__ __forceinline__ void split8(const float (&v)[8], bf16x8& hi, bf16x8& lo) {
  union { unsigned u[4]; bf16x8 v; } h, l;
#pragma unroll
  for (int i = 0; i < 4; ++i) split2(v[2 * i], v[2 * i + 1], h.u[i], l.u[i]);
  hi = h.v; lo = l.v;
}
// the stationary operand's fragments, straight from the planes: 8 consecutive head dims of one row at 16 ks + 8 half
template <int HD>
__device__ __forceinline__ void load_row_frags(const bf16raw* __restrict__ hi, const bf16raw* __restrict__ lo, bool ok, int half,
                                               bf16x8 (&fh)[HD / 16], bf16x8 (&fl)[HD / 16]) {
#pragma unroll
  for (int ks = 0; ks < HD / 16; ++ks) {
    Frag8 a, b;
    a.u = make_uint4(0u, 0u, 0u, 0u); b.u = a.u;
    if (ok) {
      a.u = *reinterpret_cast<const uint4*>(hi + 16 * ks + 8 * half);
      b.u = *reinterpret_cast<const uint4*>(lo + 16 * ks + 8 * half);
    }
    fh[ks] = a.v; fl[ks] = b.v;
  }
}
#define MFMA3(acc, ah, al, bh, bl)                                        \
  do {                                                                    \
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc, 0, 0, 0);  \
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc, 0, 0, 0);  \
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc, 0, 0, 0);  \
  } while (0)
// row of accumulator register r in a 32x32 tile: (r & 3) + 8 * (r >> 2) + 4 * half
#define ACC_ROW(r, half) (((r) & 3) + 8 * ((r) >> 2) + 4 * (half))

// The transposed fragments of the last MFMA row block read head dims up to 32 * DT - 1 >= HD: those units are never written by the
// DMA. They are zeroed once (whatever the LDS held before may be NaN patterns; their products only reach discarded rows, but zeros are safer).
template <int HD>
__device__ __forceinline__ void zero_pad_units(unsigned char* smem, int nslots, int tid) {
  constexpr int UN = HD / 8, UP = 4 * ((HD + 31) / 32) - UN;       // data units, pad units per row
  if (UP == 0) return;
  const int total = nslots * 4 * AT_KT * UP;
  for (int e = tid; e < total; e += 512) {
    const int pu = e % UP, r = (e / UP) % AT_KT, pl = e / (UP * AT_KT);
    *reinterpret_cast<uint4*>(smem + pl * AT_PLANE_B + r * 128 + (((UN + pu) ^ at_sw(r)) << 4)) = make_uint4(0u, 0u, 0u, 0u);
  }
}

// LDS-DMA as inline assembly: 64 lanes x 16 (4) bytes from per-lane global addresses to lds_base + 16 (4) * lane. Through the
// builtin (__builtin_amdgcn_global_load_lds) the compiler knows that an LDS write is in flight and puts an s_waitcnt vmcnt(0) in
// front of the NEXT LDS read of any address (it cannot tell the ring slots apart): the tile requested two trips ahead was awaited
// at the top of the trip that requested it, i.e. the ring never ran ahead (round 6; found in the ISA of every kernel of this file).
// The kernels count their pieces themselves (at_wait_vmcnt), so nothing is lost by hiding them: an operation the compiler does not
// know of only makes its own vmcnt waits stricter (completion is in issue order).
__device__ __forceinline__ void lds_dma16(const void* gsrc, unsigned lds_base) {
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(gsrc), "s"(lds_base) : "memory", "m0");
}
__device__ __forceinline__ void lds_dma4(const void* gsrc, unsigned lds_base) {
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dword %0, off" ::"v"(gsrc), "s"(lds_base) : "memory", "m0");
}

// One tile of the streamed operand pair -> ring slot `slot`: this wave's 4 pieces (rows 8 wave .. 8 wave + 7 of each of the four
// planes). `col_a` / `col_b`: halfword column of the head's slice in the two source matrices; rows beyond the bag re-read its
// last row (finite data; their scores are masked).
struct TileDma {
  int lrow, usrc;
  bool on;
  __device__ __forceinline__ void init(int wave, int lane, int UN) {
    lrow = wave * 8 + (lane >> 3);
    usrc = (lane & 7) ^ at_sw(lrow);
    on = usrc < UN;
  }
  // Source addresses = a wave-uniform tile base (scalar registers) + a 32-bit per-lane byte offset formed at the issue: no 64-bit
  // per-lane address lives across the tile loop (under the forward's 128-VGPR bound those were the values that went to scratch).
  __device__ __forceinline__ void issue(unsigned char* smem, int wave, int slot, int64_t tile_row0, int64_t Lg, const bf16raw* a_hi,
                                        const bf16raw* a_lo, int64_t lda, const bf16raw* b_hi, const bf16raw* b_lo, int64_t ldb) const {
    const int64_t rem = Lg - 1 - tile_row0;                   // rows of the bag behind the tile's first one
    const int lim = rem < (int64_t)(AT_KT - 1) ? (int)rem : AT_KT - 1;
    const int rl = lrow < lim ? lrow : lim;
    unsigned char* dst = smem + slot * AT_SLOT_B + wave * 1024;
    if (on) {
      const unsigned oa = (unsigned)(rl * (int)lda + usrc * 8) * 2u, ob = (unsigned)(rl * (int)ldb + usrc * 8) * 2u;
      const char* ta_hi = reinterpret_cast<const char*>(a_hi + tile_row0 * lda);
      const char* ta_lo = reinterpret_cast<const char*>(a_lo + tile_row0 * lda);
      const char* tb_hi = reinterpret_cast<const char*>(b_hi + tile_row0 * ldb);
      const char* tb_lo = reinterpret_cast<const char*>(b_lo + tile_row0 * ldb);
      const unsigned dl = lds_addr(dst);
      lds_dma16(ta_hi + oa, dl);
      lds_dma16(ta_lo + oa, dl + AT_PLANE_B);
      lds_dma16(tb_hi + ob, dl + 2 * AT_PLANE_B);
      lds_dma16(tb_lo + ob, dl + 3 * AT_PLANE_B);
    }
  }
};

// Four v_exp_f32 behind each other and ONE pair of wait states for the last (common.h::hw_exp2: a transcendental's consumer directly
// behind it can read stale lanes; the first three results are three issue slots old when the block ends)
__device__ __forceinline__ void hw_exp2x4(float& a, float& b, float& c, float& d) {
  asm("v_exp_f32 %0, %0\n\tv_exp_f32 %1, %1\n\tv_exp_f32 %2, %2\n\tv_exp_f32 %3, %3\n\ts_nop 1" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
}

// ---- host side: argument block + dispatch by head_dim / dropout
// D = rowsum(dO * O) and the operand planes of dO (attn.hip::attn_bwd_prep_kernel), shared by both backward forms
int attn_launch_bwd_prep(const float* dout, const float* out, int64_t Ltot, int nhead, int head_dim, float* dsum, bf16raw* g_hi,
                         bf16raw* g_lo, hipStream_t stream);
static inline size_t attn_ws_dsum_bytes(int64_t Ltot, int nhead) { return ((size_t)Ltot * (size_t)nhead * sizeof(float) + 15) / 16 * 16; }
static inline size_t attn_ws_plane_bytes(int64_t Ltot, int nhead, int head_dim) { return ((size_t)Ltot * (size_t)nhead * (size_t)head_dim * 2 + 15) / 16 * 16; }
static int attn_args(AttnArgs& a, const void* qkv_hi, const void* qkv_lo, int64_t Ltot, int nhead, int head_dim, int nseg,
                     const int64_t* ptr, int64_t max_len, float drop_p, const uint64_t* seed, uint64_t stream_id,
                     const int64_t* rng_rowoff) {
  if (!qkv_hi || !qkv_lo || Ltot <= 0 || nhead <= 0 || nseg <= 0 || max_len <= 0 || max_len > Ltot) return ADVMIL_EINVAL;
  // d_model / 8 heads of any bcb_dims the reference accepts (model/backbone.py:30-33: 384 -> 48; 128 / 256 / 512 -> 16 / 32 / 64)
  if (head_dim != 16 && head_dim != 32 && head_dim != 48 && head_dim != 64) return ADVMIL_EINVAL;
  if (nseg > 1 && !ptr) return ADVMIL_EINVAL;
  if (drop_p < 0.f || drop_p >= 1.f) return ADVMIL_EINVAL;
  if (((uintptr_t)qkv_hi & 15) || ((uintptr_t)qkv_lo & 15)) return ADVMIL_EINVAL;
  const int64_t ntile = (max_len + AT_QB - 1) / AT_QB;
  if (ntile * nseg * nhead > 0x7fffffffLL) return ADVMIL_EINVAL;
  a.qkv_hi = (const bf16raw*)qkv_hi; a.qkv_lo = (const bf16raw*)qkv_lo; a.ldq = 3 * (int64_t)nhead * head_dim;
  a.do_hi = nullptr; a.do_lo = nullptr;
  a.out = nullptr; a.lse = nullptr; a.dsum = nullptr; a.dqkv = nullptr; a.dq_part = nullptr;
  a.ptr = ptr; a.rng_rowoff = rng_rowoff; a.Ltot = Ltot; a.nseg = nseg; a.ntile = (int)ntile; a.H = nhead;
  a.scale = 1.0f / sqrtf((float)head_dim);
  a.scale_log2e = a.scale * 1.44269504088896340736f;
  const bool drop = seed && drop_p > 0.f;
  a.seed = drop ? seed : nullptr;
  a.stream_id = stream_id;
  a.drop_thr = drop ? (uint32_t)((double)drop_p * 256.0) : 0u;
  a.inv_keep = drop ? 256.0f / (256.0f - (float)a.drop_thr) : 1.0f;
  return ADVMIL_OK;
}

#define AT_LAUNCH(KERNEL, HD_, GRID, STREAM, ARGS)                                                              \
  do {                                                                                                          \
    if ((ARGS).seed) hipLaunchKernelGGL((KERNEL<HD_, true>), GRID, dim3(512), 0, STREAM, ARGS);                 \
    else hipLaunchKernelGGL((KERNEL<HD_, false>), GRID, dim3(512), 0, STREAM, ARGS);                            \
  } while (0)
#define AT_DISPATCH(KERNEL, GRID, STREAM, ARGS, HD_RT)            \
  do {                                                            \
    switch (HD_RT) {                                              \
      case 16: AT_LAUNCH(KERNEL, 16, GRID, STREAM, ARGS); break;  \
      case 32: AT_LAUNCH(KERNEL, 32, GRID, STREAM, ARGS); break;  \
      case 48: AT_LAUNCH(KERNEL, 48, GRID, STREAM, ARGS); break;  \
      default: AT_LAUNCH(KERNEL, 64, GRID, STREAM, ARGS); break;  \
    }                                                             \
  } while (0)
