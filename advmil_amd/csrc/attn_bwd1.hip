// Single-pass backward of the fused attention core (see attn.hip for the forward, the operand format and the two-launch backward).
// Reference: the autograd backward of nn.MultiheadAttention's softmax(QK^T)V (/root/reference model/backbone_utils.py:113-127).
//
// The two-launch backward recomputes S and dP twice (queries stationary for dQ, keys stationary for dK / dV): seven contractions
// where the mathematics has five. Here the keys are stationary and ONE pass over the queries produces all three gradients:
//   per trip of 32 streamed queries and per wave (32 stationary keys):  S = Q K^T,  dPd = dO V^T  ->  Pd, dS (registers, lane = key)
//     dV^T[d, key] += dO^T[d, q] . Pd[q, key]        dK^T[d, key] += Q^T[d, q] . dS[q, key]          (as attn_bwd_dkv_kernel)
//     dQ^T[d, q]    = K^T[d, key] . dS^T[key, q]   over the workgroup's 256 keys  -> one partial tile per (key block, query tile)
// dQ contracts over the keys, which live across lanes and waves: dS goes through LDS once (the hi / lo fragments that feed the dK
// contraction are written as an image [256 keys][32 queries] of 64-byte rows, 16-byte units XOR-swizzled by (row >> 2) & 3, and
// read back with ds_read_b64_tr_b16 as MFMA B fragments whose k slots walk the keys), against K^T fragments read the same way
// from a resident image of the workgroup's K rows. Four waves -- one per SIMD, the two wave quartets alternating trip by trip --
// own the four (head-dim block, key part) tasks of a trip while the other four run ahead into the next trip's scores; the key parts
// of a tile meet in LDS (fixed order) and the tile is stored UNSCALED into the partial slab of the workgroup's key block,
//   part[key block][Ltot, H*HD];   dQ = scale * sum over the bag's key blocks, in block order  (attn_dq_reduce_kernel).
// No atomics: results are deterministic. Matrix instructions per (wave, 64 queries): 108 (two-launch form: 60 + 84).
//
// LDS (152.8 KB, one workgroup per CU): ring of 2 x 16 KB [Q hi | Q lo | dO hi | dO lo] x 32 rows by LDS-DMA, K image 64 KB,
// dS image 32 KB, 24 KB of accumulator exchange (every dQ task parks its accumulator: nothing of a tile stays in registers across
// the next trip's scores), side data (lse, D, dropout row key) of two trips.
// Barriers per trip: A (the previous trip's dQ readers are done with the dS image; their accumulators are parked) and
// B (dS image complete, next tile and its side data visible, this trip's ring slot free).
// The memory counter: vmcnt counts loads AND stores in issue order. The only vector-memory operations of the loop are the ring's
// DMA pieces (2 per wave and trip, issued behind barrier B) and the partial-tile stores (behind barrier A: every wave combines and
// stores one quarter of the previous trip's tile, at most one 16-byte store per lane); a wave waits for its pieces of the next tile
// with vmcnt(stores it issued since); the side data (lse, D) ride the DMA as one more
// piece, so that no load result lives in a register and no compiler-inserted vmcnt(0) sits on the stores' latency.
#include "attn_core.h"

#define B1_QT 32                          // streamed queries per trip
#define B1_PL 4096                        // one ring plane tile: 32 rows x 128 B
#define B1_SLOT (4 * B1_PL)               // [Q hi | Q lo | dO hi | dO lo]
#define B1_RING (2 * B1_SLOT)
#define B1_KPL (AT_QB * 128)              // K image plane: 256 rows x 128 B
#define B1_KIMG_OFF B1_RING
#define B1_DSPL (AT_QB * 64)              // dS image plane: 256 key rows x 64 B (32 queries)
#define B1_DS_OFF (B1_KIMG_OFF + 2 * B1_KPL)
#define B1_SCR_OFF (B1_DS_OFF + 2 * B1_DSPL)
#define B1_SCR_B (4 * 6144)            // parked dQ accumulators: per key part [head-dim block 0: 4 KB | block 1: <= 4 KB]
#define B1_AUX_OFF (B1_SCR_OFF + B1_SCR_B)
#define B1_TOTAL (B1_AUX_OFF + 2 * 3 * B1_QT * 4)
#define B1_PPW 2                          // ring DMA pieces per wave and trip

// Transposed fragment of the dS image (64-byte rows): lane (i = lane & 31 -> query i, half) <- key rows r16 + 4 half + {0..3}
// (k slots 0-3) and r16 + 8 + 4 half + {0..3} (slots 4-7): the k-slot <-> row map of attn_core.h::frag_tr. r16 is a multiple of 16,
// so both rows' swizzle terms depend on the lane only.
struct Tr64 {
  unsigned a, b;
  __device__ __forceinline__ void init(unsigned base, int lane) {
    const int half = lane >> 5, q4 = (lane & 15) >> 2, bb = (lane >> 4) & 1, e = lane & 3;
    const int u = 2 * bb + (e >> 1);
    a = base + (unsigned)((4 * half + q4) * 64 + ((u ^ half) << 4) + (e & 1) * 8);
    b = base + (unsigned)((8 + 4 * half + q4) * 64 + ((u ^ ((half + 2) & 3)) << 4) + (e & 1) * 8);
    asm volatile("" : "+v"(a), "+v"(b));
  }
  __device__ __forceinline__ bf16x8 read(int off) const {
    union { bf16x4_t q[2]; bf16x8 v; } r;
    r.q[0] = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((LDS_AS bf16x4_t*)(size_t)(a + (unsigned)off));
    r.q[1] = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((LDS_AS bf16x4_t*)(size_t)(b + (unsigned)off));
    return r.v;
  }
};

template <int HD, bool DROP>
__global__ __launch_bounds__(512, 2) void attn_bwd_one_kernel(AttnArgs a) {
  constexpr int KS = HD / 16, DT = (HD + 31) / 32, UN = HD / 8;
  // dQ tasks of a trip: 4 (one per wave of the quartet) = (DT / NDT) head-dim block groups x KSPLIT key parts. head_dim 48 gives a
  // wave BOTH head-dim blocks of a key quarter (the dS fragments are read once for the two: 12 transposed reads per 6 matrix
  // instructions instead of 16 -- the stage is bound by LDS bandwidth) and parks 4 + 2 KB (block 1 holds 16 head dims); head_dim 64
  // would need 32 KB of exchange space for that and keeps one block per wave.
  constexpr int NDT = HD == 48 ? 2 : 1;
  constexpr int KSPLIT = 4 * NDT / DT;
  constexpr int KSTEPS = (AT_QB / 16) / KSPLIT;        // 16-key k-steps per part
  constexpr int PARTB = NDT == 2 ? 6144 : 4096 * DT;   // exchange bytes per key part: [block 0 | block 1]
  __shared__ __attribute__((aligned(16))) unsigned char smem[B1_TOTAL];
  float* const sAux = reinterpret_cast<float*>(smem + B1_AUX_OFF);       // per parity: lse[32] | D[32] | rowkey[32]
  const int tid = threadIdx.x, lane = tid & 63, j = lane & 31, half = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int H = a.H;
  const int h = blockIdx.x % H;
  const int rest = blockIdx.x / H;
  const int kt = rest % a.ntile, g = rest / a.ntile;
  const int64_t row0 = a.ptr ? a.ptr[g] : 0;
  const int64_t Lg = a.ptr ? a.ptr[g + 1] - row0 : a.Ltot;
  if ((int64_t)kt * AT_QB >= Lg) return;
  const int D = H * HD;
  const int64_t ldq = a.ldq;
  const int T = (int)((Lg + B1_QT - 1) / B1_QT);

  // ---- ring DMA: wave w brings plane w >> 1 (Q hi, Q lo, dO hi, dO lo), rows 16 (w & 1) + 8 p + (lane >> 3), p = 0, 1
  const int dpl = wave >> 1;
  const char* const dsrc = dpl < 2 ? reinterpret_cast<const char*>((dpl == 0 ? a.qkv_hi : a.qkv_lo) + row0 * ldq + h * HD)
                                   : reinterpret_cast<const char*>((dpl == 2 ? a.do_hi : a.do_lo) + row0 * D + h * HD);
  const int dpitch = dpl < 2 ? (int)ldq : D;            // halfwords
  auto ring_issue = [&](int slot, int64_t tile_row0) {
    int l2 = threadIdx.x & 63;                          // lane coordinates derived at the issue (nothing of them lives across the trip)
    asm volatile("" : "+v"(l2));
    const int64_t rem = Lg - 1 - tile_row0;
    const int lim = rem < (int64_t)(B1_QT - 1) ? (int)rem : B1_QT - 1;
    const char* tb = dsrc + tile_row0 * dpitch * 2;
    unsigned char* dst = smem + slot * B1_SLOT + dpl * B1_PL + 16 * (wave & 1) * 128;
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      const int lrow = 16 * (wave & 1) + 8 * p + (l2 >> 3);
      const int usrc = (l2 & 7) ^ at_sw(lrow);
      const int rl = lrow < lim ? lrow : lim;
      if (usrc < UN)
        lds_dma16(tb + (unsigned)(rl * dpitch + usrc * 8) * 2u, lds_addr(dst + p * 1024));
    }
  };
  const uint64_t key64 = DROP ? rng_key(*a.seed, a.stream_id) : 0;
  const int64_t rowoff = a.rng_rowoff ? a.rng_rowoff[g] : 0;
  // side data of a trip's queries: lse and D ride the DMA too -- wave 0 brings them as ONE piece of 4 bytes per lane (lanes 0-31:
  // lse of query lane, lanes 32-63: D of query lane - 32; queries past the bag read its last row, their probabilities are forced to
  // zero) straight into the trip's [lse 32 | D 32 | rowkey 32] block; the dropout row keys are hashed by lanes 0-3 of every wave.
  // (Scalar loads would count on lgkmcnt, which every LDS read of the loop waits on; vector loads would make the compiler put a
  // vmcnt(0) on the partial-tile stores.)
  auto aux_issue = [&](int tile) {
    if (wave == 0) {
      int l2 = threadIdx.x & 63;                        // (the per-lane source pointer is formed at the issue: carried, it spilled)
      asm volatile("" : "+v"(l2));
      int64_t qq = (int64_t)tile * B1_QT + (l2 & 31);
      if (qq > Lg - 1) qq = Lg - 1;
      const float* const src = ((l2 & 32) ? a.dsum : a.lse) + (row0 + qq) * H + h;
      lds_dma4(src, lds_addr(sAux + (tile & 1) * 3 * B1_QT));
    }
  };
  auto aux_keys = [&](int tile) {
    if (DROP && lane < 4) {
      const int64_t qq = (int64_t)tile * B1_QT + wave * 4 + lane;
      reinterpret_cast<uint32_t*>(sAux)[(tile & 1) * 3 * B1_QT + 2 * B1_QT + wave * 4 + lane] =
          attn_row_key(key64, (uint64_t)(row0 + rowoff + qq) * (uint64_t)H + (uint64_t)h);
    }
  };

  // ---- prologue: the workgroup's K rows as an LDS image (both planes), tiles 0 and 1, side data of tile 0
  {
    const bf16raw* const Kh = a.qkv_hi + row0 * ldq + D + h * HD;
    const bf16raw* const Kl = a.qkv_lo + row0 * ldq + D + h * HD;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const int r = 32 * wave + 8 * p + (lane >> 3);
      const int us = (lane & 7) ^ at_sw(r);
      int64_t kr = (int64_t)kt * AT_QB + r;
      if (kr > Lg - 1) kr = Lg - 1;                 // rows past the bag re-read its last row (their dS is zeroed)
      unsigned char* dst = smem + B1_KIMG_OFF + (32 * wave + 8 * p) * 128;
      if (us < UN) {
        lds_dma16(Kh + kr * ldq + us * 8, lds_addr(dst));
        lds_dma16(Kl + kr * ldq + us * 8, lds_addr(dst + B1_KPL));
      }
    }
  }
  aux_issue(0);
  ring_issue(0, 0);
  {                                                   // pad units (head dims HD .. 32 DT - 1) of every 128-byte row: ring + K image
    constexpr int UP = 4 * DT - UN;
    if (UP > 0) {
      constexpr int ROWS = (B1_RING + 2 * B1_KPL) / 128;
      for (int e = tid; e < ROWS * UP; e += 512) {
        const int pu = e % UP, r = e / UP;
        *reinterpret_cast<uint4*>(smem + r * 128 + (((UN + pu) ^ at_sw(r)) << 4)) = make_uint4(0u, 0u, 0u, 0u);
      }
    }
  }
  TrAddr tra;
  tra.init(lane);
  const unsigned smem_l = lds_addr(smem);
  unsigned trk_a[NDT], trk_b[NDT];                    // K image: this wave's dQ task (head-dim block(s), first row of its key part)
  {
    const int tx_ = wave & 3;
    const int tkp_ = NDT == 2 ? tx_ : tx_ / DT;       // the key part's first row is folded into the bases:
    const unsigned kp0 = (unsigned)(tkp_ * KSTEPS * 16);              // every read of the dQ stage is base + immediate
#pragma unroll
    for (int n = 0; n < NDT; ++n) {
      const bool hi_blk = NDT == 2 ? n == 1 : (tx_ % DT) != 0;
      trk_a[n] = smem_l + B1_KIMG_OFF + kp0 * 128 + (hi_blk ? tra.a[DT - 1] : tra.a[0]);
      trk_b[n] = smem_l + B1_KIMG_OFF + kp0 * 128 + (hi_blk ? tra.b[DT - 1] : tra.b[0]);
      asm volatile("" : "+v"(trk_a[n]), "+v"(trk_b[n]));
    }
  }
  auto ktr = [&](int n, int off) {                    // K^T fragment (head dims of block n of the task x 16 keys at byte offset off)
    union { bf16x4_t q[2]; bf16x8 v; } r;
    r.q[0] = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((LDS_AS bf16x4_t*)(size_t)(trk_a[n] + (unsigned)off));
    r.q[1] = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((LDS_AS bf16x4_t*)(size_t)(trk_b[n] + (unsigned)off));
    return r.v;
  };
  Tr64 trd;                                           // dS image, from this wave's key part on
  trd.init(smem_l + B1_DS_OFF + (unsigned)((NDT == 2 ? (wave & 3) : (wave & 3) / DT) * KSTEPS * 16 * 64), lane);

  const int64_t key = (int64_t)kt * AT_QB + wave * 32 + j;
  const bool kok = key < Lg;
  const bool ktail = (int64_t)(kt + 1) * AT_QB > Lg;  // workgroup-uniform: some key rows are past the bag
  bf16x8 kh[KS], kl[KS], vh[KS], vl[KS];
  {
    const int64_t ko = (row0 + (kok ? key : 0)) * ldq + D + h * HD;
    load_row_frags<HD>(a.qkv_hi + ko, a.qkv_lo + ko, kok, half, kh, kl);
    load_row_frags<HD>(a.qkv_hi + ko + D, a.qkv_lo + ko + D, kok, half, vh, vl);
  }
  const float ik = DROP ? a.inv_keep : 1.f;
  const float c = a.scale_log2e;
  const uint32_t kgold = (uint32_t)(key >> 2) * AT_GOLD;        // this lane's key group
  const int kbyte = (int)(key & 3) * 8;
  // dS image: this lane's row (its key) and the swizzle of that row
  // dQ task of this wave in the trips of its quartet
  const int tx = wave & 3, tdt = NDT == 2 ? 0 : tx % DT, tkp = NDT == 2 ? tx : tx / DT;

  f32x16 dk[DT], dv[DT];
#pragma unroll
  for (int dt = 0; dt < DT; ++dt)
#pragma unroll
    for (int r = 0; r < 16; ++r) { dk[dt][r] = 0.f; dv[dt][r] = 0.f; }
  // Combining a trip's dQ tile: EVERY wave sums one quarter (4 head dims per lane: accumulator registers 4 ci .. 4 ci + 3) of one
  // head-dim block over the parked key parts, in part order, and stores it UNSCALED into this key block's partial slab -- two LDS
  // reads, four adds and at most one 16-byte store per wave behind barrier A (as the work of two waves it sat on the trip's critical
  // path: 92 us of 1 108 on 16 x 2048 tokens).
  const int cdt = wave >> 2, ci = wave & 3;
  const bool cown = cdt < DT && 32 * cdt + 8 * ci < HD;          // this wave's quarter exists (wave-uniform)
  auto combine_load = [&](float4 (&x)[KSPLIT]) {
    int l2 = threadIdx.x & 63;                          // (lane offset derived here: nothing of it lives across the trip)
    asm volatile("" : "+v"(l2));
    const unsigned char* const cscr = smem + B1_SCR_OFF + cdt * 4096 + ci * 1024 + l2 * 16;
#pragma unroll
    for (int kp = 0; kp < KSPLIT; ++kp) x[kp] = *reinterpret_cast<const float4*>(cscr + kp * PARTB);
  };
  auto combine_store = [&](int tp, const float4 (&x)[KSPLIT]) {
    float4 acc = x[0];
#pragma unroll
    for (int kp = 1; kp < KSPLIT; ++kp) { acc.x += x[kp].x; acc.y += x[kp].y; acc.z += x[kp].z; acc.w += x[kp].w; }
    const int64_t q2 = (int64_t)tp * B1_QT + j;
    if (q2 < Lg)
      *reinterpret_cast<float4*>(a.dq_part + ((int64_t)kt * a.Ltot + row0 + q2) * D + h * HD + 32 * cdt + 8 * ci + 4 * half) = acc;
  };

  // dQ^T[d, q] part of the trip whose dS image is in LDS = K^T[d, key] . dS^T[key, q] over this wave's key part (two accumulators:
  // even / odd k-steps, so that consecutive matrix instructions do not wait on each other), parked for the combining wave
  auto dq_stage = [&]() {
#ifndef B1_ABL_NO_DQ
    f32x16 q0, q1;                                     // NDT == 2: head-dim blocks 0 / 1; else even / odd k-steps of the one block
#pragma unroll
    for (int r = 0; r < 16; ++r) { q0[r] = 0.f; q1[r] = 0.f; }
    if constexpr (NDT == 2) {
#pragma unroll
      for (int kk = 0; kk < KSTEPS; ++kk) {
        const bf16x8 sh = trd.read(kk * 1024), sl = trd.read(B1_DSPL + kk * 1024);
        const bf16x8 th0 = ktr(0, kk * 2048), tl0 = ktr(0, B1_KPL + kk * 2048);
        const bf16x8 th1 = ktr(1, kk * 2048), tl1 = ktr(1, B1_KPL + kk * 2048);
        q0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tl0, sh, q0, 0, 0, 0);
        q1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tl1, sh, q1, 0, 0, 0);
        q0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(th0, sl, q0, 0, 0, 0);
        q1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(th1, sl, q1, 0, 0, 0);
        q0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(th0, sh, q0, 0, 0, 0);
        q1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(th1, sh, q1, 0, 0, 0);
      }
      unsigned char* sc = smem + B1_SCR_OFF + tkp * PARTB + lane * 16;
#pragma unroll
      for (int i = 0; i < 4; ++i) *reinterpret_cast<float4*>(sc + i * 1024) = make_float4(q0[4 * i], q0[4 * i + 1], q0[4 * i + 2], q0[4 * i + 3]);
#pragma unroll
      for (int i = 0; i < (HD - 32) / 8; ++i)
        *reinterpret_cast<float4*>(sc + 4096 + i * 1024) = make_float4(q1[4 * i], q1[4 * i + 1], q1[4 * i + 2], q1[4 * i + 3]);
    } else {
#pragma unroll
      for (int kk = 0; kk < KSTEPS; kk += 2) {
        const bf16x8 th0 = ktr(0, kk * 2048), tl0 = ktr(0, B1_KPL + kk * 2048);
        const bf16x8 sh0 = trd.read(kk * 1024), sl0 = trd.read(B1_DSPL + kk * 1024);
        const bf16x8 th1 = ktr(0, (kk + 1) * 2048), tl1 = ktr(0, B1_KPL + (kk + 1) * 2048);
        const bf16x8 sh1 = trd.read((kk + 1) * 1024), sl1 = trd.read(B1_DSPL + (kk + 1) * 1024);
        q0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tl0, sh0, q0, 0, 0, 0);
        q1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tl1, sh1, q1, 0, 0, 0);
        q0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(th0, sl0, q0, 0, 0, 0);
        q1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(th1, sl1, q1, 0, 0, 0);
        q0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(th0, sh0, q0, 0, 0, 0);
        q1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(th1, sh1, q1, 0, 0, 0);
      }
      unsigned char* sc = smem + B1_SCR_OFF + tkp * PARTB + tdt * 4096 + lane * 16;
#pragma unroll
      for (int i = 0; i < 4; ++i)
        *reinterpret_cast<float4*>(sc + i * 1024) =
            make_float4(q0[4 * i] + q1[4 * i], q0[4 * i + 1] + q1[4 * i + 1], q0[4 * i + 2] + q1[4 * i + 2], q0[4 * i + 3] + q1[4 * i + 3]);
    }
#endif
  };

  at_wait_vmcnt<0>();
  aux_keys(0);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  at_barrier();                                  // K image, tile 0 and its side data are in LDS
  for (int t = 0; t < T; ++t) {
    // (behind barrier B of trip t - 1: slot (t + 1) & 1 is free)
    if (t + 1 < T) { ring_issue((t + 1) & 1, (int64_t)(t + 1) * B1_QT); aux_issue(t + 1); aux_keys(t + 1); }
    if (t > 0 && (wave >> 2) == ((t - 1) & 1)) dq_stage();
    const unsigned char* sQ = smem + (t & 1) * B1_SLOT;
    TrBase trb;
    trb.set(smem_l + (t & 1) * B1_SLOT, tra);
    const float* ax = sAux + (t & 1) * 3 * B1_QT;
    const int64_t qb = (int64_t)t * B1_QT;
    const bool tail = qb + B1_QT > Lg;
    // ---- scores and dPd of this wave's 32 keys against the trip's 32 queries
    f32x16 s, dp;
#pragma unroll
    for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const bf16x8 ah = frag_rows(sQ, j, 2 * ks + half), al = frag_rows(sQ + B1_PL, j, 2 * ks + half);
      const bf16x8 bh = frag_rows(sQ + 2 * B1_PL, j, 2 * ks + half), bl = frag_rows(sQ + 3 * B1_PL, j, 2 * ks + half);
      s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, kh[ks], s, 0, 0, 0);            // S[q, key]   (unscaled)
      dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bl, vh[ks], dp, 0, 0, 0);          // dPd[q, key] = dO . V^T
      s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, kl[ks], s, 0, 0, 0);
      dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bh, vl[ks], dp, 0, 0, 0);
      s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, kh[ks], s, 0, 0, 0);
      dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bh, vh[ks], dp, 0, 0, 0);
    }
    uint32_t hq[4];
    if (DROP) {                                  // quad lane e hashes the queries (e) + 8 rg + 4 half, rg = 0..3, for the quad's key group
      const uint32_t* rkp = reinterpret_cast<const uint32_t*>(ax) + 2 * B1_QT;
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) hq[rg] = attn_mix(rkp[8 * rg + 4 * half + (lane & 3)] + kgold);
    }
    float pd[16];
#pragma unroll
    for (int rg = 0; rg < 4; ++rg) {
      const int qo = 8 * rg + 4 * half;
      const float4 l4 = *reinterpret_cast<const float4*>(ax + qo);
      const float4 d4 = *reinterpret_cast<const float4*>(ax + B1_QT + qo);
      const float dvv[4] = {d4.x, d4.y, d4.z, d4.w};
      uint32_t hx[4] = {0u, 0u, 0u, 0u};
      if (DROP) {                                // hash of query qo + e lives in quad lane e
        hx[0] = (uint32_t)__builtin_amdgcn_mov_dpp((int)hq[rg], 0x00, 0xf, 0xf, false);
        hx[1] = (uint32_t)__builtin_amdgcn_mov_dpp((int)hq[rg], 0x55, 0xf, 0xf, false);
        hx[2] = (uint32_t)__builtin_amdgcn_mov_dpp((int)hq[rg], 0xaa, 0xf, 0xf, false);
        hx[3] = (uint32_t)__builtin_amdgcn_mov_dpp((int)hq[rg], 0xff, 0xf, 0xf, false);
      }
      float e0 = fmaf(s[4 * rg], c, -l4.x), e1 = fmaf(s[4 * rg + 1], c, -l4.y), e2 = fmaf(s[4 * rg + 2], c, -l4.z), e3 = fmaf(s[4 * rg + 3], c, -l4.w);
      hw_exp2x4(e0, e1, e2, e3);
      if (tail) {                                // queries past the bag: one uniform branch per group
        const int lim = (int)(Lg - qb);
        if (qo + 0 >= lim) e0 = 0.f;
        if (qo + 1 >= lim) e1 = 0.f;
        if (qo + 2 >= lim) e2 = 0.f;
        if (qo + 3 >= lim) e3 = 0.f;
      }
      const float pv[4] = {e0, e1, e2, e3};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int r = 4 * rg + e;
        const float p = pv[e];
        const float kf = (!DROP || ((hx[e] >> kbyte) & 0xffu) >= a.drop_thr) ? ik : 0.f;      // keep / (1 - p)
        pd[r] = p * kf;
        s[r] = p * fmaf(dp[r], kf, -dvv[e]);                    // dS[q, key]
      }
    }
    // ---- dV^T[d, key] += dO^T[d, q] . Pd[q, key]   (in front of barrier A: it does not touch the dS image)
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
      float v[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = pd[8 * s2 + e];
      bf16x8 ph, pl;
      split8(v, ph, pl);
      bf16x8 gh[DT], gl[DT];
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) { gh[dt] = frag_tr_pre(trb, 2 * B1_PL, 16 * s2, dt); gl[dt] = frag_tr_pre(trb, 3 * B1_PL, 16 * s2, dt); }
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) dv[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(gl[dt], ph, dv[dt], 0, 0, 0);
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) dv[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(gh[dt], pl, dv[dt], 0, 0, 0);
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) dv[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(gh[dt], ph, dv[dt], 0, 0, 0);
    }
    if (ktail) {                                 // keys past the bag: their dS must not reach dQ (dK / dV rows of such keys are never stored)
#pragma unroll
      for (int r = 0; r < 16; ++r) s[r] = kok ? s[r] : 0.f;
    }
    // dS fragments and the first Q^T fragments of the dK contraction are formed in front of the barrier (nothing of them depends on it)
    Frag8 fh[2], fl[2];
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
      float w[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) w[e] = s[8 * s2 + e];
      split8(w, fh[s2].v, fl[s2].v);
    }
    bf16x8 qh0[DT], ql0[DT];                     // (block 0 only: with both blocks in flight across the barrier the HD >= 48 kernels spill)
    qh0[0] = frag_tr_pre(trb, 0, 0, 0); ql0[0] = frag_tr_pre(trb, B1_PL, 0, 0);
    // ---- barrier A: trip t - 1's dQ readers have left the dS image, their accumulators are parked
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#ifndef B1_ABL_NO_A
    at_barrier();
#endif
    int stores = 0;
    float4 cx[KSPLIT];
    const bool comb = t > 0 && cown;
#ifndef B1_ABL_NO_STORE
    if (comb) combine_load(cx);                  // (requested first: the image writes run under their latency)
#endif
    // ---- dS^T image (fragment slots 0-3 / 4-7 are queries 16 s2 + 4 half + {0..3} / + 8: two 8-byte writes per plane) and
    //      dK^T[d, key] += Q^T[d, q] . dS[q, key]
    int l3 = threadIdx.x & 63;                   // dS image: this lane's row (its key) and the swizzle of that row, derived at the use
    asm volatile("" : "+v"(l3));
    unsigned char* const ds_row = smem + B1_DS_OFF + (32 * wave + (l3 & 31)) * 64 + 8 * (l3 >> 5);
    const int ds_sw = (l3 >> 2) & 3;             // ((32 wave + j) >> 2) & 3
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
      unsigned char* p0 = ds_row + (((2 * s2) ^ ds_sw) << 4);
      unsigned char* p1 = ds_row + (((2 * s2 + 1) ^ ds_sw) << 4);
      *reinterpret_cast<uint2*>(p0) = make_uint2(fh[s2].u.x, fh[s2].u.y);
      *reinterpret_cast<uint2*>(p1) = make_uint2(fh[s2].u.z, fh[s2].u.w);
      *reinterpret_cast<uint2*>(p0 + B1_DSPL) = make_uint2(fl[s2].u.x, fl[s2].u.y);
      *reinterpret_cast<uint2*>(p1 + B1_DSPL) = make_uint2(fl[s2].u.z, fl[s2].u.w);
    }
    __builtin_amdgcn_sched_barrier(0);
#ifndef B1_ABL_NO_STORE
    if (comb) { combine_store(t - 1, cx); stores = 1; }
#endif
    __builtin_amdgcn_sched_barrier(0);
    {
      bf16x8 qh1[DT], ql1[DT];
#pragma unroll
      for (int dt = 1; dt < DT; ++dt) { qh0[dt] = frag_tr_pre(trb, 0, 0, dt); ql0[dt] = frag_tr_pre(trb, B1_PL, 0, dt); }
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) { qh1[dt] = frag_tr_pre(trb, 0, 16, dt); ql1[dt] = frag_tr_pre(trb, B1_PL, 16, dt); }
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) dk[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ql0[dt], fh[0].v, dk[dt], 0, 0, 0);
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) dk[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qh0[dt], fl[0].v, dk[dt], 0, 0, 0);
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) dk[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qh0[dt], fh[0].v, dk[dt], 0, 0, 0);
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) dk[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ql1[dt], fh[1].v, dk[dt], 0, 0, 0);
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) dk[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qh1[dt], fl[1].v, dk[dt], 0, 0, 0);
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) dk[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qh1[dt], fh[1].v, dk[dt], 0, 0, 0);
    }
    if (t + 1 < T) {                             // this wave's pieces of tile t + 1: only the stores issued since may still fly
      if (stores) at_wait_vmcnt<1>(); else at_wait_vmcnt<0>();
    }
    // ---- barrier B: the dS image is complete; tile t + 1 and its side data are visible; slot t & 1 is free
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    at_barrier();
  }
  // ---- the last trip's dQ
  if ((wave >> 2) == ((T - 1) & 1)) dq_stage();
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  at_barrier();
  if (cown) {
    float4 cx[KSPLIT];
    combine_load(cx);
    combine_store(T - 1, cx);
  }

  int tid2 = threadIdx.x;                        // (lane coordinates derived again for the epilogue: see the forward)
  asm volatile("" : "+v"(tid2));
  const int half2 = (tid2 >> 5) & 1;
  const int64_t key2 = (int64_t)kt * AT_QB + (tid2 >> 6) * 32 + (tid2 & 31);
  if (key2 < Lg) {
    float* const krow = a.dqkv + (row0 + key2) * ldq + D + h * HD;
    float* const vrow = krow + D;
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) {
        const int d = 32 * dt + 8 * rg + 4 * half2;
        if (d < HD) {
          *reinterpret_cast<float4*>(krow + d) = make_float4(dk[dt][4 * rg] * a.scale, dk[dt][4 * rg + 1] * a.scale,
                                                             dk[dt][4 * rg + 2] * a.scale, dk[dt][4 * rg + 3] * a.scale);
          *reinterpret_cast<float4*>(vrow + d) = make_float4(dv[dt][4 * rg], dv[dt][4 * rg + 1], dv[dt][4 * rg + 2], dv[dt][4 * rg + 3]);
        }
      }
  }
}

// dQ[row, :] = scale * sum over the bag's key blocks (block order) of part[kb][row, :]  ->  the q columns of dqkv
// grid (row chunks of the longest bag, bags); one float4 per thread and step; the key blocks' loads of a step are independent
__global__ __launch_bounds__(256) void attn_dq_reduce_kernel(const float* __restrict__ part, int64_t slab, float* __restrict__ dqkv,
                                                             int64_t ldq, const int64_t* __restrict__ ptr, int64_t Ltot, int D,
                                                             float scale, int rows_per_wg) {
  const int g = blockIdx.y;
  const int64_t row0 = ptr ? ptr[g] : 0;
  const int64_t Lg = ptr ? ptr[g + 1] - row0 : Ltot;
  const int64_t rb = (int64_t)blockIdx.x * rows_per_wg;
  if (rb >= Lg) return;
  const int nkb = (int)((Lg + AT_QB - 1) / AT_QB);
  const int d4 = D / 4;
  int64_t rows = Lg - rb;
  if (rows > rows_per_wg) rows = rows_per_wg;
  const int64_t n = rows * d4;
  for (int64_t e = threadIdx.x; e < n; e += 256) {
    const int64_t r = row0 + rb + e / d4;
    const int c4 = (int)(e % d4);
    const float4* src = reinterpret_cast<const float4*>(part + r * D) + c4;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    int kb = 0;
    for (; kb + 4 <= nkb; kb += 4) {
      const float4 x0 = *(src + (int64_t)(kb + 0) * (slab / 4)), x1 = *(src + (int64_t)(kb + 1) * (slab / 4));
      const float4 x2 = *(src + (int64_t)(kb + 2) * (slab / 4)), x3 = *(src + (int64_t)(kb + 3) * (slab / 4));
      acc.x = (((acc.x + x0.x) + x1.x) + x2.x) + x3.x; acc.y = (((acc.y + x0.y) + x1.y) + x2.y) + x3.y;
      acc.z = (((acc.z + x0.z) + x1.z) + x2.z) + x3.z; acc.w = (((acc.w + x0.w) + x1.w) + x2.w) + x3.w;
    }
    for (; kb < nkb; ++kb) {
      const float4 x = *(src + (int64_t)kb * (slab / 4));
      acc.x += x.x; acc.y += x.y; acc.z += x.z; acc.w += x.w;
    }
    *reinterpret_cast<float4*>(dqkv + r * ldq + 4 * c4) = make_float4(acc.x * scale, acc.y * scale, acc.z * scale, acc.w * scale);
  }
}

// =====================================================================================
// C ABI
// =====================================================================================
// workspace: D [Ltot, nhead] | dO hi | dO lo | partial slabs [ceil(max_len / 256)][Ltot, nhead*head_dim] fp32
extern "C" size_t advmil_mha_bwd1_workspace_bytes(int64_t Ltot, int nhead, int head_dim, int64_t max_len) {
  const size_t ntile = (size_t)((max_len + AT_QB - 1) / AT_QB);
  return attn_ws_dsum_bytes(Ltot, nhead) + 2 * attn_ws_plane_bytes(Ltot, nhead, head_dim) +
         ntile * (size_t)Ltot * (size_t)nhead * (size_t)head_dim * sizeof(float);
}

extern "C" int advmil_mha_bwd1(const void* qkv_hi, const void* qkv_lo, const float* out, const float* dout, const float* lse,
                               int64_t Ltot, int nhead, int head_dim, int nseg, const int64_t* ptr, int64_t max_len, float drop_p,
                               const uint64_t* seed, uint64_t stream_id, const int64_t* rng_rowoff, float* dqkv, void* ws,
                               size_t ws_bytes, advmil_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  AttnArgs a;
  const int rc = attn_args(a, qkv_hi, qkv_lo, Ltot, nhead, head_dim, nseg, ptr, max_len, drop_p, seed, stream_id, rng_rowoff);
  if (rc) return rc;
  if (!out || !dout || !lse || !dqkv || !ws) return ADVMIL_EINVAL;
  if (((uintptr_t)out & 15) || ((uintptr_t)dout & 15) || ((uintptr_t)dqkv & 15) || ((uintptr_t)ws & 15)) return ADVMIL_EINVAL;
  if (ws_bytes < advmil_mha_bwd1_workspace_bytes(Ltot, nhead, head_dim, max_len)) return ADVMIL_EWORKSPACE;
  float* dsum = (float*)ws;
  bf16raw* g_hi = (bf16raw*)((char*)ws + attn_ws_dsum_bytes(Ltot, nhead));
  bf16raw* g_lo = (bf16raw*)((char*)g_hi + attn_ws_plane_bytes(Ltot, nhead, head_dim));
  float* part = (float*)((char*)g_lo + attn_ws_plane_bytes(Ltot, nhead, head_dim));
  const int prc = attn_launch_bwd_prep(dout, out, Ltot, nhead, head_dim, dsum, g_hi, g_lo, stream);
  if (prc) return prc;
  a.lse = const_cast<float*>(lse); a.do_hi = g_hi; a.do_lo = g_lo; a.dsum = dsum; a.dqkv = dqkv; a.dq_part = part;
  const dim3 grid((unsigned)(a.ntile * nseg * nhead));
  AT_DISPATCH(attn_bwd_one_kernel, grid, stream, a, head_dim);
  ADVMIL_LAUNCH_CHECK();
  const int D = nhead * head_dim;
  const int rows_per_wg = 32;
  const dim3 rgrid((unsigned)((max_len + rows_per_wg - 1) / rows_per_wg), (unsigned)nseg);
  hipLaunchKernelGGL(attn_dq_reduce_kernel, rgrid, dim3(256), 0, stream, (const float*)part, (int64_t)Ltot * D, dqkv, a.ldq, ptr, Ltot, D,
                     a.scale, rows_per_wg);
  ADVMIL_LAUNCH_CHECK();
  return ADVMIL_OK;
}
