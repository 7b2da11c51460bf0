// Fused self-attention core of the ESAT layer for gfx950 (nn.MultiheadAttention inside nn.TransformerEncoderLayer,
// /root/reference model/backbone_utils.py:113-127, called from DualTrans_HS.forward, model/backbone.py:188-196):
//   O = dropout(softmax(Q K^T / sqrt(hd))) V   per (bag, head), non-causal, bags never attend across their boundary.
// Flash-style: the [L, L] score matrix never exists in HBM. Forward keeps a running (max, sum) per query and one log-sum-exp per
// (query, head) for the backward; the backward recomputes the probabilities tile by tile in two launches (dQ with the queries
// stationary, dK/dV with the keys stationary) so that no gradient needs a float atomic: results are deterministic.
//
// Arithmetic: "bf16x3" -- every fp32 operand element is split hi + lo (bf16 each) and a product is three
// v_mfma_f32_32x32x16_bf16 (lo.hi + hi.lo + hi.hi) with fp32 accumulation, the same arithmetic as the contraction engine's
// bf16x3 mode (gemm_f32.hip); softmax statistics, exponentials and the rescaling are fp32. The operands ARRIVE split: qkv (and, in
// the backward, dO) are read as two bf16 planes (hi, lo) of the packed [L, 3*H*HD] matrix (advmil_split_planes, or the plane
// output of the in-projection's epilogue), so no kernel here converts an operand; only the probabilities / score gradients, which
// are produced in registers, are split in the kernels.
//
// Structure as shipped (512-thread workgroups = 8 waves, 256 stationary rows per workgroup):
//   * wave w owns 32 rows of the stationary operand (queries for fwd / dQ, keys for dK/dV) as MFMA *columns* (B operand,
//     fragments resident in VGPRs, loaded straight from the planes); the streamed operand comes through LDS in tiles of 64 rows;
//   * the tiles arrive by LDS-DMA (global_load_lds_dwordx4, 1 KB per wave-instruction, 4 per wave per tile: a wave-uniform tile base
//     in scalar registers + a 32-bit per-lane offset) into a ring of [K hi | K lo | V hi | V lo] (or Q / dO) slots -- forward: 2 slots
//     (64 KB), <= 128 VGPRs, `__launch_bounds__(512, 4)`: two workgroups co-resident per CU (census: tools/probe/residency_census.hip);
//     backward kernels: 3 slots (96 KB), one workgroup per CU, one barrier per tile, counted s_waitcnt vmcnt + raw s_barrier;
//     nothing is staged through registers and nothing is converted on the way;
//   * LDS image of a plane tile: [64 rows][128 B] (8 units of 16 B; head_dim <= 64, the units beyond it are never written), the unit
//     stored at position q of row r being unit q ^ sw(r), sw(r) = ((r>>1)&1)<<2 | ((r>>2)&3) -- applied to each lane's SOURCE address
//     (the DMA writes lane-linear). Row fragments (ds_read_b128: lane = row) of any 16 rows distinct mod 16 then cover all 16
//     16-byte slots of the 256-byte bank line, and the four rows of a ds_read_b64_tr_b16 block land in four different 64-byte bank
//     quarters: both read patterns are conflict-free on the same image. The swizzle terms of a transposed read depend on the lane
//     only, so every read of a tile is one of four per-tile bases + an immediate (TrAddr / TrBase);
//   * one 32-row half-tile of the streamed operand at a time goes scores -> softmax / dropout / hi-lo split -> second contraction, all
//     waves in the same phase. Round 3 built and measured the alternatives (anti-phase wave groups, an in-wave software pipeline),
//     round 4 a fully zipped stream (one MFMA : ~7 VALU, order-pinned; tools/probe/attic/): none is faster. On this chip the time of
//     these kernels follows  cycles per SIMD = 32 x MFMAs + ~4.3 x other VALU  within 5 % for all of them, so what
//     pays is removing instructions: address arithmetic folded into immediates, 32-bit tail tests behind a uniform branch, no SLP
//     packing (v_pk_* f32 costs two passes), epilogue coordinates recomputed instead of carried (0 bytes of scratch in all 24 kernels);
//   * contractions over the streamed rows (P.V, dS^T.K, P^T.dO, dS^T.Q) take their second operand straight from the accumulator
//     registers of the score tile: scores are computed transposed (rows = streamed rows, column = lane's stationary row), so a
//     lane holds, for its column, 8 row values per 16-row k-step in exactly the (lane-half, slot) positions an MFMA B fragment wants
//     with the k-slot <-> row map  slot t of half h <-> row 16*s + 8*(t>>2) + 4*h + (t&3); the transpose reads use the same map;
//   * per-query softmax statistics are lane-local (column = query); the two lane halves exchange one max per half-tile. Scores stay
//     unscaled in the accumulators: p = exp2(fma(s, log2(e)/sqrt(hd), -m)) folds the scale into the exponent's FMA.
// Workgroup -> (bag, tile, head) with head = blockIdx % nhead: for nhead = 8 every XCD (blockIdx % 8) serves one head, so the K/V
// panel of a (bag, head) is fetched into exactly one L2.
//
// Dropout on the attention probabilities (train mode): ONE 32-bit hash per (query, group of 4 consecutive keys), one byte per key:
//   keep(i, j) = byte (j & 3) of mix(rowkey(i) + (j >> 2) * 0x9E3779B9) >= floor(p * 256),
//   mix(x): x ^= x >> 15; x *= 0x7feb352d; x ^= x >> 15,   rowkey(i) = high word of splitmix64(key(seed, stream) + row_id(i) * nhead + head);
// restated on the host in advmil_amd/synth.py::attn_dropout_keep. Where a lane walks keys (forward, dQ) that is 8 hashes per 64-key
// tile and a byte compare per probability; where a lane walks queries (dK/dV) the four lanes of a key group compute 4 of the 16
// row hashes each and trade them with DPP quad broadcasts. The drop probability is thereby quantised to 1/256 (0.25 is exact) and
// the kept probabilities are scaled by 256 / (256 - floor(256 p)).
#include "attn_core.h"

// =====================================================================================
// forward
// =====================================================================================

// Forward. Measured on the way here (tools/probe/overlap_probe.hip, attn_model_probe.hip, stamp_attn.sh, rocprofv3 counters): on a
// gfx950 SIMD the time of the matrix instructions and of the vector instructions ADDS (18 ns per MFMA + ~2 ns per VALU, x 0.8 at
// best) whatever the schedule -- lockstep phases, anti-phase wave groups and an in-wave software pipeline all landed at 490-580 us for
// 16 x 2048 tokens, and a tile costs 42 MFMAs plus ~450 VALU. The probe's best overlap is at 4 waves per SIMD, and more waves hide
// the LDS / MFMA latencies without any scheduling effort: this kernel keeps its live state under 128 VGPRs -- one 32-key
// half-tile at a time: S (16) -> P split (16) -> O (32), Q fragments (24) -- so that TWO workgroups (4 waves per SIMD) share a CU,
// each with a double-buffered 64 KB ring; their barriers are independent, so one workgroup's softmax runs under the other's MFMAs.
// LSEIN: the log-sum-exp of every (query, head) is GIVEN (a.lse is read, not written): the train-mode forward of an optimizer step
// whose eval-mode forward ran over the same q | k | v (the handler's forward memo: the generator's weights do not change between
// the discriminator's and the generator's update, model_handler.py:398-425). The probabilities are then exp2(s c - lse) directly,
// as in the backward: no running maximum, no rescaling, no row sum (18 of ~56 vector instructions per 32-key half-tile and lane).
template <int HD, bool DROP, bool LSEIN = false>
__global__ __launch_bounds__(512, HD == 64 ? 2 : 4) void attn_fwd_kernel(AttnArgs a) {
  constexpr int KS = HD / 16, DT = (HD + 31) / 32, UN = HD / 8;
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * AT_SLOT_B];
  const int tid = threadIdx.x, lane = tid & 63, j = lane & 31, half = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int H = a.H;
  const int h = blockIdx.x % H;
  const int rest = blockIdx.x / H;
  const int qt = rest % a.ntile, g = rest / a.ntile;
  const int64_t row0 = a.ptr ? a.ptr[g] : 0;
  const int64_t Lg = a.ptr ? a.ptr[g + 1] - row0 : a.Ltot;
  if ((int64_t)qt * AT_QB >= Lg) return;
  const int D = H * HD;
  const int64_t ldq = a.ldq;
  const bf16raw* const Kh = a.qkv_hi + row0 * ldq + D + h * HD;
  const bf16raw* const Kl = a.qkv_lo + row0 * ldq + D + h * HD;
  const bf16raw* const Vh = Kh + D;
  const bf16raw* const Vl = Kl + D;
  const int T = (int)((Lg + AT_KT - 1) / AT_KT);

  TileDma dma;
  dma.init(wave, lane, UN);
  TrAddr tra;
  tra.init(lane);
  const unsigned smem_l = lds_addr(smem);
  dma.issue(smem, wave, 0, 0, Lg, Kh, Kl, ldq, Vh, Vl, ldq);
  if (T > 1) dma.issue(smem, wave, 1, AT_KT, Lg, Kh, Kl, ldq, Vh, Vl, ldq);
  zero_pad_units<HD>(smem, 2, tid);

  const int64_t q = (int64_t)qt * AT_QB + wave * 32 + j;
  const bool qok = q < Lg;
  bf16x8 qh[KS], ql[KS];
  {
    const int64_t qo = (row0 + (qok ? q : 0)) * ldq + h * HD;
    load_row_frags<HD>(a.qkv_hi + qo, a.qkv_lo + qo, qok, half, qh, ql);
  }
  uint32_t hbase = 0;
  if (DROP) {
    const uint64_t grow = (uint64_t)(row0 + (a.rng_rowoff ? a.rng_rowoff[g] : 0) + q);
    hbase = attn_row_key(rng_key(*a.seed, a.stream_id), grow * (uint64_t)H + (uint64_t)h) + (uint32_t)half * AT_GOLD;
  }
  const float c = a.scale_log2e;
  const uint32_t thr = a.drop_thr;

  f32x16 o[DT];
#pragma unroll
  for (int dt = 0; dt < DT; ++dt)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[dt][r] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;
  if (LSEIN) m_run = qok ? a.lse[(row0 + q) * H + h] : 0.f;       // (the given log-sum-exp, log2 domain, rides in m_run)
#ifdef AT_STAMP
  int st_i = 0;
#endif
  if (T > 1) at_wait_vmcnt<AT_PPW>(); else at_wait_vmcnt<0>();      // (the Q fragments were requested after the tiles: tile 0 has landed)
  at_barrier();
  for (int t = 0; t < T; ++t) {
    AT_ST(1);
    const unsigned char* sK = smem + (t & 1) * AT_SLOT_B;
    TrBase trb;
    trb.set(smem_l + (t & 1) * AT_SLOT_B, tra);
    const int64_t kb = (int64_t)t * AT_KT;
    const bool tail = kb + AT_KT > Lg;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      // ---- S^T[key, q] = K . Q^T for keys kb + 32 u ...
      f32x16 s;
#pragma unroll
      for (int r = 0; r < 16; ++r) s[r] = 0.f;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const bf16x8 kh = frag_rows(sK, 32 * u + j, 2 * ks + half), kl = frag_rows(sK + AT_PLANE_B, 32 * u + j, 2 * ks + half);
        MFMA3(s, kh, kl, qh[ks], ql[ks]);
      }
      if (tail) {                                // ragged tail: keys past the bag get probability 0
        const int lim = (int)(Lg - kb);
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if (32 * u + ACC_ROW(r, half) >= lim) s[r] = -INFINITY;
      }
      // ---- statistics, probabilities, dropout, split
      if constexpr (LSEIN) {
        const float nl = -m_run;
#pragma unroll
        for (int r = 0; r < 16; r += 4) {
          float e0 = fmaf(s[r], c, nl), e1 = fmaf(s[r + 1], c, nl), e2 = fmaf(s[r + 2], c, nl), e3 = fmaf(s[r + 3], c, nl);
          hw_exp2x4(e0, e1, e2, e3);
          s[r] = e0; s[r + 1] = e1; s[r + 2] = e2; s[r + 3] = e3;
        }
      } else {
        float mx = m_run;
#pragma unroll
        for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[r]);
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        if (__any(mx != m_run)) {                // rescale only when some query of this wave saw a new maximum (rare after the first tiles)
          const float alpha = hw_exp2((m_run - mx) * c);
          m_run = mx;
          l_run *= alpha;
#pragma unroll
          for (int dt = 0; dt < DT; ++dt)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[dt][r] *= alpha;
        }
        const float nmc = -mx * c;
        float psum = 0.f;
#pragma unroll
        for (int r = 0; r < 16; r += 4) {
          float e0 = fmaf(s[r], c, nmc), e1 = fmaf(s[r + 1], c, nmc), e2 = fmaf(s[r + 2], c, nmc), e3 = fmaf(s[r + 3], c, nmc);
          hw_exp2x4(e0, e1, e2, e3);
          psum += (e0 + e1) + (e2 + e3);
          s[r] = e0; s[r + 1] = e1; s[r + 2] = e2; s[r + 3] = e3;
        }
        l_run += psum;
      }
      if (DROP) {                                // register 4 rg + e <-> key kb + 32 u + 8 rg + 4 half + e: one hash per register group
        const uint32_t hb = hbase + (uint32_t)(t * (AT_KT / 4) + 8 * u) * AT_GOLD;
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
          const uint32_t hsh = attn_mix(hb + (uint32_t)(2 * rg) * AT_GOLD);
          if ((hsh & 0xffu) < thr) s[4 * rg] = 0.f;
          if (((hsh >> 8) & 0xffu) < thr) s[4 * rg + 1] = 0.f;
          if (((hsh >> 16) & 0xffu) < thr) s[4 * rg + 2] = 0.f;
          if ((hsh >> 24) < thr) s[4 * rg + 3] = 0.f;
        }
      }
      // ---- O^T[d, q] += V^T[d, key] . P^T[key, q]
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = s[8 * s2 + e];
        bf16x8 ph, pl;
        split8(v, ph, pl);
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
          const bf16x8 vh = frag_tr_pre(trb, 2 * AT_PLANE_B, 32 * u + 16 * s2, dt);
          const bf16x8 vl = frag_tr_pre(trb, 3 * AT_PLANE_B, 32 * u + 16 * s2, dt);
          MFMA3(o[dt], vh, vl, ph, pl);
        }
      }
      AT_ST(3 + u);
    }
    if (t + 1 < T) {
      at_wait_vmcnt<0>();                        // this wave's pieces of tile t + 1 (issued one trip ago)
      AT_ST(5);
      at_barrier();                              // every wave is done with tile t's slot, every piece of tile t + 1 is visible
      if (t + 2 < T) dma.issue(smem, wave, t & 1, (int64_t)(t + 2) * AT_KT, Lg, Kh, Kl, ldq, Vh, Vl, ldq);
      AT_ST(2);
    }
  }
  const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
  // The epilogue's lane coordinates are derived AGAIN here, from a thread id the optimiser cannot connect with the prologue's: under
  // the 128-VGPR bound it otherwise carries q / the output pointers through the tile loop in scratch memory (5 spilled registers).
  int tid2 = threadIdx.x;
  asm volatile("" : "+v"(tid2));
  const int j2 = tid2 & 31, half2 = (tid2 >> 5) & 1;
  const int64_t q2 = (int64_t)qt * AT_QB + (tid2 >> 6) * 32 + j2;
  if (q2 < Lg) {
    const float inv = LSEIN ? (DROP ? a.inv_keep : 1.f) : (DROP ? a.inv_keep : 1.f) * hw_rcp(l_tot);
    float* const orow = a.out + (row0 + q2) * D + h * HD;
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) {
        const int d = 32 * dt + 8 * rg + 4 * half2;
        if (d < HD)
          *reinterpret_cast<float4*>(orow + d) =
              make_float4(o[dt][4 * rg] * inv, o[dt][4 * rg + 1] * inv, o[dt][4 * rg + 2] * inv, o[dt][4 * rg + 3] * inv);
      }
    if (!LSEIN && half2 == 0) a.lse[(row0 + q2) * H + h] = m_run * c + hw_log2(l_tot);
  }
}

// =====================================================================================
// backward, queries stationary: dQ = scale * dS K
// =====================================================================================
// One workgroup (8 waves) per CU; the K / V tiles ride a ring of 3 slots: tile t + 2 is requested at the top of trip t (its slot's
// tile t - 1 was released by the barrier that closed trip t - 1), tile t + 1 is awaited at the bottom of trip t, one barrier per tile.
// A 32-key half-tile at a time through scores -> dS -> dQ: only one pair of 32x32 accumulators is live.
template <int HD, bool DROP>
__global__ __launch_bounds__(512, 2) void attn_bwd_dq_kernel(AttnArgs a) {
  constexpr int KS = HD / 16, DT = (HD + 31) / 32, UN = HD / 8;
  __shared__ __attribute__((aligned(16))) unsigned char smem[3 * AT_SLOT_B];
  const int tid = threadIdx.x, lane = tid & 63, j = lane & 31, half = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int H = a.H;
  const int h = blockIdx.x % H;
  const int rest = blockIdx.x / H;
  const int qt = rest % a.ntile, g = rest / a.ntile;
  const int64_t row0 = a.ptr ? a.ptr[g] : 0;
  const int64_t Lg = a.ptr ? a.ptr[g + 1] - row0 : a.Ltot;
  if ((int64_t)qt * AT_QB >= Lg) return;
  const int D = H * HD;
  const int64_t ldq = a.ldq;
  const bf16raw* const Kh = a.qkv_hi + row0 * ldq + D + h * HD;
  const bf16raw* const Kl = a.qkv_lo + row0 * ldq + D + h * HD;
  const bf16raw* const Vh = Kh + D;
  const bf16raw* const Vl = Kl + D;
  const int T = (int)((Lg + AT_KT - 1) / AT_KT);

  TileDma dma;
  dma.init(wave, lane, UN);
  TrAddr tra;
  tra.init(lane);
  const unsigned smem_l = lds_addr(smem);
  dma.issue(smem, wave, 0, 0, Lg, Kh, Kl, ldq, Vh, Vl, ldq);
  if (T > 1) dma.issue(smem, wave, 1, AT_KT, Lg, Kh, Kl, ldq, Vh, Vl, ldq);
  zero_pad_units<HD>(smem, 3, tid);

  const int64_t q = (int64_t)qt * AT_QB + wave * 32 + j;
  const bool qok = q < Lg;
  bf16x8 qh[KS], ql[KS], gh[KS], gl[KS];
  {
    const int64_t qo = (row0 + (qok ? q : 0)) * ldq + h * HD, go = (row0 + (qok ? q : 0)) * D + h * HD;
    load_row_frags<HD>(a.qkv_hi + qo, a.qkv_lo + qo, qok, half, qh, ql);
    load_row_frags<HD>(a.do_hi + go, a.do_lo + go, qok, half, gh, gl);
  }
  const float lse_q = qok ? a.lse[(row0 + q) * H + h] : 0.f;
  const float d_q = qok ? a.dsum[(row0 + q) * H + h] : 0.f;
  uint32_t hbase = 0;
  if (DROP) {
    const uint64_t grow = (uint64_t)(row0 + (a.rng_rowoff ? a.rng_rowoff[g] : 0) + q);
    hbase = attn_row_key(rng_key(*a.seed, a.stream_id), grow * (uint64_t)H + (uint64_t)h) + (uint32_t)half * AT_GOLD;
  }
  const float ik = DROP ? a.inv_keep : 1.f;
  const float c = a.scale_log2e;
  const uint32_t thr = a.drop_thr;

  f32x16 dq[DT];
#pragma unroll
  for (int dt = 0; dt < DT; ++dt)
#pragma unroll
    for (int r = 0; r < 16; ++r) dq[dt][r] = 0.f;

  at_wait_vmcnt<0>();
  at_barrier();                                  // tiles 0 and 1 have landed
  for (int t = 0; t < T; ++t) {
    if (t + 2 < T) dma.issue(smem, wave, (t + 2) % 3, (int64_t)(t + 2) * AT_KT, Lg, Kh, Kl, ldq, Vh, Vl, ldq);
    const unsigned char* sK = smem + (t % 3) * AT_SLOT_B;
    const unsigned char* sV = sK + 2 * AT_PLANE_B;
    TrBase trb;
    trb.set(smem_l + (t % 3) * AT_SLOT_B, tra);
    const int64_t kb = (int64_t)t * AT_KT;
    const bool tail = kb + AT_KT > Lg;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      f32x16 st, dpt;
#pragma unroll
      for (int r = 0; r < 16; ++r) { st[r] = 0.f; dpt[r] = 0.f; }
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const bf16x8 kh = frag_rows(sK, 32 * u + j, 2 * ks + half), kl = frag_rows(sK + AT_PLANE_B, 32 * u + j, 2 * ks + half);
        const bf16x8 vh = frag_rows(sV, 32 * u + j, 2 * ks + half), vl = frag_rows(sV + AT_PLANE_B, 32 * u + j, 2 * ks + half);
        st = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kl, qh[ks], st, 0, 0, 0);          // S^T[key, q]   (unscaled)
        dpt = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vl, gh[ks], dpt, 0, 0, 0);        // dPd^T[key, q] = V . dO^T
        st = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kh, ql[ks], st, 0, 0, 0);
        dpt = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vh, gl[ks], dpt, 0, 0, 0);
        st = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kh, qh[ks], st, 0, 0, 0);
        dpt = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vh, gh[ks], dpt, 0, 0, 0);
      }
      if (DROP) {                                // register 4 rg + e <-> key kb + 32 u + 8 rg + 4 half + e: one hash per register group
        const uint32_t hb = hbase + (uint32_t)(t * (AT_KT / 4) + 8 * u) * AT_GOLD;
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
          const uint32_t hsh = attn_mix(hb + (uint32_t)(2 * rg) * AT_GOLD);
          if ((hsh & 0xffu) < thr) dpt[4 * rg] = 0.f;
          if (((hsh >> 8) & 0xffu) < thr) dpt[4 * rg + 1] = 0.f;
          if (((hsh >> 16) & 0xffu) < thr) dpt[4 * rg + 2] = 0.f;
          if ((hsh >> 24) < thr) dpt[4 * rg + 3] = 0.f;
        }
      }
#pragma unroll
      for (int r = 0; r < 16; r += 4) {
        float e0 = fmaf(st[r], c, -lse_q), e1 = fmaf(st[r + 1], c, -lse_q), e2 = fmaf(st[r + 2], c, -lse_q), e3 = fmaf(st[r + 3], c, -lse_q);
        hw_exp2x4(e0, e1, e2, e3);
        st[r] = e0 * fmaf(dpt[r], ik, -d_q); st[r + 1] = e1 * fmaf(dpt[r + 1], ik, -d_q);
        st[r + 2] = e2 * fmaf(dpt[r + 2], ik, -d_q); st[r + 3] = e3 * fmaf(dpt[r + 3], ik, -d_q);     // dS^T
      }
      if (tail) {
        const int lim = (int)(Lg - kb);
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if (32 * u + ACC_ROW(r, half) >= lim) st[r] = 0.f;
      }
      // dQ^T[d, q] += K^T[d, key] . dS^T[key, q]
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = st[8 * s2 + e];
        bf16x8 dh, dl;
        split8(v, dh, dl);
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
          const bf16x8 kh = frag_tr_pre(trb, 0, 32 * u + 16 * s2, dt);
          const bf16x8 kl = frag_tr_pre(trb, AT_PLANE_B, 32 * u + 16 * s2, dt);
          MFMA3(dq[dt], kh, kl, dh, dl);
        }
      }
    }
    if (t + 1 < T) {
      if (t + 2 < T) at_wait_vmcnt<AT_PPW>(); else at_wait_vmcnt<0>();      // tile t + 1 (only tile t + 2's pieces may still fly)
      at_barrier();
    }
  }
  int tid2 = threadIdx.x;                        // (lane coordinates derived again for the epilogue: see the forward)
  asm volatile("" : "+v"(tid2));
  const int half2 = (tid2 >> 5) & 1;
  const int64_t q2 = (int64_t)qt * AT_QB + (tid2 >> 6) * 32 + (tid2 & 31);
  if (q2 < Lg) {
    float* const drow = a.dqkv + (row0 + q2) * ldq + h * HD;
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) {
        const int d = 32 * dt + 8 * rg + 4 * half2;
        if (d < HD)
          *reinterpret_cast<float4*>(drow + d) = make_float4(dq[dt][4 * rg] * a.scale, dq[dt][4 * rg + 1] * a.scale,
                                                             dq[dt][4 * rg + 2] * a.scale, dq[dt][4 * rg + 3] * a.scale);
      }
  }
}

// =====================================================================================
// backward, keys stationary: dV = Pd^T dO, dK = scale * dS^T Q
// =====================================================================================
// Same ring; the streamed tiles are [Q hi | Q lo | dO hi | dO lo] of 64 queries, consumed in two 32-query halves. The side data of
// a tile's queries (lse, D, dropout row key) sit in a double-buffered LDS array: lanes 0-7 of wave w load those of query 8 w + lane
// of tile t + 1 at the top of trip t and store them at its bottom, in front of the barrier.
template <int HD, bool DROP>
__global__ __launch_bounds__(512, 2) void attn_bwd_dkv_kernel(AttnArgs a) {
  constexpr int KS = HD / 16, DT = (HD + 31) / 32, UN = HD / 8;
  __shared__ __attribute__((aligned(16))) unsigned char smem[3 * AT_SLOT_B + 2 * 3 * AT_KT * 4];
  float* const sAux = reinterpret_cast<float*>(smem + 3 * AT_SLOT_B);     // per parity: lse[64] | D[64] | rowkey[64]
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
  const bf16raw* const Qh = a.qkv_hi + row0 * ldq + h * HD;
  const bf16raw* const Ql = a.qkv_lo + row0 * ldq + h * HD;
  const bf16raw* const Gh = a.do_hi + row0 * D + h * HD;
  const bf16raw* const Gl = a.do_lo + row0 * D + h * HD;
  const int T = (int)((Lg + AT_KT - 1) / AT_KT);

  const uint64_t key64 = DROP ? rng_key(*a.seed, a.stream_id) : 0;
  const int64_t rowoff = a.rng_rowoff ? a.rng_rowoff[g] : 0;
  float ax_l = 0.f, ax_d = 0.f;
  auto aux_load = [&](int tile) {
    if (lane < 8) {
      const int64_t qq = (int64_t)tile * AT_KT + wave * 8 + lane;
      const bool ok = qq < Lg;
      ax_l = ok ? a.lse[(row0 + qq) * H + h] : 0.f;
      ax_d = ok ? a.dsum[(row0 + qq) * H + h] : 0.f;
    }
  };
  auto aux_store = [&](int tile) {
    if (lane < 8) {
      float* p = sAux + (tile & 1) * 3 * AT_KT + wave * 8 + lane;
      p[0] = ax_l; p[AT_KT] = ax_d;
      if (DROP) {
        const int64_t qq = (int64_t)tile * AT_KT + wave * 8 + lane;
        reinterpret_cast<uint32_t*>(p)[2 * AT_KT] = attn_row_key(key64, (uint64_t)(row0 + rowoff + qq) * (uint64_t)H + (uint64_t)h);
      }
    }
  };
  TileDma dma;
  dma.init(wave, lane, UN);
  TrAddr tra;
  tra.init(lane);
  const unsigned smem_l = lds_addr(smem);
  aux_load(0);
  dma.issue(smem, wave, 0, 0, Lg, Qh, Ql, ldq, Gh, Gl, D);
  if (T > 1) dma.issue(smem, wave, 1, AT_KT, Lg, Qh, Ql, ldq, Gh, Gl, D);
  zero_pad_units<HD>(smem, 3, tid);

  const int64_t key = (int64_t)kt * AT_QB + wave * 32 + j;
  const bool kok = key < Lg;
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

  f32x16 dk[DT], dv[DT];
#pragma unroll
  for (int dt = 0; dt < DT; ++dt)
#pragma unroll
    for (int r = 0; r < 16; ++r) { dk[dt][r] = 0.f; dv[dt][r] = 0.f; }

  at_wait_vmcnt<0>();
  aux_store(0);
  at_barrier();                                  // tiles 0 and 1 and tile 0's side data are in LDS
  for (int t = 0; t < T; ++t) {
    if (t + 1 < T) aux_load(t + 1);
    if (t + 2 < T) dma.issue(smem, wave, (t + 2) % 3, (int64_t)(t + 2) * AT_KT, Lg, Qh, Ql, ldq, Gh, Gl, D);
    const unsigned char* sQ = smem + (t % 3) * AT_SLOT_B;
    const unsigned char* sG = sQ + 2 * AT_PLANE_B;
    TrBase trb;
    trb.set(smem_l + (t % 3) * AT_SLOT_B, tra);
    const float* axt = sAux + (t & 1) * 3 * AT_KT;
    const int64_t qb = (int64_t)t * AT_KT;
    const bool tail = qb + AT_KT > Lg;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      f32x16 s, dp;
#pragma unroll
      for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const bf16x8 ah = frag_rows(sQ, 32 * u + j, 2 * ks + half), al = frag_rows(sQ + AT_PLANE_B, 32 * u + j, 2 * ks + half);
        const bf16x8 bh = frag_rows(sG, 32 * u + j, 2 * ks + half), bl = frag_rows(sG + AT_PLANE_B, 32 * u + j, 2 * ks + half);
        s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, kh[ks], s, 0, 0, 0);            // S[q, key]   (unscaled)
        dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bl, vh[ks], dp, 0, 0, 0);          // dPd[q, key] = dO . V^T
        s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, kl[ks], s, 0, 0, 0);
        dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bh, vl[ks], dp, 0, 0, 0);
        s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, kh[ks], s, 0, 0, 0);
        dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bh, vh[ks], dp, 0, 0, 0);
      }
      const float* ax = axt + 32 * u;
      // rows of this half are queries: 4 consecutive ones per register group
      uint32_t hq[4];
      if (DROP) {                                // quad lane e hashes the queries (e) + 8 rg + 4 half, rg = 0..3, for the quad's key group
        const uint32_t* rkp = reinterpret_cast<const uint32_t*>(ax) + 2 * AT_KT;
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) hq[rg] = attn_mix(rkp[8 * rg + 4 * half + (lane & 3)] + kgold);
      }
      float pd[16];
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) {
        const int qo = 8 * rg + 4 * half;
        const float4 l4 = *reinterpret_cast<const float4*>(ax + qo);
        const float4 d4 = *reinterpret_cast<const float4*>(ax + AT_KT + qo);
        const float dvv[4] = {d4.x, d4.y, d4.z, d4.w};
        uint32_t hx[4] = {0u, 0u, 0u, 0u};
        if (DROP) {                              // hash of query qo + e lives in quad lane e
          hx[0] = (uint32_t)__builtin_amdgcn_mov_dpp((int)hq[rg], 0x00, 0xf, 0xf, false);
          hx[1] = (uint32_t)__builtin_amdgcn_mov_dpp((int)hq[rg], 0x55, 0xf, 0xf, false);
          hx[2] = (uint32_t)__builtin_amdgcn_mov_dpp((int)hq[rg], 0xaa, 0xf, 0xf, false);
          hx[3] = (uint32_t)__builtin_amdgcn_mov_dpp((int)hq[rg], 0xff, 0xf, 0xf, false);
        }
        float e0 = fmaf(s[4 * rg], c, -l4.x), e1 = fmaf(s[4 * rg + 1], c, -l4.y), e2 = fmaf(s[4 * rg + 2], c, -l4.z), e3 = fmaf(s[4 * rg + 3], c, -l4.w);
        hw_exp2x4(e0, e1, e2, e3);
        if (tail) {                              // queries past the bag (their lse slot holds 0): one uniform branch per group
          const int lim = (int)(Lg - qb);
          if (32 * u + qo + 0 >= lim) e0 = 0.f;
          if (32 * u + qo + 1 >= lim) e1 = 0.f;
          if (32 * u + qo + 2 >= lim) e2 = 0.f;
          if (32 * u + qo + 3 >= lim) e3 = 0.f;
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
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        float v[8], w[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) { v[e] = pd[8 * s2 + e]; w[e] = s[8 * s2 + e]; }
        bf16x8 ph, pl, dh, dl;
        split8(v, ph, pl);
        split8(w, dh, dl);
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
          const bf16x8 gh = frag_tr_pre(trb, 2 * AT_PLANE_B, 32 * u + 16 * s2, dt);
          const bf16x8 gl = frag_tr_pre(trb, 3 * AT_PLANE_B, 32 * u + 16 * s2, dt);
          const bf16x8 qh = frag_tr_pre(trb, 0, 32 * u + 16 * s2, dt);
          const bf16x8 ql = frag_tr_pre(trb, AT_PLANE_B, 32 * u + 16 * s2, dt);
          dv[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(gl, ph, dv[dt], 0, 0, 0);     // dV^T[d, key] += dO^T[d, q] . Pd[q, key]
          dk[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ql, dh, dk[dt], 0, 0, 0);     // dK^T[d, key] += Q^T[d, q] . dS[q, key]
          dv[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(gh, pl, dv[dt], 0, 0, 0);
          dk[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qh, dl, dk[dt], 0, 0, 0);
          dv[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(gh, ph, dv[dt], 0, 0, 0);
          dk[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qh, dh, dk[dt], 0, 0, 0);
        }
      }
    }
    if (t + 1 < T) {
      if (t + 2 < T) at_wait_vmcnt<AT_PPW>(); else at_wait_vmcnt<0>();      // tile t + 1 and its side data (requested before tile t + 2)
      aux_store(t + 1);
      at_barrier();
    }
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

// D[row, h] = sum_d dO[row, h*HD + d] * O[row, h*HD + d]   (the softmax backward's row constant; holds with dropout, since
// sum_j dP_j P_j = sum_j dPd_j Pd_j = dO . O), and the bf16x3 operand planes of dO for the two gradient kernels, in one pass
// One thread per float4: loads and plane stores are contiguous across the wave (the one-thread-per-(row, head) form of round 3
// walked 192-byte strides: 58 us for 150 MB); the HD / 4 partial dot products of a (row, head) meet in LDS.
template <int HD>
__global__ __launch_bounds__(HD == 48 ? 192 : 256) void attn_bwd_prep_kernel(const float* __restrict__ dout, const float* __restrict__ out,
                                                                           int64_t n /* rows * H */, float* __restrict__ dsum,
                                                                           bf16raw* __restrict__ g_hi, bf16raw* __restrict__ g_lo) {
  constexpr int BT = HD == 48 ? 192 : 256, G = HD / 4, NG = BT / G;
  __shared__ float part[BT];
  const int tid = threadIdx.x;
  const int64_t idx4 = (int64_t)blockIdx.x * BT + tid, n4 = n * G;
  float s = 0.f;
  if (idx4 < n4) {
    const float4 x = reinterpret_cast<const float4*>(dout)[idx4], y = reinterpret_cast<const float4*>(out)[idx4];
    s = (x.x * y.x + x.y * y.y) + (x.z * y.z + x.w * y.w);
    uint2 hh, ll;
    split4(x, hh, ll);
    reinterpret_cast<uint2*>(g_hi)[idx4] = hh;
    reinterpret_cast<uint2*>(g_lo)[idx4] = ll;
  }
  part[tid] = s;
  __syncthreads();
  if (tid < NG) {
    const int64_t gi = (int64_t)blockIdx.x * NG + tid;
    if (gi < n) {
      float acc = 0.f;
#pragma unroll
      for (int i = 0; i < G; ++i) acc += part[tid * G + i];
      dsum[gi] = acc;
    }
  }
}

#ifdef AT_STAMP
// Dev probe: what the occupancy API says about co-resident workgroups per CU for the head_dim-48 kernels
extern "C" int advmil_debug_attn_occupancy(int which) {
  int n = -1;
  if (which == 0) hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, attn_fwd_kernel<48, true>, 512, 0);
  else if (which == 1) hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, attn_bwd_dq_kernel<48, true>, 512, 0);
  else hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, attn_bwd_dkv_kernel<48, true>, 512, 0);
  return n;
}
#endif

// =====================================================================================
// C ABI
// =====================================================================================

extern "C" int advmil_mha_fwd(const void* qkv_hi, const void* qkv_lo, int64_t Ltot, int nhead, int head_dim, int nseg,
                              const int64_t* ptr, int64_t max_len, float drop_p, const uint64_t* seed, uint64_t stream_id,
                              const int64_t* rng_rowoff, float* out, float* lse, advmil_stream_t stream_) {
  AttnArgs a;
  const int rc = attn_args(a, qkv_hi, qkv_lo, Ltot, nhead, head_dim, nseg, ptr, max_len, drop_p, seed, stream_id, rng_rowoff);
  if (rc) return rc;
  if (!out || !lse || ((uintptr_t)out & 15)) return ADVMIL_EINVAL;
  a.out = out; a.lse = lse;
  const dim3 grid((unsigned)(a.ntile * nseg * nhead));
  AT_DISPATCH(attn_fwd_kernel, grid, (hipStream_t)stream_, a, head_dim);
  ADVMIL_LAUNCH_CHECK();
  return ADVMIL_OK;
}

// the same forward with the log-sum-exp GIVEN (dropout on: the train-mode pass behind an eval-mode pass over the same q | k | v)
extern "C" int advmil_mha_fwd_lse(const void* qkv_hi, const void* qkv_lo, int64_t Ltot, int nhead, int head_dim, int nseg,
                                  const int64_t* ptr, int64_t max_len, float drop_p, const uint64_t* seed, uint64_t stream_id,
                                  const int64_t* rng_rowoff, float* out, const float* lse, advmil_stream_t stream_) {
  AttnArgs a;
  const int rc = attn_args(a, qkv_hi, qkv_lo, Ltot, nhead, head_dim, nseg, ptr, max_len, drop_p, seed, stream_id, rng_rowoff);
  if (rc) return rc;
  if (!out || !lse || ((uintptr_t)out & 15) || !a.seed) return ADVMIL_EINVAL;       // (built for the dropout pass only)
  a.out = out; a.lse = const_cast<float*>(lse);
  const dim3 grid((unsigned)(a.ntile * nseg * nhead));
  hipStream_t stream = (hipStream_t)stream_;
  switch (head_dim) {
    case 16: hipLaunchKernelGGL((attn_fwd_kernel<16, true, true>), grid, dim3(512), 0, stream, a); break;
    case 32: hipLaunchKernelGGL((attn_fwd_kernel<32, true, true>), grid, dim3(512), 0, stream, a); break;
    case 48: hipLaunchKernelGGL((attn_fwd_kernel<48, true, true>), grid, dim3(512), 0, stream, a); break;
    default: hipLaunchKernelGGL((attn_fwd_kernel<64, true, true>), grid, dim3(512), 0, stream, a); break;
  }
  ADVMIL_LAUNCH_CHECK();
  return ADVMIL_OK;
}

// D [Ltot, nhead] fp32, then the two planes of dO [Ltot, nhead*head_dim] bf16 (each part 16-byte aligned)
extern "C" size_t advmil_mha_bwd_workspace_bytes(int64_t Ltot, int nhead, int head_dim) {
  return attn_ws_dsum_bytes(Ltot, nhead) + 2 * attn_ws_plane_bytes(Ltot, nhead, head_dim);
}

int attn_launch_bwd_prep(const float* dout, const float* out, int64_t Ltot, int nhead, int head_dim, float* dsum, bf16raw* g_hi,
                         bf16raw* g_lo, hipStream_t stream) {
  const int64_t n = Ltot * nhead;
  const int pbt = head_dim == 48 ? 192 : 256;                       // threads per block; each covers pbt / (head_dim / 4) (row, head) pairs
  const int64_t png = pbt / (head_dim / 4);
  const dim3 pg((unsigned)((n + png - 1) / png));
  switch (head_dim) {
    case 16: hipLaunchKernelGGL((attn_bwd_prep_kernel<16>), pg, dim3(pbt), 0, stream, dout, out, n, dsum, g_hi, g_lo); break;
    case 32: hipLaunchKernelGGL((attn_bwd_prep_kernel<32>), pg, dim3(pbt), 0, stream, dout, out, n, dsum, g_hi, g_lo); break;
    case 48: hipLaunchKernelGGL((attn_bwd_prep_kernel<48>), pg, dim3(pbt), 0, stream, dout, out, n, dsum, g_hi, g_lo); break;
    default: hipLaunchKernelGGL((attn_bwd_prep_kernel<64>), pg, dim3(pbt), 0, stream, dout, out, n, dsum, g_hi, g_lo); break;
  }
  ADVMIL_LAUNCH_CHECK();
  return ADVMIL_OK;
}

extern "C" int advmil_mha_bwd(const void* qkv_hi, const void* qkv_lo, const float* out, const float* dout, const float* lse,
                              int64_t Ltot, int nhead, int head_dim, int nseg, const int64_t* ptr, int64_t max_len, float drop_p,
                              const uint64_t* seed, uint64_t stream_id, const int64_t* rng_rowoff, float* dqkv, void* ws,
                              size_t ws_bytes, advmil_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  AttnArgs a;
  const int rc = attn_args(a, qkv_hi, qkv_lo, Ltot, nhead, head_dim, nseg, ptr, max_len, drop_p, seed, stream_id, rng_rowoff);
  if (rc) return rc;
  if (!out || !dout || !lse || !dqkv || !ws) return ADVMIL_EINVAL;
  if (((uintptr_t)out & 15) || ((uintptr_t)dout & 15) || ((uintptr_t)dqkv & 15) || ((uintptr_t)ws & 15)) return ADVMIL_EINVAL;
  if (ws_bytes < advmil_mha_bwd_workspace_bytes(Ltot, nhead, head_dim)) return ADVMIL_EWORKSPACE;
  float* dsum = (float*)ws;
  bf16raw* g_hi = (bf16raw*)((char*)ws + attn_ws_dsum_bytes(Ltot, nhead));
  bf16raw* g_lo = (bf16raw*)((char*)g_hi + attn_ws_plane_bytes(Ltot, nhead, head_dim));
  const int prc = attn_launch_bwd_prep(dout, out, Ltot, nhead, head_dim, dsum, g_hi, g_lo, stream);
  if (prc) return prc;
  a.lse = const_cast<float*>(lse); a.do_hi = g_hi; a.do_lo = g_lo; a.dsum = dsum; a.dqkv = dqkv;
  const dim3 grid((unsigned)(a.ntile * nseg * nhead));
  AT_DISPATCH(attn_bwd_dq_kernel, grid, stream, a, head_dim);
  ADVMIL_LAUNCH_CHECK();
  AT_DISPATCH(attn_bwd_dkv_kernel, grid, stream, a, head_dim);
  ADVMIL_LAUNCH_CHECK();
  return ADVMIL_OK;
}
