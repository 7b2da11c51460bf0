/* C ABI of libadvmil_hip.so — the MI355X (gfx950) kernels underneath the AdvMIL plugin surface.
 *
 * The reference (liupei101/AdvMIL @ v1) is pure Python/PyTorch and has no FFI: its "operator
 * interface" for the generator+discriminator training path is the set of nn.Module forwards in
 * model/backbone.py, model/backbone_utils.py, model/GANSurv.py, model/model_utils.py and the loss
 * functions in loss/utils.py. Each entry point below names the reference code it replaces
 * (file:line into /root/reference). The Python host side (advmil_amd/) binds these with ctypes
 * and mirrors the reference's class names / ctor + forward signatures / state_dict keys.
 *
 * Conventions
 *  - all pointers are DEVICE pointers borrowed for the duration of the enqueue; row-major,
 *    contiguous unless a leading dimension is given; 16-byte aligned; fp32 unless noted.
 *  - every call is asynchronous on `stream` (a hipStream_t), never allocates, never syncs; it is
 *    safe under hipStreamBeginCapture (HIP graphs).
 *  - return value: 0 = ok, <0 = ADVMIL_E* (bad argument), >0 = hipError_t of the launch.
 *  - randomness (dropout, generator noise) is counter based: u = top 24 bits of hash32(key(seed, stream_id),
 *    element_index), key = splitmix64 mix of seed and stream id, hash32 = a two-round 32-bit multiply-xorshift
 *    mixer (csrc/common.h::rng_hash32); `seed` is a DEVICE pointer to a uint64 so a captured graph can be
 *    replayed with a fresh seed; `stream_id` distinguishes call sites. NULL seed or p == 0
 *    disables dropout. advmil_amd/synth.py restates the generator on the host.
 */
#ifndef ADVMIL_HIP_H
#define ADVMIL_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* advmil_stream_t; /* hipStream_t */

enum { ADVMIL_ACT_NONE = 0, ADVMIL_ACT_RELU = 1, ADVMIL_ACT_TANH = 2, ADVMIL_ACT_SIGMOID = 3 };

int advmil_version(void);

/* ------------------------------------------------------------------------------------------
 * Dense contraction engine (fp32 MFMA v_mfma_f32_32x32x2_f32, exact fp32 accumulate).
 *   C[M,N] = epilogue( alpha * op(A)[M,K] . op(B)[K,N] )
 *   a_kc != 0 : A is [M,K] row-major (k contiguous, lda = row pitch); else A is [K,M] (m contiguous)
 *   b_kc != 0 : B is [N,K] row-major (a torch Linear weight);         else B is [K,N] (n contiguous)
 * replaces: nn.Linear / Conv2d(1x1) forwards (model/backbone.py:69,72-75,98,102;
 *   model/backbone_utils.py:16-17,35-44,150; model/model_utils.py:157-176) and the
 *   addmm/mm calls autograd issues for their backward.
 * epilogue, applied in this order per element (row m, col n):
 *   v = alpha*acc (+ bias[n]) (+ rowv[m]*colv[rowseg[m]*N + n]); v = act(v)  [act0 for n < act_split, else act1];
 *   v *= dropout(m*N+n); v *= (maskref[m*ldmask+n] > 0 ? mask_scale : 0) if maskref;
 *   C = v (+ C if accumulate).
 * splits > 1 partitions K across workgroups (needed when M*N is small and K is the bag length);
 * partials go to `ws` (advmil_gemm_f32_workspace_bytes) and a second launch reduces them.
 * Requirements: contiguous dims and leading dims multiples of 4 floats. */
typedef struct {
  const float* bias;
  int act0, act1, act_split;
  float drop_p;
  const uint64_t* seed;
  uint64_t stream_id;
  const float* rowv;
  const float* colv;
  const int32_t* rowseg; /* NULL, or the segment (bag) of each row: the rank-1 term reads colv[rowseg[m]*N + n] */
  const float* maskref;
  int ldmask;
  float mask_scale;
  int accumulate;    /* C += v. With splits > 1 (and advmil_merge_partials): while the launch stream is in DEFERRAL (advmil_defer_sums below -- opt-in
                      * per stream, never on by default) the `C += partials` fold is queued until the flush, so C must not be read or written by
                      * another launch before advmil_flush_sums / advmil_defer_sums(stream, 0); outside deferral the fold is issued at once. */
  float alpha;
  /* bf16x3 operand planes (all optional, ignored in exact mode). An fp32 matrix x can be accompanied by two bf16 matrices
   * hi = bf16(x), lo = bf16(x - hi) of the SAME shape and leading dimension (advmil_split_planes; the Adam kernel emits them
   * for the weights; c_hi/c_lo below for activations). When both planes of an operand are given (16-byte aligned, leading
   * dimension and contiguous extent multiples of 8) the engine stages them straight into LDS instead of re-splitting the fp32
   * values in every workgroup that re-reads them -- results are bit-identical to the on-the-fly split.
   * SINGLE-plane operand: a_hi (or b_hi) given with a_lo (b_lo) = NULL says the operand IS a bf16 matrix (a bag stored in bf16: the
   * x_storage = "bf16" mode of the ingest; the A / B pointer then only carries the pitch and is never dereferenced). Its products
   * are formed with two MFMAs instead of three -- built for A of the NT plane-fed forms (the embedding FCs over the slab) and B of
   * the TN plane-fed forms (the weight gradients dY^T X); the generic tiles 22 / 12 / 11 / 34 / 24 stage a zero lo plane. */
  const void* a_hi;
  const void* a_lo;
  const void* b_hi;
  const void* b_lo;
  void* c_hi; /* also write the planes of the final C values (pitch ldc) for the next contraction; with C == NULL in advmil_gemm_f32_tiled
               * (one pass, no accumulate, no second layer) the planes are written INSTEAD of C: a result consumed as an operand only */
  void* c_lo;
  /* Fused gate score (Attn_Net_Gated without its [rows, 2D] activations, model/backbone_utils.py:24-28; used by the no-grad
   * generator pass of the discriminator update, model_handler.py:398-400): B's rows are the two branches INTERLEAVED (row 2j = Wa_j,
   * row 2j+1 = Wb_j; bias likewise). Instead of C the launch writes, per row and per column block of the tile grid,
   * gate_out[m*gate_np + block] = sum_j tanh(c[2j]) * sigmoid(c[2j+1]) * gate_wc[j]; the score is the sum over blocks (+ bc).
   * gate_np = number of column blocks = n-tiles * waves along N (advmil_gemm_f32_gate_blocks). C may be NULL. No dropout, splits = 1. */
  const float* gate_wc;
  float* gate_out;
  int gate_np;
  /* Optional [M] device array: row m draws its dropout as row rng_row[m] (element index rng_row[m]*N + n). Under bag-parallel a
   * rank passes the rows its bags occupy in the single-process step slab, which makes the masks independent of the world size;
   * NULL = identity. The same optional argument exists on every entry point below that draws dropout / noise. */
  const int64_t* rng_row;
  /* Two layers over the same rows in one launch (plane-fed NT form, tile 85 only): B holds both layers' weight rows stacked
   * ([N, K]: layer 1 first); columns [0, n_split) of the product are layer 1 -- bias, output C (pitch ldc), planes c_hi/c_lo --,
   * columns [n_split, N) are layer 2 -- bias2[n - n_split], output c2[m*ldc2 + n - n_split], no planes. act_split = n_split selects
   * the two activations. n_split % 32 == 0; c2 = NULL = an ordinary launch. The generator's and the discriminator's first layers
   * over the step slab read X once this way (model/backbone.py:60-66 + model/model_utils.py:130-140). */
  float* c2;
  int64_t ldc2;
  int64_t n_split;
  const float* bias2;
  /* Column sums of the FINAL values, one partial row per (row tile, wave row) of the launch: colsum[p * N + n], p < advmil_gemm_f32_colsum_rows
   * (slab-sized bf16x3 tiles, whole tiles, splits = 1, no accumulate / dropout). With rowv + maskref in the same launch (rank-1 term, then
   * the mask) this is the fused backward of the first layer behind the gated-attention pool: dpre = (dG Wab + A dpooled) * (y > 0 ?
   * mask_scale : 0) and its bias gradient, instead of a separate pass over dh and y (model/backbone.py:79-86 autograd of ReLU + Dropout). */
  float* colsum;
  /* The mask as one bit per element instead of maskref: maskbits[m * ldbits + n / 32] bit n % 32 set = keep (x mask_scale), clear = 0
   * (advmil_act_dropout_bwd's `bits` output). Whole-tile launches of the slab-sized bf16x3 tiles only; may be combined with rowv. The
   * epilogue parks a wave's words in LDS before its first store: no per-element mask load inside the store loop. */
  const uint32_t* maskbits;
  int64_t ldbits;
  /* TRAINING form of the fused gate score (tiles 85 / 86, plain launches): gate_wc / gate_out / gate_np as above, but B's rows are the two
   * branches in BLOCKS of 32 (advmil_gate_interleave with pair32 = 1: [a_0..31 | b_0..31 | a_32..63 | ...]) and C IS written -- tanh on the
   * a blocks, sigmoid on the b blocks, in that blocked column order (advmil_gate_bwd with pair32 = 1 reads it back) --, while the row's
   * sum_j (a_j ka_j)(b_j kb_j) gate_wc[j] over each column block is reduced on the way into gate_out. ka / kb = drop_p-dropout keep
   * factors given as bits: gate_bits_a / gate_bits_b [M, ldgbits] uint32, bit j % 32 of word (m, j / 32) (advmil_dropout_planes draws them).
   * Attn_Net_Gated in train mode (model/backbone_utils.py:24-29) without a pass over its stored activations. */
  const uint32_t* gate_bits_a;
  const uint32_t* gate_bits_b;
  int64_t ldgbits;
  /* The M rows of this launch's result are in that pair-block order (dWab = dG^T h with dG from advmil_gate_bwd(pair32 = 1)) while C holds
   * [branch][unit] rows: accumulating split-K launches only (splits > 1, accumulate, no other epilogue term) -- the merge un-permutes. */
  int c_rows_pair32;
} advmil_epilogue_t;

size_t advmil_gemm_f32_workspace_bytes(int64_t M, int64_t N, int splits);
int advmil_gemm_f32(int a_kc, int b_kc, int64_t M, int64_t N, int64_t K, const float* A, int64_t lda,
                    const float* B, int64_t ldb, float* C, int64_t ldc, const advmil_epilogue_t* epi,
                    int splits, void* ws, size_t ws_bytes, advmil_stream_t stream);
/* Arithmetic of the fp32 engine (process-wide): 0 = exact fp32 MFMA; 1 = split-bf16 ("bf16x3"): every fp32 operand is
 * split into hi + lo bf16 in registers and a.b is formed as ah.bh + ah.bl + al.bh on the bf16 matrix pipe with fp32
 * accumulate -- dropped terms ~2^-17 |a||b| per product, 3/16 of the matrix-pipe time. Storage stays fp32 everywhere. */
/* lo may be NULL: only hi = bf16(x) is written (the rounding a bag undergoes on its way into a bf16 slab). */
int advmil_split_planes(const float* src, int64_t n, void* hi, void* lo, advmil_stream_t stream);
/* Glue of the fused gate score (epilogue.gate_wc): Wi[2D, D] = rows a0, b0, a1, b1, ... of the attention branches' weights Wa, Wb [D, D]
 * (reference model/backbone_utils.py Attn_Net_Gated: attention_a / attention_b), its planes (Wi_hi / Wi_lo, both or neither) and the
 * interleaved bias bi[2D]; and s[n] = sum_j partial[n][j] + bc[0] over the np per-column-block partials the epilogue wrote (bc may be NULL). */
int advmil_gate_interleave(const float* Wa, const float* Wb, const float* ba, const float* bb, int D, float* Wi, void* Wi_hi, void* Wi_lo,
                           float* bi, int pair32, advmil_stream_t stream);
int advmil_gate_partial_sum(const float* partial, int np, const float* bc, int64_t N, float* s, advmil_stream_t stream);
/* column blocks a launch with this tile writes per row in gate-score mode (see advmil_epilogue_t.gate_wc) */
int advmil_gemm_f32_gate_blocks(int tile, int64_t N);
/* partial rows a launch with this tile writes into epilogue.colsum for an M x N result (0: no column-sum form for this tile / these extents) */
int64_t advmil_gemm_f32_colsum_rows(int tile, int64_t M, int64_t N);
/* out[c] (+)= sum_{b < nblk} partial[b * stride + c], c < ncols: the merge of per-workgroup partial rows (deferrable: advmil_defer_sums) */
int advmil_merge_partials(const float* partial, int nblk, int64_t stride, int64_t ncols, float* out, int accumulate, advmil_stream_t stream);
int advmil_set_gemm_mode(int mode);
int advmil_get_gemm_mode(void);
/* The library's launch plan for a shape: block tile (see below) and K split count. Host callers size the workspace
 * from `splits` (advmil_gemm_f32_workspace_bytes) and pass both to advmil_gemm_f32_tiled. */
int advmil_gemm_f32_plan(int64_t M, int64_t N, int64_t K, int* tile, int* splits);
/* Same, knowing the operand layouts (the best tile differs between the NT forward form and the NN / TN backward forms; this is what
 * advmil_gemm_f32 itself uses). advmil_gemm_f32_plan is the a_kc = b_kc = 1 case. */
int advmil_gemm_f32_plan_layout(int a_kc, int b_kc, int64_t M, int64_t N, int64_t K, int* tile, int* splits);
/* bf16x3 mode, NT form (a_kc = b_kc = 1) with BOTH operands supplied as planes (a_hi..b_lo): the tile of the plane-fed kernel that
 * stages global -> LDS by LDS-DMA as a persistent kernel (82 / 83 = 256 x 128 / 192, 8 waves, one workgroup per CU walking the
 * tiles; M % 256 == 0, K % 32 == 0 and K >= 64, N % tile width == 0, planes < 4 GB each, splits = 1), or 0 when the shape does not
 * qualify. advmil_gemm_f32_tiled(tile = 0) makes the same choice; callers of the fused gate-score mode need the tile up front for
 * advmil_gemm_f32_gate_blocks and may pass 84 (256 x 256, instantiated for that mode only) when N % 256 == 0. Results are
 * bit-identical to the generic kernel's. */
int advmil_gemm_f32_plan_planes(int a_kc, int b_kc, int64_t M, int64_t N, int64_t K, int* tile);
/* TN form (A [K, M], B [K, N], both m-contiguous) with BOTH operands as planes -- the deep-K weight gradients dW = dY^T X over a
 * step slab (torch's addmm backward of nn.Linear / the 1x1 Conv, reference model/backbone.py:60-66, backbone_utils.py:158-168):
 * tile code (91 / 92 / 93 = 128x256 / 256x128 / 256x256, LDS-DMA fed split-K kernel) and split count to pass to
 * advmil_gemm_f32_tiled, or *tile = 0 when the shape does not qualify (then the ordinary plan applies). */
int advmil_gemm_f32_plan_tn_planes(int64_t M, int64_t N, int64_t K, int* tile, int* splits);
/* Same, with an explicit block tile: tile = 10*TM + TN selects (64*TM) x (64*TN) output tiles
 * (22 = 128x128, 23 = 128x192, 13 = 64x192, 12 = 64x128, 11 = 64x64; 43 = 256x192, 42 = 256x128, 34 = 192x256, 24 = 128x256 with 512 threads,
 * bf16x3 mode only -- in exact mode they fall back to 23 / 22); 0 = the plan's choice (what advmil_gemm_f32 uses). */
int advmil_gemm_f32_tiled(int a_kc, int b_kc, int64_t M, int64_t N, int64_t K, const float* A, int64_t lda,
                          const float* B, int64_t ldb, float* C, int64_t ldc, const advmil_epilogue_t* epi,
                          int splits, int tile, void* ws, size_t ws_bytes, advmil_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Fused self-attention core of the ESAT layer (nn.MultiheadAttention inside nn.TransformerEncoderLayer,
 * model/backbone_utils.py:113-127, reached from DualTrans_HS.forward, model/backbone.py:188-196; the reference runs one
 * call per bag, model_handler.py:352, batch_size 1):  O = dropout(softmax(Q K^T / sqrt(head_dim))) V  per (bag, head).
 * Flash-style: the [L, L] scores never exist in HBM; the backward recomputes them from `lse`.
 *   qkv_hi / qkv_lo  [Ltot, 3*nhead*head_dim] bf16: the two bf16x3 operand planes (hi = bf16(x), lo = bf16(x - hi)) of the packed
 *        in-projection output (q | k | v; head h of q at columns h*head_dim ...), as advmil_split_planes or a contraction's
 *        c_hi / c_lo plane output produce them; 16-byte aligned, row pitch 3*nhead*head_dim. The kernels stream them into LDS by
 *        LDS-DMA and never see the fp32 values.
 *   out  [Ltot, nhead*head_dim];  lse [Ltot, nhead] = log2-domain log-sum-exp of the scaled scores (opaque to the caller)
 *   Rows are a slab of `nseg` bags, bag b = rows [ptr[b], ptr[b+1]) (device int64; NULL = one bag), max_len = longest bag;
 *   attention never crosses a bag. Any bag length >= 1 (ragged tails are masked). head_dim in {16, 32, 48, 64}: d_model / 8 heads
 *   of the bcb_dims load_backbone accepts (model/backbone.py:30-33; 384 -> 48 in the shipped config).
 * Dropout on the probabilities (train mode; NULL seed or p == 0 = off): one 32-bit hash per (query, 4 consecutive keys),
 *   keep(i, j) = byte (j & 3) of mix(rowkey + (j >> 2)*0x9E3779B9) >= floor(p*256),  mix(x): x ^= x >> 15; x *= 0x7feb352d; x ^= x >> 15
 *   (p quantised to 1/256 -- 0.25 is exact --, kept probabilities scaled by 256 / (256 - floor(256 p))),
 *   rowkey = high 32 bits of splitmix64(key(seed, stream_id) + (ptr[b] + rng_rowoff[b] + i)*nhead + h); rng_rowoff (device int64
 *   [nseg], NULL = zeros) lets a rank of a bag-parallel job address the row ids the single-process run would use.
 *   Host restatement: advmil_amd/synth.py::attn_dropout_keep.
 * Arithmetic: split-bf16 ("bf16x3": hi.hi + hi.lo + lo.hi on v_mfma_f32_32x32x16_bf16, fp32 accumulate) in BOTH modes of
 *   advmil_set_gemm_mode -- ~2^-17 relative per product; softmax statistics and exponentials in fp32.
 * bwd: out / dout fp32 [Ltot, nhead*head_dim]; dqkv [Ltot, 3*nhead*head_dim] fp32 (every element written);
 *   ws >= advmil_mha_bwd_workspace_bytes (holds D = rowsum(dO * O) and the operand planes of dO). Deterministic (no atomics). */
int advmil_mha_fwd(const void* qkv_hi, const void* qkv_lo, int64_t Ltot, int nhead, int head_dim, int nseg, const int64_t* ptr,
                   int64_t max_len, float drop_p, const uint64_t* seed, uint64_t stream_id, const int64_t* rng_rowoff, float* out,
                   float* lse, advmil_stream_t stream);
/* The same forward with the log-sum-exp GIVEN (`lse` is read): the train-mode pass of an optimizer step behind the eval-mode pass over
 * the same q | k | v (the reference runs the generator twice per step with unchanged weights, model_handler.py:398-425; the softmax
 * statistics of the two passes are identical, only the dropout draw is new). Probabilities are exp2(s c - lse) directly: no running
 * maximum, no rescaling, no row sum. Dropout must be on (seed != NULL, drop_p > 0); same mask stream as advmil_mha_fwd. */
int advmil_mha_fwd_lse(const void* qkv_hi, const void* qkv_lo, int64_t Ltot, int nhead, int head_dim, int nseg, const int64_t* ptr,
                       int64_t max_len, float drop_p, const uint64_t* seed, uint64_t stream_id, const int64_t* rng_rowoff, float* out,
                       const float* lse, advmil_stream_t stream);
size_t advmil_mha_bwd_workspace_bytes(int64_t Ltot, int nhead, int head_dim);
int advmil_mha_bwd(const void* qkv_hi, const void* qkv_lo, const float* out, const float* dout, const float* lse, int64_t Ltot,
                   int nhead, int head_dim, int nseg, const int64_t* ptr, int64_t max_len, float drop_p, const uint64_t* seed,
                   uint64_t stream_id, const int64_t* rng_rowoff, float* dqkv, void* ws, size_t ws_bytes, advmil_stream_t stream);
/* The same backward in ONE pass over the scores (csrc/attn_bwd1.hip; round 6): keys stationary, dK / dV as above, and dQ from the
 * same score tiles -- dS goes through LDS once, each 256-key block leaves an unscaled partial slab [Ltot, nhead*head_dim] in the
 * workspace, and a reduce launch sums the blocks of a bag in block order into the q columns of dqkv. Five contractions instead of the
 * seven of advmil_mha_bwd (which recomputes S and dP for dQ), same arguments, same dropout stream, deterministic (no atomics); results
 * differ from advmil_mha_bwd's only in the summation order of dQ. ws >= advmil_mha_bwd1_workspace_bytes (D, the planes of dO and
 * ceil(max_len / 256) partial slabs). */
size_t advmil_mha_bwd1_workspace_bytes(int64_t Ltot, int nhead, int head_dim, int64_t max_len);
int advmil_mha_bwd1(const void* qkv_hi, const void* qkv_lo, const float* out, const float* dout, const float* lse, int64_t Ltot,
                    int nhead, int head_dim, int nseg, const int64_t* ptr, int64_t max_len, float drop_p, const uint64_t* seed,
                    uint64_t stream_id, const int64_t* rng_rowoff, float* dqkv, void* ws, size_t ws_bytes, advmil_stream_t stream);
/* Post-norm residual of the same layer (norm_first = False):  y = LayerNorm(x + dropout(o)) over rows of width d <= 512.
 * fwd also writes z = x + dropout(o), mean[R], rstd[R] for the backward; dropout element index = row*d + col on `stream_id`.
 * bwd: dx = LayerNorm'(dy); dob (may be NULL) = dx * keep; dgamma / dbeta = column sums (accumulate != 0 adds into them).
 * ws >= advmil_add_dropout_ln_bwd_workspace_bytes. */
/* y_hi / y_lo (both or neither): also the bf16x3 operand planes of y, for the plane-fed contraction that reads it next. */
int advmil_add_dropout_ln_fwd(const float* x, const float* o, const float* gamma, const float* beta, float eps, int64_t R,
                              int64_t d, float drop_p, const uint64_t* seed, uint64_t stream_id, const int64_t* rng_row, float* z,
                              float* y, float* mean, float* rstd, void* y_hi, void* y_lo, advmil_stream_t stream);
size_t advmil_add_dropout_ln_bwd_workspace_bytes(int64_t R, int64_t d);
int advmil_add_dropout_ln_bwd(const float* dy, const float* z, const float* gamma, const float* mean, const float* rstd, int64_t R,
                              int64_t d, float drop_p, const uint64_t* seed, uint64_t stream_id, const int64_t* rng_row, float* dx,
                              float* dob, float* dgamma, float* dbeta, int accumulate, void* ws, size_t ws_bytes,
                              advmil_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Gated-attention MIL pooling (Attn_Net_Gated + softmax + mm: model/backbone_utils.py:11-29,
 * model/backbone.py:81-85, 118-122, 163-167; GAPool: model/backbone_utils.py:47-56).
 *
 * gate_score: ab[N,2D] holds tanh-branch a (cols 0..D-1) and sigmoid-branch b (cols D..2D-1),
 *   post-activation, PRE-dropout. s[n] = sum_j (a*ka)(b*kb) wc[j] + bc[0], ka/kb the dropout
 *   factors (stream_a/stream_b, element index n*D+j).
 * softmax_pool_fwd: A = softmax(s) over the N instances; pooled[d] = sum_n A[n] h[n,d].
 *   ws >= advmil_softmax_pool_workspace_bytes. Deterministic two-stage reduction.
 * softmax_pool_bwd: ds[n] = A[n] * (dA[n] + dot(dpooled, h[n,:]) - sum_m A[m](dA[m] + dot(dpooled,h[m,:])));
 *   dA may be NULL. (dh gets A[n]*dpooled[d] through the rank-1 term of the gemm epilogue.)
 * gate_bwd: from ds -> dG[N,2D] = grads wrt the two pre-activations, plus dwc[D], dbc[1], dbias[2D]
 *   (column sums of dG).
 * pair32 (gate_bwd): ab and dG hold the branches in blocks of 32 columns ([a_0..31 | b_0..31 | a_32..63 | ...], the layout the fused training
 *   gate score stores: advmil_epilogue_t.gate_bits_a) instead of [a | b] halves; the parameter gradients keep their own order.
 * dG_hi / dG_lo (gate_bwd), out_hi / out_lo (act_dropout_bwd): optional (both or neither) bf16 [rows, cols] buffers that also receive
 *   the bf16x3 operand planes of the result, for the plane-fed contraction that reads it next.
 * `accumulate` (here and in the other backward entry points): non-zero ADDS the parameter gradients into the
 *   destination instead of overwriting it -- the destinations are then views of the flat gradient arena, which
 *   removes the per-parameter accumulate launches autograd would issue for every bag. */
int advmil_gate_score_fwd(const float* ab, const float* wc, const float* bc, float drop_p, const uint64_t* seed,
                          uint64_t stream_a, uint64_t stream_b, int64_t N, int64_t D, float* s, const int64_t* rng_row,
                          advmil_stream_t stream);
/* Segmented form: the N rows are a ragged slab of `nseg` bags, bag b = rows [seg_ptr[b], seg_ptr[b+1]) (device int64
 * array; NULL = one segment), max_len = longest segment. pooled / dpooled are [nseg, D]; A, s, ds are [N]. */
size_t advmil_softmax_pool_workspace_bytes(int64_t max_len, int64_t D, int nseg);
int advmil_softmax_pool_fwd(const float* s, const float* h, int64_t ldh, int64_t N, int64_t D, int nseg,
                            const int64_t* seg_ptr, int64_t max_len, float* A, float* pooled, void* ws, size_t ws_bytes,
                            advmil_stream_t stream);
/* Same, plus mean[nseg, D] = the per-bag UNWEIGHTED mean of h's rows from the same pass over h (the projection discriminator's region-level
 * inner product needs mean_r(fc_ins) beside the pooled fc_ins, GANSurv.py:96-98). D % 8 == 0. */
size_t advmil_softmax_pool_mean_workspace_bytes(int64_t max_len, int64_t D, int nseg);
int advmil_softmax_pool_mean_fwd(const float* s, const float* h, int64_t ldh, int64_t N, int64_t D, int nseg, const int64_t* seg_ptr,
                                 int64_t max_len, float* A, float* pooled, float* mean, void* ws, size_t ws_bytes, advmil_stream_t stream);
int advmil_softmax_pool_bwd(const float* dpooled, const float* dA, const float* A, const float* h, int64_t ldh,
                            int64_t N, int64_t D, int nseg, const int64_t* seg_ptr, int64_t max_len, float* ds, void* ws,
                            size_t ws_bytes, advmil_stream_t stream);
/* The same two calls with h held as its bf16x3 operand planes (h = h_hi + h_lo, exact in fp32; pitch ldh in elements, D % 8 == 0, planes
 * 16-byte aligned): the pooled tensor of a slab whose producing contraction wrote planes only (advmil_epilogue_t.c_hi with C == NULL) --
 * round 6: the generator's first layer keeps no fp32 copy of its [rows, 384] output. */
int advmil_softmax_pool_fwd_planes(const float* s, const void* h_hi, const void* h_lo, int64_t ldh, int64_t N, int64_t D, int nseg,
                                   const int64_t* seg_ptr, int64_t max_len, float* A, float* pooled, void* ws, size_t ws_bytes,
                                   advmil_stream_t stream);
int advmil_softmax_pool_bwd_planes(const float* dpooled, const float* dA, const float* A, const void* h_hi, const void* h_lo, int64_t ldh,
                                   int64_t N, int64_t D, int nseg, const int64_t* seg_ptr, int64_t max_len, float* ds, void* ws,
                                   size_t ws_bytes, advmil_stream_t stream);
/* Train-mode dropout of a tensor held as operand planes [M, N] (N % 32 == 0): out planes = split(dropout(in_hi + in_lo)), drawn at
 * (seed, stream_id, element index rng_row[m] * N + n) exactly as advmil_act_dropout_bwd's replay draws it, and bits[m * N / 32 + n / 32]
 * bit n % 32 = (out > 0) (NULL: no bits). The train-mode forward of a ReLU layer whose eval-mode output is memoized as planes
 * (model/backbone.py:60-66 run twice per optimizer step, model_handler.py:398-400 / 420-425).
 * gate_bits_a / gate_bits_b (optional, both or neither; uint32 [M, N / 32]): the KEEP bits of two more dropouts of rate gate_p over the same
 * [M, N] index space, streams gate_stream_a / _b -- the gated attention scorer's branch dropouts (model/backbone_utils.py:24-29), drawn here
 * for the fused training gate score (advmil_epilogue_t.gate_bits_a). */
int advmil_dropout_planes(const void* in_hi, const void* in_lo, int64_t M, int64_t N, float drop_p, const uint64_t* seed, uint64_t stream_id,
                          const int64_t* rng_row, void* out_hi, void* out_lo, void* bits, float gate_p, uint64_t gate_stream_a,
                          uint64_t gate_stream_b, void* gate_bits_a, void* gate_bits_b, advmil_stream_t stream);
/* dh[n, :] = A[n] * dpooled[rowseg[n], :] -- the backward of a pooling with constant weights (the per-bag mean of the region features in
 * the projection discriminator's region-level inner product, GANSurv.py:96-98). rowseg: int32 bag index per row, NULL = one bag. */
int advmil_seg_scale_rows(const float* dpooled, const float* A, const int32_t* rowseg, int64_t N, int64_t D, float* dh,
                          advmil_stream_t stream);
size_t advmil_gate_bwd_workspace_bytes(int64_t N, int64_t D);
int advmil_gate_bwd(const float* ab, const float* ds, const float* wc, float drop_p, const uint64_t* seed,
                    uint64_t stream_a, uint64_t stream_b, int64_t N, int64_t D, float* dG, float* dwc, float* dbc,
                    float* dbias, int accumulate, const int64_t* rng_row, void* dG_hi, void* dG_lo, int pair32, void* ws, size_t ws_bytes,
                    advmil_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Backward of y = dropout(act(pre)) for the Linear layers: dpre = dy * keep * act'(y) and
 * dbias[n] = sum_m dpre[m,n] (NULL to skip). y is the stored post-dropout output.
 * replaces: autograd of ReLU/Tanh/Sigmoid/Dropout modules (model/backbone.py:70-75 etc.). */
size_t advmil_colsum_workspace_bytes(int64_t M, int64_t N);
/* bits (optional, N % 32 == 0): uint32 [M, N / 32], bit c % 32 of word (m, c / 32) = (dpre[m, c] > 0) -- with act NONE and dy = y = the
 * memoized pre-dropout activations this call IS the train-mode dropout of a ReLU layer, and the bits are that layer's ReLU-and-kept mask
 * for the backward (advmil_epilogue_t.maskbits): 1/32 of the bytes of reading the stored output back. */
int advmil_act_dropout_bwd(const float* dy, const float* y, int act, float drop_p, const uint64_t* seed,
                           uint64_t stream_id, int64_t M, int64_t N, float* dpre, float* dbias, int accumulate,
                           const int64_t* rng_row, void* out_hi, void* out_lo, void* bits, void* ws, size_t ws_bytes,
                           advmil_stream_t stream);
/* out[n] (+)= sum_m x[m,n] */
int advmil_colsum(const float* x, int64_t M, int64_t N, float* out, int accumulate, void* ws, size_t ws_bytes,
                  advmil_stream_t stream);

/* Deferred merges of parameter-gradient partials. Every backward entry point above that ADDS a parameter gradient into its destination
 * (`accumulate` != 0: the destinations are slots of the optimizer's flat gradient arena, model_handler.py:405-409 / 486-497 sum the
 * per-bag gradients the same way through autograd's accumulation) first leaves per-workgroup partial rows -- or split-K partial tiles:
 * advmil_gemm_f32 with splits > 1, alpha = 1 and no other epilogue term -- in the caller's workspace and then folds them with a second,
 * launch-bound kernel. advmil_defer_sums(stream, 1) puts `stream` into deferral: those folds are queued instead (up to 16; a fold into a
 * slot that already has one queued flushes first) and advmil_flush_sums(stream) / advmil_defer_sums(stream, 0) issue them as ONE launch.
 * Contract: every workspace handed to such a call stays alive and untouched until the flush, and nothing reads the destinations before
 * it. Non-accumulating folds are never deferred. advmil_pending_sums: queued folds (-1: the stream is not in deferral). */
int advmil_defer_sums(advmil_stream_t stream, int on);
int advmil_flush_sums(advmil_stream_t stream);
int advmil_pending_sums(advmil_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Region embedding tail of AVGPoolPatchEmbedding (model/backbone_utils.py:158-168):
 * y[N,d] (the 1x1-conv / FC output) -> LayerNorm(d, eps) -> ReLU -> mean over each consecutive 16
 * rows -> emb[N/16,d]. Saves mean/rstd per row for the backward. N % 16 == 0 (backbone_utils.py:65).
 * bwd: demb[N/16,d] -> dy[N,d], dgamma[d], dbeta[d]; dycol (optional, [d]) += column sums of dy -- the bias gradient of the FC
 * that produced y, which otherwise costs a second pass over dy. dy_hi / dy_lo (both or neither): the bf16x3 operand planes of dy for
 * the weight-gradient contraction dy^T X that consumes it (advmil_epilogue_t.a_hi / a_lo); with them dy may be NULL (planes only). */
/* emb_hi / emb_lo (both or neither; d % 128 == 0): also the bf16x3 operand planes of emb (the ESAT in-projection reads it plane-fed). */
/* dup = 2 (d % 128 == 0): emb has 2 N / 16 rows, the embedding written twice ([emb; emb]: the discriminator update feeds the fake and the
 * real pass of the region level as one stacked batch, model_handler.py:373-401); the backward then takes demb [2 N / 16, d] and sums its
 * halves on load -- neither a concatenation nor an add launch. dup = 1: the plain call. */
int advmil_ln_relu_mean16_fwd(const float* y, const float* gamma, const float* beta, float eps, int64_t N, int64_t d,
                              float* emb, float* mean, float* rstd, void* emb_hi, void* emb_lo, int dup, advmil_stream_t stream);
size_t advmil_ln_relu_mean16_bwd_workspace_bytes(int64_t N, int64_t d);
int advmil_ln_relu_mean16_bwd(const float* demb, const float* y, const float* gamma, const float* beta,
                              const float* mean, const float* rstd, int64_t N, int64_t d, float* dy, float* dgamma,
                              float* dbeta, int accumulate, float* dycol, void* dy_hi, void* dy_lo, int dup, void* ws, size_t ws_bytes,
                              advmil_stream_t stream);

/* Plain row-wise LayerNorm(d) -> ReLU (the norm='layer' MLP inside GENConv; N arbitrary). Same kernels as above. */
int advmil_ln_relu_fwd(const float* y, const float* gamma, const float* beta, float eps, int64_t N, int64_t d, float* out,
                       float* mean, float* rstd, advmil_stream_t stream);
size_t advmil_ln_relu_bwd_workspace_bytes(int64_t N, int64_t d);
int advmil_ln_relu_bwd(const float* dout, const float* y, const float* gamma, const float* beta, const float* mean,
                       const float* rstd, int64_t N, int64_t d, float* dy, float* dgamma, float* dbeta, int accumulate,
                       void* ws, size_t ws_bytes, advmil_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * PatchGCN's GENConv softmax aggregation (model/backbone.py:139,157 -> torch_geometric.nn.GENConv, aggr='softmax',
 * learn_t; parity unpinned: restated from the published semantics). x[N,C]; the graph arrives as two int32 CSR images:
 *   by destination (rowptr_dst[N+1], col_src[E] = source of each in-edge)  -> forward
 *   by source      (rowptr_src[N+1], col_dst[E] = target of each out-edge) -> backward
 * fwd: out = agg + x, agg = sum_j softmax_j(t*m_j) m_j, m = relu(x)+eps; saves lse (log-sum-exp of t*m over the in-edges) and agg for
 *      the backward (both NULL: nothing is saved, an evaluation pass writes `out` only).
 * bwd: dx = dout + relu'(x) * sum_{j->i} dout_i w_ij (1 + t (m_j - agg_i)), w_ij = exp(t m_j - lse_i);
 *      dt[0] = sum_{j->i, c} dout_i w_ij m_j (m_j - agg_i), from the same edge walk (per-workgroup partials in `ws`, summed in a fixed
 *      order: run-to-run identical). ws_bytes >= advmil_genconv_bwd_workspace_bytes(N, C). No atomics in either direction. */
int advmil_genconv_fwd(const float* x, const int32_t* rowptr_dst, const int32_t* col_src, const float* t, float eps, int64_t N,
                       int64_t C, float* out, float* lse, float* agg, advmil_stream_t stream);
size_t advmil_genconv_bwd_workspace_bytes(int64_t N, int64_t C);
int advmil_genconv_bwd(const float* dout, const float* x, const float* agg, const float* lse, const int32_t* rowptr_src,
                       const int32_t* col_dst, const float* t, float eps, int64_t N, int64_t C, float* dx, float* dt, void* ws,
                       size_t ws_bytes, advmil_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Optimizer + regulariser over a flat parameter arena (torch.optim.Adam, L2-in-grad weight decay:
 * optim/optim_factory.py:25-37,76-77; model/model_handler.py:104-107; L1: loss/utils.py:6-14).
 *   g = grad*grad_scale + l1_coef*sign(p) + wd[i]*p ; Adam(m, v) ; p -= lr/(1-b1^t) * m/(sqrt(v)/sqrt(1-b2^t)+eps)
 * `step` is a device int32 incremented by the kernel (graph-replay safe). wd may be NULL. The arenas are 16-byte aligned (planes: 8). p_hi / p_lo (both or neither): bf16 arenas of
 * n elements that receive the bf16x3 operand planes of the UPDATED weights, so the contractions never re-split a weight.
 * abs_sum: out[0] = sum |p| (for the logged Loss_G_total). */
/* tick != 0: the kernel's own second launch increments `step`; tick = 0: the caller increments it (advmil_step_seed_tick folds that into the
 * RNG seed's advance at the end of an optimizer step: one one-thread launch instead of two). */
/* abs_partial (NULL: skipped): advmil_adam_blocks(n) floats, abs_partial[i] = the i-th workgroup's share of sum |p| BEFORE the update (the logged
 * value of the L1 term, summed on the host when the log is read). clear_grad != 0: grad is zeroed behind its last read (the next step then
 * needs no fill launch; p.grad reads zero afterwards -- the captured step uses it, the eager handler does not). */
int advmil_adam_blocks(int64_t n);
int advmil_adam_step(float* p, float* grad, float* m, float* v, const float* wd, int64_t n, float lr,
                     float beta1, float beta2, float eps, float grad_scale, float l1_coef, int32_t* step, void* p_hi, void* p_lo,
                     int tick, float* abs_partial, int clear_grad, advmil_stream_t stream);
/* step[0] += 1, step2[0] += 1 (the two networks' counters; NULL: skipped) and seed[0] += inc (NULL: skipped) in one launch */
int advmil_step_seed_tick(int32_t* step, int32_t* step2, uint64_t* seed, uint64_t inc, advmil_stream_t stream);
int advmil_abs_sum(const float* p, int64_t n, float* out, void* ws, size_t ws_bytes, advmil_stream_t stream);
size_t advmil_abs_sum_workspace_bytes(int64_t n);

/* Bag ingest from the device-resident bag cache (replaces the per-bag, per-epoch `.cuda()` of model/model_handler.py:315 for a bag
 * that has been seen before): one launch copies rows_bytes of fp32 rows and, when the four plane pointers are given, plane_bytes of
 * each bf16 operand plane, device to device, into the step slab. All pointers and sizes multiples of 16 bytes; planes all or none.
 * dst_hi / dst_lo given with src_hi = src_lo = NULL: the planes are DERIVED from the fp32 rows on the way (hi = bf16(x), lo = bf16(x - hi),
 * as advmil_split_planes; plane_bytes = rows_bytes / 2) -- the cache then holds the fp32 rows only. src == dst rows is allowed in that
 * form (split in place of rows that just arrived over PCIe). */
int advmil_stage_bag(void* dst_rows, const void* src_rows, size_t rows_bytes, void* dst_hi, const void* src_hi, void* dst_lo,
                     const void* src_lo, size_t plane_bytes, advmil_stream_t stream);
/* fill out[i] = U[0,1) from the counter RNG (generator noise, utils/func.py:154-164) */
int advmil_uniform_fill(float* out, int64_t n, const uint64_t* seed, uint64_t stream_id, const int64_t* rng_row, int64_t width,
                        advmil_stream_t stream);
/* seed[0] += inc  (advance the step seed between graph replays) */
/* y = x * keep/(1-p) with keep drawn at flat index i of (seed, stream_id): nn.Dropout on the [B, d]-sized head tensors
 * (model/model_utils.py:106-176 make_mlp_layer; applied to dy it is the backward). In place (y == x) allowed.
 * rng_row / width (here and in advmil_uniform_fill): optional row map, the flat tensor is [n / width, width] and row r draws as
 * row rng_row[r] (see advmil_epilogue_t.rng_row); NULL = identity. */
int advmil_dropout_apply(const float* x, float* y, int64_t n, float p, const uint64_t* seed, uint64_t stream_id,
                         const int64_t* rng_row, int64_t width, advmil_stream_t stream);
int advmil_seed_advance(uint64_t* seed, uint64_t inc, advmil_stream_t stream);
/* Measurement aid (bench.py): dst[0] = the device's constant-rate wall clock, written by a one-thread launch in stream order, so two of
 * them bracket a kernel inside a captured step graph; advmil_clock_rate_khz() = ticks per millisecond (0: unknown). Not used by the step. */
int advmil_stamp_clock(int64_t* dst, advmil_stream_t stream);
int64_t advmil_clock_rate_khz(void);
/* The step's two scalar losses over <= bp_every_batch values, value and analytic gradient in one launch each.
 * D loss (loss/utils.py:182-203 real_fake_loss with the reference's means taken over the GLOBAL counts, model_handler.py:412):
 *   which 0 = bce as shipped, 1 = hinge, 2 = wasserstein; real_mask selects the real pairs (event bags with a visible label);
 *   out3 = {loss, sum mask*real, sum fake}; g_fake / g_real = d loss / d score.
 * G loss (model_handler.py:468-486): total = reg + coef*gen, reg = inv_nv * sum vis*recon_term (loss/utils.py:21-41), gen = -inv_nf*sum fake;
 *   out3 = {total, reg, gen}; g_pred / g_fake = d total / d (pred, fake). */
/* Linear layers with in_features == 1 or out_features == 1 on [B, .] head tensors (first layer of make_embedding_y_layer,
 * model_utils.py:178-186; prj_layer, GANSurv.py:78-84; the generator's output layer): y = act(x W^T + bias), x[B,K], W[N,K].
 * bwd: dx (may be NULL), dW / dbias (may be NULL; accumulate != 0 adds into them), dpre = dy * act'(y). */
int advmil_skinny_linear_fwd(const float* x, const float* W, const float* bias, int B, int K, int N, int act, float* y,
                             advmil_stream_t stream);
int advmil_skinny_linear_bwd(const float* x, const float* W, const float* y, const float* dy, int B, int K, int N, int act, float* dx,
                             float* dW, float* dbias, int accumulate, advmil_stream_t stream);
/* Linear (+ bias, activation, dropout) on [B, d] head / tail tensors, B <= 32 rows -- the generator's rho / hop MLP, the discriminator's
 * bag-level MLPs and label embedding (reference model/GANSurv.py:30-49, 89-105; model_utils.py:116-186): y = dropout(act(x W^T + bias)),
 * x[M, K] (pitch ldx), W[N, K]; fp32 FMA; K and ldx multiples of 4, x / W 16-byte aligned. Dropout draw = the contraction epilogue's
 * (element m * N + n of stream `stream_id`, row m replaced by rng_row[m] when given).
 * bwd: dpre = dy * keep * act'(y); dW[N, K] / dbias[N] (each may be NULL; acc_* != 0 adds into them); dx[M, K] (pitch lddx; NULL = not
 * needed; needs the workspace). One launch, two with dx. */
int advmil_small_linear_fwd(const float* x, int64_t ldx, const float* W, const float* bias, int M, int N, int K, int act, float drop_p,
                            const uint64_t* seed, uint64_t stream_id, const int64_t* rng_row, float* y, advmil_stream_t stream);
size_t advmil_small_linear_bwd_workspace_bytes(int M, int N);
int advmil_small_linear_bwd(const float* dy, const float* y, const float* x, int64_t ldx, const float* W, int M, int N, int K, int act,
                            float drop_p, const uint64_t* seed, uint64_t stream_id, const int64_t* rng_row, float* dW, int acc_w,
                            float* dbias, int acc_b, float* dx, int64_t lddx, void* ws, size_t ws_bytes, advmil_stream_t stream);
/* Projection head of the discriminator (reference model/GANSurv.py:78-105, PrjDiscriminator.forward): out[b] = <u[b], t[b]> + <src[b], w> + bias[0]
 * for [B, d] head tensors -- u = the (region-mean) x embedding, t = the label embedding, src = the prj_layer's input (NULL: no projection
 * layer). bwd: du / dt / dsrc [B, d], dw [d], dbias [1] (each may be NULL; accumulate != 0 adds into dw / dbias). */
int advmil_prj_head_fwd(const float* u, const float* t, const float* src, const float* w, const float* bias, int B, int d, float* out,
                        advmil_stream_t stream);
int advmil_prj_head_bwd(const float* dout, const float* u, const float* t, const float* src, const float* w, int B, int d, float* du,
                        float* dt, float* dsrc, float* dw, float* dbias, int accumulate, advmil_stream_t stream);
/* The discriminator's bag-level tail as ONE launch each way (reference model/GANSurv.py:89-105 PrjDiscriminator.forward after the region
 * level: hid_x = net_pair_one.fc2(emb_bag), hid_t = net_pair_two(t), out = <u, hid_t> + prj_layer(hid_x | hid_t), u = hid_x ('bag') or the
 * region-mean embedding ('instance', RLIP); model_utils.py:157-186 builds the two chains). B <= 32 rows (the bags of an optimizer step, or
 * 2 x 16 for the stacked fake | real pass), every width <= 256: one workgroup walks both chains with the activations in LDS -- the six
 * forward and up to twelve backward launches of 5-10 us each that these [B, d] layers cost as separate kernels become two.
 *   layer: y = dropout(act(x W^T + bias)), W [N, K] row-major; dropout element index row * N + col on `stream_id` (row through rng_row when
 *          given: advmil_epilogue_t.rng_row); `y` [B, N] is written by the forward and read by the backward.
 *   backward: dW / dbias (NULL = not wanted) are ADDED into (optimizer arena slots); dxin / dtin / du NULL = not wanted.
 * The two chains must end at the same width d; u [B, d] or NULL (= the x chain's output); prj_src 0 none, 1 = x chain output, 2 = y chain output. */
#define ADVMIL_TAIL_MAXL 3
typedef struct {
  const float* W;
  const float* bias;
  float* dW;
  float* dbias;
  float* y;
  int K, N, act;
  float drop_p;
  uint64_t stream_id;
} advmil_dense_layer_t;
typedef struct {
  int B, nx, ny, prj_src;
  const float* xin;
  const float* tin;
  advmil_dense_layer_t x[ADVMIL_TAIL_MAXL];
  advmil_dense_layer_t y[ADVMIL_TAIL_MAXL];
  const float* u;
  const float* w_prj;
  const float* b_prj;
  float* dw_prj;
  float* db_prj;
  const uint64_t* seed;
  const int64_t* rng_row;
  float* out;        /* forward: [B] */
  const float* dout; /* backward: [B] */
  float* dxin;
  float* dtin;
  float* du;
} advmil_dtail_t;
int advmil_dtail_fwd(const advmil_dtail_t* a, advmil_stream_t stream);
int advmil_dtail_bwd(const advmil_dtail_t* a, advmil_stream_t stream);
/* Several small deep-K weight-gradient contractions C_i (+)= A_i^T B_i (A_i [K, M], B_i [K, N] row-major: the TN form of advmil_gemm_f32) as
 * ONE launch: the three weight gradients behind the fused region network are 2-8 tiles x splits each -- side by side they overlap instead of
 * paying a launch + K-walk latency apiece. 64x64 tiles, split-K per member (>= 256 of K per workgroup), the engine's arithmetic mode and k
 * order: bit-identical to advmil_gemm_f32_tiled(tile 11) with the same split count. n <= 4; M, N, lda, ldb multiples of 4; the merges of the
 * partial tiles are deferrable (advmil_defer_sums) when accumulate != 0 and ldc == N. */
typedef struct {
  int64_t M, N, K;
  const float* A;
  int64_t lda;
  const float* B;
  int64_t ldb;
  float* C;
  int64_t ldc;
  int32_t accumulate;
} advmil_gemm_tn_call_t;
size_t advmil_gemm_tn_group_workspace_bytes(const advmil_gemm_tn_call_t* calls, int n);
int advmil_gemm_tn_group(const advmil_gemm_tn_call_t* calls, int n, void* ws, size_t ws_bytes, advmil_stream_t stream);
/* The generator's bag-level head as two launches each way (csrc/ghead.hip; reference model/GANSurv.py:13-46 `Generator.forward` behind the
 * backbone's pooling: ABMIL's `rho` = Linear(d0, d1) -> ReLU -> Dropout(p1) (model/backbone.py:66-70; d1 = 0: the backbone has none),
 * MLPs[0] = Linear(d1 | d0, d2) -> ReLU -> Dropout(p2), the noise [B, d2] concatenated, MLPs[1] = Linear(2 d2 | d2, 1), out_scale
 * (model/model_utils.py:124-140 make_noise_mlp_layer, hops = 1, noise = [0, 1]). fp32 FMA, fixed summation order.
 *   noise_mode 0: no noise input (W1 is [1, d2]); 1: zeros (W1 is [1, 2 d2], the noise half contributes nothing); 2: the caller's `noise`
 *   [B, d2]; 3: drawn in the kernel, U[0, 1) at site sid_noise, element rng_row(b) * d2 + n -- the draws of advmil_uniform_fill.
 *   out_act 0: identity, 1: sigmoid. Dropout: stream sid1 at element rng_row(b) * d1 + c (rho), sid2 at rng_row(b) * d2 + n (MLPs[0]).
 * fwd writes hs [B, d1 | d2] (the first hidden layer, post-dropout), h2 [B, d2] (d1 > 0 only), pred [B]; bwd reads them back with dpred [B],
 * writes dx [B, d0] (NULL: not wanted) and ADDS the weight / bias gradients in place (arena slots; NULL: not wanted).
 * B <= 32, d0 <= 512, d0 % 4 == 0, (d1 > 0 ? d1 : d2) % 16 == 0, d2 <= 256, d2 % 4 == 0; ws: advmil_ghead_workspace_bytes. */
typedef struct {
  int32_t B, d0, d1, d2;
  int32_t noise_mode, out_act;
  const float* x;
  int64_t ldx;
  const float* Wr; /* [d1, d0] */
  const float* br;
  const float* W0; /* [d2, d1 | d0] */
  const float* b0;
  const float* W1; /* [1, d2 | 2 d2] */
  const float* b1;
  float p1, p2;
  const uint64_t* seed;
  uint64_t sid1, sid2, sid_noise;
  const int64_t* rng_row;
  const float* noise;
  float* hs;
  float* h2;
  float* pred;
  const float* dpred;
  float* dx;
  int64_t lddx;
  float* dWr;
  float* dbr;
  float* dW0;
  float* db0;
  float* dW1;
  float* db1;
  float* ws;
  size_t ws_bytes;
} advmil_ghead_t;
size_t advmil_ghead_workspace_bytes(int B, int d0, int d1, int d2);
int advmil_ghead_fwd(const advmil_ghead_t* a, advmil_stream_t stream);
int advmil_ghead_bwd(const advmil_ghead_t* a, advmil_stream_t stream);
/* The discriminator's region-level network as one launch each way (reference model/model_utils.py:188-210 EmbedXLayer: fc1 = Linear(d, d/2)
 * -> ReLU -> Dropout -> Linear(d/2, d); model/backbone_utils.py:31-56 GAPool's scorer tanh(Linear(d, d)) * sigmoid(Linear(d, d)) -> Linear(d, 1))
 * over the R region rows of a step slab, d = 128 (the shipped disc_netx_out_dim; other widths keep the layer-by-layer path):
 *   h1 = dropout_p1(relu(e W1^T + b1)) [R, 64];  fc = h1 W2^T + b2 [R, 128];  ab = tanh | sigmoid (fc Wab^T + bab) [R, 256] (pre-dropout);
 *   s[r] = sum_j drop_pg(a_j) drop_pg(b_j) wc_j + bc.
 * bf16x3 arithmetic (split-bf16 products on the bf16 matrix pipe, fp32 accumulate, the contraction engine's order). Weights arrive as their
 * operand planes (hi = bf16(w), lo = bf16(w - hi), row-major [out, in]; Wab = the tanh branch's rows, then the sigmoid branch's).
 * Dropout draws: stream sid1 at element r * 64 + c for h1, streams sida / sidb at r * 128 + j for the two gate branches (row r through
 * rng_row when given) -- the draws of the separate launches. h1 / ab may be NULL (a pass nobody differentiates).
 * bwd: from ds[R] (gradient wrt the scores), A[R] and dpooled[nseg, 128] (the pooling's direct path A[r] dpooled[bag(r)]), dmean[nseg, 128] or
 * NULL (gradient wrt the per-bag mean of fc: + dmean[bag(r)] / len(bag(r))), dfc_add[R, 128] or NULL (any other gradient wrt fc):
 *   dG [R, 256], dfc [R, 128], dpre [R, 64] (= d h1 masked by the ReLU / dropout of h1), de [R, 128] (NULL: not wanted) are written;
 *   dwc[128], dbab[256], dbc[1], db2[128], db1[64] are ADDED into (arena slots; the merge of the per-workgroup partial rows is deferrable:
 *   advmil_defer_sums). The three weight gradients dWab = dG^T fc, dW2 = dfc^T h1, dW1 = dpre^T e stay contractions of the engine.
 *   WabT / W2T / W1T: operand planes of the TRANSPOSED weights ([128, 256], [64, 128], [128, 64]), made by advmil_dx_chain_prep. */
int advmil_dx_chain_fwd(const float* e, int64_t R, int d, const void* W1_hi, const void* W1_lo, const float* b1, const void* W2_hi,
                        const void* W2_lo, const float* b2, const void* Wab_hi, const void* Wab_lo, const float* bab, const float* wc,
                        const float* bc, float p1, float pg, const uint64_t* seed, uint64_t sid1, uint64_t sida, uint64_t sidb,
                        const int64_t* rng_row, float* h1, float* fc, float* ab, float* s, advmil_stream_t stream);
size_t advmil_dx_chain_bwd_workspace_bytes(int64_t R, int d);
int advmil_dx_chain_prep(const float* W1, const float* W2, const float* Wab, int d, void* W1T_hi, void* W1T_lo, void* W2T_hi, void* W2T_lo,
                         void* WabT_hi, void* WabT_lo, advmil_stream_t stream);
int advmil_dx_chain_bwd(int64_t R, int d, const float* ds, const float* A, const float* dpooled, const float* dmean, const int32_t* rowseg,
                        const int64_t* seg_ptr, const float* dfc_add, const float* h1, const float* ab, const float* wc, float p1, float pg,
                        const uint64_t* seed, uint64_t sida, uint64_t sidb, const int64_t* rng_row, const void* WabT_hi, const void* WabT_lo,
                        const void* W2T_hi, const void* W2T_lo, const void* W1T_hi, const void* W1T_lo, float* dG, float* dfc, float* dpre,
                        float* de, float* dwc, float* dbab, float* dbc, float* db2, float* db1, void* ws, size_t ws_bytes,
                        advmil_stream_t stream);
int advmil_gan_d_loss(const float* fake, int nf, const float* real, const float* real_mask, int nr, int which, float inv_nf,
                      float inv_nr, float* out3, float* g_fake, float* g_real, advmil_stream_t stream);
int advmil_gan_g_loss(const float* pred, const float* t, const float* e, const float* vis_mask, const float* fake, int n, float alpha,
                      float gamma, int l2, float coef, float inv_nf, float inv_nv, float* out3, float* g_pred, float* g_fake,
                      advmil_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Evaluator: concordance index for right-censored data, eval/cindex.py:79-143 (`_get_comparable`,
 * `_estimate_concordance_index`; python loops over samples there, all n^2 pair tests on the device here).
 * time[n], event[n] (!= 0: event observed), estimate[n] (risk: higher = earlier event; the reference passes -prediction,
 * eval/cindex.py:35,41). out6 (device, int64) = concordant, discordant, tied_risk, tied_time, comparable pairs, number of event
 * samples the reference would register (0 -> its NoComparablePairException). cindex = (concordant + 0.5*tied_risk) / comparable. */
int advmil_cindex_counts(const float* time, const float* event, const float* estimate, int64_t n, float tied_tol, int64_t* out6,
                         advmil_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif
