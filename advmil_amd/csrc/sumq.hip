// Deferred partial sums. The backward kernels of the path leave per-workgroup partial rows (bias / gate / LayerNorm parameter gradients)
// or split-K partial tiles (the weight gradients dW = dY^T X) in a caller workspace and a second, launch-bound kernel folds them into the
// destination -- 14 such merge launches per optimizer step (8 x colsum_merge, 6 x split-K reduce: 0.10 ms of the 16-bag step, 0.08 ms of
// the 2-bag step where every launch costs its 5-6 us floor). Every destination of an ACCUMULATING merge is a slot of the optimizer's flat
// gradient arena that nothing reads before the optimizer step, so while a stream is in deferral (advmil_defer_sums) those merges are
// queued and ONE launch (multi_sum_kernel) performs all of them: one merge launch per backward instead of seven.
// The caller keeps the workspaces alive until the flush (advmil_amd/ops.py::deferred_sums holds them).
#include <mutex>
#include "common.h"
#include "sumq.h"
#include "../../include/advmil_hip.h"

struct MultiSumArgs {
  SumDesc d[ADVMIL_SUMQ_CAP];
  int blk0[ADVMIL_SUMQ_CAP + 1];   // first block of each entry
  int n;
};

// Entry with few partial rows and many columns (split-K partial tiles): a thread owns 4 consecutive columns and walks the rows with 8
// loads in flight. Entry with many partial rows (per-workgroup partial rows of a slab pass): 16 columns x 16 row-lanes per block, 16
// independent loads per lane, folded through LDS (the colsum_merge_kernel scheme of pool.hip).
__global__ __launch_bounds__(256) void multi_sum_kernel(MultiSumArgs a) {
  __shared__ float red[16][17];
  int e = 0;
  const int bid = (int)blockIdx.x;
#pragma unroll 1
  while (e + 1 < a.n && bid >= a.blk0[e + 1]) ++e;
  const SumDesc& d = a.d[e];
  const int lb = bid - a.blk0[e];
  if (d.wide) {
    const int64_t c = ((int64_t)lb * 256 + threadIdx.x) * 4;
    if (c >= d.ncols) return;
    const float* wp = d.partial + c;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    int z = 0;
    for (; z + 8 <= d.nblk; z += 8) {
      float4 p[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) p[u] = *reinterpret_cast<const float4*>(wp + (int64_t)(z + u) * d.stride);
#pragma unroll
      for (int u = 0; u < 8; ++u) { s.x += p[u].x; s.y += p[u].y; s.z += p[u].z; s.w += p[u].w; }
    }
    for (; z < d.nblk; ++z) {
      const float4 p = *reinterpret_cast<const float4*>(wp + (int64_t)z * d.stride);
      s.x += p.x; s.y += p.y; s.z += p.z; s.w += p.w;
    }
    float4* dst = reinterpret_cast<float4*>(d.out + c);
    if (d.p32n) {      // pair-block rows -> [branch][unit] rows (p32n % 4 == 0: the thread's 4 columns stay in one row)
      const int64_t r = c / d.p32n, cc = c - r * d.p32n, half = (d.ncols / d.p32n) >> 1;
      const int64_t rr = ((r >> 5) & 1) * half + ((r >> 6) << 5) + (r & 31);
      dst = reinterpret_cast<float4*>(d.out + rr * d.p32n + cc);
    }
    if (d.accumulate) { const float4 o = *dst; s.x += o.x; s.y += o.y; s.z += o.z; s.w += o.w; }
    *dst = s;
    return;
  }
  const int cl = threadIdx.x & 15, rl = threadIdx.x >> 4;
  const int64_t c = (int64_t)lb * 16 + cl;
  float sv[16];
#pragma unroll
  for (int u = 0; u < 16; ++u) sv[u] = 0.f;
  if (c < d.ncols) {
    int b = rl;
    for (; b + 240 < d.nblk; b += 256) {
#pragma unroll
      for (int u = 0; u < 16; ++u) sv[u] += d.partial[(int64_t)(b + 16 * u) * d.stride + c];
    }
    for (; b < d.nblk; b += 16) sv[0] += d.partial[(int64_t)b * d.stride + c];
  }
#pragma unroll
  for (int w = 8; w > 0; w >>= 1)
#pragma unroll
    for (int u = 0; u < w; ++u) sv[u] += sv[u + w];
  red[rl][cl] = sv[0];
  __syncthreads();
  if (rl == 0 && c < d.ncols) {
    float t = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) t += red[r][cl];
    float* dst = (d.out2 && c >= d.c2) ? d.out2 + (c - d.c2) : (d.out1 && c >= d.c1) ? d.out1 + (c - d.c1) : d.out + c;
    *dst = d.accumulate ? *dst + t : t;
  }
}

namespace {
struct Queue {
  hipStream_t stream;
  bool active;
  int n;
  SumDesc d[ADVMIL_SUMQ_CAP];
};
std::mutex g_mu;
Queue g_q[8];      // one per stream in deferral (the step runs on one stream; a side stream of the bench's probes may hold a second)

Queue* find(hipStream_t s, bool make) {
  for (auto& q : g_q)
    if (q.active && q.stream == s) return &q;
  if (make)
    for (auto& q : g_q)
      if (!q.active) { q.active = true; q.stream = s; q.n = 0; return &q; }
  return nullptr;
}

int blocks_of(const SumDesc& d) { return d.wide ? (int)((d.ncols / 4 + 255) / 256) : (int)((d.ncols + 15) / 16); }

int launch(hipStream_t stream, const SumDesc* d, int n) {
  if (n <= 0) return ADVMIL_OK;
  MultiSumArgs a;
  a.n = n;
  int b = 0;
  for (int i = 0; i < n; ++i) { a.d[i] = d[i]; a.blk0[i] = b; b += blocks_of(d[i]); }
  a.blk0[n] = b;
  hipLaunchKernelGGL(multi_sum_kernel, dim3((unsigned)b), dim3(256), 0, stream, a);
  ADVMIL_LAUNCH_CHECK();
  return ADVMIL_OK;
}

bool overlaps(const float* a0, int64_t an, const float* b0, int64_t bn) { return a0 && b0 && a0 < b0 + bn && b0 < a0 + an; }

// the ranges an entry writes: out[0, c1 or ncols), out1[0, (c2 or ncols) - c1), out2[0, ncols - c2)
void dests(const SumDesc& d, const float* (&p)[3], int64_t (&n)[3]) {
  p[0] = d.out; n[0] = d.out1 ? d.c1 : (d.out2 ? d.c2 : d.ncols);
  p[1] = d.out1; n[1] = d.out1 ? (d.out2 ? d.c2 : d.ncols) - d.c1 : 0;
  p[2] = d.out2; n[2] = d.out2 ? d.ncols - d.c2 : 0;
}
}  // namespace

int advmil_sumq_push(hipStream_t stream, SumDesc d) {
  if (!d.partial || !d.out || d.nblk <= 0 || d.ncols <= 0) return ADVMIL_EINVAL;
  d.wide = d.nblk <= 64 && !d.out1 && !d.out2 && (d.ncols & 3) == 0 && (d.stride & 3) == 0 &&
           ((((uintptr_t)d.partial) | ((uintptr_t)d.out)) & 15) == 0;
  if (d.p32n && (!d.wide || (d.p32n & 3) || d.ncols % (64 * d.p32n))) return ADVMIL_EINVAL;      // (the un-permuting merge: wide form only)
  std::lock_guard<std::mutex> lk(g_mu);
  Queue* q = d.accumulate ? find(stream, false) : nullptr;      // only merges INTO an accumulator are order-free until the flush
  if (!q) return launch(stream, &d, 1);
  const float* np[3]; int64_t nn[3];
  dests(d, np, nn);
  bool clash = q->n == ADVMIL_SUMQ_CAP;
  for (int i = 0; i < q->n && !clash; ++i) {
    const float* op[3]; int64_t on[3];
    dests(q->d[i], op, on);
    for (int u = 0; u < 3; ++u)
      for (int v = 0; v < 3; ++v) clash = clash || overlaps(np[u], nn[u], op[v], on[v]);
  }
  if (clash) {      // two merges into the same slot must not share a launch (both read-modify-write it): the queued ones go first
    const int rc = launch(stream, q->d, q->n);
    q->n = 0;
    if (rc) return rc;
  }
  q->d[q->n++] = d;
  return ADVMIL_OK;
}

extern "C" int advmil_defer_sums(advmil_stream_t stream_, int on) {
  hipStream_t stream = (hipStream_t)stream_;
  std::lock_guard<std::mutex> lk(g_mu);
  Queue* q = find(stream, on != 0);
  if (on) return q ? ADVMIL_OK : ADVMIL_EINVAL;       // (more than 8 streams in deferral at once)
  if (!q) return ADVMIL_OK;
  const int rc = launch(stream, q->d, q->n);
  q->n = 0;
  q->active = false;
  return rc;
}

extern "C" int advmil_flush_sums(advmil_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  std::lock_guard<std::mutex> lk(g_mu);
  Queue* q = find(stream, false);
  if (!q) return ADVMIL_OK;
  const int rc = launch(stream, q->d, q->n);
  q->n = 0;
  return rc;
}

extern "C" int advmil_pending_sums(advmil_stream_t stream_) {
  std::lock_guard<std::mutex> lk(g_mu);
  Queue* q = find((hipStream_t)stream_, false);
  return q ? q->n : -1;
}
