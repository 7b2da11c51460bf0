// Merge of per-workgroup partial rows into a destination, launched at once or -- while the stream is in deferral
// (advmil_defer_sums) -- queued for ONE launch per flush. See sumq.hip.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define ADVMIL_SUMQ_CAP 16

// out[c] (+)= sum_{b < nblk} partial[b * stride + c] for c < ncols; columns >= c1 go to out1[c - c1], columns >= c2 to out2[c - c2]
// (NULL: a single destination).
struct SumDesc {
  const float* partial;
  int nblk;
  int64_t stride;
  int64_t ncols;
  float* out;
  int accumulate;
  float* out1;
  int64_t c1;
  float* out2;
  int64_t c2;
  int wide;      // set by advmil_sumq_push
  // rows of the [ncols / p32n, p32n] partial matrix are in pair-block order (r = 64 (j / 32) + 32 br + j % 32: the fused training gate
  // score's layout, dWab = dG^T h) while `out` holds [br][j] rows: the merge un-permutes on its way out (0: no permutation)
  int64_t p32n;
};

int advmil_sumq_push(hipStream_t stream, SumDesc d);

static inline int advmil_sumq(hipStream_t stream, const float* partial, int nblk, int64_t stride, int64_t ncols, float* out, int accumulate,
                              float* out1 = nullptr, int64_t c1 = 0, float* out2 = nullptr, int64_t c2 = 0, int64_t p32n = 0) {
  SumDesc d;
  d.partial = partial; d.nblk = nblk; d.stride = stride; d.ncols = ncols; d.out = out; d.accumulate = accumulate;
  d.out1 = out1; d.c1 = c1; d.out2 = out2; d.c2 = c2; d.wide = 0; d.p32n = p32n;
  return advmil_sumq_push(stream, d);
}
