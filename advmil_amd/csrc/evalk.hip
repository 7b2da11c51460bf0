// Concordance index counts for right-censored data (eval/cindex.py:79-143 of the reference, the scikit-survival estimator with
// unit weights): all O(n^2) pair tests on the device, integer counters (bit-exact against the reference's python loops).
#include "common.h"
#include "../../include/advmil_hip.h"

// out[0..5] = concordant, discordant, tied_risk, tied_time, comparable, events_with_entry
__global__ __launch_bounds__(256) void cindex_counts_kernel(const float* __restrict__ time, const float* __restrict__ event,
                                                            const float* __restrict__ est, int64_t n, float tied_tol,
                                                            unsigned long long* __restrict__ out) {
  const int64_t i = blockIdx.x;
  if (!(event[i] != 0.0f)) return;                        // only samples with an event anchor comparable pairs (cindex.py:94)
  const float ti = time[i], ei = est[i];
  unsigned long long con = 0, dis = 0, tie = 0, tt = 0, comp = 0, ge = 0;
  for (int64_t j = threadIdx.x; j < n; j += 256) {
    if (j == i) continue;
    const float tj = time[j];
    const bool later = tj > ti;
    const bool same_cens = (tj == ti) && !(event[j] != 0.0f);     // censored at the same time (cindex.py:92-101)
    ge += (tj >= ti);
    if (later || same_cens) {
      ++comp;
      tt += same_cens;
      const float ej = est[j];
      if (fabsf(ej - ei) <= tied_tol) ++tie;                     // cindex.py:125
      else if (ej < ei) ++con;                                   // cindex.py:128
      else ++dis;
    }
  }
  __shared__ unsigned long long red[6][256];
  red[0][threadIdx.x] = con; red[1][threadIdx.x] = dis; red[2][threadIdx.x] = tie;
  red[3][threadIdx.x] = tt;  red[4][threadIdx.x] = comp; red[5][threadIdx.x] = ge;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s)
#pragma unroll
      for (int q = 0; q < 6; ++q) red[q][threadIdx.x] += red[q][threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
#pragma unroll
    for (int q = 0; q < 5; ++q)
      if (red[q][0]) atomicAdd(out + q, red[q][0]);
    // the reference registers an event unless it is the unique sample with the largest time (cindex.py:85-104)
    if (red[5][0]) atomicAdd(out + 5, 1ULL);
  }
}

extern "C" int advmil_cindex_counts(const float* time, const float* event, const float* estimate, int64_t n, float tied_tol,
                                    int64_t* out6, advmil_stream_t stream_) {
  if (!time || !event || !estimate || !out6 || n <= 0 || n > 0x7fffffff) return ADVMIL_EINVAL;
  hipStream_t stream = (hipStream_t)stream_;
  hipError_t err = hipMemsetAsync(out6, 0, 6 * sizeof(int64_t), stream);
  if (err != hipSuccess) return (int)err;
  hipLaunchKernelGGL(cindex_counts_kernel, dim3((unsigned)n), dim3(256), 0, stream, time, event, estimate, n, tied_tol,
                     (unsigned long long*)out6);
  ADVMIL_LAUNCH_CHECK();
  return ADVMIL_OK;
}
