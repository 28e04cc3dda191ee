// Evaluation-metric kernels (include/sdmi.h: sdmi_contingency, sdmi_sqerr_rows).  HBM-bound
// integer / streaming work: coalesced 16-byte loads, LDS histogram, one integer atomic per
// non-empty table cell and workgroup.
#include "common.h"

namespace {

// grid (chunks, B): every workgroup histograms a slice of one image's pixels into an LDS copy of
// the [Kg][Kp] table (LDS integer atomics), then adds its non-zero cells to the global table.
__global__ __launch_bounds__(256) void contingency_kernel(SdmiContingencyArgs p) {
  extern __shared__ int hist[];
  const int cells = p.Kg * p.Kp;
  for (int i = threadIdx.x; i < cells; i += 256) hist[i] = 0;
  __syncthreads();
  const int b = blockIdx.y;
  const int* __restrict__ g = p.gt + (long long)b * p.P;
  const int* __restrict__ q = p.pred + (long long)b * p.P;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < p.P;
       i += (long long)gridDim.x * 256) {
    const int c = g[i], k = q[i];
    if ((unsigned)c < (unsigned)p.Kg && (unsigned)k < (unsigned)p.Kp) atomicAdd(&hist[c * p.Kp + k], 1);
  }
  __syncthreads();
  int* out = p.counts + (long long)b * cells;
  for (int i = threadIdx.x; i < cells; i += 256)
    if (hist[i]) atomicAdd(&out[i], hist[i]);
}

// grid (nchunk, B): fp64 sum of squared differences of one slice (fixed tree: deterministic)
__global__ __launch_bounds__(256) void sqerr_rows_kernel(SdmiSqErrArgs p) {
  __shared__ double red[256];
  const int b = blockIdx.y, j = blockIdx.x;
  const long long per = (p.n + p.nchunk - 1) / p.nchunk;
  const long long lo = (long long)j * per;
  long long hi = lo + per;
  if (hi > p.n) hi = p.n;
  const float* __restrict__ x = p.x + (long long)b * p.n;
  const float* __restrict__ y = p.y + (long long)b * p.n;
  double s = 0.0;
  for (long long i = lo + threadIdx.x; i < hi; i += 256) {
    const double d = (double)x[i] - (double)y[i];
    s += p.mode ? fabs(d) : d * d;
  }
  red[threadIdx.x] = s;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if ((int)threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
    __syncthreads();
  }
  if (threadIdx.x == 0) p.partial[(long long)b * p.nchunk + j] = red[0];
}

}  // namespace

extern "C" int sdmi_contingency(const SdmiContingencyArgs* a, void* stream) {
  SDMI_REQUIRE(a && a->gt && a->pred && a->counts, "null pointer");
  SDMI_REQUIRE(a->B >= 1 && a->P >= 1 && a->Kg >= 1 && a->Kp >= 1 && (long long)a->Kg * a->Kp <= 8192,
               "bad table size");
  long long chunks = (a->P + 4095) / 4096;         // >= 16 pixels per thread and workgroup
  if (chunks > 256) chunks = 256;
  hipLaunchKernelGGL(contingency_kernel, dim3((int)chunks, a->B), dim3(256),
                     (size_t)a->Kg * a->Kp * sizeof(int), (hipStream_t)stream, *a);
  return sdmi_check_launch("contingency");
}

extern "C" int sdmi_sqerr_rows(const SdmiSqErrArgs* a, void* stream) {
  SDMI_REQUIRE(a && a->x && a->y && a->partial, "null pointer");
  SDMI_REQUIRE(a->B >= 1 && a->n >= 1 && a->nchunk >= 1 && a->nchunk <= 65535, "bad sizes");
  hipLaunchKernelGGL(sqerr_rows_kernel, dim3(a->nchunk, a->B), dim3(256), 0, (hipStream_t)stream, *a);
  return sdmi_check_launch("sqerr_rows");
}
