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

// grid (P): one workgroup per image plane; every thread walks interior pixels, 121 window taps each
__global__ __launch_bounds__(256) void ssim_kernel(SdmiSsimArgs p) {
  __shared__ double red[256];
  __shared__ double w1[11];
  if (threadIdx.x == 0) {
    double s = 0.0;
    for (int i = 0; i < 11; ++i) { w1[i] = exp(-0.5 * (double)((i - 5) * (i - 5)) / (1.5 * 1.5)); s += w1[i]; }
    for (int i = 0; i < 11; ++i) w1[i] /= s;
  }
  __syncthreads();
  const float* __restrict__ x = p.x + (long long)blockIdx.x * p.H * p.W;
  const float* __restrict__ y = p.y + (long long)blockIdx.x * p.H * p.W;
  const int h = p.H - 10, w = p.W - 10;
  const double L = (double)p.data_range;
  const double C1 = (0.01 * L) * (0.01 * L), C2 = (0.03 * L) * (0.03 * L);
  double acc = 0.0;
  for (int i = threadIdx.x; i < h * w; i += 256) {
    const int cy = i / w + 5, cx = i % w + 5;
    double ux = 0, uy = 0, uxx = 0, uyy = 0, uxy = 0;
    for (int dy = -5; dy <= 5; ++dy) {
      double rx = 0, ry = 0, rxx = 0, ryy = 0, rxy = 0;        // separable: rows first, like the filter
      const float* xr = x + (long long)(cy + dy) * p.W + cx;
      const float* yr = y + (long long)(cy + dy) * p.W + cx;
      for (int dx = -5; dx <= 5; ++dx) {
        const double a = (double)xr[dx], b = (double)yr[dx], g = w1[dx + 5];
        rx += g * a; ry += g * b; rxx += g * a * a; ryy += g * b * b; rxy += g * a * b;
      }
      const double g = w1[dy + 5];
      ux += g * rx; uy += g * ry; uxx += g * rxx; uyy += g * ryy; uxy += g * rxy;
    }
    const double vx = uxx - ux * ux, vy = uyy - uy * uy, vxy = uxy - ux * uy;
    acc += ((2.0 * ux * uy + C1) * (2.0 * vxy + C2)) / ((ux * ux + uy * uy + C1) * (vx + vy + C2));
  }
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) p.out[blockIdx.x] = red[0] / (double)(h * w);
}

}  // namespace

extern "C" int sdmi_ssim(const SdmiSsimArgs* a, void* stream) {
  SDMI_REQUIRE(a && a->x && a->y && a->out, "null pointer");
  SDMI_REQUIRE(a->P >= 1 && a->H >= 11 && a->W >= 11 && a->data_range > 0.f, "images of at least 11 x 11");
  hipLaunchKernelGGL(ssim_kernel, dim3(a->P), dim3(256), 0, (hipStream_t)stream, *a);
  return sdmi_check_launch("ssim");
}

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
