// VQ-VAE stage-1 training helpers (include/sdmi.h: sdmi_transpose2d, sdmi_softmax_rows_bwd,
// sdmi_vq_bwd).  HBM-bound streaming kernels.
#include "common.h"

namespace {

// 64x64 tile through LDS (pitch 65: conflict-free column reads); grid (C/64, R/64, Z)
template <typename T>
__global__ __launch_bounds__(256) void transpose2d_kernel(SdmiTransposeArgs p) {
  __shared__ T tile[64][65];
  const int z = blockIdx.z;
  const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
  const T* __restrict__ src = (const T*)p.src + (long long)z * p.ss;
  T* __restrict__ dst = (T*)p.dst + (long long)z * p.sd;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int r = r0 + ty + i * 4, c = c0 + tx;
    if (r < p.R && c < p.C) tile[ty + i * 4][tx] = src[(long long)r * p.lds + c];
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int c = c0 + ty + i * 4, r = r0 + tx;
    if (r < p.R && c < p.C) dst[(long long)c * p.ldd + r] = tile[tx][ty + i * 4];
  }
}

// one workgroup per row: dot = sum dp*p (fp32, fixed tree), then ds in place
template <typename T>
__global__ __launch_bounds__(256) void softmax_rows_bwd_kernel(SdmiSoftmaxBwdArgs a) {
  __shared__ float red[4];
  const long long row = blockIdx.x;
  const T* __restrict__ P = (const T*)a.p + row * a.ld;
  T* __restrict__ D = (T*)a.dp + row * a.ld;
  float s = 0.f;
  for (int j = threadIdx.x; j < a.cols; j += 256) s += Elem<T>::ld(P + j) * Elem<T>::ld(D + j);
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  const float dot = (red[0] + red[1]) + (red[2] + red[3]);
  for (int j = threadIdx.x; j < a.cols; j += 256) {
    const float pj = Elem<T>::ld(P + j);
    Elem<T>::st(D + j, a.scale * pj * (Elem<T>::ld(D + j) - dot));
  }
}

__global__ __launch_bounds__(256) void vq_bwd_kernel(SdmiVqBwdArgs p) {
  const float g = p.g ? p.g[0] : 1.f;
  const float n = (float)p.R * (float)p.dim;
  const float cz = g * 2.f / n, cc = g * 2.f * p.beta / n;
  for (long long r = (long long)blockIdx.x * 256 + threadIdx.x; r < p.R; r += (long long)gridDim.x * 256) {
    const long long code = p.idx[r];
    for (int d = 0; d < p.ldz; ++d) {
      const long long o = r * p.ldz + d;
      if (d < p.dim) {
        const float diff = p.z[o] - p.zq[o];
        p.dz[o] = (p.dzq ? p.dzq[o] : 0.f) + cz * diff;
        atomicAdd(&p.dcode[code * p.dim + d], -cc * diff);
      } else {
        p.dz[o] = 0.f;
      }
    }
  }
}

}  // namespace

extern "C" int sdmi_transpose2d(const SdmiTransposeArgs* a, void* stream) {
  SDMI_REQUIRE(a && a->src && a->dst && a->Z >= 1 && a->R >= 1 && a->C >= 1, "bad args");
  SDMI_REQUIRE(a->dtype == SDMI_F32 || a->dtype == SDMI_BF16, "bad dtype");
  SDMI_REQUIRE(a->Z <= 65535 && (a->R + 63) / 64 <= 65535, "grid too large");
  dim3 grid((a->C + 63) / 64, (a->R + 63) / 64, a->Z);
  if (a->dtype == SDMI_BF16)
    hipLaunchKernelGGL(transpose2d_kernel<bf16_t>, grid, dim3(256), 0, (hipStream_t)stream, *a);
  else
    hipLaunchKernelGGL(transpose2d_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, *a);
  return sdmi_check_launch("transpose2d");
}

extern "C" int sdmi_softmax_rows_bwd(const SdmiSoftmaxBwdArgs* a, void* stream) {
  SDMI_REQUIRE(a && a->p && a->dp && a->rows >= 1 && a->cols >= 1 && a->ld >= a->cols, "bad args");
  SDMI_REQUIRE(a->dtype == SDMI_F32 || a->dtype == SDMI_BF16, "bad dtype");
  if (a->dtype == SDMI_BF16)
    hipLaunchKernelGGL(softmax_rows_bwd_kernel<bf16_t>, dim3(a->rows), dim3(256), 0, (hipStream_t)stream, *a);
  else
    hipLaunchKernelGGL(softmax_rows_bwd_kernel<float>, dim3(a->rows), dim3(256), 0, (hipStream_t)stream, *a);
  return sdmi_check_launch("softmax_rows_bwd");
}

extern "C" int sdmi_vq_bwd(const SdmiVqBwdArgs* a, void* stream) {
  SDMI_REQUIRE(a && a->z && a->zq && a->dz && a->dcode && a->idx, "null pointer");
  SDMI_REQUIRE(a->R >= 1 && a->dim >= 1 && a->ldz >= a->dim, "bad sizes");
  long long nb = (a->R + 255) / 256;
  if (nb > 2048) nb = 2048;
  hipLaunchKernelGGL(vq_bwd_kernel, dim3((int)nb), dim3(256), 0, (hipStream_t)stream, *a);
  return sdmi_check_launch("vq_bwd");
}
