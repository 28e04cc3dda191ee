// Backward of GroupNorm(+act,+residual) on NHWC and of LayerNorm (include/sdmi.h).
//
// GroupNorm:  z = xhat*gamma + beta (+res),  y = act(z),  xhat = (x - mean_g) * rstd_g
//   dz = dy * act'(z);  dres = dz;  dbeta_c = sum dz;  dgamma_c = sum dz*xhat
//   dx = rstd_g * ( gamma_c*dz - ( S1_g + xhat * S2_g ) / n ),
//        S1_g = sum_{c in g} gamma_c * A_bc,  S2_g = sum_{c in g} gamma_c * B_bc,
//        A_bc = sum_hw dz,  B_bc = sum_hw dz*xhat      (per image b, channel c)
// so one streaming pass produces the per-(image, channel) sums, a tiny reduce turns them into the
// group terms and the parameter gradients, and a second streaming pass writes dx (and dres).
// Same fixed-channel thread organisation as the forward; fp64 combines, no atomics.
#include "common.h"

namespace {

__device__ __forceinline__ int next_pow2(int v) {
  int p = 1;
  while (p < v) p <<= 1;
  return p;
}

// partial layout: [B][nsplit][C][2]; gsum: [B][groups][2] stored right after it.
template <typename T>
__global__ __launch_bounds__(256) void gn_bwd_stats_kernel(SdmiGroupNormBwdArgs p) {
  constexpr int VEC = Elem<T>::VEC;
  __shared__ float part[256][VEC][2];
  const int b = blockIdx.y, split = blockIdx.x;
  const int CV = p.C / VEC, CVp = next_pow2(CV);
  const int R = 256 / CVp;
  const int cv = threadIdx.x % CVp, r0 = threadIdx.x / CVp;
  const int rows_per = (p.HW + p.nsplit - 1) / p.nsplit;
  const int row_begin = split * rows_per;
  int row_end = row_begin + rows_per;
  if (row_end > p.HW) row_end = p.HW;
  const int cpg = p.C / p.groups;
  float A[VEC], Bv[VEC];
#pragma unroll
  for (int j = 0; j < VEC; ++j) A[j] = Bv[j] = 0.f;
  if (cv < CV) {
    float mu[VEC], rs[VEC], ga[VEC], be[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      const int c = cv * VEC + j, g = c / cpg;
      mu[j] = p.stats[(b * p.groups + g) * 2];
      rs[j] = p.stats[(b * p.groups + g) * 2 + 1];
      ga[j] = p.gamma[c];
      be[j] = p.beta[c];
    }
    const long long base = (long long)b * p.HW * p.C + cv * VEC;
    const T* xb = (const T*)p.x + base;
    const T* db = (const T*)p.dy + base;
    const T* rb = p.residual ? (const T*)p.residual + base : nullptr;
    for (int row = row_begin + r0; row < row_end; row += R) {
      const long long o = (long long)row * p.C;
      float x[VEC], dy[VEC], rr[VEC];
      unpack16<T>(*reinterpret_cast<const uint4*>(xb + o), x);
      unpack16<T>(*reinterpret_cast<const uint4*>(db + o), dy);
      if (rb) unpack16<T>(*reinterpret_cast<const uint4*>(rb + o), rr);
#pragma unroll
      for (int j = 0; j < VEC; ++j) {
        const float xh = (x[j] - mu[j]) * rs[j];
        const float z = xh * ga[j] + be[j] + (rb ? rr[j] : 0.f);
        const float dz = dy[j] * act_grad(z, p.act);
        A[j] += dz;
        Bv[j] += dz * xh;
      }
    }
  }
#pragma unroll
  for (int j = 0; j < VEC; ++j) { part[threadIdx.x][j][0] = A[j]; part[threadIdx.x][j][1] = Bv[j]; }
  __syncthreads();
  if (r0 == 0 && cv < CV) {
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      double sa = 0.0, sb = 0.0;
      for (int r = 0; r < R; ++r) { sa += part[r * CVp + cv][j][0]; sb += part[r * CVp + cv][j][1]; }
      float* q = p.partial + ((((long long)b * p.nsplit + split) * p.C) + cv * VEC + j) * 2;
      q[0] = (float)sa;
      q[1] = (float)sb;
    }
  }
}

// per image: channel totals over splits (written back into split 0) and the group terms
__global__ __launch_bounds__(256) void gn_bwd_reduce_kernel(SdmiGroupNormBwdArgs p, float* gsum) {
  __shared__ float tot[1024][2];
  const int b = blockIdx.x;
  for (int c = threadIdx.x; c < p.C; c += 256) {
    double sa = 0.0, sb = 0.0;
    for (int k = 0; k < p.nsplit; ++k) {
      const float* q = p.partial + ((((long long)b * p.nsplit + k) * p.C) + c) * 2;
      sa += q[0];
      sb += q[1];
    }
    tot[c][0] = (float)sa;
    tot[c][1] = (float)sb;
    float* q0 = p.partial + (((long long)b * p.nsplit) * p.C + c) * 2;
    q0[0] = (float)sa;
    q0[1] = (float)sb;
  }
  __syncthreads();
  const int cpg = p.C / p.groups;
  for (int g = threadIdx.x; g < p.groups; g += 256) {
    double s1 = 0.0, s2 = 0.0;
    for (int c = g * cpg; c < (g + 1) * cpg; ++c) {
      s1 += (double)p.gamma[c] * tot[c][0];
      s2 += (double)p.gamma[c] * tot[c][1];
    }
    gsum[(b * p.groups + g) * 2] = (float)s1;
    gsum[(b * p.groups + g) * 2 + 1] = (float)s2;
  }
}

__global__ __launch_bounds__(256) void gn_bwd_param_kernel(SdmiGroupNormBwdArgs p) {
  __shared__ float red[4][64][2];
  const int c = blockIdx.x * 64 + (threadIdx.x & 63), kg = threadIdx.x >> 6;
  float sa = 0.f, sb = 0.f;
  if (c < p.C)
    for (int b = kg; b < p.B; b += 4) {
      const float* q = p.partial + (((long long)b * p.nsplit) * p.C + c) * 2;
      sa += q[0];
      sb += q[1];
    }
  red[kg][threadIdx.x & 63][0] = sa;
  red[kg][threadIdx.x & 63][1] = sb;
  __syncthreads();
  if (kg == 0 && c < p.C) {
    const int l = threadIdx.x;
    const float a = (red[0][l][0] + red[1][l][0]) + (red[2][l][0] + red[3][l][0]);
    const float b = (red[0][l][1] + red[1][l][1]) + (red[2][l][1] + red[3][l][1]);
    p.dbeta[c] = (p.accumulate ? p.dbeta[c] : 0.f) + a;
    p.dgamma[c] = (p.accumulate ? p.dgamma[c] : 0.f) + b;
  }
}

template <typename T>
__global__ __launch_bounds__(256) void gn_bwd_apply_kernel(SdmiGroupNormBwdArgs p,
                                                           const float* gsum, int rows_per) {
  constexpr int VEC = Elem<T>::VEC;
  const int b = blockIdx.y;
  const int CV = p.C / VEC, CVp = next_pow2(CV);
  const int R = 256 / CVp;
  const int cv = threadIdx.x % CVp, r0 = threadIdx.x / CVp;
  if (cv >= CV) return;
  const int row_begin = blockIdx.x * rows_per;
  int row_end = row_begin + rows_per;
  if (row_end > p.HW) row_end = p.HW;
  const int cpg = p.C / p.groups;
  const float inv_n = 1.f / ((float)p.HW * (float)cpg);
  float mu[VEC], rs[VEC], ga[VEC], be[VEC], s1[VEC], s2[VEC];
#pragma unroll
  for (int j = 0; j < VEC; ++j) {
    const int c = cv * VEC + j, g = c / cpg;
    mu[j] = p.stats[(b * p.groups + g) * 2];
    rs[j] = p.stats[(b * p.groups + g) * 2 + 1];
    ga[j] = p.gamma[c];
    be[j] = p.beta[c];
    s1[j] = gsum[(b * p.groups + g) * 2] * inv_n;
    s2[j] = gsum[(b * p.groups + g) * 2 + 1] * inv_n;
  }
  const long long base = (long long)b * p.HW * p.C + cv * VEC;
  const T* xb = (const T*)p.x + base;
  const T* db = (const T*)p.dy + base;
  const T* rb = p.residual ? (const T*)p.residual + base : nullptr;
  T* dxb = (T*)p.dx + base;
  T* drb = p.dresidual ? (T*)p.dresidual + base : nullptr;
  for (int row = row_begin + r0; row < row_end; row += R) {
    const long long o = (long long)row * p.C;
    float x[VEC], dy[VEC], rr[VEC], dx[VEC], dz[VEC];
    unpack16<T>(*reinterpret_cast<const uint4*>(xb + o), x);
    unpack16<T>(*reinterpret_cast<const uint4*>(db + o), dy);
    if (rb) unpack16<T>(*reinterpret_cast<const uint4*>(rb + o), rr);
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      const float xh = (x[j] - mu[j]) * rs[j];
      const float z = xh * ga[j] + be[j] + (rb ? rr[j] : 0.f);
      dz[j] = dy[j] * act_grad(z, p.act);
      dx[j] = rs[j] * (ga[j] * dz[j] - (s1[j] + xh * s2[j]));
    }
    *reinterpret_cast<uint4*>(dxb + o) = pack16<T>(dx);
    if (drb) *reinterpret_cast<uint4*>(drb + o) = pack16<T>(dz);
  }
}

// ------------------------------------------------------------------------------------------
// LayerNorm backward: one wave per row for dx; per-workgroup channel partials for dgamma/dbeta.
template <typename T>
__global__ __launch_bounds__(256) void ln_bwd_kernel(SdmiLayerNormBwdArgs p, int rows_per_blk) {
  __shared__ float red[4][1024][2];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int row0 = blockIdx.x * rows_per_blk;
  int row1 = row0 + rows_per_blk;
  if (row1 > p.rows) row1 = p.rows;
  float ga[16], dg[16], dbt[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int c = i * 64 + lane;
    ga[i] = c < p.C ? p.gamma[c] : 0.f;
    dg[i] = dbt[i] = 0.f;
  }
  const float invC = 1.f / (float)p.C;
  for (int row = row0 + wave; row < row1; row += 4) {
    const T* x = (const T*)p.x + (long long)row * p.C;
    const T* dy = (const T*)p.dy + (long long)row * p.C;
    T* dx = (T*)p.dx + (long long)row * p.C;
    const float mean = p.stats[row * 2], rstd = p.stats[row * 2 + 1];
    float xh[16], dxh[16];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int c = i * 64 + lane;
      if (c < p.C) {
        const float d = Elem<T>::ld(dy + c);
        xh[i] = (Elem<T>::ld(x + c) - mean) * rstd;
        dxh[i] = d * ga[i];
        dg[i] += d * xh[i];
        dbt[i] += d;
        s1 += dxh[i];
        s2 += dxh[i] * xh[i];
      } else {
        xh[i] = dxh[i] = 0.f;
      }
    }
    s1 = wave_sum(s1) * invC;
    s2 = wave_sum(s2) * invC;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int c = i * 64 + lane;
      if (c < p.C) Elem<T>::st(dx + c, rstd * (dxh[i] - s1 - xh[i] * s2));
    }
  }
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int c = i * 64 + lane;
    if (c < p.C) { red[wave][c][0] = dg[i]; red[wave][c][1] = dbt[i]; }
  }
  __syncthreads();
  for (int c = threadIdx.x; c < p.C; c += 256) {
    float* q = p.partial + ((long long)blockIdx.x * p.C + c) * 2;
    q[0] = (red[0][c][0] + red[1][c][0]) + (red[2][c][0] + red[3][c][0]);
    q[1] = (red[0][c][1] + red[1][c][1]) + (red[2][c][1] + red[3][c][1]);
  }
}
__global__ __launch_bounds__(256) void ln_bwd_param_kernel(SdmiLayerNormBwdArgs p) {
  __shared__ float red[4][64][2];
  const int c = blockIdx.x * 64 + (threadIdx.x & 63), kg = threadIdx.x >> 6;
  float sg = 0.f, sb = 0.f;
  if (c < p.C)
    for (int k = kg; k < p.nblk; k += 4) {
      const float* q = p.partial + ((long long)k * p.C + c) * 2;
      sg += q[0];
      sb += q[1];
    }
  red[kg][threadIdx.x & 63][0] = sg;
  red[kg][threadIdx.x & 63][1] = sb;
  __syncthreads();
  if (kg == 0 && c < p.C) {
    const int l = threadIdx.x;
    const float g = (red[0][l][0] + red[1][l][0]) + (red[2][l][0] + red[3][l][0]);
    const float b = (red[0][l][1] + red[1][l][1]) + (red[2][l][1] + red[3][l][1]);
    p.dgamma[c] = (p.accumulate ? p.dgamma[c] : 0.f) + g;
    p.dbeta[c] = (p.accumulate ? p.dbeta[c] : 0.f) + b;
  }
}

}  // namespace

extern "C" int sdmi_groupnorm_bwd(const SdmiGroupNormBwdArgs* a, void* stream) {
  SDMI_REQUIRE(a && a->x && a->dy && a->dx && a->gamma && a->beta && a->stats && a->dgamma &&
                   a->dbeta && a->partial, "null pointer");
  const int vec = a->dtype == SDMI_BF16 ? 8 : 4;
  SDMI_REQUIRE(a->C % vec == 0 && a->C <= 1024, "C must be a vector multiple <= 1024");
  SDMI_REQUIRE(a->groups > 0 && a->C % a->groups == 0 && a->nsplit >= 1, "bad groups/nsplit");
  hipStream_t st = (hipStream_t)stream;
  float* gsum = a->partial + (long long)a->B * a->nsplit * a->C * 2;
  dim3 g1(a->nsplit, a->B);
  const long long row_bytes = (long long)a->C * (a->dtype == SDMI_BF16 ? 2 : 4);
  int rows_per = (int)((32768 + row_bytes - 1) / row_bytes);
  if (rows_per < 4) rows_per = 4;
  if (rows_per > a->HW) rows_per = a->HW;
  dim3 g3((a->HW + rows_per - 1) / rows_per, a->B);
  if (a->dtype == SDMI_BF16)
    hipLaunchKernelGGL(gn_bwd_stats_kernel<bf16_t>, g1, dim3(256), 0, st, *a);
  else
    hipLaunchKernelGGL(gn_bwd_stats_kernel<float>, g1, dim3(256), 0, st, *a);
  hipLaunchKernelGGL(gn_bwd_reduce_kernel, dim3(a->B), dim3(256), 0, st, *a, gsum);
  hipLaunchKernelGGL(gn_bwd_param_kernel, dim3((a->C + 63) / 64), dim3(256), 0, st, *a);
  if (a->dtype == SDMI_BF16)
    hipLaunchKernelGGL(gn_bwd_apply_kernel<bf16_t>, g3, dim3(256), 0, st, *a, gsum, rows_per);
  else
    hipLaunchKernelGGL(gn_bwd_apply_kernel<float>, g3, dim3(256), 0, st, *a, gsum, rows_per);
  return sdmi_check_launch("groupnorm_bwd");
}

extern "C" int sdmi_layernorm_bwd(const SdmiLayerNormBwdArgs* a, void* stream) {
  SDMI_REQUIRE(a && a->x && a->dy && a->dx && a->gamma && a->stats && a->dgamma && a->dbeta &&
                   a->partial, "null pointer");
  SDMI_REQUIRE(a->C > 0 && a->C <= 1024 && a->nblk >= 1, "C <= 1024");
  hipStream_t st = (hipStream_t)stream;
  const int rows_per_blk = (a->rows + a->nblk - 1) / a->nblk;
  if (a->dtype == SDMI_BF16)
    hipLaunchKernelGGL(ln_bwd_kernel<bf16_t>, dim3(a->nblk), dim3(256), 0, st, *a, rows_per_blk);
  else
    hipLaunchKernelGGL(ln_bwd_kernel<float>, dim3(a->nblk), dim3(256), 0, st, *a, rows_per_blk);
  hipLaunchKernelGGL(ln_bwd_param_kernel, dim3((a->C + 63) / 64), dim3(256), 0, st, *a);
  return sdmi_check_launch("layernorm_bwd");
}
