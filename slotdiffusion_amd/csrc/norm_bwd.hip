// Backward of GroupNorm(+act,+residual) on NHWC and of LayerNorm (include/sdmi.h).
//
// GroupNorm:  z = xhat*gamma + beta (+res),  y = act(z),  xhat = (x - mean_g) * rstd_g
//   dz = dy * act'(z);  dres = dz;  dbeta_c = sum dz;  dgamma_c = sum dz*xhat
//   dx = rstd_g * ( gamma_c*dz - ( S1_g + xhat * S2_g ) / n ),
//        S1_g = sum_{c in g} gamma_c * A_bc,  S2_g = sum_{c in g} gamma_c * B_bc,
//        A_bc = sum_hw dz,  B_bc = sum_hw dz*xhat      (per image b, channel c)
// so one streaming pass produces the per-(image, split, channel) sums; the second streaming pass
// (dx, dres) first folds them into its image's group terms in LDS, and an independent small kernel
// folds them into dgamma / dbeta.  Same fixed-channel thread organisation as the forward; fp64
// combines, no atomics.
#include "common.h"
#include "gn_geom.h"
#include "gn_dev.h"
#include <type_traits>

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
    const bool drop = p.drop_p > 0.f;
    const unsigned long long dseed = drop ? sdmi_drop_seed(p.drop_seed, p.drop_seed_dev) : 0ULL;
    const unsigned thr16 = (unsigned)(p.drop_p * 65536.f);
    const float dinv = 1.f / (1.f - p.drop_p);
    for (int row = row_begin + r0; row < row_end; row += R) {
      const long long o = (long long)row * p.C;
      float x[VEC], dy[VEC], rr[VEC];
      unpack16<T>(*reinterpret_cast<const uint4*>(xb + o), x);
      unpack16<T>(*reinterpret_cast<const uint4*>(db + o), dy);
      if (drop) sdmi_drop_apply<VEC>(dy, dseed, (base + o) / VEC, thr16, dinv);
      if (rb) unpack16<T>(*reinterpret_cast<const uint4*>(rb + o), rr);
#pragma unroll
      for (int j = 0; j < VEC; ++j) {
        const float xh = (x[j] - mu[j]) * rs[j];
        const float z = xh * ga[j] + be[j] + (rb ? rr[j] : 0.f);
        const float dz = dy[j] * act_grad<sizeof(T) == 2>(z, p.act);
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

// Small images (the [HW][C/S] slab of x and dy fits one workgroup's registers, nsplit == 1): both
// streaming passes in one launch.  Same geometry as the forward single-pass kernel (gn_geom.h):
// THREADS-wide workgroups, per-thread sums folded with xor-butterflies across the lanes that share a
// vector column, one LDS entry per wave.
template <typename T, int THREADS, int NV>
__global__ __launch_bounds__(THREADS) void gn_bwd_fused_kernel(SdmiGroupNormBwdArgs p) {
  constexpr int VEC = Elem<T>::VEC;
  extern __shared__ __attribute__((aligned(16))) float gnb_smem[];
  const int b = blockIdx.x;
  const int S = gridDim.y, sidx = blockIdx.y;            // channel chunk (whole groups) of this workgroup
  const int CV = p.C / VEC / S, CVp = next_pow2(CV);
  const int c_lo = sidx * CV * VEC;
  const int R = THREADS / CVp;
  const int RR = CVp < 64 ? THREADS / 64 : R;
  float (*part)[VEC][2] = reinterpret_cast<float (*)[VEC][2]>(gnb_smem);
  float (*gsum)[2] = reinterpret_cast<float (*)[2]>(gnb_smem + RR * CVp * VEC * 2);
  const int cv = threadIdx.x % CVp, r0 = threadIdx.x / CVp;
  const bool act_c = cv < CV;
  const int cpg = p.C / p.groups;
  const int gs = p.groups / S;
  float mu[VEC], rs[VEC], ga[VEC], be[VEC];
#pragma unroll
  for (int j = 0; j < VEC; ++j) {
    const int c = c_lo + (act_c ? cv : 0) * VEC + j, g = c / cpg;
    mu[j] = p.stats[(b * p.groups + g) * 2];
    rs[j] = p.stats[(b * p.groups + g) * 2 + 1];
    ga[j] = p.gamma[c];
    be[j] = p.beta[c];
  }
  const long long base = (long long)b * p.HW * p.C + c_lo + (act_c ? cv : 0) * VEC;
  const T* xb = (const T*)p.x + base;
  const T* db = (const T*)p.dy + base;
  const T* rb = p.residual ? (const T*)p.residual + base : nullptr;
  uint4 xr[NV], dr[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int row = r0 + i * R;
    if (act_c && row < p.HW) {
      xr[i] = *reinterpret_cast<const uint4*>(xb + (long long)row * p.C);
      dr[i] = *reinterpret_cast<const uint4*>(db + (long long)row * p.C);
    }
  }
  float A[VEC], Bv[VEC];
#pragma unroll
  for (int j = 0; j < VEC; ++j) A[j] = Bv[j] = 0.f;
  const bool drop = p.drop_p > 0.f;
  const unsigned long long dseed = drop ? sdmi_drop_seed(p.drop_seed, p.drop_seed_dev) : 0ULL;
  const unsigned thr16 = (unsigned)(p.drop_p * 65536.f);
  const float dinv = 1.f / (1.f - p.drop_p);
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int row = r0 + i * R;
    if (act_c && row < p.HW) {
      float x[VEC], dy[VEC], rr[VEC];
      unpack16<T>(xr[i], x);
      unpack16<T>(dr[i], dy);
      if (drop) sdmi_drop_apply<VEC>(dy, dseed, (base + (long long)row * p.C) / VEC, thr16, dinv);
      if (rb) unpack16<T>(*reinterpret_cast<const uint4*>(rb + (long long)row * p.C), rr);
#pragma unroll
      for (int j = 0; j < VEC; ++j) {
        const float xh = (x[j] - mu[j]) * rs[j];
        const float z = xh * ga[j] + be[j] + (rb ? rr[j] : 0.f);
        const float dz = dy[j] * act_grad<sizeof(T) == 2>(z, p.act);
        dy[j] = dz;                        // keep dz for the second half
        A[j] += dz;
        Bv[j] += dz * xh;
      }
      dr[i] = pack16<T>(dy);               // (bf16: dz rounded once more, as dresidual stores it)
    }
  }
  for (int off = CVp; off < 64; off <<= 1) {
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      A[j] += __shfl_xor(A[j], off, 64);
      Bv[j] += __shfl_xor(Bv[j], off, 64);
    }
  }
  if (CVp >= 64 || (int)(threadIdx.x & 63) < CVp) {
    const int slot = CVp < 64 ? (threadIdx.x >> 6) * CVp + cv : threadIdx.x;
#pragma unroll
    for (int j = 0; j < VEC; ++j) { part[slot][j][0] = A[j]; part[slot][j][1] = Bv[j]; }
  }
  __syncthreads();
  if (r0 == 0 && act_c) {                  // channel totals -> partial[b][0][c] (for dgamma/dbeta)
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      double sa = 0.0, sb = 0.0;
      for (int r = 0; r < RR; ++r) { sa += part[r * CVp + cv][j][0]; sb += part[r * CVp + cv][j][1]; }
      part[cv][j][0] = (float)sa;          // row 0 of the column: no other thread reads rows r>0 of it now
      part[cv][j][1] = (float)sb;
      float* q = p.partial + (((long long)b * p.C) + c_lo + cv * VEC + j) * 2;
      q[0] = (float)sa;
      q[1] = (float)sb;
    }
  }
  __syncthreads();
  for (int g = threadIdx.x; g < gs; g += THREADS) {      // groups / channels local to the chunk
    double s1 = 0.0, s2 = 0.0;
    for (int c = g * cpg; c < (g + 1) * cpg; ++c) {
      s1 += (double)p.gamma[c_lo + c] * part[c / VEC][c % VEC][0];
      s2 += (double)p.gamma[c_lo + c] * part[c / VEC][c % VEC][1];
    }
    gsum[g][0] = (float)s1;
    gsum[g][1] = (float)s2;
  }
  __syncthreads();
  const float inv_n = 1.f / ((float)p.HW * (float)cpg);
  float sd[VEC];                        // per-channel sums of dx over this thread's rows (dxsum)
#pragma unroll
  for (int j = 0; j < VEC; ++j) sd[j] = 0.f;
  if (act_c) {
  float s1[VEC], s2[VEC];
#pragma unroll
  for (int j = 0; j < VEC; ++j) {
    const int g = (cv * VEC + j) / cpg;
    s1[j] = gsum[g][0] * inv_n;
    s2[j] = gsum[g][1] * inv_n;
  }
  T* dxb = (T*)p.dx + base;
  T* drb = p.dresidual ? (T*)p.dresidual + base : nullptr;
  const T* e0 = p.dextra0 ? (const T*)p.dextra0 + base : nullptr;
  const T* e1 = p.dextra1 ? (const T*)p.dextra1 + base : nullptr;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int row = r0 + i * R;
    if (row < p.HW) {
      const long long o = (long long)row * p.C;
      float x[VEC], dz[VEC], dx[VEC];
      unpack16<T>(xr[i], x);
      unpack16<T>(dr[i], dz);
#pragma unroll
      for (int j = 0; j < VEC; ++j) {
        const float xh = (x[j] - mu[j]) * rs[j];
        dx[j] = rs[j] * (ga[j] * dz[j] - (s1[j] + xh * s2[j]));
      }
      if (e0) {                           // gradients of x's other consumers, summed here
        float ex[VEC];
        unpack16<T>(*reinterpret_cast<const uint4*>(e0 + o), ex);
#pragma unroll
        for (int j = 0; j < VEC; ++j) dx[j] += ex[j];
      }
      if (e1) {
        float ex[VEC];
        unpack16<T>(*reinterpret_cast<const uint4*>(e1 + o), ex);
#pragma unroll
        for (int j = 0; j < VEC; ++j) dx[j] += ex[j];
      }
#pragma unroll
      for (int j = 0; j < VEC; ++j) sd[j] += dx[j];
      *reinterpret_cast<uint4*>(dxb + o) = pack16<T>(dx);
      if (drb) *reinterpret_cast<uint4*>(drb + o) = dr[i];
    }
  }
  }
  if (p.dxsum) {                         // (uniform: every thread of the workgroup takes this path)
    for (int off = CVp; off < 64; off <<= 1) {
#pragma unroll
      for (int j = 0; j < VEC; ++j) sd[j] += __shfl_xor(sd[j], off, 64);
    }
    __syncthreads();                     // `part` is free again (group sums were read above)
    if (CVp >= 64 || (int)(threadIdx.x & 63) < CVp) {
      const int slot = CVp < 64 ? (threadIdx.x >> 6) * CVp + cv : threadIdx.x;
#pragma unroll
      for (int j = 0; j < VEC; ++j) part[slot][j][0] = sd[j];
    }
    __syncthreads();
    if (r0 == 0 && act_c) {
#pragma unroll
      for (int j = 0; j < VEC; ++j) {
        float t = 0.f;
        for (int r = 0; r < RR; ++r) t += part[r * CVp + cv][j][0];
        p.dxsum[(long long)b * p.ld_dxsum + c_lo + cv * VEC + j] = t;
      }
    }
  }
}

// Single-pass backward, second form (the twin of gn_fused2_kernel, norm.hip): the group sums
// (sum gamma dz, sum gamma dz xhat) are folded per lane into two group slots before any cross-lane step and
// travel through two barriers; the per-channel totals (dgamma / dbeta partials) are reduced with DPP row
// rotations, folded by all threads in parallel and stored OFF the path to the second pass (the first form
// walked them with CV threads, then fetched gamma again behind a barrier); statistics and gamma / beta
// arrive as 2 + 2 vector loads instead of 4 * VEC scalar ones; the activation derivative is hoisted.
template <typename T, int THREADS, int NV>
__global__ __launch_bounds__(THREADS) void gn_bwd_fused2_kernel(SdmiGroupNormBwdArgs p) {
  constexpr int VEC = Elem<T>::VEC;
  constexpr int VSH = VEC == 8 ? 3 : 2;
  extern __shared__ __attribute__((aligned(16))) float gnb_smem[];
  const int b = blockIdx.x;
  const int S = gridDim.y, sidx = blockIdx.y;
  const int CV = p.C / VEC / S;
  const int csh = CV <= 1 ? 0 : 32 - __builtin_clz(CV - 1);
  const int CVp = 1 << csh;
  const int c_lo = sidx * CV * VEC;
  const int R = THREADS >> csh;
  const int RR = CVp < 64 ? THREADS / 64 : R;
  float4* part = reinterpret_cast<float4*>(gnb_smem);                          // [RR][CVp] group slots
  double* colsum = reinterpret_cast<double*>(gnb_smem + RR * CVp * 4);         // [CVp][4]
  float (*chpart)[VEC][2] = reinterpret_cast<float (*)[VEC][2]>(gnb_smem + RR * CVp * 4 + CVp * 8);   // [RR*CVp]
  const int tid = threadIdx.x;
  const int cv = tid & (CVp - 1), r0 = tid >> csh;
  const bool act_c = cv < CV;
  const int cvc = act_c ? cv : 0;
  const int cpg = p.C / p.groups;
  const int cl0 = cvc * VEC;
  const int g0 = cl0 / cpg;
  const int bnd = (g0 + 1) * cpg - cl0;                   // channels j >= bnd: group g0 + 1 (slot 1)
  const long long base = (long long)b * p.HW * p.C + c_lo + cl0;
  const T* xb = (const T*)p.x + base;
  const T* db = (const T*)p.dy + base;
  const T* rb = p.residual ? (const T*)p.residual + base : nullptr;
  uint4 xr[NV], dr[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int row = r0 + i * R;
    if (act_c && row < p.HW) {
      xr[i] = *reinterpret_cast<const uint4*>(xb + (long long)row * p.C);
      dr[i] = *reinterpret_cast<const uint4*>(db + (long long)row * p.C);
    }
  }
  const int gg = b * p.groups + sidx * (p.groups / S) + g0;
  const float2 st0 = *reinterpret_cast<const float2*>(p.stats + gg * 2);
  const float2 st1 = bnd < VEC ? *reinterpret_cast<const float2*>(p.stats + gg * 2 + 2) : st0;
  float ga[VEC], be[VEC];
  load_fvec<VEC>(p.gamma + c_lo + cl0, ga);
  load_fvec<VEC>(p.beta + c_lo + cl0, be);
  // per-channel statistics / group sums are selects between the two slots at their uses (no VEC-wide arrays:
  // the 1024-thread instantiation has 128 VGPRs)
#define GNB_MU(j) ((j) < bnd ? st0.x : st1.x)
#define GNB_RS(j) ((j) < bnd ? st0.y : st1.y)
  float A[VEC], Bv[VEC];
#pragma unroll
  for (int j = 0; j < VEC; ++j) A[j] = Bv[j] = 0.f;
  const bool drop = p.drop_p > 0.f;
  const unsigned long long dseed = drop ? sdmi_drop_seed(p.drop_seed, p.drop_seed_dev) : 0ULL;
  const unsigned thr16 = (unsigned)(p.drop_p * 65536.f);
  const float dinv = 1.f / (1.f - p.drop_p);
  auto first = [&](auto actc) __attribute__((always_inline)) {
    constexpr int ACT = decltype(actc)::value;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int row = r0 + i * R;
      if (act_c && row < p.HW) {
        float x[VEC], dy[VEC], rr[VEC];
        unpack16<T>(xr[i], x);
        unpack16<T>(dr[i], dy);
        if (drop) sdmi_drop_apply<VEC>(dy, dseed, (base + (long long)row * p.C) / VEC, thr16, dinv);
        if (rb) unpack16<T>(*reinterpret_cast<const uint4*>(rb + (long long)row * p.C), rr);
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
          const float xh = (x[j] - GNB_MU(j)) * GNB_RS(j);
          const float z = xh * ga[j] + be[j] + (rb ? rr[j] : 0.f);
          const float dz = dy[j] * act_grad<sizeof(T) == 2>(z, ACT);
          dy[j] = dz;                        // keep dz for the second half
          A[j] += dz;
          Bv[j] += dz * xh;
        }
        dr[i] = pack16<T>(dy);               // (bf16: dz rounded once more, as dresidual stores it)
      }
    }
  };
  if (p.act == SDMI_ACT_SILU) first(std::integral_constant<int, SDMI_ACT_SILU>{});
  else if (p.act == SDMI_ACT_RELU) first(std::integral_constant<int, SDMI_ACT_RELU>{});
  else if (p.act == SDMI_ACT_GELU) first(std::integral_constant<int, SDMI_ACT_GELU>{});
  else first(std::integral_constant<int, SDMI_ACT_NONE>{});
  // group slots: sum over the lane's channels of gamma * (dz | dz xhat)
  float a0 = 0.f, q0 = 0.f, a1 = 0.f, q1 = 0.f;
#pragma unroll
  for (int j = 0; j < VEC; ++j) {
    const bool lo = j < bnd;
    const float ta = ga[j] * A[j], tb = ga[j] * Bv[j];
    a0 += lo ? ta : 0.f;
    q0 += lo ? tb : 0.f;
    a1 += lo ? 0.f : ta;
    q1 += lo ? 0.f : tb;
  }
  if (CVp < 64) {
    a0 = col_allreduce(a0, CVp);
    q0 = col_allreduce(q0, CVp);
    a1 = col_allreduce(a1, CVp);
    q1 = col_allreduce(q1, CVp);
#pragma unroll
    for (int j = 0; j < VEC; ++j) { A[j] = col_allreduce(A[j], CVp); Bv[j] = col_allreduce(Bv[j], CVp); }
    if ((tid & 63) < CVp) {
      const int e = (tid >> 6) * CVp + cv;
      part[e] = make_float4(a0, q0, a1, q1);
#pragma unroll
      for (int j = 0; j < VEC; ++j) { chpart[e][j][0] = A[j]; chpart[e][j][1] = Bv[j]; }
    }
  } else {
    part[tid] = make_float4(a0, q0, a1, q1);
#pragma unroll
    for (int j = 0; j < VEC; ++j) { chpart[tid][j][0] = A[j]; chpart[tid][j][1] = Bv[j]; }
  }
  __syncthreads();
  if (tid < CVp) {
    double d0 = 0.0, d1 = 0.0, d2 = 0.0, d3 = 0.0;
    for (int r = 0; r < RR; ++r) {
      const float4 v = part[r * CVp + tid];
      d0 += (double)v.x; d1 += (double)v.y; d2 += (double)v.z; d3 += (double)v.w;
    }
    colsum[tid * 4 + 0] = d0; colsum[tid * 4 + 1] = d1; colsum[tid * 4 + 2] = d2; colsum[tid * 4 + 3] = d3;
  }
  // channel totals -> partial[b][c] (dgamma / dbeta): every thread takes (channel, which) pairs, starting from
  // the last waves (the first one is folding the group columns)
  for (int u = THREADS - 1 - tid; u < CVp * VEC * 2; u += THREADS) {
    const int ccv = u / (VEC * 2), j = (u >> 1) & (VEC - 1), which = u & 1;
    if (ccv < CV) {
      double sa = 0.0;
      for (int r = 0; r < RR; ++r) sa += (double)chpart[r * CVp + ccv][j][which];
      p.partial[(((long long)b * p.C) + c_lo + ccv * VEC + j) * 2 + which] = (float)sa;
    }
  }
  __syncthreads();
  const float inv_n = 1.f / ((float)p.HW * (float)cpg);
  float s1_[2], s2_[2];
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    if (k == 1 && bnd >= VEC) { s1_[1] = s1_[0]; s2_[1] = s2_[0]; break; }
    const int f = (g0 + k) * cpg, l = f + cpg - 1;
    const int ca = f >> VSH, cb = l >> VSH;
    double sm = 0.0, sq = 0.0;
    for (int c = ca; c <= cb; ++c) {
      const int o = c * 4 + (((c << VSH) >= f) ? 0 : 2);
      sm += colsum[o];
      sq += colsum[o + 1];
    }
    s1_[k] = (float)sm * inv_n;
    s2_[k] = (float)sq * inv_n;
  }
  float sd[VEC];                        // per-channel sums of dx over this thread's rows (dxsum)
#pragma unroll
  for (int j = 0; j < VEC; ++j) sd[j] = 0.f;
  if (act_c) {
    T* dxb = (T*)p.dx + base;
    T* drb = p.dresidual ? (T*)p.dresidual + base : nullptr;
    const T* e0 = p.dextra0 ? (const T*)p.dextra0 + base : nullptr;
    const T* e1 = p.dextra1 ? (const T*)p.dextra1 + base : nullptr;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int row = r0 + i * R;
      if (row < p.HW) {
        const long long o = (long long)row * p.C;
        float x[VEC], dz[VEC], dx[VEC];
        unpack16<T>(xr[i], x);
        unpack16<T>(dr[i], dz);
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
          const float rsj = GNB_RS(j);
          const float xh = (x[j] - GNB_MU(j)) * rsj;
          dx[j] = rsj * (ga[j] * dz[j] - ((j < bnd ? s1_[0] : s1_[1]) + xh * (j < bnd ? s2_[0] : s2_[1])));
        }
        if (e0) {                           // gradients of x's other consumers, summed here
          float ex[VEC];
          unpack16<T>(*reinterpret_cast<const uint4*>(e0 + o), ex);
#pragma unroll
          for (int j = 0; j < VEC; ++j) dx[j] += ex[j];
        }
        if (e1) {
          float ex[VEC];
          unpack16<T>(*reinterpret_cast<const uint4*>(e1 + o), ex);
#pragma unroll
          for (int j = 0; j < VEC; ++j) dx[j] += ex[j];
        }
#pragma unroll
        for (int j = 0; j < VEC; ++j) sd[j] += dx[j];
        *reinterpret_cast<uint4*>(dxb + o) = pack16<T>(dx);
        if (drb) *reinterpret_cast<uint4*>(drb + o) = dr[i];
      }
    }
  }
#undef GNB_MU
#undef GNB_RS
  if (p.dxsum) {                         // (uniform: every thread of the workgroup takes this path)
    if (CVp < 64) {
#pragma unroll
      for (int j = 0; j < VEC; ++j) sd[j] = col_allreduce(sd[j], CVp);
    }
    __syncthreads();                     // the channel totals above have been read
    if (CVp >= 64 || (tid & 63) < CVp) {
      const int e = CVp < 64 ? (tid >> 6) * CVp + cv : tid;
#pragma unroll
      for (int j = 0; j < VEC; ++j) chpart[e][j][0] = sd[j];
    }
    __syncthreads();
    for (int u = tid; u < CVp * VEC; u += THREADS) {
      const int ccv = u / VEC, j = u & (VEC - 1);
      if (ccv < CV) {
        float t = 0.f;
        for (int r = 0; r < RR; ++r) t += chpart[r * CVp + ccv][j][0];
        p.dxsum[(long long)b * p.ld_dxsum + c_lo + ccv * VEC + j] = t;
      }
    }
  }
}

// out0[c] (+)= sum_e partial[e][c][0], out1[c] (+)= sum_e partial[e][c][1] over nblk entries:
// 16 channels x 16 entry-lanes per workgroup (C/16 workgroups), 4 independent loads in flight per
// thread, fixed-order LDS fold.  Serves dgamma/dbeta of both GroupNorm (entries = image x split)
// and LayerNorm (entries = row blocks).
__global__ __launch_bounds__(256) void pair_colsum_kernel(const float* partial, int nblk, int C,
                                                          float* out0, float* out1,
                                                          int accumulate) {
  __shared__ float red[16][16][2];
  const int cl = threadIdx.x & 15, kg = threadIdx.x >> 4;
  const int c = blockIdx.x * 16 + cl;
  float s0 = 0.f, s1 = 0.f;
  if (c < C) {
    const float2* q = reinterpret_cast<const float2*>(partial) + c;
    int k = kg;
    for (; k + 48 < nblk; k += 64) {
      const float2 v0 = q[(long long)k * C], v1 = q[(long long)(k + 16) * C];
      const float2 v2 = q[(long long)(k + 32) * C], v3 = q[(long long)(k + 48) * C];
      s0 += (v0.x + v1.x) + (v2.x + v3.x);
      s1 += (v0.y + v1.y) + (v2.y + v3.y);
    }
    for (; k < nblk; k += 16) {
      const float2 v = q[(long long)k * C];
      s0 += v.x;
      s1 += v.y;
    }
  }
  red[kg][cl][0] = s0;
  red[kg][cl][1] = s1;
  __syncthreads();
  if (kg == 0 && c < C) {
    float a = 0.f, b = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) { a += red[k][cl][0]; b += red[k][cl][1]; }
    out0[c] = (accumulate ? out0[c] : 0.f) + a;
    out1[c] = (accumulate ? out1[c] : 0.f) + b;
  }
}

// Grouped form: blockIdx.y = item, the descriptors travel by value in the kernel arguments.
constexpr int CS_MAX = 64;
struct ColsumGroup { int n; SdmiColsumItem it[CS_MAX]; };
__global__ __launch_bounds__(256) void colsum_group_kernel(ColsumGroup g) {
  const SdmiColsumItem& q_ = g.it[blockIdx.y];
  const int C = q_.C, nblk = q_.nblk;
  if ((int)blockIdx.x * 16 >= C) return;
  __shared__ float red[16][16][2];
  const int cl = threadIdx.x & 15, kg = threadIdx.x >> 4;
  const int c = blockIdx.x * 16 + cl;
  float s0 = 0.f, s1 = 0.f;
  if (c < C) {
    const float2* q = reinterpret_cast<const float2*>(q_.partial) + c;
    int k = kg;
    for (; k + 48 < nblk; k += 64) {
      const float2 v0 = q[(long long)k * C], v1 = q[(long long)(k + 16) * C];
      const float2 v2 = q[(long long)(k + 32) * C], v3 = q[(long long)(k + 48) * C];
      s0 += (v0.x + v1.x) + (v2.x + v3.x);
      s1 += (v0.y + v1.y) + (v2.y + v3.y);
    }
    for (; k < nblk; k += 16) {
      const float2 v = q[(long long)k * C];
      s0 += v.x;
      s1 += v.y;
    }
  }
  red[kg][cl][0] = s0;
  red[kg][cl][1] = s1;
  __syncthreads();
  if (kg == 0 && c < C) {
    float a = 0.f, b = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) { a += red[k][cl][0]; b += red[k][cl][1]; }
    q_.out0[c] += a;
    q_.out1[c] += b;
  }
}

template <typename T>
__global__ __launch_bounds__(256) void gn_bwd_apply_kernel(SdmiGroupNormBwdArgs p, int rows_per) {
  constexpr int VEC = Elem<T>::VEC;
  __shared__ float gsum[256][2];     // per-group terms of this image (groups <= 256)
  const int b = blockIdx.y;
  const int cpg = p.C / p.groups;
  {  // S1_g = sum_{c in g} gamma_c A_bc, S2_g likewise with B_bc: 8 lanes per group over the
     // group's cpg * nsplit partial entries (every workgroup of the image recomputes them)
    const int l8 = threadIdx.x & 7;
    const int ne = cpg * p.nsplit;
    for (int g = threadIdx.x >> 3; g < p.groups; g += 32) {
      double s1 = 0.0, s2 = 0.0;
      for (int e = l8; e < ne; e += 8) {
        const int k = e / cpg, c = g * cpg + (e - k * cpg);
        const float2 v = reinterpret_cast<const float2*>(p.partial)[((long long)b * p.nsplit + k) * p.C + c];
        const double ga = (double)p.gamma[c];
        s1 += ga * v.x;
        s2 += ga * v.y;
      }
      for (int off = 1; off < 8; off <<= 1) {
        s1 += __shfl_xor(s1, off, 64);
        s2 += __shfl_xor(s2, off, 64);
      }
      if (l8 == 0) { gsum[g][0] = (float)s1; gsum[g][1] = (float)s2; }
    }
  }
  __syncthreads();
  const int CV = p.C / VEC, CVp = next_pow2(CV);
  const int R = 256 / CVp;
  const int cv = threadIdx.x % CVp, r0 = threadIdx.x / CVp;
  if (cv >= CV) return;
  const int row_begin = blockIdx.x * rows_per;
  int row_end = row_begin + rows_per;
  if (row_end > p.HW) row_end = p.HW;
  const float inv_n = 1.f / ((float)p.HW * (float)cpg);
  float mu[VEC], rs[VEC], ga[VEC], be[VEC], s1[VEC], s2[VEC];
#pragma unroll
  for (int j = 0; j < VEC; ++j) {
    const int c = cv * VEC + j, g = c / cpg;
    mu[j] = p.stats[(b * p.groups + g) * 2];
    rs[j] = p.stats[(b * p.groups + g) * 2 + 1];
    ga[j] = p.gamma[c];
    be[j] = p.beta[c];
    s1[j] = gsum[g][0] * inv_n;
    s2[j] = gsum[g][1] * inv_n;
  }
  const long long base = (long long)b * p.HW * p.C + cv * VEC;
  const T* xb = (const T*)p.x + base;
  const T* db = (const T*)p.dy + base;
  const T* rb = p.residual ? (const T*)p.residual + base : nullptr;
  T* dxb = (T*)p.dx + base;
  T* drb = p.dresidual ? (T*)p.dresidual + base : nullptr;
  const T* e0 = p.dextra0 ? (const T*)p.dextra0 + base : nullptr;
  const T* e1 = p.dextra1 ? (const T*)p.dextra1 + base : nullptr;
  const bool drop = p.drop_p > 0.f;
  const unsigned long long dseed = drop ? sdmi_drop_seed(p.drop_seed, p.drop_seed_dev) : 0ULL;
  const unsigned thr16 = (unsigned)(p.drop_p * 65536.f);
  const float dinv = 1.f / (1.f - p.drop_p);
  for (int row = row_begin + r0; row < row_end; row += R) {
    const long long o = (long long)row * p.C;
    float x[VEC], dy[VEC], rr[VEC], dx[VEC], dz[VEC];
    unpack16<T>(*reinterpret_cast<const uint4*>(xb + o), x);
    unpack16<T>(*reinterpret_cast<const uint4*>(db + o), dy);
    if (drop) sdmi_drop_apply<VEC>(dy, dseed, (base + o) / VEC, thr16, dinv);
    if (rb) unpack16<T>(*reinterpret_cast<const uint4*>(rb + o), rr);
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      const float xh = (x[j] - mu[j]) * rs[j];
      const float z = xh * ga[j] + be[j] + (rb ? rr[j] : 0.f);
      dz[j] = dy[j] * act_grad<sizeof(T) == 2>(z, p.act);
      dx[j] = rs[j] * (ga[j] * dz[j] - (s1[j] + xh * s2[j]));
    }
    if (e0) {
      float ex[VEC];
      unpack16<T>(*reinterpret_cast<const uint4*>(e0 + o), ex);
#pragma unroll
      for (int j = 0; j < VEC; ++j) dx[j] += ex[j];
    }
    if (e1) {
      float ex[VEC];
      unpack16<T>(*reinterpret_cast<const uint4*>(e1 + o), ex);
#pragma unroll
      for (int j = 0; j < VEC; ++j) dx[j] += ex[j];
    }
    *reinterpret_cast<uint4*>(dxb + o) = pack16<T>(dx);
    if (drb) *reinterpret_cast<uint4*>(drb + o) = pack16<T>(dz);
  }
}

// ------------------------------------------------------------------------------------------
// LayerNorm backward.  A row is covered by LPR = pow2 >= C/VEC lanes (16-byte vectors, up to VPL
// vectors per lane when C/VEC > 64), so a wave handles 64/LPR rows at once and a workgroup
// 4*64/LPR; the two row sums are xor-butterflies inside the LPR lanes.  dgamma/dbeta accumulate
// per lane over the workgroup's rows and leave as per-workgroup channel partials.
template <typename T, int VPL>
__global__ __launch_bounds__(256) void ln_bwd_kernel(SdmiLayerNormBwdArgs p, int rows_per_blk,
                                                     int LPR) {
  constexpr int VEC = Elem<T>::VEC;
  __shared__ float red[4][1024][2];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int sub = lane & (LPR - 1), slot = lane / LPR, RW = 64 / LPR;
  const int CV = p.C / VEC;
  const int row0 = blockIdx.x * rows_per_blk;
  int row1 = row0 + rows_per_blk;
  if (row1 > p.rows) row1 = p.rows;
  float ga[VPL][VEC], dg[VPL][VEC], dbt[VPL][VEC];
  bool act[VPL];
#pragma unroll
  for (int i = 0; i < VPL; ++i) {
    const int cv = sub + i * LPR;
    act[i] = cv < CV;
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      ga[i][j] = act[i] ? p.gamma[cv * VEC + j] : 0.f;
      dg[i][j] = dbt[i][j] = 0.f;
    }
  }
  const float invC = 1.f / (float)p.C;
  // rows of the NEXT pass are fetched before the current one is worked on: a workgroup walks only a few passes
  // (rows / nblk / (4 RW)) and each was a full load -> shuffle -> store latency chain (11.4 us at 16384 x 256)
  uint4 xq[VPL], dq[VPL];
  float mean_n = 0.f, rstd_n = 1.f;
  auto fetch = [&](int rbase) __attribute__((always_inline)) {
    const int row = rbase + slot;
    const long long rr = (row < row1) ? row : row0;
    const T* x = (const T*)p.x + rr * p.C;
    const T* dy = (const T*)p.dy + rr * p.C;
    mean_n = p.stats[rr * 2];
    rstd_n = p.stats[rr * 2 + 1];
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
      const int cc = act[i] ? sub + i * LPR : 0;
      xq[i] = *reinterpret_cast<const uint4*>(x + cc * VEC);
      dq[i] = *reinterpret_cast<const uint4*>(dy + cc * VEC);
    }
  };
  if (row0 + wave * RW < row1) fetch(row0 + wave * RW);
  for (int rbase = row0 + wave * RW; rbase < row1; rbase += 4 * RW) {
    const int row = rbase + slot;
    const bool rok = row < row1;
    const long long ro = (long long)(rok ? row : row0) * p.C;
    float xh[VPL][VEC], dxh[VPL][VEC];
    float s1 = 0.f, s2 = 0.f;
    const float mean = mean_n, rstd = rstd_n;
    uint4 xc[VPL], dc[VPL];
#pragma unroll
    for (int i = 0; i < VPL; ++i) { xc[i] = xq[i]; dc[i] = dq[i]; }
    const T* ex = p.dextra ? (const T*)p.dextra + ro : nullptr;
    uint4 eq[VPL];                        // gradient of the row's residual branch: in flight under the sums
    if (ex) {
#pragma unroll
      for (int i = 0; i < VPL; ++i) eq[i] = *reinterpret_cast<const uint4*>(ex + (act[i] ? sub + i * LPR : 0) * VEC);
    }
    if (rbase + 4 * RW < row1) fetch(rbase + 4 * RW);
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
      float xv[VEC], dv[VEC];
      unpack16<T>(xc[i], xv);
      unpack16<T>(dc[i], dv);
#pragma unroll
      for (int j = 0; j < VEC; ++j) {
        const float d = (act[i] && rok) ? dv[j] : 0.f;
        xh[i][j] = (xv[j] - mean) * rstd;
        dxh[i][j] = d * ga[i][j];
        dg[i][j] += d * xh[i][j];
        dbt[i][j] += d;
        s1 += dxh[i][j];
        s2 += dxh[i][j] * xh[i][j];
      }
    }
    for (int off = 1; off < LPR; off <<= 1) {
      s1 += __shfl_xor(s1, off, 64);
      s2 += __shfl_xor(s2, off, 64);
    }
    s1 *= invC;
    s2 *= invC;
    if (rok) {
      T* dx = (T*)p.dx + ro;
#pragma unroll
      for (int i = 0; i < VPL; ++i) {
        if (!act[i]) continue;
        float o[VEC];
#pragma unroll
        for (int j = 0; j < VEC; ++j) o[j] = rstd * (dxh[i][j] - s1 - xh[i][j] * s2);
        if (ex) {                         // summed here
          float ev[VEC];
          unpack16<T>(eq[i], ev);
#pragma unroll
          for (int j = 0; j < VEC; ++j) o[j] += ev[j];
        }
        *reinterpret_cast<uint4*>(dx + (sub + i * LPR) * VEC) = pack16<T>(o);
      }
    }
  }
  // fold the wave's row slots, then the four waves
  for (int off = LPR; off < 64; off <<= 1) {
#pragma unroll
    for (int i = 0; i < VPL; ++i)
#pragma unroll
      for (int j = 0; j < VEC; ++j) {
        dg[i][j] += __shfl_xor(dg[i][j], off, 64);
        dbt[i][j] += __shfl_xor(dbt[i][j], off, 64);
      }
  }
  if (slot == 0) {
#pragma unroll
    for (int i = 0; i < VPL; ++i)
      if (act[i]) {
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
          const int c = (sub + i * LPR) * VEC + j;
          red[wave][c][0] = dg[i][j];
          red[wave][c][1] = dbt[i][j];
        }
      }
  }
  __syncthreads();
  for (int c = threadIdx.x; c < p.C; c += 256) {
    float* q = p.partial + ((long long)blockIdx.x * p.C + c) * 2;
    q[0] = (red[0][c][0] + red[1][c][0]) + (red[2][c][0] + red[3][c][0]);
    q[1] = (red[0][c][1] + red[1][c][1]) + (red[2][c][1] + red[3][c][1]);
  }
}
}  // namespace

extern "C" int sdmi_groupnorm_bwd_fused(const SdmiGroupNormBwdArgs* a, void*) {
  if (!a || a->C <= 0 || a->groups <= 0) return 0;
  const int vec = a->dtype == SDMI_BF16 ? 8 : 4;
  int nv_of_T[3] = {16, 16, a->dtype == SDMI_BF16 ? 4 : 8};
  return gn_pick(a->B, a->HW, a->C, a->groups, vec, nv_of_T).T ? 1 : 0;
}

extern "C" int sdmi_groupnorm_bwd_entries(const SdmiGroupNormBwdArgs* a, void*) {
  if (!a || a->C <= 0 || a->groups <= 0) return 0;
  const int vec = a->dtype == SDMI_BF16 ? 8 : 4;
  int nv_of_T[3] = {16, 16, a->dtype == SDMI_BF16 ? 4 : 8};
  const GnGeom gg = gn_pick(a->B, a->HW, a->C, a->groups, vec, nv_of_T);
  return gg.T ? a->B : a->B * a->nsplit;
}

extern "C" int sdmi_groupnorm_bwd(const SdmiGroupNormBwdArgs* a, void* stream) {
  SDMI_REQUIRE(a && a->x && a->dy && a->dx && a->gamma && a->beta && a->stats && a->dgamma &&
                   a->dbeta && a->partial, "null pointer");
  const int vec = a->dtype == SDMI_BF16 ? 8 : 4;
  SDMI_REQUIRE(a->C % vec == 0 && a->C <= 1024, "C must be a vector multiple <= 1024");
  SDMI_REQUIRE(a->groups > 0 && a->C % a->groups == 0 && a->nsplit >= 1, "bad groups/nsplit");
  hipStream_t st = (hipStream_t)stream;
  dim3 g1(a->nsplit, a->B);
  const long long row_bytes = (long long)a->C * (a->dtype == SDMI_BF16 ? 2 : 4);
  int rows_per = (int)((32768 + row_bytes - 1) / row_bytes);
  if (rows_per < 4) rows_per = 4;
  if (rows_per > a->HW) rows_per = a->HW;
  dim3 g3((a->HW + rows_per - 1) / rows_per, a->B);
  {
    // single-pass kernel (x and dy slabs in registers), geometry shared with the forward pass
    int nv_of_T[3] = {16, 16, a->dtype == SDMI_BF16 ? 4 : 8};   // 1024 threads: 128 VGPRs each
    const GnGeom gg = gn_pick(a->B, a->HW, a->C, a->groups, vec, nv_of_T);
    if (gg.T) {
      dim3 gf(a->B, gg.S);
      const int cv = a->C / vec / gg.S;
      int cvp = 1;
      while (cvp < cv) cvp <<= 1;
      const int RR = cvp < 64 ? gg.T / 64 : gg.T / cvp;
      const size_t smem = ((size_t)RR * cvp * vec * 2 + 2 * 256) * sizeof(float);
      constexpr int v2 = 1;
      const int cpg = a->C / a->groups;
      // (the 1024-thread and the 16-vector instantiations of the second form spill at their register budgets and
      // measured slower than the first form: 43 vs 36 us at [64][32^2][128])
      const bool two = v2 && (cpg * 2 == vec || cpg >= vec - 1) && (gg.T != 1024 || v2 >= 2) && gg.need <= 8;
      const size_t smem2 = (size_t)RR * cvp * (16 + vec * 8) + (size_t)cvp * 32;
#define GNB_GO(T_, TH, NV_)                                                                        \
  do {                                                                                             \
    if (two) {                                                                                     \
      SDMI_OPTIN_LDS((gn_bwd_fused2_kernel<T_, TH, NV_>), 96 * 1024, "groupnorm_bwd");             \
      hipLaunchKernelGGL((gn_bwd_fused2_kernel<T_, TH, NV_>), gf, dim3(TH), smem2, st, *a);        \
    } else {                                                                                       \
      SDMI_OPTIN_LDS((gn_bwd_fused_kernel<T_, TH, NV_>), 80 * 1024, "groupnorm_bwd");              \
      hipLaunchKernelGGL((gn_bwd_fused_kernel<T_, TH, NV_>), gf, dim3(TH), smem, st, *a);          \
    }                                                                                              \
  } while (0)
#define GNB_PICK(T_, TH)                                                                           \
  do {                                                                                             \
    if (gg.need <= 4) GNB_GO(T_, TH, 4); else if (gg.need <= 8) GNB_GO(T_, TH, 8); else GNB_GO(T_, TH, 16); \
  } while (0)
      if (a->dtype == SDMI_BF16) {
        if (gg.T == 1024) GNB_GO(bf16_t, 1024, 4); else if (gg.T == 512) GNB_PICK(bf16_t, 512); else GNB_PICK(bf16_t, 256);
      } else {
        if (gg.T == 1024) { if (gg.need <= 4) GNB_GO(float, 1024, 4); else GNB_GO(float, 1024, 8); }
        else if (gg.T == 512) GNB_PICK(float, 512); else GNB_PICK(float, 256);
      }
#undef GNB_PICK
#undef GNB_GO
      if (!a->defer_colsum)
        hipLaunchKernelGGL(pair_colsum_kernel, dim3((a->C + 15) / 16), dim3(256), 0, st, a->partial,
                           a->B, a->C, a->dbeta, a->dgamma, a->accumulate);
      return sdmi_check_launch("groupnorm_bwd (fused)");
    }
  }
  if (a->dtype == SDMI_BF16)
    hipLaunchKernelGGL(gn_bwd_stats_kernel<bf16_t>, g1, dim3(256), 0, st, *a);
  else
    hipLaunchKernelGGL(gn_bwd_stats_kernel<float>, g1, dim3(256), 0, st, *a);
  SDMI_REQUIRE(a->groups <= 256, "at most 256 groups");
  // dbeta = sum dz, dgamma = sum dz*xhat over all (image, split) partial entries
  if (!a->defer_colsum)
    hipLaunchKernelGGL(pair_colsum_kernel, dim3((a->C + 15) / 16), dim3(256), 0, st, a->partial,
                       a->B * a->nsplit, a->C, a->dbeta, a->dgamma, a->accumulate);
  if (a->dtype == SDMI_BF16)
    hipLaunchKernelGGL(gn_bwd_apply_kernel<bf16_t>, g3, dim3(256), 0, st, *a, rows_per);
  else
    hipLaunchKernelGGL(gn_bwd_apply_kernel<float>, g3, dim3(256), 0, st, *a, rows_per);
  return sdmi_check_launch("groupnorm_bwd");
}

extern "C" int sdmi_layernorm_bwd(const SdmiLayerNormBwdArgs* a, void* stream) {
  SDMI_REQUIRE(a && a->x && a->dy && a->dx && a->gamma && a->stats && a->dgamma && a->dbeta &&
                   a->partial, "null pointer");
  SDMI_REQUIRE(a->C > 0 && a->C <= 1024 && a->nblk >= 1, "C <= 1024");
  hipStream_t st = (hipStream_t)stream;
  const int rows_per_blk = (a->rows + a->nblk - 1) / a->nblk;
  const int vec = a->dtype == SDMI_BF16 ? 8 : 4;
  SDMI_REQUIRE(a->C % vec == 0, "C must be a multiple of the 16-byte vector width");
  const int cv = a->C / vec;
  int lpr = 1;
  while (lpr < cv && lpr < 64) lpr <<= 1;
  const int vpl = (cv + 63) / 64;
#define LN_GO(T, V) \
  hipLaunchKernelGGL((ln_bwd_kernel<T, V>), dim3(a->nblk), dim3(256), 0, st, *a, rows_per_blk, lpr)
  if (a->dtype == SDMI_BF16) {
    if (vpl <= 1) LN_GO(bf16_t, 1); else LN_GO(bf16_t, 2);
  } else {
    if (vpl <= 1) LN_GO(float, 1); else if (vpl == 2) LN_GO(float, 2); else LN_GO(float, 4);
  }
#undef LN_GO
  if (!a->defer_colsum)
    hipLaunchKernelGGL(pair_colsum_kernel, dim3((a->C + 15) / 16), dim3(256), 0, st, a->partial,
                       a->nblk, a->C, a->dgamma, a->dbeta, a->accumulate);
  return sdmi_check_launch("layernorm_bwd");
}

extern "C" int sdmi_colsum_group(const SdmiColsumGroupArgs* ga, void* stream) {
  SDMI_REQUIRE(ga && ga->items && ga->n >= 1 && ga->n <= CS_MAX, "1 .. 64 items");
  const SdmiColsumItem* it = (const SdmiColsumItem*)ga->items;
  ColsumGroup g;
  g.n = ga->n;
  int cmax = 0;
  for (int i = 0; i < ga->n; ++i) {
    SDMI_REQUIRE(it[i].partial && it[i].out0 && it[i].out1 && it[i].nblk >= 1 && it[i].C >= 1, "bad item");
    g.it[i] = it[i];
    if (it[i].C > cmax) cmax = it[i].C;
  }
  hipLaunchKernelGGL(colsum_group_kernel, dim3((cmax + 15) / 16, ga->n), dim3(256), 0, (hipStream_t)stream, g);
  return sdmi_check_launch("colsum_group");
}
