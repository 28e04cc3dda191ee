// GroupNorm(+activation) on NHWC, LayerNorm, row softmax -- HBM-bound wavefront-reduction kernels.
//
// GroupNorm thread organisation: a workgroup of 256 threads views image b as [HW][C/VEC] 16-byte
// vectors.  Thread t owns vector column cv = t % CVp (CVp = next power of two >= C/VEC) and walks
// rows r = t / CVp, + 256/CVp, ... so its channels -- hence gamma/beta/group -- never change and
// every wave reads whole 128-byte lines.  Statistics are accumulated per thread in fp32 over short
// strided runs, combined across threads and splits in fp64, deterministically (no atomics).
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

// Input vector column starting at channel c of image b: pointer to its row 0 and the row pitch
// (elements) -- one tensor [B][HW][C], or the in-place concatenation [x | x2] (sdmi.h: x2, C1).
template <typename T>
__device__ __forceinline__ const T* gn_src(const SdmiGroupNormArgs& p, int b, int c, int& pitch) {
  if (p.x2) {
    if (c >= p.C1) {
      pitch = p.C - p.C1;
      return (const T*)p.x2 + (long long)b * p.HW * pitch + (c - p.C1);
    }
    pitch = p.C1;
    return (const T*)p.x + (long long)b * p.HW * pitch + c;
  }
  pitch = p.C;
  return (const T*)p.x + (long long)b * p.HW * p.C + c;
}

template <typename T>
__global__ __launch_bounds__(256) void gn_stats_kernel(SdmiGroupNormArgs p) {
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
  float s[VEC], ss[VEC];
#pragma unroll
  for (int j = 0; j < VEC; ++j) s[j] = ss[j] = 0.f;
  if (cv < CV) {
    int xp;
    const T* xb = gn_src<T>(p, b, cv * VEC, xp);
    for (int row = row_begin + r0; row < row_end; row += R) {
      const uint4 v = *reinterpret_cast<const uint4*>(xb + (long long)row * xp);
      float f[VEC];
      unpack16<T>(v, f);
#pragma unroll
      for (int j = 0; j < VEC; ++j) { s[j] += f[j]; ss[j] += f[j] * f[j]; }
    }
  }
#pragma unroll
  for (int j = 0; j < VEC; ++j) { part[threadIdx.x][j][0] = s[j]; part[threadIdx.x][j][1] = ss[j]; }
  __syncthreads();
  // 2 threads per group: (sum | sumsq)
  if (threadIdx.x < 2 * p.groups) {
    const int g = threadIdx.x >> 1, which = threadIdx.x & 1;
    const int cpg = p.C / p.groups;
    double acc = 0.0;
    for (int c = g * cpg; c < (g + 1) * cpg; ++c) {
      const int ccv = c / VEC, j = c % VEC;
      for (int r = 0; r < R; ++r) acc += (double)part[r * CVp + ccv][j][which];
    }
    p.partial[(((long long)b * p.nsplit + split) * p.groups + g) * 2 + which] = (float)acc;
  }
}

// e4m3fn copy of 8 activated outputs (one 8-byte store): the fp8 convolution's operand
__device__ __forceinline__ void gn_store_fp8(const SdmiGroupNormArgs& p, long long elem, const float* f) {
  uint2 o;
  const float s = p.y8_scale;
  o.x = f32x4_to_fp8x4(f[0] * s, f[1] * s, f[2] * s, f[3] * s);
  o.y = f32x4_to_fp8x4(f[4] * s, f[5] * s, f[6] * s, f[7] * s);
  *reinterpret_cast<uint2*>((fp8_t*)p.y8 + elem) = o;
}

template <typename T, bool F8 = false>
__global__ __launch_bounds__(256) void gn_apply_kernel(SdmiGroupNormArgs p, int rows_per) {
  constexpr int VEC = Elem<T>::VEC;
  __shared__ float s_stats[128][2];
  __shared__ double s_fold[8][32][2];
  const int b = blockIdx.y;
  // mean / rstd of this image's groups from the split partials (fp64 combine, fixed order).  With many partials
  // (the producing convolution wrote one per 32 rows: sdmi.h gn_part) eight threads per group fold a strided eighth
  // each with their loads in flight together -- one thread walking 32 dependent-latency loads cost ~10 us per launch.
  const bool wide = p.groups <= 32 && p.nsplit >= 8;
  if (wide) {
    const int g = threadIdx.x & 31, j = threadIdx.x >> 5;      // 256 threads: 8 folders per group
    if (g < p.groups) {
      double s = 0.0, ss = 0.0;
      for (int k = j; k < p.nsplit; k += 8) {
        const float2 q = *reinterpret_cast<const float2*>(p.partial + (((long long)b * p.nsplit + k) * p.groups + g) * 2);
        s += (double)q.x;
        ss += (double)q.y;
      }
      s_fold[j][g][0] = s;
      s_fold[j][g][1] = ss;
    }
    __syncthreads();
  }
  if ((int)threadIdx.x < p.groups) {
    const int g = threadIdx.x;
    double s = 0.0, ss = 0.0;
    if (wide) {
#pragma unroll
      for (int j = 0; j < 8; ++j) { s += s_fold[j][g][0]; ss += s_fold[j][g][1]; }
    } else {
      for (int k = 0; k < p.nsplit; ++k) {
        const float* q = p.partial + (((long long)b * p.nsplit + k) * p.groups + g) * 2;
        s += (double)q[0];
        ss += (double)q[1];
      }
    }
    const double n = (double)p.HW * (p.C / p.groups);
    const double mean = s / n;
    double var = ss / n - mean * mean;
    if (var < 0.0) var = 0.0;
    const float mf = (float)mean, rf = (float)(1.0 / sqrt(var + (double)p.eps));
    s_stats[g][0] = mf;
    s_stats[g][1] = rf;
    if (blockIdx.x == 0) {
      p.stats[(b * p.groups + g) * 2 + 0] = mf;
      p.stats[(b * p.groups + g) * 2 + 1] = rf;
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
  const int cpg = p.C / p.groups;
  float sc[VEC], sh[VEC];
#pragma unroll
  for (int j = 0; j < VEC; ++j) {
    const int c = cv * VEC + j;
    const int g = c / cpg;
    const float mean = s_stats[g][0];
    const float rstd = s_stats[g][1];
    sc[j] = rstd * p.gamma[c];
    sh[j] = p.beta[c] - mean * sc[j];
  }
  const long long base = (long long)b * p.HW * p.C + cv * VEC;
  int xp;
  const T* xb = gn_src<T>(p, b, cv * VEC, xp);
  T* yb = (T*)p.y + base;
  const T* rb = p.residual ? (const T*)p.residual + base : nullptr;
  const bool drop = p.drop_p > 0.f;
  const unsigned long long dseed = drop ? sdmi_drop_seed(p.drop_seed, p.drop_seed_dev) : 0ULL;
  const unsigned thr16 = (unsigned)(p.drop_p * 65536.f);
  const float dinv = 1.f / (1.f - p.drop_p);
  for (int row = row_begin + r0; row < row_end; row += R) {
    const long long o = (long long)row * p.C;
    const uint4 v = *reinterpret_cast<const uint4*>(xb + (long long)row * xp);
    float f[VEC];
    unpack16<T>(v, f);
    if (rb) {
      float rr[VEC];
      unpack16<T>(*reinterpret_cast<const uint4*>(rb + o), rr);
#pragma unroll
      for (int j = 0; j < VEC; ++j) f[j] = act_apply<sizeof(T) == 2>(f[j] * sc[j] + sh[j] + rr[j], p.act);
    } else {
#pragma unroll
      for (int j = 0; j < VEC; ++j) f[j] = act_apply<sizeof(T) == 2>(f[j] * sc[j] + sh[j], p.act);
    }
    if (drop) sdmi_drop_apply<VEC>(f, dseed, (base + o) / VEC, thr16, dinv);
    if constexpr (F8) {          // e4m3fn operand of the convolution behind the norm, and the bf16 copy when it has a reader (training)
      gn_store_fp8(p, base + o, f);
      if (p.y) *reinterpret_cast<uint4*>(yb + o) = pack16<T>(f);
    } else {
      *reinterpret_cast<uint4*>(yb + o) = pack16<T>(f);
    }
  }
}


// Single-pass GroupNorm: grid (B, S); a workgroup of THREADS (256 / 512 / 1024) threads owns 1/S of
// an image's channels (whole groups) and keeps its [HW][C/S] slab in registers (<= NV vectors per
// thread): statistics and normalisation in ONE launch, x read once.  Per-thread fp32 sums are folded
// across the lanes that share a vector column with xor-butterflies, so the LDS combine (fp64) only
// walks one entry per wave.  Wide workgroups keep whole 128-byte lines per row segment even when the
// image needs many row passes (THREADS / CVp rows per pass).
template <typename T, int THREADS, int NV, bool F8 = false>
__global__ __launch_bounds__(THREADS) void gn_fused_kernel(SdmiGroupNormArgs p) {
  constexpr int VEC = Elem<T>::VEC;
  extern __shared__ __attribute__((aligned(16))) float gn_smem[];
  const int b = blockIdx.x;
  const int S = gridDim.y, sidx = blockIdx.y;
  const int CV = p.C / VEC / S, CVp = next_pow2(CV);     // this workgroup's channel vectors
  const int c_lo = sidx * CV * VEC;                       // first channel of the chunk
  const int R = THREADS / CVp;
  const int RR = CVp < 64 ? THREADS / 64 : R;             // LDS entries per vector column
  float (*part)[VEC][2] = reinterpret_cast<float (*)[VEC][2]>(gn_smem);
  float (*s_stats)[2] = reinterpret_cast<float (*)[2]>(gn_smem + RR * CVp * VEC * 2);
  const int cv = threadIdx.x % CVp, r0 = threadIdx.x / CVp;
  const bool act_c = cv < CV;
  const long long base = (long long)b * p.HW * p.C + c_lo + (act_c ? cv : 0) * VEC;
  int xp;
  const T* xb = gn_src<T>(p, b, c_lo + (act_c ? cv : 0) * VEC, xp);
  uint4 xr[NV];
  float s[VEC], ss[VEC];
#pragma unroll
  for (int j = 0; j < VEC; ++j) s[j] = ss[j] = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int row = r0 + i * R;
    if (act_c && row < p.HW) xr[i] = *reinterpret_cast<const uint4*>(xb + (long long)row * xp);
  }
  // the affine parameters are fetched NOW, next to the slab: this kernel is one latency chain (load ->
  // reduce -> combine -> apply -> store), and a second dependent global load behind the statistics
  // costs about a microsecond of the ~10 a small image takes
  // (not in the 16-vector variants: their slab fills the register budget)
  constexpr bool PRE = NV <= 8;
  float gam[VEC], bet[VEC];
  if constexpr (PRE) {
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      const int c = c_lo + (act_c ? cv : 0) * VEC + j;
      gam[j] = p.gamma[c];
      bet[j] = p.beta[c];
    }
  }
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int row = r0 + i * R;
    if (act_c && row < p.HW) {
      float f[VEC];
      unpack16<T>(xr[i], f);
#pragma unroll
      for (int j = 0; j < VEC; ++j) { s[j] += f[j]; ss[j] += f[j] * f[j]; }
    }
  }
  for (int off = CVp; off < 64; off <<= 1) {
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      s[j] += __shfl_xor(s[j], off, 64);
      ss[j] += __shfl_xor(ss[j], off, 64);
    }
  }
  if (CVp >= 64 || (int)(threadIdx.x & 63) < CVp) {
    const int slot = CVp < 64 ? (threadIdx.x >> 6) * CVp + cv : threadIdx.x;
#pragma unroll
    for (int j = 0; j < VEC; ++j) { part[slot][j][0] = s[j]; part[slot][j][1] = ss[j]; }
  }
  __syncthreads();
  const int cpg = p.C / p.groups;
  const int gs = p.groups / S;               // groups of this chunk
  // two-stage fp64 combine (fixed order): one thread per (channel, sum) folds the RR row entries,
  // then 2 threads per group fold the group's channels
  for (int t = threadIdx.x; t < CV * VEC * 2; t += THREADS) {      // one thread per (channel, sum | sumsq)
    const int ccv = t / (VEC * 2), j = (t >> 1) % VEC, which = t & 1;
    double a = 0.0;
    for (int r = 0; r < RR; ++r) a += (double)part[r * CVp + ccv][j][which];
    // the column's entries are only read by this thread's own fold: entry 0 / 1 take the sum as a
    // (hi, lo) float pair -- the fp64 precision of the fold survives to the group stage (mean and
    // E[x^2] cancel in the variance)
    const float hi = (float)a;
    part[ccv][j][which] = hi;
    if (RR > 1) part[CVp + ccv][j][which] = (float)(a - (double)hi);
  }
  __syncthreads();
  if ((int)threadIdx.x < 2 * gs) {
    const int g = threadIdx.x >> 1, which = threadIdx.x & 1;
    double acc = 0.0;
    for (int c = g * cpg; c < (g + 1) * cpg; ++c) {
      acc += (double)part[c / VEC][c % VEC][which];
      if (RR > 1) acc += (double)part[CVp + c / VEC][c % VEC][which];
    }
    const double other = __shfl_xor(acc, 1, 64);
    const double sum = which ? other : acc, sq = which ? acc : other;
    const double n = (double)p.HW * cpg;
    const double mean = sum / n;
    double var = sq / n - mean * mean;
    if (var < 0.0) var = 0.0;
    if (which == 0) {
      const float mf = (float)mean, rf = (float)(1.0 / sqrt(var + (double)p.eps));
      s_stats[g][0] = mf;
      s_stats[g][1] = rf;
      p.stats[(b * p.groups + sidx * gs + g) * 2 + 0] = mf;
      p.stats[(b * p.groups + sidx * gs + g) * 2 + 1] = rf;
    }
  }
  __syncthreads();
  if (!act_c) return;
  float sc[VEC], sh[VEC];
#pragma unroll
  for (int j = 0; j < VEC; ++j) {
    const int cl = cv * VEC + j, g = cl / cpg;            // chunk-local channel / group
    sc[j] = s_stats[g][1] * (PRE ? gam[j] : p.gamma[c_lo + cl]);
    sh[j] = (PRE ? bet[j] : p.beta[c_lo + cl]) - s_stats[g][0] * sc[j];
  }
  T* yb = (T*)p.y + base;
  const T* rb = p.residual ? (const T*)p.residual + base : nullptr;
  const bool drop = p.drop_p > 0.f;
  const unsigned long long dseed = drop ? sdmi_drop_seed(p.drop_seed, p.drop_seed_dev) : 0ULL;
  const unsigned thr16 = (unsigned)(p.drop_p * 65536.f);
  const float dinv = 1.f / (1.f - p.drop_p);
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int row = r0 + i * R;
    if (row < p.HW) {
      const long long o = (long long)row * p.C;
      float f[VEC];
      unpack16<T>(xr[i], f);
      if (rb) {
        float rr[VEC];
        unpack16<T>(*reinterpret_cast<const uint4*>(rb + o), rr);
#pragma unroll
        for (int j = 0; j < VEC; ++j) f[j] = act_apply<sizeof(T) == 2>(f[j] * sc[j] + sh[j] + rr[j], p.act);
      } else {
#pragma unroll
        for (int j = 0; j < VEC; ++j) f[j] = act_apply<sizeof(T) == 2>(f[j] * sc[j] + sh[j], p.act);
      }
      if (drop) sdmi_drop_apply<VEC>(f, dseed, (base + o) / VEC, thr16, dinv);
      // (the e4m3fn output is its own instantiation: as a run-time branch it pushed the 16-vector
      // variants of the plain kernel into scratch)
      if constexpr (F8) {          // e4m3fn operand of the convolution behind the norm, and the bf16 copy when it has a reader (training)
        gn_store_fp8(p, base + o, f);
        if (p.y) *reinterpret_cast<uint4*>(yb + o) = pack16<T>(f);
      } else {
        *reinterpret_cast<uint4*>(yb + o) = pack16<T>(f);
      }
    }
  }
}

// ------------------------------------------------------------------------------------------
// Single-pass GroupNorm, second form (round 3).  Same geometry and data movement as gn_fused_kernel, but the
// part between the loads and the stores is cut down: in a dependent chain the first form took 2 - 3.5x the
// time of a copy of the same tensor (tools/exp/gn_chain.py: 10.9 vs 4.8 us at [64][16^2][256]), the
// difference being VALU work and the three-barrier fp64 fold, not memory.
//   * a 16-byte vector spans at most TWO groups whenever C/groups >= VEC/2: each lane folds its VEC channel
//     sums into two (sum, sum of squares) slots BEFORE any cross-lane step, so the butterflies move 4 values
//     instead of 2 * VEC (in fp64), the in-row steps as DPP row rotations (no LDS crossbar), an LDS entry is 32 bytes;
//   * fold: one thread per vector column sums the wave entries in fp64; after the second (last) barrier every
//     thread gathers its own two groups' columns (<= 5 reads each) and derives mean / rstd itself -- no third
//     barrier, no single-thread fp64 division + sqrt on the critical path (v_rsq_f64 + one Newton step);
//   * lane coordinates by shifts (CVp is a power of two), gamma / beta as 16-byte loads, the activation
//     hoisted out of the element loop, SiLU through v_rcp_f32 (common.h).
//   * PART (bf16): the slab is not read from x but formed from the split-K partials of the GEMM in front
//     (sdmi.h: part) -- the launch-boundary reduce: the second stage of that GEMM costs no launch of its own.
template <typename T, int THREADS, int NV, bool F8 = false, bool PART = false>
__global__ __launch_bounds__(THREADS) void gn_fused2_kernel(SdmiGroupNormArgs p) {
  constexpr int VEC = Elem<T>::VEC;
  constexpr int VSH = VEC == 8 ? 3 : 2;
  extern __shared__ __attribute__((aligned(16))) float gn_smem[];
  const int b = blockIdx.x;
  const int S = gridDim.y, sidx = blockIdx.y;
  const int CV = p.C / VEC / S;
  const int csh = CV <= 1 ? 0 : 32 - __builtin_clz(CV - 1);
  const int CVp = 1 << csh;
  const int c_lo = sidx * CV * VEC;
  const int R = THREADS >> csh;
  const int RR = CVp < 64 ? THREADS / 64 : R;             // LDS entries per vector column
  // sums: fp32 for bf16 storage, fp64 for fp32 storage (the parity configuration: a near-tie of the evaluation
  // argmax of the 15-slot video fixture depends on the order of an fp32 fold)
  using acc_t = std::conditional_t<sizeof(T) == 4, double, float>;
  acc_t* part = reinterpret_cast<acc_t*>(gn_smem);                           // [RR][CVp][4]
  double* colsum = reinterpret_cast<double*>(part + RR * CVp * 4);           // [CVp][4]
  const int tid = threadIdx.x;
  const int cv = tid & (CVp - 1), r0 = tid >> csh;
  const bool act_c = cv < CV;
  const int cvc = act_c ? cv : 0;
  const int cpg = p.C / p.groups;
  const int cl0 = cvc * VEC;                              // chunk-local first channel of this lane
  const int g0 = cl0 / cpg;                               // its (chunk-local) group = slot 0
  const int bnd = (g0 + 1) * cpg - cl0;                   // channels j >= bnd belong to group g0 + 1 = slot 1
  const long long base = (long long)b * p.HW * p.C + c_lo + cl0;
  int xp;
  const T* xb = gn_src<T>(p, b, c_lo + cl0, xp);
  uint4 xr[NV];
  if constexpr (PART) {
    // in = bf16(sum_k part[k] * alpha + bias + rowvec + residual): igemm.hip's splitk_epilogue_kernel, same order
    // (two-source input: the partials are the FIRST tensor, [B][HW][C1]; a vector never straddles C1)
    static_assert(VEC == 8, "partials source: bf16 storage");
    const int pc = p.x2 ? p.C1 : p.C;
    const bool from_p = c_lo + cl0 < pc;
    const long long total = (long long)p.B * p.HW * pc;
    float eb[VEC], er[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) eb[j] = er[j] = 0.f;
    if (from_p && p.part_bias) load_fvec<VEC>(p.part_bias + c_lo + cl0, eb);
    if (from_p && p.part_rowvec) load_fvec<VEC>(p.part_rowvec + (long long)b * p.part_ldrv + c_lo + cl0, er);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int row = r0 + i * R;
      if (act_c && row < p.HW && !from_p) xr[i] = *reinterpret_cast<const uint4*>(xb + (long long)row * xp);
      if (act_c && row < p.HW && from_p) {
        const long long e = ((long long)b * p.HW + row) * pc + c_lo + cl0;
        const float* pp = p.part + e;
        float v[VEC];
#pragma unroll
        for (int j = 0; j < VEC; ++j) v[j] = 0.f;
#pragma unroll 4
        for (int k = 0; k < p.part_splits; ++k) {
          const float4 lo = *reinterpret_cast<const float4*>(pp + (long long)k * total);
          const float4 hi = *reinterpret_cast<const float4*>(pp + (long long)k * total + 4);
          v[0] += lo.x; v[1] += lo.y; v[2] += lo.z; v[3] += lo.w;
          v[4] += hi.x; v[5] += hi.y; v[6] += hi.z; v[7] += hi.w;
        }
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
          v[j] *= p.part_alpha;
          if (p.part_bias) v[j] += eb[j];
          if (p.part_rowvec) v[j] += er[j];
        }
        if (p.part_residual) {
          float rr[VEC];
          unpack16<T>(*reinterpret_cast<const uint4*>((const T*)p.part_residual + e), rr);
#pragma unroll
          for (int j = 0; j < VEC; ++j) v[j] += rr[j];
        }
        xr[i] = pack16<T>(v);
        *reinterpret_cast<uint4*>((T*)p.raw_out + e) = xr[i];
      }
    }
  } else {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int row = r0 + i * R;
      if (act_c && row < p.HW) xr[i] = *reinterpret_cast<const uint4*>(xb + (long long)row * xp);
    }
  }
  float gam[VEC], bet[VEC];                               // fetched next to the slab (one latency chain)
  load_fvec<VEC>(p.gamma + c_lo + cl0, gam);
  load_fvec<VEC>(p.beta + c_lo + cl0, bet);
  acc_t s[VEC], ss[VEC];
#pragma unroll
  for (int j = 0; j < VEC; ++j) s[j] = ss[j] = (acc_t)0;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int row = r0 + i * R;
    if (act_c && row < p.HW) {
      float f[VEC];
      unpack16<T>(xr[i], f);
#pragma unroll
      for (int j = 0; j < VEC; ++j) { s[j] += (acc_t)f[j]; ss[j] += (acc_t)f[j] * (acc_t)f[j]; }
    }
  }
  acc_t a0 = 0, q0 = 0, a1 = 0, q1 = 0;
#pragma unroll
  for (int j = 0; j < VEC; ++j) {
    const bool lo = j < bnd;
    a0 += lo ? s[j] : (acc_t)0;
    q0 += lo ? ss[j] : (acc_t)0;
    a1 += lo ? (acc_t)0 : s[j];
    q1 += lo ? (acc_t)0 : ss[j];
  }
  if (CVp < 64) {
    a0 = col_allreduce(a0, CVp);
    q0 = col_allreduce(q0, CVp);
    a1 = col_allreduce(a1, CVp);
    q1 = col_allreduce(q1, CVp);
    if ((tid & 63) < CVp) {
      acc_t* e = part + ((tid >> 6) * CVp + cv) * 4;
      e[0] = a0; e[1] = q0; e[2] = a1; e[3] = q1;
    }
  } else {
    acc_t* e = part + tid * 4;
    e[0] = a0; e[1] = q0; e[2] = a1; e[3] = q1;
  }
  __syncthreads();
  if (tid < CVp) {
    double d0 = 0.0, d1 = 0.0, d2 = 0.0, d3 = 0.0;
    for (int r = 0; r < RR; ++r) {
      const acc_t* e = part + (r * CVp + tid) * 4;
      d0 += (double)e[0]; d1 += (double)e[1]; d2 += (double)e[2]; d3 += (double)e[3];
    }
    colsum[tid * 4 + 0] = d0; colsum[tid * 4 + 1] = d1; colsum[tid * 4 + 2] = d2; colsum[tid * 4 + 3] = d3;
  }
  __syncthreads();
  if (!act_c) return;
  const double inv_n = 1.0 / ((double)p.HW * (double)cpg);
  float mean_[2], rstd_[2];
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    if (k == 1 && bnd >= VEC) { mean_[1] = mean_[0]; rstd_[1] = rstd_[0]; break; }
    const int f = (g0 + k) * cpg, l = f + cpg - 1;        // chunk-local channel range of the group
    const int ca = f >> VSH, cb = l >> VSH;
    double sm = 0.0, sq = 0.0;
    for (int c = ca; c <= cb; ++c) {
      const int o = c * 4 + (((c << VSH) >= f) ? 0 : 2);  // the column starts inside the group: its slot 0
      sm += colsum[o];
      sq += colsum[o + 1];
    }
    const double mean = sm * inv_n;
    double var = sq * inv_n - mean * mean;
    if (var < 0.0) var = 0.0;
    const double v = var + (double)p.eps;
    double y = __builtin_amdgcn_rsq(v);
    y = y * (1.5 - 0.5 * v * y * y);
    mean_[k] = (float)mean;
    rstd_[k] = (float)y;
  }
  if (r0 == 0) {                                          // each group's statistics leave through one lane
    const int gg = b * p.groups + sidx * (p.groups / S) + g0;
    if (cl0 == g0 * cpg) { p.stats[gg * 2] = mean_[0]; p.stats[gg * 2 + 1] = rstd_[0]; }
    if (bnd < VEC) { p.stats[gg * 2 + 2] = mean_[1]; p.stats[gg * 2 + 3] = rstd_[1]; }
  }
  float sc[VEC], sh[VEC];
#pragma unroll
  for (int j = 0; j < VEC; ++j) {
    const bool lo = j < bnd;
    sc[j] = (lo ? rstd_[0] : rstd_[1]) * gam[j];
    sh[j] = bet[j] - (lo ? mean_[0] : mean_[1]) * sc[j];
  }
  T* yb = (T*)p.y + base;
  const T* rb = p.residual ? (const T*)p.residual + base : nullptr;
  const bool drop = p.drop_p > 0.f;
  const unsigned long long dseed = drop ? sdmi_drop_seed(p.drop_seed, p.drop_seed_dev) : 0ULL;
  const unsigned thr16 = (unsigned)(p.drop_p * 65536.f);
  const float dinv = 1.f / (1.f - p.drop_p);
  auto finish = [&](auto actc) __attribute__((always_inline)) {
    constexpr int ACT = decltype(actc)::value;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int row = r0 + i * R;
      if (row < p.HW) {
        const long long o = (long long)row * p.C;
        float f[VEC];
        unpack16<T>(xr[i], f);
        if (rb) {
          float rr[VEC];
          unpack16<T>(*reinterpret_cast<const uint4*>(rb + o), rr);
#pragma unroll
          for (int j = 0; j < VEC; ++j) f[j] = act_apply<sizeof(T) == 2>(fmaf(f[j], sc[j], sh[j]) + rr[j], ACT);
        } else {
#pragma unroll
          for (int j = 0; j < VEC; ++j) f[j] = act_apply<sizeof(T) == 2>(fmaf(f[j], sc[j], sh[j]), ACT);
        }
        if (drop) sdmi_drop_apply<VEC>(f, dseed, (base + o) / VEC, thr16, dinv);
        if constexpr (F8) {          // e4m3fn operand of the convolution behind the norm, and the bf16 copy when it has a reader (training)
          gn_store_fp8(p, base + o, f);
          if (p.y) *reinterpret_cast<uint4*>(yb + o) = pack16<T>(f);
        } else {
          *reinterpret_cast<uint4*>(yb + o) = pack16<T>(f);
        }
      }
    }
  };
  if (p.act == SDMI_ACT_SILU) finish(std::integral_constant<int, SDMI_ACT_SILU>{});
  else if (p.act == SDMI_ACT_RELU) finish(std::integral_constant<int, SDMI_ACT_RELU>{});
  else if (p.act == SDMI_ACT_GELU) finish(std::integral_constant<int, SDMI_ACT_GELU>{});
  else finish(std::integral_constant<int, SDMI_ACT_NONE>{});
}

// ------------------------------------------------------------------------------------------
// LayerNorm: a row is covered by LPR = pow2 >= C/VEC lanes holding one 16-byte vector each (two
// when C/VEC > 64), so a wave handles 64/LPR rows per pass and every access is a whole 16-byte
// vector; mean and variance (two-pass, in registers) are xor-butterflies inside the LPR lanes.
template <typename T, int VPL>
__global__ __launch_bounds__(256) void layernorm_vec_kernel(SdmiLayerNormArgs p, int LPR) {
  constexpr int VEC = Elem<T>::VEC;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int sub = lane & (LPR - 1), slot = lane / LPR, RW = 64 / LPR;
  const int CV = p.C / VEC;
  const long long row = ((long long)blockIdx.x * 4 + wave) * RW + slot;
  const bool rok = row < p.rows;
  const T* x = (const T*)p.x + (rok ? row : 0) * p.ldx;
  float v[VPL][VEC];
  bool act[VPL];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < VPL; ++i) {
    const int cv = sub + i * LPR;
    act[i] = cv < CV;
    unpack16<T>(*reinterpret_cast<const uint4*>(x + (act[i] ? cv : 0) * VEC), v[i]);
#pragma unroll
    for (int j = 0; j < VEC; ++j) s += act[i] ? v[i][j] : 0.f;
  }
  for (int off = 1; off < LPR; off <<= 1) s += __shfl_xor(s, off, 64);
  const float mean = s / (float)p.C;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < VPL; ++i)
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      const float d = act[i] ? v[i][j] - mean : 0.f;
      q += d * d;
    }
  for (int off = 1; off < LPR; off <<= 1) q += __shfl_xor(q, off, 64);
  const float rstd = rsqrtf(q / (float)p.C + p.eps);
  if (!rok) return;
  if (p.stats && sub == 0) { p.stats[row * 2] = mean; p.stats[row * 2 + 1] = rstd; }
  T* y = (T*)p.y + row * p.ldy;
#pragma unroll
  for (int i = 0; i < VPL; ++i) {
    if (!act[i]) continue;
    const int c0 = (sub + i * LPR) * VEC;
    float o[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) o[j] = (v[i][j] - mean) * rstd * p.gamma[c0 + j] + p.beta[c0 + j];
    *reinterpret_cast<uint4*>(y + c0) = pack16<T>(o);
  }
}

// scalar form (rows whose pitch or width is not a multiple of the 16-byte vector)
template <typename T>
__global__ __launch_bounds__(256) void layernorm_kernel(SdmiLayerNormArgs p) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= p.rows) return;
  const T* x = (const T*)p.x + (long long)row * p.ldx;
  T* y = (T*)p.y + (long long)row * p.ldy;
  float v[16];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int c = i * 64 + lane;
    v[i] = c < p.C ? Elem<T>::ld(x + c) : 0.f;
    s += v[i];
  }
  const float mean = wave_sum(s) / (float)p.C;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int c = i * 64 + lane;
    const float d = c < p.C ? v[i] - mean : 0.f;
    q += d * d;
  }
  const float rstd = rsqrtf(wave_sum(q) / (float)p.C + p.eps);
  if (p.stats && lane == 0) { p.stats[row * 2] = mean; p.stats[row * 2 + 1] = rstd; }
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int c = i * 64 + lane;
    if (c < p.C) Elem<T>::st(y + c, (v[i] - mean) * rstd * p.gamma[c] + p.beta[c]);
  }
}

// row softmax in place with scale: one wave per row, three L2-resident sweeps.
template <typename T>
__global__ __launch_bounds__(256) void softmax_rows_kernel(SdmiSoftmaxArgs p) {
  const int lane = threadIdx.x & 63;
  const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= p.rows) return;
  T* x = (T*)p.x + row * p.ld;
  float m = -INFINITY;
  for (int c = lane; c < p.cols; c += 64) m = fmaxf(m, Elem<T>::ld(x + c) * p.scale);
  m = wave_max(m);
  float s = 0.f;
  for (int c = lane; c < p.cols; c += 64) s += __expf(Elem<T>::ld(x + c) * p.scale - m);
  s = wave_sum(s);
  const float inv = 1.f / s;
  for (int c = lane; c < p.cols; c += 64)
    Elem<T>::st(x + c, __expf(Elem<T>::ld(x + c) * p.scale - m) * inv);
}

// Rows that fit a wave's registers (cols <= 64 lanes x NV 16-byte vectors: the 32^2 ... 56^2-token attention of
// the VQ-VAE mid block, [B][S][S] scores of 134 - 315 MB): one read, one write, 16-byte accesses -- the scalar
// three-sweep kernel above took 356 us on the 3136-column rows of the 224^2 configuration.
template <typename T, int NV>
__global__ __launch_bounds__(256) void softmax_rows_vec_kernel(SdmiSoftmaxArgs p) {
  constexpr int VEC = Elem<T>::VEC;
  const int lane = threadIdx.x & 63;
  const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= p.rows) return;
  T* x = (T*)p.x + row * p.ld;
  float f[NV][VEC];
  float m = -INFINITY;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = (i * 64 + lane) * VEC;
    if (c < p.cols) {
      unpack16<T>(*reinterpret_cast<const uint4*>(x + c), f[i]);
#pragma unroll
      for (int j = 0; j < VEC; ++j) {
        f[i][j] *= p.scale;
        m = fmaxf(m, f[i][j]);
      }
    }
  }
  m = wave_max(m);
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    if ((i * 64 + lane) * VEC < p.cols) {
#pragma unroll
      for (int j = 0; j < VEC; ++j) {
        f[i][j] = __expf(f[i][j] - m);
        s += f[i][j];
      }
    }
  }
  s = wave_sum(s);
  const float inv = 1.f / s;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = (i * 64 + lane) * VEC;
    if (c < p.cols) {
#pragma unroll
      for (int j = 0; j < VEC; ++j) f[i][j] *= inv;
      *reinterpret_cast<uint4*>(x + c) = pack16<T>(f[i]);
    }
  }
}

}  // namespace

static int gn_validate(const SdmiGroupNormArgs* a) {
  SDMI_REQUIRE(a && a->x && a->gamma && a->beta && a->stats, "null pointer");
  const int vec = a->dtype == SDMI_BF16 ? 8 : 4;
  SDMI_REQUIRE(a->C % vec == 0 && a->C / vec <= 256, "C must be a multiple of the vector width");
  SDMI_REQUIRE(a->groups > 0 && a->groups <= 128 && a->C % a->groups == 0, "bad groups");
  SDMI_REQUIRE(a->nsplit >= 1, "nsplit");
  return SDMI_OK;
}

extern "C" int sdmi_groupnorm_stats(const SdmiGroupNormArgs* a, void* stream) {
  int rc = gn_validate(a);
  if (rc) return rc;
  SDMI_REQUIRE(a->partial, "partial workspace required");
  hipStream_t st = (hipStream_t)stream;
  dim3 grid(a->nsplit, a->B);
  if (a->dtype == SDMI_BF16) hipLaunchKernelGGL(gn_stats_kernel<bf16_t>, grid, dim3(256), 0, st, *a);
  else hipLaunchKernelGGL(gn_stats_kernel<float>, grid, dim3(256), 0, st, *a);
  return sdmi_check_launch("groupnorm_stats");
}

extern "C" int sdmi_groupnorm_apply(const SdmiGroupNormArgs* a, void* stream) {
  int rc = gn_validate(a);
  if (rc) return rc;
  SDMI_REQUIRE((a->y || a->y8) && a->partial, "null output / partial");
  SDMI_REQUIRE(!a->y8 || a->dtype == SDMI_BF16, "fp8 output: bf16 input only");
  hipStream_t st = (hipStream_t)stream;
  // ~32 KiB of activations per workgroup, at least one row-sweep each
  const long long row_bytes = (long long)a->C * (a->dtype == SDMI_BF16 ? 2 : 4);
  int rows_per = (int)((32768 + row_bytes - 1) / row_bytes);
  if (rows_per < 4) rows_per = 4;
  if (rows_per > a->HW) rows_per = a->HW;
  dim3 grid((a->HW + rows_per - 1) / rows_per, a->B);
  if (a->dtype == SDMI_BF16 && a->y8)
    hipLaunchKernelGGL((gn_apply_kernel<bf16_t, true>), grid, dim3(256), 0, st, *a, rows_per);
  else if (a->dtype == SDMI_BF16)
    hipLaunchKernelGGL(gn_apply_kernel<bf16_t>, grid, dim3(256), 0, st, *a, rows_per);
  else
    hipLaunchKernelGGL(gn_apply_kernel<float>, grid, dim3(256), 0, st, *a, rows_per);
  return sdmi_check_launch("groupnorm_apply");
}

// Partials source without a single-pass kernel behind it (shape / fp8 output): the reduction as its own launch,
// the norm then reads raw_out.  Same arithmetic as the fused form.
__global__ __launch_bounds__(256) void gn_part_reduce_kernel(SdmiGroupNormArgs p) {
  const int pc = p.x2 ? p.C1 : p.C;
  const long long total = (long long)p.B * p.HW * pc;
  const long long nvec = total / 8;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (long long)gridDim.x * 256) {
    const long long e = i * 8;
    const int c = (int)(e % pc);
    const int b = (int)(e / ((long long)p.HW * pc));
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = 0.f;
    for (int k = 0; k < p.part_splits; ++k) {
      const float4 lo = *reinterpret_cast<const float4*>(p.part + (long long)k * total + e);
      const float4 hi = *reinterpret_cast<const float4*>(p.part + (long long)k * total + e + 4);
      v[0] += lo.x; v[1] += lo.y; v[2] += lo.z; v[3] += lo.w;
      v[4] += hi.x; v[5] += hi.y; v[6] += hi.z; v[7] += hi.w;
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      v[j] *= p.part_alpha;
      if (p.part_bias) v[j] += p.part_bias[c + j];
      if (p.part_rowvec) v[j] += p.part_rowvec[(long long)b * p.part_ldrv + c + j];
    }
    if (p.part_residual) {
      float rr[8];
      unpack16<bf16_t>(*reinterpret_cast<const uint4*>((const bf16_t*)p.part_residual + e), rr);
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] += rr[j];
    }
    *reinterpret_cast<uint4*>((bf16_t*)p.raw_out + e) = pack16<bf16_t>(v);
  }
}

static constexpr int GN_PART_MIN_SLAB = 2048;

static constexpr bool gn_v2_enabled() {
  return true;
}

extern "C" int sdmi_groupnorm(const SdmiGroupNormArgs* a, void* stream) {
  int rc = gn_validate(a);
  if (rc) return rc;
  SDMI_REQUIRE(a->y || a->y8, "null output");
  SdmiGroupNormArgs loc;
  if (a->part) {
    SDMI_REQUIRE(a->dtype == SDMI_BF16 && a->part_splits > 1 && a->raw_out && a->raw_out != a->y &&
                     ((uintptr_t)a->part & 15) == 0,
                 "partials source: bf16, raw_out required");
    int nv_of_T[3] = {16, 16, 16};
    const GnGeom gg = gn_pick(a->B, a->HW, a->C, a->groups, 8, nv_of_T, GN_PART_MIN_SLAB);
    const int cpg = a->C / a->groups;
    if (a->y8 || !gg.T || !gn_v2_enabled() || !(cpg * 2 == 8 || cpg >= 7)) {
      const long long nvec = (long long)a->B * a->HW * (a->x2 ? a->C1 : a->C) / 8;
      int blocks = (int)((nvec + 255) / 256);
      if (blocks > 2048) blocks = 2048;
      hipLaunchKernelGGL(gn_part_reduce_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, *a);
      rc = sdmi_check_launch("groupnorm (partials reduce)");
      if (rc) return rc;
      loc = *a;
      loc.part = nullptr;
      loc.x = a->raw_out;
      a = &loc;
    }
  }
  SDMI_REQUIRE(!a->y8 || a->dtype == SDMI_BF16, "fp8 output: bf16 input only");
  SDMI_REQUIRE(!a->x2 || (a->C1 > 0 && a->C1 < a->C && a->C1 % (a->dtype == SDMI_BF16 ? 8 : 4) == 0 &&
                          (a->C - a->C1) % (a->dtype == SDMI_BF16 ? 8 : 4) == 0 && a->y != a->x),
               "two-source input: C1 splits the channels on 16-byte vectors; not in place");
  const int vec = a->dtype == SDMI_BF16 ? 8 : 4;
  // single-pass kernel when the image (or a whole-group channel chunk of it) fits a workgroup's registers
  {
    int nv_of_T[3] = {16, 16, 16};
    const GnGeom gg = gn_pick(a->B, a->HW, a->C, a->groups, vec, nv_of_T, a->part ? GN_PART_MIN_SLAB : 8192);
    if (gg.T) {
      hipStream_t st = (hipStream_t)stream;
      dim3 grid(a->B, gg.S);
      const int cv = a->C / vec / gg.S;
      int cvp = 1;
      while (cvp < cv) cvp <<= 1;
      const int RR = cvp < 64 ? gg.T / 64 : gg.T / cvp;
      const size_t smem = ((size_t)RR * cvp * vec * 2 + 2 * 128) * sizeof(float);
      // second form (two group slots per lane): a 16-byte vector must span at most two groups
      const int cpg = a->C / a->groups;
      const bool two = gn_v2_enabled() && (cpg * 2 == vec || cpg >= vec - 1);
      const bool from_part = a->part != nullptr;          // (validated above: implies the second form, bf16, no fp8)
      const size_t smem2 = (size_t)RR * cvp * (vec == 4 ? 32 : 16) + (size_t)cvp * 32;
#define GN_GO3(T_, TH, NV_, F8_)                                                                   \
  do {                                                                                             \
    if (two && from_part) {                                                                        \
      hipLaunchKernelGGL((gn_fused2_kernel<T_, TH, NV_, F8_, sizeof(T_) == 2 && !F8_>), grid, dim3(TH), smem2, st, *a); \
    } else if (two) {                                                                              \
      hipLaunchKernelGGL((gn_fused2_kernel<T_, TH, NV_, F8_>), grid, dim3(TH), smem2, st, *a);     \
    } else {                                                                                       \
      SDMI_OPTIN_LDS((gn_fused_kernel<T_, TH, NV_, F8_>), 80 * 1024, "groupnorm");                 \
      hipLaunchKernelGGL((gn_fused_kernel<T_, TH, NV_, F8_>), grid, dim3(TH), smem, st, *a);       \
    }                                                                                              \
  } while (0)
#define GN_GO2(T_, TH, NV_) GN_GO3(T_, TH, NV_, false)
      // fewest registers that hold the slab: more workgroups per CU
#define GN_GO(T_, TH)                                                                              \
  do {                                                                                             \
    if (gg.need <= 4) GN_GO2(T_, TH, 4); else if (gg.need <= 8) GN_GO2(T_, TH, 8); else GN_GO2(T_, TH, 16); \
  } while (0)
      if (a->dtype == SDMI_BF16 && a->y8) {           // e4m3fn output (inference)
#define GN_GO8(TH)                                                                                 \
  do {                                                                                             \
    if (gg.need <= 4) GN_GO3(bf16_t, TH, 4, true); else if (gg.need <= 8) GN_GO3(bf16_t, TH, 8, true); \
    else GN_GO3(bf16_t, TH, 16, true);                                                             \
  } while (0)
        if (gg.T == 1024) GN_GO8(1024); else if (gg.T == 512) GN_GO8(512); else GN_GO8(256);
#undef GN_GO8
      } else if (a->dtype == SDMI_BF16) {
        if (gg.T == 1024) GN_GO(bf16_t, 1024); else if (gg.T == 512) GN_GO(bf16_t, 512); else GN_GO(bf16_t, 256);
      } else {
        if (gg.T == 1024) GN_GO(float, 1024); else if (gg.T == 512) GN_GO(float, 512); else GN_GO(float, 256);
      }
#undef GN_GO2
#undef GN_GO3
#undef GN_GO
      return sdmi_check_launch("groupnorm (fused)");
    }
  }
  rc = sdmi_groupnorm_stats(a, stream);
  return rc ? rc : sdmi_groupnorm_apply(a, stream);
}

extern "C" int sdmi_layernorm(const SdmiLayerNormArgs* a, void* stream) {
  SDMI_REQUIRE(a && a->x && a->y && a->gamma && a->beta, "null pointer");
  SDMI_REQUIRE(a->C > 0 && a->C <= 1024, "C must be <= 1024");
  hipStream_t st = (hipStream_t)stream;
  const int vec = a->dtype == SDMI_BF16 ? 8 : 4;
  if (a->C % vec == 0 && a->ldx % vec == 0 && a->ldy % vec == 0 && ((uintptr_t)a->x & 15) == 0 &&
      ((uintptr_t)a->y & 15) == 0) {
    const int cv = a->C / vec;
    int lpr = 1;
    while (lpr < cv && lpr < 64) lpr <<= 1;
    const int vpl = (cv + 63) / 64;
    const int rows_per_wg = 4 * (64 / lpr);
    dim3 gridv((unsigned)((a->rows + rows_per_wg - 1) / rows_per_wg));
#define LNF_GO(T, V) hipLaunchKernelGGL((layernorm_vec_kernel<T, V>), gridv, dim3(256), 0, st, *a, lpr)
    if (a->dtype == SDMI_BF16) {
      if (vpl <= 1) LNF_GO(bf16_t, 1); else LNF_GO(bf16_t, 2);
    } else {
      if (vpl <= 1) LNF_GO(float, 1); else if (vpl == 2) LNF_GO(float, 2); else LNF_GO(float, 4);
    }
#undef LNF_GO
    return sdmi_check_launch("layernorm");
  }
  dim3 grid((a->rows + 3) / 4);
  if (a->dtype == SDMI_BF16) hipLaunchKernelGGL(layernorm_kernel<bf16_t>, grid, dim3(256), 0, st, *a);
  else hipLaunchKernelGGL(layernorm_kernel<float>, grid, dim3(256), 0, st, *a);
  return sdmi_check_launch("layernorm");
}

extern "C" int sdmi_softmax_rows(const SdmiSoftmaxArgs* a, void* stream) {
  SDMI_REQUIRE(a && a->x && a->rows > 0 && a->cols > 0, "bad args");
  hipStream_t st = (hipStream_t)stream;
  dim3 grid((a->rows + 3) / 4);
  {
    const int vec = a->dtype == SDMI_BF16 ? 8 : 4;
    const int nv = (a->cols / vec + 63) / 64;
    if (a->cols % vec == 0 && a->ld % vec == 0 && ((uintptr_t)a->x & 15) == 0 && nv <= 8) {
#define SM_GO(T, NV) hipLaunchKernelGGL((softmax_rows_vec_kernel<T, NV>), grid, dim3(256), 0, st, *a)
      if (a->dtype == SDMI_BF16) {
        if (nv <= 2) SM_GO(bf16_t, 2); else if (nv <= 4) SM_GO(bf16_t, 4); else SM_GO(bf16_t, 8);
      } else {
        if (nv <= 2) SM_GO(float, 2); else if (nv <= 4) SM_GO(float, 4); else SM_GO(float, 8);
      }
#undef SM_GO
      return sdmi_check_launch("softmax_rows");
    }
  }
  if (a->dtype == SDMI_BF16)
    hipLaunchKernelGGL(softmax_rows_kernel<bf16_t>, grid, dim3(256), 0, st, *a);
  else
    hipLaunchKernelGGL(softmax_rows_kernel<float>, grid, dim3(256), 0, st, *a);
  return sdmi_check_launch("softmax_rows");
}
