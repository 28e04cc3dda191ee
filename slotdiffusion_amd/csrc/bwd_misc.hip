// Backward helper kernels and the fused optimiser (include/sdmi.h "Backward helpers", "Optimiser").
#include "common.h"

namespace {

constexpr int TH = 256;
static inline int nblocks(long long n) {
  long long b = (n + TH - 1) / TH;
  if (b > 4096) b = 4096;
  if (b < 1) b = 1;
  return (int)b;
}
#define GRID_STRIDE(i, n)                                                         \
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < (n); \
       i += (long long)gridDim.x * blockDim.x)

template <typename T>
__global__ void pack_dgrad_kernel(SdmiPackDgradArgs p) {
  const long long n = (long long)p.Cout * p.KH * p.KW * p.Cin;
  GRID_STRIDE(i, n) {                       // i indexes dst [ci][kh'][kw'][co]
    const int co = (int)(i % p.Cout);
    long long r = i / p.Cout;
    const int kw = (int)(r % p.KW);
    r /= p.KW;
    const int kh = (int)(r % p.KH);
    const int ci = (int)(r / p.KH);
    const long long s = (((long long)co * p.KH + (p.KH - 1 - kh)) * p.KW + (p.KW - 1 - kw)) * p.Cin + ci;
    ((T*)p.dst)[(i / p.Cout) * p.CoutPad + co] = ((const T*)p.src)[s];
  }
}

// Batched version: a workgroup transposes one 64(co) x 64(ci) tile of one tap of one operand
// through LDS -- 16-byte reads along ci, 16-byte writes along co (the per-element version above
// gathers 2-byte elements with a stride of a whole filter).
template <typename T>
__global__ __launch_bounds__(256) void pack_dgrad_batch_kernel(SdmiPackBatchArgs p) {
  constexpr int VEC = Elem<T>::VEC;
  constexpr int VPR = 64 / VEC;                  // vectors per 64-element tile row
  __shared__ T tile[64][64 + 2];
  const SdmiPackDesc* descs = (const SdmiPackDesc*)p.descs;
  int lo = 0, hi = p.n_desc - 1;          // last descriptor with block_begin <= blockIdx.x
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (descs[mid].block_begin <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
  }
  const SdmiPackDesc d = descs[lo];
  int w = (int)blockIdx.x - d.block_begin;
  const int tci = (d.Cin + 63) / 64, tco = (d.Cout + 63) / 64;
  const int ci0 = (w % tci) * 64;
  w /= tci;
  const int co0 = (w % tco) * 64;
  const int tap = w / tco;                       // destination tap slot
  const bool sel = d.nkh > 0;                    // tap selection (parity sub-filter) or the whole filter
  const int nkw = sel ? d.nkw : d.KW;
  const int ti = tap / nkw, tj = tap - ti * nkw;
  const int kh = sel ? d.kh0 + ti * d.kstep : ti, kw = sel ? d.kw0 + tj * d.kstep : tj;   // (kh', kw')
  const int stap = (d.KH - 1 - kh) * d.KW + (d.KW - 1 - kw);
  const int taps = d.KH * d.KW;                  // source taps
  const int dtaps = sel ? d.nkh * d.nkw : taps;  // destination taps
  const T* src = (const T*)d.src;
  T* dst = (T*)d.dst;
  for (int i = threadIdx.x; i < 64 * VPR; i += 256) {
    const int r = i / VPR, c = (i % VPR) * VEC;  // r: co offset, c: ci offset
    const int co = co0 + r, ci = ci0 + c;
    uint4 v = make_uint4(0u, 0u, 0u, 0u);
    if (co < d.Cout && ci < d.Cin)
      v = *reinterpret_cast<const uint4*>(src + ((long long)co * taps + stap) * d.Cin + ci);
    const T* e = reinterpret_cast<const T*>(&v);
#pragma unroll
    for (int j = 0; j < VEC; ++j) tile[r][c + j] = e[j];
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 64 * VPR; i += 256) {
    const int r = i / VPR, c = (i % VPR) * VEC;  // r: ci offset, c: co offset
    const int ci = ci0 + r, co = co0 + c;
    if (ci >= d.Cin || co >= d.CoutPad) continue;
    uint4 v;
    T* e = reinterpret_cast<T*>(&v);
#pragma unroll
    for (int j = 0; j < VEC; ++j) e[j] = tile[c + j][r];   // rows co >= Cout hold zeros
    *reinterpret_cast<uint4*>(dst + ((long long)ci * dtaps + tap) * d.CoutPad + co) = v;
  }
}

// out[g][n] = sum_{r < rows_per} x[g*rows_per + r][n]; block = 256 threads over n, grid (n-blocks, groups)
template <typename T>
__global__ __launch_bounds__(256) void rowgroup_sum_kernel(SdmiRowGroupSumArgs p) {
  const int n = blockIdx.x * 256 + threadIdx.x;
  const int g = blockIdx.y;
  if (n >= p.N) return;
  const T* x = (const T*)p.x + (long long)g * p.rows_per * p.ldx + n;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  int r = 0;
  for (; r + 3 < p.rows_per; r += 4) {
    s0 += Elem<T>::ld(x + (long long)r * p.ldx);
    s1 += Elem<T>::ld(x + (long long)(r + 1) * p.ldx);
    s2 += Elem<T>::ld(x + (long long)(r + 2) * p.ldx);
    s3 += Elem<T>::ld(x + (long long)(r + 3) * p.ldx);
  }
  for (; r < p.rows_per; ++r) s0 += Elem<T>::ld(x + (long long)r * p.ldx);
  p.out[(long long)g * (p.ldo ? p.ldo : p.N) + n] = (s0 + s1) + (s2 + s3);
}

// the same sums for long groups (e.g. 784 rows of a 28 x 28 map per image, few images): the column-per-thread
// walk above is one latency-bound chain per thread; here a workgroup owns 64 columns (32 lanes x 2) and
// its 8 row lanes stride the rows 8 loads deep, partial sums meet in LDS.  N and ldx even.
template <typename T>
__global__ __launch_bounds__(256) void rowgroup_sum_tall_kernel(SdmiRowGroupSumArgs p) {
  __shared__ float part[8][64];
  const int cl = threadIdx.x & 31, rl = threadIdx.x >> 5;
  const int n = blockIdx.x * 64 + 2 * cl;
  const int g = blockIdx.y;
  float s0 = 0.f, s1 = 0.f;
  if (n < p.N) {
    const T* x = (const T*)p.x + (long long)g * p.rows_per * p.ldx + n;
    int r = rl;
    for (; r + 56 < p.rows_per; r += 64) {
      float a[8][2];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const T* q = x + (long long)(r + 8 * u) * p.ldx;
        if constexpr (sizeof(T) == 2) {
          const unsigned w = *reinterpret_cast<const unsigned*>(q);
          a[u][0] = bf16_to_f32((bf16_t)(w & 0xffffu));
          a[u][1] = bf16_to_f32((bf16_t)(w >> 16));
        } else {
          const float2 w = *reinterpret_cast<const float2*>(q);
          a[u][0] = w.x; a[u][1] = w.y;
        }
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) { s0 += a[u][0]; s1 += a[u][1]; }
    }
    for (; r < p.rows_per; r += 8) {
      s0 += Elem<T>::ld(x + (long long)r * p.ldx);
      s1 += Elem<T>::ld(x + (long long)r * p.ldx + 1);
    }
  }
  part[rl][2 * cl] = s0;
  part[rl][2 * cl + 1] = s1;
  __syncthreads();
  if (threadIdx.x < 64) {
    const int c = blockIdx.x * 64 + threadIdx.x;
    if (c < p.N) {
      float t = 0.f;
#pragma unroll
      for (int i = 0; i < 8; ++i) t += part[i][threadIdx.x];
      p.out[(long long)g * (p.ldo ? p.ldo : p.N) + c] = t;
    }
  }
}

template <typename T>
__global__ void pool2x2_kernel(SdmiPool2x2Args p) {      // x [B,2H,2W,C] -> y [B,H,W,C]
  constexpr int VEC = Elem<T>::VEC;
  const int cv = p.C / VEC;
  const long long n = (long long)p.B * p.H * p.W * cv;
  GRID_STRIDE(i, n) {
    const int c = (int)(i % cv) * VEC;
    long long r = i / cv;
    const int x = (int)(r % p.W);
    r /= p.W;
    const int y = (int)(r % p.H);
    const int b = (int)(r / p.H);
    const T* src = (const T*)p.x + ((((long long)b * 2 * p.H + 2 * y) * 2 * p.W) + 2 * x) * p.C + c;
    float a0[VEC], a1[VEC], a2[VEC], a3[VEC];
    unpack16<T>(*reinterpret_cast<const uint4*>(src), a0);
    unpack16<T>(*reinterpret_cast<const uint4*>(src + p.C), a1);
    unpack16<T>(*reinterpret_cast<const uint4*>(src + (long long)2 * p.W * p.C), a2);
    unpack16<T>(*reinterpret_cast<const uint4*>(src + (long long)2 * p.W * p.C + p.C), a3);
#pragma unroll
    for (int j = 0; j < VEC; ++j) a0[j] = (a0[j] + a1[j]) + (a2[j] + a3[j]);
    *reinterpret_cast<uint4*>((T*)p.y + (((long long)b * p.H + y) * p.W + x) * p.C + c) = pack16<T>(a0);
  }
}

template <typename T>
__global__ void add_kernel(SdmiAddArgs p) {
  constexpr int VEC = Elem<T>::VEC;
  const long long nv = p.n / VEC;
  GRID_STRIDE(i, nv) {
    float a[VEC], b[VEC];
    unpack16<T>(*reinterpret_cast<const uint4*>((const T*)p.x + i * VEC), a);
    unpack16<T>(*reinterpret_cast<const uint4*>((const T*)p.z + i * VEC), b);
#pragma unroll
    for (int j = 0; j < VEC; ++j) a[j] += b[j];
    *reinterpret_cast<uint4*>((T*)p.y + i * VEC) = pack16<T>(a);
  }
  // tail
  for (long long i = nv * VEC + (long long)blockIdx.x * blockDim.x + threadIdx.x; i < p.n;
       i += (long long)gridDim.x * blockDim.x)
    Elem<T>::st((T*)p.y + i, Elem<T>::ld((const T*)p.x + i) + Elem<T>::ld((const T*)p.z + i));
}

// grid (blocks, segment): segment blockIdx.y of the table, float4 body + scalar tail
constexpr int SCATTER_MAX = 32;
struct ScatterTable { SdmiScatterItem it[SCATTER_MAX]; };
__global__ __launch_bounds__(256) void scatter_add_kernel(ScatterTable t) {
  const SdmiScatterItem s = t.it[blockIdx.y];
  const bool vec = ((((uintptr_t)s.src) | ((uintptr_t)s.dst)) & 15) == 0;
  const long long nv = vec ? s.count / 4 : 0;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < nv; i += (long long)gridDim.x * 256) {
    f32x4 a = reinterpret_cast<const f32x4*>(s.src)[i];
    a += reinterpret_cast<const f32x4*>(s.dst)[i];
    reinterpret_cast<f32x4*>(s.dst)[i] = a;
  }
  for (long long i = nv * 4 + (long long)blockIdx.x * 256 + threadIdx.x; i < s.count;
       i += (long long)gridDim.x * 256)
    s.dst[i] += s.src[i];
}

struct CopyTable { SdmiCopyItem it[SCATTER_MAX]; };
__global__ __launch_bounds__(256) void copy_group_kernel(CopyTable t) {
  const SdmiCopyItem s = t.it[blockIdx.y];
  const char* src = (const char*)s.src;
  char* dst = (char*)s.dst;
  const bool vec = ((((uintptr_t)src) | ((uintptr_t)dst)) & 15) == 0;
  const long long nv = vec ? s.bytes / 16 : 0;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < nv; i += (long long)gridDim.x * 256)
    reinterpret_cast<uint4*>(dst)[i] = reinterpret_cast<const uint4*>(src)[i];
  for (long long i = nv * 16 + (long long)blockIdx.x * 256 + threadIdx.x; i < s.bytes;
       i += (long long)gridDim.x * 256)
    dst[i] = src[i];
}

template <typename T>
__global__ void split_kernel(SdmiSplitArgs p) {
  constexpr int VEC = Elem<T>::VEC;
  const int cva = p.Ca / VEC, cvb = p.Cb / VEC, cv = cva + cvb;
  const long long n = p.rows * cv;
  GRID_STRIDE(i, n) {
    const long long r = i / cv;
    const int c = (int)(i - r * cv);
    const uint4 v = *reinterpret_cast<const uint4*>((const T*)p.y + r * (p.Ca + p.Cb) + c * VEC);
    if (c < cva) *reinterpret_cast<uint4*>((T*)p.a + r * p.Ca + c * VEC) = v;
    else *reinterpret_cast<uint4*>((T*)p.b + r * p.Cb + (c - cva) * VEC) = v;
  }
}

template <typename T>
__global__ void act_bwd_kernel(SdmiActBwdArgs p) {
  GRID_STRIDE(i, p.n) {
    Elem<T>::st((T*)p.dx + i,
                Elem<T>::ld((const T*)p.dy + i) * act_grad<sizeof(T) == 2>(Elem<T>::ld((const T*)p.x + i), p.act));
  }
}

__device__ __forceinline__ unsigned mix64(unsigned long long z) {   // splitmix64 finaliser
  z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL;
  z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL;
  z = z ^ (z >> 31);
  return (unsigned)(z >> 32);
}
template <typename T>
__global__ void dropout_kernel(SdmiDropoutArgs p) {
  const float inv = 1.f / (1.f - p.p);
  const unsigned thr = (unsigned)((double)p.p * 4294967296.0);
  const unsigned long long seed =
      (unsigned long long)p.seed + (p.seed_dev ? (unsigned long long)(*p.seed_dev) * 0x100000001b3ULL : 0ULL);
  GRID_STRIDE(i, p.n) {
    const unsigned r = mix64(seed + 0x9e3779b97f4a7c15ULL * (unsigned long long)(i + 1));
    const float v = Elem<T>::ld((const T*)p.x + i);
    Elem<T>::st((T*)p.y + i, r >= thr ? v * inv : 0.f);
  }
}

// ---- optimiser --------------------------------------------------------------------------
template <typename GT>
__device__ __forceinline__ void sqsum_body(const SdmiSqSumArgs& p) {
  __shared__ double red[4];
  double acc = 0.0;
  const GT* __restrict__ gp = (const GT*)p.g;
  const long long per = (p.n + p.nblk - 1) / p.nblk;
  const long long i0 = (long long)blockIdx.x * per;
  long long i1 = i0 + per;
  if (i1 > p.n) i1 = p.n;
  // 16-byte loads over the aligned middle of the slice, scalar head / tail
  constexpr int V = Elem<GT>::VEC;
  long long a0 = (i0 + V - 1) & ~(long long)(V - 1), a1 = i1 & ~(long long)(V - 1);
  if (a0 > a1) a0 = a1 = i0;
  for (long long i = i0 + threadIdx.x; i < a0; i += 256) { const double g = Elem<GT>::ld(gp + i); acc += g * g; }
  const uint4* __restrict__ g4 = reinterpret_cast<const uint4*>(gp);
  for (long long i = a0 / V + threadIdx.x; i < a1 / V; i += 256) {
    float f[V];
    unpack16<GT>(g4[i], f);
    double s = 0.0;
#pragma unroll
    for (int j = 0; j < V; j += 4)
      s += ((double)f[j] * (double)f[j] + (double)f[j + 1] * (double)f[j + 1]) +
           ((double)f[j + 2] * (double)f[j + 2] + (double)f[j + 3] * (double)f[j + 3]);
    acc += s;
  }
  for (long long i = a1 + threadIdx.x; i < i1; i += 256) { const double g = Elem<GT>::ld(gp + i); acc += g * g; }
  acc = wave_sum_d(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) p.partial[blockIdx.x] = (float)((red[0] + red[1]) + (red[2] + red[3]));
}
// (un-templated names: bench.py calibrates the FETCH_SIZE counter on `sqsum_kernel`'s known byte count)
__global__ __launch_bounds__(256) void sqsum_kernel(SdmiSqSumArgs p) { sqsum_body<float>(p); }
__global__ __launch_bounds__(256) void sqsum_bf16_kernel(SdmiSqSumArgs p) { sqsum_body<bf16_t>(p); }

__global__ __launch_bounds__(256) void ema_kernel(SdmiEmaArgs p) {
  // the reference's three roundings (ema.py:48-50: sub, mul, sub_): no fused multiply-add here
#pragma clang fp contract(off)
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < p.n;
       i += (long long)gridDim.x * 256) {
    const float s = p.shadow[i];
    const float d = s - p.p[i];
    const float u = p.one_minus_decay * d;
    p.shadow[i] = s - u;
  }
}

// VEC4: four parameters per lane and iteration (16-byte loads of p / g / m / v, 16-byte stores, one
// 8-byte store of the bf16 shadow); the launch picks it when every pointer is 16-byte aligned, the
// tail n % 4 is finished by the first lanes in scalar form.
template <bool VEC4, typename GT = float>
__global__ __launch_bounds__(256) void adam_kernel(SdmiAdamArgs p) {
  // global grad norm from the block partials (every block recomputes the same scalar)
  __shared__ double s_part[256];
  __shared__ float s_coef;
  {
    double s = 0.0;
    for (int i = threadIdx.x; i < p.nblk; i += 256) s += (double)p.sq_partial[i];
    s_part[threadIdx.x] = s;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
      if ((int)threadIdx.x < w) s_part[threadIdx.x] += s_part[threadIdx.x + w];
      __syncthreads();
    }
    if (threadIdx.x == 0) {
      const float total = (float)sqrt(s_part[0]) * (p.gscale != 0.f ? p.gscale : 1.f);   // norm of the scaled gradients
      float c = p.clip > 0.f ? p.clip / (total + 1e-6f) : 1.f;
      s_coef = c < 1.f ? c : 1.f;
    }
    __syncthreads();
  }
  const float coef = s_coef * (p.gscale != 0.f ? p.gscale : 1.f);
  const GT* __restrict__ gp = (const GT*)p.g;
  const int step = p.step_dev ? *p.step_dev : p.step;
  const float lr = p.lr_dev ? *p.lr_dev : p.lr;
  const float bc1 = 1.f - powf(p.beta1, (float)step);
  const float bc2 = 1.f - powf(p.beta2, (float)step);
  const float step_size = lr / bc1;
  const float inv_sqrt_bc2 = 1.f / sqrtf(bc2);
  auto update = [&](float g, float& m, float& v, float& w) {
    g *= coef;
    m = p.beta1 * m + (1.f - p.beta1) * g;
    v = p.beta2 * v + (1.f - p.beta2) * g * g;
    const float denom = sqrtf(v) * inv_sqrt_bc2 + p.eps;
    w = w - step_size * (m / denom);
  };
  long long done = 0;
  if constexpr (VEC4) {
    const long long n4 = p.n / 4;
    GRID_STRIDE(i, n4) {
      f32x4 g4;
      if constexpr (sizeof(GT) == 4) {
        g4 = reinterpret_cast<const f32x4*>(gp)[i];
      } else {
        const uint2 gb = reinterpret_cast<const uint2*>(gp)[i];
        g4 = f32x4{__uint_as_float(gb.x << 16), __uint_as_float(gb.x & 0xffff0000u), __uint_as_float(gb.y << 16),
                   __uint_as_float(gb.y & 0xffff0000u)};
      }
      f32x4 m4 = reinterpret_cast<f32x4*>(p.m)[i];
      f32x4 v4 = reinterpret_cast<f32x4*>(p.v)[i];
      f32x4 w4 = reinterpret_cast<f32x4*>(p.p)[i];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float m = m4[j], v = v4[j], w = w4[j];
        update(g4[j], m, v, w);
        m4[j] = m; v4[j] = v; w4[j] = w;
      }
      reinterpret_cast<f32x4*>(p.m)[i] = m4;
      reinterpret_cast<f32x4*>(p.v)[i] = v4;
      reinterpret_cast<f32x4*>(p.p)[i] = w4;
      if (p.shadow_bf16) {
        uint2 sh;
        sh.x = (unsigned)f32_to_bf16(w4[0]) | ((unsigned)f32_to_bf16(w4[1]) << 16);
        sh.y = (unsigned)f32_to_bf16(w4[2]) | ((unsigned)f32_to_bf16(w4[3]) << 16);
        reinterpret_cast<uint2*>(p.shadow_bf16)[i] = sh;
      }
    }
    done = n4 * 4;
  }
  for (long long i = done + (long long)blockIdx.x * blockDim.x + threadIdx.x; i < p.n;
       i += (long long)gridDim.x * blockDim.x) {
    float m = p.m[i], v = p.v[i], w = p.p[i];
    update(Elem<GT>::ld(gp + i), m, v, w);
    p.m[i] = m;
    p.v[i] = v;
    p.p[i] = w;
    if (p.shadow_bf16) ((bf16_t*)p.shadow_bf16)[i] = f32_to_bf16(w);
  }
}

}  // namespace

#define ST ((hipStream_t)stream)
#define DISPATCH_T(kern, grid, args)                                                    \
  do {                                                                                  \
    if ((args)->dtype == SDMI_BF16)                                                     \
      hipLaunchKernelGGL(kern<bf16_t>, grid, dim3(TH), 0, ST, *(args));                 \
    else                                                                                \
      hipLaunchKernelGGL(kern<float>, grid, dim3(TH), 0, ST, *(args));                  \
  } while (0)

extern "C" int sdmi_pack_dgrad(const SdmiPackDgradArgs* a, void* stream) {
  SDMI_REQUIRE(a && a->src && a->dst && a->CoutPad >= a->Cout, "bad args");
  DISPATCH_T(pack_dgrad_kernel, dim3(nblocks((long long)a->Cout * a->KH * a->KW * a->Cin)), a);
  return sdmi_check_launch("pack_dgrad");
}
extern "C" int sdmi_pack_dgrad_batch(const SdmiPackBatchArgs* a, void* stream) {
  SDMI_REQUIRE(a && a->descs && a->n_desc >= 1 && a->total_blocks >= 1, "bad args");
  DISPATCH_T(pack_dgrad_batch_kernel, dim3(a->total_blocks), a);
  return sdmi_check_launch("pack_dgrad_batch");
}
extern "C" int sdmi_ema_update(const SdmiEmaArgs* a, void* stream) {
  SDMI_REQUIRE(a && a->shadow && a->p && a->n >= 0, "bad args");
  long long nb = (a->n + 255) / 256;
  if (nb > 4096) nb = 4096;
  if (nb < 1) nb = 1;
  hipLaunchKernelGGL(ema_kernel, dim3((int)nb), dim3(256), 0, (hipStream_t)stream, *a);
  return sdmi_check_launch("ema_update");
}
extern "C" int sdmi_rowgroup_sum(const SdmiRowGroupSumArgs* a, void* stream) {
  SDMI_REQUIRE(a && a->x && a->out && a->groups >= 1 && a->rows_per >= 1, "bad args");
  const int es = a->dtype == SDMI_BF16 ? 2 : 4;
  const bool tall = a->rows_per >= 128 && a->N % 2 == 0 && a->ldx % 2 == 0 &&
                    ((uintptr_t)a->x % (2 * es)) == 0 && (long long)a->groups * ((a->N + 255) / 256) < 1024;
  if (tall) DISPATCH_T(rowgroup_sum_tall_kernel, dim3((a->N + 63) / 64, a->groups), a);
  else DISPATCH_T(rowgroup_sum_kernel, dim3((a->N + 255) / 256, a->groups), a);
  return sdmi_check_launch("rowgroup_sum");
}
extern "C" int sdmi_pool2x2_sum(const SdmiPool2x2Args* a, void* stream) {
  SDMI_REQUIRE(a && a->x && a->y, "null pointer");
  const int vec = a->dtype == SDMI_BF16 ? 8 : 4;
  SDMI_REQUIRE(a->C % vec == 0, "C must be a vector multiple");
  DISPATCH_T(pool2x2_kernel, dim3(nblocks((long long)a->B * a->H * a->W * (a->C / vec))), a);
  return sdmi_check_launch("pool2x2_sum");
}
extern "C" int sdmi_add(const SdmiAddArgs* a, void* stream) {
  SDMI_REQUIRE(a && a->x && a->z && a->y, "null pointer");
  DISPATCH_T(add_kernel, dim3(nblocks(a->n / 4 + 1)), a);
  return sdmi_check_launch("add");
}
extern "C" int sdmi_scatter_add(const SdmiScatterAddArgs* a, void* stream) {
  SDMI_REQUIRE(a && a->items && a->n >= 1 && a->n <= SCATTER_MAX, "1 .. 32 segments");
  const SdmiScatterItem* it = (const SdmiScatterItem*)a->items;
  ScatterTable t;
  long long mx = 0;
  for (int i = 0; i < a->n; ++i) {
    SDMI_REQUIRE(it[i].src && it[i].dst && it[i].count >= 0, "bad segment");
    t.it[i] = it[i];
    if (it[i].count > mx) mx = it[i].count;
  }
  long long blocks = (mx / 4 + 255) / 256;
  if (blocks < 1) blocks = 1;
  if (blocks > 256) blocks = 256;
  hipLaunchKernelGGL(scatter_add_kernel, dim3((unsigned)blocks, a->n), dim3(256), 0, (hipStream_t)stream, t);
  return sdmi_check_launch("scatter_add");
}
extern "C" int sdmi_copy_group(const SdmiCopyGroupArgs* a, void* stream) {
  SDMI_REQUIRE(a && a->items && a->n >= 1 && a->n <= SCATTER_MAX, "1 .. 32 copies");
  const SdmiCopyItem* it = (const SdmiCopyItem*)a->items;
  CopyTable t;
  long long mx = 0;
  for (int i = 0; i < a->n; ++i) {
    SDMI_REQUIRE(it[i].src && it[i].dst && it[i].bytes >= 0, "bad item");
    t.it[i] = it[i];
    if (it[i].bytes > mx) mx = it[i].bytes;
  }
  long long blocks = (mx / 16 + 255) / 256;
  if (blocks < 1) blocks = 1;
  if (blocks > 256) blocks = 256;
  hipLaunchKernelGGL(copy_group_kernel, dim3((unsigned)blocks, a->n), dim3(256), 0, (hipStream_t)stream, t);
  return sdmi_check_launch("copy_group");
}
extern "C" int sdmi_split_channels(const SdmiSplitArgs* a, void* stream) {
  SDMI_REQUIRE(a && a->y && a->a && a->b, "null pointer");
  const int vec = a->dtype == SDMI_BF16 ? 8 : 4;
  SDMI_REQUIRE(a->Ca % vec == 0 && a->Cb % vec == 0, "channel counts must be vector multiples");
  DISPATCH_T(split_kernel, dim3(nblocks(a->rows * ((a->Ca + a->Cb) / vec))), a);
  return sdmi_check_launch("split_channels");
}
extern "C" int sdmi_act_bwd(const SdmiActBwdArgs* a, void* stream) {
  SDMI_REQUIRE(a && a->x && a->dy && a->dx, "null pointer");
  DISPATCH_T(act_bwd_kernel, dim3(nblocks(a->n)), a);
  return sdmi_check_launch("act_bwd");
}
extern "C" int sdmi_dropout(const SdmiDropoutArgs* a, void* stream) {
  SDMI_REQUIRE(a && a->x && a->y && a->p >= 0.f && a->p < 1.f, "bad args");
  DISPATCH_T(dropout_kernel, dim3(nblocks(a->n)), a);
  return sdmi_check_launch("dropout");
}
extern "C" int sdmi_sqsum_partial(const SdmiSqSumArgs* a, void* stream) {
  SDMI_REQUIRE(a && a->g && a->partial && a->nblk >= 1 && (a->g_dtype == SDMI_F32 || a->g_dtype == SDMI_BF16), "bad args");
  if (a->g_dtype == SDMI_BF16) hipLaunchKernelGGL(sqsum_bf16_kernel, dim3(a->nblk), dim3(256), 0, ST, *a);
  else hipLaunchKernelGGL(sqsum_kernel, dim3(a->nblk), dim3(256), 0, ST, *a);
  return sdmi_check_launch("sqsum_partial");
}
extern "C" int sdmi_adam_clip(const SdmiAdamArgs* a, void* stream) {
  SDMI_REQUIRE(a && a->p && a->g && a->m && a->v && a->sq_partial && a->nblk >= 1 &&
                   (a->step >= 1 || a->step_dev) && (a->g_dtype == SDMI_F32 || a->g_dtype == SDMI_BF16) &&
                   a->gscale >= 0.f, "bad args");
  const bool g16 = a->g_dtype == SDMI_BF16;
  const uintptr_t al = (uintptr_t)a->p | ((uintptr_t)a->g << (g16 ? 1 : 0)) | (uintptr_t)a->m | (uintptr_t)a->v |
                       ((uintptr_t)a->shadow_bf16 << 1);        // bf16 buffers need 8-byte alignment
  const bool vec = al % 16 == 0 && a->n >= 4;
  if (g16) {
    if (vec) hipLaunchKernelGGL((adam_kernel<true, bf16_t>), dim3(nblocks(a->n / 4)), dim3(256), 0, ST, *a);
    else hipLaunchKernelGGL((adam_kernel<false, bf16_t>), dim3(nblocks(a->n)), dim3(256), 0, ST, *a);
  } else if (vec) {
    hipLaunchKernelGGL((adam_kernel<true, float>), dim3(nblocks(a->n / 4)), dim3(256), 0, ST, *a);
  } else {
    hipLaunchKernelGGL((adam_kernel<false, float>), dim3(nblocks(a->n)), dim3(256), 0, ST, *a);
  }
  return sdmi_check_launch("adam_clip");
}
