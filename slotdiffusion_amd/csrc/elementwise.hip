// Small HBM-bound fused elementwise kernels (include/sdmi.h, "Small fused elementwise kernels").
#include "common.h"

namespace {

constexpr int EW_THREADS = 256;
static inline int ew_blocks(long long n, int per_thread = 1) {
  long long b = (n + (long long)EW_THREADS * per_thread - 1) / ((long long)EW_THREADS * per_thread);
  if (b > 4096) b = 4096;
  if (b < 1) b = 1;
  return (int)b;
}
#define GRID_STRIDE(i, n)                                                         \
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < (n); \
       i += (long long)gridDim.x * blockDim.x)

__global__ void lincomb_kernel(SdmiLincombArgs p) {
  GRID_STRIDE(i, p.n) {
    float v = 0.f;
    if (p.x0) v = p.c0 * p.x0[i];
    if (p.x1) {
      const float t = p.c1 * p.x1[i];
      v = p.x0 ? v + t : t;
    }
    if (p.x2) {
      const float d = p.x3 ? p.x2[i] - p.x3[i] : p.x2[i];
      v = v + p.c2 * d;
    }
    if (p.div != 0.f) v = v / p.div;
    p.y[i] = v;
  }
}

__global__ void row_lincomb_kernel(SdmiRowLincombArgs p) {
  const long long n = (long long)p.B * p.per;
  GRID_STRIDE(i, n) {
    const int b = (int)(i / p.per);
    p.y[i] = p.ca[b] * p.x0[i] + p.cb[b] * p.x1[i];
  }
}

template <typename T>
__global__ void nchw_to_nhwc_kernel(SdmiNchwToNhwcArgs p) {
  const long long n = (long long)p.B * p.H * p.W * p.Cpad;
  GRID_STRIDE(i, n) {
    const int c = (int)(i % p.Cpad);
    const long long pix = i / p.Cpad;
    const int hw = p.H * p.W;
    const int b = (int)(pix / hw);
    const int r = (int)(pix - (long long)b * hw);
    const float v = c < p.C ? p.src[((long long)b * p.C + c) * hw + r] : 0.f;
    Elem<T>::st((T*)p.dst + i, v);
  }
}

template <typename T>
__global__ void nhwc_to_nchw_kernel(SdmiNhwcToNchwArgs p) {
  const long long n = (long long)p.B * p.C * p.H * p.W;
  GRID_STRIDE(i, n) {
    const int hw = p.H * p.W;
    const int r = (int)(i % hw);
    const long long bc = i / hw;
    const int c = (int)(bc % p.C);
    const int b = (int)(bc / p.C);
    p.dst[i] = Elem<T>::ld((const T*)p.src + ((long long)b * hw + r) * p.Cpad + c);
  }
}

template <typename S, typename D>
__global__ void cast2d_kernel(SdmiCast2dArgs p) {
  const int wc = p.zpad ? p.ldd : p.cols;          // columns written per row
  const long long n = p.rows * wc;
  GRID_STRIDE(i, n) {
    const long long r = i / wc;
    const int c = (int)(i - r * wc);
    Elem<D>::st((D*)p.dst + r * p.ldd + c, c < p.cols ? Elem<S>::ld((const S*)p.src + r * p.lds + c) : 0.f);
  }
}

// per-tensor fp8 (e4m3fn) quantisation, 16 values per thread (one 16-byte store)
template <typename S>
__global__ void quant_fp8_kernel(SdmiQuantFp8Args p) {
  const int vpr = p.ldd / 16;                       // 16-byte output vectors per row
  const long long n = p.rows * vpr;
  GRID_STRIDE(i, n) {
    const long long r = i / vpr;
    const int c0 = (int)(i - r * vpr) * 16;
    const S* src = (const S*)p.src + r * p.lds + c0;
    float f[16];
    if (c0 + 16 <= p.cols && (p.lds % Elem<S>::VEC) == 0) {
#pragma unroll
      for (int q = 0; q < 16 / Elem<S>::VEC; ++q)
        unpack16<S>(*reinterpret_cast<const uint4*>(src + q * Elem<S>::VEC), f + q * Elem<S>::VEC);
    } else {
#pragma unroll
      for (int j = 0; j < 16; ++j) f[j] = c0 + j < p.cols ? Elem<S>::ld(src + j) : 0.f;
    }
    uint4 o;
    o.x = f32x4_to_fp8x4(f[0] * p.scale, f[1] * p.scale, f[2] * p.scale, f[3] * p.scale);
    o.y = f32x4_to_fp8x4(f[4] * p.scale, f[5] * p.scale, f[6] * p.scale, f[7] * p.scale);
    o.z = f32x4_to_fp8x4(f[8] * p.scale, f[9] * p.scale, f[10] * p.scale, f[11] * p.scale);
    o.w = f32x4_to_fp8x4(f[12] * p.scale, f[13] * p.scale, f[14] * p.scale, f[15] * p.scale);
    *reinterpret_cast<uint4*>((fp8_t*)p.dst + r * p.ldd + c0) = o;
  }
}

// grouped per-tensor fp8 quantisation with device-side scales (sdmi.h: sdmi_fp8_quant_group): a workgroup serves
// 4096 consecutive elements of one tensor of the table; pass 1 folds max |x| into amax_bits[d] (non-negative floats
// order like their bit patterns: atomicMax on the bits), pass 2 derives the scale from it
__device__ __forceinline__ int fp8_desc_of(const SdmiFp8Desc* descs, int n_desc, int b) {
  int lo = 0, hi = n_desc - 1;            // last descriptor with block_begin <= b
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (descs[mid].block_begin <= b) lo = mid; else hi = mid - 1;
  }
  return lo;
}
template <typename S>
__device__ __forceinline__ void fp8_load16(const S* src, float* f) {
#pragma unroll
  for (int q = 0; q < 16 / Elem<S>::VEC; ++q)
    unpack16<S>(*reinterpret_cast<const uint4*>(src + q * Elem<S>::VEC), f + q * Elem<S>::VEC);
}
template <typename S>
__global__ __launch_bounds__(256) void fp8_group_amax_kernel(SdmiFp8GroupArgs p) {
  const SdmiFp8Desc* descs = (const SdmiFp8Desc*)p.descs;
  const int di = fp8_desc_of(descs, p.n_desc, (int)blockIdx.x);
  const SdmiFp8Desc d = descs[di];
  const long long i0 = ((long long)((int)blockIdx.x - d.block_begin) * 256 + threadIdx.x) * 16;
  float m = 0.f;
  if (i0 < d.n) {
    float f[16];
    fp8_load16<S>((const S*)d.src + i0, f);
#pragma unroll
    for (int j = 0; j < 16; ++j) m = fmaxf(m, fabsf(f[j]));
  }
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
  __shared__ float wm[4];
  if ((threadIdx.x & 63) == 0) wm[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    m = fmaxf(fmaxf(wm[0], wm[1]), fmaxf(wm[2], wm[3]));
    atomicMax(p.amax_bits + di, __float_as_uint(m));
  }
}
template <typename S>
__global__ __launch_bounds__(256) void fp8_group_quant_kernel(SdmiFp8GroupArgs p) {
  const SdmiFp8Desc* descs = (const SdmiFp8Desc*)p.descs;
  const int di = fp8_desc_of(descs, p.n_desc, (int)blockIdx.x);
  const SdmiFp8Desc d = descs[di];
  const float amax = fmaxf(__uint_as_float(p.amax_bits[di]), 1e-12f);
  const float scale = 448.f / amax;
  if ((int)blockIdx.x == d.block_begin && threadIdx.x == 0) p.inv_scale[di] = amax / 448.f;
  const long long i0 = ((long long)((int)blockIdx.x - d.block_begin) * 256 + threadIdx.x) * 16;
  if (i0 >= d.n) return;
  float f[16];
  fp8_load16<S>((const S*)d.src + i0, f);
  uint4 o;
  o.x = f32x4_to_fp8x4(f[0] * scale, f[1] * scale, f[2] * scale, f[3] * scale);
  o.y = f32x4_to_fp8x4(f[4] * scale, f[5] * scale, f[6] * scale, f[7] * scale);
  o.z = f32x4_to_fp8x4(f[8] * scale, f[9] * scale, f[10] * scale, f[11] * scale);
  o.w = f32x4_to_fp8x4(f[12] * scale, f[13] * scale, f[14] * scale, f[15] * scale);
  *reinterpret_cast<uint4*>((fp8_t*)d.dst + i0) = o;
}

template <typename T>
__global__ void expand_heads_kernel(SdmiExpandHeadsArgs p) {
  const int gw = p.gw == 16 ? 16 : 8;
  const int R = p.heads * gw, hd = p.C / p.heads;
  const long long n = (long long)p.B * R * p.C;
  GRID_STRIDE(i, n) {
    const int c = (int)(i % p.C);
    const long long br = i / p.C;
    const int row = (int)(br % R), b = (int)(br / R);
    const int h = row / gw, j = row - h * gw;
    float k = 0.f, v = 0.f;
    if (j < p.S && c / hd == h) {
      const T* src = (const T*)p.kv + ((long long)b * p.S + j) * p.ldkv;
      k = Elem<T>::ld(src + c) * p.scale;
      v = Elem<T>::ld(src + p.C + c);
    }
    Elem<T>::st((T*)p.kexp + i, k);
    Elem<T>::st((T*)p.vexp + i, v);
  }
}

template <typename T>
__global__ void scale_dev_kernel(SdmiScaleDevArgs p) {
  const float s = p.s[0];
  GRID_STRIDE(i, p.n) Elem<T>::st((T*)p.y + i, Elem<T>::ld((const T*)p.x + i) * s);
}

__global__ void counters_inc_kernel(SdmiCountersArgs p) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    if (p.step) p.step[0] += 1;
    if (p.seed) p.seed[0] += 1;
  }
}

// training draws: uniform timesteps + Box-Muller normals from a splitmix64 counter generator
__global__ void draw_tn_kernel(SdmiDrawTnArgs p) {
  const unsigned long long seed = sdmi_drop_seed(p.seed, p.seed_dev);
  const long long n = (long long)p.B * p.per;
  GRID_STRIDE(i, n + p.B) {
    if (i >= n) {                                   // the B timestep draws ride in the tail threads
      const int b = (int)(i - n);
      const unsigned long long r = sdmi_mix64((seed ^ 0x7f4a7c159e3779b9ULL) + 0x9e3779b97f4a7c15ULL * (unsigned long long)(b + 1));
      const long long t = (long long)(r % (unsigned long long)p.T);
      p.t[b] = t;
      if (p.tf) p.tf[b] = (float)t;
      if (p.ca) p.ca[b] = p.tab_a[t];
      if (p.cb) p.cb[b] = p.tab_b[t];
      continue;
    }
    float z = 0.f;
    if ((i & 3) != 3) {
      const unsigned long long r = sdmi_mix64(seed + 0x9e3779b97f4a7c15ULL * (unsigned long long)(i + 1));
      const float u1 = ((float)(unsigned)(r >> 40) + 0.5f) * (1.f / 16777216.f);        // (0, 1)
      const float u2 = ((float)(unsigned)((r >> 8) & 0xffffffu)) * (1.f / 16777216.f);  // [0, 1)
      z = sqrtf(-2.f * logf(u1)) * cosf(6.28318530717958647692f * u2);
    }
    p.noise[i] = z;
  }
}

__global__ void time_emb_kernel(SdmiTimeEmbArgs p) {
  const int half = p.dim / 2;
  const long long n = (long long)p.B * p.dim;
  GRID_STRIDE(i, n) {
    const int b = (int)(i / p.dim), j = (int)(i % p.dim);
    const int f = j < half ? j : j - half;
    const float freq = expf(-logf(p.max_period) * (float)f / (float)half);
    const float arg = p.t[b] * freq;
    p.out[i] = j < half ? cosf(arg) : sinf(arg);
  }
}

template <typename S, typename D>
__global__ void act_kernel(SdmiActArgs p) {
  GRID_STRIDE(i, p.n) {
    Elem<D>::st((D*)p.y + i, act_apply<sizeof(D) == 2>(Elem<S>::ld((const S*)p.x + i), p.act));
  }
}

template <typename T>
__global__ void geglu_kernel(SdmiGegluArgs p) {
  constexpr int VEC = Elem<T>::VEC;
  const int cv = p.C / VEC;
  const long long n = p.rows * cv;
  GRID_STRIDE(i, n) {
    const long long r = i / cv;
    const int c = (int)(i - r * cv) * VEC;
    const T* h = (const T*)p.h + r * 2 * p.C;
    float x[VEC], g[VEC];
    unpack16<T>(*reinterpret_cast<const uint4*>(h + c), x);
    unpack16<T>(*reinterpret_cast<const uint4*>(h + p.C + c), g);
#pragma unroll
    for (int j = 0; j < VEC; ++j) x[j] *= act_apply<sizeof(T) == 2>(g[j], SDMI_ACT_GELU);
    *reinterpret_cast<uint4*>((T*)p.y + r * p.C + c) = pack16<T>(x);
  }
}

template <typename T>
__global__ void geglu_bwd_kernel(SdmiGegluBwdArgs p) {
  constexpr int VEC = Elem<T>::VEC;
  const int cv = p.C / VEC;
  const long long n = p.rows * cv;
  GRID_STRIDE(i, n) {
    const long long r = i / cv;
    const int c = (int)(i - r * cv) * VEC;
    const T* h = (const T*)p.h + r * 2 * p.C;
    T* dh = (T*)p.dh + r * 2 * p.C;
    float x[VEC], g[VEC], dy[VEC], dx[VEC], dg[VEC];
    unpack16<T>(*reinterpret_cast<const uint4*>(h + c), x);
    unpack16<T>(*reinterpret_cast<const uint4*>(h + p.C + c), g);
    unpack16<T>(*reinterpret_cast<const uint4*>((const T*)p.dy + r * p.C + c), dy);
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      dx[j] = dy[j] * act_apply<sizeof(T) == 2>(g[j], SDMI_ACT_GELU);
      dg[j] = dy[j] * x[j] * act_grad<sizeof(T) == 2>(g[j], SDMI_ACT_GELU);
    }
    *reinterpret_cast<uint4*>(dh + c) = pack16<T>(dx);
    *reinterpret_cast<uint4*>(dh + p.C + c) = pack16<T>(dg);
  }
}

template <typename T>
__global__ void add_pos_kernel(SdmiAddPosArgs p) {
  constexpr int VEC = Elem<T>::VEC;
  const long long per_v = p.per / VEC;
  const long long n = (long long)p.B * per_v;
  GRID_STRIDE(i, n) {
    const long long o = (i % per_v) * VEC;
    float x[VEC];
    unpack16<T>(*reinterpret_cast<const uint4*>((const T*)p.x + i * VEC), x);
#pragma unroll
    for (int j = 0; j < VEC; ++j) x[j] += p.pos[o + j];
    *reinterpret_cast<uint4*>((T*)p.y + i * VEC) = pack16<T>(x);
  }
}


template <typename T>
__global__ void broadcast_pos_kernel(SdmiBroadcastPosArgs p) {
  constexpr int VEC = Elem<T>::VEC;
  const int cv = p.C / VEC;
  const long long n = (long long)p.G * p.R * cv;
  GRID_STRIDE(i, n) {
    const int c = (int)(i % cv) * VEC;
    const long long gr = i / cv;
    const int r = (int)(gr % p.R);
    const long long g = gr / p.R;
    float v[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) v[j] = p.x[g * p.C + c + j] + p.pos[(long long)r * p.C + c + j];
    *reinterpret_cast<uint4*>((T*)p.y + i * VEC) = pack16<T>(v);
  }
}

template <typename T>
__global__ void sa_combine_kernel(SdmiSaCombineArgs p) {
  const long long n = (long long)p.B * p.HW;
  GRID_STRIDE(i, n) {
    const long long b = i / p.HW;
    const int px = (int)(i - b * p.HW);
    const T* o = (const T*)p.o + ((b * p.N) * p.HW + px) * p.ldo;
    const long long sn = (long long)p.HW * p.ldo;       // slot stride
    float mx = -INFINITY;
    for (int s = 0; s < p.N; ++s) mx = fmaxf(mx, Elem<T>::ld(o + s * sn + 3));
    float den = 0.f, r0 = 0.f, r1 = 0.f, r2 = 0.f;
    for (int s = 0; s < p.N; ++s) {
      const float e = __expf(Elem<T>::ld(o + s * sn + 3) - mx);
      den += e;
      r0 += e * Elem<T>::ld(o + s * sn);
      r1 += e * Elem<T>::ld(o + s * sn + 1);
      r2 += e * Elem<T>::ld(o + s * sn + 2);
    }
    const float inv = 1.f / den;
    for (int s = 0; s < p.N; ++s)
      p.masks[(b * p.N + s) * p.HW + px] = __expf(Elem<T>::ld(o + s * sn + 3) - mx) * inv;
    float4 w = make_float4(r0 * inv, r1 * inv, r2 * inv, 0.f);
    reinterpret_cast<float4*>(p.recon)[i] = w;
  }
}

template <typename T>
__global__ void sa_combine_bwd_kernel(SdmiSaCombineBwdArgs p) {
  const long long n = (long long)p.B * p.HW;
  GRID_STRIDE(i, n) {
    const long long b = i / p.HW;
    const int px = (int)(i - b * p.HW);
    const long long sn = (long long)p.HW * p.ldo;
    const T* o = (const T*)p.o + ((b * p.N) * p.HW + px) * p.ldo;
    T* d = (T*)p.dout + ((b * p.N) * p.HW + px) * p.ldo;
    const float4 dr = reinterpret_cast<const float4*>(p.drecon)[i];
    float t = 0.f;
    for (int s = 0; s < p.N; ++s) {
      const float m = p.masks[(b * p.N + s) * p.HW + px];
      const float sdot = Elem<T>::ld(o + s * sn) * dr.x + Elem<T>::ld(o + s * sn + 1) * dr.y +
                         Elem<T>::ld(o + s * sn + 2) * dr.z;
      t += m * sdot;
    }
    for (int s = 0; s < p.N; ++s) {
      const float m = p.masks[(b * p.N + s) * p.HW + px];
      const float sdot = Elem<T>::ld(o + s * sn) * dr.x + Elem<T>::ld(o + s * sn + 1) * dr.y +
                         Elem<T>::ld(o + s * sn + 2) * dr.z;
      Elem<T>::st(d + s * sn, m * dr.x);
      Elem<T>::st(d + s * sn + 1, m * dr.y);
      Elem<T>::st(d + s * sn + 2, m * dr.z);
      Elem<T>::st(d + s * sn + 3, m * (sdot - t));
      for (int c = 4; c < p.ldo; ++c) Elem<T>::st(d + s * sn + c, 0.f);
    }
  }
}

template <typename T>
__global__ void concat_kernel(SdmiConcatArgs p) {
  constexpr int VEC = Elem<T>::VEC;
  const int cva = p.Ca / VEC, cvb = p.Cb / VEC, cv = cva + cvb;
  const long long n = p.rows * cv;
  GRID_STRIDE(i, n) {
    const long long r = i / cv;
    const int c = (int)(i - r * cv);
    const uint4 v = c < cva
                        ? *reinterpret_cast<const uint4*>((const T*)p.a + r * p.Ca + c * VEC)
                        : *reinterpret_cast<const uint4*>((const T*)p.b + r * p.Cb + (c - cva) * VEC);
    *reinterpret_cast<uint4*>((T*)p.y + r * (p.Ca + p.Cb) + c * VEC) = v;
  }
}

// bilinear upsample (align_corners=False, torch's area_pixel source index) + argmax over slots
__global__ void mask_up_kernel(SdmiMaskUpArgs p) {
  const long long n = (long long)p.B * p.H * p.W;
  const float sy = (float)p.h / (float)p.H, sx = (float)p.w / (float)p.W;
  GRID_STRIDE(i, n) {
    const int x = (int)(i % p.W);
    const int y = (int)((i / p.W) % p.H);
    const int b = (int)(i / ((long long)p.W * p.H));
    float fy = sy * ((float)y + 0.5f) - 0.5f;
    float fx = sx * ((float)x + 0.5f) - 0.5f;
    if (fy < 0.f) fy = 0.f;
    if (fx < 0.f) fx = 0.f;
    const int y0 = (int)fy, x0 = (int)fx;
    const int y1 = y0 + (y0 < p.h - 1 ? 1 : 0), x1 = x0 + (x0 < p.w - 1 ? 1 : 0);
    const float ly1 = fy - (float)y0, ly0 = 1.f - ly1;
    const float lx1 = fx - (float)x0, lx0 = 1.f - lx1;
    const float* s = p.seg + (long long)b * p.h * p.w * p.N;
    float best = -INFINITY;
    int bi = 0;
    for (int k = 0; k < p.N; ++k) {
      const float v00 = s[((long long)y0 * p.w + x0) * p.N + k];
      const float v01 = s[((long long)y0 * p.w + x1) * p.N + k];
      const float v10 = s[((long long)y1 * p.w + x0) * p.N + k];
      const float v11 = s[((long long)y1 * p.w + x1) * p.N + k];
      const float v = ly0 * (lx0 * v00 + lx1 * v01) + ly1 * (lx0 * v10 + lx1 * v11);
      if (p.up) p.up[(((long long)b * p.N + k) * p.H + y) * p.W + x] = v;
      if (v > best) { best = v; bi = k; }
    }
    if (p.idx) p.idx[i] = bi;
  }
}

template <typename T>
__global__ __launch_bounds__(256) void mse_kernel(SdmiMseArgs p) {
  __shared__ double red[4];
  double acc = 0.0;
  const float gs = (p.l1 ? 1.f : 2.f) * p.gscale / (float)p.n;
  GRID_STRIDE(i, p.n) {
    const float d = Elem<T>::ld((const T*)p.pred + i) - p.target[i];
    if (p.l1) {
      acc += (double)fabsf(d);
      if (p.dpred) Elem<T>::st((T*)p.dpred + i, d > 0.f ? gs : (d < 0.f ? -gs : 0.f));
    } else {
      acc += (double)d * (double)d;
      if (p.dpred) Elem<T>::st((T*)p.dpred + i, d * gs);
    }
  }
  acc = wave_sum_d(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) p.partial[blockIdx.x] = (float)(red[0] + red[1] + red[2] + red[3]);
}
__global__ void mse_final_kernel(SdmiMseArgs p) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    double s = 0.0;
    for (int i = 0; i < p.nblk; ++i) s += (double)p.partial[i];
    const float m = (float)(s / (double)p.n);
    p.out[0] = p.oscale != 0.f ? m * p.oscale : m;
  }
}

}  // namespace

#define ST ((hipStream_t)stream)

extern "C" int sdmi_lincomb(const SdmiLincombArgs* a, void* stream) {
  SDMI_REQUIRE(a && a->y && a->n >= 0, "bad args");
  if (a->n == 0) return SDMI_OK;
  hipLaunchKernelGGL(lincomb_kernel, dim3(ew_blocks(a->n)), dim3(EW_THREADS), 0, ST, *a);
  return sdmi_check_launch("lincomb");
}
extern "C" int sdmi_row_lincomb(const SdmiRowLincombArgs* a, void* stream) {
  SDMI_REQUIRE(a && a->y && a->x0 && a->x1 && a->ca && a->cb, "null pointer");
  hipLaunchKernelGGL(row_lincomb_kernel, dim3(ew_blocks((long long)a->B * a->per)),
                     dim3(EW_THREADS), 0, ST, *a);
  return sdmi_check_launch("row_lincomb");
}
extern "C" int sdmi_nchw_to_nhwc(const SdmiNchwToNhwcArgs* a, void* stream) {
  SDMI_REQUIRE(a && a->src && a->dst && a->Cpad >= a->C, "bad args");
  const long long n = (long long)a->B * a->H * a->W * a->Cpad;
  if (a->dtype == SDMI_BF16)
    hipLaunchKernelGGL(nchw_to_nhwc_kernel<bf16_t>, dim3(ew_blocks(n)), dim3(EW_THREADS), 0, ST, *a);
  else
    hipLaunchKernelGGL(nchw_to_nhwc_kernel<float>, dim3(ew_blocks(n)), dim3(EW_THREADS), 0, ST, *a);
  return sdmi_check_launch("nchw_to_nhwc");
}
extern "C" int sdmi_nhwc_to_nchw(const SdmiNhwcToNchwArgs* a, void* stream) {
  SDMI_REQUIRE(a && a->src && a->dst && a->Cpad >= a->C, "bad args");
  const long long n = (long long)a->B * a->H * a->W * a->C;
  if (a->dtype == SDMI_BF16)
    hipLaunchKernelGGL(nhwc_to_nchw_kernel<bf16_t>, dim3(ew_blocks(n)), dim3(EW_THREADS), 0, ST, *a);
  else
    hipLaunchKernelGGL(nhwc_to_nchw_kernel<float>, dim3(ew_blocks(n)), dim3(EW_THREADS), 0, ST, *a);
  return sdmi_check_launch("nhwc_to_nchw");
}
extern "C" int sdmi_cast2d(const SdmiCast2dArgs* a, void* stream) {
  SDMI_REQUIRE(a && a->src && a->dst, "null pointer");
  const int g = ew_blocks(a->rows * a->cols);
  const bool sb = a->src_dtype == SDMI_BF16, db = a->dst_dtype == SDMI_BF16;
  if (sb && db) hipLaunchKernelGGL((cast2d_kernel<bf16_t, bf16_t>), dim3(g), dim3(EW_THREADS), 0, ST, *a);
  else if (sb) hipLaunchKernelGGL((cast2d_kernel<bf16_t, float>), dim3(g), dim3(EW_THREADS), 0, ST, *a);
  else if (db) hipLaunchKernelGGL((cast2d_kernel<float, bf16_t>), dim3(g), dim3(EW_THREADS), 0, ST, *a);
  else hipLaunchKernelGGL((cast2d_kernel<float, float>), dim3(g), dim3(EW_THREADS), 0, ST, *a);
  return sdmi_check_launch("cast2d");
}
// dst[b][i] = src[b][idx[i]] over 2-byte elements: eight gathered elements per thread, one 16-byte store
__global__ __launch_bounds__(256) void gather_rows_kernel(SdmiGatherRowsArgs p) {
  const long long v = (long long)blockIdx.x * 256 + threadIdx.x;
  if (v * 8 >= p.n) return;
  const unsigned short* s = (const unsigned short*)p.src + (long long)blockIdx.y * p.s_src;
  const long long* ix = p.idx + v * 8;
  unsigned e[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) e[j] = s[ix[j]];
  *reinterpret_cast<uint4*>((unsigned short*)p.dst + (long long)blockIdx.y * p.s_dst + v * 8) =
      make_uint4(e[0] | (e[1] << 16), e[2] | (e[3] << 16), e[4] | (e[5] << 16), e[6] | (e[7] << 16));
}
extern "C" int sdmi_gather_rows(const SdmiGatherRowsArgs* a, void* stream) {
  SDMI_REQUIRE(a && a->src && a->dst && a->idx, "null pointer");
  SDMI_REQUIRE(a->B >= 1 && a->n >= 8 && a->n % 8 == 0 && a->s_dst % 8 == 0 && ((uintptr_t)a->dst & 15) == 0,
               "n and the destination pitch are multiples of eight 2-byte elements");
  hipLaunchKernelGGL(gather_rows_kernel, dim3((unsigned)((a->n / 8 + 255) / 256), a->B), dim3(256), 0, ST, *a);
  return sdmi_check_launch("gather_rows");
}
extern "C" int sdmi_expand_heads(const SdmiExpandHeadsArgs* a, void* stream) {
  SDMI_REQUIRE(a && a->kv && a->kexp && a->vexp, "null pointer");
  SDMI_REQUIRE((a->gw == 0 || a->gw == 8 || a->gw == 16) && a->S >= 1 && a->S <= (a->gw == 16 ? 16 : 8) && a->heads > 0 &&
                   a->C % a->heads == 0, "1 .. gw slots (gw = 8 or 16), C = heads * head_dim");
  const int g = ew_blocks((long long)a->B * a->heads * (a->gw == 16 ? 16 : 8) * a->C);
  if (a->dtype == SDMI_BF16) hipLaunchKernelGGL(expand_heads_kernel<bf16_t>, dim3(g), dim3(EW_THREADS), 0, ST, *a);
  else hipLaunchKernelGGL(expand_heads_kernel<float>, dim3(g), dim3(EW_THREADS), 0, ST, *a);
  return sdmi_check_launch("expand_heads");
}
extern "C" int sdmi_quant_fp8(const SdmiQuantFp8Args* a, void* stream) {
  SDMI_REQUIRE(a && a->src && a->dst, "null pointer");
  SDMI_REQUIRE(a->ldd % 16 == 0 && a->ldd >= a->cols && ((uintptr_t)a->dst & 15) == 0, "dst rows are 16-byte vectors");
  SDMI_REQUIRE(a->src_dtype == SDMI_F32 || a->src_dtype == SDMI_BF16, "bad src_dtype");
  const int g = ew_blocks(a->rows * (a->ldd / 16));
  if (a->src_dtype == SDMI_BF16) hipLaunchKernelGGL(quant_fp8_kernel<bf16_t>, dim3(g), dim3(EW_THREADS), 0, ST, *a);
  else hipLaunchKernelGGL(quant_fp8_kernel<float>, dim3(g), dim3(EW_THREADS), 0, ST, *a);
  return sdmi_check_launch("quant_fp8");
}
extern "C" int sdmi_fp8_quant_group(const SdmiFp8GroupArgs* a, void* stream) {
  SDMI_REQUIRE(a && a->descs && a->amax_bits && a->inv_scale && a->n_desc >= 1 && a->total_blocks >= 1, "bad args");
  SDMI_REQUIRE(a->src_dtype == SDMI_F32 || a->src_dtype == SDMI_BF16, "bad src_dtype");
  if (hipMemsetAsync(a->amax_bits, 0, sizeof(unsigned) * (size_t)a->n_desc, ST) != hipSuccess) {
    sdmi_set_error("fp8_quant_group: hipMemsetAsync failed");
    return SDMI_ELAUNCH;
  }
  if (a->src_dtype == SDMI_BF16) {
    hipLaunchKernelGGL(fp8_group_amax_kernel<bf16_t>, dim3(a->total_blocks), dim3(256), 0, ST, *a);
    hipLaunchKernelGGL(fp8_group_quant_kernel<bf16_t>, dim3(a->total_blocks), dim3(256), 0, ST, *a);
  } else {
    hipLaunchKernelGGL(fp8_group_amax_kernel<float>, dim3(a->total_blocks), dim3(256), 0, ST, *a);
    hipLaunchKernelGGL(fp8_group_quant_kernel<float>, dim3(a->total_blocks), dim3(256), 0, ST, *a);
  }
  return sdmi_check_launch("fp8_quant_group");
}
extern "C" int sdmi_memset0(const SdmiMemsetArgs* a, void* stream) {
  SDMI_REQUIRE(a && a->ptr && a->bytes >= 0, "bad args");
  if (a->bytes == 0) return SDMI_OK;
  if (hipMemsetAsync(a->ptr, 0, (size_t)a->bytes, ST) != hipSuccess) {
    sdmi_set_error("memset0: hipMemsetAsync failed");
    return SDMI_ELAUNCH;
  }
  return SDMI_OK;
}
extern "C" int sdmi_scale_dev(const SdmiScaleDevArgs* a, void* stream) {
  SDMI_REQUIRE(a && a->x && a->y && a->s, "null pointer");
  if (a->dtype == SDMI_BF16)
    hipLaunchKernelGGL(scale_dev_kernel<bf16_t>, dim3(ew_blocks(a->n)), dim3(EW_THREADS), 0, ST, *a);
  else
    hipLaunchKernelGGL(scale_dev_kernel<float>, dim3(ew_blocks(a->n)), dim3(EW_THREADS), 0, ST, *a);
  return sdmi_check_launch("scale_dev");
}
extern "C" int sdmi_counters_inc(const SdmiCountersArgs* a, void* stream) {
  SDMI_REQUIRE(a && (a->step || a->seed), "null pointer");
  hipLaunchKernelGGL(counters_inc_kernel, dim3(1), dim3(64), 0, ST, *a);
  return sdmi_check_launch("counters_inc");
}
extern "C" int sdmi_draw_tn(const SdmiDrawTnArgs* a, void* stream) {
  SDMI_REQUIRE(a && a->t && a->noise && a->B > 0 && a->T > 0 && a->per > 0 && a->per % 4 == 0, "bad args");
  SDMI_REQUIRE((!a->ca && !a->cb) || (a->tab_a && a->tab_b), "coefficient tables missing");
  hipLaunchKernelGGL(draw_tn_kernel, dim3(ew_blocks((long long)a->B * a->per + a->B)), dim3(EW_THREADS),
                     0, ST, *a);
  return sdmi_check_launch("draw_tn");
}
extern "C" int sdmi_timestep_embedding(const SdmiTimeEmbArgs* a, void* stream) {
  SDMI_REQUIRE(a && a->t && a->out && a->dim % 2 == 0, "bad args");
  hipLaunchKernelGGL(time_emb_kernel, dim3(ew_blocks((long long)a->B * a->dim)), dim3(EW_THREADS),
                     0, ST, *a);
  return sdmi_check_launch("timestep_embedding");
}
extern "C" int sdmi_act(const SdmiActArgs* a, void* stream) {
  SDMI_REQUIRE(a && a->x && a->y, "null pointer");
  const int g = ew_blocks(a->n);
  const bool sb = a->src_dtype == SDMI_BF16, db = a->dst_dtype == SDMI_BF16;
  if (sb && db) hipLaunchKernelGGL((act_kernel<bf16_t, bf16_t>), dim3(g), dim3(EW_THREADS), 0, ST, *a);
  else if (sb) hipLaunchKernelGGL((act_kernel<bf16_t, float>), dim3(g), dim3(EW_THREADS), 0, ST, *a);
  else if (db) hipLaunchKernelGGL((act_kernel<float, bf16_t>), dim3(g), dim3(EW_THREADS), 0, ST, *a);
  else hipLaunchKernelGGL((act_kernel<float, float>), dim3(g), dim3(EW_THREADS), 0, ST, *a);
  return sdmi_check_launch("act");
}
extern "C" int sdmi_geglu(const SdmiGegluArgs* a, void* stream) {
  SDMI_REQUIRE(a && a->h && a->y, "null pointer");
  const int vec = a->dtype == SDMI_BF16 ? 8 : 4;
  SDMI_REQUIRE(a->C % vec == 0, "C must be a multiple of the vector width");
  const int g = ew_blocks(a->rows * (a->C / vec));
  if (a->dtype == SDMI_BF16) hipLaunchKernelGGL(geglu_kernel<bf16_t>, dim3(g), dim3(EW_THREADS), 0, ST, *a);
  else hipLaunchKernelGGL(geglu_kernel<float>, dim3(g), dim3(EW_THREADS), 0, ST, *a);
  return sdmi_check_launch("geglu");
}
extern "C" int sdmi_geglu_bwd(const SdmiGegluBwdArgs* a, void* stream) {
  SDMI_REQUIRE(a && a->h && a->dy && a->dh, "null pointer");
  const int vec = a->dtype == SDMI_BF16 ? 8 : 4;
  SDMI_REQUIRE(a->C % vec == 0, "C must be a multiple of the vector width");
  const int g = ew_blocks(a->rows * (a->C / vec));
  if (a->dtype == SDMI_BF16) hipLaunchKernelGGL(geglu_bwd_kernel<bf16_t>, dim3(g), dim3(EW_THREADS), 0, ST, *a);
  else hipLaunchKernelGGL(geglu_bwd_kernel<float>, dim3(g), dim3(EW_THREADS), 0, ST, *a);
  return sdmi_check_launch("geglu_bwd");
}
extern "C" int sdmi_add_pos(const SdmiAddPosArgs* a, void* stream) {
  SDMI_REQUIRE(a && a->x && a->pos && a->y, "null pointer");
  const int vec = a->dtype == SDMI_BF16 ? 8 : 4;
  SDMI_REQUIRE(a->per % vec == 0, "per-image size must be a multiple of the vector width");
  const int g = ew_blocks((long long)a->B * (a->per / vec));
  if (a->dtype == SDMI_BF16) hipLaunchKernelGGL(add_pos_kernel<bf16_t>, dim3(g), dim3(EW_THREADS), 0, ST, *a);
  else hipLaunchKernelGGL(add_pos_kernel<float>, dim3(g), dim3(EW_THREADS), 0, ST, *a);
  return sdmi_check_launch("add_pos");
}
extern "C" int sdmi_broadcast_pos(const SdmiBroadcastPosArgs* a, void* stream) {
  SDMI_REQUIRE(a && a->x && a->pos && a->y, "null pointer");
  const int vec = a->dtype == SDMI_BF16 ? 8 : 4;
  SDMI_REQUIRE(a->C % vec == 0, "C must be a multiple of the vector width");
  const int g = ew_blocks((long long)a->G * a->R * (a->C / vec));
  if (a->dtype == SDMI_BF16) hipLaunchKernelGGL(broadcast_pos_kernel<bf16_t>, dim3(g), dim3(EW_THREADS), 0, ST, *a);
  else hipLaunchKernelGGL(broadcast_pos_kernel<float>, dim3(g), dim3(EW_THREADS), 0, ST, *a);
  return sdmi_check_launch("broadcast_pos");
}
extern "C" int sdmi_sa_combine(const SdmiSaCombineArgs* a, void* stream) {
  SDMI_REQUIRE(a && a->o && a->recon && a->masks && a->ldo >= 4 && a->N >= 1, "bad args");
  const int g = ew_blocks((long long)a->B * a->HW);
  if (a->dtype == SDMI_BF16) hipLaunchKernelGGL(sa_combine_kernel<bf16_t>, dim3(g), dim3(EW_THREADS), 0, ST, *a);
  else hipLaunchKernelGGL(sa_combine_kernel<float>, dim3(g), dim3(EW_THREADS), 0, ST, *a);
  return sdmi_check_launch("sa_combine");
}
extern "C" int sdmi_sa_combine_bwd(const SdmiSaCombineBwdArgs* a, void* stream) {
  SDMI_REQUIRE(a && a->o && a->masks && a->drecon && a->dout && a->ldo >= 4, "bad args");
  const int g = ew_blocks((long long)a->B * a->HW);
  if (a->dtype == SDMI_BF16) hipLaunchKernelGGL(sa_combine_bwd_kernel<bf16_t>, dim3(g), dim3(EW_THREADS), 0, ST, *a);
  else hipLaunchKernelGGL(sa_combine_bwd_kernel<float>, dim3(g), dim3(EW_THREADS), 0, ST, *a);
  return sdmi_check_launch("sa_combine_bwd");
}
extern "C" int sdmi_concat_channels(const SdmiConcatArgs* a, void* stream) {
  SDMI_REQUIRE(a && a->a && a->b && a->y, "null pointer");
  const int vec = a->dtype == SDMI_BF16 ? 8 : 4;
  SDMI_REQUIRE(a->Ca % vec == 0 && a->Cb % vec == 0, "channel counts must be vector multiples");
  const int g = ew_blocks(a->rows * ((a->Ca + a->Cb) / vec));
  if (a->dtype == SDMI_BF16) hipLaunchKernelGGL(concat_kernel<bf16_t>, dim3(g), dim3(EW_THREADS), 0, ST, *a);
  else hipLaunchKernelGGL(concat_kernel<float>, dim3(g), dim3(EW_THREADS), 0, ST, *a);
  return sdmi_check_launch("concat_channels");
}
extern "C" int sdmi_mask_upsample_argmax(const SdmiMaskUpArgs* a, void* stream) {
  SDMI_REQUIRE(a && a->seg && (a->up || a->idx), "null pointer");
  hipLaunchKernelGGL(mask_up_kernel, dim3(ew_blocks((long long)a->B * a->H * a->W)),
                     dim3(EW_THREADS), 0, ST, *a);
  return sdmi_check_launch("mask_upsample_argmax");
}
extern "C" int sdmi_mse(const SdmiMseArgs* a, void* stream) {
  SDMI_REQUIRE(a && a->pred && a->target && a->out && a->partial && a->nblk >= 1, "bad args");
  if (a->dtype == SDMI_BF16) hipLaunchKernelGGL(mse_kernel<bf16_t>, dim3(a->nblk), dim3(256), 0, ST, *a);
  else hipLaunchKernelGGL(mse_kernel<float>, dim3(a->nblk), dim3(256), 0, ST, *a);
  hipLaunchKernelGGL(mse_final_kernel, dim3(1), dim3(64), 0, ST, *a);
  return sdmi_check_launch("mse");
}
