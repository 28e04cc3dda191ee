// Shared device helpers for libsdmi (gfx950 only: wave64, MFMA, 160 KiB LDS).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <atomic>
#include "../../include/sdmi.h"

typedef unsigned short bf16_t;  // raw bf16 bits
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) void lds_void;   // LDS destination of buffer_load ... lds

void sdmi_set_error(const char* fmt, ...);
int sdmi_check_launch(const char* what);
// dynamic-LDS opt-in, once per (call site, device): `static std::atomic<unsigned long long> done{0};`
int sdmi_optin_lds(std::atomic<unsigned long long>& done, const void* fn, int bytes, const char* what);
#define SDMI_OPTIN_LDS(fn, bytes, what)                                                  \
  do {                                                                                   \
    static std::atomic<unsigned long long> sdmi_lds_done_{0};                            \
    const int sdmi_lds_rc_ = sdmi_optin_lds(sdmi_lds_done_, (const void*)(fn), (bytes), (what)); \
    if (sdmi_lds_rc_) return sdmi_lds_rc_;                                               \
  } while (0)

#define SDMI_REQUIRE(cond, msg)                       \
  do {                                                \
    if (!(cond)) {                                    \
      sdmi_set_error("%s: %s", __func__, msg);        \
      return SDMI_EINVAL;                             \
    }                                                 \
  } while (0)

__device__ __forceinline__ float bf16_to_f32(bf16_t h) {
  return __uint_as_float(((uint32_t)h) << 16);
}
typedef float sdmi_f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 sdmi_bf16x2 __attribute__((ext_vector_type(2)));
// round-to-nearest-even through gfx950's v_cvt_pk_bf16_f32 (one instruction per PAIR instead of the
// ~6 integer ops per value of the bit-twiddling form; same results for every finite input)
__device__ __forceinline__ bf16_t f32_to_bf16(float f) {
  return __builtin_bit_cast(bf16_t, (__bf16)f);
}
__device__ __forceinline__ uint32_t f32x2_to_bf16x2(float lo, float hi) {
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(sdmi_f32x2{lo, hi}, sdmi_bf16x2));
}

typedef unsigned char fp8_t;     // raw OCP e4m3fn bits (gfx950's native fp8)
// four floats -> four e4m3fn bytes (round to nearest even), saturating at +-448 (the cvt itself
// would produce NaN beyond the format's range)
__device__ __forceinline__ uint32_t f32x4_to_fp8x4(float a, float b, float c, float d) {
  a = fminf(fmaxf(a, -448.f), 448.f);
  b = fminf(fmaxf(b, -448.f), 448.f);
  c = fminf(fmaxf(c, -448.f), 448.f);
  d = fminf(fmaxf(d, -448.f), 448.f);
  int v = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, 0, false);
  v = __builtin_amdgcn_cvt_pk_fp8_f32(c, d, v, true);
  return (uint32_t)v;
}

template <typename T> struct Elem;
template <> struct Elem<float> {
  static constexpr int VEC = 4;  // elements per 16-byte vector
  __device__ static __forceinline__ float ld(const float* p) { return *p; }
  __device__ static __forceinline__ void st(float* p, float v) { *p = v; }
};
template <> struct Elem<bf16_t> {
  static constexpr int VEC = 8;
  __device__ static __forceinline__ float ld(const bf16_t* p) { return bf16_to_f32(*p); }
  __device__ static __forceinline__ void st(bf16_t* p, float v) { *p = f32_to_bf16(v); }
};

// 16-byte vector <-> VEC floats
template <typename T> __device__ __forceinline__ void unpack16(const uint4& v, float* f);
template <> __device__ __forceinline__ void unpack16<float>(const uint4& v, float* f) {
  f[0] = __uint_as_float(v.x); f[1] = __uint_as_float(v.y);
  f[2] = __uint_as_float(v.z); f[3] = __uint_as_float(v.w);
}
template <> __device__ __forceinline__ void unpack16<bf16_t>(const uint4& v, float* f) {
  f[0] = __uint_as_float(v.x << 16); f[1] = __uint_as_float(v.x & 0xffff0000u);
  f[2] = __uint_as_float(v.y << 16); f[3] = __uint_as_float(v.y & 0xffff0000u);
  f[4] = __uint_as_float(v.z << 16); f[5] = __uint_as_float(v.z & 0xffff0000u);
  f[6] = __uint_as_float(v.w << 16); f[7] = __uint_as_float(v.w & 0xffff0000u);
}
template <typename T> __device__ __forceinline__ uint4 pack16(const float* f);
template <> __device__ __forceinline__ uint4 pack16<float>(const float* f) {
  return make_uint4(__float_as_uint(f[0]), __float_as_uint(f[1]), __float_as_uint(f[2]),
                    __float_as_uint(f[3]));
}
template <> __device__ __forceinline__ uint4 pack16<bf16_t>(const float* f) {
  uint4 v;
  v.x = f32x2_to_bf16x2(f[0], f[1]);
  v.y = f32x2_to_bf16x2(f[2], f[3]);
  v.z = f32x2_to_bf16x2(f[4], f[5]);
  v.w = f32x2_to_bf16x2(f[6], f[7]);
  return v;
}

// erf without branches (both ranges evaluated, one select): minimax polynomials, the tail through v_exp_f32;
// < 1.5 ulp (N. Juffa's single-precision erff, max error 0.995 ulp with a correctly rounded exp).  The library
// erff is two divergent paths with a full-precision expf -- ~36 VALU instructions per element in the GEGLU
// epilogue of a GEMM whose K loop is four tiles long.
__device__ __forceinline__ float sdmi_erff(float a) {
  const float t = fabsf(a), s = a * a;
  float r = fmaf(-1.72853470e-5f, t, 3.83197126e-4f);
  const float u = fmaf(-3.88396438e-3f, t, 2.42546219e-2f);
  r = fmaf(r, s, u);
  r = fmaf(r, t, -1.06777877e-1f);
  r = fmaf(r, t, -6.34846687e-1f);
  r = fmaf(r, t, -1.28717512e-1f);
  r = fmaf(r, t, -t);
  r = copysignf(1.0f - __expf(r), a);
  float q = -5.96761703e-4f;
  q = fmaf(q, s, 4.99119423e-3f);
  q = fmaf(q, s, -2.67681349e-2f);
  q = fmaf(q, s, 1.12819925e-1f);
  q = fmaf(q, s, -3.76125336e-1f);
  q = fmaf(q, s, 1.28379166e-1f);
  q = fmaf(q, a, a);
  return t > 0.927734375f ? r : q;
}

// FAST = the value is rounded to bf16 afterwards (bf16 storage): SiLU through v_rcp_f32 (1 ulp) instead of the
// correctly rounded division (~11 VALU instructions per element: the single-pass GroupNorm + SiLU kernels were
// VALU bound at 2x the time of a copy of the same tensor) and the branch-free erf.  fp32 storage (the parity
// configuration) keeps the correctly rounded forms: the 20-step v-prediction sampler amplifies 1-ulp differences
// of every activation into its 1e-3 latent bound.
template <bool FAST = false>
__device__ __forceinline__ float act_apply(float x, int act) {
  switch (act) {
    case SDMI_ACT_RELU: return x > 0.f ? x : 0.f;
    case SDMI_ACT_SILU:
      if constexpr (FAST) return x * __builtin_amdgcn_rcpf(1.f + __expf(-x));
      else return x / (1.f + __expf(-x));
    case SDMI_ACT_GELU:
      if constexpr (FAST) return 0.5f * x * (1.f + sdmi_erff(x * 0.70710678118654752440f));
      else return 0.5f * x * (1.f + erff(x * 0.70710678118654752440f));
    default: return x;
  }
}
// derivative of act wrt its pre-activation input x
template <bool FAST = false>
__device__ __forceinline__ float act_grad(float x, int act) {
  switch (act) {
    case SDMI_ACT_RELU: return x > 0.f ? 1.f : 0.f;
    case SDMI_ACT_SILU: {
      float s;
      if constexpr (FAST) s = __builtin_amdgcn_rcpf(1.f + __expf(-x));
      else s = 1.f / (1.f + __expf(-x));
      return s * (1.f + x * (1.f - s));
    }
    case SDMI_ACT_GELU: {
      float c;
      if constexpr (FAST) c = 0.5f * (1.f + sdmi_erff(x * 0.70710678118654752440f));
      else c = 0.5f * (1.f + erff(x * 0.70710678118654752440f));
      return c + x * 0.39894228040143267794f * __expf(-0.5f * x * x);
    }
    default: return 1.f;
  }
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// ---- counter-based dropout shared by the fused GroupNorm kernels: one splitmix64 finaliser yields
// four 16-bit lots, element j of 16-byte vector `vi` is kept when its lot >= thr16 = p * 65536.
__device__ __forceinline__ unsigned long long sdmi_mix64(unsigned long long z) {
  z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL;
  z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL;
  return z ^ (z >> 31);
}
__device__ __forceinline__ unsigned long long sdmi_drop_seed(long long seed, const long long* seed_dev) {
  return (unsigned long long)seed + (seed_dev ? (unsigned long long)(*seed_dev) * 0x100000001b3ULL : 0ULL);
}
template <int VEC>
__device__ __forceinline__ void sdmi_drop_apply(float* f, unsigned long long seed, long long vi,
                                                unsigned thr16, float inv) {
#pragma unroll
  for (int h = 0; h < VEC / 4; ++h) {
    const unsigned long long r =
        sdmi_mix64(seed + 0x9e3779b97f4a7c15ULL * (unsigned long long)(vi * (VEC / 4) + h + 1));
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const unsigned lot = (unsigned)(r >> (16 * j)) & 0xffffu;
      f[h * 4 + j] = lot >= thr16 ? f[h * 4 + j] * inv : 0.f;
    }
  }
}

static inline int ceil_div(long long a, long long b) { return (int)((a + b - 1) / b); }
