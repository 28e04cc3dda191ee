// Multi-head attention with head_dim 32 (UNet self-attention over 16..256 tokens, slot
// cross-attention over 7..15 keys).  K and V of one (image, head) are staged once in LDS as fp32
// (Skv*32*4 B each); every lane owns one query (q and the output row live in registers) and walks
// the keys with an online softmax, reading K/V rows as LDS broadcasts.  fp32 math throughout.
#include "common.h"

namespace {

template <typename T, int HD>
__global__ __launch_bounds__(256) void attn_fwd_kernel(SdmiAttnArgs p) {
  constexpr int VEC = Elem<T>::VEC;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* Ks = reinterpret_cast<float*>(smem);
  float* Vs = Ks + p.Skv * HD;
  const int b = blockIdx.z, h = blockIdx.y;
  const int tid = threadIdx.x;
  // stage K, V
  {
    const int vecs_per_row = HD / VEC;
    const int total = p.Skv * vecs_per_row;
    const T* kb = (const T*)p.k + (long long)b * p.Skv * p.ldk + h * HD;
    const T* vb = (const T*)p.v + (long long)b * p.Skv * p.ldv + h * HD;
    for (int i = tid; i < total; i += blockDim.x) {
      const int row = i / vecs_per_row, c = (i % vecs_per_row) * VEC;
      float f[VEC];
      unpack16<T>(*reinterpret_cast<const uint4*>(kb + (long long)row * p.ldk + c), f);
#pragma unroll
      for (int j = 0; j < VEC; ++j) Ks[row * HD + c + j] = f[j];
      unpack16<T>(*reinterpret_cast<const uint4*>(vb + (long long)row * p.ldv + c), f);
#pragma unroll
      for (int j = 0; j < VEC; ++j) Vs[row * HD + c + j] = f[j];
    }
  }
  __syncthreads();
  const int qi = blockIdx.x * blockDim.x + tid;
  if (qi >= p.Sq) return;
  float q[HD], o[HD];
  {
    const T* qp = (const T*)p.q + ((long long)b * p.Sq + qi) * p.ldq + h * HD;
#pragma unroll
    for (int c = 0; c < HD; c += VEC) {
      float f[VEC];
      unpack16<T>(*reinterpret_cast<const uint4*>(qp + c), f);
#pragma unroll
      for (int j = 0; j < VEC; ++j) q[c + j] = f[j] * p.scale;
    }
  }
#pragma unroll
  for (int d = 0; d < HD; ++d) o[d] = 0.f;
  float m = -INFINITY, l = 0.f;
  for (int j0 = 0; j0 < p.Skv; j0 += 4) {
    float s[4];
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
      const int j = j0 + jj;
      if (j < p.Skv) {
        const f32x4* kr = reinterpret_cast<const f32x4*>(Ks + j * HD);
        float a = 0.f;
#pragma unroll
        for (int c = 0; c < HD / 4; ++c) {
          const f32x4 kv = kr[c];
          a += q[4 * c] * kv[0] + q[4 * c + 1] * kv[1] + q[4 * c + 2] * kv[2] + q[4 * c + 3] * kv[3];
        }
        s[jj] = a;
      } else {
        s[jj] = -INFINITY;
      }
    }
    const float mx = fmaxf(fmaxf(s[0], s[1]), fmaxf(s[2], s[3]));
    if (mx > m) {
      const float alpha = __expf(m - mx);
      l *= alpha;
#pragma unroll
      for (int d = 0; d < HD; ++d) o[d] *= alpha;
      m = mx;
    }
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
      const int j = j0 + jj;
      if (j < p.Skv) {
        const float pj = __expf(s[jj] - m);
        l += pj;
        const f32x4* vr = reinterpret_cast<const f32x4*>(Vs + j * HD);
#pragma unroll
        for (int c = 0; c < HD / 4; ++c) {
          const f32x4 vv = vr[c];
          o[4 * c] += pj * vv[0]; o[4 * c + 1] += pj * vv[1];
          o[4 * c + 2] += pj * vv[2]; o[4 * c + 3] += pj * vv[3];
        }
      }
    }
  }
  const float inv = 1.f / l;
  T* op = (T*)p.out + ((long long)b * p.Sq + qi) * p.ldo + h * HD;
#pragma unroll
  for (int c = 0; c < HD; c += VEC) {
    float f[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) f[j] = o[c + j] * inv;
    *reinterpret_cast<uint4*>(op + c) = pack16<T>(f);
  }
  if (p.lse) p.lse[((long long)b * p.heads + h) * p.Sq + qi] = m + __logf(l);
}

}  // namespace

extern "C" int sdmi_attention(const SdmiAttnArgs* a, void* stream) {
  SDMI_REQUIRE(a && a->q && a->k && a->v && a->out, "null pointer");
  const int vec = a->dtype == SDMI_BF16 ? 8 : 4;
  SDMI_REQUIRE(a->ldq % vec == 0 && a->ldk % vec == 0 && a->ldv % vec == 0 && a->ldo % vec == 0,
               "row pitches must keep 16-byte alignment");
  SDMI_REQUIRE(a->Skv >= 1 && a->Skv <= 400, "Skv must be in [1, 400] (K/V staged in LDS)");
  hipStream_t st = (hipStream_t)stream;
  int threads = ((a->Sq + 63) / 64) * 64;
  if (threads > 256) threads = 256;
  dim3 grid((a->Sq + threads - 1) / threads, a->heads, a->B);
  const int hd = a->head_dim > 0 ? a->head_dim : 32;
  SDMI_REQUIRE(hd == 32 || hd == 48, "head_dim must be 32 (UNet) or 48 (SAVi predictor)");
  const int smem = 2 * a->Skv * hd * 4;
#define ATTN_GO(T, HDV)                                                                      \
  do {                                                                                       \
    static bool done = false;                                                                \
    if (!done) {                                                                             \
      (void)hipFuncSetAttribute((const void*)attn_fwd_kernel<T, HDV>,                        \
                                hipFuncAttributeMaxDynamicSharedMemorySize,                   \
                                (2 * 512 * HDV * 4 < 160 * 1024 ? 2 * 512 * HDV * 4 : 160 * 1024)); \
      done = true;                                                                           \
    }                                                                                        \
    hipLaunchKernelGGL((attn_fwd_kernel<T, HDV>), grid, dim3(threads), smem, st, *a);        \
  } while (0)
  if (a->dtype == SDMI_BF16) { if (hd == 32) ATTN_GO(bf16_t, 32); else ATTN_GO(bf16_t, 48); }
  else { if (hd == 32) ATTN_GO(float, 32); else ATTN_GO(float, 48); }
#undef ATTN_GO
  return sdmi_check_launch("attention");
}
