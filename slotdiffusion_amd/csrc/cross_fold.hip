// Folded slot cross-attention of one transformer block in ONE launch (include/sdmi.h: sdmi_cross_fold; reference
// attention.py:182-206, 247-251: norm2 -> CrossAttention(context = slots) -> to_out + residual).
//
// With the 7 slot keys folded into the query projection and the values into the output projection
// (kern.Kern.cross_prepare) the layer is, per image b,
//     P   = softmax_over_each_head's_slots( LayerNorm(tok) Wq[b]^T + bq[b] )      [HW, R]      R = heads * 8
//     out = P W2[b]^T + bias + tok                                                [HW, C]
// -- two GEMMs whose weights are PER IMAGE (2 R C bf16 = 256 KB at C = 512), so at the low-resolution levels the
// layer is a weight stream with 16 - 64 token rows per image: as two batched igemm launches on 64 x 64 tiles it
// took 11.3 + 7.5 us at [64 images][16 tokens] with three quarters of every tile empty.  Here a workgroup owns 16
// tokens of one image: row statistics, scores^T = Wq[b] tok^T on 16x16x32 MFMAs with the weights as the A operand
// loaded straight from global memory (each byte is used once per workgroup: no LDS staging), the LayerNorm fold and
// the slot softmax in registers, probabilities through LDS, out^T = W2[b] P^T the same way, 8-byte stores of four
// consecutive channels of a token.
#include "common.h"

namespace {

typedef unsigned cf_u32x4 __attribute__((ext_vector_type(4)));
constexpr int CF_RMAX = 256;             // score columns (heads * 8) a workgroup's LDS image holds
constexpr int CF_PP = CF_RMAX + 8;       // row pitch of the probability image (bf16 elements): 16-byte aligned, skewed

__device__ __forceinline__ bf16x8 cf_ld16(const bf16_t* q) {
  return __builtin_bit_cast(bf16x8, *reinterpret_cast<const cf_u32x4*>(q));
}

// KS1 = C / 32, KS2 = R / 32: the K loops are unrolled with every operand load of a tile in flight before its MFMAs
// (a loop with a run-time trip count serialises load -> wait -> MFMA: one L2 round trip per 32 k)
template <int KS1, int KS2>
__global__ __launch_bounds__(256) void cross_fold_kernel(SdmiCrossFoldArgs p) {
  __shared__ float st[16][2];
  __shared__ __attribute__((aligned(16))) bf16_t P[16 * CF_PP];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int tiles = p.HW >> 4;
  const int b = blockIdx.x / tiles, t0 = (blockIdx.x - b * tiles) << 4;
  const bf16_t* tok = (const bf16_t*)p.tok + ((long long)b * p.HW + t0) * p.C;
  const bf16_t* wq = (const bf16_t*)p.wq + (long long)b * p.s_wq;
  const bf16_t* w2 = (const bf16_t*)p.w2 + (long long)b * p.s_w2;
  const int j = lane & 15, q = lane >> 4;          // MFMA operand row / output column; k group / output row group
  // ---- LayerNorm statistics of the 16 token rows (fp32, of the bf16 values): wave w takes rows 4w .. 4w + 3
  {
    float s4[4] = {0.f, 0.f, 0.f, 0.f}, q4[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
#pragma unroll
      for (int c0 = 0; c0 < KS1 * 32; c0 += 512) {
        const int c = c0 + lane * 8;
        if (c < KS1 * 32) {
          float f[8];
          unpack16<bf16_t>(*reinterpret_cast<const uint4*>(tok + (long long)(4 * w + rr) * p.C + c), f);
#pragma unroll
          for (int e = 0; e < 8; ++e) { s4[rr] += f[e]; q4[rr] = fmaf(f[e], f[e], q4[rr]); }
        }
      }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {              // eight interleaved butterflies
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {
        s4[rr] += __shfl_xor(s4[rr], o, 64);
        q4[rr] += __shfl_xor(q4[rr], o, 64);
      }
    }
    if (lane < 4) {
      const float s = lane == 0 ? s4[0] : lane == 1 ? s4[1] : lane == 2 ? s4[2] : s4[3];
      const float ss = lane == 0 ? q4[0] : lane == 1 ? q4[1] : lane == 2 ? q4[2] : q4[3];
      const float mean = s / (float)p.C;
      float var = ss / (float)p.C - mean * mean;
      if (var < 0.f) var = 0.f;
      st[4 * w + lane][0] = mean;
      st[4 * w + lane][1] = rsqrtf(var + p.ln_eps);
    }
  }
  __syncthreads();
  // ---- scores^T tile [16 score columns][16 tokens] = Wq[b] rows x tok^T, the norm folded in the epilogue
  const float mean = st[j][0], rstd = st[j][1];
  const float* colsum = p.colsum + (long long)b * p.s_colsum;
  const float* biasq = p.biasq + (long long)b * p.s_bias;
  bf16x8 tk[KS1];                                 // this lane's token fragments: shared by every score tile
  {
    const bf16_t* bp = tok + (long long)j * p.C + q * 8;
#pragma unroll
    for (int ks = 0; ks < KS1; ++ks) tk[ks] = cf_ld16(bp + ks * 32);
  }
  constexpr int NT1 = 2 * KS2, T1 = (NT1 + 3) / 4;          // score tiles in all / per wave (compile time: the tile
#pragma unroll                                            // loop unrolls and every tile's loads go out up front)
  for (int ti = 0; ti < T1; ++ti) {
    const int tile = w + 4 * ti;
    if (tile >= NT1) break;
    // (packed: fragment-major storage, a wave's load = 1 KB of consecutive memory; else 16 rows x 64 bytes)
    const bf16_t* ap = p.packed ? wq + ((long long)tile * KS1 * 64 + lane) * 8 : wq + (long long)(tile * 16 + j) * p.ld_wq + q * 8;
    const int astep = p.packed ? 512 : 32;
    bf16x8 a[KS1];
#pragma unroll
    for (int ks = 0; ks < KS1; ++ks) a[ks] = cf_ld16(ap + ks * astep);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < KS1; ++ks) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[ks], tk[ks], acc, 0, 0, 0);
    // lane: token j, score columns r0 .. r0 + 3 = slots (4 q & 4) .. + 3 of head (tile * 2 + (q >> 1))
    const int r0 = tile * 16 + 4 * q;
    float v[4], m = -INFINITY;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      v[e] = rstd * (acc[e] - mean * colsum[r0 + e]) + biasq[r0 + e];
      if (((4 * q + e) & 7) < p.slots) m = fmaxf(m, v[e]);
    }
    m = fmaxf(m, __shfl_xor(m, 16, 64));
    float sum = 0.f;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      v[e] = ((4 * q + e) & 7) < p.slots ? __expf(v[e] - m) : 0.f;
      sum += v[e];
    }
    sum += __shfl_xor(sum, 16, 64);
    const float inv = 1.f / sum;
    uint2 o;
    o.x = f32x2_to_bf16x2(v[0] * inv, v[1] * inv);
    o.y = f32x2_to_bf16x2(v[2] * inv, v[3] * inv);
    *reinterpret_cast<uint2*>(&P[j * CF_PP + r0]) = o;
  }
  __syncthreads();
  // ---- out^T tile [16 channels][16 tokens] = W2[b] rows x P^T, + bias + residual
  bf16_t* outp = (bf16_t*)p.out + ((long long)b * p.HW + t0) * p.C;
  bf16x8 pf[KS2];                                 // this lane's probability fragments
#pragma unroll
  for (int ks = 0; ks < KS2; ++ks) pf[ks] = cf_ld16(&P[j * CF_PP + ks * 32 + q * 8]);
  constexpr int T2 = KS1 / 2;                               // output tiles per wave (C / 16 / 4)
  bf16x8 a2[T2][KS2];
  uint2 rs2[T2];
  float4 b2[T2];
#pragma unroll
  for (int ti = 0; ti < T2; ++ti) {                         // every operand of the wave's tiles in flight together
    const int tile = w + 4 * ti;
    const bf16_t* ap = p.packed ? w2 + ((long long)tile * KS2 * 64 + lane) * 8 : w2 + (long long)(tile * 16 + j) * p.ld_w2 + q * 8;
    const int astep = p.packed ? 512 : 32;
#pragma unroll
    for (int ks = 0; ks < KS2; ++ks) a2[ti][ks] = cf_ld16(ap + ks * astep);
    const int n0 = tile * 16 + 4 * q;
    rs2[ti] = *reinterpret_cast<const uint2*>(tok + (long long)j * p.C + n0);
    b2[ti] = p.bias ? *reinterpret_cast<const float4*>(p.bias + n0) : float4{0.f, 0.f, 0.f, 0.f};
  }
#pragma unroll
  for (int ti = 0; ti < T2; ++ti) {
    const int tile = w + 4 * ti;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < KS2; ++ks) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a2[ti][ks], pf[ks], acc, 0, 0, 0);
    const int n0 = tile * 16 + 4 * q;
    const uint2 rs = rs2[ti];
    const float bi[4] = {b2[ti].x, b2[ti].y, b2[ti].z, b2[ti].w};
    uint2 o;
    o.x = f32x2_to_bf16x2(acc[0] + bi[0] + __uint_as_float(rs.x << 16), acc[1] + bi[1] + __uint_as_float(rs.x & 0xffff0000u));
    o.y = f32x2_to_bf16x2(acc[2] + bi[2] + __uint_as_float(rs.y << 16), acc[3] + bi[3] + __uint_as_float(rs.y & 0xffff0000u));
    *reinterpret_cast<uint2*>(outp + (long long)j * p.C + n0) = o;
  }
}

}  // namespace

extern "C" int sdmi_cross_fold(const SdmiCrossFoldArgs* a, void* stream) {
  SDMI_REQUIRE(a && a->tok && a->out && a->wq && a->w2 && a->colsum && a->biasq, "null pointer");
  SDMI_REQUIRE(a->B > 0 && a->HW > 0 && a->HW % 16 == 0 && a->C % 32 == 0 && a->C >= 32 && a->R % 32 == 0 && a->R >= 32 &&
                   a->R <= CF_RMAX && a->slots >= 1 && a->slots <= 8,
               "shape: HW % 16, C % 32, R % 32 (R <= 256), 1..8 slots per head");
  SDMI_REQUIRE(a->s_wq % 8 == 0 && a->s_w2 % 8 == 0 &&
                   (a->packed || (a->ld_wq % 8 == 0 && a->ld_w2 % 8 == 0 && a->ld_wq >= a->C && a->ld_w2 >= a->R)) && ((uintptr_t)a->wq & 15) == 0 && ((uintptr_t)a->w2 & 15) == 0 &&
                   ((uintptr_t)a->tok & 15) == 0 && ((uintptr_t)a->out & 15) == 0 && (!a->bias || ((uintptr_t)a->bias & 15) == 0),
               "operands are read in 16-byte vectors");
  const dim3 grid(a->B * (a->HW / 16));
  hipStream_t st = (hipStream_t)stream;
  if (a->C == 512 && a->R == 128) hipLaunchKernelGGL((cross_fold_kernel<16, 4>), grid, dim3(256), 0, st, *a);
  else if (a->C == 384 && a->R == 96) hipLaunchKernelGGL((cross_fold_kernel<12, 3>), grid, dim3(256), 0, st, *a);
  else if (a->C == 256 && a->R == 64) hipLaunchKernelGGL((cross_fold_kernel<8, 2>), grid, dim3(256), 0, st, *a);
  else if (a->C == 128 && a->R == 32) hipLaunchKernelGGL((cross_fold_kernel<4, 1>), grid, dim3(256), 0, st, *a);
  else {
    sdmi_set_error("cross_fold: (C, R) must be one of (512, 128), (384, 96), (256, 64), (128, 32)");
    return SDMI_EINVAL;
  }
  return sdmi_check_launch("cross_fold");
}
