// Fused Slot Attention: every iteration of the slot update for one image inside ONE workgroup.
// (include/sdmi.h: sdmi_slot_attention; reference loop img_based/models/sa_diffusion.py:40-68.)
//
// Per iteration:
//   1. q = Wq . LayerNorm(slots)                          [N][D]   (LDS resident)
//   2. one streaming pass over the image's M tokens: each wave takes tokens w, w+8, ...; its 64
//      lanes split the D channels of the k / v rows (coalesced row reads), the N logits are
//      reduced across the wave, softmax over the N slots is computed redundantly in every lane,
//      and sum_m (attn+eps) v  /  sum_m (attn+eps) are accumulated in registers in the same pass
//      (k and v are read exactly once per iteration; the [M][N] attention map never exists in
//      memory except for the last iteration's softmax, which IS the segmentation mask output).
//   3. GRUCell and the residual MLP on the [N][D] slot matrix (tiny mat-vecs from L2-resident
//      fp32 weights, activations in LDS).
// All math fp32; k/v may be bf16 or fp32.
#include "common.h"

namespace {

constexpr int SA_THREADS = 512;
constexpr int SA_WAVES = SA_THREADS / 64;

// out[n][o] = sum_k in[n][k] * W[o][k] (+ bias[o]), for n < N, o < O.  in/out in LDS, W global.
template <int NMAX>
__device__ __forceinline__ void small_matmul(float* out, int ldo, const float* in, int ldi,
                                             const float* __restrict__ W,
                                             const float* __restrict__ bias, int N, int O, int K) {
  for (int o = threadIdx.x; o < O; o += SA_THREADS) {
    float acc[NMAX];
    const float b = bias ? bias[o] : 0.f;
#pragma unroll
    for (int n = 0; n < NMAX; ++n) acc[n] = b;
    const f32x4* wr = reinterpret_cast<const f32x4*>(W + (long long)o * K);
    for (int k4 = 0; k4 < K / 4; ++k4) {
      const f32x4 w = wr[k4];
#pragma unroll
      for (int n = 0; n < NMAX; ++n) {
        if (n < N) {
          const f32x4 x = *reinterpret_cast<const f32x4*>(in + n * ldi + k4 * 4);
          acc[n] += x[0] * w[0] + x[1] * w[1] + x[2] * w[2] + x[3] * w[3];
        }
      }
    }
#pragma unroll
    for (int n = 0; n < NMAX; ++n)
      if (n < N) out[n * ldo + o] = acc[n];
  }
}

// LayerNorm rows of an [N][D] LDS matrix: wave w normalises rows w, w+8, ...
__device__ __forceinline__ void ln_rows(float* out, const float* in, const float* __restrict__ g,
                                        const float* __restrict__ be, int N, int D, float eps) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int n = wave; n < N; n += SA_WAVES) {
    float s = 0.f;
    for (int c = lane; c < D; c += 64) s += in[n * D + c];
    const float mean = wave_sum(s) / (float)D;
    float q = 0.f;
    for (int c = lane; c < D; c += 64) {
      const float d = in[n * D + c] - mean;
      q += d * d;
    }
    const float rstd = rsqrtf(wave_sum(q) / (float)D + eps);
    for (int c = lane; c < D; c += 64) out[n * D + c] = (in[n * D + c] - mean) * rstd * g[c] + be[c];
  }
}

__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + __expf(-x)); }

template <typename T, int NMAX, int DPL>
__global__ __launch_bounds__(SA_THREADS) void slot_attn_kernel(SdmiSlotAttnArgs p) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int N = p.N, D = p.D, Hd = p.Hid;
  float* s_slots = sm;                 // [N][D]
  float* s_a = s_slots + NMAX * D;     // [N][D]  LN output / q / LN for the MLP
  float* s_q = s_a + NMAX * D;         // [N][D]  q * scale
  float* s_upd = s_q + NMAX * D;       // [N][D]  updates
  float* s_den = s_upd + NMAX * D;     // [NMAX]
  float* s_gi = s_den + 16;            // [N][3D]
  float* s_gh = s_gi + NMAX * 3 * D;   // [N][3D]
  float* s_hid = s_gh + NMAX * 3 * D;  // [N][Hid]

  const int b = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const T* kb = (const T*)p.k + (long long)b * p.M * p.ldkv;
  const T* vb = (const T*)p.v + (long long)b * p.M * p.ldkv;
  const float* sin_ = p.slots_in + (long long)b * p.slots_bstride;
  for (int i = tid; i < N * D; i += SA_THREADS) s_slots[i] = sin_[i];
  __syncthreads();

  for (int it = 0; it < p.iters; ++it) {
    const bool last = it == p.iters - 1;
    // 1. q = Wq LN(slots), pre-scaled
    ln_rows(s_a, s_slots, p.lnq_g, p.lnq_b, N, D, 1e-5f);
    __syncthreads();
    small_matmul<NMAX>(s_q, D, s_a, D, p.wq, nullptr, N, D, D);
    __syncthreads();
    for (int i = tid; i < N * D; i += SA_THREADS) s_q[i] *= p.scale;
    __syncthreads();

    // 2. streaming attention pass
    float qreg[NMAX][DPL];
#pragma unroll
    for (int n = 0; n < NMAX; ++n)
#pragma unroll
      for (int i = 0; i < DPL; ++i) {
        const int c = lane + 64 * i;
        qreg[n][i] = (n < N && c < D) ? s_q[n * D + c] : 0.f;
      }
    float upd[NMAX][DPL], den[NMAX];
#pragma unroll
    for (int n = 0; n < NMAX; ++n) {
      den[n] = 0.f;
#pragma unroll
      for (int i = 0; i < DPL; ++i) upd[n][i] = 0.f;
    }
    for (int m = wave; m < p.M; m += SA_WAVES) {
      float kx[DPL], vx[DPL];
#pragma unroll
      for (int i = 0; i < DPL; ++i) {
        const int c = lane + 64 * i;
        kx[i] = c < D ? Elem<T>::ld(kb + (long long)m * p.ldkv + c) : 0.f;
        vx[i] = c < D ? Elem<T>::ld(vb + (long long)m * p.ldkv + c) : 0.f;
      }
      float lg[NMAX];
#pragma unroll
      for (int n = 0; n < NMAX; ++n) {
        float a = 0.f;
#pragma unroll
        for (int i = 0; i < DPL; ++i) a += kx[i] * qreg[n][i];
        lg[n] = wave_sum(a);
      }
      float mx = -INFINITY;
#pragma unroll
      for (int n = 0; n < NMAX; ++n)
        if (n < N) mx = fmaxf(mx, lg[n]);
      float se = 0.f;
#pragma unroll
      for (int n = 0; n < NMAX; ++n) {
        lg[n] = n < N ? __expf(lg[n] - mx) : 0.f;
        se += lg[n];
      }
      const float inv = 1.f / se;
      float mine = 0.f;
#pragma unroll
      for (int n = 0; n < NMAX; ++n) {
        const float a = lg[n] * inv;      // softmax over slots
        if (lane == n) mine = a;
        const float ae = a + p.eps;
        if (n < N) {
          den[n] += ae;
#pragma unroll
          for (int i = 0; i < DPL; ++i) upd[n][i] += ae * vx[i];
        }
      }
      if (last && lane < N) p.seg[((long long)b * p.M + m) * N + lane] = mine;
    }
    // cross-wave reduction of upd / den through LDS, wave by wave (deterministic order)
    for (int w = 0; w < SA_WAVES; ++w) {
      if (wave == w) {
#pragma unroll
        for (int n = 0; n < NMAX; ++n) {
          if (n < N) {
#pragma unroll
            for (int i = 0; i < DPL; ++i) {
              const int c = lane + 64 * i;
              if (c < D) s_upd[n * D + c] = (w == 0 ? 0.f : s_upd[n * D + c]) + upd[n][i];
            }
            if (lane == 0) s_den[n] = (w == 0 ? 0.f : s_den[n]) + den[n];
          }
        }
      }
      __syncthreads();
    }
    for (int i = tid; i < N * D; i += SA_THREADS) s_upd[i] = s_upd[i] / s_den[i / D];
    __syncthreads();

    // 3. GRUCell(updates, slots_prev)
    small_matmul<NMAX>(s_gi, 3 * D, s_upd, D, p.w_ih, p.b_ih, N, 3 * D, D);
    small_matmul<NMAX>(s_gh, 3 * D, s_slots, D, p.w_hh, p.b_hh, N, 3 * D, D);
    __syncthreads();
    for (int i = tid; i < N * D; i += SA_THREADS) {
      const int n = i / D, c = i - n * D;
      const float* gi = s_gi + n * 3 * D;
      const float* gh = s_gh + n * 3 * D;
      const float r = sigmoidf_(gi[c] + gh[c]);
      const float z = sigmoidf_(gi[D + c] + gh[D + c]);
      const float nn = tanhf(gi[2 * D + c] + r * gh[2 * D + c]);
      s_slots[i] = (1.f - z) * nn + z * s_slots[i];
    }
    __syncthreads();
    // residual MLP
    ln_rows(s_a, s_slots, p.lnm_g, p.lnm_b, N, D, 1e-5f);
    __syncthreads();
    small_matmul<NMAX>(s_hid, Hd, s_a, D, p.w1, p.b1, N, Hd, D);
    __syncthreads();
    for (int i = tid; i < N * Hd; i += SA_THREADS) s_hid[i] = fmaxf(s_hid[i], 0.f);
    __syncthreads();
    small_matmul<NMAX>(s_a, D, s_hid, Hd, p.w2, p.b2, N, D, Hd);
    __syncthreads();
    for (int i = tid; i < N * D; i += SA_THREADS) s_slots[i] += s_a[i];
    __syncthreads();
  }
  float* so = p.slots_out + (long long)b * N * D;
  for (int i = tid; i < N * D; i += SA_THREADS) so[i] = s_slots[i];
}

template <typename T, int NMAX, int DPL>
int launch_sa(const SdmiSlotAttnArgs& a, hipStream_t st) {
  const int smem = (4 * NMAX * a.D + 16 + 2 * NMAX * 3 * a.D + NMAX * a.Hid) * 4;
  auto kern = slot_attn_kernel<T, NMAX, DPL>;
  SDMI_OPTIN_LDS(kern, 160 * 1024, "slot_attention");
  if (smem > 160 * 1024) {
    sdmi_set_error("slot_attention: LDS budget exceeded (%d B)", smem);
    return SDMI_EUNSUPPORTED;
  }
  hipLaunchKernelGGL(kern, dim3(a.B), dim3(SA_THREADS), smem, st, a);
  return sdmi_check_launch("slot_attention");
}

}  // namespace

extern "C" int sdmi_slot_attention(const SdmiSlotAttnArgs* a, void* stream) {
  SDMI_REQUIRE(a && a->k && a->v && a->slots_in && a->slots_out && a->seg, "null pointer");
  SDMI_REQUIRE(a->N >= 1 && a->N <= 16, "1 <= num_slots <= 16");
  SDMI_REQUIRE(a->D % 4 == 0 && a->D <= 256 && a->Hid % 4 == 0, "slot_size <= 256, multiples of 4");
  SDMI_REQUIRE(a->iters >= 1, "iters");
  hipStream_t st = (hipStream_t)stream;
  const bool small = a->N <= 8;
  const int dpl = (a->D + 63) / 64;
  const bool bf = a->dtype == SDMI_BF16;
#define SA_GO(NM, DP)                                                        \
  return bf ? launch_sa<bf16_t, NM, DP>(*a, st) : launch_sa<float, NM, DP>(*a, st)
  if (dpl <= 2) { if (small) { SA_GO(8, 2); } else { SA_GO(16, 2); } }
  if (dpl == 3) { if (small) { SA_GO(8, 3); } else { SA_GO(16, 3); } }
  if (small) { SA_GO(8, 4); } else { SA_GO(16, 4); }
#undef SA_GO
}
