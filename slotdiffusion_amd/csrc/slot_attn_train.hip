// Training-mode Slot Attention pieces (include/sdmi.h: sdmi_sa_attend_fwd / _bwd, sdmi_gru_gates*).
// The inference kernel (slot_attn.hip) fuses all iterations; for training each iteration's
// streaming attention pass is a kernel with a matching backward, and the tiny [B*N, D] mat-vecs
// (q projection, GRU, MLP) run on the GEMM kernel so their weight gradients come from sdmi_wgrad.
//
// forward (one iteration):   L[m][n] = q_n . k_m  (q pre-scaled by D^-1/2)
//     a = softmax_n(L);  w = a + eps;  den_n = sum_m w[m][n];  upd_n = sum_m w[m][n] v_m / den_n
// backward, given dupd:      c_n = dupd_n . upd_n
//     g[m][n] = dupd_n . v_m;  dw = (g - c_n) / den_n;  dL = a * (dw - sum_n' a dw)
//     dv_m = sum_n (w[m][n]/den_n) dupd_n;  dk_m = sum_n dL[m][n] q_n;  dq_n = sum_m dL[m][n] k_m
// One workgroup per image; wave w owns tokens w, w+8, ...; 64 lanes split the D channels.
#include "common.h"

namespace {

constexpr int SA_THREADS = 512;
constexpr int SA_WAVES = SA_THREADS / 64;

template <typename T, int NMAX, int DPL>
__global__ __launch_bounds__(SA_THREADS) void sa_attend_fwd_kernel(SdmiSaAttendArgs p) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int N = p.N, D = p.D;
  float* s_upd = sm;                 // [N][D]
  float* s_den = s_upd + NMAX * D;   // [NMAX]
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const T* kb = (const T*)p.k + (long long)b * p.M * p.ldkv;
  const T* vb = (const T*)p.v + (long long)b * p.M * p.ldkv;
  const float* qb = p.q + (long long)b * N * D;
  float qreg[NMAX][DPL];
#pragma unroll
  for (int n = 0; n < NMAX; ++n)
#pragma unroll
    for (int i = 0; i < DPL; ++i) {
      const int c = lane + 64 * i;
      qreg[n][i] = (n < N && c < D) ? qb[n * D + c] * p.scale : 0.f;
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
      const float a = lg[n] * inv;
      if (lane == n) mine = a;
      const float ae = a + p.eps;
      if (n < N) {
        den[n] += ae;
#pragma unroll
        for (int i = 0; i < DPL; ++i) upd[n][i] += ae * vx[i];
      }
    }
    if (lane < N) p.attn[((long long)b * p.M + m) * N + lane] = mine;
  }
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
  for (int i = tid; i < N * D; i += SA_THREADS)
    p.upd[(long long)b * N * D + i] = s_upd[i] / s_den[i / D];
  if (tid < N) p.den[b * N + tid] = s_den[tid];
}

template <typename T, int NMAX, int DPL>
__global__ __launch_bounds__(SA_THREADS) void sa_attend_bwd_kernel(SdmiSaAttendBwdArgs p) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int N = p.N, D = p.D;
  float* s_dq = sm;                  // [N][D]
  float* s_c = s_dq + NMAX * D;      // [NMAX]
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const T* kb = (const T*)p.k + (long long)b * p.M * p.ldkv;
  const T* vb = (const T*)p.v + (long long)b * p.M * p.ldkv;
  T* dkb = (T*)p.dk + (long long)b * p.M * p.ldkv;
  T* dvb = (T*)p.dv + (long long)b * p.M * p.ldkv;
  const float* qb = p.q + (long long)b * N * D;
  const float* ub = p.upd + (long long)b * N * D;
  const float* dub = p.dupd + (long long)b * N * D;
  // c_n = dupd_n . upd_n (one wave per slot)
  for (int n = wave; n < N; n += SA_WAVES) {
    float a = 0.f;
    for (int c = lane; c < D; c += 64) a += dub[n * D + c] * ub[n * D + c];
    a = wave_sum(a);
    if (lane == 0) s_c[n] = a;
  }
  __syncthreads();
  float qreg[NMAX][DPL], dureg[NMAX][DPL], dq[NMAX][DPL], cden[NMAX], iden[NMAX];
#pragma unroll
  for (int n = 0; n < NMAX; ++n) {
    iden[n] = n < N ? 1.f / p.den[b * N + n] : 0.f;
    cden[n] = n < N ? s_c[n] : 0.f;
#pragma unroll
    for (int i = 0; i < DPL; ++i) {
      const int c = lane + 64 * i;
      const bool ok = n < N && c < D;
      qreg[n][i] = ok ? qb[n * D + c] * p.scale : 0.f;
      dureg[n][i] = ok ? dub[n * D + c] : 0.f;
      dq[n][i] = 0.f;
    }
  }
  for (int m = wave; m < p.M; m += SA_WAVES) {
    float kx[DPL], vx[DPL];
#pragma unroll
    for (int i = 0; i < DPL; ++i) {
      const int c = lane + 64 * i;
      kx[i] = c < D ? Elem<T>::ld(kb + (long long)m * p.ldkv + c) : 0.f;
      vx[i] = c < D ? Elem<T>::ld(vb + (long long)m * p.ldkv + c) : 0.f;
    }
    float a[NMAX], dw[NMAX];
    float dot = 0.f;
#pragma unroll
    for (int n = 0; n < NMAX; ++n) {
      float g = 0.f;
#pragma unroll
      for (int i = 0; i < DPL; ++i) g += dureg[n][i] * vx[i];
      g = wave_sum(g);
      a[n] = n < N ? p.attn[((long long)b * p.M + m) * N + n] : 0.f;
      dw[n] = (g - cden[n]) * iden[n];
      dot += a[n] * dw[n];
    }
    float dkx[DPL], dvx[DPL];
#pragma unroll
    for (int i = 0; i < DPL; ++i) dkx[i] = dvx[i] = 0.f;
#pragma unroll
    for (int n = 0; n < NMAX; ++n) {
      if (n < N) {
        const float dL = a[n] * (dw[n] - dot);
        const float wn = (a[n] + p.eps) * iden[n];
#pragma unroll
        for (int i = 0; i < DPL; ++i) {
          dkx[i] += dL * qreg[n][i];
          dvx[i] += wn * dureg[n][i];
          dq[n][i] += dL * kx[i];
        }
      }
    }
#pragma unroll
    for (int i = 0; i < DPL; ++i) {
      const int c = lane + 64 * i;
      if (c < D) {
        Elem<T>::st(dkb + (long long)m * p.ldkv + c, dkx[i]);
        Elem<T>::st(dvb + (long long)m * p.ldkv + c, dvx[i]);
      }
    }
  }
  for (int w = 0; w < SA_WAVES; ++w) {
    if (wave == w) {
#pragma unroll
      for (int n = 0; n < NMAX; ++n)
        if (n < N) {
#pragma unroll
          for (int i = 0; i < DPL; ++i) {
            const int c = lane + 64 * i;
            if (c < D) s_dq[n * D + c] = (w == 0 ? 0.f : s_dq[n * D + c]) + dq[n][i];
          }
        }
    }
    __syncthreads();
  }
  // dq is wrt the UNSCALED q: L = scale * q.k
  for (int i = tid; i < N * D; i += SA_THREADS) p.dq[(long long)b * N * D + i] = s_dq[i] * p.scale;
}

__device__ __forceinline__ float sigm(float x) { return 1.f / (1.f + __expf(-x)); }

// GRUCell gate math on precomputed gi = W_ih x + b_ih, gh = W_hh h + b_hh ([R][3D] each, fp32)
__global__ void gru_gates_fwd_kernel(SdmiGruGatesArgs p) {
  const long long n = (long long)p.R * p.D;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (long long)gridDim.x * blockDim.x) {
    const long long r = i / p.D;
    const int c = (int)(i - r * p.D);
    const float* gi = p.gi + r * 3 * p.D;
    const float* gh = p.gh + r * 3 * p.D;
    const float rg = sigm(gi[c] + gh[c]);
    const float z = sigm(gi[p.D + c] + gh[p.D + c]);
    const float nn = tanhf(gi[2 * p.D + c] + rg * gh[2 * p.D + c]);
    p.hout[i] = (1.f - z) * nn + z * p.h[i];
  }
}
__global__ void gru_gates_bwd_kernel(SdmiGruGatesBwdArgs p) {
  const long long n = (long long)p.R * p.D;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (long long)gridDim.x * blockDim.x) {
    const long long r = i / p.D;
    const int c = (int)(i - r * p.D);
    const float* gi = p.gi + r * 3 * p.D;
    const float* gh = p.gh + r * 3 * p.D;
    float* dgi = p.dgi + r * 3 * p.D;
    float* dgh = p.dgh + r * 3 * p.D;
    const float rg = sigm(gi[c] + gh[c]);
    const float z = sigm(gi[p.D + c] + gh[p.D + c]);
    const float ghn = gh[2 * p.D + c];
    const float nn = tanhf(gi[2 * p.D + c] + rg * ghn);
    const float dh = p.dhout[i];
    const float dn = dh * (1.f - z);
    const float dz = dh * (p.h[i] - nn);
    const float dpre_n = dn * (1.f - nn * nn);
    const float dr = dpre_n * ghn;
    const float dpre_r = dr * rg * (1.f - rg);
    const float dpre_z = dz * z * (1.f - z);
    dgi[c] = dpre_r; dgh[c] = dpre_r;
    dgi[p.D + c] = dpre_z; dgh[p.D + c] = dpre_z;
    dgi[2 * p.D + c] = dpre_n; dgh[2 * p.D + c] = dpre_n * rg;
    p.dh[i] = dh * z;
  }
}

template <typename T, int NMAX, int DPL>
int launch_fwd(const SdmiSaAttendArgs& a, hipStream_t st) {
  const int smem = (NMAX * a.D + 16) * 4;
  hipLaunchKernelGGL((sa_attend_fwd_kernel<T, NMAX, DPL>), dim3(a.B), dim3(SA_THREADS), smem, st, a);
  return sdmi_check_launch("sa_attend_fwd");
}
template <typename T, int NMAX, int DPL>
int launch_bwd(const SdmiSaAttendBwdArgs& a, hipStream_t st) {
  const int smem = (NMAX * a.D + 16) * 4;
  hipLaunchKernelGGL((sa_attend_bwd_kernel<T, NMAX, DPL>), dim3(a.B), dim3(SA_THREADS), smem, st, a);
  return sdmi_check_launch("sa_attend_bwd");
}

}  // namespace

#define SA_DISPATCH(FN, a, st)                                                              \
  do {                                                                                      \
    const bool small = (a)->N <= 8;                                                         \
    const int dpl = ((a)->D + 63) / 64;                                                     \
    const bool bf = (a)->dtype == SDMI_BF16;                                                \
    if (dpl <= 2) {                                                                         \
      if (small) return bf ? FN<bf16_t, 8, 2>(*(a), st) : FN<float, 8, 2>(*(a), st);        \
      return bf ? FN<bf16_t, 16, 2>(*(a), st) : FN<float, 16, 2>(*(a), st);                 \
    }                                                                                       \
    if (dpl == 3) {                                                                         \
      if (small) return bf ? FN<bf16_t, 8, 3>(*(a), st) : FN<float, 8, 3>(*(a), st);        \
      return bf ? FN<bf16_t, 16, 3>(*(a), st) : FN<float, 16, 3>(*(a), st);                 \
    }                                                                                       \
    if (small) return bf ? FN<bf16_t, 8, 4>(*(a), st) : FN<float, 8, 4>(*(a), st);          \
    return bf ? FN<bf16_t, 16, 4>(*(a), st) : FN<float, 16, 4>(*(a), st);                   \
  } while (0)

extern "C" int sdmi_sa_attend_fwd(const SdmiSaAttendArgs* a, void* stream) {
  SDMI_REQUIRE(a && a->k && a->v && a->q && a->attn && a->upd && a->den, "null pointer");
  SDMI_REQUIRE(a->N >= 1 && a->N <= 16 && a->D <= 256, "N <= 16, D <= 256");
  hipStream_t st = (hipStream_t)stream;
  SA_DISPATCH(launch_fwd, a, st);
}
extern "C" int sdmi_sa_attend_bwd(const SdmiSaAttendBwdArgs* a, void* stream) {
  SDMI_REQUIRE(a && a->k && a->v && a->q && a->attn && a->upd && a->den && a->dupd && a->dq &&
                   a->dk && a->dv, "null pointer");
  SDMI_REQUIRE(a->N >= 1 && a->N <= 16 && a->D <= 256, "N <= 16, D <= 256");
  hipStream_t st = (hipStream_t)stream;
  SA_DISPATCH(launch_bwd, a, st);
}
extern "C" int sdmi_gru_gates(const SdmiGruGatesArgs* a, void* stream) {
  SDMI_REQUIRE(a && a->gi && a->gh && a->h && a->hout, "null pointer");
  const long long n = (long long)a->R * a->D;
  int blocks = (int)((n + 255) / 256);
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(gru_gates_fwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, *a);
  return sdmi_check_launch("gru_gates");
}
extern "C" int sdmi_gru_gates_bwd(const SdmiGruGatesBwdArgs* a, void* stream) {
  SDMI_REQUIRE(a && a->gi && a->gh && a->h && a->dhout && a->dgi && a->dgh && a->dh, "null pointer");
  const long long n = (long long)a->R * a->D;
  int blocks = (int)((n + 255) / 256);
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(gru_gates_bwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, *a);
  return sdmi_check_launch("gru_gates_bwd");
}
