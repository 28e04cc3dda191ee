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
// Two implementations:
//  * token-tiled (workspace given, N <= 8): one workgroup per tile of 128 (bf16) / 64 (fp32)
//    tokens, K and V tiles staged in LDS; phase A = per-token slot terms (a few threads per token),
//    phase B = the per-channel contractions over the tile's tokens; per-tile partials of the
//    token sums (upd, den, dq) are folded by a small finalize kernel.  B*M/128 workgroups fill
//    the chip; the first version below used B workgroups and one wave reduction per (token, slot).
//  * one workgroup per image; wave w owns tokens w, w+8, ...; 64 lanes split the D channels
//    (N up to 16, any D <= 256).
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
        T* dko = dkb + (long long)m * p.ldkv + c;
        T* dvo = dvb + (long long)m * p.ldkv + c;
        Elem<T>::st(dko, dkx[i] + (p.accumulate ? Elem<T>::ld(dko) : 0.f));
        Elem<T>::st(dvo, dvx[i] + (p.accumulate ? Elem<T>::ld(dvo) : 0.f));
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


// ==========================================================================================
// token-tiled kernels
// ==========================================================================================
// slots are padded to NP = 8 or 16 (template parameter)

template <typename T> struct SatCfg;
template <> struct SatCfg<bf16_t> { static constexpr int TM = 128; };
template <> struct SatCfg<float> { static constexpr int TM = 64; };

template <typename T>
__device__ __forceinline__ void sat_stage(const T* kb, const T* vb, int ldkv, int m0, int mvalid,
                                          int D, char* Kt, char* Vt, int PK) {
  constexpr int VEC = Elem<T>::VEC;
  constexpr int TM = SatCfg<T>::TM;
  const int cpr = D / VEC;                       // 16-byte pieces per row
  const u32x4 zero4 = {0u, 0u, 0u, 0u};
  for (int i = threadIdx.x; i < TM * cpr; i += 256) {
    const int row = i / cpr, c = i - row * cpr;
    const bool ok = row < mvalid;
    const long long go = (long long)(m0 + row) * ldkv + c * VEC;
    const u32x4 kv = ok ? *reinterpret_cast<const u32x4*>(kb + go) : zero4;
    const u32x4 vv = ok ? *reinterpret_cast<const u32x4*>(vb + go) : zero4;
    *reinterpret_cast<u32x4*>(Kt + row * PK + c * 16) = kv;
    *reinterpret_cast<u32x4*>(Vt + row * PK + c * 16) = vv;
  }
}

// partial dots of one token row (LDS, this thread's channel range) with N fp32 vectors in LDS
template <typename T, int NP>
__device__ __forceinline__ void sat_dots(const char* row, const float* vecs, int D, int c0, int cn,
                                         int N, float (&out)[NP]) {
  constexpr int VEC = Elem<T>::VEC;
#pragma unroll
  for (int n = 0; n < NP; ++n) out[n] = 0.f;
  for (int c = c0; c < c0 + cn; c += VEC) {
    float x[VEC];
    unpack16<T>(*reinterpret_cast<const uint4*>(row + c * sizeof(T)), x);
#pragma unroll
    for (int n = 0; n < NP; ++n) {
      if (n < N) {
        const float* qv = vecs + n * D + c;
#pragma unroll
        for (int j = 0; j < VEC; j += 4) {
          const f32x4 q4 = *reinterpret_cast<const f32x4*>(qv + j);
          out[n] += x[j] * q4[0] + x[j + 1] * q4[1] + x[j + 2] * q4[2] + x[j + 3] * q4[3];
        }
      }
    }
  }
}

template <typename T, int NP>
__global__ __launch_bounds__(256) void sat_fwd_kernel(SdmiSaAttendArgs p, int tiles) {
  constexpr int TM = SatCfg<T>::TM;
  constexpr int TPT = 256 / TM;                  // threads per token in phase A
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int N = p.N, D = p.D;
  const int PK = D * (int)sizeof(T) + 16;
  char* Kt = smem;
  char* Vt = Kt + TM * PK;
  float* q_s = reinterpret_cast<float*>(Vt + TM * PK);     // [N][D], pre-scaled
  float* w_s = q_s + NP * D;                           // [TM][NP]
  const int b = blockIdx.y, tile = blockIdx.x, tid = threadIdx.x;
  const int m0 = tile * TM;
  const int mvalid = min(TM, p.M - m0);
  const T* kb = (const T*)p.k + (long long)b * p.M * p.ldkv;
  const T* vb = (const T*)p.v + (long long)b * p.M * p.ldkv;
  sat_stage<T>(kb, vb, p.ldkv, m0, mvalid, D, Kt, Vt, PK);
  for (int i = tid; i < N * D; i += 256) q_s[i] = p.q[(long long)b * N * D + i] * p.scale;
  __syncthreads();
  {  // phase A: logits, softmax over slots, w = a + eps
    const int tok = tid / TPT, sub = tid % TPT;
    const int cn = D / TPT;
    float lg[NP];
    sat_dots<T, NP>(Kt + tok * PK, q_s, D, sub * cn, cn, N, lg);
#pragma unroll
    for (int n = 0; n < NP; ++n)
      for (int off = 1; off < TPT; off <<= 1) lg[n] += __shfl_xor(lg[n], off, 64);
    float mx = -INFINITY;
#pragma unroll
    for (int n = 0; n < NP; ++n)
      if (n < N) mx = fmaxf(mx, lg[n]);
    float se = 0.f;
#pragma unroll
    for (int n = 0; n < NP; ++n) {
      lg[n] = n < N ? __expf(lg[n] - mx) : 0.f;
      se += lg[n];
    }
    const float inv = 1.f / se;
    if (sub == 0) {
      const bool ok = tok < mvalid;
      float* ap = p.attn + ((long long)b * p.M + m0 + tok) * N;
#pragma unroll
      for (int n = 0; n < NP; ++n) {
        const float a = lg[n] * inv;
        if (ok && n < N) ap[n] = a;
        w_s[tok * NP + n] = (ok && n < N) ? a + p.eps : 0.f;
      }
    }
  }
  __syncthreads();
  // phase B: thread d owns channel d: upd_partial[n][d] = sum_m w[m][n] v[m][d]; the last N threads
  // (idle when D < 256, shared with a channel when D == 256): den
  float* wsu = p.workspace + ((long long)b * tiles + tile) * N * (D + 1);
  if (tid < D) {
    float acc[NP];
#pragma unroll
    for (int n = 0; n < NP; ++n) acc[n] = 0.f;
    const char* vcol = Vt + tid * sizeof(T);
    for (int m = 0; m < mvalid; ++m) {
      const float v = Elem<T>::ld(reinterpret_cast<const T*>(vcol + m * PK));
#pragma unroll
      for (int q = 0; q < NP / 4; ++q) {
        const f32x4 w4 = *reinterpret_cast<const f32x4*>(w_s + m * NP + 4 * q);
        acc[4 * q] += w4[0] * v; acc[4 * q + 1] += w4[1] * v;
        acc[4 * q + 2] += w4[2] * v; acc[4 * q + 3] += w4[3] * v;
      }
    }
#pragma unroll
    for (int n = 0; n < NP; ++n)
      if (n < N) wsu[n * D + tid] = acc[n];
  }
  if (tid >= 256 - N) {
    const int n = tid - (256 - N);
    float s = 0.f;
    for (int m = 0; m < mvalid; ++m) s += w_s[m * NP + n];
    wsu[N * D + n] = s;
  }
}

__global__ __launch_bounds__(256) void sat_fwd_finalize_kernel(SdmiSaAttendArgs p, int tiles) {
  __shared__ float den_s[16];
  const int b = blockIdx.x, N = p.N, D = p.D;
  const float* ws = p.workspace + (long long)b * tiles * N * (D + 1);
  if (threadIdx.x < N) {
    float s = 0.f;
    for (int t = 0; t < tiles; ++t) s += ws[(long long)t * N * (D + 1) + N * D + threadIdx.x];
    den_s[threadIdx.x] = s;
    p.den[b * N + threadIdx.x] = s;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < N * D; i += 256) {
    float s = 0.f;
    for (int t = 0; t < tiles; ++t) s += ws[(long long)t * N * (D + 1) + i];
    p.upd[(long long)b * N * D + i] = s / den_s[i / D];
  }
}

template <typename T, int NP>
__global__ __launch_bounds__(256) void sat_bwd_kernel(SdmiSaAttendBwdArgs p, int tiles) {
  constexpr int VEC = Elem<T>::VEC;
  constexpr int TM = SatCfg<T>::TM;
  constexpr int TPT = 256 / TM;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int N = p.N, D = p.D;
  const int PK = D * (int)sizeof(T) + 16;
  char* Kt = smem;
  char* Vt = Kt + TM * PK;
  float* q_s = reinterpret_cast<float*>(Vt + TM * PK);     // [NP][D] scaled q
  float* du_s = q_s + NP * D;                          // [NP][D] dupd
  float* dL_s = du_s + NP * D;                         // [TM][NP]
  float* wn_s = dL_s + TM * NP;                        // [TM][NP]
  float* c_s = wn_s + TM * NP;                         // [NP] c_n, [NP] 1/den_n
  const int b = blockIdx.y, tile = blockIdx.x, tid = threadIdx.x;
  const int m0 = tile * TM;
  const int mvalid = min(TM, p.M - m0);
  const T* kb = (const T*)p.k + (long long)b * p.M * p.ldkv;
  const T* vb = (const T*)p.v + (long long)b * p.M * p.ldkv;
  sat_stage<T>(kb, vb, p.ldkv, m0, mvalid, D, Kt, Vt, PK);
  for (int i = tid; i < NP * D; i += 256) {
    const bool ok = i < N * D;
    q_s[i] = ok ? p.q[(long long)b * N * D + i] * p.scale : 0.f;
    du_s[i] = ok ? p.dupd[(long long)b * N * D + i] : 0.f;
  }
  // c_n = dupd_n . upd_n : 32 lanes per slot, 8 slots per round
  for (int n = tid >> 5; n < NP; n += 8) {
    const int l = tid & 31;
    float a = 0.f;
    if (n < N)
      for (int c = l; c < D; c += 32)
        a += p.dupd[((long long)b * N + n) * D + c] * p.upd[((long long)b * N + n) * D + c];
    for (int off = 1; off < 32; off <<= 1) a += __shfl_xor(a, off, 64);
    if (l == 0) {
      c_s[n] = a;
      c_s[NP + n] = n < N ? 1.f / p.den[b * N + n] : 0.f;
    }
  }
  __syncthreads();
  {  // phase A: per-token slot terms dL[m][n], wn[m][n]
    const int tok = tid / TPT, sub = tid % TPT;
    const int cn = D / TPT;
    float g[NP];
    sat_dots<T, NP>(Vt + tok * PK, du_s, D, sub * cn, cn, N, g);
#pragma unroll
    for (int n = 0; n < NP; ++n)
      for (int off = 1; off < TPT; off <<= 1) g[n] += __shfl_xor(g[n], off, 64);
    if (sub == 0) {
      const bool ok = tok < mvalid;
      const float* ap = p.attn + ((long long)b * p.M + m0 + (ok ? tok : 0)) * N;
      float a[NP], dw[NP];
      float dot = 0.f;
#pragma unroll
      for (int n = 0; n < NP; ++n) {
        a[n] = (ok && n < N) ? ap[n] : 0.f;
        dw[n] = (g[n] - c_s[n]) * c_s[NP + n];
        dot += a[n] * dw[n];
      }
#pragma unroll
      for (int n = 0; n < NP; ++n) {
        dL_s[tok * NP + n] = a[n] * (dw[n] - dot);
        wn_s[tok * NP + n] = (ok && n < N) ? (a[n] + p.eps) * c_s[NP + n] : 0.f;
      }
    }
  }
  __syncthreads();
  // phase B: thread (chunk ch of CW channels, token lane tl): dk, dv rows out; dq partial.
  // CW = the 16-byte vector, or half of it for 16 padded slots in bf16 (register budget:
  // 3 * NP * CW values per thread)
  constexpr int CW = (NP == 16 && VEC == 8) ? 4 : VEC;
  const int CH = D / CW, TL = 256 / CH;
  const int ch = tid % CH, tl = tid / CH;
  float dq[NP][CW];
#pragma unroll
  for (int n = 0; n < NP; ++n)
#pragma unroll
    for (int j = 0; j < CW; ++j) dq[n][j] = 0.f;
  if (tl < TL) {
    float qr[NP][CW], dur[NP][CW];
#pragma unroll
    for (int n = 0; n < NP; ++n)
#pragma unroll
      for (int j = 0; j < CW; ++j) {
        qr[n][j] = q_s[n * D + ch * CW + j];
        dur[n][j] = du_s[n * D + ch * CW + j];
      }
    T* dkb = (T*)p.dk + (long long)b * p.M * p.ldkv;
    T* dvb = (T*)p.dv + (long long)b * p.M * p.ldkv;
    for (int m = tl; m < mvalid; m += TL) {
      float kx[CW], dkx[CW], dvx[CW];
      const T* kp = reinterpret_cast<const T*>(Kt + m * PK) + ch * CW;
#pragma unroll
      for (int j = 0; j < CW; ++j) kx[j] = Elem<T>::ld(kp + j);
      float dl[NP], wn[NP];
#pragma unroll
      for (int q = 0; q < NP / 4; ++q) {
        *reinterpret_cast<f32x4*>(dl + 4 * q) = *reinterpret_cast<const f32x4*>(dL_s + m * NP + 4 * q);
        *reinterpret_cast<f32x4*>(wn + 4 * q) = *reinterpret_cast<const f32x4*>(wn_s + m * NP + 4 * q);
      }
#pragma unroll
      for (int j = 0; j < CW; ++j) dkx[j] = dvx[j] = 0.f;
#pragma unroll
      for (int n = 0; n < NP; ++n)
#pragma unroll
        for (int j = 0; j < CW; ++j) {
          dkx[j] += dl[n] * qr[n][j];
          dvx[j] += wn[n] * dur[n][j];
          dq[n][j] += dl[n] * kx[j];
        }
      T* dkp = dkb + (long long)(m0 + m) * p.ldkv + ch * CW;
      T* dvp = dvb + (long long)(m0 + m) * p.ldkv + ch * CW;
#pragma unroll
      for (int j = 0; j < CW; ++j) {
        Elem<T>::st(dkp + j, dkx[j] + (p.accumulate ? Elem<T>::ld(dkp + j) : 0.f));
        Elem<T>::st(dvp + j, dvx[j] + (p.accumulate ? Elem<T>::ld(dvp + j) : 0.f));
      }
    }
  }
  __syncthreads();                   // K/V tiles are dead: reuse them for the dq fold
  float* red = reinterpret_cast<float*>(smem);             // [TL][NP][D]
  if (tl < TL) {
#pragma unroll
    for (int n = 0; n < NP; ++n)
#pragma unroll
      for (int j = 0; j < CW; ++j) red[(tl * NP + n) * D + ch * CW + j] = dq[n][j];
  }
  __syncthreads();
  float* wsq = p.workspace + ((long long)b * tiles + tile) * N * D;
  for (int i = tid; i < N * D; i += 256) {
    float s = 0.f;
    for (int l = 0; l < TL; ++l) s += red[l * NP * D + i];       // i = n*D + d, n < N <= NP
    wsq[i] = s;
  }
}

__global__ __launch_bounds__(256) void sat_bwd_finalize_kernel(SdmiSaAttendBwdArgs p, int tiles) {
  const int b = blockIdx.x, N = p.N, D = p.D;
  const float* ws = p.workspace + (long long)b * tiles * N * D;
  for (int i = threadIdx.x; i < N * D; i += 256) {
    float s = 0.f;
    for (int t = 0; t < tiles; ++t) s += ws[(long long)t * N * D + i];
    p.dq[(long long)b * N * D + i] = s * p.scale;      // wrt the UNSCALED q: L = scale * q.k
  }
}

template <typename T>
bool sat_ok(int N, int D, int ldkv) {
  constexpr int VEC = Elem<T>::VEC;
  constexpr int TPT = 256 / SatCfg<T>::TM;
  const int np = N <= 8 ? 8 : 16;
  const int tm = SatCfg<T>::TM;
  // LDS of the (larger) backward kernel must fit the 160 KB of a CU
  const int smem_bwd = 2 * tm * (D * (int)sizeof(T) + 16) + (2 * np * D + 2 * tm * np + 2 * np) * 4;
  return N <= 16 && D % (TPT * VEC) == 0 && D / VEC <= 256 && ldkv % VEC == 0 && D <= 256 &&
         smem_bwd <= 160 * 1024;
}
template <typename T, int NP>
int sat_launch_fwd(const SdmiSaAttendArgs& a, hipStream_t st) {
  constexpr int TM = SatCfg<T>::TM;
  const int tiles = (a.M + TM - 1) / TM;
  const int smem = 2 * TM * (a.D * (int)sizeof(T) + 16) + (NP * a.D + TM * NP) * 4;
  SDMI_OPTIN_LDS((sat_fwd_kernel<T, NP>), 160 * 1024, "sa_attend_fwd");
  hipLaunchKernelGGL((sat_fwd_kernel<T, NP>), dim3(tiles, a.B), dim3(256), smem, st, a, tiles);
  hipLaunchKernelGGL(sat_fwd_finalize_kernel, dim3(a.B), dim3(256), 0, st, a, tiles);
  return sdmi_check_launch("sa_attend_fwd (tiled)");
}
template <typename T, int NP>
int sat_launch_bwd(const SdmiSaAttendBwdArgs& a, hipStream_t st) {
  constexpr int TM = SatCfg<T>::TM;
  constexpr int VEC = Elem<T>::VEC;
  const int tiles = (a.M + TM - 1) / TM;
  int smem = 2 * TM * (a.D * (int)sizeof(T) + 16) + (2 * NP * a.D + 2 * TM * NP + 2 * NP) * 4;
  const int cw = (NP == 16 && VEC == 8) ? 4 : VEC;
  const int red = (256 / (a.D / cw)) * NP * a.D * 4;
  if (red > smem) smem = red;
  SDMI_OPTIN_LDS((sat_bwd_kernel<T, NP>), 160 * 1024, "sa_attend_bwd");
  hipLaunchKernelGGL((sat_bwd_kernel<T, NP>), dim3(tiles, a.B), dim3(256), smem, st, a, tiles);
  hipLaunchKernelGGL(sat_bwd_finalize_kernel, dim3(a.B), dim3(256), 0, st, a, tiles);
  return sdmi_check_launch("sa_attend_bwd (tiled)");
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
  if (a->workspace) {
    if (a->dtype == SDMI_BF16 && sat_ok<bf16_t>(a->N, a->D, a->ldkv)) return a->N <= 8 ? sat_launch_fwd<bf16_t, 8>(*a, st) : sat_launch_fwd<bf16_t, 16>(*a, st);
    if (a->dtype == SDMI_F32 && sat_ok<float>(a->N, a->D, a->ldkv)) return a->N <= 8 ? sat_launch_fwd<float, 8>(*a, st) : sat_launch_fwd<float, 16>(*a, st);
  }
  SA_DISPATCH(launch_fwd, a, st);
}
extern "C" int sdmi_sa_attend_bwd(const SdmiSaAttendBwdArgs* a, void* stream) {
  SDMI_REQUIRE(a && a->k && a->v && a->q && a->attn && a->upd && a->den && a->dupd && a->dq &&
                   a->dk && a->dv, "null pointer");
  SDMI_REQUIRE(a->N >= 1 && a->N <= 16 && a->D <= 256, "N <= 16, D <= 256");
  hipStream_t st = (hipStream_t)stream;
  if (a->workspace) {
    if (a->dtype == SDMI_BF16 && sat_ok<bf16_t>(a->N, a->D, a->ldkv)) return a->N <= 8 ? sat_launch_bwd<bf16_t, 8>(*a, st) : sat_launch_bwd<bf16_t, 16>(*a, st);
    if (a->dtype == SDMI_F32 && sat_ok<float>(a->N, a->D, a->ldkv)) return a->N <= 8 ? sat_launch_bwd<float, 8>(*a, st) : sat_launch_bwd<float, 16>(*a, st);
  }
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
