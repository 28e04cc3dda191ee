// Backward of the head_dim-32 attention (include/sdmi.h: sdmi_attention_bwd).
//   P_ij = exp(scale*q_i.k_j - lse_i),  D_i = dout_i . out_i
//   dV_j = sum_i P_ij dout_i,  dS_ij = P_ij (dout_i . v_j - D_i)
//   dQ_i = scale * sum_j dS_ij k_j,   dK_j = scale * sum_i dS_ij q_i
// Two kernels, both recomputing P from the saved log-sum-exp (no [Sq][Skv] matrix in memory):
//   dq kernel : lane per query, K/V of the (image, head) staged in LDS (as the forward);
//   dkv kernel: self-attention  -> lane per key, Q/dO/lse/D staged in LDS;
//               slot cross-attention (Skv <= 16) -> one thread per (key, channel) with 32-lane
//               shuffles for the two dot products, so all lanes work although only 7..15 keys exist.
#include "common.h"

namespace {

template <typename T, int HD>
__device__ __forceinline__ void load_row32(const T* p, float* f) {
  constexpr int VEC = Elem<T>::VEC;
#pragma unroll
  for (int c = 0; c < HD; c += VEC) unpack16<T>(*reinterpret_cast<const uint4*>(p + c), f + c);
}
template <typename T, int HD>
__device__ __forceinline__ void store_row32(T* p, const float* f) {
  constexpr int VEC = Elem<T>::VEC;
#pragma unroll
  for (int c = 0; c < HD; c += VEC) *reinterpret_cast<uint4*>(p + c) = pack16<T>(f + c);
}

template <typename T, int HD>
__global__ __launch_bounds__(256) void attn_bwd_dq_kernel(SdmiAttnBwdArgs p) {
  constexpr int VEC = Elem<T>::VEC;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* Ks = reinterpret_cast<float*>(smem);
  float* Vs = Ks + p.Skv * HD;
  const int b = blockIdx.z, h = blockIdx.y, tid = threadIdx.x;
  {
    const int vpr = HD / VEC, total = p.Skv * vpr;
    const T* kb = (const T*)p.k + (long long)b * p.Skv * p.ldk + h * HD;
    const T* vb = (const T*)p.v + (long long)b * p.Skv * p.ldv + h * HD;
    for (int i = tid; i < total; i += blockDim.x) {
      const int row = i / vpr, c = (i % vpr) * VEC;
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
  float q[HD], dO[HD], o[HD], dq[HD];
  load_row32<T, HD>((const T*)p.q + ((long long)b * p.Sq + qi) * p.ldq + h * HD, q);
  load_row32<T, HD>((const T*)p.dout + ((long long)b * p.Sq + qi) * p.ldo + h * HD, dO);
  load_row32<T, HD>((const T*)p.out + ((long long)b * p.Sq + qi) * p.ldo + h * HD, o);
  float D = 0.f;
#pragma unroll
  for (int d = 0; d < HD; ++d) { D += dO[d] * o[d]; dq[d] = 0.f; q[d] *= p.scale; }
  const float lse = p.lse[((long long)b * p.heads + h) * p.Sq + qi];
  for (int j = 0; j < p.Skv; ++j) {
    float s = 0.f, dp = 0.f;
#pragma unroll
    for (int d = 0; d < HD; ++d) { s += q[d] * Ks[j * HD + d]; dp += dO[d] * Vs[j * HD + d]; }
    const float ds = __expf(s - lse) * (dp - D);
#pragma unroll
    for (int d = 0; d < HD; ++d) dq[d] += ds * Ks[j * HD + d];
  }
#pragma unroll
  for (int d = 0; d < HD; ++d) dq[d] *= p.scale;
  store_row32<T, HD>((T*)p.dq + ((long long)b * p.Sq + qi) * p.ldq + h * HD, dq);
}

// self-attention dK/dV: lane per key; queries (q, dO, lse, D) staged in LDS in chunks.
template <typename T, int HD>
__global__ __launch_bounds__(256) void attn_bwd_dkv_kernel(SdmiAttnBwdArgs p) {
  constexpr int VEC = Elem<T>::VEC;
  constexpr int QC = 64;                       // queries per staged chunk
  __shared__ float Qs[QC][HD], Os[QC][HD], Ts[QC][HD], Ls[QC], Ds[QC];
  const int b = blockIdx.z, h = blockIdx.y, tid = threadIdx.x;
  const int kj = blockIdx.x * blockDim.x + tid;
  const bool act = kj < p.Skv;
  float k[HD], v[HD], dk[HD], dv[HD];
  if (act) {
    load_row32<T, HD>((const T*)p.k + ((long long)b * p.Skv + kj) * p.ldk + h * HD, k);
    load_row32<T, HD>((const T*)p.v + ((long long)b * p.Skv + kj) * p.ldv + h * HD, v);
  }
#pragma unroll
  for (int d = 0; d < HD; ++d) { dk[d] = dv[d] = 0.f; if (!act) k[d] = v[d] = 0.f; }
  for (int q0 = 0; q0 < p.Sq; q0 += QC) {
    __syncthreads();
    // stage QC queries: each thread handles (row, vector) items
    const int vpr = HD / VEC;
    for (int i = tid; i < QC * vpr; i += blockDim.x) {
      const int r = i / vpr, c = (i % vpr) * VEC, qi = q0 + r;
      float fq[VEC], fo[VEC], fout[VEC];
      if (qi < p.Sq) {
        const long long ro = ((long long)b * p.Sq + qi);
        unpack16<T>(*reinterpret_cast<const uint4*>((const T*)p.q + ro * p.ldq + h * HD + c), fq);
        unpack16<T>(*reinterpret_cast<const uint4*>((const T*)p.dout + ro * p.ldo + h * HD + c), fo);
        unpack16<T>(*reinterpret_cast<const uint4*>((const T*)p.out + ro * p.ldo + h * HD + c), fout);
      } else {
#pragma unroll
        for (int j = 0; j < VEC; ++j) fq[j] = fo[j] = fout[j] = 0.f;
      }
#pragma unroll
      for (int j = 0; j < VEC; ++j) { Qs[r][c + j] = fq[j]; Os[r][c + j] = fo[j]; Ts[r][c + j] = fout[j]; }
    }
    __syncthreads();
    for (int r = tid; r < QC; r += blockDim.x) {       // D_r = dO_r . O_r, lse_r
      const int qi = q0 + r;
      float D = 0.f;
#pragma unroll
      for (int d = 0; d < HD; ++d) D += Os[r][d] * Ts[r][d];
      Ds[r] = D;
      Ls[r] = qi < p.Sq ? p.lse[((long long)b * p.heads + h) * p.Sq + qi] : INFINITY;
    }
    __syncthreads();
    const int qn = min(QC, p.Sq - q0);
    for (int r = 0; r < qn; ++r) {
      float s = 0.f, dp = 0.f;
#pragma unroll
      for (int d = 0; d < HD; ++d) { s += Qs[r][d] * k[d]; dp += Os[r][d] * v[d]; }
      const float pij = __expf(s * p.scale - Ls[r]);
      const float ds = pij * (dp - Ds[r]);
#pragma unroll
      for (int d = 0; d < HD; ++d) { dv[d] += pij * Os[r][d]; dk[d] += ds * Qs[r][d]; }
    }
  }
  if (!act) return;
#pragma unroll
  for (int d = 0; d < HD; ++d) dk[d] *= p.scale;
  store_row32<T, HD>((T*)p.dk + ((long long)b * p.Skv + kj) * p.ldk + h * HD, dk);
  store_row32<T, HD>((T*)p.dv + ((long long)b * p.Skv + kj) * p.ldv + h * HD, dv);
}

// cross-attention dK/dV (Skv <= 16): thread (j, d) = (tid/32, tid%32); blockDim = Skv*32.
template <typename T, int HD>
__global__ __launch_bounds__(512) void attn_bwd_dkv_small_kernel(SdmiAttnBwdArgs p) {
  const int b = blockIdx.y, h = blockIdx.x;
  const int j = threadIdx.x / HD, d = threadIdx.x % HD;
  const float kd = Elem<T>::ld((const T*)p.k + ((long long)b * p.Skv + j) * p.ldk + h * HD + d);
  const float vd = Elem<T>::ld((const T*)p.v + ((long long)b * p.Skv + j) * p.ldv + h * HD + d);
  const float* lse = p.lse + ((long long)b * p.heads + h) * p.Sq;
  float dk = 0.f, dv = 0.f;
  for (int i = 0; i < p.Sq; ++i) {
    const long long ro = (long long)b * p.Sq + i;
    const float qd = Elem<T>::ld((const T*)p.q + ro * p.ldq + h * HD + d);
    const float od = Elem<T>::ld((const T*)p.dout + ro * p.ldo + h * HD + d);
    const float outd = Elem<T>::ld((const T*)p.out + ro * p.ldo + h * HD + d);
    float s = qd * kd, dp = od * vd, D = od * outd;
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) {
      s += __shfl_xor(s, o, 64);
      dp += __shfl_xor(dp, o, 64);
      D += __shfl_xor(D, o, 64);
    }
    const float pij = __expf(s * p.scale - lse[i]);
    dv += pij * od;
    dk += pij * (dp - D) * qd;
  }
  Elem<T>::st((T*)p.dk + ((long long)b * p.Skv + j) * p.ldk + h * HD + d, dk * p.scale);
  Elem<T>::st((T*)p.dv + ((long long)b * p.Skv + j) * p.ldv + h * HD + d, dv);
}

template <typename T, int HD>
int launch_attn_bwd(const SdmiAttnBwdArgs& a, hipStream_t st) {
  int threads = ((a.Sq + 63) / 64) * 64;
  if (threads > 256) threads = 256;
  dim3 grid((a.Sq + threads - 1) / threads, a.heads, a.B);
  const int smem = 2 * a.Skv * HD * 4;
  static bool done = false;
  if (!done) {
    (void)hipFuncSetAttribute((const void*)attn_bwd_dq_kernel<T, HD>,
                              hipFuncAttributeMaxDynamicSharedMemorySize,
                              (2 * 512 * HD * 4 < 160 * 1024 ? 2 * 512 * HD * 4 : 160 * 1024));
    done = true;
  }
  hipLaunchKernelGGL((attn_bwd_dq_kernel<T, HD>), grid, dim3(threads), smem, st, a);
  if (a.Skv <= 16 && HD == 32) {
    hipLaunchKernelGGL((attn_bwd_dkv_small_kernel<T, HD>), dim3(a.heads, a.B), dim3(a.Skv * HD), 0,
                       st, a);
  } else {
    int kt = ((a.Skv + 63) / 64) * 64;
    if (kt > 256) kt = 256;
    dim3 g2((a.Skv + kt - 1) / kt, a.heads, a.B);
    hipLaunchKernelGGL((attn_bwd_dkv_kernel<T, HD>), g2, dim3(kt), 0, st, a);
  }
  return sdmi_check_launch("attention_bwd");
}

}  // namespace

extern "C" int sdmi_attention_bwd(const SdmiAttnBwdArgs* a, void* stream) {
  SDMI_REQUIRE(a && a->q && a->k && a->v && a->out && a->dout && a->lse && a->dq && a->dk && a->dv,
               "null pointer");
  const int vec = a->dtype == SDMI_BF16 ? 8 : 4;
  SDMI_REQUIRE(a->ldq % vec == 0 && a->ldk % vec == 0 && a->ldv % vec == 0 && a->ldo % vec == 0,
               "row pitches must keep 16-byte alignment");
  SDMI_REQUIRE(a->Skv >= 1 && a->Skv <= 400, "Skv must be in [1, 400]");
  const int hd = a->head_dim > 0 ? a->head_dim : 32;
  SDMI_REQUIRE(hd == 32 || hd == 48, "head_dim must be 32 or 48");
  hipStream_t st = (hipStream_t)stream;
  if (a->dtype == SDMI_BF16)
    return hd == 32 ? launch_attn_bwd<bf16_t, 32>(*a, st) : launch_attn_bwd<bf16_t, 48>(*a, st);
  return hd == 32 ? launch_attn_bwd<float, 32>(*a, st) : launch_attn_bwd<float, 48>(*a, st);
}
