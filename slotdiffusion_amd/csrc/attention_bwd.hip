// Backward of the head_dim-32 attention (include/sdmi.h: sdmi_attention_bwd).
//   P_ij = exp(scale*q_i.k_j - lse_i),  D_i = dout_i . out_i
//   dV_j = sum_i P_ij dout_i,  dS_ij = P_ij (dout_i . v_j - D_i)
//   dQ_i = scale * sum_j dS_ij k_j,   dK_j = scale * sum_i dS_ij q_i
// bf16 / head_dim 32 runs on the matrix cores (attn_bwd_*_mfma_kernel at the end of this file);
// fp32 (parity mode), head_dim 48 and very long sequences use the VALU kernels:
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
  SDMI_OPTIN_LDS((attn_bwd_dq_kernel<T, HD>), (2 * 512 * HD * 4 < 160 * 1024 ? 2 * 512 * HD * 4 : 160 * 1024),
                 "attention_bwd");
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


// ------------------------------------------------------------------------------------------
// Matrix-core backward (bf16, head_dim 32).  Same transposed formulation as the forward kernel
// (attention.hip): a lane's accumulator registers belong to ONE column, so the softmax terms are
// per-lane, and a C-layout accumulator converted to bf16 is directly the B operand of the next
// MFMA (k-slot h*8 + jj*4 + i  <->  row 16*mm + 8*jj + 4*h + i of the 32-row block); the matching
// A operands (K^T, Q^T, dO^T) come from row-major LDS images through ds_read_b64_tr_b16.
//   dQ kernel : wave = 32 queries (columns), loop over key blocks
//       S^T = K Q^T, dP^T = V dO^T, dS^T = P^T o (dP^T - D_q), dQ^T += K^T dS^T
//   dKV kernel: wave = 32 keys (columns), loop over query blocks
//       S = Q K^T, dP = dO V^T, P = exp(S - lse_q), dV^T += dO^T P, dK^T += Q^T (P o (dP - D_q))
// Row operands read with ds_read_b128 live in an 80-byte-pitch image, transposed ones in a
// 64-byte-pitch image (both conflict free), so K (dQ kernel) and Q, dO (dKV kernel) are staged twice.
// ------------------------------------------------------------------------------------------
typedef short b_s16x4 __attribute__((ext_vector_type(4)));
typedef short b_s16x8 __attribute__((ext_vector_type(8)));
#define ATB_LDS_V4(p) ((__attribute__((address_space(3))) b_s16x4*)(p))
constexpr int P80 = 80, P64 = 64;

__device__ __forceinline__ unsigned bpack2(float lo, float hi) {
  return (unsigned)f32_to_bf16(lo) | ((unsigned)f32_to_bf16(hi) << 16);
}
__device__ __forceinline__ bf16x8 bpack8(const float* v) {
  const u32x4 w = {bpack2(v[0], v[1]), bpack2(v[2], v[3]), bpack2(v[4], v[5]), bpack2(v[6], v[7])};
  return __builtin_bit_cast(bf16x8, w);
}
// A operand (rows = d, k-slots = rows 16*mm.. of the block) out of a 64-byte-pitch row-major image
__device__ __forceinline__ bf16x8 tr_operand(const char* img_lane, int row0) {
  const char* q = img_lane + row0 * P64;
  const b_s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(ATB_LDS_V4(q));
  const b_s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(ATB_LDS_V4(q + 8 * P64));
  const b_s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
  return __builtin_bit_cast(bf16x8, v);
}
__device__ __forceinline__ float dot8(const u32x4& a, const u32x4& b) {
  float fa[8], fb[8];
  unpack16<bf16_t>(reinterpret_cast<const uint4&>(a), fa);
  unpack16<bf16_t>(reinterpret_cast<const uint4&>(b), fb);
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += fa[i] * fb[i];
  return s;
}
// store a transposed accumulator column (this lane's row `rp`, d = 8j + 4hh + i) scaled by `sc`
__device__ __forceinline__ void store_col(bf16_t* rp, const f32x16& acc, float sc) {
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    uint2 w;
    w.x = bpack2(acc[4 * j] * sc, acc[4 * j + 1] * sc);
    w.y = bpack2(acc[4 * j + 2] * sc, acc[4 * j + 3] * sc);
    *reinterpret_cast<uint2*>(rp + 8 * j) = w;
  }
}

// Keys (dQ body) / queries (dKV body) pass through LDS in chunks of `chunk` rows (a multiple of 32, at
// most ATB_CHUNK), so sequences beyond one LDS image (the 784 tokens of the 224^2 configs) take the same
// kernel: the accumulators stay in registers across chunks, only the staged images are replaced
// between two barriers.  Slot cross-attention (a handful of keys) uses short query chunks: the launch
// allocates the larger of the two bodies' images for every workgroup, and occupancy follows it.
constexpr int ATB_CHUNK = 512;

__device__ __forceinline__ void attn_bwd_dq_mfma_body(const SdmiAttnBwdArgs& p, char* smem, int bx, int chunk) {
  const int skv_pad = (p.Skv + 31) / 32 * 32;
  const int cap = skv_pad < chunk ? skv_pad : chunk;
  char* K80 = smem;
  char* K64 = K80 + cap * P80;
  char* V80 = K64 + cap * P64;
  const int b = blockIdx.z, h = blockIdx.y;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int q0 = (bx * 4 + wave) * 32;
  const bool active = q0 < p.Sq;          // idle waves still help staging and keep the barriers
  const int ql = lane & 31, hh = lane >> 5;
  const int qi = q0 + ql, qc = qi < p.Sq ? qi : p.Sq - 1;
  bf16x8 bq[2], bdo[2];
  float D;
  {
    const long long ro = (long long)b * p.Sq + qc;
    const bf16_t* qp = (const bf16_t*)p.q + ro * p.ldq + h * 32 + hh * 8;
    const bf16_t* dp_ = (const bf16_t*)p.dout + ro * p.ldo + h * 32 + hh * 8;
    const bf16_t* op = (const bf16_t*)p.out + ro * p.ldo + h * 32 + hh * 8;
    const u32x4 q0v = *reinterpret_cast<const u32x4*>(qp), q1v = *reinterpret_cast<const u32x4*>(qp + 16);
    const u32x4 d0v = *reinterpret_cast<const u32x4*>(dp_), d1v = *reinterpret_cast<const u32x4*>(dp_ + 16);
    const u32x4 o0v = *reinterpret_cast<const u32x4*>(op), o1v = *reinterpret_cast<const u32x4*>(op + 16);
    bq[0] = __builtin_bit_cast(bf16x8, q0v); bq[1] = __builtin_bit_cast(bf16x8, q1v);
    bdo[0] = __builtin_bit_cast(bf16x8, d0v); bdo[1] = __builtin_bit_cast(bf16x8, d1v);
    D = dot8(d0v, o0v) + dot8(d1v, o1v);
    D += __shfl_xor(D, 32, 64);
  }
  const float lse = p.lse[((long long)b * p.heads + h) * p.Sq + qc];
  f32x16 dq;
#pragma unroll
  for (int r = 0; r < 16; ++r) dq[r] = 0.f;
  const int g = lane >> 4, t = lane & 15;
  const char* kfrag = K80 + ql * P80 + hh * 16;
  const char* vfrag = V80 + ql * P80 + hh * 16;
  const char* ktr = K64 + (4 * hh + (t >> 2)) * P64 + ((g & 1) * 16 + (t & 3) * 4) * 2;
  const bf16_t* kbase = (const bf16_t*)p.k + (long long)b * p.Skv * p.ldk + h * 32;
  const bf16_t* vbase = (const bf16_t*)p.v + (long long)b * p.Skv * p.ldv + h * 32;
  for (int c0 = 0; c0 < skv_pad; c0 += chunk) {
    const int rows = skv_pad - c0 < chunk ? skv_pad - c0 : chunk;
    if (c0) __syncthreads();              // everyone is done with the previous chunk's images
    {
      const u32x4 zero4 = {0u, 0u, 0u, 0u};
      for (int i = tid; i < rows * 4; i += 256) {
        const int row = i >> 2, c = i & 3;
        const bool ok = c0 + row < p.Skv;
        const u32x4 kv = ok ? *reinterpret_cast<const u32x4*>(kbase + (long long)(c0 + row) * p.ldk + c * 8) : zero4;
        const u32x4 vv = ok ? *reinterpret_cast<const u32x4*>(vbase + (long long)(c0 + row) * p.ldv + c * 8) : zero4;
        *reinterpret_cast<u32x4*>(K80 + row * P80 + c * 16) = kv;
        *reinterpret_cast<u32x4*>(K64 + row * P64 + c * 16) = kv;
        *reinterpret_cast<u32x4*>(V80 + row * P80 + c * 16) = vv;
      }
    }
    __syncthreads();
    if (!active) continue;
    for (int kb = 0; kb < rows / 32; ++kb) {
      f32x16 s, dp;
#pragma unroll
      for (int r = 0; r < 16; ++r) s[r] = dp[r] = 0.f;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const u32x4 ak = *reinterpret_cast<const u32x4*>(kfrag + kb * 32 * P80 + ks * 32);
        const u32x4 av = *reinterpret_cast<const u32x4*>(vfrag + kb * 32 * P80 + ks * 32);
        s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ak), bq[ks], s, 0, 0, 0);
        dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, av), bdo[ks], dp, 0, 0, 0);
      }
      float ds[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int key = c0 + kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
        const float pr = key < p.Skv ? __expf(s[r] * p.scale - lse) : 0.f;
        ds[r] = pr * (dp[r] - D);
      }
#pragma unroll
      for (int mm = 0; mm < 2; ++mm)
        dq = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_operand(ktr, kb * 32 + 16 * mm),
                                                     bpack8(ds + 8 * mm), dq, 0, 0, 0);
    }
  }
  if (active && qi < p.Sq)
    store_col((bf16_t*)p.dq + ((long long)b * p.Sq + qi) * p.ldq + h * 32 + 4 * hh, dq, p.scale);
}

__device__ __forceinline__ void attn_bwd_dkv_mfma_body(const SdmiAttnBwdArgs& p, char* smem, int bx, int chunk) {
  const int sq_pad = (p.Sq + 31) / 32 * 32;
  const int cap = sq_pad < chunk ? sq_pad : chunk;
  char* Q80 = smem;
  char* Q64 = Q80 + cap * P80;
  char* O80 = Q64 + cap * P64;
  char* O64 = O80 + cap * P80;
  float* lse_s = reinterpret_cast<float*>(O64 + cap * P64);
  float* D_s = lse_s + cap;
  const int b = blockIdx.z, h = blockIdx.y;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int k0 = (bx * 4 + wave) * 32;
  const bool active = k0 < p.Skv;
  const int kl = lane & 31, hh = lane >> 5;
  const int kj = k0 + kl, kc = kj < p.Skv ? kj : p.Skv - 1;
  bf16x8 bk[2], bv[2];
  {
    const bf16_t* kp = (const bf16_t*)p.k + ((long long)b * p.Skv + kc) * p.ldk + h * 32 + hh * 8;
    const bf16_t* vp = (const bf16_t*)p.v + ((long long)b * p.Skv + kc) * p.ldv + h * 32 + hh * 8;
    bk[0] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(kp));
    bk[1] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(kp + 16));
    bv[0] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(vp));
    bv[1] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(vp + 16));
  }
  f32x16 dk, dv;
#pragma unroll
  for (int r = 0; r < 16; ++r) dk[r] = dv[r] = 0.f;
  const int g = lane >> 4, t = lane & 15;
  const char* qfrag = Q80 + kl * P80 + hh * 16;     // A rows = queries of the block
  const char* ofrag = O80 + kl * P80 + hh * 16;
  const int troff = (4 * hh + (t >> 2)) * P64 + ((g & 1) * 16 + (t & 3) * 4) * 2;
  const char* qtr = Q64 + troff;
  const char* otr = O64 + troff;
  const bf16_t* qb_ = (const bf16_t*)p.q + (long long)b * p.Sq * p.ldq + h * 32;
  const bf16_t* db = (const bf16_t*)p.dout + (long long)b * p.Sq * p.ldo + h * 32;
  const bf16_t* ob = (const bf16_t*)p.out + (long long)b * p.Sq * p.ldo + h * 32;
  for (int c0 = 0; c0 < sq_pad; c0 += chunk) {
    const int rows = sq_pad - c0 < chunk ? sq_pad - c0 : chunk;
    if (c0) __syncthreads();
    {
      const u32x4 zero4 = {0u, 0u, 0u, 0u};
      for (int i = tid; i < rows * 4; i += 256) {
        const int row = i >> 2, c = i & 3;
        const bool ok = c0 + row < p.Sq;
        const u32x4 qv = ok ? *reinterpret_cast<const u32x4*>(qb_ + (long long)(c0 + row) * p.ldq + c * 8) : zero4;
        const u32x4 dv4 = ok ? *reinterpret_cast<const u32x4*>(db + (long long)(c0 + row) * p.ldo + c * 8) : zero4;
        *reinterpret_cast<u32x4*>(Q80 + row * P80 + c * 16) = qv;
        *reinterpret_cast<u32x4*>(Q64 + row * P64 + c * 16) = qv;
        *reinterpret_cast<u32x4*>(O80 + row * P80 + c * 16) = dv4;
        *reinterpret_cast<u32x4*>(O64 + row * P64 + c * 16) = dv4;
      }
      // D_q = dO_q . O_q and lse_q; pad queries get lse = +inf (P = 0) and D = 0
      for (int r = tid; r < rows; r += 256) {
        const int qi = c0 + r;
        float d = 0.f, l = INFINITY;
        if (qi < p.Sq) {
#pragma unroll
          for (int c = 0; c < 4; ++c)
            d += dot8(*reinterpret_cast<const u32x4*>(db + (long long)qi * p.ldo + c * 8),
                      *reinterpret_cast<const u32x4*>(ob + (long long)qi * p.ldo + c * 8));
          l = p.lse[((long long)b * p.heads + h) * p.Sq + qi];
        }
        D_s[r] = d;
        lse_s[r] = l;
      }
    }
    __syncthreads();
    if (!active) continue;
    for (int qb = 0; qb < rows / 32; ++qb) {
      f32x16 s, dp;
#pragma unroll
      for (int r = 0; r < 16; ++r) s[r] = dp[r] = 0.f;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const u32x4 aq = *reinterpret_cast<const u32x4*>(qfrag + qb * 32 * P80 + ks * 32);
        const u32x4 ao = *reinterpret_cast<const u32x4*>(ofrag + qb * 32 * P80 + ks * 32);
        s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, aq), bk[ks], s, 0, 0, 0);
        dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ao), bv[ks], dp, 0, 0, 0);
      }
      float pr[16], ds[16];
#pragma unroll
      for (int j = 0; j < 4; ++j) {   // rows (queries) c0 + qb*32 + 8j + 4hh + i
        const f32x4 l4 = *reinterpret_cast<const f32x4*>(lse_s + qb * 32 + 8 * j + 4 * hh);
        const f32x4 d4 = *reinterpret_cast<const f32x4*>(D_s + qb * 32 + 8 * j + 4 * hh);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int r = 4 * j + i;
          pr[r] = __expf(s[r] * p.scale - l4[i]);
          ds[r] = pr[r] * (dp[r] - d4[i]);
        }
      }
#pragma unroll
      for (int mm = 0; mm < 2; ++mm) {
        dv = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_operand(otr, qb * 32 + 16 * mm),
                                                     bpack8(pr + 8 * mm), dv, 0, 0, 0);
        dk = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_operand(qtr, qb * 32 + 16 * mm),
                                                     bpack8(ds + 8 * mm), dk, 0, 0, 0);
      }
    }
  }
  if (active && kj < p.Skv) {
    store_col((bf16_t*)p.dk + ((long long)b * p.Skv + kj) * p.ldk + h * 32 + 4 * hh, dk, p.scale);
    store_col((bf16_t*)p.dv + ((long long)b * p.Skv + kj) * p.ldv + h * 32 + 4 * hh, dv, 1.f);
  }
}

// dq and dk / dv of one attention in ONE launch: workgroups [0, nqx) take query blocks, the rest key
// blocks (two dependent-free halves of the same backward: one launch instead of two)
__global__ __launch_bounds__(256) void attn_bwd_mfma_kernel(SdmiAttnBwdArgs p, int nqx, int chunk_k, int chunk_q) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  if ((int)blockIdx.x < nqx) attn_bwd_dq_mfma_body(p, smem, blockIdx.x, chunk_k);
  else attn_bwd_dkv_mfma_body(p, smem, (int)blockIdx.x - nqx, chunk_q);
}

int launch_attn_bwd_mfma(const SdmiAttnBwdArgs& a, hipStream_t st) {
  SDMI_OPTIN_LDS(attn_bwd_mfma_kernel, 160 * 1024, "attention_bwd (mfma)");
  int skv_pad = (a.Skv + 31) / 32 * 32, sq_pad = (a.Sq + 31) / 32 * 32;
  if (skv_pad > ATB_CHUNK) skv_pad = ATB_CHUNK;
  const int qmax = a.Skv <= 32 ? 128 : ATB_CHUNK;
  if (sq_pad > qmax) sq_pad = qmax;
  const int smem_q = skv_pad * (2 * P80 + P64), smem_k = sq_pad * (2 * P80 + 2 * P64 + 8);
  const int nqx = (a.Sq + 127) / 128, nkx = (a.Skv + 127) / 128;
  hipLaunchKernelGGL(attn_bwd_mfma_kernel, dim3(nqx + nkx, a.heads, a.B), dim3(256),
                     smem_q > smem_k ? smem_q : smem_k, st, a, nqx, skv_pad, sq_pad);
  return sdmi_check_launch("attention_bwd (mfma)");
}

}  // namespace

extern "C" int sdmi_attention_bwd(const SdmiAttnBwdArgs* a, void* stream) {
  SDMI_REQUIRE(a && a->q && a->k && a->v && a->out && a->dout && a->lse && a->dq && a->dk && a->dv,
               "null pointer");
  const int vec = a->dtype == SDMI_BF16 ? 8 : 4;
  SDMI_REQUIRE(a->ldq % vec == 0 && a->ldk % vec == 0 && a->ldv % vec == 0 && a->ldo % vec == 0,
               "row pitches must keep 16-byte alignment");
  const int hd = a->head_dim > 0 ? a->head_dim : 32;
  SDMI_REQUIRE(hd == 32 || hd == 48, "head_dim must be 32 or 48");
  hipStream_t st = (hipStream_t)stream;
  // chunked staging: any sequence length on either side (slot cross-attention under 28 x 28 queries
  // included: one wave walking the query chunks with MFMAs beats the thread-per-(key, channel) kernel
  // below by an order of magnitude there)
  if (a->dtype == SDMI_BF16 && hd == 32)
    return launch_attn_bwd_mfma(*a, st);
  SDMI_REQUIRE(a->Skv >= 1 && a->Skv <= 400, "Skv must be in [1, 400] (fp32 / head_dim 48: K/V staged whole in LDS)");
  if (a->dtype == SDMI_BF16)
    return hd == 32 ? launch_attn_bwd<bf16_t, 32>(*a, st) : launch_attn_bwd<bf16_t, 48>(*a, st);
  return hd == 32 ? launch_attn_bwd<float, 32>(*a, st) : launch_attn_bwd<float, 48>(*a, st);
}
