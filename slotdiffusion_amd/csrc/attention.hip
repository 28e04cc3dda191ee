// Multi-head attention (UNet self-attention over 16..256 tokens, slot cross-attention over 7..15
// keys; head_dim 32, or 48 in the SAVi predictor).
//
// bf16, head_dim 32: matrix-core kernel `attn_fwd_mfma_kernel` (below).
// fp32 (parity mode) and head_dim 48: `attn_fwd_kernel` -- K and V of one (image, head) staged
// once in LDS as fp32; every lane owns one query (q and the output row live in registers) and
// walks the keys with an online softmax, reading K/V rows as LDS broadcasts.
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


// ------------------------------------------------------------------------------------------
// Matrix-core forward (bf16, head_dim 32).  One wave owns 32 queries; everything is computed
// TRANSPOSED so that a lane's accumulator registers all belong to one query:
//     S^T[k][q] = K_blk Q^T      : v_mfma 32x32x16 x2, A = K rows (16-byte LDS reads), B = Q rows
//     softmax statistics of query q = lane & 31: register reductions + one exchange with lane ^ 32
//     O^T[d][q] += V^T_blk P^T   : B operand = the P^T accumulator registers converted to bf16 IN
//                                  PLACE (MFMA k-slot s of lane half h  <->  key 8*(s>>2&1 ...)
//                                  see `key of slot` below), A = V^T through ds_read_b64_tr_b16
// so no value ever moves between lanes except the two half-wave exchanges.
// C/D layout of the 32x32 MFMA: column = lane & 31, row(r) = (r&3) + 8*(r>>2) + 4*(lane>>5).
// key of slot: MFMA m (keys 16m..16m+15 of the block) uses C registers r = 8m .. 8m+7 of both
// lane halves; slot (h*8 + jj*4 + i)  <->  key 16m + 8jj + 4h + i  -- the same map is used to
// gather the V^T operand (two transposing reads of 4 keys each).
// ------------------------------------------------------------------------------------------
typedef short a_s16x4 __attribute__((ext_vector_type(4)));
typedef short a_s16x8 __attribute__((ext_vector_type(8)));
#define ATT_LDS_V4(p) ((__attribute__((address_space(3))) a_s16x4*)(p))
constexpr int ATT_VP = 64;   // V row pitch: = 64 (mod 256), what the transposing read wants

__device__ __forceinline__ unsigned pack_bf16x2(float lo, float hi) {
  return (unsigned)f32_to_bf16(lo) | ((unsigned)f32_to_bf16(hi) << 16);
}

// HD = 32 (UNet) or 64 (DINO ViT).  Keys pass through LDS in chunks of CHUNK rows (one chunk covers the
// UNet's sequences at 128^2; the ViT's 785 keys x 64 channels take four), the online-softmax state and the
// output accumulators stay in registers across chunks.  V is kept as HD/32 separate 32-channel images
// so the transposing read sees the same 64-byte pitch for either head size.
template <int HD, int CHUNK>
__global__ __launch_bounds__(256) void attn_fwd_mfma_kernel(SdmiAttnArgs p) {
  constexpr int ND = HD / 32;            // 32-channel groups of the head
  constexpr int KP = HD * 2 + 16;        // K row pitch in LDS (conflict-free ds_read_b128)
  constexpr int PIECES = HD / 8;         // 16-byte pieces per row
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int skv_pad = (p.Skv + 31) / 32 * 32;
  const int cap = skv_pad < CHUNK ? skv_pad : CHUNK;
  char* Ks = smem;
  char* Vs = smem + cap * KP;            // image j (channels 32j..32j+31) at Vs + j * cap * ATT_VP
  const int b = blockIdx.z, h = blockIdx.y;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int q0 = (blockIdx.x * 4 + wave) * 32;
  const bool active = q0 < p.Sq;         // idle waves still stage and keep the barriers
  const int ql = lane & 31, hh = lane >> 5;
  const int qi = q0 + ql;
  const int qc = qi < p.Sq ? qi : p.Sq - 1;
  // B operand of S^T: this lane's query row, d = ks*16 + hh*8 .. +8
  bf16x8 bq[2 * ND];
  {
    const bf16_t* qp = (const bf16_t*)p.q + ((long long)b * p.Sq + qc) * p.ldq + h * HD + hh * 8;
#pragma unroll
    for (int ks = 0; ks < 2 * ND; ++ks)
      bq[ks] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(qp + 16 * ks));
  }
  f32x16 o[ND];
#pragma unroll
  for (int j = 0; j < ND; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[j][r] = 0.f;
  float m = -INFINITY, lsum = 0.f;            // m: running maximum in the exp2 domain
  const float sc2 = p.scale * 1.4426950408889634f;
  const int g = lane >> 4, t = lane & 15;
  const char* kfrag = Ks + ql * KP + hh * 16;
  // transposing read: lane addresses piece (row t>>2, 4 columns at (t&3)*4) of its group's block
  const char* vfrag = Vs + (4 * hh + (t >> 2)) * ATT_VP + ((g & 1) * 16 + (t & 3) * 4) * 2;
  const bf16_t* kbase = (const bf16_t*)p.k + (long long)b * p.Skv * p.ldk + h * HD;
  const bf16_t* vbase = (const bf16_t*)p.v + (long long)b * p.Skv * p.ldv + h * HD;
  for (int c0 = 0; c0 < skv_pad; c0 += CHUNK) {
    const int rows = skv_pad - c0 < CHUNK ? skv_pad - c0 : CHUNK;
    if (c0) __syncthreads();
    {  // stage K, V rows c0 .. c0+rows of this (image, head); pad rows zeroed
      const u32x4 zero4 = {0u, 0u, 0u, 0u};
      for (int i = tid; i < rows * PIECES; i += 256) {
        const int row = i / PIECES, c = i % PIECES;
        const bool ok = c0 + row < p.Skv;
        const u32x4 kv = ok ? *reinterpret_cast<const u32x4*>(kbase + (long long)(c0 + row) * p.ldk + c * 8) : zero4;
        const u32x4 vv = ok ? *reinterpret_cast<const u32x4*>(vbase + (long long)(c0 + row) * p.ldv + c * 8) : zero4;
        *reinterpret_cast<u32x4*>(Ks + row * KP + c * 16) = kv;
        *reinterpret_cast<u32x4*>(Vs + (c >> 2) * cap * ATT_VP + row * ATT_VP + (c & 3) * 16) = vv;
      }
    }
    __syncthreads();
    if (!active) continue;
    for (int kb = 0; kb < rows / 32; ++kb) {
      f32x16 s;
#pragma unroll
      for (int r = 0; r < 16; ++r) s[r] = 0.f;
#pragma unroll
      for (int ks = 0; ks < 2 * ND; ++ks) {
        const u32x4 a = *reinterpret_cast<const u32x4*>(kfrag + kb * 32 * KP + ks * 32);
        s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), bq[ks], s, 0, 0, 0);
      }
      // The softmax is this kernel's VALU bill (head dim 32: 2 x 8 MFMA passes per 16 scores of a lane next to
      // ~12 VALU issue slots per score, and MFMA and VALU of a SIMD do not overlap), so it is kept minimal:
      // scores live in the exp2 domain (scale * log2(e) folded into one multiply, v_exp_f32 directly), the key
      // mask is applied only in a block that crosses Skv, and the running output is rescaled only when some
      // lane's maximum moved (alpha == 1 exactly otherwise).
      float bmax = -INFINITY;
      if (c0 + kb * 32 + 32 > p.Skv) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int key = c0 + kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
          s[r] = key < p.Skv ? s[r] * sc2 : -INFINITY;
          bmax = fmaxf(bmax, s[r]);
        }
      } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          s[r] *= sc2;
          bmax = fmaxf(bmax, s[r]);
        }
      }
      bmax = fmaxf(bmax, __shfl_xor(bmax, 32, 64));
      const float m_new = fmaxf(m, bmax);
      float psum = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        s[r] = __builtin_amdgcn_exp2f(s[r] - m_new);
        psum += s[r];
      }
      psum += __shfl_xor(psum, 32, 64);
      if (__builtin_amdgcn_ballot_w64(m_new > m) != 0) {
        const float alpha = __builtin_amdgcn_exp2f(m - m_new);
        lsum = lsum * alpha + psum;
#pragma unroll
        for (int j = 0; j < ND; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) o[j][r] *= alpha;
      } else {
        lsum += psum;
      }
      m = m_new;
#pragma unroll
      for (int mm = 0; mm < 2; ++mm) {
        const u32x4 pb = {pack_bf16x2(s[8 * mm + 0], s[8 * mm + 1]), pack_bf16x2(s[8 * mm + 2], s[8 * mm + 3]),
                          pack_bf16x2(s[8 * mm + 4], s[8 * mm + 5]), pack_bf16x2(s[8 * mm + 6], s[8 * mm + 7])};
#pragma unroll
        for (int j = 0; j < ND; ++j) {
          const char* vp = vfrag + j * cap * ATT_VP + (kb * 32 + 16 * mm) * ATT_VP;
          const a_s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(ATT_LDS_V4(vp));
          const a_s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(ATT_LDS_V4(vp + 8 * ATT_VP));
          const a_s16x8 av = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
          o[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, av),
                                                         __builtin_bit_cast(bf16x8, pb), o[j], 0, 0, 0);
        }
      }
    }
  }
  if (active && qi < p.Sq) {
    const float inv = 1.f / lsum;
    bf16_t* op = (bf16_t*)p.out + ((long long)b * p.Sq + qi) * p.ldo + h * HD + 4 * hh;
#pragma unroll
    for (int jd = 0; jd < ND; ++jd)
#pragma unroll
      for (int j = 0; j < 4; ++j) {    // d = 32jd + 8j + 4hh + (0..3)
        uint2 w;
        w.x = pack_bf16x2(o[jd][4 * j] * inv, o[jd][4 * j + 1] * inv);
        w.y = pack_bf16x2(o[jd][4 * j + 2] * inv, o[jd][4 * j + 3] * inv);
        *reinterpret_cast<uint2*>(op + 32 * jd + 8 * j) = w;
      }
    if (p.lse && hh == 0) p.lse[((long long)b * p.heads + h) * p.Sq + qi] = m * 0.6931471805599453f + __logf(lsum);
  }
}

}  // namespace

extern "C" int sdmi_attention(const SdmiAttnArgs* a, void* stream) {
  SDMI_REQUIRE(a && a->q && a->k && a->v && a->out, "null pointer");
  const int vec = a->dtype == SDMI_BF16 ? 8 : 4;
  SDMI_REQUIRE(a->ldq % vec == 0 && a->ldk % vec == 0 && a->ldv % vec == 0 && a->ldo % vec == 0,
               "row pitches must keep 16-byte alignment");
  const int hd = a->head_dim > 0 ? a->head_dim : 32;
  SDMI_REQUIRE(hd == 32 || hd == 48 || hd == 64,
               "head_dim must be 32 (UNet), 48 (SAVi predictor) or 64 (DINO ViT, bf16 only)");
  hipStream_t st = (hipStream_t)stream;
  if (a->dtype == SDMI_BF16 && (hd == 32 || hd == 64)) {
    // matrix-core kernel, any key count: K and V of one (image, head) pass through LDS in chunks of 512
    // keys x 144 B (head_dim 32; one chunk covers every 128^2 configuration, the 28 x 28 self-attention
    // of the 224^2 configs takes two) or 256 keys x 272 B (head_dim 64): two workgroups per CU either way.
    // (per device, return code checked: sdmi_optin_lds)
    SDMI_OPTIN_LDS((attn_fwd_mfma_kernel<32, 512>), 512 * (32 * 2 + 16 + ATT_VP), "attention (mfma, head dim 32)");
    SDMI_OPTIN_LDS((attn_fwd_mfma_kernel<64, 256>), 256 * (64 * 2 + 16 + 2 * ATT_VP), "attention (mfma, head dim 64)");
    SDMI_REQUIRE(a->Skv >= 1, "Skv must be positive");
    const int skv_pad = (a->Skv + 31) / 32 * 32;
    dim3 g2((a->Sq + 127) / 128, a->heads, a->B);
    if (hd == 32) {
      const int cap = skv_pad < 512 ? skv_pad : 512;
      hipLaunchKernelGGL((attn_fwd_mfma_kernel<32, 512>), g2, dim3(256), cap * (32 * 2 + 16 + ATT_VP), st, *a);
    } else {
      const int cap = skv_pad < 256 ? skv_pad : 256;
      hipLaunchKernelGGL((attn_fwd_mfma_kernel<64, 256>), g2, dim3(256), cap * (64 * 2 + 16 + 2 * ATT_VP),
                         st, *a);
    }
    return sdmi_check_launch("attention (mfma)");
  }
  SDMI_REQUIRE(hd != 64, "head_dim 64 is implemented for bf16 only");
  SDMI_REQUIRE(a->Skv >= 1 && a->Skv <= 400, "Skv must be in [1, 400] (fp32 / head_dim 48: K/V staged whole in LDS)");
  int threads = ((a->Sq + 63) / 64) * 64;
  if (threads > 256) threads = 256;
  dim3 grid((a->Sq + threads - 1) / threads, a->heads, a->B);
  const int smem = 2 * a->Skv * hd * 4;
#define ATTN_GO(T, HDV)                                                                      \
  do {                                                                                       \
    SDMI_OPTIN_LDS((attn_fwd_kernel<T, HDV>),                                                \
                   (2 * 512 * HDV * 4 < 160 * 1024 ? 2 * 512 * HDV * 4 : 160 * 1024), "attention"); \
    hipLaunchKernelGGL((attn_fwd_kernel<T, HDV>), grid, dim3(threads), smem, st, *a);        \
  } while (0)
  if (a->dtype == SDMI_BF16) { if (hd == 32) ATTN_GO(bf16_t, 32); else ATTN_GO(bf16_t, 48); }
  else { if (hd == 32) ATTN_GO(float, 32); else ATTN_GO(float, 48); }
#undef ATTN_GO
  return sdmi_check_launch("attention");
}
