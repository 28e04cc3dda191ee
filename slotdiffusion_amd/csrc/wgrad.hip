// Weight gradient of conv / linear as an MFMA GEMM contracting over the B*Ho*Wo rows
// (include/sdmi.h: sdmi_wgrad):
//     dW[n][k] = sum_m dY[m][n] * A[m][k],    k = (kh, kw, ci),  A gathered like the forward.
//
// Both operands are m-major in HBM while the contraction runs over m, so the staging pass
// transposes: each thread loads a VEC x VEC block (VEC rows m, one 16-byte vector of n or k each),
// transposes it in registers, and writes VEC 16-byte vectors into LDS tiles laid out [n][m] /
// [k][m] -- m-contiguous rows -- after which the MFMA part is the forward kernel's: lane l reads
// row (l&31), bytes [ks*32 + (l>>5)*16, +16) of both tiles (conflict-free 16-byte padded pitch).
//
// Parallelism: the output has only (N/128)*(K/128) tiles, so M is split across `splits`
// workgroups per tile; partials go to a workspace [splits][N][K] and a deterministic second kernel
// reduces them (optionally accumulating into dW).  No atomics.
#include "common.h"

namespace {

template <typename T> struct WCfg;
template <> struct WCfg<bf16_t> { static constexpr int MTB = 256; };  // 128 m per iteration
template <> struct WCfg<float> { static constexpr int MTB = 128; };   // 32 m per iteration

// in-register VEC x VEC transpose of 16-byte vectors
__device__ __forceinline__ void transpose_block(const u32x4 (&r)[8], u32x4 (&c)[8], bf16_t) {
#pragma unroll
  for (int n = 0; n < 8; ++n) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const unsigned a = r[2 * e][n >> 1], b = r[2 * e + 1][n >> 1];
      c[n][e] = (n & 1) ? ((a >> 16) | (b & 0xffff0000u)) : ((a & 0xffffu) | (b << 16));
    }
  }
}
__device__ __forceinline__ void transpose_block(const u32x4 (&r)[4], u32x4 (&c)[4], float) {
#pragma unroll
  for (int n = 0; n < 4; ++n)
#pragma unroll
    for (int e = 0; e < 4; ++e) c[n][e] = r[e][n];
}

template <typename T, int TN, int TK>
__global__ __launch_bounds__(256) void wgrad_kernel(SdmiWgradArgs p, int tiles_n, int tiles_k,
                                                    int m_per_split) {
  constexpr int VEC = 16 / sizeof(T);
  constexpr int MTB = WCfg<T>::MTB;
  constexpr int MT = MTB / sizeof(T);        // m rows per iteration
  constexpr int ROWB = MTB + 16;
  constexpr int KSTEPS = MTB / 32;
  constexpr int MBLK = MT / VEC;             // blocks along m
  constexpr int Y_BLOCKS = (TN / VEC) * MBLK;
  constexpr int A_BLOCKS = (TK / VEC) * MBLK;
  constexpr int Y_PER = (Y_BLOCKS + 255) / 256;
  constexpr int A_PER = (A_BLOCKS + 255) / 256;
  constexpr int WTN = TN / 2, WTK = TK / 2;
  constexpr int FN = WTN / 32, FK = WTK / 32;

  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* Ys = smem;                  // [TN][ROWB]
  char* As = smem + TN * ROWB;      // [TK][ROWB]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wn = wave >> 1, wk = wave & 1;
  const int tile = blockIdx.x;
  const int tile_n = tile / tiles_k, tile_k = tile - tile_n * tiles_k;
  const int split = blockIdx.y;
  const int n0 = tile_n * TN, k0 = tile_k * TK;
  const int m_begin = split * m_per_split;
  int m_end = m_begin + m_per_split;
  if (m_end > p.M) m_end = p.M;

  const T* __restrict__ Ag = (const T*)p.a;
  const T* __restrict__ Yg = (const T*)p.dy;
  const int HoWo = p.Ho * p.Wo;

  // ---- per-thread block coordinates (fixed for the whole kernel except the m position)
  int y_mb[Y_PER], y_n[Y_PER];
  bool y_act[Y_PER];
#pragma unroll
  for (int i = 0; i < Y_PER; ++i) {
    const int blk = tid + i * 256;
    y_act[i] = blk < Y_BLOCKS;
    y_mb[i] = blk % MBLK;
    y_n[i] = n0 + (blk / MBLK) * VEC;
  }
  int a_mb[A_PER], a_kb[A_PER], a_ci[A_PER], a_kh[A_PER], a_kw[A_PER];
  bool a_act[A_PER];
  // position state of the first row of each A block: (b, oy, ox)
  int a_b[A_PER], a_oy[A_PER], a_ox[A_PER];
#pragma unroll
  for (int i = 0; i < A_PER; ++i) {
    const int blk = tid + i * 256;
    a_mb[i] = blk % MBLK;
    a_kb[i] = blk / MBLK;
    const int k = k0 + a_kb[i] * VEC;
    a_act[i] = blk < A_BLOCKS && k < p.K;
    const int kk = a_act[i] ? k : 0;
    const int tap = kk / p.Cin;
    a_ci[i] = kk - tap * p.Cin;
    a_kh[i] = tap / p.KW;
    a_kw[i] = tap - a_kh[i] * p.KW;
    const int m = m_begin + a_mb[i] * VEC;
    a_b[i] = m / HoWo;
    const int rem = m - a_b[i] * HoWo;
    a_oy[i] = rem / p.Wo;
    a_ox[i] = rem - a_oy[i] * p.Wo;
  }

  u32x4 ry[Y_PER][VEC], ra[A_PER][VEC];
  const u32x4 zero4 = {0u, 0u, 0u, 0u};
  unsigned ymask[Y_PER], amask[A_PER];

  auto load_iter = [&](int mt) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < Y_PER; ++i) {
      ymask[i] = 0;
      const int mrow = mt + y_mb[i] * VEC;
#pragma unroll
      for (int j = 0; j < VEC; ++j) {
        const int m = mrow + j;
        const bool ok = y_act[i] && m < m_end && y_n[i] < p.N;
        ymask[i] |= (ok ? 1u : 0u) << j;
        const long long off = ok ? (long long)m * p.ldy + y_n[i] : 0;
        ry[i][j] = *reinterpret_cast<const u32x4*>(Yg + off);
      }
    }
#pragma unroll
    for (int i = 0; i < A_PER; ++i) {
      amask[i] = 0;
      int b = a_b[i], oy = a_oy[i], ox = a_ox[i];
      const int mrow = mt + a_mb[i] * VEC;
#pragma unroll
      for (int j = 0; j < VEC; ++j) {
        bool ok = a_act[i] && (mrow + j) < m_end;
        int iy = oy * p.stride - p.pad_t + a_kh[i], ix = ox * p.stride - p.pad_l + a_kw[i];
        if (p.ups) {
          ok = ok && iy >= 0 && iy < 2 * p.H && ix >= 0 && ix < 2 * p.W;
          iy >>= 1;
          ix >>= 1;
        } else {
          ok = ok && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
        }
        amask[i] |= (ok ? 1u : 0u) << j;
        const long long off =
            ok ? (((long long)b * p.H + iy) * p.W + ix) * p.lda + a_ci[i] : 0;
        ra[i][j] = *reinterpret_cast<const u32x4*>(Ag + off);
        // next pixel (row m+1)
        if (++ox == p.Wo) { ox = 0; if (++oy == p.Ho) { oy = 0; ++b; } }
      }
      // advance this block's first row by MT for the next iteration
      int adv = MT;
      ox = a_ox[i] + adv;
      oy = a_oy[i];
      b = a_b[i];
      while (ox >= p.Wo) { ox -= p.Wo; if (++oy == p.Ho) { oy = 0; ++b; } }
      a_ox[i] = ox; a_oy[i] = oy; a_b[i] = b;
    }
  };
  auto store_iter = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < Y_PER; ++i) {
      if (!y_act[i]) continue;
      u32x4 in[VEC], out[VEC];
#pragma unroll
      for (int j = 0; j < VEC; ++j) in[j] = ((ymask[i] >> j) & 1u) ? ry[i][j] : zero4;
      transpose_block(in, out, T());
      const int nloc = y_n[i] - n0;
#pragma unroll
      for (int c = 0; c < VEC; ++c)
        *reinterpret_cast<u32x4*>(Ys + (nloc + c) * ROWB + y_mb[i] * 16) = out[c];
    }
#pragma unroll
    for (int i = 0; i < A_PER; ++i) {
      if (tid + i * 256 >= A_BLOCKS) continue;
      u32x4 in[VEC], out[VEC];
#pragma unroll
      for (int j = 0; j < VEC; ++j) in[j] = ((amask[i] >> j) & 1u) ? ra[i][j] : zero4;
      transpose_block(in, out, T());
#pragma unroll
      for (int c = 0; c < VEC; ++c)
        *reinterpret_cast<u32x4*>(As + (a_kb[i] * VEC + c) * ROWB + a_mb[i] * 16) = out[c];
    }
  };

  f32x16 acc[FN][FK];
#pragma unroll
  for (int i = 0; i < FN; ++i)
#pragma unroll
    for (int j = 0; j < FK; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int frag_row = lane & 31, frag_kb = (lane >> 5) * 16;
  if (m_begin < m_end) {
    load_iter(m_begin);
    for (int mt = m_begin; mt < m_end; mt += MT) {
      store_iter();
      __syncthreads();
      if (mt + MT < m_end) load_iter(mt + MT);
      const char* Yf = Ys + (wn * WTN + frag_row) * ROWB + frag_kb;
      const char* Af = As + (wk * WTK + frag_row) * ROWB + frag_kb;
#pragma unroll
      for (int ks = 0; ks < KSTEPS; ++ks) {
        u32x4 fy[FN], fa[FK];
#pragma unroll
        for (int i = 0; i < FN; ++i)
          fy[i] = *reinterpret_cast<const u32x4*>(Yf + i * 32 * ROWB + ks * 32);
#pragma unroll
        for (int j = 0; j < FK; ++j)
          fa[j] = *reinterpret_cast<const u32x4*>(Af + j * 32 * ROWB + ks * 32);
#pragma unroll
        for (int i = 0; i < FN; ++i)
#pragma unroll
          for (int j = 0; j < FK; ++j) {
            if constexpr (sizeof(T) == 2) {
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
                  __builtin_bit_cast(bf16x8, fy[i]), __builtin_bit_cast(bf16x8, fa[j]), acc[i][j],
                  0, 0, 0);
            } else {
#pragma unroll
              for (int c = 0; c < 4; ++c)
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(
                    __uint_as_float(fy[i][c]), __uint_as_float(fa[j][c]), acc[i][j], 0, 0, 0);
            }
          }
      }
      __syncthreads();
    }
  }
  // ---- write the partial tile: ws[split][n][k]
  float* ws = p.workspace + (long long)split * p.N * p.K;
  const int col_l = lane & 31, row_l = (lane >> 5) * 4;
#pragma unroll
  for (int i = 0; i < FN; ++i)
#pragma unroll
    for (int j = 0; j < FK; ++j) {
      const int k = k0 + wk * WTK + j * 32 + col_l;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int n = n0 + wn * WTN + i * 32 + (r & 3) + 8 * (r >> 2) + row_l;
        if (n < p.N && k < p.K) ws[(long long)n * p.K + k] = acc[i][j][r];
      }
    }
}

__global__ __launch_bounds__(256) void wgrad_reduce_kernel(SdmiWgradArgs p) {
  const long long total = (long long)p.N * p.K;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total;
       i += (long long)gridDim.x * 256) {
    float s = p.accumulate ? p.dw[i] : 0.f;
    for (int k = 0; k < p.splits; ++k) s += p.workspace[(long long)k * total + i];
    p.dw[i] = s;
  }
}

// column sums of dY (bias gradient).  Stage 1: workgroup = one row chunk; thread t owns the 16-byte
// column vector cv = t % CVp and walks rows t / CVp, + 256/CVp, ... (whole 128-byte lines per wave,
// several independent loads in flight); LDS reduce over the row threads -> partial[chunk][N].
// Stage 2: 64 columns x 4 chunk-groups per workgroup.
__device__ __forceinline__ int next_pow2_w(int v) {
  int p = 1;
  while (p < v) p <<= 1;
  return p;
}
template <typename T>
__global__ __launch_bounds__(256) void colsum_partial_kernel(const T* y, float* part, int M, int N,
                                                             int ldy, int rows_per) {
  constexpr int VEC = Elem<T>::VEC;
  __shared__ float red[256][VEC];
  const int CVtot = (N + VEC - 1) / VEC;
  int CVp = next_pow2_w(CVtot);
  if (CVp > 256) CVp = 256;                    // wide N: column blocks along grid.y
  const int R = 256 / CVp;
  const int cv = blockIdx.y * CVp + threadIdx.x % CVp, r0 = threadIdx.x / CVp;
  const int CV = CVtot;
  const int m0 = blockIdx.x * rows_per;
  int m1 = m0 + rows_per;
  if (m1 > M) m1 = M;
  float acc[VEC];
#pragma unroll
  for (int j = 0; j < VEC; ++j) acc[j] = 0.f;
  if (cv < CV) {
    const T* base = y + cv * VEC;
    int m = m0 + r0;
    for (; m + 3 * R < m1; m += 4 * R) {
      float f0[VEC], f1[VEC], f2[VEC], f3[VEC];
      unpack16<T>(*reinterpret_cast<const uint4*>(base + (long long)m * ldy), f0);
      unpack16<T>(*reinterpret_cast<const uint4*>(base + (long long)(m + R) * ldy), f1);
      unpack16<T>(*reinterpret_cast<const uint4*>(base + (long long)(m + 2 * R) * ldy), f2);
      unpack16<T>(*reinterpret_cast<const uint4*>(base + (long long)(m + 3 * R) * ldy), f3);
#pragma unroll
      for (int j = 0; j < VEC; ++j) acc[j] += (f0[j] + f1[j]) + (f2[j] + f3[j]);
    }
    for (; m < m1; m += R) {
      float f0[VEC];
      unpack16<T>(*reinterpret_cast<const uint4*>(base + (long long)m * ldy), f0);
#pragma unroll
      for (int j = 0; j < VEC; ++j) acc[j] += f0[j];
    }
  }
#pragma unroll
  for (int j = 0; j < VEC; ++j) red[threadIdx.x][j] = acc[j];
  __syncthreads();
  if (r0 == 0 && cv < CV) {
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      const int n = cv * VEC + j;
      if (n < N) {
        float s = 0.f;
        for (int r = 0; r < R; ++r) s += red[r * CVp + (threadIdx.x % CVp)][j];
        part[(long long)blockIdx.x * N + n] = s;
      }
    }
  }
}
__global__ __launch_bounds__(256) void colsum_final_kernel(const float* part, float* out, int N,
                                                           int chunks, int accumulate) {
  __shared__ float red[4][64];
  const int n = blockIdx.x * 64 + (threadIdx.x & 63), kg = threadIdx.x >> 6;
  float s = 0.f;
  if (n < N)
    for (int c = kg; c < chunks; c += 4) s += part[(long long)c * N + n];
  red[kg][threadIdx.x & 63] = s;
  __syncthreads();
  if (kg == 0 && n < N) {
    const float t = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
    out[n] = (accumulate ? out[n] : 0.f) + t;
  }
}

template <typename T, int TN, int TK>
int launch_wgrad(const SdmiWgradArgs& a, hipStream_t st) {
  constexpr int MTB = WCfg<T>::MTB;
  constexpr int smem = (TN + TK) * (MTB + 16);
  auto kern = wgrad_kernel<T, TN, TK>;
  static bool done = false;
  if (!done) {
    if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, smem) !=
        hipSuccess) {
      sdmi_set_error("wgrad: hipFuncSetAttribute failed");
      return SDMI_ELAUNCH;
    }
    done = true;
  }
  const int tiles_n = (a.N + TN - 1) / TN, tiles_k = (a.K + TK - 1) / TK;
  const int MT = MTB / (int)sizeof(T);
  int mps = (a.M + a.splits - 1) / a.splits;
  mps = (mps + MT - 1) / MT * MT;
  dim3 grid(tiles_n * tiles_k, a.splits);
  hipLaunchKernelGGL(kern, grid, dim3(256), smem, st, a, tiles_n, tiles_k, mps);
  return sdmi_check_launch("wgrad");
}

}  // namespace

extern "C" int sdmi_wgrad(const SdmiWgradArgs* a, void* stream) {
  SDMI_REQUIRE(a && a->a && a->dy && a->dw && a->workspace, "null pointer");
  SDMI_REQUIRE(a->dtype == SDMI_F32 || a->dtype == SDMI_BF16, "bad dtype");
  const int vec = a->dtype == SDMI_BF16 ? 8 : 4;
  SDMI_REQUIRE(a->K == a->KH * a->KW * a->Cin, "K != KH*KW*Cin");
  SDMI_REQUIRE(a->Cin % vec == 0 && a->lda % vec == 0 && a->ldy % vec == 0,
               "Cin/lda/ldy must be multiples of the 16-byte vector width");
  SDMI_REQUIRE(a->M == a->B * a->Ho * a->Wo, "M != B*Ho*Wo");
  SDMI_REQUIRE(a->splits >= 1, "splits");
  hipStream_t st = (hipStream_t)stream;
  int rc;
  const bool small = a->N <= 64 || a->K <= 64;
  if (a->dtype == SDMI_BF16)
    rc = small ? launch_wgrad<bf16_t, 64, 64>(*a, st) : launch_wgrad<bf16_t, 128, 128>(*a, st);
  else
    rc = small ? launch_wgrad<float, 64, 64>(*a, st) : launch_wgrad<float, 128, 128>(*a, st);
  if (rc) return rc;
  {
    const long long total = (long long)a->N * a->K;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(blocks), dim3(256), 0, st, *a);
    rc = sdmi_check_launch("wgrad reduce");
    if (rc) return rc;
  }
  if (a->dbias) {
    // reuse the (now consumed) workspace for the column-sum partials
    int chunks = (a->M + 255) / 256;
    if (chunks > 256) chunks = 256;
    const int rows_per = (a->M + chunks - 1) / chunks;
    if (a->dtype == SDMI_BF16)
      hipLaunchKernelGGL(colsum_partial_kernel<bf16_t>, dim3(chunks, ((a->N + 7) / 8 + 255) / 256), dim3(256), 0, st,
                         (const bf16_t*)a->dy, a->workspace, a->M, a->N, a->ldy, rows_per);
    else
      hipLaunchKernelGGL(colsum_partial_kernel<float>, dim3(chunks, ((a->N + 3) / 4 + 255) / 256), dim3(256), 0, st,
                         (const float*)a->dy, a->workspace, a->M, a->N, a->ldy, rows_per);
    hipLaunchKernelGGL(colsum_final_kernel, dim3((a->N + 63) / 64), dim3(256), 0, st,
                       a->workspace, a->dbias, a->N, chunks, a->accumulate);
    rc = sdmi_check_launch("wgrad dbias");
  }
  return rc;
}
