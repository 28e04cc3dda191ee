// Weight-gradient bodies shared by wgrad.hip (stand-alone launches) and bwd_pair.hip (data gradient and
// weight gradient of one layer in ONE launch).  See wgrad.hip for the method.
#pragma once
#include "common.h"

namespace {

// Bias gradient for one (column tile, split): thread t owns the 16-byte column vector t % CV and
// walks rows t / CV, + R, ... of the split (4 independent loads in flight), LDS reduce over the R
// row threads -> dbias (splits == 1) or the bias partials behind the dW partials.
template <typename T, int TN>
__device__ __forceinline__ void bias_tile(const SdmiWgradArgs& p, int tile_n, int split,
                                          int m_per_split, char* smem) {
  constexpr int VEC = 16 / sizeof(T);
  constexpr int CV = TN / VEC, R = 256 / CV;
  float* red = reinterpret_cast<float*>(smem);      // [256][VEC]
  const int tid = threadIdx.x;
  if (tid >= 256) return;                           // no barrier below involves the upper waves
  const int cv = tid % CV, r0 = tid / CV;
  const int n = tile_n * TN + cv * VEC;
  const int m0 = split * m_per_split;
  int m1 = m0 + m_per_split;
  if (m1 > p.M) m1 = p.M;
  float acc[VEC];
#pragma unroll
  for (int j = 0; j < VEC; ++j) acc[j] = 0.f;
  if (n < p.N) {
    const T* base = (const T*)p.dy + n;
    int m = m0 + r0;
    for (; m + 3 * R < m1; m += 4 * R) {
      float f0[VEC], f1[VEC], f2[VEC], f3[VEC];
      unpack16<T>(*reinterpret_cast<const uint4*>(base + (long long)m * p.ldy), f0);
      unpack16<T>(*reinterpret_cast<const uint4*>(base + (long long)(m + R) * p.ldy), f1);
      unpack16<T>(*reinterpret_cast<const uint4*>(base + (long long)(m + 2 * R) * p.ldy), f2);
      unpack16<T>(*reinterpret_cast<const uint4*>(base + (long long)(m + 3 * R) * p.ldy), f3);
#pragma unroll
      for (int j = 0; j < VEC; ++j) acc[j] += (f0[j] + f1[j]) + (f2[j] + f3[j]);
    }
    for (; m < m1; m += R) {
      float f0[VEC];
      unpack16<T>(*reinterpret_cast<const uint4*>(base + (long long)m * p.ldy), f0);
#pragma unroll
      for (int j = 0; j < VEC; ++j) acc[j] += f0[j];
    }
  }
#pragma unroll
  for (int j = 0; j < VEC; ++j) red[tid * VEC + j] = acc[j];
  __syncthreads();
  if (tid < TN && tile_n * TN + tid < p.N) {
    const int c = tid / VEC, j = tid % VEC;
    float s = 0.f;
    for (int r = 0; r < R; ++r) s += red[(r * CV + c) * VEC + j];
    const int nn = tile_n * TN + tid;
    if (p.splits == 1)
      p.dbias[nn] = (p.accumulate ? p.dbias[nn] : 0.f) + s;
    else
      p.workspace[(long long)p.splits * p.N * p.K + (long long)split * p.N + nn] = s;
  }
}


// ------------------------------------------------------------------------------------------
// bf16 kernel: no register transposes at all.  The loader waves copy the operands as they lie in
// HBM -- row-major [m][n] / [m][k], every lane a 16-byte piece of a row, 16 lanes per 256-byte row
// (coalesced) -- into LDS with a row pitch = 64 (mod 256) bytes, and the MFMA waves fetch their
// fragments with the gfx950 transposing LDS read: a 16-lane group of ds_read_b64_tr_b16 reads a
// [4 m][16 n] block and hands lane t column t, i.e. 4 consecutive contraction elements of output
// row t -- exactly half of a 32x32x16 MFMA operand (semantics pinned by tools/probes/tr16.hip).
// With that pitch the 32 lanes serviced per LDS cycle touch 64 distinct banks.
// ------------------------------------------------------------------------------------------
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
#define SDMI_LDS_V4(p) ((__attribute__((address_space(3))) s16x4*)(p))

// MODE 0: general convolution (stride, nearest-x2 fold, any image size): per-step address
//         arithmetic in the loaders.
// MODE 1 (1x1 / linear) and MODE 2 (stride-1 "same" convolution on a power-of-two image, where the
// input pixel of output row m is m + const): scalar-only loaders -- buffer loads whose per-lane
// byte offset is fixed for the whole launch, the walk over m in an SGPR offset, out-of-range
// offsets (zeros) for inactive columns, the split's tail and image borders.  A loader wave's VALU
// work serialises with the MFMAs of the wave next to it (tools/probes/ldsdma.hip), so MODE 2
// keeps only the border test (~10 VALU per vector and step) and MODE 1 none.
template <int TN, int TK, int MODE>
__device__ __forceinline__ void wgrad_tr_body(const SdmiWgradArgs& p, int tiles_n, int tiles_k,
                                              int m_per_split, int tile, int split_idx) {
  typedef bf16_t T;
  constexpr bool IS1X1 = MODE == 1;
  constexpr bool FAST = MODE != 0;
  constexpr int MT = 64;                        // m rows per step
  constexpr int PY = TN * 2 + 64, PA = TK * 2 + 64;   // row pitches (bytes)
  constexpr int STAGE = MT * (PY + PA);
  constexpr int CHY = TN / 8, CHA = TK / 8;     // 16-byte chunks per row
  constexpr int Y_PER = MT * CHY / 256, A_PER = MT * CHA / 256;
  constexpr int RSY = 256 / CHY, RSA = 256 / CHA;     // row stride between a thread's chunks
  constexpr int WTN = TN / 2, WTK = TK / 2;
  constexpr int FN = WTN / 32, FK = WTK / 32;

  extern __shared__ __attribute__((aligned(16))) char smem[];

  if (tile >= tiles_n * tiles_k) {   // trailing workgroups: bias gradient (column sums of dY)
    bias_tile<T, TN>(p, tile - tiles_n * tiles_k, split_idx, m_per_split, smem);
    return;
  }
  const int tile_n = tile / tiles_k, tile_k = tile - tile_n * tiles_k;
  const int split = split_idx;
  const int n0 = tile_n * TN, k0 = tile_k * TK;
  const int m_begin = split * m_per_split;
  int m_end = m_begin + m_per_split;
  if (m_end > p.M) m_end = p.M;
  const int n_steps = m_begin < m_end ? (m_end - m_begin + MT - 1) / MT : 0;

  if (threadIdx.x >= 256) {
    // =============================== loader waves ===============================
    const int tid = threadIdx.x - 256;
    const T* __restrict__ Ag = (const T*)p.a;
    const T* __restrict__ Yg = (const T*)p.dy;
    const int ycc = tid % CHY, yr0 = tid / CHY;
    const int acc_ = tid % CHA, ar0 = tid / CHA;
    const int yn = n0 + ycc * 8;
    const bool y_act = yn < p.N;
    const int ak = k0 + acc_ * 8;
    const bool a_act = ak < p.K;
    int a_ci = a_act ? ak : 0, a_kh = 0, a_kw = 0;
    // pixel state (b, oy, ox) of each of this thread's A rows, advanced by MT rows per step
    int pb[A_PER], poy[A_PER], pox[A_PER];
    int adv_b = 0, adv_y = 0, adv_x = 0;
    if constexpr (!IS1X1) {
      const int tap = a_ci / p.Cin;
      a_ci -= tap * p.Cin;
      a_kh = tap / p.KW;
      a_kw = tap - a_kh * p.KW;
      const int HoWo = p.Ho * p.Wo;
#pragma unroll
      for (int i = 0; i < A_PER; ++i) {
        const int m = m_begin + ar0 + i * RSA;
        pb[i] = m / HoWo;
        const int rem = m - pb[i] * HoWo;
        poy[i] = rem / p.Wo;
        pox[i] = rem - poy[i] * p.Wo;
      }
      adv_b = MT / HoWo;
      const int rr = MT - adv_b * HoWo;
      adv_y = rr / p.Wo;
      adv_x = rr - adv_y * p.Wo;
    }
    const u32x4 zero4 = {0u, 0u, 0u, 0u};
    // ---- scalar-only loaders (MODE 1 / 2): offsets relative to the split's first row
    constexpr unsigned OOB = 0x80000000u;       // == num_records
    const long long a_bias = MODE == 2 ? (long long)p.pad_t * p.W + p.pad_l : 0;   // offsets >= 0
    const __amdgpu_buffer_rsrc_t rsY = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(Yg + (long long)m_begin * p.ldy), 0, (int)OOB, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(Ag + ((long long)m_begin - a_bias) * p.lda), 0, (int)OOB, 0x00020000);
    unsigned y_vo[Y_PER], a_vo[A_PER];
    int lw_sh = 0;
    if constexpr (FAST) {
#pragma unroll
      for (int i = 0; i < Y_PER; ++i)
        y_vo[i] = y_act ? ((unsigned)(yr0 + i * RSY) * (unsigned)p.ldy + yn) * 2u : OOB;
#pragma unroll
      for (int i = 0; i < A_PER; ++i) {
        const unsigned r = ar0 + i * RSA + (MODE == 2 ? a_kh * p.W + a_kw : 0);
        a_vo[i] = a_act ? (r * (unsigned)p.lda + a_ci) * 2u : OOB;
      }
      while ((1 << lw_sh) < p.W) ++lw_sh;
    }
    const bool pow2 = (p.H & (p.H - 1)) == 0 && (p.W & (p.W - 1)) == 0;

    auto issue = [&](int mt, u32x4 (&ry)[Y_PER], u32x4 (&ra)[A_PER], unsigned& mask)
                     __attribute__((always_inline)) {
      mask = 0;
      if constexpr (FAST) {
        const int rel = mt - m_begin;            // wave-uniform
        const bool tail = mt + MT > m_end;
        const unsigned so_y = (unsigned)rel * (unsigned)p.ldy * 2u;
        const unsigned so_a = (unsigned)rel * (unsigned)p.lda * 2u;
#pragma unroll
        for (int i = 0; i < Y_PER; ++i) {
          unsigned vo = y_vo[i];
          if (tail && mt + yr0 + i * RSY >= m_end) vo = OOB;
          ry[i] = __builtin_amdgcn_raw_buffer_load_b128(rsY, (int)vo, (int)so_y, 0);
        }
#pragma unroll
        for (int i = 0; i < A_PER; ++i) {
          unsigned vo = a_vo[i];
          const int m = mt + ar0 + i * RSA;
          if constexpr (MODE == 2) {
            int ox, oy;
            if (pow2) {
              ox = m & (p.W - 1);
              oy = (m >> lw_sh) & (p.H - 1);
            } else {
              // any image size (28^2 latents of the 224^2 configurations): the row's pixel is carried along,
              // advanced by the MT rows of a step (steps are issued in order)
              ox = pox[i];
              oy = poy[i];
              pox[i] += adv_x;
              poy[i] += adv_y;
              if (pox[i] >= p.Wo) { pox[i] -= p.Wo; ++poy[i]; }
              if (poy[i] >= p.Ho) poy[i] -= p.Ho;
            }
            const bool bad = (unsigned)(oy + a_kh - p.pad_t) >= (unsigned)p.H ||
                             (unsigned)(ox + a_kw - p.pad_l) >= (unsigned)p.W;
            vo = bad ? OOB : vo;
          }
          if (tail && m >= m_end) vo = OOB;
          ra[i] = __builtin_amdgcn_raw_buffer_load_b128(rsA, (int)vo, (int)so_a, 0);
        }
        return;
      }
#pragma unroll
      for (int i = 0; i < Y_PER; ++i) {
        const int m = mt + yr0 + i * RSY;
        const bool ok = y_act && m < m_end;
        mask |= (ok ? 1u : 0u) << i;
        const long long off = ok ? (long long)m * p.ldy + yn : 0;
        ry[i] = *reinterpret_cast<const u32x4*>(Yg + off);
      }
#pragma unroll
      for (int i = 0; i < A_PER; ++i) {
        const int m = mt + ar0 + i * RSA;
        bool ok = a_act && m < m_end;
        long long off;
        if constexpr (IS1X1) {
          off = (long long)m * p.lda + a_ci;
        } else {
          int iy = poy[i] * p.stride - p.pad_t + a_kh, ix = pox[i] * p.stride - p.pad_l + a_kw;
          if (p.ups) {
            ok = ok && iy >= 0 && iy < 2 * p.H && ix >= 0 && ix < 2 * p.W;
            iy >>= 1;
            ix >>= 1;
          } else {
            ok = ok && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
          }
          off = (((long long)pb[i] * p.H + iy) * p.W + ix) * p.lda + a_ci;
          // advance the pixel state by MT rows
          pb[i] += adv_b; poy[i] += adv_y; pox[i] += adv_x;
          if (pox[i] >= p.Wo) { pox[i] -= p.Wo; ++poy[i]; }
          if (poy[i] >= p.Ho) { poy[i] -= p.Ho; ++pb[i]; }
        }
        mask |= (ok ? 1u : 0u) << (16 + i);
        ra[i] = *reinterpret_cast<const u32x4*>(Ag + (ok ? off : 0));
      }
    };
    auto commit = [&](char* buf, const u32x4 (&ry)[Y_PER], const u32x4 (&ra)[A_PER],
                      unsigned mask) __attribute__((always_inline)) {
      char* Ys = buf;
      char* As = buf + MT * PY;
#pragma unroll
      for (int i = 0; i < Y_PER; ++i)
        *reinterpret_cast<u32x4*>(Ys + (yr0 + i * RSY) * PY + ycc * 16) =
            (FAST || ((mask >> i) & 1u)) ? ry[i] : zero4;
#pragma unroll
      for (int i = 0; i < A_PER; ++i)
        *reinterpret_cast<u32x4*>(As + (ar0 + i * RSA) * PA + acc_ * 16) =
            (FAST || ((mask >> (16 + i)) & 1u)) ? ra[i] : zero4;
    };

    u32x4 ry0[Y_PER], ra0[A_PER], ry1[Y_PER], ra1[A_PER];
    unsigned mk0 = 0, mk1 = 0;
    if (n_steps > 0) issue(m_begin, ry0, ra0, mk0);
    if (n_steps > 1) issue(m_begin + MT, ry1, ra1, mk1);
    for (int s = 0; s < n_steps; s += 2) {
      commit(smem, ry0, ra0, mk0);
      if (s + 2 < n_steps) issue(m_begin + (s + 2) * MT, ry0, ra0, mk0);
      __syncthreads();
      if (s + 1 < n_steps) {
        commit(smem + STAGE, ry1, ra1, mk1);
        if (s + 3 < n_steps) issue(m_begin + (s + 3) * MT, ry1, ra1, mk1);
        __syncthreads();
      }
    }
    return;
  }

  // ================================= MFMA waves =================================
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wn = wave >> 1, wk = wave & 1;
  f32x16 acc[FN][FK];
#pragma unroll
  for (int i = 0; i < FN; ++i)
#pragma unroll
    for (int j = 0; j < FK; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // lane (g = lane >> 4, t = lane & 15) addresses the 8-byte piece (row t / 4, 4 columns at
  // (t % 4) * 4) of its group's [4 m][16 n] block: m rows (g >> 1) * 8 (+4 for the second read),
  // columns (g & 1) * 16 of the 32-row fragment.
  const int g = lane >> 4, t = lane & 15;
  const int lrow = (g >> 1) * 8 + (t >> 2);
  const int lcol = (g & 1) * 16 + (t & 3) * 4;
  const int yoff = lrow * PY + (wn * WTN + lcol) * 2;
  const int aoff = MT * PY + lrow * PA + (wk * WTK + lcol) * 2;
  constexpr int KS = MT / 16;
  auto read_frags = [&](const char* Yf, const char* Af, int ks, s16x8 (&fy)[FN], s16x8 (&fa)[FK])
                        __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < FN; ++i) {
      const s16x4 lo =
          __builtin_amdgcn_ds_read_tr16_b64_v4i16(SDMI_LDS_V4(Yf + (ks * 16) * PY + i * 64));
      const s16x4 hi =
          __builtin_amdgcn_ds_read_tr16_b64_v4i16(SDMI_LDS_V4(Yf + (ks * 16 + 4) * PY + i * 64));
      fy[i] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    }
#pragma unroll
    for (int j = 0; j < FK; ++j) {
      const s16x4 lo =
          __builtin_amdgcn_ds_read_tr16_b64_v4i16(SDMI_LDS_V4(Af + (ks * 16) * PA + j * 64));
      const s16x4 hi =
          __builtin_amdgcn_ds_read_tr16_b64_v4i16(SDMI_LDS_V4(Af + (ks * 16 + 4) * PA + j * 64));
      fa[j] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    }
  };
  for (int s = 0; s < n_steps; ++s) {
    __syncthreads();                 // stage s & 1 holds step s
    const char* buf = smem + (s & 1) * STAGE;
    const char* Yf = buf + yoff;
    const char* Af = buf + aoff;
    // fragments of k-step ks+1 are fetched under the MFMAs of k-step ks
    s16x8 fy[2][FN], fa[2][FK];
    read_frags(Yf, Af, 0, fy[0], fa[0]);
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      if (ks + 1 < KS) read_frags(Yf, Af, ks + 1, fy[(ks + 1) & 1], fa[(ks + 1) & 1]);
      __builtin_amdgcn_sched_barrier(0);   // keep the prefetch ahead of this k-step's MFMAs
#pragma unroll
      for (int i = 0; i < FN; ++i)
#pragma unroll
        for (int j = 0; j < FK; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
              __builtin_bit_cast(bf16x8, fy[ks & 1][i]), __builtin_bit_cast(bf16x8, fa[ks & 1][j]),
              acc[i][j], 0, 0, 0);
    }
  }
  const long long NK = (long long)p.N * p.K;
  const bool direct = p.splits == 1;
  float* ws = direct ? p.dw : p.workspace + (long long)split * NK;
  const bool accum = direct && p.accumulate;
  const int col_l = lane & 31, row_l = (lane >> 5) * 4;
#pragma unroll
  for (int i = 0; i < FN; ++i) {
#pragma unroll
    for (int j = 0; j < FK; ++j) {
      const int k = k0 + wk * WTK + j * 32 + col_l;
      const int nb = n0 + wn * WTN + i * 32 + row_l;
      float old[16];
      if (accum) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int n = nb + (r & 3) + 8 * (r >> 2);
          old[r] = (n < p.N && k < p.K) ? ws[(long long)n * p.K + k] : 0.f;
        }
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int n = nb + (r & 3) + 8 * (r >> 2);
        if (n < p.N && k < p.K)
          ws[(long long)n * p.K + k] = accum ? old[r] + acc[i][j][r] : acc[i][j][r];
      }
    }
  }
}


// Fold of the M-split partials, deterministic: output vector i is owned by G adjacent lanes (G = 1 below 64
// splits, else 8); lane g sums the contiguous split range [g * ceil(S / G), ...) in split order, the G range sums are
// added in range order behind the destination's previous value.  With one lane per output a 256-split fold
// (the direct 3x3 kernel's one slot per CU) was a chain of 32 dependent load batches on 36 workgroups: 72 us.
__host__ __device__ inline int wgrad_fold_lanes(int splits) { return splits >= 64 ? 8 : 1; }
// workgroups of THREADS threads that cover the fold of one problem (callers cap it)
static inline long long wgrad_fold_blocks(const SdmiWgradArgs& a, int threads = 256) {
  return ((long long)a.N * a.K / 4 * wgrad_fold_lanes(a.splits) + threads - 1) / threads;
}

template <int THREADS = 256>
__device__ __forceinline__ void wgrad_reduce_body(const SdmiWgradArgs& p, int blk, int nblk) {
  const long long total = (long long)p.N * p.K;
  const long long total4 = total >> 2;
  const int G = wgrad_fold_lanes(p.splits);
  const int per = (p.splits + G - 1) / G;
  const int gl = threadIdx.x & (G - 1);
  const int k0 = gl * per, k1 = min(p.splits, k0 + per);
  for (long long i = ((long long)blk * THREADS + threadIdx.x) / G;; i += (long long)nblk * THREADS / G) {
    // (the G lanes of an output leave the loop together: i is the same for all of them)
    if (i >= total4) break;
    const f32x4* src = reinterpret_cast<const f32x4*>(p.workspace) + i;
    f32x4* dst = reinterpret_cast<f32x4*>(p.dw) + i;
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    if (G == 1 && p.accumulate) s = *dst;
    int k = k0;
    for (; k + 8 <= k1; k += 8) {
      f32x4 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = src[(long long)(k + u) * total4];
#pragma unroll
      for (int u = 0; u < 8; ++u) s += v[u];
    }
    for (; k < k1; ++k) s += src[(long long)k * total4];
    if (G > 1) {
      f32x4 t = {0.f, 0.f, 0.f, 0.f};
      if (p.accumulate && gl == 0) t = *dst;
      const int base = (threadIdx.x & 63) & ~(G - 1);
      for (int g = 0; g < G; ++g) {
#pragma unroll
        for (int c = 0; c < 4; ++c) t[c] += __shfl(s[c], base + g, 64);
      }
      s = t;
    }
    if (gl == 0) *dst = s;
  }
  if (p.dbias)
    for (long long n = ((long long)blk * THREADS + threadIdx.x) / G;; n += (long long)nblk * THREADS / G) {
      if (n >= p.N) break;
      const float* src = p.workspace + (long long)p.splits * total + n;
      float s = (G == 1 && p.accumulate) ? p.dbias[n] : 0.f;
      for (int k = k0; k < k1; ++k) s += src[(long long)k * p.N];
      if (G > 1) {
        float t = (p.accumulate && gl == 0) ? p.dbias[n] : 0.f;
        const int base = (threadIdx.x & 63) & ~(G - 1);
        for (int g = 0; g < G; ++g) t += __shfl(s, base + g, 64);
        s = t;
      }
      if (gl == 0) p.dbias[n] = s;
    }
}


}  // namespace
