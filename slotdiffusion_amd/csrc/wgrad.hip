// Weight gradient of conv / linear as an MFMA GEMM contracting over the B*Ho*Wo rows
// (include/sdmi.h: sdmi_wgrad):
//     dW[n][k] = sum_m dY[m][n] * A[m][k],    k = (kh, kw, ci),  A gathered like the forward.
//
// Both operands are m-major in HBM while the contraction runs over m, so the staging pass
// transposes: each loader thread loads a VEC x VEC block (VEC rows m, one 16-byte vector of n or k
// each), transposes it in registers (v_perm_b32), and writes VEC 16-byte vectors into LDS tiles
// laid out [n][m] / [k][m] -- m-contiguous rows -- after which the MFMA part is the forward
// kernel's: lane l reads row (l&31), bytes [ks*32 + (l>>5)*16, +16) of both tiles (conflict-free
// 16-byte padded pitch).
//
// Wave specialisation: a workgroup is 8 waves -- 4 MFMA waves (2x2 over the output tile) and 4
// loader waves, one of each per SIMD, so the matrix pipe of a SIMD runs under the other wave's
// address / transpose VALU work.  The loaders keep two register sets (two m-steps) of global loads
// in flight and fill a double-buffered LDS image one step ahead of the MFMA waves; one barrier per
// m-step.  (The previous single-role version serialised load latency, transposes and MFMAs in
// every wave: 3.4 us per 128-row step.)
//
// Parallelism: the output has only (N/128)*(K/128) tiles, so M is split across `splits`
// workgroups per tile; partials go to a workspace [splits][N][K] and a deterministic second kernel
// reduces them (optionally accumulating into dW).  No atomics.  Trailing workgroups of the grid
// compute the bias gradient (column sums of dY).
#include "common.h"

namespace {

template <typename T> struct WCfg;
template <> struct WCfg<bf16_t> { [[maybe_unused]] static constexpr int MTB = 256; };  // 128 m per iteration
template <> struct WCfg<float> { static constexpr int MTB = 128; };   // 32 m per iteration

// in-register VEC x VEC transpose of 16-byte vectors
__device__ __forceinline__ void transpose_block(const u32x4 (&r)[8], u32x4 (&c)[8], bf16_t) {
#pragma unroll
  for (int n = 0; n < 8; ++n) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const unsigned a = r[2 * e][n >> 1], b = r[2 * e + 1][n >> 1];
      // {b.hi16 : a.hi16} / {b.lo16 : a.lo16}: one v_perm_b32 each
      c[n][e] = __builtin_amdgcn_perm(b, a, (n & 1) ? 0x07060302u : 0x05040100u);
    }
  }
}
__device__ __forceinline__ void transpose_block(const u32x4 (&r)[4], u32x4 (&c)[4], float) {
#pragma unroll
  for (int n = 0; n < 4; ++n)
#pragma unroll
    for (int e = 0; e < 4; ++e) c[n][e] = r[e][n];
}

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

template <typename T, int TN, int TK, bool IS1X1>
__global__ __launch_bounds__(512) void wgrad_kernel(SdmiWgradArgs p, int tiles_n, int tiles_k,
                                                    int m_per_split) {
  constexpr int VEC = 16 / sizeof(T);
  constexpr int MTB = WCfg<T>::MTB;
  constexpr int MT = MTB / sizeof(T);        // m rows per step
  constexpr int ROWB = MTB + 16;
  constexpr int KSTEPS = MTB / 32;
  constexpr int MBLK = MT / VEC;             // blocks along m
  constexpr int Y_BLOCKS = (TN / VEC) * MBLK;
  constexpr int A_BLOCKS = (TK / VEC) * MBLK;
  constexpr int Y_PER = (Y_BLOCKS + 255) / 256;
  constexpr int A_PER = (A_BLOCKS + 255) / 256;
  constexpr int WTN = TN / 2, WTK = TK / 2;
  constexpr int FN = WTN / 32, FK = WTK / 32;
  constexpr int BUFB = (TN + TK) * ROWB;     // one LDS stage: Ys [TN][ROWB] | As [TK][ROWB]

  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tile = blockIdx.x;
  if (tile >= tiles_n * tiles_k) {   // trailing workgroups: bias gradient (column sums of dY)
    bias_tile<T, TN>(p, tile - tiles_n * tiles_k, blockIdx.y, m_per_split, smem);
    return;
  }
  const int tile_n = tile / tiles_k, tile_k = tile - tile_n * tiles_k;
  const int split = blockIdx.y;
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
    const int HoWo = p.Ho * p.Wo;
    int y_mb[Y_PER], y_n[Y_PER];
    bool y_act[Y_PER];
#pragma unroll
    for (int i = 0; i < Y_PER; ++i) {
      const int blk = tid + i * 256;
      y_mb[i] = blk % MBLK;
      y_n[i] = n0 + (blk / MBLK) * VEC;
      y_act[i] = blk < Y_BLOCKS && y_n[i] < p.N;
    }
    int a_mb[A_PER], a_kb[A_PER], a_ci[A_PER], a_kh[A_PER], a_kw[A_PER];
    bool a_act[A_PER];
#pragma unroll
    for (int i = 0; i < A_PER; ++i) {
      const int blk = tid + i * 256;
      a_mb[i] = blk % MBLK;
      a_kb[i] = blk / MBLK;
      const int k = k0 + a_kb[i] * VEC;
      a_act[i] = blk < A_BLOCKS && k < p.K;
      const int kk = a_act[i] ? k : 0;
      if constexpr (IS1X1) {
        a_ci[i] = kk; a_kh[i] = 0; a_kw[i] = 0;
      } else {
        const int tap = kk / p.Cin;
        a_ci[i] = kk - tap * p.Cin;
        a_kh[i] = tap / p.KW;
        a_kw[i] = tap - a_kh[i] * p.KW;
      }
    }
    const u32x4 zero4 = {0u, 0u, 0u, 0u};

    // one m-step of global loads into a register set (masks: bit j = row j of the block valid)
    auto issue = [&](int mt, u32x4 (&ry)[Y_PER][VEC], u32x4 (&ra)[A_PER][VEC],
                     unsigned (&ymask)[Y_PER], unsigned (&amask)[A_PER])
                     __attribute__((always_inline)) {
#pragma unroll
      for (int i = 0; i < Y_PER; ++i) {
        ymask[i] = 0;
        const int mrow = mt + y_mb[i] * VEC;
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
          const int m = mrow + j;
          const bool ok = y_act[i] && m < m_end;
          ymask[i] |= (ok ? 1u : 0u) << j;
          const long long off = ok ? (long long)m * p.ldy + y_n[i] : 0;
          ry[i][j] = *reinterpret_cast<const u32x4*>(Yg + off);
        }
      }
#pragma unroll
      for (int i = 0; i < A_PER; ++i) {
        amask[i] = 0;
        const int mrow = mt + a_mb[i] * VEC;
        if constexpr (IS1X1) {
#pragma unroll
          for (int j = 0; j < VEC; ++j) {
            const int m = mrow + j;
            const bool ok = a_act[i] && m < m_end;
            amask[i] |= (ok ? 1u : 0u) << j;
            const long long off = ok ? (long long)m * p.lda + a_ci[i] : 0;
            ra[i][j] = *reinterpret_cast<const u32x4*>(Ag + off);
          }
        } else {
          int b = mrow / HoWo;
          const int rem = mrow - b * HoWo;
          int oy = rem / p.Wo;
          int ox = rem - oy * p.Wo;
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
            if (++ox == p.Wo) { ox = 0; if (++oy == p.Ho) { oy = 0; ++b; } }   // next pixel
          }
        }
      }
    };
    // transpose a register set into one LDS stage
    auto commit = [&](char* buf, u32x4 (&ry)[Y_PER][VEC], u32x4 (&ra)[A_PER][VEC],
                      unsigned (&ymask)[Y_PER], unsigned (&amask)[A_PER])
                      __attribute__((always_inline)) {
      constexpr unsigned FULL = (1u << VEC) - 1u;
      char* Ys = buf;
      char* As = buf + TN * ROWB;
#pragma unroll
      for (int i = 0; i < Y_PER; ++i) {
        if (tid + i * 256 >= Y_BLOCKS) continue;
        u32x4 in[VEC], out[VEC];
#pragma unroll
        for (int j = 0; j < VEC; ++j) in[j] = ry[i][j];
        if (ymask[i] != FULL) {
#pragma unroll
          for (int j = 0; j < VEC; ++j) in[j] = ((ymask[i] >> j) & 1u) ? in[j] : zero4;
        }
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
        for (int j = 0; j < VEC; ++j) in[j] = ra[i][j];
        if (amask[i] != FULL) {
#pragma unroll
          for (int j = 0; j < VEC; ++j) in[j] = ((amask[i] >> j) & 1u) ? in[j] : zero4;
        }
        transpose_block(in, out, T());
#pragma unroll
        for (int c = 0; c < VEC; ++c)
          *reinterpret_cast<u32x4*>(As + (a_kb[i] * VEC + c) * ROWB + a_mb[i] * 16) = out[c];
      }
    };

    u32x4 ry0[Y_PER][VEC], ra0[A_PER][VEC], ry1[Y_PER][VEC], ra1[A_PER][VEC];
    unsigned ym0[Y_PER], am0[A_PER], ym1[Y_PER], am1[A_PER];
    if (n_steps > 0) issue(m_begin, ry0, ra0, ym0, am0);
    if (n_steps > 1) issue(m_begin + MT, ry1, ra1, ym1, am1);
    for (int s = 0; s < n_steps; s += 2) {
      commit(smem, ry0, ra0, ym0, am0);
      if (s + 2 < n_steps) issue(m_begin + (s + 2) * MT, ry0, ra0, ym0, am0);
      __syncthreads();
      if (s + 1 < n_steps) {
        commit(smem + BUFB, ry1, ra1, ym1, am1);
        if (s + 3 < n_steps) issue(m_begin + (s + 3) * MT, ry1, ra1, ym1, am1);
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

  const int frag_row = lane & 31, frag_kb = (lane >> 5) * 16;
  for (int s = 0; s < n_steps; ++s) {
    __syncthreads();                 // stage s & 1 holds step s
    const char* buf = smem + (s & 1) * BUFB;
    const char* Yf = buf + (wn * WTN + frag_row) * ROWB + frag_kb;
    const char* Af = buf + TN * ROWB + (wk * WTK + frag_row) * ROWB + frag_kb;
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
  }
  // ---- epilogue.  splits == 1: straight into dW (one launch).  Otherwise the partial tile goes
  // to ws[split][n][k] and wgrad_reduce_kernel folds the partials in split order.  (A variant in
  // which the last workgroup of a tile folds them was measured 3x slower: one workgroup cannot
  // pull splits*64 KB fast.)
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
      if (accum) {   // all 16 loads in flight before the first add (not load-add-store chains)
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
            const int ox = m & (p.W - 1), oy = (m >> lw_sh) & (p.H - 1);
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

template <int TN, int TK, int MODE>
__global__ __launch_bounds__(512) void wgrad_tr_kernel(SdmiWgradArgs p, int tiles_n, int tiles_k,
                                                       int m_per_split) {
  wgrad_tr_body<TN, TK, MODE>(p, tiles_n, tiles_k, m_per_split, blockIdx.x, blockIdx.y);
}

// ------------------------------------------------------------------------------------------
// Grouped launch: the workgroups of up to 16 independent 1x1 / linear problems in one grid.  The
// descriptor travels BY VALUE in the kernel arguments (a HIP graph captures it with the launch; no
// table in device memory).  item_begin[i] .. item_begin[i+1] are problem i's workgroups, laid out
// [split][tile (+ bias tiles)] like the single-problem grid.
// ------------------------------------------------------------------------------------------
constexpr int WG_MAX = 16;
struct WgradGroup {
  int n;
  int item_begin[WG_MAX + 1];
  int tiles_n[WG_MAX], tiles_k[WG_MAX], per_split[WG_MAX], mps[WG_MAX];
  int red_begin[WG_MAX + 1];        // fold launch: 256-thread blocks of problem i (0 when splits == 1)
  SdmiWgradArgs p[WG_MAX];
};

__global__ __launch_bounds__(512) void wgrad_group_kernel(WgradGroup g) {
  const int bid = blockIdx.x;
  int i = 0;
  while (i + 1 < g.n && bid >= g.item_begin[i + 1]) ++i;      // uniform: scalar loop over <= 16 entries
  const int local = bid - g.item_begin[i];
  const int split = local / g.per_split[i];
  const int tile = local - split * g.per_split[i];
  wgrad_tr_body<128, 128, 1>(g.p[i], g.tiles_n[i], g.tiles_k[i], g.mps[i], tile, split);
}

__device__ __forceinline__ void wgrad_reduce_body(const SdmiWgradArgs& p, int blk, int nblk) {
  const long long total = (long long)p.N * p.K;
  const long long total4 = total >> 2;
  for (long long i = (long long)blk * 256 + threadIdx.x; i < total4; i += (long long)nblk * 256) {
    const f32x4* src = reinterpret_cast<const f32x4*>(p.workspace) + i;
    f32x4* dst = reinterpret_cast<f32x4*>(p.dw) + i;
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    if (p.accumulate) s = *dst;
    int k = 0;
    for (; k + 8 <= p.splits; k += 8) {
      f32x4 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = src[(long long)(k + u) * total4];
#pragma unroll
      for (int u = 0; u < 8; ++u) s += v[u];
    }
    for (; k < p.splits; ++k) s += src[(long long)k * total4];
    *dst = s;
  }
  if (p.dbias)
    for (long long n = (long long)blk * 256 + threadIdx.x; n < p.N; n += (long long)nblk * 256) {
      float s = p.accumulate ? p.dbias[n] : 0.f;
      for (int k = 0; k < p.splits; ++k)
        s += p.workspace[(long long)p.splits * total + (long long)k * p.N + n];
      p.dbias[n] = s;
    }
}

__global__ __launch_bounds__(256) void wgrad_group_reduce_kernel(WgradGroup g) {
  const int bid = blockIdx.x;
  int i = 0;
  while (i + 1 < g.n && bid >= g.red_begin[i + 1]) ++i;
  wgrad_reduce_body(g.p[i], bid - g.red_begin[i], g.red_begin[i + 1] - g.red_begin[i]);
}

// Fold the split partials in split order (deterministic).  One thread per 16-byte output vector
// (N*K is a multiple of 4), 8 partial loads in flight per thread before the ordered adds.
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(SdmiWgradArgs p) {
  wgrad_reduce_body(p, blockIdx.x, gridDim.x);
}

template <typename T, int TN, int TK, bool IS1X1>
int launch_wgrad(const SdmiWgradArgs& a, hipStream_t st) {
  constexpr int MTB = WCfg<T>::MTB;
  constexpr int smem = 2 * (TN + TK) * (MTB + 16);
  auto kern = wgrad_kernel<T, TN, TK, IS1X1>;
  SDMI_OPTIN_LDS(kern, smem, "wgrad");
  const int tiles_n = (a.N + TN - 1) / TN, tiles_k = (a.K + TK - 1) / TK;
  const int MT = MTB / (int)sizeof(T);
  int mps = (a.M + a.splits - 1) / a.splits;
  mps = (mps + MT - 1) / MT * MT;
  dim3 grid(tiles_n * tiles_k + (a.dbias ? tiles_n : 0), a.splits);
  hipLaunchKernelGGL(kern, grid, dim3(512), smem, st, a, tiles_n, tiles_k, mps);
  return sdmi_check_launch("wgrad");
}

template <int TN, int TK, int MODE>
int launch_wgrad_tr(const SdmiWgradArgs& a, hipStream_t st) {
  constexpr int MT = 64;
  constexpr int smem = 2 * MT * (TN * 2 + 64 + TK * 2 + 64);
  auto kern = wgrad_tr_kernel<TN, TK, MODE>;
  SDMI_OPTIN_LDS(kern, smem, "wgrad");
  const int tiles_n = (a.N + TN - 1) / TN, tiles_k = (a.K + TK - 1) / TK;
  int mps = (a.M + a.splits - 1) / a.splits;
  mps = (mps + MT - 1) / MT * MT;
  dim3 grid(tiles_n * tiles_k + (a.dbias ? tiles_n : 0), a.splits);
  hipLaunchKernelGGL(kern, grid, dim3(512), smem, st, a, tiles_n, tiles_k, mps);
  return sdmi_check_launch("wgrad");
}

static bool wgrad_is1x1(const SdmiWgradArgs& a) {
  return a.KH == 1 && a.KW == 1 && a.stride == 1 && a.pad_t == 0 && a.pad_l == 0 && !a.ups &&
         a.H == a.Ho && a.W == a.Wo;
}

int dispatch_wgrad_f32(const SdmiWgradArgs& a, hipStream_t st) {
  const bool small = a.N <= 64 || a.K <= 64;
  const bool is1x1 = wgrad_is1x1(a);
  if (small)
    return is1x1 ? launch_wgrad<float, 64, 64, true>(a, st)
                 : launch_wgrad<float, 64, 64, false>(a, st);
  return is1x1 ? launch_wgrad<float, 128, 128, true>(a, st)
               : launch_wgrad<float, 128, 128, false>(a, st);
}

int dispatch_wgrad_bf16(const SdmiWgradArgs& a, hipStream_t st) {
  const bool is1x1 = wgrad_is1x1(a);
  const bool n64 = a.N <= 64, k64 = a.K <= 64;
  // "same" stride-1 convolution on a power-of-two image: the input pixel is linear in m
  const bool lin = !is1x1 && !a.ups && a.stride == 1 && a.H == a.Ho && a.W == a.Wo &&
                   (a.H & (a.H - 1)) == 0 && (a.W & (a.W - 1)) == 0;
  // the scalar-only loaders address a split's rows with 31-bit byte offsets
  const long long mps = ((long long)a.M + a.splits - 1) / a.splits + 64;
  const long long ld = a.lda > a.ldy ? a.lda : a.ldy;
  const bool fits = (mps + (long long)(a.KH + 1) * a.W + 64) * ld * 2 < (1ll << 31);
#define WG_TR(TN, TK)                                                              \
  (is1x1 && fits ? launch_wgrad_tr<TN, TK, 1>(a, st)                               \
   : lin && fits ? launch_wgrad_tr<TN, TK, 2>(a, st) : launch_wgrad_tr<TN, TK, 0>(a, st))
  if (n64) return k64 ? WG_TR(64, 64) : WG_TR(64, 128);
  return k64 ? WG_TR(128, 64) : WG_TR(128, 128);
#undef WG_TR
}

}  // namespace

extern "C" int sdmi_wgrad_group(const SdmiWgradGroupArgs* ga, void* stream) {
  SDMI_REQUIRE(ga && ga->problems && ga->n >= 1 && ga->n <= WG_MAX, "1 .. 16 problems");
  const SdmiWgradArgs* ps = (const SdmiWgradArgs*)ga->problems;
  hipStream_t st = (hipStream_t)stream;
  WgradGroup g;
  g.n = ga->n;
  int items = 0, red = 0;
  for (int i = 0; i < ga->n; ++i) {
    const SdmiWgradArgs& a = ps[i];
    SDMI_REQUIRE(a.a && a.dy && a.dw, "null pointer");
    SDMI_REQUIRE(a.dtype == SDMI_BF16 && wgrad_is1x1(a) && a.N > 64 && a.K > 64,
                 "grouped wgrad: bf16 1x1 / linear problems with N, K > 64");
    SDMI_REQUIRE(a.K == a.Cin && a.Cin % 8 == 0 && a.lda % 8 == 0 && a.ldy % 8 == 0 && a.M == a.B * a.Ho * a.Wo,
                 "bad geometry");
    SDMI_REQUIRE(a.splits >= 1 && (a.splits == 1 || a.workspace), "splits / workspace");
    const long long mps_ = ((long long)a.M + a.splits - 1) / a.splits + 64;
    const long long ld = a.lda > a.ldy ? a.lda : a.ldy;
    SDMI_REQUIRE((mps_ + 64) * ld * 2 < (1ll << 31), "split too large for 31-bit offsets");
    g.p[i] = a;
    g.tiles_n[i] = (a.N + 127) / 128;
    g.tiles_k[i] = (a.K + 127) / 128;
    g.per_split[i] = g.tiles_n[i] * g.tiles_k[i] + (a.dbias ? g.tiles_n[i] : 0);
    int mps = (a.M + a.splits - 1) / a.splits;
    g.mps[i] = (mps + 63) / 64 * 64;
    g.item_begin[i] = items;
    items += g.per_split[i] * a.splits;
    g.red_begin[i] = red;
    if (a.splits > 1) {
      long long blocks = ((long long)a.N * a.K / 4 + 255) / 256;
      red += (int)(blocks > 512 ? 512 : blocks);
    }
  }
  for (int i = ga->n; i <= WG_MAX; ++i) { g.item_begin[i] = items; g.red_begin[i] = red; }
  constexpr int smem = 2 * 64 * (128 * 2 + 64 + 128 * 2 + 64);
  SDMI_OPTIN_LDS(wgrad_group_kernel, smem, "wgrad group");
  hipLaunchKernelGGL(wgrad_group_kernel, dim3(items), dim3(512), smem, st, g);
  int rc = sdmi_check_launch("wgrad group");
  if (rc || red == 0) return rc;
  hipLaunchKernelGGL(wgrad_group_reduce_kernel, dim3(red), dim3(256), 0, st, g);
  return sdmi_check_launch("wgrad group reduce");
}

extern "C" int sdmi_wgrad_fold_group(const SdmiWgradGroupArgs* ga, void* stream) {
  SDMI_REQUIRE(ga && ga->problems && ga->n >= 1 && ga->n <= WG_MAX, "1 .. 16 problems");
  const SdmiWgradArgs* ps = (const SdmiWgradArgs*)ga->problems;
  WgradGroup g;
  g.n = ga->n;
  int red = 0;
  for (int i = 0; i < ga->n; ++i) {
    SDMI_REQUIRE(ps[i].dw && ps[i].workspace && ps[i].splits > 1 && ps[i].N > 0 && ps[i].K > 0 &&
                 ((long long)ps[i].N * ps[i].K) % 4 == 0, "bad problem");
    g.p[i] = ps[i];
    g.item_begin[i] = 0;
    g.red_begin[i] = red;
    long long blocks = ((long long)ps[i].N * ps[i].K / 4 + 255) / 256;
    red += (int)(blocks > 512 ? 512 : blocks);
  }
  for (int i = ga->n; i <= WG_MAX; ++i) { g.item_begin[i] = 0; g.red_begin[i] = red; }
  hipLaunchKernelGGL(wgrad_group_reduce_kernel, dim3(red), dim3(256), 0, (hipStream_t)stream, g);
  return sdmi_check_launch("wgrad fold group");
}

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
  rc = a->dtype == SDMI_BF16 ? dispatch_wgrad_bf16(*a, st) : dispatch_wgrad_f32(*a, st);
  if (rc || a->splits == 1 || a->defer_fold) return rc;
  const long long total = (long long)a->N * a->K;     // K % 4 == 0 (Cin % vec == 0)
  int blocks = (int)((total / 4 + 255) / 256);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(blocks), dim3(256), 0, st, *a);
  return sdmi_check_launch("wgrad reduce");
}
