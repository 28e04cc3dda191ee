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
#include "wgrad_body.h"

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

// (body of wgrad_kernel: `tile` / `split` are the launch's block indices, or a slice of a grouped grid)
template <typename T, int TN, int TK, bool IS1X1>
__device__ __forceinline__ void wgrad_std_body(const SdmiWgradArgs& p, int tiles_n, int tiles_k, int m_per_split,
                                               const int tile, const int split, char* smem) {
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

  if (tile >= tiles_n * tiles_k) {   // trailing workgroups: bias gradient (column sums of dY)
    bias_tile<T, TN>(p, tile - tiles_n * tiles_k, split, m_per_split, smem);
    return;
  }
  const int tile_n = tile / tiles_k, tile_k = tile - tile_n * tiles_k;
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

template <typename T, int TN, int TK, bool IS1X1>
__global__ __launch_bounds__(512) void wgrad_kernel(SdmiWgradArgs p, int tiles_n, int tiles_k,
                                                    int m_per_split) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  wgrad_std_body<T, TN, TK, IS1X1>(p, tiles_n, tiles_k, m_per_split, (int)blockIdx.x, (int)blockIdx.y, smem);
}

template <int TN, int TK, int MODE>
__global__ __launch_bounds__(512) void wgrad_tr_kernel(SdmiWgradArgs p, int tiles_n, int tiles_k,
                                                       int m_per_split) {
  // XCD-aware order (as in bwd_pair.hip): the tiles of one M split read the same rows of x and dY, so each XCD takes a
  // contiguous range of (split, tile) indices instead of every eighth one (the dispatcher walks x fastest and places
  // linear block id b on XCD b % 8).  Same work per workgroup, bit-identical results.
  const int gx = (int)gridDim.x, nb = gx * (int)gridDim.y;
  const int j = (int)blockIdx.y * gx + (int)blockIdx.x;
  const int xc = j & 7, q8 = nb >> 3, r8 = nb & 7;
  const int idx = (xc < r8 ? xc * (q8 + 1) : r8 * (q8 + 1) + (xc - r8) * q8) + (j >> 3);
  const int split = idx / gx;
  wgrad_tr_body<TN, TK, MODE>(p, tiles_n, tiles_k, m_per_split, idx - split * gx, split);
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

// fp32 form: the exact-fp32 64 x 64 tiles of the Slot Attention / predictor layers (M = images x slots rows: each of
// these problems alone is 9 - 36 workgroups and ~20 us of latency; dispatch_wgrad_f32)
__global__ __launch_bounds__(512) void wgrad_group_f32_kernel(WgradGroup g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int bid = blockIdx.x;
  int i = 0;
  while (i + 1 < g.n && bid >= g.item_begin[i + 1]) ++i;
  const int local = bid - g.item_begin[i];
  const int split = local / g.per_split[i];
  const int tile = local - split * g.per_split[i];
  wgrad_std_body<float, 64, 64, true>(g.p[i], g.tiles_n[i], g.tiles_k[i], g.mps[i], tile, split, smem);
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


// ------------------------------------------------------------------------------------------
// Direct weight gradient of the 64 -> 64 channel 3x3 convolutions at full resolution (slot encoder /
// VQ-VAE at 128^2: M = 1M pixels, N = 64, K = 576).  As an implicit GEMM (wgrad_tr_kernel<64, 128, 2>) this
// shape re-fetches every activation row for each of the 9 taps and dY for each of the 5 column blocks --
// 1.9 GB of L2 -> LDS traffic per launch, 211 us.  Here (the weight-gradient twin of conv3x3_c64_kernel):
//   * persistent workgroup s of `splits` (256 threads = 4 waves, one per CU) walks output tiles of 4 image
//     rows x 64 pixels; the tile's dY (256 px x 64 ch) and its 6 x 66 pixel input halo (zeros outside the
//     image) are staged ONCE and serve all 9 taps: a tap only shifts the fragment address;
//   * the contraction runs over pixels, both operands are pixel-major: fragments come from the transposing
//     LDS read (ds_read_b64_tr_b16, pixel pitch 192 B = 64 mod 256: conflict free), 16 pixels of an image
//     row per MFMA k-step;
//   * wave w owns output channels [32 (w & 1), +32) x taps {w >> 1, + 2, ...} (5 / 5 / 4 / 4 of the 18
//     (tap, half) units): 160 accumulator registers, kept across ALL tiles of the workgroup;
//   * the next tile's operands are prefetched into registers under the MFMAs;
//   * the workgroup's partial dW [64][576] (and column sums of dY) go to slot s of the M-split workspace,
//     folded by the usual deterministic reduce.
constexpr int W33_PIX = 192;
constexpr int W33_HALO = 6 * 66, W33_DY = 4 * 64;
constexpr int W33_HV = (W33_HALO * 8 + 255) / 256, W33_DV = W33_DY * 8 / 256;     // uint4 per thread
constexpr int W33_SMEM = (W33_HALO + W33_DY) * W33_PIX;

__global__ __launch_bounds__(256) void wgrad3x3_c64_kernel(SdmiWgradArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const Xs = smem;
  char* const Ys = smem + W33_HALO * W33_PIX;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bf16_t* __restrict__ Ag = (const bf16_t*)p.a;
  const bf16_t* __restrict__ Yg = (const bf16_t*)p.dy;
  const int tiles_x = p.W / 64, tiles_y = p.H / 4;
  const int n_tiles = p.B * tiles_y * tiles_x;
  const int split = blockIdx.x, nsplit = gridDim.x;

  u32x4 prex[W33_HV], prey[W33_DV];
  float bsum[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) bsum[j] = 0.f;
  auto fetch = [&](int t) __attribute__((always_inline)) {
    const int b = t / (tiles_y * tiles_x), r = t - b * tiles_y * tiles_x;
    const int ty = r / tiles_x, tx = r - ty * tiles_x;
    const int iy0 = ty * 4 - 1, ix0 = tx * 64 - 1;
#pragma unroll
    for (int i = 0; i < W33_HV; ++i) {
      const int v = tid + i * 256, px = v >> 3, ch = v & 7;
      const int hy = px / 66, hx = px - hy * 66;
      const int iy = iy0 + hy, ix = ix0 + hx;
      const bool ok = px < W33_HALO && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
      prex[i] = ok ? *reinterpret_cast<const u32x4*>(Ag + ((long long)(b * p.H + iy) * p.W + ix) * p.lda + ch * 8)
                   : u32x4{0u, 0u, 0u, 0u};
    }
#pragma unroll
    for (int i = 0; i < W33_DV; ++i) {
      const int v = tid + i * 256, px = v >> 3, ch = v & 7;
      const long long m = (long long)(b * p.H + ty * 4 + (px >> 6)) * p.W + tx * 64 + (px & 63);
      prey[i] = *reinterpret_cast<const u32x4*>(Yg + m * p.ldy + ch * 8);
    }
  };
  auto stash = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < W33_HV; ++i) {
      const int v = tid + i * 256, px = v >> 3, ch = v & 7;
      if (px < W33_HALO) *reinterpret_cast<u32x4*>(Xs + px * W33_PIX + ch * 16) = prex[i];
    }
#pragma unroll
    for (int i = 0; i < W33_DV; ++i) {
      const int v = tid + i * 256, px = v >> 3, ch = v & 7;
      *reinterpret_cast<u32x4*>(Ys + px * W33_PIX + ch * 16) = prey[i];
      float f[8];
      unpack16<bf16_t>(__builtin_bit_cast(uint4, prey[i]), f);     // column sums of dY: channels (tid & 7) * 8 ...
#pragma unroll
      for (int j = 0; j < 8; ++j) bsum[j] += f[j];
    }
  };

  // fragment addressing of the transposing read (wgrad_body.h): lane (g, t) -> pixel (g >> 1) * 8 + t / 4
  // (+4 for the second read), channels (g & 1) * 16 + (t % 4) * 4 of a 32-channel fragment
  const int g = lane >> 4, tt = lane & 15;
  const int lrow = (g >> 1) * 8 + (tt >> 2), lcol = (g & 1) * 16 + (tt & 3) * 4;
  const int chh = wave & 1, tp0 = wave >> 1;
  const char* const yb = Ys + lrow * W33_PIX + (chh * 32 + lcol) * 2;
  const char* const xb = Xs + lrow * W33_PIX + lcol * 2;
  int tapoff[5];
#pragma unroll
  for (int u = 0; u < 5; ++u) {
    const int tap = tp0 + 2 * u, kh = tap / 3, kw = tap - kh * 3;
    tapoff[u] = (kh * 66 + kw) * W33_PIX;
  }
  const int n_units = tp0 == 0 ? 5 : 4;
  f32x16 acc[5][2];
#pragma unroll
  for (int u = 0; u < 5; ++u)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[u][j][r] = 0.f;

  auto read_y = [&](int ks) __attribute__((always_inline)) {
    const char* q = yb + ((ks >> 2) * 64 + (ks & 3) * 16) * W33_PIX;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(SDMI_LDS_V4(q));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(SDMI_LDS_V4(q + 4 * W33_PIX));
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
  };
  auto read_x = [&](int ks, int toff, s16x8 (&fx)[2]) __attribute__((always_inline)) {
    const char* q = xb + ((ks >> 2) * 66 + (ks & 3) * 16) * W33_PIX + toff;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(SDMI_LDS_V4(q + j * 64));
      const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(SDMI_LDS_V4(q + j * 64 + 4 * W33_PIX));
      fx[j] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    }
  };

  int t = split;
  if (t < n_tiles) fetch(t);
  for (; t < n_tiles; t += nsplit) {
    __syncthreads();                       // previous tile's fragment reads are done
    stash();
    __syncthreads();
    if (t + nsplit < n_tiles) fetch(t + nsplit);     // in flight under the MFMAs below
    s16x8 fy = read_y(0), fx[3][2];            // unit u uses buffer u % 3 (five units: two would collide)
    read_x(0, tapoff[0], fx[0]);
    for (int ks = 0; ks < 16; ++ks) {
      s16x8 fy_n = fy;
#pragma unroll
      for (int u = 0; u < 5; ++u) {
        // fragments of the next unit (or of the next k-step's first unit) are fetched under this unit's MFMAs
        if (u + 1 < 5) {
          if (u + 1 < n_units) read_x(ks, tapoff[u + 1], fx[(u + 1) % 3]);
        } else if (ks + 1 < 16) {
          fy_n = read_y(ks + 1);
          read_x(ks + 1, tapoff[0], fx[0]);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (u < n_units) {
#pragma unroll
          for (int j = 0; j < 2; ++j)
            acc[u][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
                __builtin_bit_cast(bf16x8, fy), __builtin_bit_cast(bf16x8, fx[u % 3][j]), acc[u][j], 0, 0, 0);
        }
      }
      fy = fy_n;
    }
  }

  // ---- this workgroup's partials: dW [64][576] and the column sums of dY
  float* const ws = p.workspace + (long long)split * p.N * p.K;
  const int col = lane & 31, row_l = (lane >> 5) * 4;
#pragma unroll
  for (int u = 0; u < 5; ++u) {
    if (u < n_units) {
      const int tap = tp0 + 2 * u;
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int n = chh * 32 + (r & 3) + 8 * (r >> 2) + row_l;
          ws[(long long)n * p.K + tap * 64 + j * 32 + col] = acc[u][j][r];
        }
    }
  }
  __syncthreads();
  float* red = reinterpret_cast<float*>(smem);       // [256][8]
#pragma unroll
  for (int j = 0; j < 8; ++j) red[tid * 8 + j] = bsum[j];
  __syncthreads();
  if (tid < 64) {
    const int c = tid >> 3, j = tid & 7;              // channel tid = chunk c, element j
    float sacc = 0.f;
    for (int r = 0; r < 32; ++r) sacc += red[(r * 8 + c) * 8 + j];
    p.workspace[(long long)nsplit * p.N * p.K + (long long)split * p.N + tid] = sacc;
  }
}

static int launch_wgrad3x3_c64(const SdmiWgradArgs& a, hipStream_t st) {
  SDMI_OPTIN_LDS(wgrad3x3_c64_kernel, W33_SMEM, "wgrad (direct 3x3 c64)");
  hipLaunchKernelGGL(wgrad3x3_c64_kernel, dim3(a.splits), dim3(256), W33_SMEM, st, a);
  return sdmi_check_launch("wgrad (direct 3x3 c64)");
}

// ------------------------------------------------------------------------------------------
// The same direct weight gradient for ANY 3x3 stride-1 layer on a 16 / 32 / 64-column power-of-two image (round 5;
// the twin of igemm_halo.h): grid = (M-split slots, pairs), pair (nb, cb) = 64 output channels x 64 input channels.
// A workgroup walks 256-pixel tiles (TH = 256 / W whole image rows) of its slot: the tile's dY (256 px x its 64
// output channels) and the (TH + 2) x (W + 2) pixel input patch (its 64 input channels, zeros outside the image) are
// staged ONCE and serve all 9 taps -- as an implicit GEMM the activation rows are fetched again for every tap and the
// dY rows for every 128-column block of the filter.  Wave roles, fragment addressing (transposing LDS reads over
// pixel-major operands) and the partial layout [slot][N][9 Cin] are wgrad3x3_c64_kernel's; the usual deterministic fold
// follows.  The dY column sums (bias gradient) are taken by the pairs with cb = 0.
template <int LOGW>
__global__ __launch_bounds__(256) void wgrad3x3_halo_kernel(SdmiWgradArgs p, int ncb) {
  constexpr int W = 1 << LOGW, TH = 256 / W, PW = W + 2, PH = TH + 2, HALO = PH * PW, KPR = W / 16;
  constexpr int HV = (HALO * 8 + 255) / 256, DV = 8;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const Xs = smem;
  char* const Ys = smem + HALO * W33_PIX;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int pair = blockIdx.y, nb = pair / ncb, cb = pair - nb * ncb;
  const bf16_t* __restrict__ Ag = (const bf16_t*)p.a + cb * 64;
  const bf16_t* __restrict__ Yg = (const bf16_t*)p.dy + nb * 64;
  const int tiles_y = p.H / TH;
  const int n_tiles = p.B * tiles_y;
  const int split = blockIdx.x, nsplit = gridDim.x;
  const bool want_bias = cb == 0;

  u32x4 prex[HV], prey[DV];
  float bsum[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) bsum[j] = 0.f;
  auto fetch = [&](int t) __attribute__((always_inline)) {
    const int b = t / tiles_y, ty = t - b * tiles_y;
    const int iy0 = ty * TH - 1;
#pragma unroll
    for (int i = 0; i < HV; ++i) {
      const int v = tid + i * 256, px = v >> 3, ch = v & 7;
      const int hy = px / PW, hx = px - hy * PW;
      const int iy = iy0 + hy, ix = hx - 1;
      const bool ok = px < HALO && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)W;
      prex[i] = ok ? *reinterpret_cast<const u32x4*>(Ag + ((long long)(b * p.H + iy) * W + ix) * p.lda + ch * 8)
                   : u32x4{0u, 0u, 0u, 0u};
    }
#pragma unroll
    for (int i = 0; i < DV; ++i) {
      const int v = tid + i * 256, px = v >> 3, ch = v & 7;
      prey[i] = *reinterpret_cast<const u32x4*>(Yg + ((long long)t * 256 + px) * p.ldy + ch * 8);
    }
  };
  auto stash = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < HV; ++i) {
      const int v = tid + i * 256, px = v >> 3, ch = v & 7;
      if (px < HALO) *reinterpret_cast<u32x4*>(Xs + px * W33_PIX + ch * 16) = prex[i];
    }
#pragma unroll
    for (int i = 0; i < DV; ++i) {
      const int v = tid + i * 256, px = v >> 3, ch = v & 7;
      *reinterpret_cast<u32x4*>(Ys + px * W33_PIX + ch * 16) = prey[i];
      if (want_bias) {
        float f[8];
        unpack16<bf16_t>(__builtin_bit_cast(uint4, prey[i]), f);     // column sums of dY: channels (tid & 7) * 8 ...
#pragma unroll
        for (int j = 0; j < 8; ++j) bsum[j] += f[j];
      }
    }
  };

  const int g = lane >> 4, tt = lane & 15;
  const int lrow = (g >> 1) * 8 + (tt >> 2), lcol = (g & 1) * 16 + (tt & 3) * 4;
  const int chh = wave & 1, tp0 = wave >> 1;
  const char* const yb = Ys + lrow * W33_PIX + (chh * 32 + lcol) * 2;
  const char* const xb = Xs + lrow * W33_PIX + lcol * 2;
  int tapoff[5];
#pragma unroll
  for (int u = 0; u < 5; ++u) {
    const int tap = tp0 + 2 * u, kh = tap / 3, kw = tap - kh * 3;
    tapoff[u] = (kh * PW + kw) * W33_PIX;
  }
  const int n_units = tp0 == 0 ? 5 : 4;
  f32x16 acc[5][2];
#pragma unroll
  for (int u = 0; u < 5; ++u)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[u][j][r] = 0.f;

  auto read_y = [&](int ks) __attribute__((always_inline)) {
    const char* q = yb + ks * 16 * W33_PIX;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(SDMI_LDS_V4(q));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(SDMI_LDS_V4(q + 4 * W33_PIX));
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
  };
  auto read_x = [&](int ks, int toff, s16x8 (&fx)[2]) __attribute__((always_inline)) {
    const char* q = xb + ((ks / KPR) * PW + (ks % KPR) * 16) * W33_PIX + toff;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(SDMI_LDS_V4(q + j * 64));
      const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(SDMI_LDS_V4(q + j * 64 + 4 * W33_PIX));
      fx[j] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    }
  };

  int t = split;
  if (t < n_tiles) fetch(t);
  for (; t < n_tiles; t += nsplit) {
    __syncthreads();                       // previous tile's fragment reads are done
    stash();
    __syncthreads();
    if (t + nsplit < n_tiles) fetch(t + nsplit);     // in flight under the MFMAs below
    s16x8 fy = read_y(0), fx[3][2];            // unit u uses buffer u % 3 (five units: two would collide)
    read_x(0, tapoff[0], fx[0]);
    for (int ks = 0; ks < 16; ++ks) {
      s16x8 fy_n = fy;
#pragma unroll
      for (int u = 0; u < 5; ++u) {
        if (u + 1 < 5) {
          if (u + 1 < n_units) read_x(ks, tapoff[u + 1], fx[(u + 1) % 3]);
        } else if (ks + 1 < 16) {
          fy_n = read_y(ks + 1);
          read_x(ks + 1, tapoff[0], fx[0]);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (u < n_units) {
#pragma unroll
          for (int j = 0; j < 2; ++j)
            acc[u][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
                __builtin_bit_cast(bf16x8, fy), __builtin_bit_cast(bf16x8, fx[u % 3][j]), acc[u][j], 0, 0, 0);
        }
      }
      fy = fy_n;
    }
  }

  // ---- this workgroup's partials: rows [64 nb, +64), columns tap * Cin + [64 cb, +64) of slot `split`'s dW [N][9 Cin]
  float* const ws = p.workspace + (long long)split * p.N * p.K;
  const int col = lane & 31, row_l = (lane >> 5) * 4;
#pragma unroll
  for (int u = 0; u < 5; ++u) {
    if (u < n_units) {
      const int tap = tp0 + 2 * u;
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int n = nb * 64 + chh * 32 + (r & 3) + 8 * (r >> 2) + row_l;
          ws[(long long)n * p.K + tap * p.Cin + cb * 64 + j * 32 + col] = acc[u][j][r];
        }
    }
  }
  if (!want_bias) return;
  __syncthreads();
  float* red = reinterpret_cast<float*>(smem);       // [256][8]
#pragma unroll
  for (int j = 0; j < 8; ++j) red[tid * 8 + j] = bsum[j];
  __syncthreads();
  if (tid < 64) {
    const int c = tid >> 3, j = tid & 7;              // channel tid = chunk c, element j
    float sacc = 0.f;
    for (int r = 0; r < 32; ++r) sacc += red[(r * 8 + c) * 8 + j];
    p.workspace[(long long)nsplit * p.N * p.K + (long long)split * p.N + nb * 64 + tid] = sacc;
  }
}

template <int LOGW>
static int launch_wgrad3x3_halo(const SdmiWgradArgs& a, hipStream_t st) {
  constexpr int W = 1 << LOGW, HALO = (256 / W + 2) * (W + 2);
  constexpr int smem = (HALO + 256) * W33_PIX;
  auto kern = wgrad3x3_halo_kernel<LOGW>;
  SDMI_OPTIN_LDS(kern, smem, "wgrad (direct 3x3, halo)");
  const int ncb = a.Cin / 64, nnb = a.N / 64;
  hipLaunchKernelGGL(kern, dim3(a.splits, nnb * ncb), dim3(256), smem, st, a, ncb);
  return sdmi_check_launch("wgrad (direct 3x3, halo)");
}

static bool wgrad_is1x1(const SdmiWgradArgs& a) {
  return a.KH == 1 && a.KW == 1 && a.stride == 1 && a.pad_t == 0 && a.pad_l == 0 && !a.ups &&
         a.H == a.Ho && a.W == a.Wo;
}

int dispatch_wgrad_f32(const SdmiWgradArgs& a, hipStream_t st) {
  // exact-fp32 MFMA runs at 1/16 of the bf16 rate (0.6 TFLOP/s per CU), so a launch with few workgroups is
  // COMPUTE bound on the handful of CUs it occupies: the Slot Attention / predictor layers (M = clips x slots
  // = 240 ... 448 rows, N, K <= 768) gave 4 - 36 workgroups of 128 x 128 and took 34 - 50 us each.  Below 96
  // workgroups the 64 x 64 tiles (four times the workgroups, a quarter of the work each) are used.
  const long long wg128 = (long long)((a.N + 127) / 128) * ((a.K + 127) / 128) * a.splits;
  const bool small = a.N <= 64 || a.K <= 64 || wg128 < 96;
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
  const bool lin = !is1x1 && !a.ups && a.stride == 1 && a.H == a.Ho && a.W == a.Wo;
  // the scalar-only loaders address a split's rows with 31-bit byte offsets
  const long long mps = ((long long)a.M + a.splits - 1) / a.splits + 64;
  const long long ld = a.lda > a.ldy ? a.lda : a.ldy;
  const bool fits = (mps + (long long)(a.KH + 1) * a.W + 64) * ld * 2 < (1ll << 31);
  // direct kernel for the 64 -> 64 channel 3x3 layers at full resolution (one workgroup per M-split slot)
  {
    if (a.KH == 3 && a.KW == 3 && a.stride == 1 && a.pad_t == 1 && a.pad_l == 1 && !a.ups && a.Cin == 64 &&
        a.N == 64 && a.K == 576 && a.H == a.Ho && a.W == a.Wo && a.W % 64 == 0 && a.H % 4 == 0 && a.splits >= 2 &&
        (long long)a.B * (a.H / 4) * (a.W / 64) >= a.splits && (long long)a.B * a.H * a.W * (a.lda > a.ldy ? a.lda : a.ldy) < (1ll << 40))
      return launch_wgrad3x3_c64(a, st);
  }
  // ... and for every other 3x3 stride-1 layer on 16 / 32 / 64-column power-of-two images (pairs of 64 output x 64
  // input channels; the caller sizes `splits` so that pairs x splits fills the chip: kern.py).  The kernel always
  // writes M-split partials for the fold behind it, so -- like the 64-channel kernel above -- it takes splits >= 2 only:
  // splits == 1 means "written straight into dw" at the C ABI and goes to the implicit-GEMM kernel below.
  {
    static int halo = -1;
    if (halo < 0) {
      const char* e = getenv("SDMI_WGRAD_HALO");
      halo = e ? atoi(e) : 1;
    }
    const int logw = a.W == 16 ? 4 : (a.W == 32 ? 5 : (a.W == 64 ? 6 : 0));
    const long long hw = (long long)a.H * a.W;
    if (halo && logw && a.KH == 3 && a.KW == 3 && a.stride == 1 && a.pad_t == 1 && a.pad_l == 1 && !a.ups && a.H == a.Ho &&
        a.W == a.Wo && a.Cin % 64 == 0 && a.N % 64 == 0 && a.K == 9 * a.Cin && hw % 256 == 0 && (hw & (hw - 1)) == 0 &&
        a.splits >= 2 && a.M / 256 >= a.splits && (a.N / 64) * (a.Cin / 64) * a.splits >= 128 &&
        (long long)a.M * (a.lda > a.ldy ? a.lda : a.ldy) < (1ll << 40))
      return logw == 4 ? launch_wgrad3x3_halo<4>(a, st) : (logw == 5 ? launch_wgrad3x3_halo<5>(a, st) : launch_wgrad3x3_halo<6>(a, st));
  }
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
  if (ps[0].dtype == SDMI_F32) {          // exact-fp32 problems on 64 x 64 tiles (any N, K)
    constexpr int MT = WCfg<float>::MTB / 4;
    for (int i = 0; i < ga->n; ++i) {
      const SdmiWgradArgs& a = ps[i];
      SDMI_REQUIRE(a.a && a.dy && a.dw, "null pointer");
      SDMI_REQUIRE(a.dtype == SDMI_F32 && wgrad_is1x1(a), "grouped wgrad: one dtype per group, 1x1 / linear problems");
      SDMI_REQUIRE(a.K == a.Cin && a.Cin % 4 == 0 && a.lda % 4 == 0 && a.ldy % 4 == 0 && a.M == a.B * a.Ho * a.Wo,
                   "bad geometry");
      SDMI_REQUIRE(a.splits >= 1 && (a.splits == 1 || a.workspace), "splits / workspace");
      g.p[i] = a;
      g.tiles_n[i] = (a.N + 63) / 64;
      g.tiles_k[i] = (a.K + 63) / 64;
      g.per_split[i] = g.tiles_n[i] * g.tiles_k[i] + (a.dbias ? g.tiles_n[i] : 0);
      const int mps = (a.M + a.splits - 1) / a.splits;
      g.mps[i] = (mps + MT - 1) / MT * MT;
      g.item_begin[i] = items;
      items += g.per_split[i] * a.splits;
      g.red_begin[i] = red;
      if (a.splits > 1) {
        long long blocks = wgrad_fold_blocks(a);
        red += (int)(blocks > 512 ? 512 : blocks);
      }
    }
    for (int i = ga->n; i <= WG_MAX; ++i) { g.item_begin[i] = items; g.red_begin[i] = red; }
    constexpr int smem32 = 2 * (64 + 64) * (WCfg<float>::MTB + 16);
    SDMI_OPTIN_LDS(wgrad_group_f32_kernel, smem32, "wgrad group (fp32)");
    hipLaunchKernelGGL(wgrad_group_f32_kernel, dim3(items), dim3(512), smem32, st, g);
    int rc = sdmi_check_launch("wgrad group (fp32)");
    if (rc || red == 0) return rc;
    hipLaunchKernelGGL(wgrad_group_reduce_kernel, dim3(red), dim3(256), 0, st, g);
    return sdmi_check_launch("wgrad group reduce");
  }
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
      long long blocks = wgrad_fold_blocks(a);
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
    long long blocks = wgrad_fold_blocks(ps[i]);
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
  int blocks = (int)wgrad_fold_blocks(*a);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(blocks), dim3(256), 0, st, *a);
  return sdmi_check_launch("wgrad reduce");
}
