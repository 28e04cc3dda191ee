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
