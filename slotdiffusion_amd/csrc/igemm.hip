// Implicit-GEMM convolution / GEMM on the CDNA4 matrix cores (see include/sdmi.h: sdmi_igemm).
//
//   out[m][n] = epi( alpha * sum_k A[m][k] * W[n][k] )      M = B*Ho*Wo, N = Cout, K = KH*KW*Cin
//
// Tiling (per 256-thread workgroup = 4 waves as 2x2):
//   block tile BM x BN (128x128 or 64x64), K tile = BKB bytes of K per row (64 or 128);
//   each wave owns (BM/2)x(BN/2) = TMxTN MFMA 32x32 tiles, fp32 accumulators in VGPR/AGPR.
//   bf16:  v_mfma_f32_32x32x16_bf16  -- one MFMA per 32 bytes of K per (row tile, col tile)
//   fp32:  v_mfma_f32_32x32x2_f32 x4 -- same 32 bytes (8 floats) of K, exact fp32 (fmaf chain)
// Both operand tiles live in LDS as [rows][BKB bytes] with a 16-byte row pad (row pitch 80/144 B:
// odd multiples of 16 B => the 16 distinct rows of every ds_read_b128 lane group hit 16 distinct
// 16-byte bank slots: conflict free).  Lane l reads row (l&31), K bytes [ks*32 + (l>>5)*16, +16).
// Any consistent k permutation is valid for a contraction, so A and B use the same mapping.
//
// Global->LDS staging is register staged (zero fill for image borders / K tail needs predication),
// software pipelined: the global loads of tile t+1 are issued before the MFMAs of tile t and
// written to the other LDS buffer afterwards; one __syncthreads() per K tile.
//
// blockIdx.x -> tile mapping is XCD aware: the 8 XCDs (block b runs on XCD b%8) each get a
// contiguous range of tile ids, with the n-tiles of one m-tile adjacent, so an activation tile is
// fetched into one L2 and re-used by its n-tiles there.
#include "common.h"

namespace {

struct ConvGeom {
  int H, W, Cin, HoWo, Wo, KH, KW, stride, pad_t, pad_l, ups;
};

template <typename T, int BM, int BN, int BKB, bool IS1X1>
__global__ __launch_bounds__(256) void igemm_kernel(SdmiGemmArgs p, int tiles_m, int tiles_n,
                                                    int kt_per_split, int hw_shift) {
  constexpr int VEC = 16 / sizeof(T);
  constexpr int BK = BKB / sizeof(T);
  constexpr int VPR = BKB / 16;
  constexpr int ROWB = BKB + 16;
  constexpr int A_VECS = BM * VPR / 256;
  constexpr int B_VECS = BN * VPR / 256;
  constexpr int WTM = BM / 2, WTN = BN / 2;
  constexpr int TM = WTM / 32, TN = WTN / 32;
  constexpr int KSTEPS = BKB / 32;
  constexpr int BUF_BYTES = (BM + BN) * ROWB;

  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;

  // ---- XCD-aware tile id
  int tile_m, tile_n;
  {
    const int nwg = tiles_m * tiles_n;
    const int bid = blockIdx.x;
    const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
    const int id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    tile_m = id / tiles_n;
    tile_n = id - tile_m * tiles_n;
  }
  const int zb = blockIdx.y / p.split_k;
  const int ksplit = blockIdx.y - zb * p.split_k;
  const int m0 = tile_m * BM, n0 = tile_n * BN;

  const T* __restrict__ Ag = (const T*)p.a + (long long)zb * p.sa;
  const T* __restrict__ Wg = (const T*)p.w + (long long)zb * p.sw;

  const int nk_total = (p.K + BK - 1) / BK;
  const int kt_begin = ksplit * kt_per_split;
  int kt_end = kt_begin + kt_per_split;
  if (kt_end > nk_total) kt_end = nk_total;

  // ---- per-thread staging coordinates
  const int kc = tid % VPR;  // vector column inside the K tile (same for all of a thread's vectors)
  int kk = kt_begin * BK + kc * VEC;  // this thread's k index for the current tile
  int ci = 0, kh = 0, kw = 0;
  if (!IS1X1) {
    const int tap = kk / p.Cin;
    ci = kk - tap * p.Cin;
    kh = tap / p.KW;
    kw = tap - kh * p.KW;
  }
  long long a_pix[A_VECS];   // IS1X1: m*lda ; conv: b*H*W (pixel index base)
  int a_iy0[A_VECS], a_ix0[A_VECS];
  bool a_ok[A_VECS];
#pragma unroll
  for (int i = 0; i < A_VECS; ++i) {
    const int row = (tid + i * 256) / VPR;
    const int m = m0 + row;
    a_ok[i] = m < p.M;
    if (IS1X1) {
      a_pix[i] = (long long)m * p.lda;
      a_iy0[i] = a_ix0[i] = 0;
    } else {
      const int HoWo = p.Ho * p.Wo;
      const int b = m / HoWo;
      const int rem = m - b * HoWo;
      const int oy = rem / p.Wo;
      const int ox = rem - oy * p.Wo;
      a_pix[i] = (long long)b * p.H * p.W;
      a_iy0[i] = oy * p.stride - p.pad_t;
      a_ix0[i] = ox * p.stride - p.pad_l;
    }
  }
  long long b_off[B_VECS];
  bool b_ok[B_VECS];
#pragma unroll
  for (int i = 0; i < B_VECS; ++i) {
    const int row = (tid + i * 256) / VPR;
    const int n = n0 + row;
    b_ok[i] = n < p.N;
    b_off[i] = (long long)n * p.ldw;
  }

  u32x4 ra[A_VECS], rb[B_VECS];
  const u32x4 zero4 = {0u, 0u, 0u, 0u};

  // Loads are issued unconditionally from a clamped (always valid) address so the compiler can
  // keep all of a tile's global loads in flight together; out-of-image / K-tail vectors are
  // zeroed when the registers are written to LDS (mask bits travel with the tile).
  unsigned okmask = 0;
  auto load_tile = [&]() __attribute__((always_inline)) {
    const bool k_ok = kk < p.K;
    okmask = 0;
#pragma unroll
    for (int i = 0; i < A_VECS; ++i) {
      bool ok = a_ok[i] && k_ok;
      long long off;
      if (IS1X1) {
        off = a_pix[i] + kk;
      } else {
        int iy = a_iy0[i] + kh, ix = a_ix0[i] + kw;
        if (p.ups) {
          ok = ok && iy >= 0 && iy < 2 * p.H && ix >= 0 && ix < 2 * p.W;
          iy >>= 1;
          ix >>= 1;
        } else if (p.zins > 1) {
          ok = ok && iy >= 0 && ix >= 0 && (iy % p.zins) == 0 && (ix % p.zins) == 0;
          iy /= p.zins;
          ix /= p.zins;
          ok = ok && iy < p.H && ix < p.W;
        } else {
          ok = ok && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
        }
        off = (a_pix[i] + (long long)iy * p.W + ix) * p.lda + ci;
      }
      off = ok ? off : 0;
      okmask |= (ok ? 1u : 0u) << i;
      ra[i] = *reinterpret_cast<const u32x4*>(Ag + off);
    }
#pragma unroll
    for (int i = 0; i < B_VECS; ++i) {
      const bool ok = b_ok[i] && k_ok;
      okmask |= (ok ? 1u : 0u) << (A_VECS + i);
      rb[i] = *reinterpret_cast<const u32x4*>(Wg + (ok ? b_off[i] + kk : 0));
    }
    // advance k state to the next tile
    kk += BK;
    if (!IS1X1) {
      ci += BK;
      while (ci >= p.Cin) {
        ci -= p.Cin;
        if (++kw == p.KW) { kw = 0; ++kh; }
      }
    }
  };
  auto store_tile = [&](int buf) __attribute__((always_inline)) {
    char* base = smem + buf * BUF_BYTES;
#pragma unroll
    for (int i = 0; i < A_VECS; ++i) {
      const int row = (tid + i * 256) / VPR;
      *reinterpret_cast<u32x4*>(base + row * ROWB + kc * 16) =
          ((okmask >> i) & 1u) ? ra[i] : zero4;
    }
#pragma unroll
    for (int i = 0; i < B_VECS; ++i) {
      const int row = (tid + i * 256) / VPR;
      *reinterpret_cast<u32x4*>(base + BM * ROWB + row * ROWB + kc * 16) =
          ((okmask >> (A_VECS + i)) & 1u) ? rb[i] : zero4;
    }
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int frag_row = lane & 31;
  const int frag_kb = (lane >> 5) * 16;

  if (kt_begin < kt_end) {
    load_tile();
    store_tile(0);
    __syncthreads();
    int buf = 0;
    for (int kt = kt_begin; kt < kt_end; ++kt) {
      const bool more = kt + 1 < kt_end;
      if (more) load_tile();
      const char* As = smem + buf * BUF_BYTES + (wm * WTM + frag_row) * ROWB + frag_kb;
      const char* Bs = smem + buf * BUF_BYTES + BM * ROWB + (wn * WTN + frag_row) * ROWB + frag_kb;
#pragma unroll
      for (int ks = 0; ks < KSTEPS; ++ks) {
        u32x4 fa[TM], fb[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i)
          fa[i] = *reinterpret_cast<const u32x4*>(As + i * 32 * ROWB + ks * 32);
#pragma unroll
        for (int j = 0; j < TN; ++j)
          fb[j] = *reinterpret_cast<const u32x4*>(Bs + j * 32 * ROWB + ks * 32);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j) {
            if constexpr (sizeof(T) == 2) {
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
                  __builtin_bit_cast(bf16x8, fa[i]), __builtin_bit_cast(bf16x8, fb[j]), acc[i][j],
                  0, 0, 0);
            } else {
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(
                  __uint_as_float(fa[i][0]), __uint_as_float(fb[j][0]), acc[i][j], 0, 0, 0);
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(
                  __uint_as_float(fa[i][1]), __uint_as_float(fb[j][1]), acc[i][j], 0, 0, 0);
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(
                  __uint_as_float(fa[i][2]), __uint_as_float(fb[j][2]), acc[i][j], 0, 0, 0);
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(
                  __uint_as_float(fa[i][3]), __uint_as_float(fb[j][3]), acc[i][j], 0, 0, 0);
            }
          }
      }
      if (more) store_tile(buf ^ 1);
      __syncthreads();
      buf ^= 1;
    }
  }

  // ---- epilogue. C/D layout of the 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
  const int col_l = lane & 31;
  const int row_l = (lane >> 5) * 4;
  if (p.split_k > 1) {
    float* ws = p.workspace + ((long long)blockIdx.y) * p.M * p.N;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int n = n0 + wn * WTN + j * 32 + col_l;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m = m0 + wm * WTM + i * 32 + (r & 3) + 8 * (r >> 2) + row_l;
          if (m < p.M && n < p.N) ws[(long long)m * p.N + n] = acc[i][j][r];
        }
      }
    return;
  }
  char* outp = (char*)p.out;
  const char* resp = (const char*)p.residual;
  const long long zc = (long long)zb * p.sc, zr = (long long)zb * p.sr;
  const int HoWo = p.Ho * p.Wo;
  const bool out_bf16 = p.out_dtype == SDMI_BF16;
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int n = n0 + wn * WTN + j * 32 + col_l;
      const bool n_ok = n < p.N;
      const float bn = (p.bias && !p.bias_m && n_ok) ? p.bias[n] : 0.f;
      const int mbase = m0 + wm * WTM + i * 32 + row_l;
      // phase 1: gather every epilogue operand of this 32x32 tile.  Indices are clamped into
      // range so all loads are unconditional and in flight together (`out` may alias
      // `residual`, so no load may be interleaved with the stores of phase 2).
      const int nc = n_ok ? n : p.N - 1;
      float add[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) add[r] = bn;
      if (p.bias && p.bias_m) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m = min(mbase + (r & 3) + 8 * (r >> 2), p.M - 1);
          add[r] += p.bias[m];
        }
      }
      if (p.rowvec) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m = min(mbase + (r & 3) + 8 * (r >> 2), p.M - 1);
          const int b = hw_shift >= 0 ? (m >> hw_shift) : (m / HoWo);
          add[r] += p.rowvec[(long long)b * p.ldrv + nc];
        }
      }
      if (resp) {
        if (out_bf16) {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int m = min(mbase + (r & 3) + 8 * (r >> 2), p.M - 1);
            add[r] += bf16_to_f32(((const bf16_t*)resp)[zr + (long long)m * p.ldr + nc]);
          }
        } else {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int m = min(mbase + (r & 3) + 8 * (r >> 2), p.M - 1);
            add[r] += ((const float*)resp)[zr + (long long)m * p.ldr + nc];
          }
        }
      }
      // phase 2: finish and store (uniform switches hoisted out of the element loops)
      float v[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) v[r] = acc[i][j][r] * p.alpha + add[r];
      if (p.act == SDMI_ACT_SILU) {
#pragma unroll
        for (int r = 0; r < 16; ++r) v[r] = act_apply(v[r], SDMI_ACT_SILU);
      } else if (p.act == SDMI_ACT_RELU) {
#pragma unroll
        for (int r = 0; r < 16; ++r) v[r] = fmaxf(v[r], 0.f);
      } else if (p.act == SDMI_ACT_GELU) {
#pragma unroll
        for (int r = 0; r < 16; ++r) v[r] = act_apply(v[r], SDMI_ACT_GELU);
      }
      if (out_bf16) {
        bf16_t* o = (bf16_t*)outp + zc + n;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m = mbase + (r & 3) + 8 * (r >> 2);
          if (n_ok && m < p.M) o[(long long)m * p.ldc] = f32_to_bf16(v[r]);
        }
      } else {
        float* o = (float*)outp + zc + n;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m = mbase + (r & 3) + 8 * (r >> 2);
          if (n_ok && m < p.M) o[(long long)m * p.ldc] = v[r];
        }
      }
    }
}

// split-K second stage: sum partials, apply the same epilogue
__global__ __launch_bounds__(256) void splitk_epilogue_kernel(SdmiGemmArgs p, int hw_shift) {
  const long long total = (long long)p.M * p.N;
  const int HoWo = p.Ho * p.Wo;
  for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total;
       idx += (long long)gridDim.x * 256) {
    const int m = (int)(idx / p.N);
    const int n = (int)(idx - (long long)m * p.N);
    float s = 0.f;
    for (int k = 0; k < p.split_k; ++k) s += p.workspace[(long long)k * total + idx];
    float v = s * p.alpha;
    if (p.bias) v += p.bias_m ? p.bias[m] : p.bias[n];
    if (p.rowvec) {
      const int b = hw_shift >= 0 ? (m >> hw_shift) : (m / HoWo);
      v += p.rowvec[(long long)b * p.ldrv + n];
    }
    if (p.residual) {
      const long long ro = (long long)m * p.ldr + n;
      v += p.out_dtype == SDMI_BF16 ? bf16_to_f32(((const bf16_t*)p.residual)[ro])
                                    : ((const float*)p.residual)[ro];
    }
    v = act_apply(v, p.act);
    const long long oo = (long long)m * p.ldc + n;
    if (p.out_dtype == SDMI_BF16) ((bf16_t*)p.out)[oo] = f32_to_bf16(v);
    else ((float*)p.out)[oo] = v;
  }
}

template <typename T, int BM, int BN, int BKB, bool IS1X1>
int launch_cfg(const SdmiGemmArgs& p, int split_k, int hw_shift, hipStream_t st) {
  constexpr int BK = BKB / sizeof(T);
  constexpr int smem = 2 * (BM + BN) * (BKB + 16);
  static bool attr_done = false;
  auto kern = igemm_kernel<T, BM, BN, BKB, IS1X1>;
  if (!attr_done) {
    if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, smem) !=
        hipSuccess) {
      sdmi_set_error("igemm: hipFuncSetAttribute failed");
      return SDMI_ELAUNCH;
    }
    attr_done = true;
  }
  SdmiGemmArgs q = p;
  q.split_k = split_k;
  const int tiles_m = (p.M + BM - 1) / BM, tiles_n = (p.N + BN - 1) / BN;
  const int nk = (p.K + BK - 1) / BK;
  const int ktps = (nk + split_k - 1) / split_k;
  dim3 grid(tiles_m * tiles_n, split_k * (p.batch > 0 ? p.batch : 1));
  hipLaunchKernelGGL(kern, grid, dim3(256), smem, st, q, tiles_m, tiles_n, ktps, hw_shift);
  int rc = sdmi_check_launch("igemm");
  if (rc) return rc;
  if (split_k > 1) {
    const long long total = (long long)p.M * p.N;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(splitk_epilogue_kernel, dim3(blocks), dim3(256), 0, st, q, hw_shift);
    rc = sdmi_check_launch("igemm splitk epilogue");
  }
  return rc;
}

template <typename T>
int dispatch(const SdmiGemmArgs& p, hipStream_t st) {
  constexpr int VEC = 16 / sizeof(T);
  const bool is1x1 = p.KH == 1 && p.KW == 1 && p.stride == 1 && p.pad_t == 0 && p.pad_l == 0 &&
                     !p.ups && p.zins <= 1;
  int hw_shift = -1;
  {
    const int hw = p.Ho * p.Wo;
    if (hw > 0 && (hw & (hw - 1)) == 0) {
      hw_shift = 0;
      while ((1 << hw_shift) < hw) ++hw_shift;
    }
  }
  const int batch = p.batch > 0 ? p.batch : 1;
  const long long t128 = (long long)((p.M + 127) / 128) * ((p.N + 127) / 128) * batch;
  const bool big = p.N > 64 && t128 >= 192;
  // K tile: 128 bytes of K per row when K is deep enough, else 64
  const int kbytes = p.K * (int)sizeof(T);
  const bool wide = kbytes >= 512;
  int split_k = 1;
  if (p.split_k > 0) split_k = p.split_k;       // caller override
  else if (!big && batch == 1 && p.workspace) {
    const long long t64 = (long long)((p.M + 63) / 64) * ((p.N + 63) / 64);
    const int nk = (kbytes + (wide ? 127 : 63)) / (wide ? 128 : 64);
    while (t64 * split_k < 384 && split_k * 2 <= nk / 4 && split_k < 16) split_k *= 2;
  }
  if (split_k > 1 && !p.workspace) split_k = 1;
  (void)VEC;
#define SDMI_GO(BM, BN, BKB)                                                           \
  return is1x1 ? launch_cfg<T, BM, BN, BKB, true>(p, split_k, hw_shift, st)            \
               : launch_cfg<T, BM, BN, BKB, false>(p, split_k, hw_shift, st)
  if (big) {
    if (wide) { SDMI_GO(128, 128, 128); } else { SDMI_GO(128, 128, 64); }
  } else {
    if (wide) { SDMI_GO(64, 64, 128); } else { SDMI_GO(64, 64, 64); }
  }
#undef SDMI_GO
}

}  // namespace

extern "C" int sdmi_igemm(const SdmiGemmArgs* a, void* stream) {
  SDMI_REQUIRE(a && a->a && a->w && a->out, "null pointer");
  SDMI_REQUIRE(a->dtype == SDMI_F32 || a->dtype == SDMI_BF16, "bad dtype");
  SDMI_REQUIRE(a->out_dtype == SDMI_F32 || a->out_dtype == SDMI_BF16, "bad out_dtype");
  const int vec = a->dtype == SDMI_BF16 ? 8 : 4;
  SDMI_REQUIRE(a->M > 0 && a->N > 0 && a->K > 0, "empty problem");
  SDMI_REQUIRE(a->K == a->KH * a->KW * a->Cin, "K != KH*KW*Cin");
  SDMI_REQUIRE(a->Cin % vec == 0 && a->lda % vec == 0 && a->ldw % vec == 0,
               "Cin/lda/ldw must be multiples of the 16-byte vector width");
  SDMI_REQUIRE(((uintptr_t)a->a & 15) == 0 && ((uintptr_t)a->w & 15) == 0, "unaligned operand");
  SDMI_REQUIRE(a->M == a->B * a->Ho * a->Wo, "M != B*Ho*Wo");
  SDMI_REQUIRE(!(a->zins > 1 && (a->ups || a->stride != 1)), "zins excludes ups / stride");
  SDMI_REQUIRE(!(a->batch > 1) || (a->KH == 1 && a->KW == 1), "batched mode is 1x1 only");
  SDMI_REQUIRE(!(a->batch > 1 && a->split_k > 1), "batched split-K unsupported");
  SDMI_REQUIRE(a->sa % vec == 0 && a->sw % vec == 0, "batch strides must keep 16-byte alignment");
  hipStream_t st = (hipStream_t)stream;
  return a->dtype == SDMI_BF16 ? dispatch<bf16_t>(*a, st) : dispatch<float>(*a, st);
}
