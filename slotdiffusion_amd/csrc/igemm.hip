// Implicit-GEMM convolution / GEMM on the CDNA4 matrix cores (see include/sdmi.h: sdmi_igemm).
//
//   out[m][n] = epi( alpha * sum_k A[m][k] * W[n][k] )      M = B*Ho*Wo, N = Cout, K = KH*KW*Cin
//
// Tiling (per 512-thread workgroup = 4 MFMA waves as 2x2 + 4 loader waves):
//   block tile BM x BN (128x128, 128x64 or 64x64), K tile = BKB bytes of K per row (64 or
//   128); each MFMA wave owns (BM/2)x(BN/2) = TMxTN MFMA 32x32 tiles, fp32 accumulators.
//   bf16:  v_mfma_f32_32x32x16_bf16  -- one MFMA per 32 bytes of K per (row tile, col tile)
//   fp32:  v_mfma_f32_32x32x2_f32 x4 -- same 32 bytes (8 floats) of K, exact fp32 (fmaf chain)
// Both operand tiles live in LDS as [rows][BKB bytes] with a 16-byte row pad (row pitch 80/144 B:
// odd multiples of 16 B => the 16 distinct rows of every ds_read_b128 lane group hit 16 distinct
// 16-byte bank slots: conflict free).  Lane l reads row (l&31), K bytes [ks*32 + (l>>5)*16, +16).
// Any consistent k permutation is valid for a contraction, so A and B use the same mapping.
//
// Wave specialisation: the loader waves (one per SIMD, next to one MFMA wave) do the im2col
// address arithmetic, keep two K tiles of global loads in flight in registers (zero fill for
// image borders / K tail needs predication, so the copy is register staged) and fill a
// double-buffered LDS image one K tile ahead; the MFMA waves only read fragments and issue
// MFMAs, prefetching the next k-step's fragments under the current MFMAs.  One barrier per K tile.
//
// blockIdx.x -> tile mapping is XCD aware: the 8 XCDs (block b runs on XCD b%8) each get a
// contiguous range of tile ids, with the n-tiles of one m-tile adjacent, so an activation tile is
// fetched into one L2 and re-used by its n-tiles there.
#include "common.h"
#include <stdlib.h>

#ifndef SDMI_IGEMM_DMA
#define SDMI_IGEMM_DMA 3      // 0 off, 1 / 2 experiments (all eligible shapes), 3 the measured-faster shapes only
#endif

#include "igemm_body.h"
#include "epi_rows.h"
#include "igemm_sym.h"

// 3x3 stride-1 convolution with the activation patch staged once per 64-channel chunk (igemm_halo.h)
int sdmi_launch_halo(const SdmiGemmArgs& p, int logw, int nj, int hw_shift, hipStream_t st, int n_cu);

namespace {

// sdmi.h parity4: batch index z = 2 py + px of the four parity convolutions of an upsample layer selects padding and
// output placement (the filter advances by z * sw like any batch); wave-uniform scalar arithmetic on the by-value args
__device__ __forceinline__ void parity_select(SdmiGemmArgs& p, int by) {
  if (p.parity4) {
    const int z = by / (p.split_k > 0 ? p.split_k : 1);
    p.pad_t = 1 - (z >> 1);
    p.pad_l = 1 - (z & 1);
    p.ooy = z >> 1;
    p.oox = z & 1;
  }
}


// ------------------------------------------------------------------------------------------
// LDS-DMA variant (MODE 1 = 1x1 / linear, MODE 2 = plain convolution with Cin % BK == 0; K tile =
// 128 bytes of K per row).  Measured on MI355X (tools/probes/ldsdma.hip): VALU instructions of a
// loader wave and the MFMAs of the wave next to it on the SIMD serialise -- a loader that computes
// 64-bit addresses per K tile makes fill time and MFMA time ADD UP instead of overlapping.  So the
// loaders here are scalar-only in the steady state:
//   * operands are fetched with `buffer_load_dwordx4 ... lds` (global -> LDS, no VGPR staging, no
//     ds_write): the per-lane byte offset (voffset) is computed once per output tile, the walk over
//     K is a wave-uniform SGPR offset (soffset);
//   * image borders / the K tail are out-of-range voffsets (the buffer returns zeros); a filter
//     tap's validity mask is applied once per tap, not per K tile;
//   * the activation base pointer is biased by -(pad_t*W + pad_l) pixels so every offset is >= 0.
// One DMA instruction writes 64 lanes x 16 B = 1 KB of LDS contiguously = 8 rows of 128 B, so the
// tile rows are unpadded and bank conflicts are avoided by an XOR swizzle instead: logical 16-byte
// chunk c of row r lives at chunk c ^ ((r >> 1) & 7); each lane simply FETCHES the chunk that
// belongs at its position.  The 16 rows of every ds_read_b128 lane group then cover all 64 banks.
// Prefetch depth comes from NSTAGE LDS stages (NSTAGE-1 K tiles in flight), one s_barrier per K
// tile: before barrier g the loaders have waited for K tile g to land; after it they refill the
// stage that K tile g-1 just vacated.
// MFMA waves: always FOUR (2 x 2 over the tile, one per SIMD), each owning (BM/2) x (BN/2): at
// 256 x 128 a wave computes 128 x 64 = 4 x 2 MFMA tiles and reads 6 fragments per 8 MFMAs -- 96 B/clk
// of LDS reads per CU at full matrix rate, where 64 x 64 wave tiles (4 fragments per 4 MFMAs) need
// the LDS's whole 256 B/clk (the limit the 128 x 128 kernels run into).
// LW = loader waves (4 or 8).  PMC + issue-cost arithmetic (round 3): a wave issues one 1 KB LDS-DMA
// piece per ~150 cycles next to MFMA traffic, so FOUR loader waves deliver ~27 B/clk per CU -- the 25 B/clk the
// 128 x 128 kernel was observed at (a K tile every ~1300 cycles against 512 cycles of MFMA).  Eight loader
// waves halve the issue time per K tile; the workgroup is then 12 waves (one per CU, three per SIMD).
template <typename T, int BM, int BN, int NSTAGE, int MODE, int LW = 4, bool XS = false, int EPI = 0>
__global__ __launch_bounds__(256 + 64 * LW, (BM * BN <= 64 * 64 ? 4 : 0)) void igemm_dma_kernel(
    SdmiGemmArgs p, int tiles_m, int tiles_n, int kt_per_split, int hw_shift) {
  if constexpr (MODE == 2 && EPI == 0 && !XS) parity_select(p, (int)blockIdx.y);
  constexpr int VEC = 16 / sizeof(T);
  constexpr int BK = 128 / sizeof(T);
  constexpr int WM = BM / 2, WN = BN / 2;            // wave tile
  constexpr int TMf = WM / 32, TNf = WN / 32;
  constexpr int NMFMA = 256;                         // MFMA threads; 256 loader threads follow
  constexpr int STAGE = (BM + BN) * 128;
  constexpr int A_PC = BM / (8 * LW), B_PC = BN / (8 * LW);      // 1 KB pieces per loader wave per K tile
  constexpr int NLOAD = A_PC + B_PC;
  constexpr unsigned OOB = 0x80000000u;              // == num_records: always out of range
  static_assert((NSTAGE - 2) * NLOAD <= 63, "vmcnt field");
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int nwg = tiles_m * tiles_n;
  auto tile_of = [&](int vb, int& m0, int& n0) __attribute__((always_inline)) {
    const int xcd = vb & 7, q = nwg >> 3, r = nwg & 7;
    const int id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (vb >> 3);
    const int tm = id / tiles_n;
    m0 = tm * BM;
    n0 = (id - tm * tiles_n) * BN;
  };
  const int my_tiles = ((int)blockIdx.x < nwg) ? (nwg - 1 - (int)blockIdx.x) / (int)gridDim.x + 1 : 0;
  const int zb = blockIdx.y / p.split_k;
  const int ksplit = blockIdx.y - zb * p.split_k;
  const int nk_total = (p.K + BK - 1) / BK;
  const int kt_begin = ksplit * kt_per_split;
  int kt_end = kt_begin + kt_per_split;
  if (kt_end > nk_total) kt_end = nk_total;
  const int n_kt = kt_end > kt_begin ? kt_end - kt_begin : 0;
  const int total = my_tiles * n_kt;

  if (threadIdx.x >= NMFMA) {
    // =============================== loader waves ===============================
    // (a split-K slice without K tiles -- 16 splits of 129 K tiles leave the last one empty -- still writes its
    // zero partial: only the loaders leave)
    if (total == 0) return;
    const int lt = threadIdx.x - NMFMA, l = lt & 63;
    const int lw = __builtin_amdgcn_readfirstlane(lt >> 6);      // scalar: LDS piece addresses stay in SGPRs
    const int kc = (l & 7) ^ ((4 * (lw & 1) + (l >> 4)) & 7);   // logical chunk fetched by this lane
    const T* Ag = (const T*)p.a + (long long)zb * p.sa;
    if (MODE == 2) Ag -= (long long)(p.pad_t * p.W + p.pad_l) * p.lda;
    const T* Wg = (const T*)p.w + (long long)zb * p.sw;
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)Ag, 0, (int)OOB, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc((void*)Wg, 0, (int)OOB, 0x00020000);
    unsigned a_vo[A_PC], a_cur[A_PC], a_inv[A_PC], b_vo[B_PC], b_cur[B_PC];
    // extra A sources appended along K (sdmi.h: a2 / a3): 1x1 taps at the output pixel
    unsigned a_vo2[XS ? A_PC : 1], a_vo3[XS ? A_PC : 1];
    int ld_tile = 0, ld_kt = 0, k0 = 0, ci = 0, kh = 0, kw = 0;   // wave-uniform
    auto begin_tile = [&]() __attribute__((always_inline)) {
      int m0, n0;
      tile_of((int)blockIdx.x + ld_tile * (int)gridDim.x, m0, n0);
      k0 = kt_begin * BK;
      if (MODE == 2) {
        const int tap = k0 / p.Cin;
        ci = k0 - tap * p.Cin;
        kh = tap / p.KW;
        kw = tap - kh * p.KW;
      }
#pragma unroll
      for (int i = 0; i < A_PC; ++i) {
        const int row = (lw + LW * i) * 8 + (l >> 3);
        const int m = min(m0 + row, p.M - 1);
        if constexpr (XS) {        // (stride-1 "same" convolutions: output pixel m = input pixel m)
          a_vo2[i] = ((unsigned)m * (unsigned)p.lda2 + kc * VEC) * (unsigned)sizeof(T);
          a_vo3[i] = ((unsigned)m * (unsigned)p.lda3 + kc * VEC) * (unsigned)sizeof(T);
        }
        if (MODE == 1) {
          a_vo[i] = ((unsigned)m * (unsigned)p.lda + kc * VEC) * (unsigned)sizeof(T);
          a_inv[i] = 0;
        } else {
          const int HoWo = p.Ho * p.Wo;
          const int b = hw_shift >= 0 ? (m >> hw_shift) : (m / HoWo);
          const int rem = m - b * HoWo;
          const int oy = rem / p.Wo;
          const int ox = rem - oy * p.Wo;
          const int iy0 = oy * p.stride - p.pad_t, ix0 = ox * p.stride - p.pad_l;
          a_vo[i] = ((unsigned)((b * p.H + oy * p.stride) * p.W + ox * p.stride) * (unsigned)p.lda +
                     kc * VEC) * (unsigned)sizeof(T);
          unsigned rb = 0, cb = 0, inv = 0;   // bad rows / columns of the filter window
          for (int q = 0; q < p.KH; ++q) rb |= ((unsigned)(iy0 + q) < (unsigned)p.H ? 0u : 1u) << q;
          for (int q = 0; q < p.KW; ++q) cb |= ((unsigned)(ix0 + q) < (unsigned)p.W ? 0u : 1u) << q;
          for (int q = 0; q < p.KH; ++q) inv |= (((rb >> q) & 1u) ? ((1u << p.KW) - 1u) : cb) << (q * p.KW);
          a_inv[i] = inv;
        }
        a_cur[i] = a_vo[i];
      }
#pragma unroll
      for (int i = 0; i < B_PC; ++i) {
        const int row = (lw + LW * i) * 8 + (l >> 3);
        const int n = min(n0 + row, p.N - 1);
        b_vo[i] = ((unsigned)n * (unsigned)p.ldw + kc * VEC) * (unsigned)sizeof(T);
        b_cur[i] = b_vo[i];
      }
    };
    int ld_stage = 0;
    auto issue = [&]() __attribute__((always_inline)) {
      if (ld_kt == 0) begin_tile();
      unsigned so_a;
      if (MODE == 2) {
        if (ci == 0 || ld_kt == 0) {               // new filter tap: apply its validity mask
          const int tap = kh * p.KW + kw;
#pragma unroll
          for (int i = 0; i < A_PC; ++i) a_cur[i] = ((a_inv[i] >> tap) & 1u) ? OOB : a_vo[i];
        }
        so_a = (unsigned)((kh * p.W + kw) * p.lda + ci) * (unsigned)sizeof(T);
      } else {
        if (k0 + BK > p.K) {                       // K tail (last K tile of a tile only)
          const bool k_ok = k0 + kc * VEC < p.K;
#pragma unroll
          for (int i = 0; i < A_PC; ++i) a_cur[i] = k_ok ? a_vo[i] : OOB;
#pragma unroll
          for (int i = 0; i < B_PC; ++i) b_cur[i] = k_ok ? b_vo[i] : OOB;
        }
        so_a = (unsigned)k0 * (unsigned)sizeof(T);
      }
      const unsigned so_b = (unsigned)k0 * (unsigned)sizeof(T);
      char* st = smem + ld_stage * STAGE + lw * 1024;
      if constexpr (XS) {
        // which source this K tile reads (wave-uniform selects; the descriptor is rebuilt per K tile)
        const bool s1 = p.a2 != nullptr && k0 >= p.K1;
        const bool s2 = s1 && p.a3 != nullptr && k0 >= p.K2;
        const unsigned so_x = s2 ? (unsigned)(k0 - p.K2) * (unsigned)sizeof(T)
                                 : (s1 ? (unsigned)(k0 - p.K1) * (unsigned)sizeof(T) : so_a);
        const void* base_x = s2 ? p.a3 : (s1 ? p.a2 : (const void*)Ag);
        const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc((void*)base_x, 0, (int)OOB, 0x00020000);
#pragma unroll
        for (int i = 0; i < A_PC; ++i) {
          const unsigned vo = s2 ? a_vo3[i] : (s1 ? a_vo2[i] : a_cur[i]);
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x, (lds_void*)(st + i * (LW * 1024)), 16, (int)vo, (int)so_x, 0, 0);
        }
      } else {
#pragma unroll
      for (int i = 0; i < A_PC; ++i)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lds_void*)(st + i * (LW * 1024)), 16, (int)a_cur[i],
                                                 (int)so_a, 0, 0);
      }
#pragma unroll
      for (int i = 0; i < B_PC; ++i)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, (lds_void*)(st + BM * 128 + i * (LW * 1024)), 16,
                                                 (int)b_cur[i], (int)so_b, 0, 0);
      if (MODE == 2) {
        ci += BK;
        if (ci == p.Cin) {
          ci = 0;
          if (++kw == p.KW) { kw = 0; ++kh; }
        }
      }
      k0 += BK;
      if (++ld_kt == n_kt) { ld_kt = 0; ++ld_tile; }
      if (++ld_stage == NSTAGE) ld_stage = 0;
    };
    // Steps past the last one re-fetch clamped rows of a non-existent tile into a stage nobody
    // reads any more: the number of DMA groups in flight stays static, so a fixed vmcnt works.
#pragma unroll
    for (int s = 0; s < NSTAGE - 1; ++s) issue();
    for (int g = 0; g < total; ++g) {
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NSTAGE - 2) * NLOAD) : "memory");   // K tile g landed
      __builtin_amdgcn_s_barrier();
      issue();                                     // K tile g + NSTAGE - 1 -> stage of K tile g - 1
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // no DMA may outlive the workgroup's LDS
    return;
  }

  // ================================= MFMA waves =================================
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int R = lane & 31;
  int swz[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) swz[ks] = ((2 * ks + (lane >> 5)) ^ ((R >> 1) & 7)) * 16;
  const int a_off = (wm * WM + R) * 128;
  const int b_off = BM * 128 + (wn * WN + R) * 128;
  auto read_frags = [&](const char* base, int ks, u32x4 (&fa)[TMf], u32x4 (&fb)[TNf])
                        __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < TMf; ++i)
      fa[i] = *reinterpret_cast<const u32x4*>(base + a_off + i * 4096 + swz[ks]);
#pragma unroll
    for (int j = 0; j < TNf; ++j)
      fb[j] = *reinterpret_cast<const u32x4*>(base + b_off + j * 4096 + swz[ks]);
  };
  int stage = 0;
  for (int ti = 0; ti < my_tiles; ++ti) {
    int m0, n0;
    tile_of((int)blockIdx.x + ti * (int)gridDim.x, m0, n0);
    f32x16 acc[TMf][TNf];
#pragma unroll
    for (int i = 0; i < TMf; ++i)
#pragma unroll
      for (int j = 0; j < TNf; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    float sx[TMf], sxx[TMf];                      // LayerNorm fold (EPI bit 0): row sums from the A fragments
#pragma unroll
    for (int i = 0; i < TMf; ++i) sx[i] = sxx[i] = 0.f;
    for (int t = 0; t < n_kt; ++t) {
      __builtin_amdgcn_s_barrier();               // K tile landed; previous stage may be refilled
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
      const char* base = smem + stage * STAGE;
      if (++stage == NSTAGE) stage = 0;
      u32x4 fa[2][TMf], fb[2][TNf];
      read_frags(base, 0, fa[0], fb[0]);
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        if (ks + 1 < 4) read_frags(base, ks + 1, fa[(ks + 1) & 1], fb[(ks + 1) & 1]);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr ((EPI & 1) != 0 && sizeof(T) == 2) {
#pragma unroll
          for (int i = 0; i < TMf; ++i) {
            const sdmi_bf16x2 ones = __builtin_bit_cast(sdmi_bf16x2, 0x3F803F80u);
            const bf16x8 a8 = __builtin_bit_cast(bf16x8, fa[ks & 1][i]);
            const sdmi_bf16x2 v0 = {a8[0], a8[1]}, v1 = {a8[2], a8[3]}, v2 = {a8[4], a8[5]}, v3 = {a8[6], a8[7]};
            sx[i] = __builtin_amdgcn_fdot2_f32_bf16(v0, ones, sx[i], false);
            sxx[i] = __builtin_amdgcn_fdot2_f32_bf16(v0, v0, sxx[i], false);
            sx[i] = __builtin_amdgcn_fdot2_f32_bf16(v1, ones, sx[i], false);
            sxx[i] = __builtin_amdgcn_fdot2_f32_bf16(v1, v1, sxx[i], false);
            sx[i] = __builtin_amdgcn_fdot2_f32_bf16(v2, ones, sx[i], false);
            sxx[i] = __builtin_amdgcn_fdot2_f32_bf16(v2, v2, sxx[i], false);
            sx[i] = __builtin_amdgcn_fdot2_f32_bf16(v3, ones, sx[i], false);
            sxx[i] = __builtin_amdgcn_fdot2_f32_bf16(v3, v3, sxx[i], false);
          }
        }
#pragma unroll
        for (int i = 0; i < TMf; ++i)
#pragma unroll
          for (int j = 0; j < TNf; ++j) {
            const u32x4 a4 = fa[ks & 1][i], b4 = fb[ks & 1][j];
            if constexpr (sizeof(T) == 2) {
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
                  __builtin_bit_cast(bf16x8, a4), __builtin_bit_cast(bf16x8, b4), acc[i][j], 0, 0, 0);
            } else {
#pragma unroll
              for (int c = 0; c < 4; ++c)
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(
                    __uint_as_float(a4[c]), __uint_as_float(b4[c]), acc[i][j], 0, 0, 0);
            }
          }
      }
    }
    if constexpr (EPI == 0)
      wave_epilogue<TMf, TNf>(p, acc, m0 + wm * WM, n0 + wn * WN, zb, hw_shift, lane, (int)blockIdx.y);
    else
      fused_epilogue<TMf, TNf, EPI>(p, acc, sx, sxx, m0 + wm * WM, n0 + wn * WN, lane, zb);
  }
}

// ------------------------------------------------------------------------------------------
// Direct 3x3 convolution for the 64 -> 64 channel layers at full resolution (slot encoder / VQ-VAE
// at 128^2: M = 1M pixels, N = 64 -- as a GEMM this shape is bound by re-fetching the activation
// tile for each of the 9 taps, 0.4 PF).  bf16, stride 1, pad 1, W % 64 == 0, H % 4 == 0.
//   * persistent workgroup (256 threads = 4 MFMA waves), one per CU; the WHOLE filter
//     ([64 cout][9 taps x 64 cin], 74 KB) is loaded into LDS once;
//   * output tile = 4 image rows x 64 pixels; its 6 x 66 pixel input halo (57 KB, zeros outside
//     the image) is staged once and serves all 9 taps: the tap only shifts the fragment address;
//   * the NEXT tile's halo is prefetched into registers while the current one is multiplied.
// Wave w owns image row oy0 + w: 64 consecutive output pixels x 64 channels (2x2 MFMA tiles), so
// the generic wave epilogue applies unchanged.  Pixel pitch 144 B / filter row pitch 1168 B (odd
// multiples of 16 B): conflict-free ds_read_b128.
constexpr int D33_PIX = 144, D33_WROW = 9 * 64 * 2 + 16;
constexpr int D33_HALO = 6 * 66, D33_HALO_V = (D33_HALO * 8 + 255) / 256;     // uint4 per thread
constexpr int D33_STAGE = 32 * D33_PIX;                       // per-wave epilogue staging (32 px x 32 ch fp32)
constexpr int D33_SMEM = 64 * D33_WROW + D33_HALO * D33_PIX + 4 * D33_STAGE;

// ROWS (round 5; N = 64): MFMA operands swapped -- the accumulators hold C^T, lane = output pixel -- and the row-major
// epilogue of epi_rows.h: a lane packs its four-channel groups, 16 ds_write_b64 + 8 ds_read_b128 per wave and 32-pixel
// half instead of 64 scalar fp32 LDS stores per lane, residual added before the single rounding.
template <bool ROWS>
__global__ __launch_bounds__(256) void conv3x3_c64_kernel(SdmiGemmArgs p, int hw_shift) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const Ws = smem;
  char* const Xs = smem + 64 * D33_WROW;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const bf16_t* __restrict__ Ag = (const bf16_t*)p.a;
  const bf16_t* __restrict__ Wg = (const bf16_t*)p.w;
  const int tiles_x = p.W / 64, tiles_y = p.H / 4;
  const int n_tiles = p.B * tiles_y * tiles_x;
  // ---- the filter: [n][576] rows -> LDS rows of pitch D33_WROW
  for (int v = tid; v < 64 * 72; v += 256) {
    const int n = v / 72, c = v - n * 72;
    const u32x4 w4 = n < p.N ? *reinterpret_cast<const u32x4*>(Wg + (long long)n * p.ldw + c * 8)
                             : u32x4{0u, 0u, 0u, 0u};
    *reinterpret_cast<u32x4*>(Ws + n * D33_WROW + c * 16) = w4;
  }
  // ---- halo fetch of tile t into registers (zeros outside the image)
  u32x4 pre[D33_HALO_V];
  auto fetch = [&](int t) __attribute__((always_inline)) {
    const int b = t / (tiles_y * tiles_x), r = t - b * tiles_y * tiles_x;
    const int ty = r / tiles_x, tx = r - ty * tiles_x;
    const int iy0 = ty * 4 - 1, ix0 = tx * 64 - 1;
#pragma unroll
    for (int i = 0; i < D33_HALO_V; ++i) {
      const int v = tid + i * 256, px = v >> 3, ch = v & 7;
      const int hy = px / 66, hx = px - hy * 66;
      const int iy = iy0 + hy, ix = ix0 + hx;
      const bool ok = px < D33_HALO && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
      pre[i] = ok ? *reinterpret_cast<const u32x4*>(Ag + ((long long)(b * p.H + iy) * p.W + ix) * p.lda + ch * 8)
                  : u32x4{0u, 0u, 0u, 0u};
    }
  };
  auto stash = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < D33_HALO_V; ++i) {
      const int v = tid + i * 256, px = v >> 3, ch = v & 7;
      if (px < D33_HALO) *reinterpret_cast<u32x4*>(Xs + px * D33_PIX + ch * 16) = pre[i];
    }
  };
  int t = blockIdx.x;
  if (t < n_tiles) fetch(t);
  const int R = lane & 31, kb = (lane >> 5) * 16;
  for (; t < n_tiles; t += gridDim.x) {
    __syncthreads();                       // previous tile's fragment reads are done
    stash();
    __syncthreads();
    if (t + (int)gridDim.x < n_tiles) fetch(t + gridDim.x);     // in flight under the MFMAs below
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    // wave's output row = halo row wave + kh; pixel ox = i*32 + R -> halo column ox + kw
    const char* xrow = Xs + (wave * 66 + R) * D33_PIX + kb;
    const char* wrow = Ws + R * D33_WROW + kb;
    u32x4 fa[2][2], fb[2][2];
    auto frags = [&](int step, u32x4 (&a)[2], u32x4 (&b)[2]) __attribute__((always_inline)) {
      const int tap = step >> 2, c = step & 3;
      const int kh = tap / 3, kw = tap - kh * 3;
      const char* xa = xrow + (kh * 66 + kw) * D33_PIX + c * 32;
      const char* wb = wrow + (tap * 64 + c * 16) * 2;
      a[0] = *reinterpret_cast<const u32x4*>(xa);
      a[1] = *reinterpret_cast<const u32x4*>(xa + 32 * D33_PIX);
      b[0] = *reinterpret_cast<const u32x4*>(wb);
      b[1] = *reinterpret_cast<const u32x4*>(wb + 32 * D33_WROW);
    };
    frags(0, fa[0], fb[0]);
#pragma unroll
    for (int step = 0; step < 36; ++step) {
      if (step + 1 < 36) frags(step + 1, fa[(step + 1) & 1], fb[(step + 1) & 1]);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = ROWS ? __builtin_amdgcn_mfma_f32_32x32x16_bf16(
                                 __builtin_bit_cast(bf16x8, fb[step & 1][j]), __builtin_bit_cast(bf16x8, fa[step & 1][i]),
                                 acc[i][j], 0, 0, 0)
                           : __builtin_amdgcn_mfma_f32_32x32x16_bf16(
                                 __builtin_bit_cast(bf16x8, fa[step & 1][i]), __builtin_bit_cast(bf16x8, fb[step & 1][j]),
                                 acc[i][j], 0, 0, 0);
    }
    const int b = t / (tiles_y * tiles_x), r = t - b * tiles_y * tiles_x;
    const int ty = r / tiles_x, tx = r - ty * tiles_x;
    const int m0 = (b * p.H + ty * 4 + wave) * p.W + tx * 64;
    // ---- epilogue: the generic per-element store (64 two-byte stores per lane) is store-issue
    // bound and nothing hides it here, so the wave transposes its tile through LDS (32 pixels at a
    // time) and writes whole 128-byte pixel rows: bias / per-image row vector in registers,
    // residual + activation on the way out.
    char* stg = smem + 64 * D33_WROW + D33_HALO * D33_PIX + wave * D33_STAGE;
    if constexpr (ROWS) {
      static_assert(D33_STAGE >= EpiRows<2>::PATCH, "epilogue patch");
      if (epilogue_rows_ok<2>(p, m0, 0, hw_shift)) wave_epilogue_rows<2>(p, acc, m0, 0, hw_shift, lane, stg);
      else wave_epilogue_rows_generic<2>(p, acc, m0, 0, hw_shift, lane);
      continue;
    }
    const float* rv = p.rowvec ? p.rowvec + (long long)b * p.ldrv : nullptr;
    const bf16_t* resp = (const bf16_t*)p.residual;
    bf16_t* outp = (bf16_t*)p.out;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      // fp32 values are needed until residual + activation are applied: the staging holds fp32,
      // one 32-pixel x 32-channel quarter (4 KB) at a time
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int n = j * 32 + R;
        const int nc = n < p.N ? n : p.N - 1;
        const float add = (p.bias ? p.bias[nc] : 0.f) + (rv ? rv[nc] : 0.f);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // the quarter before was read out (LDS only:
                                                               // global stores / prefetch stay in flight)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
          *reinterpret_cast<float*>(stg + row * D33_PIX + R * 4) = acc[i][j][r] * p.alpha + add;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        // 32 rows x 32 channels fp32: lane -> (row = lane >> 1 [+0], 16-channel half = lane & 1)
        const int row = lane >> 1, hf = lane & 1;
        const int m = m0 + i * 32 + row;
        const int c0 = j * 32 + hf * 16;
        if (c0 < p.N) {
          float v[16];
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const f32x4 t4 = *reinterpret_cast<const f32x4*>(stg + row * D33_PIX + (hf * 16 + q * 4) * 4);
            v[q * 4 + 0] = t4[0]; v[q * 4 + 1] = t4[1]; v[q * 4 + 2] = t4[2]; v[q * 4 + 3] = t4[3];
          }
          if (resp) {
            float rr[8];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
              unpack16<bf16_t>(*reinterpret_cast<const uint4*>(resp + (long long)m * p.ldr + c0 + h * 8), rr);
#pragma unroll
              for (int e = 0; e < 8; ++e) v[h * 8 + e] += rr[e];
            }
          }
          if (p.act) {
#pragma unroll
            for (int e = 0; e < 16; ++e) v[e] = act_apply(v[e], p.act);
          }
#pragma unroll
          for (int h = 0; h < 2; ++h)
            if (c0 + h * 8 < p.N)
              *reinterpret_cast<uint4*>(outp + (long long)m * p.ldc + c0 + h * 8) = pack16<bf16_t>(v + h * 8);
        }
      }
    }
  }
}

// Two entry points over the same body: <= 128 VGPRs (two workgroups per CU) for the tiles whose
// double-buffered LDS image allows it, unconstrained for the 256-row tile (92 KB of LDS).
template <typename T, int BM, int BN, int BKB, int MODE, int EPI = 0, bool XS = false>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(4, 4))) void igemm_kernel(
    SdmiGemmArgs p, int tiles_m, int tiles_n, int kt_per_split, int hw_shift) {
  if constexpr (sizeof(T) == 1) {          // fp8 operands: the device-side half of the scale (sdmi.h: alpha_dev)
    if (p.alpha_dev) p.alpha *= *p.alpha_dev;
  }
  if constexpr (MODE != 1 && EPI == 0 && !XS) parity_select(p, (int)blockIdx.y);
  igemm_body<T, BM, BN, BKB, MODE, EPI, XS>(p, tiles_m, tiles_n, kt_per_split, hw_shift, (int)blockIdx.x,
                                            (int)gridDim.x, (int)blockIdx.y);
}
template <typename T, int BM, int BN, int BKB, int MODE>
__global__ __launch_bounds__(512) void igemm_kernel_tall(SdmiGemmArgs p, int tiles_m, int tiles_n,
                                                         int kt_per_split, int hw_shift) {
  if constexpr (sizeof(T) == 1) {
    if (p.alpha_dev) p.alpha *= *p.alpha_dev;
  }
  if constexpr (MODE != 1) parity_select(p, (int)blockIdx.y);
  igemm_body<T, BM, BN, BKB, MODE>(p, tiles_m, tiles_n, kt_per_split, hw_shift, (int)blockIdx.x, (int)gridDim.x,
                                   (int)blockIdx.y);
}

// split-K second stage: sum partials, apply the same epilogue
__global__ __launch_bounds__(256) void splitk_epilogue_kernel(SdmiGemmArgs p, int hw_shift) {
  const long long total = (long long)p.M * p.N;
  const int HoWo = p.Ho * p.Wo;
  if (p.alpha_dev) p.alpha *= *p.alpha_dev;
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

static int launch_splitk_epilogue(const SdmiGemmArgs& q, int hw_shift, hipStream_t st) {
  const long long total = (long long)q.M * q.N;
  int blocks = (int)((total + 255) / 256);
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(splitk_epilogue_kernel, dim3(blocks), dim3(256), 0, st, q, hw_shift);
  return sdmi_check_launch("igemm splitk epilogue");
}

static int device_cus();

template <typename T, int BM, int BN, int BKB, int MODE, int EPI = 0, bool XS = false>
int launch_cfg(const SdmiGemmArgs& p, int split_k, int hw_shift, hipStream_t st) {
  constexpr int BK = BKB / sizeof(T);
  constexpr int smem = 2 * (BM + BN) * (BKB + 16);
  constexpr int BN_OUT = (EPI & 2) ? BN / 2 : BN;          // GEGLU: value + gate rows per output column
  void (*kern)(SdmiGemmArgs, int, int, int, int);
  if constexpr (BM > 128) kern = igemm_kernel_tall<T, BM, BN, BKB, MODE>;
  else kern = igemm_kernel<T, BM, BN, BKB, MODE, EPI, XS>;
  SDMI_OPTIN_LDS(kern, smem, "igemm");
  SdmiGemmArgs q = p;
  q.split_k = split_k;
  const int tiles_m = (p.M + BM - 1) / BM, tiles_n = (p.N + BN_OUT - 1) / BN_OUT;
  const int nk = (p.K + BK - 1) / BK;
  const int ktps = (nk + split_k - 1) / split_k;
  // persistent workgroups: at most what the chip holds at once (registers / LDS allow 2 workgroups
  // per CU for the 128-row tiles, 3 for 64x64); the rest of the tiles are walked in-kernel
  const int n_cu = device_cus();
  const int ny = split_k * (p.batch > 0 ? p.batch : 1);
  int cap = n_cu * (BM * BN >= 128 * 64 ? 2 : 3) / ny;
  cap = cap < 8 ? 8 : (cap & ~7);          // multiple of 8: a virtual block id keeps its XCD
  const int nwg = tiles_m * tiles_n;
  dim3 grid(nwg <= cap ? nwg : cap, ny);
  hipLaunchKernelGGL(kern, grid, dim3(512), smem, st, q, tiles_m, tiles_n, ktps, hw_shift);
  int rc = sdmi_check_launch("igemm");
  if (rc) return rc;
  if (split_k > 1 && !p.defer_epilogue) rc = launch_splitk_epilogue(q, hw_shift, st);
  return rc;
}

static int device_cus() {
  static int n_cu = 0;
  if (!n_cu) {
    int dev = 0;
    hipDeviceProp_t prop;
    n_cu = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess &&
            prop.multiProcessorCount > 0)
               ? prop.multiProcessorCount
               : 256;
  }
  return n_cu;
}

template <typename T, int BM, int BN, int NSTAGE, int MODE, int LW = 4, bool XS = false, int EPI = 0>
int launch_dma(const SdmiGemmArgs& p, int hw_shift, hipStream_t st, int split_k = 1) {
  constexpr int BK = 128 / sizeof(T);
  constexpr int smem = NSTAGE * (BM + BN) * 128;
  constexpr int threads = 256 + 64 * LW;
  auto kern = igemm_dma_kernel<T, BM, BN, NSTAGE, MODE, LW, XS, EPI>;
  SDMI_OPTIN_LDS(kern, smem, "igemm (lds-dma)");
  SdmiGemmArgs q = p;
  q.split_k = split_k;
  const int tiles_m = (p.M + BM - 1) / BM, tiles_n = (p.N + BN - 1) / BN;
  const int nk = (p.K + BK - 1) / BK;
  const int ktps = (nk + split_k - 1) / split_k;
  const int ny = split_k * (p.batch > 0 ? p.batch : 1);
  // workgroups the chip holds at once: one per CU for the large tiles (their LDS stages fill the CU),
  // two for the 64 x 64 tile
  int cap = device_cus() * (smem <= 72 * 1024 ? 2 : 1) / ny;
  cap = cap < 8 ? 8 : (cap & ~7);
  const int nwg = tiles_m * tiles_n;
  dim3 grid(nwg <= cap ? nwg : cap, ny);
  hipLaunchKernelGGL(kern, grid, dim3(threads), smem, st, q, tiles_m, tiles_n, ktps, hw_shift);
  int rc = sdmi_check_launch("igemm (lds-dma)");
  if (rc) return rc;
  if (split_k > 1 && !p.defer_epilogue) rc = launch_splitk_epilogue(q, hw_shift, st);
  return rc;
}

// SDMI_IGEMM_DMA: 0 off, 1 / 2 experiments (all eligible shapes), 3 the measured-faster shapes, 4 = 3 without the
// extra-source form (A/B of the merged ResBlock tails)
static int dma_mode() {
  static int dma_env = -1;
  if (dma_env < 0) {
    const char* e = getenv("SDMI_IGEMM_DMA");
    dma_env = e ? atoi(e) : SDMI_IGEMM_DMA;
  }
  return dma_env;
}

// smallest K (bytes per row) that takes the 64 x 64 LDS-DMA kernel
static constexpr int dma64_min() { return 256; }

// SDMI_IGEMM_SYM: LDS stages of the symmetric-wave kernel (2: two workgroups per CU, 3 / 4: one), 0 = off,
// unset = by shape (see dispatch)
static int sym_stages() {
  static int v = -1;
  if (v < 0) {
    const char* e = getenv("SDMI_IGEMM_SYM");
    v = e ? atoi(e) : -1;                 // -1 (default): two stages where there are >= 2 tiles per CU
    if (v > 0 && (v < 2 || v > 4)) v = 4;
  }
  return v;
}

// plan_only: return the K split the launch would use (sdmi_igemm_split_plan), no launch
template <typename T>
int dispatch(const SdmiGemmArgs& p, hipStream_t st, bool plan_only = false) {
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
  const long long tm128 = (p.M + 127) / 128;
  // (parity4: the tile shape of ONE parity convolution -- the launch is four of those side by side)
  const long long t128 = tm128 * ((p.N + 127) / 128) * (p.parity4 ? 1 : batch);
  // tile shape: 128x128 when that still gives >= 192 workgroups; narrow outputs (N <= 64: the
  // 64-channel encoder convs at 128x128 pixels) 128x64; everything else 64x64 (+ split-K)
  enum { T128x128, T128x64, T64x64 } shape = T64x64;
  if (p.N > 64) {
    if (t128 >= 192) shape = T128x128;
  } else if (p.N > 32) {
    // (a 256x64 tile was measured 1.6x slower here: 92 KB of LDS = one workgroup per CU, and
    // with K = 576 the prologue / epilogue of each tile is no longer hidden by a neighbour)
    if (tm128 * batch >= 192) shape = T128x64;
  }
  {
    // shallow-K 1x1 problems (<= 512 bytes of K per row: two K tiles) on 64x64 tiles: more, shorter
    // workgroups.  Slower per launch in isolation (10.4 vs 9.7 us at 16384 x 256 x 256), faster inside
    // the sampler (same-box A/B twice: 104.2 / 104.6 vs 105.1 / 105.2 ms per 20-NFE pass); train neutral.
    constexpr int t64_kb = 512;
    if (is1x1 && p.K * (int)sizeof(T) <= t64_kb && shape == T128x128) shape = T64x64;
  }
  const bool big = shape != T64x64;
  // K tile: 128 bytes of K per row when K is deep enough, else 64
  const int kbytes = p.K * (int)sizeof(T);
  const bool wide = kbytes >= 512;
  const bool plain = !is1x1 && !p.ups && p.zins <= 1;
  // the scalar-offset loaders (MODE 1 / 2) address operands with 31-bit byte offsets
  const long long a_bytes =
      ((long long)p.B * p.H * p.W + (long long)(p.KH + 1) * p.W) * p.lda * (long long)sizeof(T);
  const long long w_bytes = (long long)p.N * (p.geglu ? 2 : 1) * p.ldw * (long long)sizeof(T);
  const bool fits31 = a_bytes < (1ll << 31) && w_bytes < (1ll << 31);
  // 3x3 stride-1 same-size convolutions on power-of-two images of 16 / 32 / 64 columns whose 256-pixel tiles are whole
  // image rows (igemm_halo.h): the activation patch of a tile goes to LDS once per 64-channel chunk and serves all nine
  // taps -- 21 KB of operand traffic per K tile instead of 48.  256 x 128 tiles when they fill the chip, else 256 x 64
  // (the 16^2 level at B = 64: 64 row tiles); decided before the K split (these launches never split K)
  int halo_nj = 0, halo_logw = 0;
  if constexpr (sizeof(T) == 2) {              // (bf16: the fp8 instantiation is not built -- igemm_halo.h)
    constexpr int CPC = 128 / (int)sizeof(T);          // channels per 128-byte chunk
    static int halo_min = -1;                // SDMI_IGEMM_HALO: fewest tiles that take it (0 = off)
    if (halo_min < 0) {
      const char* e = getenv("SDMI_IGEMM_HALO");
      halo_min = e ? atoi(e) : 192;
    }
    // (256 x 64 tiles where 256 x 128 leave CUs idle lost: 26.3 vs 27.3 us at 256 -> 256 @16^2 but 44.2 vs 41.4 us at 512 -> 256;
    //  the 64 -> 64 channel layers through 256 x 64 halo tiles lost to conv3x3_c64_kernel: 135.1 vs 120.4 us at 128^2)
    const long long tm256 = (long long)(p.M + 255) / 256;
    const int nj = 2;
    const long long t256 = tm256 * ((p.N + 64 * nj - 1) / (64 * nj));
    // tile width = image width (16 / 32 / 64), or 64-column tiles of four rows on wider images (128^2: the 64-channel layers)
    halo_logw = p.W == 16 ? 4 : (p.W == 32 ? 5 : ((p.W % 64 == 0 && p.H % 4 == 0) ? 6 : 0));
    if (halo_min > 0 && halo_logw && p.KH == 3 && p.KW == 3 && p.stride == 1 && p.pad_t == 1 && p.pad_l == 1 && plain &&
        !p.a2 && p.H == p.Ho && p.W == p.Wo && hw_shift >= 8 && p.Cin % CPC == 0 && p.N > 64 &&
        t256 >= halo_min && p.split_k <= 1 && batch == 1 && fits31 && p.osy == 0 && !p.ln_colsum && !p.geglu && !p.softmax8 &&
        !p.out2 && !p.gn_part && !p.defer_epilogue && p.out_dtype == SDMI_BF16)
      halo_nj = nj;
  }
  if (halo_nj) {
    if (plan_only) return 1;
    return sdmi_launch_halo(p, halo_logw, halo_nj, hw_shift, st, device_cus());
  }
  int split_k = 1;
  if (p.split_k > 0) split_k = p.split_k;       // caller override
  else if (!big && batch == 1 && p.workspace && p.osy == 0) {
    const long long t64 = (long long)((p.M + 63) / 64) * ((p.N + 63) / 64);
    const int nk = (kbytes + (wide ? 127 : 63)) / (wide ? 128 : 64);
    constexpr int sk_target = 384;       // workgroups a split-K launch aims for
    while (t64 * split_k < sk_target && split_k * 2 <= nk / 4 && split_k < 16) split_k *= 2;
  }
  if (split_k > 1 && !p.workspace) split_k = 1;
  if (plan_only) return split_k;
  // direct 3x3 kernel for the 64 -> 64 channel convolutions at full resolution
  if (sizeof(T) == 2 && p.KH == 3 && p.KW == 3 && p.stride == 1 && p.pad_t == 1 && p.pad_l == 1 &&
      !p.a2 && !p.ups && p.zins <= 1 && p.osy == 0 && p.Cin == 64 && p.N <= 64 && p.N > 32 && batch == 1 &&
      p.split_k <= 1 && p.H == p.Ho && p.W == p.Wo && p.W % 64 == 0 && p.H % 4 == 0 &&
      p.out_dtype == SDMI_BF16 && p.ldc % 8 == 0 && p.N % 8 == 0 && !p.bias_m &&
      (!p.residual || p.ldr % 8 == 0) &&
      (long long)p.B * (p.H / 4) * (p.W / 64) >= 2 * device_cus()) {
    SDMI_OPTIN_LDS(conv3x3_c64_kernel<false>, D33_SMEM, "igemm (direct 3x3 c64)");
    SdmiGemmArgs q = p;
    q.split_k = 1;
    int grid = device_cus();
    const long long n_tiles = (long long)p.B * (p.H / 4) * (p.W / 64);
    if (grid > n_tiles) grid = (int)n_tiles;
    hipLaunchKernelGGL(conv3x3_c64_kernel<false>, dim3(grid), dim3(256), D33_SMEM, st, q, hw_shift);
    return sdmi_check_launch("igemm (direct 3x3 c64)");
  }
  // symmetric-wave kernel (igemm_sym.h): 128 x 128 tiles, bf16, 1x1 / plain convolutions, plain epilogue
  if constexpr (sizeof(T) == 2) {
    // default (-1): the two-stage form (two workgroups per CU) where the launch has at least two 128 x 128 tiles
    // per CU -- +5 ... 8 % there in dependent chains (32^2 / 64^2 convolutions at B = 64); at one tile per CU the
    // loader / MFMA kernels stay (profiles/r04_sym_timeline.txt: the four DMA issues of a wave cost as much as its MFMAs)
    int sym = sym_stages();
    if (sym < 0) sym = t128 >= 2 * device_cus() ? 2 : 0;
    const int bk = 64;
    bool ok = sym && shape == T128x128 && wide && split_k == 1 && batch == 1 && fits31 && !p.ln_colsum &&
              !p.geglu && !p.softmax8 && !p.out2 && (is1x1 || (plain && p.KH * p.KW <= 32 && p.Cin % bk == 0));
    if (ok && p.a2) {
      const long long a2_bytes = (long long)p.M * p.lda2 * 2, a3_bytes = p.a3 ? (long long)p.M * p.lda3 * 2 : 0;
      const int kend2 = p.a3 ? p.K2 : p.K;
      const bool same = p.stride == 1 && p.H == p.Ho && p.W == p.Wo;
      ok = (is1x1 || same) && p.K1 == p.KH * p.KW * p.Cin && kend2 > p.K1 && (!p.a3 || p.K > p.K2) && p.K1 % bk == 0 &&
           (kend2 - p.K1) % bk == 0 && (p.K - kend2) % bk == 0 && p.lda2 % VEC == 0 && (!p.a3 || p.lda3 % VEC == 0) &&
           a2_bytes < (1ll << 31) && a3_bytes < (1ll << 31);
    }
    if (ok) {
      const int n_cu = device_cus();
#define SDMI_SYM(NS)                                                                               \
  do {                                                                                             \
    if (p.a2) return is1x1 ? launch_sym<1, NS, true>(p, hw_shift, st, n_cu) : launch_sym<2, NS, true>(p, hw_shift, st, n_cu); \
    return is1x1 ? launch_sym<1, NS>(p, hw_shift, st, n_cu) : launch_sym<2, NS>(p, hw_shift, st, n_cu); \
  } while (0)
      if (sym == 2) SDMI_SYM(2);
      if (sym == 3) SDMI_SYM(3);
      SDMI_SYM(4);
#undef SDMI_SYM
    }
  }
  if (p.a2) {       // extra A sources (sdmi.h: a2 / a3): 1x1, or a stride-1 "same" convolution on the fast path
    const int bk = (wide ? 128 : 64) / (int)sizeof(T);
    const long long a2_bytes = (long long)p.M * p.lda2 * (long long)sizeof(T);
    const long long a3_bytes = p.a3 ? (long long)p.M * p.lda3 * (long long)sizeof(T) : 0;
    const int kend2 = p.a3 ? p.K2 : p.K;
    const bool same = p.stride == 1 && p.H == p.Ho && p.W == p.Wo && !p.ups && p.zins <= 1;
    const bool conv_ok = plain && same && p.KH * p.KW <= 32 && p.Cin % bk == 0;
    if (!(is1x1 || conv_ok) || batch != 1 || p.osy != 0 || p.ln_colsum || p.geglu ||
        p.K1 != p.KH * p.KW * p.Cin || kend2 <= p.K1 || (p.a3 && p.K <= p.K2) || p.K1 % bk || (kend2 - p.K1) % bk ||
        (p.K - kend2) % bk || p.lda2 % VEC || (p.a3 && p.lda3 % VEC) || !fits31 || a2_bytes >= (1ll << 31) ||
        a3_bytes >= (1ll << 31)) {
      sdmi_set_error("igemm: extra A sources need a 1x1 / stride-1 same-size problem with whole K tiles per segment");
      return SDMI_EUNSUPPORTED;
    }
    // their own instantiations (128-byte K tile; 128 x 128 or 64 x 64 output tiles)
    if constexpr (sizeof(T) != 1) {
      if (!wide || shape == T128x64) {
        sdmi_set_error("igemm: extra A sources need K >= 512 bytes per row and N > 64");
        return SDMI_EUNSUPPORTED;
      }
      if (shape == T128x128) {
        // merged ResBlock tails of the 16^2 level (3x3 over h + 1x1 over the skip sources, about one tile per CU): the
        // twelve-wave LDS-DMA kernel the plain 3x3 layers of that level take (below), with the extra-source loaders
        if constexpr (sizeof(T) == 2) {
          const long long t256x = ((p.M + 255) / 256) * ((p.N + 127) / 128);
          if (dma_mode() == 3 && !is1x1 && split_k == 1 && t128 >= 192 && t256x < 192 && kbytes >= 2048 * 2)
            return launch_dma<T, 128, 128, 4, 2, 8, true>(p, hw_shift, st);
        }
        return is1x1 ? launch_cfg<T, 128, 128, 128, 1, 0, true>(p, split_k, hw_shift, st)
                     : launch_cfg<T, 128, 128, 128, 2, 0, true>(p, split_k, hw_shift, st);
      }
      if (sizeof(T) == 2 && dma64_min())
        return is1x1 ? launch_dma<T, 64, 64, 4, 1, 4, true>(p, hw_shift, st, split_k)
                     : launch_dma<T, 64, 64, 4, 2, 4, true>(p, hw_shift, st, split_k);
      return is1x1 ? launch_cfg<T, 64, 64, 128, 1, 0, true>(p, split_k, hw_shift, st)
                   : launch_cfg<T, 64, 64, 128, 2, 0, true>(p, split_k, hw_shift, st);
    } else {
      sdmi_set_error("igemm: extra A sources: bf16 / fp32 operands only");
      return SDMI_EUNSUPPORTED;
    }
  }
  // fused LayerNorm-fold / GEGLU epilogues (sdmi.h: ln_colsum, geglu): 1x1 / linear problems only
  {
    const int epi = (p.ln_colsum ? 1 : 0) | (p.geglu ? 2 : 0) | (p.softmax8 ? 4 : 0);
    if (p.out2 && (!p.geglu || p.ldc2 < 2 * p.N)) {
      sdmi_set_error("igemm: out2 (pre-activation copy) goes with geglu and needs ldc2 >= 2N");
      return SDMI_EUNSUPPORTED;
    }
    if (epi == 5) {            // LayerNorm fold + softmax over 8-column groups (per-image batches allowed)
      if (sizeof(T) == 1 || !is1x1 || !fits31 || p.osy != 0 || p.split_k > 1 || p.bias_m || p.residual ||
          p.rowvec || (p.N & (p.softmax8 > 8 ? 15 : 7)) || p.softmax8 > 16) {
        sdmi_set_error("igemm: softmax8 epilogue needs a plain 1x1 problem with N a multiple of the group width (8; 16 from 9 slots)");
        return SDMI_EUNSUPPORTED;
      }
      if constexpr (sizeof(T) == 2)
        if (dma64_min() && kbytes >= dma64_min()) return launch_dma<T, 64, 64, 4, 1, 4, false, 5>(p, hw_shift, st, 1);
      if constexpr (sizeof(T) != 1)
        return wide ? launch_cfg<T, 64, 64, 128, 1, 5>(p, 1, hw_shift, st)
                    : launch_cfg<T, 64, 64, 64, 1, 5>(p, 1, hw_shift, st);
    } else if (epi) {
      if (sizeof(T) == 1 || (epi & 4) || !is1x1 || !fits31 || batch != 1 || p.osy != 0 || p.split_k > 1 || p.bias_m || p.residual || p.rowvec) {
        sdmi_set_error("igemm: LayerNorm-fold / GEGLU epilogues need a plain 1x1 problem without residual / rowvec");
        return SDMI_EUNSUPPORTED;
      }
      // GEGLU pairs tiles inside a 64-column wave block: 128 x 128 only
      const bool t128 = (epi & 2) || shape == T128x128;
      if constexpr (sizeof(T) == 2)
        if (epi == 1 && !t128 && dma64_min() && kbytes >= dma64_min())
          return launch_dma<T, 64, 64, 4, 1, 4, false, 1>(p, hw_shift, st, 1);
#define SDMI_EPI(E)                                                                              \
  do {                                                                                           \
    if (t128) return wide ? launch_cfg<T, 128, 128, 128, 1, E>(p, 1, hw_shift, st)               \
                          : launch_cfg<T, 128, 128, 64, 1, E>(p, 1, hw_shift, st);               \
    if constexpr (((E) & 2) == 0)                                                                \
      return wide ? launch_cfg<T, 64, 64, 128, 1, E>(p, 1, hw_shift, st)                         \
                  : launch_cfg<T, 64, 64, 64, 1, E>(p, 1, hw_shift, st);                         \
  } while (0)
      if (epi == 1) SDMI_EPI(1);
      if (epi == 2) SDMI_EPI(2);
      SDMI_EPI(3);
#undef SDMI_EPI
    }
  }
  // LDS-DMA kernels (one workgroup per CU, 3-4 LDS stages): deep-K 1x1 / plain convolutions with
  // wide outputs, no split-K
  {
    const int dma_env = dma_mode();
    const bool dma_ok = sizeof(T) != 1 && !p.a2 && !p.parity4 && dma_env && wide && kbytes >= 8 * 128 && split_k == 1 && p.N > 64 && fits31 &&
                        (is1x1 || (plain && p.Cin % (128 / (int)sizeof(T)) == 0)) && p.KH * p.KW <= 32;
    if (dma_ok && dma_env >= 3) {
      // default: the 4-stage 128 x 128 LDS-DMA kernel where it measured faster in dependent chains on MI355X
      // (tools/exp/conv_chain.py, B = 64): about one tile per CU and a deep K -- the 16^2 level's 3x3
      // convolutions, 30.0 -> 27.8 us (256 -> 256) and 50.4 -> 42.4 us (512 -> 256); neutral or slower elsewhere
      const long long t256 = ((p.M + 255) / 256) * ((p.N + 127) / 128) * batch;
      // (eight loader waves: a wave issues one 1 KB DMA piece per ~150 cycles, four cap the feed at ~27 B/clk/CU)
      if (t128 >= 192 && t256 < 192 && kbytes >= 2048 * 2 && !is1x1) return launch_dma<T, 128, 128, 4, 2, 8>(p, hw_shift, st);
    } else if (dma_ok) {
      const long long t256 = ((p.M + 255) / 256) * ((p.N + 127) / 128) * batch;
      if (t256 >= 192) {
        if (is1x1) return launch_dma<T, 256, 128, 3, 1>(p, hw_shift, st);
        return launch_dma<T, 256, 128, 3, 2>(p, hw_shift, st);
      }
      if (t128 >= 192 && dma_env >= 2) {
        if (is1x1) return launch_dma<T, 128, 128, 4, 1>(p, hw_shift, st);
        return launch_dma<T, 128, 128, 4, 2>(p, hw_shift, st);
      }
    }
  }
  if constexpr (sizeof(T) == 2) {
    // LDS-DMA kernel on 64 x 64 tiles (4 stages of 16 KB: two workgroups per CU), split-K as below: the 8^2 /
    // 4^2 levels' convolutions 12 - 17 % faster in dependent chains (29.0 -> 24.3 us 384 -> 384 @8^2, 50.6 -> 41.9
    // 768 -> 384), sampling pass 94.7 -> 92.9 ms, train step 29.58 -> 29.33 ms (same-box A/B, twice each).
    // (128 x 64 tiles for the 8^2 level: faster per launch in isolation -- 384 -> 384 25.1 -> 23.6 us -- and slower inside the
    // replayed sampler, 74.84 -> 75.41 ms, and the train step, 26.78 -> 26.97 ms: 192 workgroups leave a quarter of the CUs
    // idle; deleted)
    const int dma64 = dma64_min();
    if (dma64 && shape == T64x64 && !p.a2 && kbytes >= dma64 && fits31 && (batch == 1 || p.parity4)) {
      if (is1x1) return launch_dma<T, 64, 64, 4, 1, 4>(p, hw_shift, st, split_k);
      if (plain && p.KH * p.KW <= 32 && p.Cin % 64 == 0) return launch_dma<T, 64, 64, 4, 2, 4>(p, hw_shift, st, split_k);
    }
  }
#define SDMI_GO(BM, BN, BKB)                                                                    \
  do {                                                                                          \
    if (is1x1 && fits31) return launch_cfg<T, BM, BN, BKB, 1>(p, split_k, hw_shift, st);        \
    if (plain && fits31 && p.KH * p.KW <= 32 && p.Cin % (BKB / (int)sizeof(T)) == 0)            \
      return launch_cfg<T, BM, BN, BKB, 2>(p, split_k, hw_shift, st);                           \
    return launch_cfg<T, BM, BN, BKB, 0>(p, split_k, hw_shift, st);                             \
  } while (0)
  switch (shape) {
    case T128x128: if (wide) { SDMI_GO(128, 128, 128); } else { SDMI_GO(128, 128, 64); }
    case T128x64: if (wide) { SDMI_GO(128, 64, 128); } else { SDMI_GO(128, 64, 64); }
    default: if (wide) { SDMI_GO(64, 64, 128); } else { SDMI_GO(64, 64, 64); }
  }
#undef SDMI_GO
}

}  // namespace

extern "C" int sdmi_igemm(const SdmiGemmArgs* a, void* stream) {
  SDMI_REQUIRE(a && a->a && a->w && a->out, "null pointer");
  SDMI_REQUIRE(a->dtype == SDMI_F32 || a->dtype == SDMI_BF16 || a->dtype == SDMI_FP8, "bad dtype");
  SDMI_REQUIRE(a->out_dtype == SDMI_F32 || a->out_dtype == SDMI_BF16, "bad out_dtype");
  const int vec = a->dtype == SDMI_FP8 ? 16 : (a->dtype == SDMI_BF16 ? 8 : 4);
  SDMI_REQUIRE(!a->alpha_dev || (a->dtype == SDMI_FP8 && !a->defer_epilogue), "alpha_dev goes with fp8 operands (no deferred epilogue)");
  SDMI_REQUIRE(a->dtype != SDMI_FP8 || (!(a->batch > 1) && !a->ups && a->zins <= 1),
               "fp8 operands: plain convolution / linear only (no batch, upsample fold)");
  SDMI_REQUIRE(a->M > 0 && a->N > 0 && a->K > 0, "empty problem");
  SDMI_REQUIRE(a->a2 ? (a->K1 == a->KH * a->KW * a->Cin && a->K > a->K1) : a->K == a->KH * a->KW * a->Cin,
               "K != KH*KW*Cin (+ extra sources)");
  SDMI_REQUIRE(a->Cin % vec == 0 && a->lda % vec == 0 && a->ldw % vec == 0,
               "Cin/lda/ldw must be multiples of the 16-byte vector width");
  SDMI_REQUIRE(((uintptr_t)a->a & 15) == 0 && ((uintptr_t)a->w & 15) == 0, "unaligned operand");
  SDMI_REQUIRE(a->M == a->B * a->Ho * a->Wo, "M != B*Ho*Wo");
  SDMI_REQUIRE(!(a->zins > 1 && (a->ups || a->stride != 1)), "zins excludes ups / stride");
  SDMI_REQUIRE(!a->parity4 || (a->batch == 4 && a->KH == 2 && a->KW == 2 && a->stride == 1 && a->osy == 2 && a->osx == 2 &&
                               a->sa == 0 && a->sc == 0 && a->sw > 0 && !a->a2 && !a->ups && a->zins <= 1 && a->split_k <= 1 &&
                               !a->residual && !a->ln_colsum && !a->geglu && !a->softmax8 && !a->gn_part &&
                               !a->defer_epilogue && a->dtype != SDMI_FP8),
               "parity4: four 2x2 parity convolutions (batch 4, osy = osx = 2, shared input and output)");
  SDMI_REQUIRE(!(a->batch > 1) || a->parity4 || (a->KH == 1 && a->KW == 1), "batched mode is 1x1 only");
  SDMI_REQUIRE(!(a->batch > 1 && a->split_k > 1), "batched split-K unsupported");
  SDMI_REQUIRE(a->sa % vec == 0 && a->sw % vec == 0, "batch strides must keep 16-byte alignment");
  SDMI_REQUIRE(a->osy == 0 || (a->osy > 0 && a->osx > 0 && a->oH > 0 && a->oW > 0 &&
                               a->split_k <= 1 && (!(a->batch > 1) || a->parity4)),
               "sub-sampled output: needs osy/osx/oH/oW > 0, no split-K / batch");
  SDMI_REQUIRE(a->osy == 0 || !a->residual || a->ldr == a->ldc,
               "sub-sampled output: the residual shares the output's layout");
  SDMI_REQUIRE(a->osy == 0 || (!a->rowvec && !a->act && !a->bias_m), "sub-sampled output: plain epilogue only");
  if (a->gn_part) {
    const int hw = a->Ho * a->Wo, cpg = a->gn_groups > 0 ? a->N / a->gn_groups : 0;
    SDMI_REQUIRE(a->out_dtype == SDMI_BF16 && a->dtype != SDMI_F32 && a->M % 128 == 0 && a->N % 128 == 0 && hw >= 32 &&
                     (hw & (hw - 1)) == 0 && a->gn_groups > 0 && a->N % a->gn_groups == 0 && cpg >= 1 && cpg <= 32 &&
                     (cpg & (cpg - 1)) == 0 && a->split_k == 1 && !a->bias_m && !a->ln_colsum && !a->geglu &&
                     !a->softmax8 && !a->out2 && a->osy == 0 && !(a->batch > 1) && a->ldc == a->N &&
                     (long long)a->M * (a->ldc > a->ldr ? a->ldc : a->ldr) < (1ll << 30),
                 "gn_part: bf16, whole 128 x 128 tiles, Ho*Wo = 2^k >= 32, N / gn_groups a power of two <= 32, plain epilogue");
  }
  SDMI_REQUIRE(!a->defer_epilogue || (a->workspace && !a->act && a->ldc == a->N && !a->bias_m && !a->ln_colsum &&
                                      !a->geglu && !a->softmax8 && !a->out2 && a->osy == 0 && !(a->batch > 1)),
               "defer_epilogue: plain epilogue (alpha / bias / rowvec / residual), ldc = N, workspace required");
  hipStream_t st = (hipStream_t)stream;
  if (a->dtype == SDMI_FP8) return dispatch<fp8_t>(*a, st);
  return a->dtype == SDMI_BF16 ? dispatch<bf16_t>(*a, st) : dispatch<float>(*a, st);
}

extern "C" int sdmi_igemm_split_plan(const SdmiGemmArgs* a, void*) {
  if (!a || a->M <= 0 || a->N <= 0 || a->K <= 0) return 1;
  if (a->dtype == SDMI_FP8) return dispatch<fp8_t>(*a, nullptr, true);
  return a->dtype == SDMI_BF16 ? dispatch<bf16_t>(*a, nullptr, true) : dispatch<float>(*a, nullptr, true);
}

extern "C" int sdmi_splitk_finish(const SdmiGemmArgs* a, void* stream) {
  SDMI_REQUIRE(a && a->workspace && a->out && a->split_k > 1 && a->M > 0 && a->N > 0, "bad args");
  int hw_shift = -1;
  const int hw = a->Ho * a->Wo;
  if (hw > 0 && (hw & (hw - 1)) == 0) {
    hw_shift = 0;
    while ((1 << hw_shift) < hw) ++hw_shift;
  }
  return launch_splitk_epilogue(*a, hw_shift, (hipStream_t)stream);
}
