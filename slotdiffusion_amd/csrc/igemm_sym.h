// Symmetric-wave implicit GEMM (round 4): the main loop section 8.1 of the round-3 design asked for.
//
// igemm_kernel / igemm_dma_kernel split a workgroup into loader waves and MFMA waves that meet at one
// barrier per K tile: the phases of a K tile (operand fetch, fragment reads + MFMAs, epilogue) ADD UP, and
// each SIMD has a single MFMA wave whose LDS latency nothing covers (33 us where the MFMAs need
// 9).  Here all EIGHT waves of the workgroup are the same program:
//   * 128 x 128 output tile, wave grid 2 (M) x 4 (N): a wave owns 64 x 32 = two 32x32 MFMA tiles (32
//     accumulator registers), so every SIMD hosts TWO MFMA-issuing waves that cover each other's
//     fragment-read latency, and 128 x 128 tiles still give 256 - 512 workgroups at B = 64;
//   * every wave issues its own share of the K tile's operand pieces (`buffer_load_dwordx4 ... lds`, 1 KB
//     per instruction: two A pieces + two B pieces per wave and K tile) right after the barrier that
//     retires the stage they refill, and waits for them with a COUNTED `s_waitcnt vmcnt((NSTAGE-2)*4)`
//     NSTAGE-1 K tiles later -- never vmcnt(0) inside the loop, no workgroup-scope fence (it would drain
//     the DMA queue);
//   * address generation is scalar in the steady state (per-tile voffsets, K walk in an SGPR soffset,
//     borders as out-of-range offsets) exactly like igemm_dma_kernel's loaders;
//   * the pipeline is flat over (output tile, K tile): the first K tiles of the next output tile are in
//     flight while the epilogue of the current one stores.
// LDS image: NSTAGE stages of [128 A rows | 128 B rows] x 128 B, XOR swizzled (16-byte chunk c of row r at
// chunk c ^ ((r >> 1) & 7)) by fetching the chunk that belongs at each lane's position.
#pragma once
#include "igemm_body.h"

namespace {

#ifdef SDMI_SYM_TIMELINE   // experiment (tools/exp/sym_timeline.py): s_memtime sums of wave 0 per workgroup -> p.workspace
#define SYM_TL_DECL unsigned long long tl_t = __builtin_amdgcn_s_memtime(), tl_a[6] = {0, 0, 0, 0, 0, 0};
#define SYM_TL_LAP(i) do { const unsigned long long n_ = __builtin_amdgcn_s_memtime(); tl_a[i] += n_ - tl_t; tl_t = n_; } while (0)
#define SYM_TL_FLUSH do { if (threadIdx.x == 0 && p.workspace) for (int i_ = 0; i_ < 6; ++i_) ((unsigned long long*)p.workspace)[blockIdx.x * 8 + i_] = tl_a[i_]; } while (0)
#else
#define SYM_TL_DECL
#define SYM_TL_LAP(i)
#define SYM_TL_FLUSH
#endif

template <int MODE, int NSTAGE, bool XS = false>
__global__ __launch_bounds__(512, (NSTAGE == 2 ? 4 : 2)) void igemm_sym_kernel(SdmiGemmArgs p, int tiles_m, int tiles_n, int hw_shift,
                                                                              int kt_per_split) {
  typedef bf16_t T;
  constexpr int VEC = 8, BK = 64, BM = 128, BN = 128;
  constexpr int STAGE = (BM + BN) * 128;
  constexpr int NLOAD = 4;                             // DMA instructions per wave and K tile
  constexpr unsigned OOB = 0x80000000u;
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
  // split-K (grid y = K slice): partial sums to p.workspace, folded by splitk_epilogue_kernel
  const int nk_total = (p.K + BK - 1) / BK;
  const int kt_begin = (int)blockIdx.y * kt_per_split;
  const int kt_end = kt_begin + kt_per_split < nk_total ? kt_begin + kt_per_split : nk_total;
  const int n_kt = kt_end > kt_begin ? kt_end - kt_begin : 0;
  const int total = my_tiles * n_kt;
  if (total == 0 && p.split_k <= 1) return;      // (an empty K slice still writes its zero partial)

  const int tid = threadIdx.x, l = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  // ------------------------------- operand fetch state (per wave) -------------------------------
  const int kc = (l & 7) ^ ((4 * (w & 1) + (l >> 4)) & 7);     // logical chunk fetched by this lane
  const T* Ag = (const T*)p.a;
  if (MODE == 2) Ag -= (long long)(p.pad_t * p.W + p.pad_l) * p.lda;
  const T* Wg = (const T*)p.w;
  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)Ag, 0, (int)OOB, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc((void*)Wg, 0, (int)OOB, 0x00020000);
  unsigned a_vo[2], a_cur[2], a_inv[2], b_vo[2], b_cur[2];
  unsigned a_vo2[XS ? 2 : 1], a_vo3[XS ? 2 : 1];
  int ld_tile = 0, ld_kt = 0, k0 = 0, ci = 0, kh = 0, kw = 0;   // wave-uniform
  auto begin_tile = [&]() __attribute__((always_inline)) {
    int m0, n0;
    tile_of((int)blockIdx.x + ld_tile * (int)gridDim.x, m0, n0);
    k0 = kt_begin * BK; ci = 0; kh = 0; kw = 0;
    if (MODE == 2) {
      const int tap = k0 / p.Cin;
      ci = k0 - tap * p.Cin;
      kh = tap / p.KW;
      kw = tap - kh * p.KW;
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int row = (w + 8 * i) * 8 + (l >> 3);
      const int m = min(m0 + row, p.M - 1);
      if constexpr (XS) {
        a_vo2[i] = ((unsigned)m * (unsigned)p.lda2 + kc * VEC) * 2u;
        a_vo3[i] = ((unsigned)m * (unsigned)p.lda3 + kc * VEC) * 2u;
      }
      if (MODE == 1) {
        a_vo[i] = ((unsigned)m * (unsigned)p.lda + kc * VEC) * 2u;
        a_inv[i] = 0;
      } else {
        const int HoWo = p.Ho * p.Wo;
        const bool wo2 = (p.Wo & (p.Wo - 1)) == 0;
        const int b = hw_shift >= 0 ? (m >> hw_shift) : (m / HoWo);
        const int rem = m - b * HoWo;
        const int oy = wo2 ? (rem >> (31 - __builtin_clz(p.Wo))) : rem / p.Wo;
        const int ox = rem - oy * p.Wo;
        const int iy0 = oy * p.stride - p.pad_t, ix0 = ox * p.stride - p.pad_l;
        a_vo[i] = ((unsigned)((b * p.H + oy * p.stride) * p.W + ox * p.stride) * (unsigned)p.lda + kc * VEC) * 2u;
        unsigned rb = 0, cb = 0, inv = 0;
        for (int q = 0; q < p.KH; ++q) rb |= ((unsigned)(iy0 + q) < (unsigned)p.H ? 0u : 1u) << q;
        for (int q = 0; q < p.KW; ++q) cb |= ((unsigned)(ix0 + q) < (unsigned)p.W ? 0u : 1u) << q;
        for (int q = 0; q < p.KH; ++q) inv |= (((rb >> q) & 1u) ? ((1u << p.KW) - 1u) : cb) << (q * p.KW);
        a_inv[i] = inv;
      }
      a_cur[i] = a_vo[i];
      const int n = min(n0 + row, p.N - 1);
      b_vo[i] = ((unsigned)n * (unsigned)p.ldw + kc * VEC) * 2u;
      b_cur[i] = b_vo[i];
    }
  };
  int ld_stage = 0;
  auto issue = [&]() __attribute__((always_inline)) {
    if (ld_kt == 0) begin_tile();
    unsigned so_a;
    if (MODE == 2) {
      if (ci == 0 || ld_kt == 0) {                   // new filter tap: its validity mask
        const int tap = kh * p.KW + kw;
#pragma unroll
        for (int i = 0; i < 2; ++i) a_cur[i] = ((a_inv[i] >> tap) & 1u) ? OOB : a_vo[i];
      }
      so_a = (unsigned)((kh * p.W + kw) * p.lda + ci) * 2u;
    } else {
      if (k0 + BK > p.K) {                           // K tail (last K tile of an output tile only)
        const bool k_ok = k0 + kc * VEC < p.K;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          a_cur[i] = k_ok ? a_vo[i] : OOB;
          b_cur[i] = k_ok ? b_vo[i] : OOB;
        }
      }
      so_a = (unsigned)k0 * 2u;
    }
    const unsigned so_b = (unsigned)k0 * 2u;
    char* st = smem + ld_stage * STAGE + w * 1024;
    if constexpr (XS) {
      const bool s1 = p.a2 != nullptr && k0 >= p.K1;
      const bool s2 = s1 && p.a3 != nullptr && k0 >= p.K2;
      const unsigned so_x = s2 ? (unsigned)(k0 - p.K2) * 2u : (s1 ? (unsigned)(k0 - p.K1) * 2u : so_a);
      const void* base_x = s2 ? p.a3 : (s1 ? p.a2 : (const void*)Ag);
      const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc((void*)base_x, 0, (int)OOB, 0x00020000);
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const unsigned vo = s2 ? a_vo3[i] : (s1 ? a_vo2[i] : a_cur[i]);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x, (lds_void*)(st + i * 8192), 16, (int)vo, (int)so_x, 0, 0);
      }
    } else {
#pragma unroll
      for (int i = 0; i < 2; ++i)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lds_void*)(st + i * 8192), 16, (int)a_cur[i], (int)so_a, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, (lds_void*)(st + BM * 128 + i * 8192), 16, (int)b_cur[i],
                                               (int)so_b, 0, 0);
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

  // ------------------------------------ MFMA side (per wave) ------------------------------------
  const int wm = w >> 2, wn = w & 3;
  const int R = l & 31;
  int swz[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) swz[ks] = ((2 * ks + (l >> 5)) ^ ((R >> 1) & 7)) * 16;
  const int a_off = (wm * 64 + R) * 128;
  const int b_off = BM * 128 + (wn * 32 + R) * 128;
  auto read_frags = [&](const char* base, int ks, u32x4 (&fa)[2], u32x4& fb) __attribute__((always_inline)) {
    fa[0] = *reinterpret_cast<const u32x4*>(base + a_off + swz[ks]);
    fa[1] = *reinterpret_cast<const u32x4*>(base + a_off + 4096 + swz[ks]);
    fb = *reinterpret_cast<const u32x4*>(base + b_off + swz[ks]);
  };

  // Steps past the last one re-fetch clamped rows of a non-existent tile into a stage nobody reads any
  // more: the number of DMA groups in flight stays static, so a fixed vmcnt works.
  if (total > 0) {
#pragma unroll
    for (int s = 0; s < NSTAGE - 1; ++s) issue();
  }
  int stage = 0;
  SYM_TL_DECL
  SYM_TL_LAP(5);                            // prologue: tile setup + first DMA issues
  for (int ti = 0; ti < my_tiles; ++ti) {
    int m0, n0;
    tile_of((int)blockIdx.x + ti * (int)gridDim.x, m0, n0);
    f32x16 acc[2][1];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][0][r] = 0.f;
    for (int t = 0; t < n_kt; ++t) {
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NSTAGE - 2) * NLOAD) : "memory");   // this wave's pieces of the K tile
      SYM_TL_LAP(0);
      __builtin_amdgcn_s_barrier();          // everybody's pieces landed; the previous stage is free
      asm volatile("" ::: "memory");
      SYM_TL_LAP(1);
      issue();                               // K tile + NSTAGE - 1 -> the stage the previous K tile vacated
      SYM_TL_LAP(2);
      const char* base = smem + stage * STAGE;
      if (++stage == NSTAGE) stage = 0;
      u32x4 fa[2][2], fb[2];
      read_frags(base, 0, fa[0], fb[0]);
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        if (ks + 1 < 4) read_frags(base, ks + 1, fa[(ks + 1) & 1], fb[(ks + 1) & 1]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < 2; ++i)
          acc[i][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fa[ks & 1][i]),
                                                              __builtin_bit_cast(bf16x8, fb[ks & 1]), acc[i][0], 0, 0, 0);
      }
#ifdef SDMI_SYM_TIMELINE
      asm volatile("s_nop 0" :: "v"(acc[0][0][0]), "v"(acc[1][0][0]));      // (the MFMAs of this K tile have issued)
#endif
      SYM_TL_LAP(3);
    }
    // The streamlined epilogue is called directly: behind the generic one (loads and stores under per-element
    // branches) the compiler's wait-count pass carries pending-load state around the loop and parks an
    // `s_waitcnt vmcnt(4)` in front of the fragment reads of EVERY K tile -- a drain of the prefetched stages.
    // The generic path (edge tiles, fp32 out) therefore ends in a compiler-visible vmcnt(0).
    const int mw0 = m0 + wm * 64, nw0 = n0 + wn * 32;
    if (epilogue_fast_ok<2, 1>(p, mw0, nw0, hw_shift)) {
      wave_epilogue_fast<2, 1>(p, acc, mw0, nw0, 0, hw_shift, l);
    } else {
      wave_epilogue<2, 1>(p, acc, mw0, nw0, 0, hw_shift, l, (int)blockIdx.y);
      __builtin_amdgcn_s_waitcnt(0x0F70);       // vmcnt(0)
    }
    SYM_TL_LAP(4);
  }
  SYM_TL_FLUSH;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // no DMA may outlive the workgroup's LDS
}

template <int MODE, int NSTAGE, bool XS = false>
int launch_sym(const SdmiGemmArgs& p, int hw_shift, hipStream_t st, int n_cu, int split_k = 1) {
  constexpr int smem = NSTAGE * 256 * 128;
  auto kern = igemm_sym_kernel<MODE, NSTAGE, XS>;
  SDMI_OPTIN_LDS(kern, smem, "igemm (symmetric waves)");
  SdmiGemmArgs q = p;
  q.split_k = split_k;
  const int tiles_m = (p.M + 127) / 128, tiles_n = (p.N + 127) / 128;
  const int nk = (p.K + 63) / 64, ktps = (nk + split_k - 1) / split_k;
  int cap = n_cu * (smem <= 80 * 1024 ? 2 : 1) / split_k;
  cap = cap < 8 ? 8 : (cap & ~7);
  const int nwg = tiles_m * tiles_n;
  hipLaunchKernelGGL(kern, dim3(nwg <= cap ? nwg : cap, split_k), dim3(512), smem, st, q, tiles_m, tiles_n, hw_shift, ktps);
  return sdmi_check_launch("igemm (symmetric waves)");
}

}  // namespace
