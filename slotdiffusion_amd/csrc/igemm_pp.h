// Ping-pong implicit GEMM on a 256 x 128 tile with 64 x 64 wave tiles (round 5).
//
// What bounded every 128 x 128 kernel of rounds 1 - 4: operand bytes per flop (a 128 x 128 tile asks the
// L2 -> LDS feed for 64 B/clk per CU at full matrix rate), 12 `ds_read_b128` per 8 MFMAs in a 64 x 32 wave tile, and
// a loop in which all eight waves do the same thing at the same time -- issue DMA, then MFMA -- so the phases add up.
// Here:
//   * 256 x 128 output tile, eight waves as 4 (M) x 2 (N), each a 64 x 64 block = four 32x32 accumulators: 16 MFMAs
//     per 16 `ds_read_b128` and K tile, 48 KB of operands per 2 x 256 x 128 x 64 flops (3/4 of the 128 x 128 tile's
//     bytes per flop);
//   * the two waves of a SIMD belong to different GROUPS (waves 0-3 / 4-7) that run the same program one barrier
//     apart: while one group is in its MFMA segment (8 back-to-back `v_mfma_f32_32x32x16_bf16` = 256 cycles of the
//     SIMD's matrix pipe) the other is in its LOAD segment (8 fragment reads for its next half K tile + 3 LDS-DMA
//     pieces of the K tile two ahead), then they swap -- matrix beside memory on every SIMD, four barrier intervals
//     per K tile (guide: "two waves per SIMD", the 8-phase template's `if (wr == 1) s_barrier` stagger);
//   * three LDS stages of [256 A rows | 128 B rows] x 128 B (XOR-swizzled chunks, igemm_sym.h's image), filled by
//     `buffer_load_dwordx4 ... lds` with scalar-only address walks; a wave issues 6 pieces per K tile and waits for
//     its pieces of the NEXT K tile with one counted `s_waitcnt vmcnt(6)` per K tile (never 0 in the loop);
//   * hazards by barrier count (interval i = between barrier i and i + 1; group 0 runs L(t,0) M(t,0) L(t,1) M(t,1)
//     in intervals 4t .. 4t+3, group 1 one interval later): the last `ds_read` of K tile t is group 1's L(t,1) in
//     interval 4t+3, retired by `lgkmcnt(0)` before barrier 4t+4; the stage is refilled (K tile t+3) from the L
//     segments of K tile t+1, interval >= 4t+4.  A wave's pieces of K tile t+1 are waited for at the end of its
//     L(t,1) (interval <= 4t+3), a barrier follows, the first read is group 0's L(t+1,0) in interval 4t+4;
//   * the pipeline is flat over (output tile, K tile); at an output-tile boundary group 0 idles one interval so that
//     both groups store their accumulators in the same interval (one workgroup per CU: nothing else would overlap
//     a serial pair of epilogues), then group 1 idles one to restore the stagger.
#pragma once
#include "igemm_body.h"
#include "epi_rows.h"

namespace {

#ifndef SDMI_PP_KSEG
#define SDMI_PP_KSEG 2
#endif
#ifndef SDMI_PP_EXP       // ablation bit mask (tools/exp/pp_ablate.sh): 1 no MFMA, 2 no DMA in the loop, 4 no fragment reads,
#define SDMI_PP_EXP 0     // 8 un-swizzled DMA source, 16 only the A pieces, 32 only the B pieces, 64 no s_setprio (1 - 32: results are wrong with any bit set)
#endif
#ifdef SDMI_PP_TIMELINE   // experiment (tools/exp/pp_timeline.py): s_memtime sums of waves 0 and 4 per workgroup -> p.workspace
#define PP_TL_DECL unsigned long long tl_t = __builtin_amdgcn_s_memtime(), tl_a[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define PP_TL_LAP(i) do { const unsigned long long n_ = __builtin_amdgcn_s_memtime(); tl_a[i] += n_ - tl_t; tl_t = n_; } while (0)
#define PP_TL_FLUSH do { if ((threadIdx.x & 255) == 0 && p.workspace) for (int i_ = 0; i_ < 8; ++i_) ((unsigned long long*)p.workspace)[(blockIdx.x * 2 + (threadIdx.x >> 8)) * 8 + i_] = tl_a[i_]; } while (0)
#else
#define PP_TL_DECL
#define PP_TL_LAP(i)
#define PP_TL_FLUSH
#endif

template <int MODE, bool XS = false>
__global__ __launch_bounds__(512, 2) void igemm_pp_kernel(SdmiGemmArgs p, int tiles_m, int tiles_n, int hw_shift) {
  typedef bf16_t T;
  constexpr int VEC = 8, BK = 64, BM = 256, BN = 128, NSTAGE = 3;
  constexpr int STAGE = (BM + BN) * 128;               // 48 KB
  constexpr int NA = BM / 64, NB = BN / 64;            // DMA pieces (8 rows x 128 B) per wave and K tile: 4 A + 2 B
  constexpr unsigned OOB = 0x80000000u;
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
  const int n_kt = (p.K + BK - 1) / BK;
  if (my_tiles == 0) return;

  const int tid = threadIdx.x, l = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = w >> 2;                              // the two waves of a SIMD (w, w + 4) sit in different groups
  // ------------------------------- operand fetch state (per wave) -------------------------------
  // piece q of a stage = rows 8q .. 8q+7; wave w owns A pieces w + 8i (i < 4) and B pieces w + 8j (j < 2); lane l
  // fetches the 16-byte chunk that belongs at its position of the swizzled image (chunk c of row r at c ^ ((r >> 1) & 7))
  const int kc = (SDMI_PP_EXP & 8) ? (l & 7) : ((l & 7) ^ ((4 * (w & 1) + (l >> 4)) & 7));
  const T* Ag = (const T*)p.a;
  if (MODE == 2) Ag -= (long long)(p.pad_t * p.W + p.pad_l) * p.lda;
  const T* Wg = (const T*)p.w;
  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)Ag, 0, (int)OOB, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc((void*)Wg, 0, (int)OOB, 0x00020000);
  unsigned a_vo[NA], a_cur[NA], a_inv[NA], b_vo[NB], b_cur[NB];
  unsigned a_vo2[XS ? NA : 1], a_vo3[XS ? NA : 1];
  int ld_tile = 0, ld_kt = 0, k0 = 0, ci = 0, kh = 0, kw = 0;   // wave-uniform
  unsigned so_a = 0, so_b = 0;                                  // this K tile's scalar offsets (set by issue0)
  auto begin_tile = [&]() __attribute__((always_inline)) {
    int m0, n0;
    tile_of((int)blockIdx.x + ld_tile * (int)gridDim.x, m0, n0);
    k0 = 0; ci = 0; kh = 0; kw = 0;
#pragma unroll
    for (int i = 0; i < NA; ++i) {
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
    }
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      const int row = (w + 8 * j) * 8 + (l >> 3);
      const int n = min(n0 + row, p.N - 1);
      b_vo[j] = ((unsigned)n * (unsigned)p.ldw + kc * VEC) * 2u;
      b_cur[j] = b_vo[j];
    }
  };
  int ld_stage = 0;
  // one A piece of the K tile being fetched (its source: the convolution's activation, or an extra source appended
  // along K -- sdmi.h: a2 / a3)
  bool dma_on = true;
  auto dma_a = [&](int i, char* st) __attribute__((always_inline)) {
    if ((SDMI_PP_EXP & 32) || ((SDMI_PP_EXP & 2) && !dma_on)) return;
    if constexpr (XS) {
      const bool s1 = p.a2 != nullptr && k0 >= p.K1;
      const bool s2 = s1 && p.a3 != nullptr && k0 >= p.K2;
      const unsigned so_x = s2 ? (unsigned)(k0 - p.K2) * 2u : (s1 ? (unsigned)(k0 - p.K1) * 2u : so_a);
      const void* base_x = s2 ? p.a3 : (s1 ? p.a2 : (const void*)Ag);
      const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc((void*)base_x, 0, (int)OOB, 0x00020000);
      const unsigned vo = s2 ? a_vo3[i] : (s1 ? a_vo2[i] : a_cur[i]);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x, (lds_void*)(st + (w + 8 * i) * 1024), 16, (int)vo, (int)so_x, 0, 0);
    } else {
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lds_void*)(st + (w + 8 * i) * 1024), 16, (int)a_cur[i], (int)so_a, 0, 0);
    }
  };
  auto dma_b = [&](int j, char* st) __attribute__((always_inline)) {
    if ((SDMI_PP_EXP & 16) || ((SDMI_PP_EXP & 2) && !dma_on)) return;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, (lds_void*)(st + BM * 128 + (w + 8 * j) * 1024), 16, (int)b_cur[j],
                                             (int)so_b, 0, 0);
  };
  // first half of a K tile's pieces (A0, A1, B0): also sets up the K tile's scalar state
  auto issue0 = [&]() __attribute__((always_inline)) {
    if (ld_kt == 0) begin_tile();
    if (MODE == 2) {
      if (ci == 0 || ld_kt == 0) {                   // new filter tap: its validity mask
        const int tap = kh * p.KW + kw;
#pragma unroll
        for (int i = 0; i < NA; ++i) a_cur[i] = ((a_inv[i] >> tap) & 1u) ? OOB : a_vo[i];
      }
      so_a = (unsigned)((kh * p.W + kw) * p.lda + ci) * 2u;
    } else {
      if (k0 + BK > p.K) {                           // K tail (last K tile of an output tile only)
        const bool k_ok = k0 + kc * VEC < p.K;
#pragma unroll
        for (int i = 0; i < NA; ++i) a_cur[i] = k_ok ? a_vo[i] : OOB;
#pragma unroll
        for (int j = 0; j < NB; ++j) b_cur[j] = k_ok ? b_vo[j] : OOB;
      }
      so_a = (unsigned)k0 * 2u;
    }
    so_b = (unsigned)k0 * 2u;
    char* st = smem + ld_stage * STAGE;
    dma_a(0, st);
    dma_a(1, st);
    dma_b(0, st);
  };
  // second half (A2, A3, B1), then the walk to the next K tile
  auto issue1 = [&]() __attribute__((always_inline)) {
    char* st = smem + ld_stage * STAGE;
    dma_a(2, st);
    dma_a(3, st);
    dma_b(1, st);
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
  const int wm = w & 3, wn = w >> 2;
  const int R = l & 31;
  int swz[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) swz[ks] = ((2 * ks + (l >> 5)) ^ ((R >> 1) & 7)) * 16;
  const int a_off = (wm * 64 + R) * 128;
  const int b_off = BM * 128 + (wn * 64 + R) * 128;
  // KSEG k-steps (16 of the K tile's 64 columns each) per LOAD / MFMA segment: 2 = four barrier intervals per K tile
  // (8 MFMAs each), 4 = two intervals (16 MFMAs each; 64 fragment registers)
  constexpr int KSEG = SDMI_PP_KSEG, NSEG = 4 / KSEG;
  u32x4 fa[KSEG][2] = {}, fb[KSEG][2] = {};       // [k-step of the segment][row / column block]
  auto read_seg = [&](const char* base, int h) __attribute__((always_inline)) {
    if (SDMI_PP_EXP & 4) return;
#pragma unroll
    for (int s = 0; s < KSEG; ++s) {
      fa[s][0] = *reinterpret_cast<const u32x4*>(base + a_off + swz[KSEG * h + s]);
      fa[s][1] = *reinterpret_cast<const u32x4*>(base + a_off + 4096 + swz[KSEG * h + s]);
      fb[s][0] = *reinterpret_cast<const u32x4*>(base + b_off + swz[KSEG * h + s]);
      fb[s][1] = *reinterpret_cast<const u32x4*>(base + b_off + 4096 + swz[KSEG * h + s]);
    }
  };
  f32x16 acc[2][2];
  auto mfma_seg = [&]() __attribute__((always_inline)) {
    if (SDMI_PP_EXP & 1) return;
    if (!(SDMI_PP_EXP & 64)) __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int s = 0; s < KSEG; ++s)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          // operands swapped (weights as the A operand): acc[i][j] holds C^T -- lane = output row, registers = four
          // groups of four consecutive output columns (epi_rows.h)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fb[s][j]),
                                                              __builtin_bit_cast(bf16x8, fa[s][i]), acc[i][j], 0, 0, 0);
    if (!(SDMI_PP_EXP & 64)) __builtin_amdgcn_s_setprio(0);
  };
#define PP_BARRIER()                         \
  do {                                       \
    __builtin_amdgcn_sched_barrier(0);       \
    __builtin_amdgcn_s_barrier();            \
    asm volatile("" ::: "memory");           \
    __builtin_amdgcn_sched_barrier(0);       \
  } while (0)

  // K tiles 0 and 1 in flight; K tile 0 landed and visible before anybody reads.  Steps past the last K tile re-fetch
  // clamped rows of a non-existent tile into a stage nobody reads any more: the number of DMA pieces in flight stays
  // static, so the fixed vmcnt works to the end.
  issue0(); issue1();
  issue0(); issue1();
  asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  PP_BARRIER();
  dma_on = false;
  int stage = 0;
  PP_TL_DECL
  for (int ti = 0; ti < my_tiles; ++ti) {
    int m0, n0;
    tile_of((int)blockIdx.x + ti * (int)gridDim.x, m0, n0);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    if (grp == 1) PP_BARRIER();              // stagger: group 1 runs one interval behind group 0
    PP_TL_LAP(7);
    for (int t = 0; t < n_kt; ++t) {
      const char* base = smem + stage * STAGE;
      if (++stage == NSTAGE) stage = 0;
#pragma unroll
      for (int h = 0; h < NSEG; ++h) {
        // ---- L(t, h): the segment's fragments; pieces of K tile t + 2 (first / second three when a K tile has two
        // segments); in the LAST segment this wave's pieces of K tile t + 1 must have landed
        read_seg(base, h);
        __builtin_amdgcn_sched_barrier(0);
        if (NSEG == 1 || h == 0) issue0();
        if (NSEG == 1 || h == 1) issue1();
        PP_TL_LAP(0);
        if (h == NSEG - 1) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        PP_TL_LAP(4);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        PP_TL_LAP(1);
        PP_BARRIER();
        PP_TL_LAP(2);
        // ---- M(t, h)
        mfma_seg();
#ifdef SDMI_PP_TIMELINE
        asm volatile("s_nop 0" :: "v"(acc[0][0][0]), "v"(acc[1][1][0]));
#endif
        PP_TL_LAP(3);
        PP_BARRIER();
        PP_TL_LAP(2);
      }
    }
    if (grp == 0) PP_BARRIER();              // both groups store in the same interval
    PP_TL_LAP(5);
    // Row-major epilogue through a wave-private patch of the stage the last K tile vacated (every wave has finished
    // its reads of it: group 0 passed the un-stagger barrier behind group 1's last MFMA segment).  The barrier behind
    // the epilogue keeps the next K tile's DMA pieces (they refill that stage) off the other waves' patches.
    const int mw0 = m0 + wm * 64, nw0 = n0 + wn * 64;
    if (epilogue_rows_ok<2>(p, mw0, nw0, hw_shift)) {
      char* patch = smem + (stage == 0 ? NSTAGE - 1 : stage - 1) * STAGE + w * EPI_ROWS_PATCH;
      wave_epilogue_rows<2>(p, acc, mw0, nw0, hw_shift, l, patch);
    } else {
      wave_epilogue_rows_generic<2>(p, acc, mw0, nw0, hw_shift, l);
      __builtin_amdgcn_s_waitcnt(0x0F70);       // vmcnt(0)
    }
    if (ti + 1 < my_tiles) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      PP_BARRIER();
    }
    PP_TL_LAP(6);
  }
  PP_TL_FLUSH;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // no DMA may outlive the workgroup's LDS
#undef PP_BARRIER
}

template <int MODE, bool XS = false>
int launch_pp(const SdmiGemmArgs& p, int hw_shift, hipStream_t st, int n_cu) {
  constexpr int smem = 3 * (256 + 128) * 128;
  auto kern = igemm_pp_kernel<MODE, XS>;
  SDMI_OPTIN_LDS(kern, smem, "igemm (ping-pong 256x128)");
  SdmiGemmArgs q = p;
  q.split_k = 1;
  const int tiles_m = (p.M + 255) / 256, tiles_n = (p.N + 127) / 128;
  int cap = n_cu < 8 ? 8 : (n_cu & ~7);
  const int nwg = tiles_m * tiles_n;
  hipLaunchKernelGGL(kern, dim3(nwg <= cap ? nwg : cap), dim3(512), smem, st, q, tiles_m, tiles_n, hw_shift);
  return sdmi_check_launch("igemm (ping-pong 256x128)");
}

}  // namespace
