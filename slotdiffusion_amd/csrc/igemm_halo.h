// 3x3 stride-1 convolution as a ping-pong implicit GEMM whose activation operand is staged ONCE per 64-channel chunk
// (round 5).
//
// Ablations of the same ping-pong tile WITHOUT halo staging (igemm_pp, deleted in round 6: DESIGN section 8;
// profiles/r05_pp_ablation.txt; 3x3 256 -> 256 at 32^2, B = 64, us per launch): full kernel 87.7,
// MFMAs + fragment reads without any operand fetch 51.0 (the matrix pipes at the clock the chip sustains under this
// load, ~1.67 GHz), with the weight pieces only 59.2, with the activation pieces only 69.7 -- the activation operand
// of an implicit GEMM (every input pixel fetched again for each of the nine taps: 32 KB per K tile, cold in the
// vector L1) is what a 3x3 convolution waits for.  Here a workgroup's 256 output pixels are TH = 256 / W whole image
// rows, and per 64-channel chunk the (TH + 2) x (W + 2) pixel patch around them -- zero columns left and right, zero
// rows above / below the image -- goes into LDS once (43 KB at W = 32) and serves all nine taps: a tap only shifts the
// fragment address by (kh (W + 2) + kw) patch pixels.  Activation traffic per K tile 32 KB -> 4.8 KB; with the 16 KB
// weight tile 21 KB instead of 48.
//   * structure: 256 x 128 output tile, eight waves as 4 (M) x 2 (N), each a 64 x 64 block = four 32x32 accumulators
//     (16 MFMAs per 16 ds_read_b128 and K tile).  The two waves of a SIMD belong to different GROUPS (waves 0-3 / 4-7)
//     that run the same program one barrier apart: while one group is in its MFMA segment (8 back-to-back
//     v_mfma_f32_32x32x16_bf16 = 256 cycles of the SIMD's matrix pipe) the other is in its LOAD segment (fragment reads
//     for its next half K tile + its DMA pieces), then they swap -- matrix beside memory on every SIMD, four barrier
//     intervals per K tile (step);
//   * hazards by barrier count (interval i = between barrier i and i + 1; group 0 runs L(s,0) M(s,0) L(s,1) M(s,1) in
//     intervals 4s .. 4s+3, group 1 one interval later): the last ds_read of step s is group 1's L(s,1) in interval 4s+3,
//     retired by lgkmcnt(0) before barrier 4s+4; a weight stage is refilled (step s+3's tile) from the L segments of
//     step s+1, interval >= 4s+4.  At an output-tile boundary group 0 idles one interval so that both groups store their
//     accumulators in the same interval, then group 1 idles one to restore the stagger;
//   * LDS: two patch buffers (chunk c + 1 arrives while chunk c is multiplied: wave w issues patch piece 8 t + w
//     during tap t <= 6) + three 16 KB weight stages (the tile of step s + 2 is issued during step s); a wave waits
//     for the weight pieces of step s + 1 at the end of step s with vmcnt(2 or 3) -- patch pieces are older than the
//     weight pieces behind them, so they have landed by then too;
//   * the patch image is pixel-major, 128 B per pixel, 16-byte chunk c of patch pixel q at chunk c ^ ((q >> 1) & 7)
//     (the DMA writes lane-linear: each lane FETCHES the chunk that belongs at its position); a fragment read
//     computes the same key from its shifted pixel index: one add, one bit-field extract, one xor per k-step;
//   * K order (chunk, tap): the weight tile of a step is columns (tap Cin + 64 chunk) of the [N][9 Cin] filter;
//   * accumulators in C^T layout, row-major epilogue through a wave-private patch (epi_rows.h) in the patch buffer of
//     the tile's last chunk.
#pragma once
#include "igemm_body.h"
#include "epi_rows.h"

namespace {

// LOGW: log2 of the tile width = the image width (16 / 32 / 64 columns), or 6 on images of a multiple of 64 columns.  NJ: 32-column blocks per wave -- 2: 256 x 128 tiles (64 x 64
// wave blocks, two segments of 8 MFMAs per K tile); 1: 256 x 64 tiles (64 x 32 wave blocks, one segment of 8 MFMAs per
// K tile) for the layers whose 256 x 128 grid would leave half the chip idle (the 16^2 level at B = 64).
// T: bf16_t.  The fp8_t form (e4m3fn operands: a 128-byte patch row holds 128 channels, two 16-byte k-steps feed one
// v_mfma_scale_f32_32x32x64_f8f6f4 as in igemm_body.h's fp8 path) is written out below and NOT instantiated: with that
// builtin in this kernel hipcc (ROCm 7.2) allocates 256 VGPRs and spills the fragments across the segment barrier
// (1.2 - 1.4 KB of scratch per lane, reloads + vmcnt(0) in the K loop); the same body with the bf16 MFMA in its place
// takes 199 VGPRs and no scratch.  Left for an inline-asm MFMA or a later compiler.
template <int LOGW, int NJ, typename T = bf16_t>
__global__ __launch_bounds__(512, 2) void igemm_halo_kernel(SdmiGemmArgs p, int tiles_m, int tiles_n, int hw_shift) {
  constexpr int ELT = sizeof(T), CPC = 128 / ELT;      // bytes per element, channels per 128-byte chunk
  constexpr int W = 1 << LOGW, TH = 256 / W, PW = W + 2, PH = TH + 2, Q = PH * PW;
  constexpr int NPIECE = (Q + 7) / 8;                  // 1 KB pieces (8 patch pixels) of a chunk's patch
  constexpr int BN = 64 * NJ;
  constexpr int ABUF = NPIECE * 1024, BSTAGE = BN * 128, NBST = 3;
  constexpr int ATAPS = 7;                             // taps during which the next patch is issued (7 x 8 waves >= NPIECE)
  constexpr int KSEG = NJ == 2 ? 2 : 4, NSEG = 4 / KSEG; // k-steps per LOAD / MFMA segment: 8 MFMAs per segment either way
  static_assert(ATAPS * 8 >= NPIECE && ABUF >= 8 * EpiRows<NJ>::PATCH, "patch geometry");
  constexpr unsigned OOB = 0x80000000u;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const Bst = smem + 2 * ABUF;

  const int nwg = tiles_m * tiles_n;
  auto tile_of = [&](int vb, int& m0, int& n0) __attribute__((always_inline)) {
    const int xcd = vb & 7, q = nwg >> 3, r = nwg & 7;
    const int id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (vb >> 3);
    const int tm = id / tiles_n;
    m0 = tm * 256;
    n0 = (id - tm * tiles_n) * BN;
  };
  const int my_tiles = ((int)blockIdx.x < nwg) ? (nwg - 1 - (int)blockIdx.x) / (int)gridDim.x + 1 : 0;
  const int nchunk = p.Cin / CPC;
  if (my_tiles == 0) return;

  const int tid = threadIdx.x, l = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = w >> 2;
  // ------------------------------------ operand fetch state ------------------------------------
  // Tile = TH rows x W columns of one image whose width p.W is W (the 16 / 32 / 64-column levels) or a multiple of it
  // (W = 64 on wider images: the tile's left / right patch columns are then real pixels except at the image border).
  // The activation base is biased by -(p.W + 1) pixels: patch pixel (py, px) of a tile whose first pixel is P0 lies at
  // P0 + py p.W + px in biased coordinates (offsets >= 0; the pixels in front of the tensor are never valid)
  const int tpx = p.W >> LOGW, per_img = (p.H / TH) * tpx;            // tiles per row band / per image
  auto tile_pix0 = [&](int m0, unsigned& mask) __attribute__((always_inline)) {
    const int tm = m0 >> 8, b = tm / per_img, r = tm - b * per_img;
    const int ty = r / tpx, tx = r - ty * tpx;
    mask = (ty == 0 ? 1u : 0u) | ((ty + 1) * TH == p.H ? 2u : 0u) | (tx == 0 ? 4u : 0u) | ((tx + 1) * W == p.W ? 8u : 0u);
    return (b * p.H + ty * TH) * p.W + tx * W;
  };
  const char* Ag = (const char*)p.a - (long long)(p.W + 1) * p.lda * ELT;
  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)Ag, 0, (int)OOB, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, (int)OOB, 0x00020000);
  // patch pieces of this wave: piece 8 t + w (t < ATAPS); lane l: patch pixel q = 8 piece + (l >> 3), chunk position
  // l & 7.  Low four bits of a_vo: 1 / 2 = row above / below the tile, 4 / 8 = column left / right of it (zero when the
  // tile touches that image border); pixels beyond the patch are out of range for good
  unsigned a_vo[ATAPS], a_cur[ATAPS];
#pragma unroll
  for (int t = 0; t < ATAPS; ++t) {
    const int q = (8 * t + w) * 8 + (l >> 3);
    const int py = q / PW, px = q - py * PW;
    const int kc = (l & 7) ^ ((q >> 1) & 7);
    const unsigned f = (py == 0 ? 1u : 0u) | (py == PH - 1 ? 2u : 0u) | (px == 0 ? 4u : 0u) | (px == PW - 1 ? 8u : 0u);
    a_vo[t] = q >= Q ? OOB : (((unsigned)(py * p.W + px) * (unsigned)p.lda * (unsigned)ELT + (unsigned)kc * 16u) | f);
    a_cur[t] = OOB;
  }
  unsigned b_vo[NJ];
  // A side: the chunk being fetched (tile a_ti, chunk a_c); B side: the step being fetched (tile b_ti, chunk b_c, tap b_t)
  int a_ti = 0, a_c = 0, a_buf = 0;
  unsigned a_so = 0;
  auto a_begin_tile = [&]() __attribute__((always_inline)) {
    int m0, n0;
    tile_of((int)blockIdx.x + a_ti * (int)gridDim.x, m0, n0);
    const bool live = a_ti < my_tiles;
    unsigned mask;
    const int pix0 = tile_pix0(live ? m0 : 0, mask);
#pragma unroll
    for (int t = 0; t < ATAPS; ++t) a_cur[t] = (!live || (a_vo[t] & mask)) ? OOB : (a_vo[t] & ~15u);
    a_so = (unsigned)pix0 * (unsigned)p.lda * (unsigned)ELT;
  };
  auto issue_a = [&](int t) __attribute__((always_inline)) {        // piece 8 t + w of chunk (a_ti, a_c) -> buffer a_buf
    if (8 * t + w < NPIECE)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lds_void*)(smem + a_buf * ABUF + (8 * t + w) * 1024), 16, (int)a_cur[t],
                                               (int)(a_so + (unsigned)a_c * 128u), 0, 0);
  };
  auto a_advance = [&]() __attribute__((always_inline)) {           // next chunk to fetch
    a_buf ^= 1;
    if (++a_c == nchunk) {
      a_c = 0;
      ++a_ti;
      a_begin_tile();
    }
  };
  int b_ti = 0, b_c = 0, b_t = 0, b_stage = 0;
  auto b_begin_tile = [&]() __attribute__((always_inline)) {
    int m0, n0;
    tile_of((int)blockIdx.x + b_ti * (int)gridDim.x, m0, n0);
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int row = (w + 8 * j) * 8 + (l >> 3);
      const int kc = (l & 7) ^ ((4 * (w & 1) + (l >> 4)) & 7);
      const int n = min(n0 + row, p.N - 1);
      b_vo[j] = (unsigned)n * (unsigned)p.ldw * (unsigned)ELT + (unsigned)kc * 16u;
    }
  };
  auto issue_b = [&](int j) __attribute__((always_inline)) {        // piece w + 8 j of step (b_ti, b_c, b_t) -> stage b_stage
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, (lds_void*)(Bst + b_stage * BSTAGE + (w + 8 * j) * 1024), 16, (int)b_vo[j],
                                             (int)((unsigned)(b_t * p.Cin + b_c * CPC) * (unsigned)ELT), 0, 0);
  };
  auto b_advance = [&]() __attribute__((always_inline)) {
    if (++b_stage == NBST) b_stage = 0;
    if (++b_t == 9) {
      b_t = 0;
      if (++b_c == nchunk) {
        b_c = 0;
        ++b_ti;
        b_begin_tile();
      }
    }
  };

  // ------------------------------------ MFMA side (per wave) ------------------------------------
  const int wm = w & 3, wn = w >> 2;
  const int R = l & 31, hsel = l >> 5;
  int qb[2];                                           // patch pixel of tap (0, 0) for this lane's row of row block i
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int r = wm * 64 + i * 32 + R;
    qb[i] = (r >> LOGW) * PW + (r & (W - 1));
  }
  int swz[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) swz[ks] = ((2 * ks + hsel) ^ ((R >> 1) & 7)) * 16;
  const int b_off = (wn * 32 * NJ + R) * 128;
  // fragments: bf16 -- one 16-byte k-step per register quad; fp8 -- the two k-steps of a 32x32x64 MFMA side by side in
  // one 8-register tuple, assembled when they are READ (assembling them in front of the MFMAs made the register
  // allocator spill the fragments across the barrier between the two segments)
  constexpr int FV = ELT == 2 ? 1 : 2, NF = KSEG / FV;        // k-steps per fragment tuple, tuples per segment
  typedef unsigned int frag_t __attribute__((ext_vector_type(4 * FV)));
  frag_t fa[NF][2], fb[NF][NJ];                        // [tuple of the segment][row / column block]
  f32x16 acc[2][NJ];
  auto ld_frag = [&](const char* q0, const char* q1) __attribute__((always_inline)) {
    if constexpr (FV == 1) {
      return *reinterpret_cast<const frag_t*>(q0);
    } else {
      const u32x4 lo = *reinterpret_cast<const u32x4*>(q0), hi = *reinterpret_cast<const u32x4*>(q1);
      return frag_t{lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
    }
  };
  auto read_seg = [&](const char* Ab, const char* Bb, int dq, int h) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int q = qb[i] + dq;
      const int key = (q >> 1) & 7;
      const char* row = Ab + q * 128;
#pragma unroll
      for (int f = 0; f < NF; ++f) {
        const int s0 = KSEG * h + f * FV;
        fa[f][i] = ld_frag(row + (((2 * s0 + hsel) ^ key) << 4), row + (((2 * (s0 + FV - 1) + hsel) ^ key) << 4));
      }
    }
#pragma unroll
    for (int f = 0; f < NF; ++f)
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        const int s0 = KSEG * h + f * FV;
        fb[f][j] = ld_frag(Bb + b_off + j * 4096 + swz[s0], Bb + b_off + j * 4096 + swz[s0 + FV - 1]);
      }
  };
  auto mfma_seg = [&]() __attribute__((always_inline)) {
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int f = 0; f < NF; ++f)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
          if constexpr (ELT == 2) {
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fb[f][j]),
                                                                __builtin_bit_cast(bf16x8, fa[f][i]), acc[i][j], 0, 0, 0);
          } else {
            typedef int i32x8 __attribute__((ext_vector_type(8)));
            acc[i][j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(__builtin_bit_cast(i32x8, fb[f][j]),
                                                                        __builtin_bit_cast(i32x8, fa[f][i]), acc[i][j],
                                                                        0, 0, 0, 127, 0, 127);
          }
        }
    __builtin_amdgcn_s_setprio(0);
  };
#define HALO_BARRIER()                       \
  do {                                       \
    __builtin_amdgcn_sched_barrier(0);       \
    __builtin_amdgcn_s_barrier();            \
    asm volatile("" ::: "memory");           \
    __builtin_amdgcn_sched_barrier(0);       \
  } while (0)

  // ---- prologue: the whole patch of (tile 0, chunk 0), the weight tiles of steps 0 and 1
  a_begin_tile();
  b_begin_tile();
#pragma unroll
  for (int t = 0; t < ATAPS; ++t) issue_a(t);
  a_advance();
#pragma unroll
  for (int s = 0; s < 2; ++s) {
#pragma unroll
    for (int j = 0; j < NJ; ++j) issue_b(j);
    b_advance();
  }
  if (NJ == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");     // everything but step 1's weight pieces
  else asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
  HALO_BARRIER();
  int c_buf = 0, c_stage = 0;                          // patch buffer / weight stage of the step being multiplied
  for (int ti = 0; ti < my_tiles; ++ti) {
    int m0, n0;
    tile_of((int)blockIdx.x + ti * (int)gridDim.x, m0, n0);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    if (grp == 1) HALO_BARRIER();              // stagger: group 1 runs one interval behind group 0
    for (int c = 0; c < nchunk; ++c) {
      const char* Ab = smem + c_buf * ABUF;
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        const char* Bb = Bst + c_stage * BSTAGE;
        if (++c_stage == NBST) c_stage = 0;
        // dq is a compile-time constant per unrolled tap: hipcc hoists the fragment addresses of all nine taps out of the
        // chunk loop (256 VGPRs; 17 - 27 spilled registers in the W = 16 / 64 instantiations, none in the K loop).  Hiding
        // the shift from the optimiser (-DSDMI_HALO_NO_HOIST: 199 VGPRs, no scratch, five VALU per tap and segment in the
        // LOAD segments) measured SLOWER: 71.1 vs 67.0 us (256 -> 256 @32^2), 76.2 vs 67.5 (256 -> 256 @16^2, B = 256).
        int dq = (t / 3) * PW + (t % 3);
#ifdef SDMI_HALO_NO_HOIST
        asm volatile("" : "+s"(dq));
#endif
        const bool has_a = t < ATAPS && 8 * t + w < NPIECE;        // wave-uniform
#pragma unroll
        for (int h = 0; h < NSEG; ++h) {
          // ---- L(s, h): the segment's fragments; in the first segment a piece of the next chunk's patch and the first
          // weight piece of step s + 2, in the second the other weight piece.  Behind the LAST segment's issues this
          // wave's weight pieces of step s + 1 (and every patch piece issued before them) must have landed.
          read_seg(Ab, Bb, dq, h);
          __builtin_amdgcn_sched_barrier(0);
          if (h == 0) {
            if (t < ATAPS) issue_a(t);
            issue_b(0);
          }
          if constexpr (NJ == 2) {
            if (h == NSEG - 1) issue_b(1);
          }
          if (h == NSEG - 1) {
            b_advance();
            if (t == ATAPS - 1) a_advance();                          // the next chunk's patch is fully issued
            if (NJ == 2) {
              if (has_a) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
              else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
            } else {
              if (has_a) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
              else asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
            }
          }
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          HALO_BARRIER();
          mfma_seg();
          HALO_BARRIER();
        }
      }
      c_buf ^= 1;
    }
    if (grp == 0) HALO_BARRIER();              // both groups store in the same interval
    // (a wave's 64 rows are consecutive pixels: whole image rows, or one 64-pixel run of a wider image)
    unsigned mask_;
    const int mw0 = tile_pix0(m0, mask_) + ((wm * 64) >> LOGW) * p.W + ((wm * 64) & (W - 1)), nw0 = n0 + wn * 32 * NJ;
    if (epilogue_rows_ok<NJ>(p, mw0, nw0, hw_shift)) {
      // wave-private patch in the patch buffer the tile's last chunk vacated (c_buf already points at the other one)
      wave_epilogue_rows<NJ>(p, acc, mw0, nw0, hw_shift, l, smem + (c_buf ^ 1) * ABUF + w * EpiRows<NJ>::PATCH);
    } else {
      wave_epilogue_rows_generic<NJ>(p, acc, mw0, nw0, hw_shift, l);
      __builtin_amdgcn_s_waitcnt(0x0F70);       // vmcnt(0)
    }
    if (ti + 1 < my_tiles) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      HALO_BARRIER();
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // no DMA may outlive the workgroup's LDS
#undef HALO_BARRIER
}

template <int LOGW, int NJ, typename T = bf16_t>
int launch_halo(const SdmiGemmArgs& p, int hw_shift, hipStream_t st, int n_cu) {
  constexpr int W = 1 << LOGW, Q = (256 / W + 2) * (W + 2);
  constexpr int smem = 2 * ((Q + 7) / 8) * 1024 + 3 * 64 * NJ * 128;
  auto kern = igemm_halo_kernel<LOGW, NJ, T>;
  SDMI_OPTIN_LDS(kern, smem, "igemm (3x3 halo ping-pong)");
  SdmiGemmArgs q = p;
  q.split_k = 1;
  const int tiles_m = p.M / 256, tiles_n = (p.N + 64 * NJ - 1) / (64 * NJ);
  int cap = n_cu < 8 ? 8 : (n_cu & ~7);
  const int nwg = tiles_m * tiles_n;
  hipLaunchKernelGGL(kern, dim3(nwg <= cap ? nwg : cap), dim3(512), smem, st, q, tiles_m, tiles_n, hw_shift);
  return sdmi_check_launch("igemm (3x3 halo ping-pong)");
}

}  // namespace
