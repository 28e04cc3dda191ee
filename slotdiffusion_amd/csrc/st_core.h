// Shared core of the fused SpatialTransformer kernels (st_fused.hip: bf16 inference; st_train.hip: the training
// forward / backward twins): geometry, the per-wave weight ring over LDS-DMA, one K tile of the weights-as-A GEMM
// and the hand-off barrier.  See st_fused.hip for the layout description.
#pragma once
#include "common.h"
#include <type_traits>

namespace {

typedef __attribute__((address_space(3))) char lds_char;
typedef short st_s16x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef short st_s16x8 __attribute__((ext_vector_type(8)));
#define ST_LDS_V4(p) ((__attribute__((address_space(3))) st_s16x4*)(p))

constexpr int ST_UNIT = 2048;          // 16 weight rows x 64 k x 2 B
constexpr int ST_KP = 80, ST_VP = 64;  // attention staging: K row pitch (64 B + pad), V row pitch

__device__ __forceinline__ unsigned st_pack2(float lo, float hi) { return f32x2_to_bf16x2(lo, hi); }

// Epilogue vectors (biases, LayerNorm-fold column sums): lane group lg's four of the 16 floats of a slice.
__device__ __forceinline__ f32x4 st_vec4(const float* base16, int lg) {
  return *reinterpret_cast<const f32x4*>(base16 + 4 * lg);
}

__device__ __forceinline__ int st_xcd_id(int bid, int nwg) {
  const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
}

template <int C, int TT>
struct StGeom {
  static constexpr int ROWS = 16 * TT;          // token rows of a workgroup (TT 16-row tiles: 64, or 32 for small grids)
  static constexpr int NSL = C / 128;          // 16-column slices per wave of an N = C GEMM
  static constexpr int KT = C / 64;            // 64-wide K tiles of a K = C GEMM
  static constexpr int HEADS = C / 32;
  static constexpr int R = HEADS * 8;          // head-expanded slot rows of the folded cross-attention
  static constexpr int RP = 128;               // ... padded (N of the score GEMM, K of the output GEMM)
  static constexpr int NHC = C / 32;           // hidden chunks of 128 (4C hidden units)
  static constexpr int PITCH = C * 2;          // row pitch of the activation operand buffer
  static constexpr int Y_BYTES = ROWS * PITCH;
  static constexpr int G_BYTES = ROWS * 256;   // one GEGLU chunk / the cross-attention probabilities
  static constexpr int FREE2 = 160 * 1024 - Y_BYTES - 2 * G_BYTES;     // what two chunk buffers leave for the rings
  // ring depth in units per wave: as deep as the LDS allows next to two chunk buffers (8 at most), 6 otherwise
  static constexpr int D = FREE2 >= 8 * 8 * ST_UNIT ? 8 : (FREE2 >= 8 * 7 * ST_UNIT ? 7 : 6);
  static constexpr int GBUF = (Y_BYTES + 2 * G_BYTES + 8 * D * ST_UNIT <= 160 * 1024) ? 2 : 1;
  static constexpr int Y_OFF = 0;
  static constexpr int RING_OFF = Y_BYTES + GBUF * G_BYTES;
  static constexpr int SMEM_GEMM = Y_BYTES + GBUF * G_BYTES + 8 * D * ST_UNIT;
  // units per wave
  static constexpr int UA = KT * NSL + 3 * KT * NSL;                       // phase A: proj_in, q, k, v
  static constexpr int UB1 = KT * NSL;                                     // phase B shared, part 1: to_out
  static constexpr int UIMG = KT * 1 + 2 * NSL;                            // per image: scores, cross output
  static constexpr int UB2 = KT * NSL + NHC * (KT * 2 + 2 * NSL);          // part 2: x2 @ Wpo, FF chunks
};

// ---------------------------------------------------------------------------------------------------------
// per-wave weight stream + ring (all state wave-uniform)
// ---------------------------------------------------------------------------------------------------------
template <int D>
struct StRing {
  __amdgpu_buffer_rsrc_t rs_sh, rs_img;
  int g_iss;            // next unit to fetch
  int n1, n_img, total; // shared part 1 | per-image | shared part 2 (unit counts)
  int pos_iss;          // ring position of unit g_iss
  int pos_con;          // ring position of the next unit to consume
  lds_char* ring;       // this wave's ring
  int voff;             // lane * 16
  int jump_at = 1 << 30, jump = 0;   // shared-stream units from `jump_at` on lie `jump` units further (a workgroup that skips
                                     // part of the stream: the feed-forward split of st_fused.hip's phase B)

  __device__ __forceinline__ void issue_one() {
    int g = g_iss < total ? g_iss : total - 1;          // (steps past the end re-fetch the last unit: static vmcnt)
    lds_char* dst = ring + pos_iss * ST_UNIT;
    if (g >= n1 && g < n1 + n_img) {
      const int so = (g - n1) * ST_UNIT;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_img, (lds_void*)dst, 16, voff, so, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_img, (lds_void*)(dst + 1024), 16, voff, so + 1024, 0, 0);
    } else {
      int gs = g >= n1 ? g - n_img : g;
      if (gs >= jump_at) gs += jump;
      const int so = gs * ST_UNIT;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_sh, (lds_void*)dst, 16, voff, so, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_sh, (lds_void*)(dst + 1024), 16, voff, so + 1024, 0, 0);
    }
    ++g_iss;
    if (++pos_iss == D) pos_iss = 0;
  }
};

// EXTRA = global stores of the epilogue in front of this step: they are younger than every DMA in flight and
// vmcnt retires in issue order, so they simply stay outstanding on top of the D - n units (a vmcnt(0) here waited
// for the stores' round trip: ~2 k cycles per q / k / v pass of phase A).  -DST_DRAIN: vmcnt(0) everywhere (experiment).
#if defined(ST_DRAIN)
#define ST_WAIT_UNITS(D_, n_, extra_) asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
#else
#define ST_WAIT_UNITS(D_, n_, extra_) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * ((D_) - (n_)) + (extra_)) : "memory")
#endif
// Workgroup barrier that leaves the DMA queue alone: `__syncthreads()` carries a fence that waits vmcnt(0) while
// LDS-DMA is in flight (it is a pending LDS write); LDS stores / reads of this wave are retired explicitly.
#ifdef ST_TIMELINE      // experiment: s_memtime of wave 0 at the phase boundaries of phase B -> p.gn_gamma[wg][8] (u64)
#define ST_STAMP(i) do { if (threadIdx.x == 0) ((unsigned long long*)p.gn_gamma)[blockIdx.x * 16 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define ST_STAMP(i)
#endif
#ifdef ST_TIMELINE
#define ST_TL_DECL unsigned long long tl_t = 0, tl_a[4] = {0, 0, 0, 0};
#define ST_TL_BEGIN tl_t = __builtin_amdgcn_s_memtime();
#define ST_TL_LAP(i) do { const unsigned long long n_ = __builtin_amdgcn_s_memtime(); tl_a[i] += n_ - tl_t; tl_t = n_; } while (0)
#define ST_TL_FLUSH do { if (threadIdx.x == 0) for (int i_ = 0; i_ < 4; ++i_) ((unsigned long long*)p.gn_gamma)[blockIdx.x * 16 + 12 + i_] = tl_a[i_]; } while (0)
#else
#define ST_TL_DECL
#define ST_TL_BEGIN
#define ST_TL_LAP(i)
#define ST_TL_FLUSH
#endif
#ifdef ST_TIMELINE
#define STA_STAMP(i) do { if (threadIdx.x == 0) ((unsigned long long*)p.vec_img)[blockIdx.x * 8 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define STA_STAMP(i)
#endif
#define ST_BARRIER()                                         \
  do {                                                       \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       \
    __builtin_amdgcn_s_barrier();                            \
    asm volatile("" ::: "memory");                           \
  } while (0)

// One K tile (64 k = two MFMA k-steps) of a GEMM whose wave block is NU 16-column slices x 64 rows:
// acc[s][tt] += W_unit(s) . act[rows 16 tt ..][k tile kt].  STATS: LayerNorm-fold row sums from the
// activation fragments (lane: row lane & 15 of tile tt, its 8 k of each 32).
template <int D, int NU, bool STATS, int EXTRA = 0, int TT = 4>
__device__ __forceinline__ void st_gemm_step(StRing<D>& rg, const lds_char* act, const int (&yaddr)[4], int kt,
                                             int tt_stride, const int (&woff)[2], f32x4 (&acc)[NU][TT],
                                             float (&sx)[TT], float (&sxx)[TT]) {
#ifdef ST_STEP_BARRIER       // experiment: the eight waves in lock step (the default lets them run free)
  __builtin_amdgcn_s_barrier();
#endif
  ST_WAIT_UNITS(D, NU, EXTRA);
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    const int kk = kt * 2 + ks;
    bf16x8 b[TT];
#pragma unroll
    for (int tt = 0; tt < TT; ++tt)
      b[tt] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const __attribute__((address_space(3))) u32x4*>(
                                             act + yaddr[kk & 3] + tt * tt_stride + (kk >> 2) * 256));
    if constexpr (STATS) {
      const sdmi_bf16x2 ones = __builtin_bit_cast(sdmi_bf16x2, 0x3F803F80u);
#pragma unroll
      for (int tt = 0; tt < TT; ++tt) {
        const sdmi_bf16x2 v0 = {b[tt][0], b[tt][1]}, v1 = {b[tt][2], b[tt][3]}, v2 = {b[tt][4], b[tt][5]},
                          v3 = {b[tt][6], b[tt][7]};
        sx[tt] = __builtin_amdgcn_fdot2_f32_bf16(v0, ones, sx[tt], false);
        sxx[tt] = __builtin_amdgcn_fdot2_f32_bf16(v0, v0, sxx[tt], false);
        sx[tt] = __builtin_amdgcn_fdot2_f32_bf16(v1, ones, sx[tt], false);
        sxx[tt] = __builtin_amdgcn_fdot2_f32_bf16(v1, v1, sxx[tt], false);
        sx[tt] = __builtin_amdgcn_fdot2_f32_bf16(v2, ones, sx[tt], false);
        sxx[tt] = __builtin_amdgcn_fdot2_f32_bf16(v2, v2, sxx[tt], false);
        sx[tt] = __builtin_amdgcn_fdot2_f32_bf16(v3, ones, sx[tt], false);
        sxx[tt] = __builtin_amdgcn_fdot2_f32_bf16(v3, v3, sxx[tt], false);
      }
    }
#pragma unroll
    for (int s = 0; s < NU; ++s) {
      int pos = rg.pos_con + s;
      if (pos >= D) pos -= D;
      const bf16x8 a = __builtin_bit_cast(
          bf16x8, *reinterpret_cast<const __attribute__((address_space(3))) u32x4*>(rg.ring + pos * ST_UNIT + woff[ks]));
#pragma unroll
      for (int tt = 0; tt < TT; ++tt) acc[s][tt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b[tt], acc[s][tt], 0, 0, 0);
    }
  }
  // The units just multiplied are free: refill their ring positions -- but only once this step's fragment reads have
  // RETURNED.  To the instruction scheduler a `buffer_load ... lds` is a load, not an LDS store: without the wait and
  // the scheduling barrier it moves the refill in front of the ds_reads of the very slot it overwrites, and when the
  // stream hits in L2 the DMA wins the race now and then: with free-running waves 2 - 6 % of the 16-column output
  // slices were wrong (exactly one 1 KB piece of one K tile each; tools/exp/st_forensic.py, st_stress.py), in lock step
  // still ~1e-3.  Neither vmcnt(0) per step, a barrier behind the wait, nor waiting a step ahead cured it; this does:
  // 0 / 102400 slices over 40 runs, bitwise repeatable.
#ifndef ST_UNSAFE_REFILL
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
#endif
  rg.pos_con += NU;
  if (rg.pos_con >= D) rg.pos_con -= D;
#pragma unroll
  for (int s = 0; s < NU; ++s) rg.issue_one();
}

// row statistics of the LayerNorm fold: after all K tiles, fold the four 8-k lane groups of a row
template <int TT>
__device__ __forceinline__ void st_ln_stats(float (&sx)[TT], float (&sxx)[TT], float inv_k, float eps, float (&mean)[TT],
                                            float (&rstd)[TT]) {
#pragma unroll
  for (int tt = 0; tt < TT; ++tt) {
    float a = sx[tt], b = sxx[tt];
    a += __shfl_xor(a, 16, 64);
    b += __shfl_xor(b, 16, 64);
    a += __shfl_xor(a, 32, 64);
    b += __shfl_xor(b, 32, 64);
    mean[tt] = a * inv_k;
    rstd[tt] = rsqrtf(fmaxf(b * inv_k - mean[tt] * mean[tt], 0.f) + eps);
  }
}

}  // namespace
