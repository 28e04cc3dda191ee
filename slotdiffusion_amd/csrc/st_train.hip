// Training twin of the fused SpatialTransformer block (sdmi.h: sdmi_st_train_fwd / sdmi_st_train_bwd, sdmi_st_pack).
//
// Reference: video_based/models/unet/attention.py:297-308 (SpatialTransformer), 247-251 (BasicTransformerBlock),
// 182-206 (CrossAttention), 44-65 (GEGLU / FeedForward).
//
// Same skeleton as st_fused.hip (a workgroup owns 64 / 32 token rows of one image, eight symmetric waves, each streaming
// the weight rows it multiplies through a private ring of LDS-DMA units, weights as the MFMA A operand so that a lane
// holds four consecutive output columns of one token row) -- but in the form a TRAINING step needs:
//   * plain weights (the weights change every step: no LayerNorm fold, no ff.net.2 / proj_out merge), re-packed into
//     unit streams by st_pack_kernel after every optimiser step;
//   * LayerNorm explicitly on the fp32 residual stream (row statistics exchanged between the waves through LDS): the
//     normalised rows are written out because the weight gradients of the layers behind them multiply them;
//   * slot cross-attention explicit (q projection, per-head softmax over the slots, output projection) -- the slot
//     keys / values receive gradients;
//   * every tensor the backward pass reads is stored on the way (sdmi.h lists them).
// Forward: phase A (GroupNorm -> proj_in -> LN1 -> q | k | v), phase B (self-attention -> to_out -> LN2 -> cross
// attention -> LN3 -> GEGLU feed-forward -> proj_out).
#include "st_core.h"

namespace {

template <int C, int TT, int KVS>
struct StTrGeom {
  static constexpr int ROWS = 16 * TT, NSL = C / 128, KT = C / 64, HEADS = C / 32, NHC = C / 32, PITCH = C * 2;
  static constexpr int Y_BYTES = ROWS * PITCH;
  static constexpr int G_BYTES = ROWS * 256;               // one GEGLU chunk (128 hidden units)
  static constexpr int RED_BYTES = 8 * ROWS * 8;           // LayerNorm partials: [wave][row] (sum, sum of squares)
  static constexpr int KV_BYTES = KVS * 2 * C * 2;         // slot keys | values of the image
  static constexpr int FIX1 = Y_BYTES + G_BYTES + RED_BYTES + KV_BYTES;
  static constexpr int GBUF = (FIX1 + G_BYTES + 8 * 6 * ST_UNIT <= 160 * 1024) ? 2 : 1;
  static constexpr int FIXED = FIX1 + (GBUF - 1) * G_BYTES;
  static constexpr int DFREE = (160 * 1024 - FIXED) / (8 * ST_UNIT);
  static constexpr int D = DFREE >= 8 ? 8 : DFREE;          // ring depth (units per wave)
  static_assert(D >= 4 && D >= NSL + 1, "ring too shallow");
  static constexpr int Y_OFF = 0, G_OFF = Y_BYTES, RED_OFF = G_OFF + GBUF * G_BYTES, KV_OFF = RED_OFF + RED_BYTES,
                       RING_OFF = KV_OFF + KV_BYTES;
  static constexpr int SMEM_B = RING_OFF + 8 * D * ST_UNIT;
  // phase A: operand buffer | GroupNorm partials (8 KB + statistics) | LayerNorm partials | rings
  static constexpr int GNRED_BYTES = 9 * 1024;
  static constexpr int A_GN_OFF = Y_BYTES, A_RED_OFF = A_GN_OFF + GNRED_BYTES, A_RING_OFF = A_RED_OFF + RED_BYTES;
  static constexpr int DA_FREE = (160 * 1024 - A_RING_OFF) / (8 * ST_UNIT);
  static constexpr int DA = DA_FREE >= 8 ? 8 : DA_FREE;
  static constexpr int SMEM_A = A_RING_OFF + 8 * DA * ST_UNIT;
  static constexpr int UA = 4 * KT * NSL;                                   // proj_in, q, k, v
  static constexpr int UB = 4 * KT * NSL + NHC * (2 * KT + 2 * NSL);        // to_out, q2, to_out2, FF chunks, proj_out
};

// Cross-lane sums without the LDS path (ds_bpermute): v_permlane16_swap / v_permlane32_swap exchange 16- / 32-lane halves
// in the VALU, DPP row rotations sum the 16 lanes of a row (no LDS round trip next to the weight DMAs).
//
// BUILD NOTE: this file is compiled with -fno-slp-vectorize (csrc/build.py).  With the SLP vectoriser on, the LayerNorm
// backward below is emitted with packed fp32 VALU ops (v_pk_add_f32 / v_pk_mul_f32 with op_sel / neg modifiers behind
// v_lshlrev_b32 unpacks), and on MI355X those were NOT repeatable run to run: in ~1 % of the workgroups the LOW half of a
// packed result came out wrong in lanes 48 - 63 of a wave (the raw loaded words, the MFMA accumulators and every scalar
// form of the same arithmetic were bit-stable; tools/exp/st_bwd_repeat.py: 39 of 39 runs differed at B = 64 with the
// packed ops, 0 of 351 without, same source).  Looks like a packed-fp32 hazard next to a draining MFMA pipeline that the
// compiler does not cover; the scalar forms cost nothing measurable here.
__device__ __forceinline__ float st_sum_x16(float a) {        // a(lane) + a(lane ^ 16), identical in both lanes
  const unsigned i = __float_as_uint(a);
  const auto r = __builtin_amdgcn_permlane16_swap(i, i, false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float st_sum_x32(float a) {        // a(lane) + a(lane ^ 32)
  const unsigned i = __float_as_uint(a);
  const auto r = __builtin_amdgcn_permlane32_swap(i, i, false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float st_sum_row16(float v) {      // sum over the 16 lanes of a DPP row, in every lane
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x128, 0xf, 0xf, false));   // row_ror:8
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x124, 0xf, 0xf, false));   // row_ror:4
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x122, 0xf, 0xf, false));   // row_ror:2
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x121, 0xf, 0xf, false));   // row_ror:1
  return v;
}

template <int D>
__device__ __forceinline__ void st_ring_init(StRing<D>& rg, const void* stream, int units, lds_char* ring, int lane, int w) {
  rg.rs_sh = __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)stream + (long long)w * units * ST_UNIT), 0,
                                               units * ST_UNIT, 0x00020000);
  rg.rs_img = rg.rs_sh;
  rg.g_iss = 0; rg.n1 = units; rg.n_img = 0; rg.total = units; rg.pos_iss = 0; rg.pos_con = 0;
  rg.ring = ring + w * D * ST_UNIT;
  rg.voff = lane * 16;
#pragma unroll
  for (int i = 0; i < D; ++i) rg.issue_one();
}

// LayerNorm of the fp32 rows held across the workgroup (lane: row 16 tt + l15, columns (w NSL + s) 16 + 4 lg + j):
// row sums exchanged through `red` ([wave][row] float2), then n = (x - mean) rstd gamma + beta as bf16 into the operand
// buffer Y and to `nout`; (mean, rstd) to `stout` (wave 0).  Two workgroup barriers: the first also tells that every
// wave is done reading Y as the previous GEMM's operand.
template <int C, int TT, int NSL>
__device__ __forceinline__ void st_layernorm_rows(const f32x4 (&x)[NSL][TT], const f32x4 (&gm)[NSL], const f32x4 (&bt)[NSL],
                                                  float eps, lds_char* Y, float* red, bf16_t* nout, float* stout, int w,
                                                  int l15, int lg) {
  constexpr int ROWS = 16 * TT, PITCH = C * 2;
#pragma unroll
  for (int tt = 0; tt < TT; ++tt) {
    float a = 0.f, b = 0.f;
#pragma unroll
    for (int s = 0; s < NSL; ++s)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        a += x[s][tt][j];
        b += x[s][tt][j] * x[s][tt][j];
      }
    a = st_sum_x32(st_sum_x16(a));
    b = st_sum_x32(st_sum_x16(b));
    if (lg == 0) *reinterpret_cast<float2*>(red + (w * ROWS + tt * 16 + l15) * 2) = make_float2(a, b);
  }
  ST_BARRIER();
#pragma unroll
  for (int tt = 0; tt < TT; ++tt) {
    const int r = tt * 16 + l15;
    float a = 0.f, b = 0.f;
#pragma unroll
    for (int ww = 0; ww < 8; ++ww) {
      const float2 v = *reinterpret_cast<const float2*>(red + (ww * ROWS + r) * 2);
      a += v.x;
      b += v.y;
    }
    const float mean = a * (1.f / (float)C);
    const float rstd = rsqrtf(fmaxf(b * (1.f / (float)C) - mean * mean, 0.f) + eps);
    if (w == 0 && lg == 0) *reinterpret_cast<float2*>(stout + (long long)r * 2) = make_float2(mean, rstd);
#pragma unroll
    for (int s = 0; s < NSL; ++s) {
      const int n0 = (w * NSL + s) * 16 + 4 * lg;
      const int c = n0 >> 3;
      const int phys = (c & ~15) | ((c ^ r) & 15);
      uint2 o;
      o.x = st_pack2((x[s][tt][0] - mean) * rstd * gm[s][0] + bt[s][0], (x[s][tt][1] - mean) * rstd * gm[s][1] + bt[s][1]);
      o.y = st_pack2((x[s][tt][2] - mean) * rstd * gm[s][2] + bt[s][2], (x[s][tt][3] - mean) * rstd * gm[s][3] + bt[s][3]);
      *reinterpret_cast<__attribute__((address_space(3))) u32x2*>(Y + r * PITCH + phys * 16 + (lg & 1) * 8) = u32x2{o.x, o.y};
      *reinterpret_cast<uint2*>(nout + (long long)r * C + n0) = o;
    }
  }
  ST_BARRIER();
}

// bf16 rows held across the workgroup (fp32 values x) -> operand buffer Y (+ optional global copy)
template <int C, int TT, int NSL, bool GLOBAL>
__device__ __forceinline__ void st_rows_to_y(const f32x4 (&x)[NSL][TT], lds_char* Y, bf16_t* gout, int w, int l15, int lg) {
  constexpr int PITCH = C * 2;
#pragma unroll
  for (int s = 0; s < NSL; ++s) {
    const int n0 = (w * NSL + s) * 16 + 4 * lg;
    const int c = n0 >> 3;
#pragma unroll
    for (int tt = 0; tt < TT; ++tt) {
      const int r = tt * 16 + l15;
      const int phys = (c & ~15) | ((c ^ r) & 15);
      uint2 o;
      o.x = st_pack2(x[s][tt][0], x[s][tt][1]);
      o.y = st_pack2(x[s][tt][2], x[s][tt][3]);
      *reinterpret_cast<__attribute__((address_space(3))) u32x2*>(Y + r * PITCH + phys * 16 + (lg & 1) * 8) = u32x2{o.x, o.y};
      if constexpr (GLOBAL) *reinterpret_cast<uint2*>(gout + (long long)r * C + n0) = o;
    }
  }
}

template <int C, int TT, int NSL>
__device__ __forceinline__ void st_rows_store(const f32x4 (&x)[NSL][TT], bf16_t* gout, int ld, int w, int l15, int lg) {
#pragma unroll
  for (int s = 0; s < NSL; ++s) {
    const int n0 = (w * NSL + s) * 16 + 4 * lg;
#pragma unroll
    for (int tt = 0; tt < TT; ++tt) {
      uint2 o;
      o.x = st_pack2(x[s][tt][0], x[s][tt][1]);
      o.y = st_pack2(x[s][tt][2], x[s][tt][3]);
      *reinterpret_cast<uint2*>(gout + (long long)(tt * 16 + l15) * ld + n0) = o;
    }
  }
}

// ---------------------------------------------------------------------------------------------------------
// forward phase A: GroupNorm -> hgn -> proj_in -> tok ; LayerNorm1 -> n1 ; q | k | v
// ---------------------------------------------------------------------------------------------------------
template <int C, int TT, int KVS>
__global__ __launch_bounds__(512) void st_train_a_kernel(SdmiStTrainArgs p) {
  typedef StTrGeom<C, TT, KVS> G;
  constexpr int ROWS = G::ROWS, NSL = G::NSL, KT = G::KT, D = G::DA, PITCH = G::PITCH;
  extern __shared__ __attribute__((aligned(16))) char smem_[];
  lds_char* const smem = (lds_char*)smem_;
  lds_char* const Y = smem;
  float* const red = (float*)(smem_ + G::A_GN_OFF);
  float* const lnred = (float*)(smem_ + G::A_RED_OFF);
  const int tid = threadIdx.x, lane = tid & 63, l15 = lane & 15, lg = lane >> 4;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wgs_per_img = p.S / ROWS;
  const int vid = st_xcd_id((int)blockIdx.x, (int)gridDim.x);
  const int b = vid / wgs_per_img, rb = vid - b * wgs_per_img;
  const long long row0 = (long long)b * p.S + rb * ROWS;

  StRing<D> rg;
  st_ring_init<D>(rg, p.wstream_a, G::UA, smem + G::A_RING_OFF, lane, w);      // weights in flight under the GroupNorm

  // ---- GroupNorm statistics of the image (32 groups), recomputed by each of its workgroups (L2 resident)
  constexpr int VPR = C / 8, RPP = 512 / VPR, GS4 = C / 32 / 4;
  {
    const bf16_t* xi = (const bf16_t*)p.x + (long long)b * p.S * C;
    float s0 = 0.f, s1 = 0.f, q0 = 0.f, q1 = 0.f;
    const int vr = tid / VPR, vc = tid - vr * VPR;
    if (vr < RPP) {
      for (int r0 = vr; r0 < p.S; r0 += 8 * RPP) {
        uint4 v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int r = r0 + j * RPP;
          v[j] = r < p.S ? *reinterpret_cast<const uint4*>(xi + (long long)r * C + vc * 8) : make_uint4(0u, 0u, 0u, 0u);
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          float f[8];
          unpack16<bf16_t>(v[j], f);
          s0 += (f[0] + f[1]) + (f[2] + f[3]);
          s1 += (f[4] + f[5]) + (f[6] + f[7]);
          q0 += f[0] * f[0] + f[1] * f[1] + f[2] * f[2] + f[3] * f[3];
          q1 += f[4] * f[4] + f[5] * f[5] + f[6] * f[6] + f[7] * f[7];
        }
      }
      f32x4 v = {s0, q0, s1, q1};
      *reinterpret_cast<f32x4*>(red + (vr * VPR + vc) * 4) = v;
    }
    ST_BARRIER();
    if (tid < 32) {
      float s = 0.f, q = 0.f;
      for (int r = 0; r < RPP; ++r)
        for (int h = 0; h < GS4; ++h) {
          const int hv = tid * GS4 + h;
          const float* e = red + ((r * VPR + (hv >> 1)) * 4 + (hv & 1) * 2);
          s += e[0];
          q += e[1];
        }
      const float n = (float)(p.S * (C / 32));
      const float mean = s / n;
      const float rstd = rsqrtf(fmaxf(q / n - mean * mean, 0.f) + p.gn_eps);
      red[512 * 4 + tid * 2] = mean;
      red[512 * 4 + tid * 2 + 1] = rstd;
      if (rb == 0) *reinterpret_cast<float2*>(p.gn_stats + ((long long)b * 32 + tid) * 2) = make_float2(mean, rstd);
    }
    ST_BARRIER();
  }
  // ---- normalise this workgroup's rows into the operand buffer and to hgn
  {
    const float* st = red + 512 * 4;
    const bf16_t* xr = (const bf16_t*)p.x + row0 * C;
    bf16_t* ho = (bf16_t*)p.hgn + row0 * C;
    static_assert((ROWS * VPR) % 512 == 0, "whole passes");
#pragma unroll
    for (int it = 0; it < ROWS * VPR / 512; ++it) {
      const int i = tid + it * 512;
      const int r = i / VPR, vc = i - r * VPR;
      float f[8], gm[8], bt[8];
      unpack16<bf16_t>(*reinterpret_cast<const uint4*>(xr + (long long)r * C + vc * 8), f);
      *reinterpret_cast<f32x4*>(gm) = *reinterpret_cast<const f32x4*>(p.gn_gamma + vc * 8);
      *reinterpret_cast<f32x4*>(gm + 4) = *reinterpret_cast<const f32x4*>(p.gn_gamma + vc * 8 + 4);
      *reinterpret_cast<f32x4*>(bt) = *reinterpret_cast<const f32x4*>(p.gn_beta + vc * 8);
      *reinterpret_cast<f32x4*>(bt + 4) = *reinterpret_cast<const f32x4*>(p.gn_beta + vc * 8 + 4);
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int g = (vc * 8 + h * 4) / (C / 32);
        const float mean = st[g * 2], rstd = st[g * 2 + 1];
#pragma unroll
        for (int j = 0; j < 4; ++j) f[h * 4 + j] = (f[h * 4 + j] - mean) * rstd * gm[h * 4 + j] + bt[h * 4 + j];
      }
      const int phys = (vc & ~15) | ((vc ^ r) & 15);
      const uint4 pk = pack16<bf16_t>(f);
      *reinterpret_cast<__attribute__((address_space(3))) u32x4*>(Y + r * PITCH + phys * 16) = u32x4{pk.x, pk.y, pk.z, pk.w};
      *reinterpret_cast<uint4*>(ho + (long long)r * C + vc * 8) = pk;
    }
  }
  ST_BARRIER();

  int yaddr[4], woff[2];
#pragma unroll
  for (int j = 0; j < 4; ++j) yaddr[j] = l15 * PITCH + ((((4 * j + lg) ^ l15) & 15) * 16);
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) woff[ks] = l15 * 128 + (((4 * ks + lg) ^ ((l15 >> 1) & 7)) * 16);
  float sx[TT], sxx[TT];
#pragma unroll
  for (int tt = 0; tt < TT; ++tt) sx[tt] = sxx[tt] = 0.f;

  // ---- proj_in: tok = hgn Win^T + bin   (epilogue vectors are fetched BEFORE the GEMM: st_fused.hip's note)
  f32x4 acc[NSL][TT], ev0[NSL], evg[NSL], evb[NSL];
#pragma unroll
  for (int s = 0; s < NSL; ++s) {
    ev0[s] = st_vec4(p.b_in + (w * NSL + s) * 16, lg);
    evg[s] = st_vec4(p.ln1_g + (w * NSL + s) * 16, lg);
    evb[s] = st_vec4(p.ln1_b + (w * NSL + s) * 16, lg);
#pragma unroll
    for (int tt = 0; tt < TT; ++tt) acc[s][tt] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
#pragma unroll
  for (int kt = 0; kt < KT; ++kt) st_gemm_step<D, NSL, false, 0, TT>(rg, Y, yaddr, kt, 16 * PITCH, woff, acc, sx, sxx);
#pragma unroll
  for (int s = 0; s < NSL; ++s)
#pragma unroll
    for (int tt = 0; tt < TT; ++tt)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[s][tt][j] += ev0[s][j];
  st_rows_store<C, TT, NSL>(acc, (bf16_t*)p.tok + row0 * C, C, w, l15, lg);
  // ---- LayerNorm1 -> n1 (operand buffer + global), statistics
  st_layernorm_rows<C, TT, NSL>(acc, evg, evb, p.ln_eps, Y, lnred, (bf16_t*)p.n1 + row0 * C, p.st1 + row0 * 2, w, l15, lg);

  // ---- q | k | v = n1 W^T: three passes of N = C over the same operand
#pragma unroll 1
  for (int pass = 0; pass < 3; ++pass) {
#pragma unroll
    for (int s = 0; s < NSL; ++s)
#pragma unroll
      for (int tt = 0; tt < TT; ++tt) acc[s][tt] = f32x4{0.f, 0.f, 0.f, 0.f};
    st_gemm_step<D, NSL, false, NSL * TT, TT>(rg, Y, yaddr, 0, 16 * PITCH, woff, acc, sx, sxx);   // (EXTRA: the stores / loads just issued stay outstanding)
#pragma unroll
    for (int kt = 1; kt < KT; ++kt) st_gemm_step<D, NSL, false, 0, TT>(rg, Y, yaddr, kt, 16 * PITCH, woff, acc, sx, sxx);
    st_rows_store<C, TT, NSL>(acc, (bf16_t*)p.qkv + row0 * 3 * C + pass * C, 3 * C, w, l15, lg);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // no DMA may outlive the workgroup's LDS
}

// ---------------------------------------------------------------------------------------------------------
// forward phase B
// ---------------------------------------------------------------------------------------------------------
template <int C, int TT, int KVS>
__global__ __launch_bounds__(512) void st_train_b_kernel(SdmiStTrainArgs p) {
  typedef StTrGeom<C, TT, KVS> G;
  constexpr int ROWS = G::ROWS, NSL = G::NSL, KT = G::KT, D = G::D, PITCH = G::PITCH, HEADS = G::HEADS;
  extern __shared__ __attribute__((aligned(16))) char smem_[];
  lds_char* const smem = (lds_char*)smem_;
  lds_char* const Y = smem + G::Y_OFF;
  lds_char* const Gb = smem + G::G_OFF;
  float* const lnred = (float*)(smem_ + G::RED_OFF);
  lds_char* const KV = smem + G::KV_OFF;
  const int tid = threadIdx.x, lane = tid & 63, l15 = lane & 15, lg = lane >> 4;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wgs_per_img = p.S / ROWS;
  const int vid = st_xcd_id((int)blockIdx.x, (int)gridDim.x);
  const int b = vid / wgs_per_img, rb = vid - b * wgs_per_img;
  const long long row0 = (long long)b * p.S + rb * ROWS;
  const int S = p.S;

  // =========================== self-attention: ROWS queries x HEADS heads over S keys ===========================
  // (st_fused.hip's phase B; additionally lse1 for the backward pass)
  unsigned opack[HEADS / 4][8];
  {
    const int hs = TT == 4 ? (w >> 1) : (w & 3), qh = TT == 4 ? (w & 1) : 0;
    const bool att_active = TT == 4 || w < 4;
    const int ql = lane & 31, hh = lane >> 5;
    const bf16_t* qkv_img = (const bf16_t*)p.qkv + (long long)b * S * 3 * C;
    const int head_bytes = S * (ST_KP + ST_VP);
    const float sc2 = p.attn_scale * 1.4426950408889634f;
    const int g4 = lane >> 4, t16 = lane & 15;
    const bool all_heads = HEADS * head_bytes <= 160 * 1024;
    const int hb = all_heads ? HEADS : 4;
    bf16x8 bq[HEADS / 4][2];
#pragma unroll
    for (int ri = 0; ri < HEADS / 4; ++ri) {
      const bf16_t* qp = (const bf16_t*)p.qkv + (row0 + qh * 32 + ql) * 3 * C + (ri * 4 + hs) * 32 + hh * 8;
      bq[ri][0] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(qp));
      bq[ri][1] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(qp + 16));
    }
    auto stage = [&](int h0, auto nb_) __attribute__((always_inline)) {
      constexpr int NB = decltype(nb_)::value;
      const int ppr = hb * 8;
      for (int i0 = tid; i0 < S * ppr; i0 += NB * 512) {
        u32x4 v[NB];
        int dsto[NB];
#pragma unroll
        for (int j = 0; j < NB; ++j) {
          const int i = i0 + j * 512, row = i / ppr, rem = i - row * ppr;
          const int isv = rem >= hb * 4, r2 = rem - isv * hb * 4, hl = r2 >> 2, c = r2 & 3;
          v[j] = *reinterpret_cast<const u32x4*>(qkv_img + (long long)row * 3 * C + (1 + isv) * C + (h0 + hl) * 32 + c * 8);
          dsto[j] = hl * head_bytes + (isv ? S * ST_KP + row * ST_VP : row * ST_KP) + c * 16;
        }
#pragma unroll
        for (int j = 0; j < NB; ++j) *reinterpret_cast<__attribute__((address_space(3))) u32x4*>(smem + dsto[j]) = v[j];
      }
    };
#pragma unroll
    for (int ri = 0; ri < HEADS / 4; ++ri) {
      if (ri == 0 || !all_heads) {
        if (ri) __syncthreads();
        const int per_thread = S * hb / 64;
        if (per_thread % 16 == 0) stage(ri * 4, std::integral_constant<int, 16>());
        else if (per_thread % 12 == 0) stage(ri * 4, std::integral_constant<int, 12>());
        else stage(ri * 4, std::integral_constant<int, 4>());
        __syncthreads();
      }
      const int hl = (all_heads ? ri * 4 : 0) + hs;
      const lds_char* Ks = smem + hl * head_bytes;
      const lds_char* Vs = Ks + S * ST_KP;
      const lds_char* kfrag = Ks + ql * ST_KP + hh * 16;
      const lds_char* vfrag = Vs + (4 * hh + (t16 >> 2)) * ST_VP + ((g4 & 1) * 16 + (t16 & 3) * 4) * 2;
      if (!att_active) continue;
      f32x16 o;
#pragma unroll
      for (int r = 0; r < 16; ++r) o[r] = 0.f;
      float m = -INFINITY, lsum = 0.f;
      for (int kb = 0; kb < S / 32; ++kb) {
        f32x16 s;
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          const u32x4 a = *reinterpret_cast<const __attribute__((address_space(3))) u32x4*>(kfrag + kb * 32 * ST_KP + ks * 32);
          s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), bq[ri][ks], s, 0, 0, 0);
        }
        float bmax = -INFINITY;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          s[r] *= sc2;
          bmax = fmaxf(bmax, s[r]);
        }
        bmax = fmaxf(bmax, __shfl_xor(bmax, 32, 64));
        const float m_new = fmaxf(m, bmax);
        float psum = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          s[r] = __builtin_amdgcn_exp2f(s[r] - m_new);
          psum += s[r];
        }
        psum += __shfl_xor(psum, 32, 64);
        if (__builtin_amdgcn_ballot_w64(m_new > m) != 0) {
          const float alpha = __builtin_amdgcn_exp2f(m - m_new);
          lsum = lsum * alpha + psum;
#pragma unroll
          for (int r = 0; r < 16; ++r) o[r] *= alpha;
        } else {
          lsum += psum;
        }
        m = m_new;
#pragma unroll
        for (int mm = 0; mm < 2; ++mm) {
          const u32x4 pb = {st_pack2(s[8 * mm + 0], s[8 * mm + 1]), st_pack2(s[8 * mm + 2], s[8 * mm + 3]),
                            st_pack2(s[8 * mm + 4], s[8 * mm + 5]), st_pack2(s[8 * mm + 6], s[8 * mm + 7])};
          const lds_char* vp = vfrag + (kb * 32 + 16 * mm) * ST_VP;
          const st_s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(ST_LDS_V4(vp));
          const st_s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(ST_LDS_V4(vp + 8 * ST_VP));
          const st_s16x8 av = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
          o = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, av), __builtin_bit_cast(bf16x8, pb), o, 0, 0, 0);
        }
      }
      const float inv = 1.f / lsum;
#pragma unroll
      for (int j = 0; j < 4; ++j) {       // d = 8 j + 4 hh + (0..3)
        opack[ri][2 * j] = st_pack2(o[4 * j] * inv, o[4 * j + 1] * inv);
        opack[ri][2 * j + 1] = st_pack2(o[4 * j + 2] * inv, o[4 * j + 3] * inv);
      }
      if (hh == 0)
        p.lse1[((long long)b * HEADS + (ri * 4 + hs)) * S + rb * ROWS + qh * 32 + ql] = m * 0.6931471805599453f + __logf(lsum);
    }
    __syncthreads();                      // the staging region becomes operand buffers + rings
  }

  // the token residual of the first epilogue and the slot keys / values: fetched and RETIRED before any weight DMA
  uint2 rsd[NSL][TT];
  {
    const bf16_t* tok = (const bf16_t*)p.tok + row0 * C;
#pragma unroll
    for (int s = 0; s < NSL; ++s)
#pragma unroll
      for (int tt = 0; tt < TT; ++tt)
        rsd[s][tt] = *reinterpret_cast<const uint2*>(tok + (long long)(tt * 16 + l15) * C + (w * NSL + s) * 16 + 4 * lg);
    const bf16_t* kvg = (const bf16_t*)p.kv2 + (long long)b * p.slots * p.ldkv;
    const int vpr = 2 * C / 8;                                 // 16-byte vectors per slot row
    for (int i = tid; i < p.slots * vpr; i += 512) {
      const int j = i / vpr, c = i - j * vpr;
      const u32x4 v = *reinterpret_cast<const u32x4*>(kvg + (long long)j * p.ldkv + c * 8);
      *reinterpret_cast<__attribute__((address_space(3))) u32x4*>(KV + j * (4 * C) + c * 16) = v;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  if (w >= 4) __builtin_amdgcn_s_setprio(1);
  StRing<D> rg;
  st_ring_init<D>(rg, p.wstream_b, G::UB, smem + G::RING_OFF, lane, w);
  {
    const int hs = TT == 4 ? (w >> 1) : (w & 3), qh = TT == 4 ? (w & 1) : 0;
    const int ql = lane & 31, hh = lane >> 5;
    const int r = qh * 32 + ql;
    bf16_t* a1 = (bf16_t*)p.a1 + (row0 + r) * C;
    if (TT == 4 || w < 4)
#pragma unroll
      for (int rd = 0; rd < HEADS / 4; ++rd) {
        const int h = rd * 4 + hs;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int c = h * 4 + j;
          const int phys = (c & ~15) | ((c ^ r) & 15);
          uint2 v;
          v.x = opack[rd][2 * j];
          v.y = opack[rd][2 * j + 1];
          *reinterpret_cast<__attribute__((address_space(3))) u32x2*>(Y + r * PITCH + phys * 16 + hh * 8) = u32x2{v.x, v.y};
          *reinterpret_cast<uint2*>(a1 + h * 32 + 8 * j + 4 * hh) = v;
        }
      }
  }
  ST_BARRIER();                           // attention output complete in Y, slot keys / values in KV

  int yaddr[4], gaddr[4], woff[2];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int sw = (((4 * j + lg) ^ l15) & 15) * 16;
    yaddr[j] = l15 * PITCH + sw;
    gaddr[j] = l15 * 256 + sw;
  }
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) woff[ks] = l15 * 128 + (((4 * ks + lg) ^ ((l15 >> 1) & 7)) * 16);
  float sx[TT], sxx[TT];
#pragma unroll
  for (int tt = 0; tt < TT; ++tt) sx[tt] = sxx[tt] = 0.f;

  f32x4 res[NSL][TT], acc[NSL][TT], ev0[NSL], evg[NSL], evb[NSL];
  auto zero_acc = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int s = 0; s < NSL; ++s)
#pragma unroll
      for (int tt = 0; tt < TT; ++tt) acc[s][tt] = f32x4{0.f, 0.f, 0.f, 0.f};
  };
  auto load_vecs = [&](const float* bias, const float* gm, const float* bt) __attribute__((always_inline)) {
#pragma unroll
    for (int s = 0; s < NSL; ++s) {
      ev0[s] = st_vec4(bias + (w * NSL + s) * 16, lg);
      evg[s] = st_vec4(gm + (w * NSL + s) * 16, lg);
      evb[s] = st_vec4(bt + (w * NSL + s) * 16, lg);
    }
  };

  // ---- attn1.to_out + tok -> x1 ; LayerNorm2 -> n2
  load_vecs(p.b_o, p.ln2_g, p.ln2_b);
  zero_acc();
#pragma unroll
  for (int kt = 0; kt < KT; ++kt) st_gemm_step<D, NSL, false, 0, TT>(rg, Y, yaddr, kt, 16 * PITCH, woff, acc, sx, sxx);
#pragma unroll
  for (int s = 0; s < NSL; ++s)
#pragma unroll
    for (int tt = 0; tt < TT; ++tt) {
      const uint2 t2 = rsd[s][tt];
      res[s][tt][0] = acc[s][tt][0] + ev0[s][0] + __uint_as_float(t2.x << 16);
      res[s][tt][1] = acc[s][tt][1] + ev0[s][1] + __uint_as_float(t2.x & 0xffff0000u);
      res[s][tt][2] = acc[s][tt][2] + ev0[s][2] + __uint_as_float(t2.y << 16);
      res[s][tt][3] = acc[s][tt][3] + ev0[s][3] + __uint_as_float(t2.y & 0xffff0000u);
    }
  st_rows_store<C, TT, NSL>(res, (bf16_t*)p.x1 + row0 * C, C, w, l15, lg);
  st_layernorm_rows<C, TT, NSL>(res, evg, evb, p.ln_eps, Y, lnred, (bf16_t*)p.n2 + row0 * C, p.st2 + row0 * 2, w, l15, lg);

  // ---- slot cross-attention: q2 = n2 Wq2^T ; per (row, head): softmax over the slots ; a2
  load_vecs(p.b_o2, p.ln3_g, p.ln3_b);            // (epilogue vectors of the GEMM BEHIND the attention: fetched early)
  zero_acc();
  st_gemm_step<D, NSL, false, NSL * TT, TT>(rg, Y, yaddr, 0, 16 * PITCH, woff, acc, sx, sxx);   // (EXTRA: the stores / loads just issued stay outstanding)
#pragma unroll
  for (int kt = 1; kt < KT; ++kt) st_gemm_step<D, NSL, false, 0, TT>(rg, Y, yaddr, kt, 16 * PITCH, woff, acc, sx, sxx);
  ST_BARRIER();                                    // every wave is done reading n2
  st_rows_to_y<C, TT, NSL, true>(acc, Y, (bf16_t*)p.q2 + row0 * C, w, l15, lg);
  ST_BARRIER();
  {
    const int slots = p.slots;
    for (int it = tid; it < ROWS * HEADS; it += 512) {
      const int r = it % ROWS, hd = it / ROWS;
      float q[32];
#pragma unroll
      for (int c4 = 0; c4 < 4; ++c4) {
        const int c = hd * 4 + c4;
        const int phys = (c & ~15) | ((c ^ r) & 15);
        const u32x4 v = *reinterpret_cast<const __attribute__((address_space(3))) u32x4*>(Y + r * PITCH + phys * 16);
        unpack16<bf16_t>(make_uint4(v[0], v[1], v[2], v[3]), q + c4 * 8);
      }
      float sc[KVS];
      float mx = -INFINITY;
#pragma unroll
      for (int j = 0; j < KVS; ++j) {
        sc[j] = -INFINITY;
        if (j < slots) {
          float d = 0.f;
#pragma unroll
          for (int c4 = 0; c4 < 4; ++c4) {
            const u32x4 v = *reinterpret_cast<const __attribute__((address_space(3))) u32x4*>(KV + j * (4 * C) + hd * 64 + c4 * 16);
            float kf[8];
            unpack16<bf16_t>(make_uint4(v[0], v[1], v[2], v[3]), kf);
#pragma unroll
            for (int e = 0; e < 8; ++e) d = fmaf(q[c4 * 8 + e], kf[e], d);
          }
          sc[j] = d * p.attn_scale;
          mx = fmaxf(mx, sc[j]);
        }
      }
      float sm = 0.f;
#pragma unroll
      for (int j = 0; j < KVS; ++j) {
        sc[j] = j < slots ? __expf(sc[j] - mx) : 0.f;
        sm += sc[j];
      }
      const float inv = 1.f / sm;
      float o[32];
#pragma unroll
      for (int e = 0; e < 32; ++e) o[e] = 0.f;
#pragma unroll
      for (int j = 0; j < KVS; ++j)
        if (j < slots) {
          const float pj = sc[j] * inv;
#pragma unroll
          for (int c4 = 0; c4 < 4; ++c4) {
            const u32x4 v = *reinterpret_cast<const __attribute__((address_space(3))) u32x4*>(KV + j * (4 * C) + 2 * C + hd * 64 + c4 * 16);
            float vf[8];
            unpack16<bf16_t>(make_uint4(v[0], v[1], v[2], v[3]), vf);
#pragma unroll
            for (int e = 0; e < 8; ++e) o[c4 * 8 + e] = fmaf(pj, vf[e], o[c4 * 8 + e]);
          }
        }
      bf16_t* a2 = (bf16_t*)p.a2 + (row0 + r) * C + hd * 32;
#pragma unroll
      for (int c4 = 0; c4 < 4; ++c4) {
        const int c = hd * 4 + c4;
        const int phys = (c & ~15) | ((c ^ r) & 15);
        const uint4 pk = pack16<bf16_t>(o + c4 * 8);
        *reinterpret_cast<__attribute__((address_space(3))) u32x4*>(Y + r * PITCH + phys * 16) = u32x4{pk.x, pk.y, pk.z, pk.w};
        *reinterpret_cast<uint4*>(a2 + c4 * 8) = pk;
      }
      p.lse2[((long long)b * HEADS + hd) * S + rb * ROWS + r] = mx + __logf(sm);
    }
  }
  ST_BARRIER();

  // ---- attn2.to_out + x1 -> x2 ; LayerNorm3 -> n3
  zero_acc();
#pragma unroll
  for (int kt = 0; kt < KT; ++kt) st_gemm_step<D, NSL, false, 0, TT>(rg, Y, yaddr, kt, 16 * PITCH, woff, acc, sx, sxx);
#pragma unroll
  for (int s = 0; s < NSL; ++s)
#pragma unroll
    for (int tt = 0; tt < TT; ++tt)
#pragma unroll
      for (int j = 0; j < 4; ++j) res[s][tt][j] += acc[s][tt][j] + ev0[s][j];
  st_rows_store<C, TT, NSL>(res, (bf16_t*)p.x2 + row0 * C, C, w, l15, lg);
  st_layernorm_rows<C, TT, NSL>(res, evg, evb, p.ln_eps, Y, lnred, (bf16_t*)p.n3 + row0 * C, p.st3 + row0 * 2, w, l15, lg);

  // ---- GEGLU feed-forward, hidden chunk by hidden chunk: h = n3 W1^T + b1 ; g = value * gelu(gate) ; acc += g Wff2^T
#pragma unroll
  for (int s = 0; s < NSL; ++s) ev0[s] = st_vec4(p.b_ff2 + (w * NSL + s) * 16, lg);
  zero_acc();
  bf16_t* hrow = (bf16_t*)p.h + row0 * 8 * C;
  bf16_t* grow = (bf16_t*)p.g + row0 * 4 * C;
#pragma unroll 1
  for (int hc = 0; hc < G::NHC; ++hc) {
    f32x4 vg[2][TT];
    const f32x4 biv = st_vec4(p.b_ff1 + hc * 128 + w * 16, lg);
    const f32x4 big = st_vec4(p.b_ff1 + 4 * C + hc * 128 + w * 16, lg);
#pragma unroll
    for (int tt = 0; tt < TT; ++tt) vg[0][tt] = vg[1][tt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kt = 0; kt < KT; ++kt) st_gemm_step<D, 2, false, 0, TT>(rg, Y, yaddr, kt, 16 * PITCH, woff, vg, sx, sxx);
    lds_char* gb = Gb + (G::GBUF == 2 ? (hc & 1) * G::G_BYTES : 0);
    if (G::GBUF == 1) ST_BARRIER();                     // the previous chunk's readers are done
    {
      const int n0 = hc * 128 + w * 16 + 4 * lg;
      const int c = (w * 16 + 4 * lg) >> 3;
#pragma unroll
      for (int tt = 0; tt < TT; ++tt) {
        const int r = tt * 16 + l15;
        float v[4], gt[4], y[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          v[j] = vg[0][tt][j] + biv[j];
          gt[j] = vg[1][tt][j] + big[j];
          y[j] = v[j] * act_apply<true>(gt[j], SDMI_ACT_GELU);
        }
        uint2 hv, hg, o;
        hv.x = st_pack2(v[0], v[1]);
        hv.y = st_pack2(v[2], v[3]);
        hg.x = st_pack2(gt[0], gt[1]);
        hg.y = st_pack2(gt[2], gt[3]);
        o.x = st_pack2(y[0], y[1]);
        o.y = st_pack2(y[2], y[3]);
        *reinterpret_cast<uint2*>(hrow + (long long)r * 8 * C + n0) = hv;
        *reinterpret_cast<uint2*>(hrow + (long long)r * 8 * C + 4 * C + n0) = hg;
        *reinterpret_cast<uint2*>(grow + (long long)r * 4 * C + n0) = o;
        const int phys = (c & ~15) | ((c ^ r) & 15);
        *reinterpret_cast<__attribute__((address_space(3))) u32x2*>(gb + r * 256 + phys * 16 + (lg & 1) * 8) = u32x2{o.x, o.y};
      }
    }
    ST_BARRIER();
    st_gemm_step<D, NSL, false, 3 * TT, TT>(rg, gb, gaddr, 0, 16 * 256, woff, acc, sx, sxx);   // (EXTRA: the stores / loads just issued stay outstanding)
#pragma unroll
    for (int kt = 1; kt < 2; ++kt) st_gemm_step<D, NSL, false, 0, TT>(rg, gb, gaddr, kt, 16 * 256, woff, acc, sx, sxx);
  }
  // ---- x3 = acc + bff2 + x2 -> operand buffer ; out = x3 Wpo^T + bpo + x
#pragma unroll
  for (int s = 0; s < NSL; ++s)
#pragma unroll
    for (int tt = 0; tt < TT; ++tt)
#pragma unroll
      for (int j = 0; j < 4; ++j) res[s][tt][j] += acc[s][tt][j] + ev0[s][j];
#pragma unroll
  for (int s = 0; s < NSL; ++s) ev0[s] = st_vec4(p.b_po + (w * NSL + s) * 16, lg);
  ST_BARRIER();                                         // (Y = n3: every wave's last GEGLU GEMM is done with it)
  st_rows_to_y<C, TT, NSL, true>(res, Y, (bf16_t*)p.x3 + row0 * C, w, l15, lg);
  ST_BARRIER();
  zero_acc();
  st_gemm_step<D, NSL, false, NSL * TT, TT>(rg, Y, yaddr, 0, 16 * PITCH, woff, acc, sx, sxx);   // (EXTRA: the stores / loads just issued stay outstanding)
#pragma unroll
  for (int kt = 1; kt < KT; ++kt) st_gemm_step<D, NSL, false, 0, TT>(rg, Y, yaddr, kt, 16 * PITCH, woff, acc, sx, sxx);
  // (only dummy re-fetches are in flight now: drain them, then ordinary loads are safe)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  {
    const bf16_t* xr = (const bf16_t*)p.x + row0 * C;
#pragma unroll
    for (int s = 0; s < NSL; ++s)
#pragma unroll
      for (int tt = 0; tt < TT; ++tt)
        rsd[s][tt] = *reinterpret_cast<const uint2*>(xr + (long long)(tt * 16 + l15) * C + (w * NSL + s) * 16 + 4 * lg);
    bf16_t* outp = (bf16_t*)p.out + row0 * C;
#pragma unroll
    for (int s = 0; s < NSL; ++s) {
      const int n0 = (w * NSL + s) * 16 + 4 * lg;
      const f32x4 bi = ev0[s];
#pragma unroll
      for (int tt = 0; tt < TT; ++tt) {
        const uint2 x2 = rsd[s][tt];
        uint2 o;
        o.x = st_pack2(acc[s][tt][0] + bi[0] + __uint_as_float(x2.x << 16),
                       acc[s][tt][1] + bi[1] + __uint_as_float(x2.x & 0xffff0000u));
        o.y = st_pack2(acc[s][tt][2] + bi[2] + __uint_as_float(x2.y << 16),
                       acc[s][tt][3] + bi[3] + __uint_as_float(x2.y & 0xffff0000u));
        *reinterpret_cast<uint2*>(outp + (long long)(tt * 16 + l15) * C + n0) = o;
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// ---------------------------------------------------------------------------------------------------------
// weight units from the parameter arena (sdmi_st_pack): two units per workgroup, one 16-byte chunk per thread
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void st_pack_kernel(const SdmiStPackDesc* __restrict__ descs, int n_units) {
  const int u = (int)blockIdx.x * 2 + ((int)threadIdx.x >> 7);
  if (u >= n_units) return;
  const SdmiStPackDesc d = descs[u];
  const int ch = threadIdx.x & 127, r = ch >> 3, pc = ch & 7;
  const int lc = pc ^ ((r >> 1) & 7);
  const bf16_t* src = (const bf16_t*)d.src + (long long)r * d.rs + (long long)lc * 8 * d.cs;
  uint4 v;
  if (d.cs == 1) {
    v = *reinterpret_cast<const uint4*>(src);
  } else {
    unsigned e[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) e[i] = src[(long long)i * d.cs];
    v = make_uint4(e[0] | (e[1] << 16), e[2] | (e[3] << 16), e[4] | (e[5] << 16), e[6] | (e[7] << 16));
  }
  *reinterpret_cast<uint4*>((char*)d.dst + ch * 16) = v;
}

// =========================================================================================================
// backward data path (sdmi_st_train_bwd): three launches around the two attention backward passes
// =========================================================================================================
template <int C, int TT>
struct StTrBwdGeom {
  static constexpr int ROWS = 16 * TT, NSL = C / 128, KT = C / 64, NHC = C / 32, PITCH = C * 2;
  static constexpr int Y_BYTES = ROWS * PITCH, G_BYTES = ROWS * 256, RED_BYTES = 8 * ROWS * 8;
  // phase B1: operand buffer | value-gradient chunk | gate-gradient chunk | LayerNorm partials | rings
  static constexpr int GV_OFF = Y_BYTES, GG_OFF = GV_OFF + G_BYTES, RED1_OFF = GG_OFF + G_BYTES, RING1_OFF = RED1_OFF + RED_BYTES;
  static constexpr int D1F = (160 * 1024 - RING1_OFF) / (8 * ST_UNIT);
  static constexpr int D1 = D1F >= 8 ? 8 : D1F;
  static_assert(D1 >= 4 && D1 >= NSL + 1, "ring too shallow");
  static constexpr int SMEM1 = RING1_OFF + 8 * D1 * ST_UNIT;
  // phases B2 / A: operand buffer | LayerNorm partials | rings
  static constexpr int RED2_OFF = Y_BYTES, RING2_OFF = RED2_OFF + RED_BYTES;
  static constexpr int D2F = (160 * 1024 - RING2_OFF) / (8 * ST_UNIT);
  static constexpr int D2 = D2F >= 8 ? 8 : D2F;
  static constexpr int SMEM2 = RING2_OFF + 8 * D2 * ST_UNIT;
  static constexpr int UB1 = 2 * KT * NSL + NHC * (KT + 4 * NSL);      // proj_out^T, chunks (ff2^T, ff1^T value | gate), to_out2^T
  static constexpr int UB2 = 2 * KT * NSL;                             // to_q2^T, to_out^T
  static constexpr int UA = 4 * KT * NSL;                              // q^T, k^T, v^T, proj_in^T
};

// ROWS x C bf16 rows from global memory into the swizzled operand buffer (all 512 threads, 16-byte vectors)
template <int C, int TT>
__device__ __forceinline__ void st_rows_global_to_y(const bf16_t* src, int ld, lds_char* Y, int tid) {
  constexpr int ROWS = 16 * TT, VPR = C / 8, PITCH = 2 * C;
  static_assert((ROWS * VPR) % 512 == 0, "whole passes");
  u32x4 v[ROWS * VPR / 512];
#pragma unroll
  for (int it = 0; it < ROWS * VPR / 512; ++it) {
    const int i = tid + it * 512, r = i / VPR, vc = i - r * VPR;
    v[it] = *reinterpret_cast<const u32x4*>(src + (long long)r * ld + vc * 8);
  }
#pragma unroll
  for (int it = 0; it < ROWS * VPR / 512; ++it) {
    const int i = tid + it * 512, r = i / VPR, vc = i - r * VPR;
    const int phys = (vc & ~15) | ((vc ^ r) & 15);
    *reinterpret_cast<__attribute__((address_space(3))) u32x4*>(Y + r * PITCH + phys * 16) = v[it];
  }
}

// LayerNorm backward on rows held across the workgroup: dn (gradient of the normalised rows, fp32) ->
//   dx = rstd (g - mean_k g - xhat mean_k (g xhat)) + residual gradient,   g = dn gamma, xhat = (x - mean) rstd
// written back into dn; per-workgroup column sums (sum_rows dn xhat, sum_rows dn) -> colpart [C] float2.
// x rows / statistics / the bf16 residual gradient come from global memory; RES_REG: the residual gradient is `resreg`.
// One workgroup barrier (also: every wave is done reading the operand buffer of the GEMM in front).
template <int C, int TT, int NSL, bool RES_REG>
__device__ __forceinline__ void st_ln_bwd_rows(f32x4 (&dn)[NSL][TT], const bf16_t* xrows, const float* strows,
                                               const float* gamma, const bf16_t* resrows, const f32x4 (&resreg)[NSL][TT],
                                               float* red, float* colpart, int w, int l15, int lg) {
  constexpr int ROWS = 16 * TT;
  uint2 xv[NSL][TT], rv[NSL][TT];
  float2 ms[TT];
  f32x4 gm[NSL];
#pragma unroll
  for (int s = 0; s < NSL; ++s) {
    const int n0 = (w * NSL + s) * 16 + 4 * lg;
    gm[s] = *reinterpret_cast<const f32x4*>(gamma + n0);
#pragma unroll
    for (int tt = 0; tt < TT; ++tt) {
      xv[s][tt] = *reinterpret_cast<const uint2*>(xrows + (long long)(tt * 16 + l15) * C + n0);
      if constexpr (!RES_REG) rv[s][tt] = *reinterpret_cast<const uint2*>(resrows + (long long)(tt * 16 + l15) * C + n0);
    }
  }
#pragma unroll
  for (int tt = 0; tt < TT; ++tt) ms[tt] = *reinterpret_cast<const float2*>(strows + (long long)(tt * 16 + l15) * 2);
  float xh[NSL][TT][4];
  float cg[NSL][4], cb[NSL][4];
#pragma unroll
  for (int s = 0; s < NSL; ++s)
#pragma unroll
    for (int j = 0; j < 4; ++j) cg[s][j] = cb[s][j] = 0.f;
#pragma unroll
  for (int tt = 0; tt < TT; ++tt) {
    float a = 0.f, b = 0.f;
#pragma unroll
    for (int s = 0; s < NSL; ++s) {
      const uint2 t2 = xv[s][tt];
      const float xf[4] = {__uint_as_float(t2.x << 16), __uint_as_float(t2.x & 0xffff0000u), __uint_as_float(t2.y << 16),
                           __uint_as_float(t2.y & 0xffff0000u)};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float h_ = (xf[j] - ms[tt].x) * ms[tt].y;
        xh[s][tt][j] = h_;
        const float g = dn[s][tt][j] * gm[s][j];
        a += g;
        b += g * h_;
        cg[s][j] += dn[s][tt][j] * h_;
        cb[s][j] += dn[s][tt][j];
      }
    }
    a = st_sum_x32(st_sum_x16(a));
    b = st_sum_x32(st_sum_x16(b));
    if (lg == 0) *reinterpret_cast<float2*>(red + (w * ROWS + tt * 16 + l15) * 2) = make_float2(a, b);
  }
  // column sums over this workgroup's rows: the 16 row lanes of a column group, then one lane writes
#pragma unroll
  for (int s = 0; s < NSL; ++s)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float a = st_sum_row16(cg[s][j]), b = st_sum_row16(cb[s][j]);
      if (l15 == 0) *reinterpret_cast<float2*>(colpart + (long long)((w * NSL + s) * 16 + 4 * lg + j) * 2) = make_float2(a, b);
    }
  ST_BARRIER();
#pragma unroll
  for (int tt = 0; tt < TT; ++tt) {
    const int r = tt * 16 + l15;
    float a = 0.f, b = 0.f;
#pragma unroll
    for (int ww = 0; ww < 8; ++ww) {
      const float2 v = *reinterpret_cast<const float2*>(red + (ww * ROWS + r) * 2);
      a += v.x;
      b += v.y;
    }
    const float c1 = a * (1.f / (float)C), c2 = b * (1.f / (float)C), rstd = ms[tt].y;
#pragma unroll
    for (int s = 0; s < NSL; ++s) {
      float rr[4];
      if constexpr (RES_REG) {
#pragma unroll
        for (int j = 0; j < 4; ++j) rr[j] = resreg[s][tt][j];
      } else {
        const uint2 t2 = rv[s][tt];
        rr[0] = __uint_as_float(t2.x << 16);
        rr[1] = __uint_as_float(t2.x & 0xffff0000u);
        rr[2] = __uint_as_float(t2.y << 16);
        rr[3] = __uint_as_float(t2.y & 0xffff0000u);
      }
#pragma unroll
      for (int j = 0; j < 4; ++j)
        dn[s][tt][j] = rstd * (dn[s][tt][j] * gm[s][j] - c1 - xh[s][tt][j] * c2) + rr[j];
    }
  }
}

#define ST_BWD_PROLOGUE(GEOM)                                                                         \
  extern __shared__ __attribute__((aligned(16))) char smem_[];                                        \
  lds_char* const smem = (lds_char*)smem_;                                                            \
  lds_char* const Y = smem;                                                                           \
  const int tid = threadIdx.x, lane = tid & 63, l15 = lane & 15, lg = lane >> 4;                      \
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);                                             \
  const int vid = st_xcd_id((int)blockIdx.x, (int)gridDim.x);                                         \
  const long long row0 = (long long)vid * GEOM::ROWS;                                                 \
  int yaddr[4], gaddr[4], woff[2];                                                                    \
  _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                     \
    const int sw = (((4 * j + lg) ^ l15) & 15) * 16;                                                  \
    yaddr[j] = l15 * GEOM::PITCH + sw;                                                                \
    gaddr[j] = l15 * 256 + sw;                                                                        \
  }                                                                                                   \
  _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) woff[ks] = l15 * 128 + (((4 * ks + lg) ^ ((l15 >> 1) & 7)) * 16); \
  float sx[TT], sxx[TT];                                                                              \
  _Pragma("unroll") for (int tt = 0; tt < TT; ++tt) sx[tt] = sxx[tt] = 0.f;                           \
  (void)gaddr

// ---- phase B1: dout -> dx3 -> (dh) -> dn3 -> dx2 -> da2
template <int C, int TT>
__global__ __launch_bounds__(512) void st_train_bwd_b1_kernel(SdmiStTrainBwdArgs p) {
  typedef StTrBwdGeom<C, TT> G;
  constexpr int NSL = G::NSL, KT = G::KT, D = G::D1, PITCH = G::PITCH;
  ST_BWD_PROLOGUE(G);
  lds_char* const Gv = smem + G::GV_OFF;
  lds_char* const Gg = smem + G::GG_OFF;
  float* const red = (float*)(smem_ + G::RED1_OFF);
  StRing<D> rg;
  st_ring_init<D>(rg, p.wstream_b1, G::UB1, smem + G::RING1_OFF, lane, w);
  st_rows_global_to_y<C, TT>((const bf16_t*)p.dout + row0 * C, C, Y, tid);
  ST_BARRIER();
  f32x4 res[NSL][TT], acc[NSL][TT];
  auto zero_acc = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int s = 0; s < NSL; ++s)
#pragma unroll
      for (int tt = 0; tt < TT; ++tt) acc[s][tt] = f32x4{0.f, 0.f, 0.f, 0.f};
  };
  // ---- dx3 = dout Wpo
  zero_acc();
#pragma unroll
  for (int kt = 0; kt < KT; ++kt) st_gemm_step<D, NSL, false, 0, TT>(rg, Y, yaddr, kt, 16 * PITCH, woff, acc, sx, sxx);
#pragma unroll
  for (int s = 0; s < NSL; ++s)
#pragma unroll
    for (int tt = 0; tt < TT; ++tt) res[s][tt] = acc[s][tt];
  ST_BARRIER();
  st_rows_to_y<C, TT, NSL, true>(res, Y, (bf16_t*)p.dx3 + row0 * C, w, l15, lg);
  ST_BARRIER();
  // ---- feed-forward backward, hidden chunk by hidden chunk: dg = dx3 Wff2 ; GEGLU' ; dn3 += [dval | dgate] W1
  zero_acc();
  const bf16_t* hrow = (const bf16_t*)p.h + row0 * 8 * C;
  bf16_t* dhrow = (bf16_t*)p.dh + row0 * 8 * C;
#pragma unroll 1
  for (int hc = 0; hc < G::NHC; ++hc) {
    const int n0 = hc * 128 + w * 16 + 4 * lg;
    uint2 hv[TT], hg[TT];
#pragma unroll
    for (int tt = 0; tt < TT; ++tt) {
      hv[tt] = *reinterpret_cast<const uint2*>(hrow + (long long)(tt * 16 + l15) * 8 * C + n0);
      hg[tt] = *reinterpret_cast<const uint2*>(hrow + (long long)(tt * 16 + l15) * 8 * C + 4 * C + n0);
    }
    f32x4 dg[1][TT];
#pragma unroll
    for (int tt = 0; tt < TT; ++tt) dg[0][tt] = f32x4{0.f, 0.f, 0.f, 0.f};
    st_gemm_step<D, 1, false, 2 * TT, TT>(rg, Y, yaddr, 0, 16 * PITCH, woff, dg, sx, sxx);   // (EXTRA: the stores / loads just issued stay outstanding)
#pragma unroll
    for (int kt = 1; kt < KT; ++kt) st_gemm_step<D, 1, false, 0, TT>(rg, Y, yaddr, kt, 16 * PITCH, woff, dg, sx, sxx);
    ST_BARRIER();                                       // the previous chunk's readers are done with Gv / Gg
    {
      const int c = (w * 16 + 4 * lg) >> 3;
#pragma unroll
      for (int tt = 0; tt < TT; ++tt) {
        const int r = tt * 16 + l15;
        const float v[4] = {__uint_as_float(hv[tt].x << 16), __uint_as_float(hv[tt].x & 0xffff0000u),
                            __uint_as_float(hv[tt].y << 16), __uint_as_float(hv[tt].y & 0xffff0000u)};
        const float gt[4] = {__uint_as_float(hg[tt].x << 16), __uint_as_float(hg[tt].x & 0xffff0000u),
                             __uint_as_float(hg[tt].y << 16), __uint_as_float(hg[tt].y & 0xffff0000u)};
        float dv[4], dgt[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          dv[j] = dg[0][tt][j] * act_apply<true>(gt[j], SDMI_ACT_GELU);
          dgt[j] = dg[0][tt][j] * v[j] * act_grad<true>(gt[j], SDMI_ACT_GELU);
        }
        uint2 ov, og;
        ov.x = st_pack2(dv[0], dv[1]);
        ov.y = st_pack2(dv[2], dv[3]);
        og.x = st_pack2(dgt[0], dgt[1]);
        og.y = st_pack2(dgt[2], dgt[3]);
        *reinterpret_cast<uint2*>(dhrow + (long long)r * 8 * C + n0) = ov;
        *reinterpret_cast<uint2*>(dhrow + (long long)r * 8 * C + 4 * C + n0) = og;
        const int phys = (c & ~15) | ((c ^ r) & 15);
        *reinterpret_cast<__attribute__((address_space(3))) u32x2*>(Gv + r * 256 + phys * 16 + (lg & 1) * 8) = u32x2{ov.x, ov.y};
        *reinterpret_cast<__attribute__((address_space(3))) u32x2*>(Gg + r * 256 + phys * 16 + (lg & 1) * 8) = u32x2{og.x, og.y};
      }
    }
    ST_BARRIER();
    st_gemm_step<D, NSL, false, 2 * TT, TT>(rg, Gv, gaddr, 0, 16 * 256, woff, acc, sx, sxx);   // (EXTRA: the stores / loads just issued stay outstanding)
#pragma unroll
    for (int kt = 1; kt < 2; ++kt) st_gemm_step<D, NSL, false, 0, TT>(rg, Gv, gaddr, kt, 16 * 256, woff, acc, sx, sxx);
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) st_gemm_step<D, NSL, false, 0, TT>(rg, Gg, gaddr, kt, 16 * 256, woff, acc, sx, sxx);
  }
  // ---- LayerNorm3 backward + dx3 -> dx2
  st_ln_bwd_rows<C, TT, NSL, true>(acc, (const bf16_t*)p.x2 + row0 * C, p.st3 + row0 * 2, p.ln3_g, nullptr, res, red,
                                   p.ln3_part + (long long)vid * C * 2, w, l15, lg);
  st_rows_to_y<C, TT, NSL, true>(acc, Y, (bf16_t*)p.dx2 + row0 * C, w, l15, lg);
  ST_BARRIER();
  // ---- da2 = dx2 Wo2
  zero_acc();
  st_gemm_step<D, NSL, false, NSL * TT, TT>(rg, Y, yaddr, 0, 16 * PITCH, woff, acc, sx, sxx);   // (EXTRA: the stores / loads just issued stay outstanding)
#pragma unroll
  for (int kt = 1; kt < KT; ++kt) st_gemm_step<D, NSL, false, 0, TT>(rg, Y, yaddr, kt, 16 * PITCH, woff, acc, sx, sxx);
  st_rows_store<C, TT, NSL>(acc, (bf16_t*)p.da2 + row0 * C, C, w, l15, lg);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// ---- phase B2: dq2 -> dn2 -> LN2' + dx2 -> dx1 -> da1
template <int C, int TT>
__global__ __launch_bounds__(512) void st_train_bwd_b2_kernel(SdmiStTrainBwdArgs p) {
  typedef StTrBwdGeom<C, TT> G;
  constexpr int NSL = G::NSL, KT = G::KT, D = G::D2, PITCH = G::PITCH;
  ST_BWD_PROLOGUE(G);
  float* const red = (float*)(smem_ + G::RED2_OFF);
  StRing<D> rg;
  st_ring_init<D>(rg, p.wstream_b2, G::UB2, smem + G::RING2_OFF, lane, w);
  st_rows_global_to_y<C, TT>((const bf16_t*)p.dq2 + row0 * C, C, Y, tid);
  ST_BARRIER();
  f32x4 acc[NSL][TT];
#pragma unroll
  for (int s = 0; s < NSL; ++s)
#pragma unroll
    for (int tt = 0; tt < TT; ++tt) acc[s][tt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int kt = 0; kt < KT; ++kt) st_gemm_step<D, NSL, false, 0, TT>(rg, Y, yaddr, kt, 16 * PITCH, woff, acc, sx, sxx);
  st_ln_bwd_rows<C, TT, NSL, false>(acc, (const bf16_t*)p.x1 + row0 * C, p.st2 + row0 * 2, p.ln2_g,
                                    (const bf16_t*)p.dx2 + row0 * C, acc, red, p.ln2_part + (long long)vid * C * 2, w, l15, lg);
  st_rows_to_y<C, TT, NSL, true>(acc, Y, (bf16_t*)p.dx1 + row0 * C, w, l15, lg);
  ST_BARRIER();
#pragma unroll
  for (int s = 0; s < NSL; ++s)
#pragma unroll
    for (int tt = 0; tt < TT; ++tt) acc[s][tt] = f32x4{0.f, 0.f, 0.f, 0.f};
  st_gemm_step<D, NSL, false, NSL * TT, TT>(rg, Y, yaddr, 0, 16 * PITCH, woff, acc, sx, sxx);   // (EXTRA: the stores / loads just issued stay outstanding)
#pragma unroll
  for (int kt = 1; kt < KT; ++kt) st_gemm_step<D, NSL, false, 0, TT>(rg, Y, yaddr, kt, 16 * PITCH, woff, acc, sx, sxx);
  st_rows_store<C, TT, NSL>(acc, (bf16_t*)p.da1 + row0 * C, C, w, l15, lg);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// ---- phase A: dqkv -> dn1 -> LN1' + dx1 -> dtok -> dhgn
template <int C, int TT>
__global__ __launch_bounds__(512) void st_train_bwd_a_kernel(SdmiStTrainBwdArgs p) {
  typedef StTrBwdGeom<C, TT> G;
  constexpr int NSL = G::NSL, KT = G::KT, D = G::D2, PITCH = G::PITCH;
  ST_BWD_PROLOGUE(G);
  float* const red = (float*)(smem_ + G::RED2_OFF);
  StRing<D> rg;
  st_ring_init<D>(rg, p.wstream_a, G::UA, smem + G::RING2_OFF, lane, w);
  const bf16_t* dqkv = (const bf16_t*)p.dqkv + row0 * 3 * C;
  st_rows_global_to_y<C, TT>(dqkv, 3 * C, Y, tid);
  ST_BARRIER();
  f32x4 acc[NSL][TT];
#pragma unroll
  for (int s = 0; s < NSL; ++s)
#pragma unroll
    for (int tt = 0; tt < TT; ++tt) acc[s][tt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int pass = 0; pass < 3; ++pass) {
    if (pass) {
      ST_BARRIER();                                     // every wave is done with the previous part of dqkv
      st_rows_global_to_y<C, TT>(dqkv + pass * C, 3 * C, Y, tid);
      ST_BARRIER();
    }
#pragma unroll
    for (int kt = 0; kt < KT; ++kt) st_gemm_step<D, NSL, false, 0, TT>(rg, Y, yaddr, kt, 16 * PITCH, woff, acc, sx, sxx);
  }
  st_ln_bwd_rows<C, TT, NSL, false>(acc, (const bf16_t*)p.tok + row0 * C, p.st1 + row0 * 2, p.ln1_g,
                                    (const bf16_t*)p.dx1 + row0 * C, acc, red, p.ln1_part + (long long)vid * C * 2, w, l15, lg);
  st_rows_to_y<C, TT, NSL, true>(acc, Y, (bf16_t*)p.dtok + row0 * C, w, l15, lg);
  ST_BARRIER();
#pragma unroll
  for (int s = 0; s < NSL; ++s)
#pragma unroll
    for (int tt = 0; tt < TT; ++tt) acc[s][tt] = f32x4{0.f, 0.f, 0.f, 0.f};
  st_gemm_step<D, NSL, false, NSL * TT, TT>(rg, Y, yaddr, 0, 16 * PITCH, woff, acc, sx, sxx);   // (EXTRA: the stores / loads just issued stay outstanding)
#pragma unroll
  for (int kt = 1; kt < KT; ++kt) st_gemm_step<D, NSL, false, 0, TT>(rg, Y, yaddr, kt, 16 * PITCH, woff, acc, sx, sxx);
  st_rows_store<C, TT, NSL>(acc, (bf16_t*)p.dhgn + row0 * C, C, w, l15, lg);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

template <int C, int TT>
int st_train_bwd_launch(const SdmiStTrainBwdArgs& a, hipStream_t st) {
  typedef StTrBwdGeom<C, TT> G;
  const int grid = a.B * (a.S / G::ROWS);
  if (a.phase == 1) {
    SDMI_OPTIN_LDS((st_train_bwd_b1_kernel<C, TT>), G::SMEM1, "st_train_bwd (B1)");
    hipLaunchKernelGGL((st_train_bwd_b1_kernel<C, TT>), dim3(grid), dim3(512), G::SMEM1, st, a);
    return sdmi_check_launch("st_train_bwd (B1)");
  }
  if (a.phase == 2) {
    SDMI_OPTIN_LDS((st_train_bwd_b2_kernel<C, TT>), G::SMEM2, "st_train_bwd (B2)");
    hipLaunchKernelGGL((st_train_bwd_b2_kernel<C, TT>), dim3(grid), dim3(512), G::SMEM2, st, a);
    return sdmi_check_launch("st_train_bwd (B2)");
  }
  SDMI_OPTIN_LDS((st_train_bwd_a_kernel<C, TT>), G::SMEM2, "st_train_bwd (A)");
  hipLaunchKernelGGL((st_train_bwd_a_kernel<C, TT>), dim3(grid), dim3(512), G::SMEM2, st, a);
  return sdmi_check_launch("st_train_bwd (A)");
}

template <int C, int TT, int KVS>
int st_train_launch(const SdmiStTrainArgs& a, hipStream_t st) {
  typedef StTrGeom<C, TT, KVS> G;
  const int heads_bytes = a.S * (ST_KP + ST_VP);
  const int att = (G::HEADS * heads_bytes <= 160 * 1024 ? G::HEADS : 4) * heads_bytes;
  const int smem_b = att > G::SMEM_B ? att : G::SMEM_B;
  const int grid = a.B * (a.S / G::ROWS);
  if (a.phase == 0 || a.phase == 1) {
    SDMI_OPTIN_LDS((st_train_a_kernel<C, TT, KVS>), G::SMEM_A, "st_train_fwd (phase A)");
    hipLaunchKernelGGL((st_train_a_kernel<C, TT, KVS>), dim3(grid), dim3(512), G::SMEM_A, st, a);
    const int rc = sdmi_check_launch("st_train_fwd (phase A)");
    if (rc) return rc;
  }
  if (a.phase == 0 || a.phase == 2) {
    SDMI_OPTIN_LDS((st_train_b_kernel<C, TT, KVS>), 160 * 1024, "st_train_fwd (phase B)");
    hipLaunchKernelGGL((st_train_b_kernel<C, TT, KVS>), dim3(grid), dim3(512), smem_b, st, a);
    return sdmi_check_launch("st_train_fwd (phase B)");
  }
  return SDMI_OK;
}

}  // namespace

extern "C" int sdmi_st_train_fwd(const SdmiStTrainArgs* a, void* stream) {
  SDMI_REQUIRE(a && a->x && a->out && a->hgn && a->gn_stats && a->tok && a->n1 && a->st1 && a->qkv && a->a1 && a->lse1 &&
                   a->x1 && a->n2 && a->st2 && a->q2 && a->a2 && a->lse2 && a->x2 && a->n3 && a->st3 && a->h && a->g &&
                   a->x3 && a->kv2,
               "null tensor pointer");
  SDMI_REQUIRE(a->wstream_a && a->wstream_b && a->gn_gamma && a->gn_beta && a->b_in && a->ln1_g && a->ln1_b && a->b_o &&
                   a->ln2_g && a->ln2_b && a->b_o2 && a->ln3_g && a->ln3_b && a->b_ff1 && a->b_ff2 && a->b_po,
               "null parameter pointer");
  SDMI_REQUIRE(a->C == 256 || a->C == 384, "C must be 256 or 384");
  SDMI_REQUIRE(a->rows == 0 || a->rows == 64 || a->rows == 32, "rows per workgroup: 64 (0) or 32");
  const int rows = a->rows ? a->rows : 64;
  SDMI_REQUIRE(a->S >= rows && a->S % rows == 0 && a->S % 32 == 0 && 4 * a->S * (ST_KP + ST_VP) <= 160 * 1024,
               "S must be a multiple of the rows per workgroup and at most 256 tokens per image");
  SDMI_REQUIRE(a->slots >= 1 && a->slots <= 16 && a->ldkv >= 2 * a->C && a->ldkv % 8 == 0, "1..16 slots, kv rows of 2C");
  SDMI_REQUIRE(a->phase >= 0 && a->phase <= 2, "phase: 0 = both, 1 = A, 2 = B");
  hipStream_t st = (hipStream_t)stream;
  const bool wide = a->slots > 8;
#define ST_TR_GO(C_, TT_) (wide ? st_train_launch<C_, TT_, 16>(*a, st) : st_train_launch<C_, TT_, 8>(*a, st))
  if (rows == 64) return a->C == 256 ? ST_TR_GO(256, 4) : ST_TR_GO(384, 4);
  return a->C == 256 ? ST_TR_GO(256, 2) : ST_TR_GO(384, 2);
#undef ST_TR_GO
}

extern "C" int sdmi_st_pack(const SdmiStPackArgs* a, void* stream) {
  SDMI_REQUIRE(a && a->descs && a->n_units >= 1, "null / empty descriptor table");
  hipLaunchKernelGGL(st_pack_kernel, dim3((a->n_units + 1) / 2), dim3(256), 0, (hipStream_t)stream,
                     (const SdmiStPackDesc*)a->descs, a->n_units);
  return sdmi_check_launch("st_pack");
}

extern "C" int sdmi_st_train_bwd(const SdmiStTrainBwdArgs* a, void* stream) {
  SDMI_REQUIRE(a && a->phase >= 1 && a->phase <= 3, "phase: 1 = B1, 2 = B2, 3 = A");
  SDMI_REQUIRE(a->C == 256 || a->C == 384, "C must be 256 or 384");
  SDMI_REQUIRE(a->rows == 0 || a->rows == 64 || a->rows == 32, "rows per workgroup: 64 (0) or 32");
  const int rows = a->rows ? a->rows : 64;
  SDMI_REQUIRE(a->B >= 1 && a->S >= rows && a->S % rows == 0, "S must be a multiple of the rows per workgroup");
  if (a->phase == 1)
    SDMI_REQUIRE(a->dout && a->h && a->x2 && a->st3 && a->ln3_g && a->dx3 && a->dh && a->dx2 && a->da2 && a->ln3_part &&
                     a->wstream_b1, "phase B1: null pointer");
  else if (a->phase == 2)
    SDMI_REQUIRE(a->dq2 && a->x1 && a->st2 && a->ln2_g && a->dx2 && a->dx1 && a->da1 && a->ln2_part && a->wstream_b2,
                 "phase B2: null pointer");
  else
    SDMI_REQUIRE(a->dqkv && a->tok && a->st1 && a->ln1_g && a->dx1 && a->dtok && a->dhgn && a->ln1_part && a->wstream_a,
                 "phase A: null pointer");
  hipStream_t st = (hipStream_t)stream;
  if (rows == 64) return a->C == 256 ? st_train_bwd_launch<256, 4>(*a, st) : st_train_bwd_launch<384, 4>(*a, st);
  return a->C == 256 ? st_train_bwd_launch<256, 2>(*a, st) : st_train_bwd_launch<384, 2>(*a, st);
}
