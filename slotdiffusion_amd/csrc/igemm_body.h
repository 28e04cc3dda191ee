// Implicit-GEMM body shared by igemm.hip (forward / data-gradient launches) and bwd_pair.hip (a data
// gradient and the weight gradient of one layer in ONE launch): operand loaders, MFMA loop, epilogues.
// See igemm.hip for the tiling / wave-specialisation notes.
#pragma once
#include "common.h"

namespace {

// Out-of-image / K-tail / out-of-range operand vectors are FETCHED from this zero line instead of
// being zeroed with per-dword selects after the load (pointer select = 2 VALU ops per vector).
__device__ uint4 g_zero_line[8];

struct ConvGeom {
  int H, W, Cin, HoWo, Wo, KH, KW, stride, pad_t, pad_l, ups;
};

// Streamlined epilogue of a wave block that lies entirely inside the output (the common case): bf16
// out, alpha / bias / per-image row vector / residual / activation.  Measured on MI355X: with
// every global load, LDS write and MFMA switched off, a [16384 x 2048 x 256] linear layer still spent 21 of
// its 52 us in the generic epilogue -- 64-bit address arithmetic, a bounds test and an exec-mask branch
// around each of a lane's 64 two-byte stores.  Here a row's byte offset is wave-uniform and travels in the
// SGPR offset of a buffer store, the lane part (its column) is one VGPR per column tile: no per-element
// address arithmetic, no predication (edge blocks take the generic path).  The column-per-lane layout
// stays: a store instruction writes two 64-byte row segments; the row-per-lane alternative (MFMA operands
// swapped, 8-byte vector stores) touches 32 cache lines per instruction and measured 17 % slower.
template <int TM, int TN>
__device__ __forceinline__ void wave_epilogue_fast(const SdmiGemmArgs& p, f32x16 (&acc)[TM][TN], int mw0,
                                                   int nw0, int zb, int hw_shift, int lane) {
  const int col_l = lane & 31, row_l = (lane >> 5) * 4;
  const int mwu = __builtin_amdgcn_readfirstlane(mw0);          // wave-uniform by construction
  const unsigned ldc2 = (unsigned)p.ldc * 2u, ldr2 = (unsigned)p.ldr * 2u;
  // Sub-sampled placement (sdmi.h: osy; epilogue_fast_ok admits Ho * Wo = 2^k with Wo = 4 or a power of two >= 8):
  // rows 8g .. 8g+7 of a tile are eight consecutive pixels of one output row -- or two rows of four -- so a row's
  // offset is still (wave-uniform pixel of row 8g) + (r & 3) pixel steps, and the half-waves sit four pixel steps
  // (or one row step) apart.  The residual shares the output's layout (ldr = ldc).
  const bool sub = p.osy > 0;
  const unsigned step2 = sub ? (unsigned)p.osx * ldc2 : ldc2;                      // bytes between rows m, m + 1
  const unsigned half2 = sub ? (p.Wo >= 8 ? 4u * (unsigned)p.osx : (unsigned)(p.osy * p.oW)) * ldc2 : 4u * ldc2;
  const int wo_shift = sub ? 31 - __builtin_clz((unsigned)p.Wo) : 0;
  bf16_t* ob = (bf16_t*)p.out + (long long)zb * p.sc;
  const __amdgpu_buffer_rsrc_t rsO = __builtin_amdgcn_make_buffer_rsrc((void*)ob, 0, 0x7fffffff, 0x00020000);
  const bool has_res = p.residual != nullptr;
  const bf16_t* rbp = has_res ? (const bf16_t*)p.residual + (long long)zb * p.sr : ob;
  const __amdgpu_buffer_rsrc_t rsR = __builtin_amdgcn_make_buffer_rsrc((void*)rbp, 0, 0x7fffffff, 0x00020000);
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int mrow = mwu + i * 32;                               // uniform: first row of this row tile
    unsigned g2[4];                                              // byte offset of row mrow + 8g (uniform)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      if (sub) {
        const int m8 = mrow + 8 * g;
        const int b = m8 >> hw_shift, rem = m8 & ((1 << hw_shift) - 1);
        const int oy = rem >> wo_shift, ox = rem & (p.Wo - 1);
        g2[g] = (unsigned)((b * p.oH + oy * p.osy + p.ooy) * p.oW + ox * p.osx + p.oox) * ldc2;
      } else {
        g2[g] = (unsigned)(mrow + 8 * g) * ldc2;
      }
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int n = nw0 + j * 32 + col_l;
      const float bn = p.bias ? p.bias[n] : 0.f;
      float add[16];
      if (p.rowvec) {
        // rows 8g .. 8g+7 of the tile belong to one image (Ho*Wo is a power of two >= 8)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int b = (mrow + 8 * g) >> hw_shift;              // uniform
          const float rv = p.rowvec[(long long)b * p.ldrv + n] + bn;
#pragma unroll
          for (int q = 0; q < 4; ++q) add[4 * g + q] = rv;
        }
      } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) add[r] = bn;
      }
      const int vo_c = (int)((unsigned)(row_l >> 2) * half2 + (unsigned)n * 2u);   // lane part of the offsets
      if (has_res) {
        const int vo_r = sub ? vo_c : (int)((unsigned)row_l * ldr2 + (unsigned)n * 2u);
        unsigned short rr[16];
#pragma unroll
        for (int r = 0; r < 16; ++r)
          rr[r] = __builtin_amdgcn_raw_buffer_load_b16(
              rsR, vo_r, sub ? (int)(g2[r >> 2] + (unsigned)(r & 3) * step2)
                             : (int)((unsigned)(mrow + (r & 3) + 8 * (r >> 2)) * ldr2), 0);
#pragma unroll
        for (int r = 0; r < 16; ++r) add[r] += bf16_to_f32(rr[r]);
      }
      float v[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) v[r] = acc[i][j][r] * p.alpha + add[r];
      if (p.act == SDMI_ACT_SILU) {
#pragma unroll
        for (int r = 0; r < 16; ++r) v[r] = act_apply<true>(v[r], SDMI_ACT_SILU);
      } else if (p.act == SDMI_ACT_RELU) {
#pragma unroll
        for (int r = 0; r < 16; ++r) v[r] = fmaxf(v[r], 0.f);
      } else if (p.act == SDMI_ACT_GELU) {
#pragma unroll
        for (int r = 0; r < 16; ++r) v[r] = act_apply<true>(v[r], SDMI_ACT_GELU);
      }
      if (p.gn_part) {
        // GroupNorm statistics of this 32 x 32 tile (sdmi.h: gn_part): sums of the ROUNDED values over the lane's 16
        // rows, the other half-wave's 16, then the cpg lanes (columns) of a group; one lane per group stores them
        float gs = 0.f, gq = 0.f;
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
          const uint32_t pk = f32x2_to_bf16x2(v[r], v[r + 1]);
          const float lo = __uint_as_float(pk << 16), hi = __uint_as_float(pk & 0xffff0000u);
          gs += lo + hi;
          gq = fmaf(lo, lo, fmaf(hi, hi, gq));
        }
        gs += __shfl_xor(gs, 32, 64);
        gq += __shfl_xor(gq, 32, 64);
        const int cpg = p.N / p.gn_groups;
        for (int o = 1; o < cpg; o <<= 1) {
          gs += __shfl_xor(gs, o, 64);
          gq += __shfl_xor(gq, o, 64);
        }
        if (lane < 32 && (col_l & (cpg - 1)) == 0) {
          const int b = mrow >> hw_shift, blk = (mrow & ((1 << hw_shift) - 1)) >> 5;
          float* dst = p.gn_part + ((((long long)b << (hw_shift - 5)) + blk) * p.gn_groups + n / cpg) * 2;
          dst[0] = gs;
          dst[1] = gq;
        }
      }
#pragma unroll
      for (int r = 0; r < 16; r += 2) {
        const uint32_t pk = f32x2_to_bf16x2(v[r], v[r + 1]);
        __builtin_amdgcn_raw_buffer_store_b16((unsigned short)(pk & 0xffffu), rsO, vo_c,
                                              (int)(g2[r >> 2] + (unsigned)(r & 3) * step2), 0);
        __builtin_amdgcn_raw_buffer_store_b16((unsigned short)(pk >> 16), rsO, vo_c,
                                              (int)(g2[(r + 1) >> 2] + (unsigned)((r + 1) & 3) * step2), 0);
      }
    }
  }
}

// The streamlined epilogue serves: interior blocks, bf16 out, offsets within 31 bits; plain placement, or a
// sub-sampled one whose 8-row groups are whole pixel runs (see wave_epilogue_fast).
template <int TM, int TN>
__device__ __forceinline__ bool epilogue_fast_ok(const SdmiGemmArgs& p, int mw0, int nw0, int hw_shift) {
  if (!(p.split_k <= 1 && p.out_dtype == SDMI_BF16 && !p.bias_m && mw0 + TM * 32 <= p.M && nw0 + TN * 32 <= p.N &&
        (!p.rowvec || hw_shift >= 3)))
    return false;
  if (p.osy > 0)
    return hw_shift >= 3 && (p.Wo & (p.Wo - 1)) == 0 && p.Wo >= 4 && (!p.residual || p.ldr == p.ldc) &&
           (long long)p.B * p.oH * p.oW * p.ldc < (1ll << 30);
  return (long long)p.M * (p.ldc > p.ldr ? p.ldc : p.ldr) < (1ll << 30);
}

// Epilogue of one MFMA wave's (TM*32)x(TN*32) accumulator block whose top-left output element is
// (mw0, nw0): split-K partial store, or alpha / bias / per-image row vector / residual / activation
// and the typed store.
template <int TM, int TN>
__device__ __forceinline__ void wave_epilogue(const SdmiGemmArgs& p, f32x16 (&acc)[TM][TN], int mw0,
                                              int nw0, int zb, int hw_shift, int lane, int by) {
  // ---- epilogue. C/D layout of the 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
  // the common case takes the streamlined path: plain placement, bf16 out, the whole wave block inside
  // the output, row offsets within 31 bits
  if (epilogue_fast_ok<TM, TN>(p, mw0, nw0, hw_shift)) {
    wave_epilogue_fast<TM, TN>(p, acc, mw0, nw0, zb, hw_shift, lane);
    return;
  }
  const int col_l = lane & 31;
  const int row_l = (lane >> 5) * 4;
  if (p.split_k > 1) {
    float* ws = p.workspace + ((long long)by) * p.M * p.N;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int n = nw0 + j * 32 + col_l;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m = mw0 + i * 32 + (r & 3) + 8 * (r >> 2) + row_l;
          if (m < p.M && n < p.N) ws[(long long)m * p.N + n] = acc[i][j][r];
        }
      }
    return;
  }
  if (p.osy > 0) {
    // sub-sampled output placement (data gradient of a strided conv, one parity per launch):
    // plain alpha * acc (+ bias), row m = (b, oy, ox) -> pixel (oy*osy + ooy, ox*osx + oox)
    const int HoWo_ = p.Ho * p.Wo;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int n = nw0 + j * 32 + col_l;
        const float bn = (p.bias && n < p.N) ? p.bias[n] : 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m = mw0 + i * 32 + (r & 3) + 8 * (r >> 2) + row_l;
          if (m < p.M && n < p.N) {
            const int b = hw_shift >= 0 ? (m >> hw_shift) : (m / HoWo_);
            const int rem = m - b * HoWo_;
            const int oy = rem / p.Wo, ox = rem - oy * p.Wo;
            const long long o = (((long long)b * p.oH + oy * p.osy + p.ooy) * p.oW + ox * p.osx + p.oox) *
                                    p.ldc + n;
            float v = acc[i][j][r] * p.alpha + bn;
            if (p.residual)             // same layout as the output (gradient of another consumer)
              v += p.out_dtype == SDMI_BF16 ? bf16_to_f32(((const bf16_t*)p.residual)[o])
                                            : ((const float*)p.residual)[o];
            if (p.out_dtype == SDMI_BF16) ((bf16_t*)p.out)[o] = f32_to_bf16(v);
            else ((float*)p.out)[o] = v;
          }
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
      const int n = nw0 + j * 32 + col_l;
      const bool n_ok = n < p.N;
      const float bn = (p.bias && !p.bias_m && n_ok) ? p.bias[n] : 0.f;
      const int mbase = mw0 + i * 32 + row_l;
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
        if (out_bf16) {
#pragma unroll
          for (int r = 0; r < 16; ++r) v[r] = act_apply<true>(v[r], SDMI_ACT_SILU);
        } else {
#pragma unroll
          for (int r = 0; r < 16; ++r) v[r] = act_apply(v[r], SDMI_ACT_SILU);
        }
      } else if (p.act == SDMI_ACT_RELU) {
#pragma unroll
        for (int r = 0; r < 16; ++r) v[r] = fmaxf(v[r], 0.f);
      } else if (p.act == SDMI_ACT_GELU) {
        if (out_bf16) {
#pragma unroll
          for (int r = 0; r < 16; ++r) v[r] = act_apply<true>(v[r], SDMI_ACT_GELU);
        } else {
#pragma unroll
          for (int r = 0; r < 16; ++r) v[r] = act_apply(v[r], SDMI_ACT_GELU);
        }
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


// Fused LayerNorm-fold / GEGLU epilogue of one MFMA wave's block (sdmi.h: ln_colsum, geglu; 1x1 /
// linear problems only).  EPI bit 0: the GEMM ran on raw rows with gamma-scaled weights, sx / sxx are
// this lane's partial sums (its 16 bytes of every 32-byte k-step) of row (lane & 31) of row tile i:
//   out = rstd * (acc - mean * colsum[n]) + bias[n]
// EPI bit 1: column tiles j < TN/2 hold value columns, j + TN/2 the matching gate columns:
//   out[m][n] = value * gelu(gate).
typedef __bf16 sdmi_bf16x2 __attribute__((ext_vector_type(2)));

template <int TM, int TN, int EPI>
__device__ __forceinline__ void fused_epilogue(const SdmiGemmArgs& p, f32x16 (&acc)[TM][TN],
                                               const float (&sx)[TM], const float (&sxx)[TM], int mw0,
                                               int nw0, int lane, int zb = 0) {
  constexpr bool LNF = (EPI & 1) != 0, GEGLU = (EPI & 2) != 0, SM8 = (EPI & 4) != 0;
  const float* bias_p = p.bias ? p.bias + (long long)zb * p.s_bias : nullptr;
  const float* colsum_p = p.ln_colsum ? p.ln_colsum + (long long)zb * p.s_colsum : nullptr;
  const long long zc = (long long)zb * p.sc;
  constexpr int NOUT = GEGLU ? TN / 2 : TN;
  // C layout of the 32x32 MFMA: col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
  const int col_l = lane & 31, row_l = (lane >> 5) * 4;
  const bool out_bf16 = p.out_dtype == SDMI_BF16;
  // streamlined stores for interior blocks (see wave_epilogue_fast)
  const int mwu = __builtin_amdgcn_readfirstlane(mw0);
  const unsigned ldc2 = (unsigned)p.ldc * 2u;
  const bool fast = out_bf16 && mw0 + TM * 32 <= p.M && nw0 + NOUT * 32 <= p.N &&
                    (long long)p.M * p.ldc < (1ll << 30);
  const __amdgpu_buffer_rsrc_t rsO =
      __builtin_amdgcn_make_buffer_rsrc((void*)((bf16_t*)p.out + zc), 0, 0x7fffffff, 0x00020000);
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    float mean = 0.f, rstd = 1.f;
    if constexpr (LNF) {
      const float tot = sx[i] + __shfl_xor(sx[i], 32, 64), tot2 = sxx[i] + __shfl_xor(sxx[i], 32, 64);
      const float inv_k = 1.f / (float)p.K;
      mean = tot * inv_k;
      rstd = rsqrtf(fmaxf(tot2 * inv_k - mean * mean, 0.f) + p.ln_eps);      // lane l: row l & 31
    }
    // per-column epilogue operands first (NOUT columns per lane), then row by row: the row's
    // statistics are fetched from their owner lane when needed (no 32-register staging arrays)
    float c0[NOUT], s0[NOUT], c1[NOUT], s1[NOUT];
    int ncol[NOUT];
#pragma unroll
    for (int j = 0; j < NOUT; ++j) {
      const int n = nw0 + j * 32 + col_l;
      ncol[j] = n;
      const int nc = n < p.N ? n : p.N - 1;
      c0[j] = bias_p ? bias_p[nc] : 0.f;
      s0[j] = LNF ? colsum_p[nc] : 0.f;
      c1[j] = (GEGLU && bias_p) ? bias_p[p.N + nc] : 0.f;
      s1[j] = (GEGLU && LNF) ? colsum_p[p.N + nc] : 0.f;
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int rr = (r & 3) + 8 * (r >> 2) + row_l;
      const float mu = LNF ? __shfl(mean, rr, 64) : 0.f;
      const float rs = LNF ? __shfl(rstd, rr, 64) : 1.f;
      const int m = mw0 + i * 32 + rr;
#pragma unroll
      for (int j = 0; j < NOUT; ++j) {
        float v = rs * (acc[i][j][r] * p.alpha - mu * s0[j]) + c0[j];
        if constexpr (GEGLU) {
          float g = rs * (acc[i][j + TN / 2][r] * p.alpha - mu * s1[j]) + c1[j];
          if (p.out2) {          // keep the pre-activation (training) and continue from its rounding
            if (out_bf16) {
              const bf16_t vb = f32_to_bf16(v), gb = f32_to_bf16(g);
              v = bf16_to_f32(vb);
              g = bf16_to_f32(gb);
              if (ncol[j] < p.N && m < p.M) {
                bf16_t* h = (bf16_t*)p.out2 + (long long)m * p.ldc2 + ncol[j];
                h[0] = vb;
                h[p.N] = gb;
              }
            } else if (ncol[j] < p.N && m < p.M) {
              float* h = (float*)p.out2 + (long long)m * p.ldc2 + ncol[j];
              h[0] = v;
              h[p.N] = g;
            }
          }
          if (out_bf16) v *= act_apply<true>(g, SDMI_ACT_GELU);
          else v *= act_apply(g, SDMI_ACT_GELU);
        } else if constexpr (SM8) {
          // softmax over the aligned 8-column (16-column from 9 slots) group (softmax8 slot scores + pad columns):
          // the group's lanes are neighbours of this 32-lane half, every lane takes part
          const bool wide = p.softmax8 > 8;
          const float x = ((ncol[j] & (wide ? 15 : 7)) >= p.softmax8) ? -INFINITY : v;
          float mx = x;
          mx = fmaxf(mx, __shfl_xor(mx, 1, 64));
          mx = fmaxf(mx, __shfl_xor(mx, 2, 64));
          mx = fmaxf(mx, __shfl_xor(mx, 4, 64));
          if (wide) mx = fmaxf(mx, __shfl_xor(mx, 8, 64));
          const float e = __expf(x - mx);
          float sm = e;
          sm += __shfl_xor(sm, 1, 64);
          sm += __shfl_xor(sm, 2, 64);
          sm += __shfl_xor(sm, 4, 64);
          if (wide) sm += __shfl_xor(sm, 8, 64);
          v = e / sm;
        } else {
          v = act_apply(v, p.act);
        }
        if (fast) {
          // interior block, bf16 out: the row's byte offset is wave-uniform (SGPR offset of the buffer store),
          // the lane part is fixed per column tile -- no 64-bit address, bounds test or branch per element
          __builtin_amdgcn_raw_buffer_store_b16(
              f32_to_bf16(v), rsO, (int)((unsigned)row_l * ldc2 + (unsigned)ncol[j] * 2u),
              (int)((unsigned)(mwu + i * 32 + (r & 3) + 8 * (r >> 2)) * ldc2), 0);
        } else if (ncol[j] < p.N && m < p.M) {
          const long long o = zc + (long long)m * p.ldc + ncol[j];
          if (out_bf16) ((bf16_t*)p.out)[o] = f32_to_bf16(v);
          else ((float*)p.out)[o] = v;
        }
      }
    }
  }
}

template <typename T, int BM, int BN, int BKB, int MODE, int EPI = 0, bool XS = false>
__device__ __forceinline__ void igemm_body(const SdmiGemmArgs& p, int tiles_m, int tiles_n,
                                           int kt_per_split, int hw_shift, int bid, int nblk, int by) {
  // bid / nblk / by: this workgroup's index, the number of workgroups walking the tiles, and the
  // (batch, split-K) index -- block index x / grid size x / block index y in a plain launch; bid keeps
  // the launch's block index modulo 8 (its XCD) when the grid hosts other work in front of these blocks
  // MODE: 0 = general gather (nearest-x2 fold, zero insertion, K tiles that straddle filter taps),
  //       1 = 1x1 / linear, 2 = plain convolution whose K tiles lie inside one filter tap
  //       (Cin % BK == 0): the tap (kh, kw) is wave-uniform, so the per-vector work shrinks to two
  //       adds, two compares and one 64-bit multiply-add
  constexpr bool IS1X1 = MODE == 1;
  constexpr bool TAPU = MODE == 2;
  constexpr bool LNF = (EPI & 1) != 0, GEGLU = (EPI & 2) != 0;
  static_assert(EPI == 0 || MODE == 1, "fused LayerNorm / GEGLU epilogues: 1x1 / linear only");
  static_assert(!GEGLU || (BN / 2) % 64 == 0, "GEGLU pairs value and gate tiles inside a wave");
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
  constexpr int MTH_ROWS = 256 / VPR;         // rows covered by one pass of the 256 loader threads

  extern __shared__ __attribute__((aligned(16))) char smem[];

  // ---- persistent workgroups: this one walks tiles bid, + nblk, ... (nblk is a
  // multiple of 8 whenever it is smaller than the tile count, so a virtual block id keeps its
  // XCD).  XCD-aware id: each XCD gets a contiguous range of tile ids.
  const int nwg = tiles_m * tiles_n;
  auto tile_of = [&](int vb, int& m0, int& n0) __attribute__((always_inline)) {
    const int xcd = vb & 7, q = nwg >> 3, r = nwg & 7;
    const int id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (vb >> 3);
    const int tm = id / tiles_n;
    m0 = tm * BM;
    n0 = (id - tm * tiles_n) * (GEGLU ? BN / 2 : BN);     // GEGLU: BN/2 output columns per tile
  };
  const int my_tiles = (bid < nwg) ? (nwg - 1 - bid) / nblk + 1 : 0;
  const int zb = by / p.split_k;
  const int ksplit = by - zb * p.split_k;

  const int nk_total = (p.K + BK - 1) / BK;
  const int kt_begin = ksplit * kt_per_split;
  int kt_end = kt_begin + kt_per_split;
  if (kt_end > nk_total) kt_end = nk_total;
  const int n_kt = kt_end > kt_begin ? kt_end - kt_begin : 0;
  const int total = my_tiles * n_kt;          // pipeline steps of this workgroup

  if (threadIdx.x >= 256) {
    // =============================== loader waves ===============================
    const int tid = threadIdx.x - 256;
    const T* __restrict__ Ag = (const T*)p.a + (long long)zb * p.sa;
    const T* __restrict__ Wg = (const T*)p.w + (long long)zb * p.sw;
    const int kc = tid % VPR;  // vector column inside the K tile (same for all of a thread's vectors)
    const T* zero_src = reinterpret_cast<const T*>(g_zero_line);
    int ld_tile = 0, ld_kt = 0;               // (tile, K tile) the next load_tile() fetches
    // Rows m >= M / n >= N only feed output rows / columns that are never stored, so they are
    // CLAMPED to the last valid row instead of being zero-filled.
    //
    // MODE 1 / 2 loaders are scalar-only in the steady state (VALU work of a loader wave and the
    // MFMAs of the wave next to it on the SIMD serialise -- tools/probes/ldsdma.hip): operands are
    // fetched with buffer loads whose per-lane byte offset (voffset) is computed once per output
    // tile, the walk over K is a wave-uniform SGPR offset; image borders / the K tail are
    // out-of-range voffsets (the buffer returns zeros), a filter tap's validity mask is applied
    // once per tap.  The activation base is biased by -(pad_t*W + pad_l) pixels: offsets >= 0.
    constexpr unsigned OOB = 0x80000000u;       // == num_records
    const T* Abase = Ag;
    if (TAPU) Abase -= (long long)(p.pad_t * p.W + p.pad_l) * p.lda;
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)Abase, 0, (int)OOB, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc((void*)Wg, 0, (int)OOB, 0x00020000);
    // extra A sources appended along K (sdmi.h: a2 / a3): 1x1 taps at the output pixel
    constexpr bool XSRC = XS && (IS1X1 || TAPU);      // (its own instantiation: 8 more loader VGPRs)
    // (their buffer descriptors are built per K tile from the pointer: three more live descriptors
    // pushed the loaders' scalar state into scratch)
    unsigned a_vo2[XSRC ? A_VECS : 1], a_vo3[XSRC ? A_VECS : 1];
    unsigned a_vo[A_VECS], a_cur[A_VECS], a_inv[A_VECS], b_vo[B_VECS], b_cur[B_VECS];
    int k0 = 0, ci = 0, kh = 0, kw = 0;       // wave-uniform k state of the next K tile (MODE 1/2)
    // MODE 0 (general gather) state
    int kk = 0;
    int a_pix[A_VECS], a_iy0[A_VECS], a_ix0[A_VECS];
    int b_row[B_VECS];
    auto begin_tile = [&]() __attribute__((always_inline)) {
      int m0, n0;
      tile_of(bid + ld_tile * nblk, m0, n0);
      k0 = kt_begin * BK;
      if (TAPU) {               // uniform: derived from the K tile index only
        const int tap = k0 / p.Cin;
        ci = k0 - tap * p.Cin;  // channel base of the K tile
        kh = tap / p.KW;
        kw = tap - kh * p.KW;
      } else if (!IS1X1) {
        kk = k0 + kc * VEC;
        const int tap = kk / p.Cin;
        ci = kk - tap * p.Cin;
        kh = tap / p.KW;
        kw = tap - kh * p.KW;
      }
#pragma unroll
      for (int i = 0; i < A_VECS; ++i) {
        const int row = (tid + i * 256) / VPR;
        const int m = min(m0 + row, p.M - 1);
        if constexpr (XSRC) {      // (stride-1 "same" convolutions: output pixel m = input pixel m)
          a_vo2[i] = ((unsigned)m * (unsigned)p.lda2 + kc * VEC) * (unsigned)sizeof(T);
          a_vo3[i] = ((unsigned)m * (unsigned)p.lda3 + kc * VEC) * (unsigned)sizeof(T);
        }
        if (IS1X1) {
          a_vo[i] = ((unsigned)m * (unsigned)p.lda + kc * VEC) * (unsigned)sizeof(T);
          a_inv[i] = 0;
          a_cur[i] = a_vo[i];
        } else {
          const int HoWo = p.Ho * p.Wo;
          // (power-of-two images -- every UNet / encoder level -- take shifts: an integer division is
          // ~35 instructions, and this runs per operand vector and output tile)
          const bool wo2 = (p.Wo & (p.Wo - 1)) == 0;
          const int b = hw_shift >= 0 ? (m >> hw_shift) : m / HoWo;
          const int rem = m - b * HoWo;
          const int oy = wo2 ? (rem >> (31 - __builtin_clz(p.Wo))) : rem / p.Wo;
          const int ox = rem - oy * p.Wo;
          const int iy0 = oy * p.stride - p.pad_t, ix0 = ox * p.stride - p.pad_l;
          if (TAPU) {
            a_vo[i] = ((unsigned)((b * p.H + oy * p.stride) * p.W + ox * p.stride) * (unsigned)p.lda +
                       kc * VEC) * (unsigned)sizeof(T);
            unsigned rb = 0, cb = 0, inv = 0;   // bad rows / columns of the filter window
            for (int q = 0; q < p.KH; ++q) rb |= ((unsigned)(iy0 + q) < (unsigned)p.H ? 0u : 1u) << q;
            for (int q = 0; q < p.KW; ++q) cb |= ((unsigned)(ix0 + q) < (unsigned)p.W ? 0u : 1u) << q;
            for (int q = 0; q < p.KH; ++q)
              inv |= (((rb >> q) & 1u) ? ((1u << p.KW) - 1u) : cb) << (q * p.KW);
            a_inv[i] = inv;
            a_cur[i] = a_vo[i];
          } else {
            a_iy0[i] = iy0;
            a_ix0[i] = ix0;
            a_pix[i] = b * p.H * p.W;
          }
        }
      }
#pragma unroll
      for (int i = 0; i < B_VECS; ++i) {
        const int row = (tid + i * 256) / VPR;
        int n;
        if constexpr (GEGLU) {
          // B-tile row -> weight row: each wave's BN/2 rows are [value columns | their gate columns]
          constexpr int HW_ = BN / 4;                       // output columns per wave
          const int wnr = row / (BN / 2), rem = row - wnr * (BN / 2);
          const int jg = rem / HW_, c = rem - jg * HW_;
          n = jg * p.N + min(n0 + wnr * HW_ + c, p.N - 1);
        } else {
          n = min(n0 + row, p.N - 1);
        }
        b_row[i] = n;
        b_vo[i] = ((unsigned)n * (unsigned)p.ldw + kc * VEC) * (unsigned)sizeof(T);
        b_cur[i] = b_vo[i];
      }
    };
    // Loads are unconditional (all of a tile's global loads in flight together): border / K-tail
    // vectors read zeros (out-of-range buffer offset, or the zero line in MODE 0).
    auto load_tile = [&](u32x4 (&ra)[A_VECS], u32x4 (&rb)[B_VECS])
                         __attribute__((always_inline)) {
      if (ld_kt == 0) begin_tile();
      if constexpr (TAPU || IS1X1) {
        unsigned so_a;
        if constexpr (TAPU) {
          if (ci == 0 || ld_kt == 0) {             // new filter tap: apply its validity mask
            const int tap = kh * p.KW + kw;
#pragma unroll
            for (int i = 0; i < A_VECS; ++i) a_cur[i] = ((a_inv[i] >> tap) & 1u) ? OOB : a_vo[i];
          }
          so_a = (unsigned)((kh * p.W + kw) * p.lda + ci) * (unsigned)sizeof(T);
        } else {
          if (k0 + BK > p.K) {                     // K tail (last K tile of an output tile only)
            const bool k_ok = k0 + kc * VEC < p.K;
#pragma unroll
            for (int i = 0; i < A_VECS; ++i) a_cur[i] = k_ok ? a_vo[i] : OOB;
#pragma unroll
            for (int i = 0; i < B_VECS; ++i) b_cur[i] = k_ok ? b_vo[i] : OOB;
          }
          so_a = (unsigned)k0 * (unsigned)sizeof(T);
        }
        const unsigned so_b = (unsigned)k0 * (unsigned)sizeof(T);
        // which source this K tile reads (wave-uniform): selects, not branches -- separate load loops per
        // source pushed the loaders' register arrays into scratch
        const bool s1 = XSRC && p.a2 != nullptr && k0 >= p.K1;
        const bool s2 = s1 && p.a3 != nullptr && k0 >= p.K2;
        const unsigned so_x = s2 ? (unsigned)(k0 - p.K2) * (unsigned)sizeof(T)
                                 : (s1 ? (unsigned)(k0 - p.K1) * (unsigned)sizeof(T) : so_a);
        const void* base_x = s2 ? p.a3 : (s1 ? p.a2 : (const void*)Abase);
        const __amdgpu_buffer_rsrc_t rs_x =
            XSRC ? __builtin_amdgcn_make_buffer_rsrc((void*)base_x, 0, (int)OOB, 0x00020000) : rsA;
#pragma unroll
        for (int i = 0; i < A_VECS; ++i) {
          const unsigned vo = s2 ? a_vo3[XSRC ? i : 0] : (s1 ? a_vo2[XSRC ? i : 0] : a_cur[i]);
          ra[i] = __builtin_amdgcn_raw_buffer_load_b128(rs_x, (int)vo, (int)so_x, 0);
        }
#pragma unroll
        for (int i = 0; i < B_VECS; ++i)
          rb[i] = __builtin_amdgcn_raw_buffer_load_b128(rsW, (int)b_cur[i], (int)so_b, 0);
        if constexpr (TAPU) {
          ci += BK;
          if (ci == p.Cin) {
            ci = 0;
            if (++kw == p.KW) { kw = 0; ++kh; }
          }
        }
      } else {
        const bool k_ok = kk < p.K;
#pragma unroll
        for (int i = 0; i < A_VECS; ++i) {
          bool ok = k_ok;
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
          const long long off = (long long)(a_pix[i] + iy * p.W + ix) * p.lda + ci;
          ra[i] = *reinterpret_cast<const u32x4*>(ok ? Ag + off : zero_src);
        }
#pragma unroll
        for (int i = 0; i < B_VECS; ++i)
          rb[i] = *reinterpret_cast<const u32x4*>(
              k_ok ? Wg + ((long long)b_row[i] * p.ldw + kk) : zero_src);
        kk += BK;
        ci += BK;
        while (ci >= p.Cin) {
          ci -= p.Cin;
          if (++kw == p.KW) { kw = 0; ++kh; }
        }
      }
      k0 += BK;
      if (++ld_kt == n_kt) { ld_kt = 0; ++ld_tile; }
    };
    // LDS destinations: this thread's first row + compile-time row strides (immediate offsets)
    constexpr int RSTEP = MTH_ROWS;
    char* const st_a = smem + (tid / VPR) * ROWB + kc * 16;
    char* const st_b = st_a + BM * ROWB;
    auto store_tile = [&](int stage_off, const u32x4 (&ra)[A_VECS], const u32x4 (&rb)[B_VECS])
                          __attribute__((always_inline)) {
#pragma unroll
      for (int i = 0; i < A_VECS; ++i)
        *reinterpret_cast<u32x4*>(st_a + stage_off + i * RSTEP * ROWB) = ra[i];
#pragma unroll
      for (int i = 0; i < B_VECS; ++i)
        *reinterpret_cast<u32x4*>(st_b + stage_off + i * RSTEP * ROWB) = rb[i];
    };

    // flat pipeline over (output tile, K tile) steps: while the MFMA waves finish a tile and run
    // its epilogue, the first K tiles of the next one are already staged / in flight.  DEPTH K tiles
    // of global loads are in flight in registers (16 vectors per thread: 2 tiles of the 128 x 128 x
    // 128-byte configuration, 4 - 8 of the small-tile ones, whose launches are bound by the memory
    // latency of each K step rather than by bandwidth); the LDS image stays double buffered.
    constexpr int DEPTH_ = 16 / (A_VECS + B_VECS);
    constexpr int DEPTH = DEPTH_ >= 8 ? 8 : (DEPTH_ >= 4 ? 4 : 2);
    u32x4 ra[DEPTH][A_VECS], rb[DEPTH][B_VECS];
#pragma unroll
    for (int d = 0; d < DEPTH; ++d)
      if (total > d) load_tile(ra[d], rb[d]);
    for (int g = 0; g < total; g += DEPTH) {
#pragma unroll
      for (int d = 0; d < DEPTH; ++d) {
        if (g + d < total) {
          store_tile((d & 1) * BUF_BYTES, ra[d], rb[d]);
          if (g + d + DEPTH < total) load_tile(ra[d], rb[d]);
          __syncthreads();
        }
      }
    }
    return;
  }

  // ================================= MFMA waves =================================
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;

  const int frag_row = lane & 31;
  const int frag_kb = (lane >> 5) * 16;
  auto read_frags = [&](const char* As, const char* Bs, int ks, u32x4 (&fa)[TM], u32x4 (&fb)[TN])
                        __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < TM; ++i)
      fa[i] = *reinterpret_cast<const u32x4*>(As + i * 32 * ROWB + ks * 32);
#pragma unroll
    for (int j = 0; j < TN; ++j)
      fb[j] = *reinterpret_cast<const u32x4*>(Bs + j * 32 * ROWB + ks * 32);
  };
  int g = 0;                          // pipeline step (stage = g & 1)
  for (int ti = 0; ti < my_tiles; ++ti) {
  int m0, n0;
  tile_of(bid + ti * nblk, m0, n0);
  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  float sx[TM], sxx[TM];                 // LayerNorm fold: row sums taken from the A fragments
#pragma unroll
  for (int i = 0; i < TM; ++i) sx[i] = sxx[i] = 0.f;
  for (int t = 0; t < n_kt; ++t, ++g) {
    __syncthreads();                 // stage g & 1 holds this K tile
    const char* base = smem + (g & 1) * BUF_BYTES;
    const char* As = base + (wm * WTM + frag_row) * ROWB + frag_kb;
    const char* Bs = base + BM * ROWB + (wn * WTN + frag_row) * ROWB + frag_kb;
    if constexpr (sizeof(T) == 1) {
      // fp8 (e4m3fn): one v_mfma_scale_f32_32x32x64_f8f6f4 (unit block scales) per 64 bytes of K
      // = two 32-byte k-steps; a lane's 32 operand bytes are its 16 bytes of each of the two
      // steps (the same K subset for A and B, so the contraction is unchanged)
      typedef int i32x8 __attribute__((ext_vector_type(8)));
#pragma unroll
      for (int kp = 0; kp < KSTEPS / 2; ++kp) {
        u32x4 fa0[TM], fa1[TM], fb0[TN], fb1[TN];
        read_frags(As, Bs, 2 * kp, fa0, fb0);
        read_frags(As, Bs, 2 * kp + 1, fa1, fb1);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j) {
            const i32x8 a8 = {(int)fa0[i].x, (int)fa0[i].y, (int)fa0[i].z, (int)fa0[i].w,
                              (int)fa1[i].x, (int)fa1[i].y, (int)fa1[i].z, (int)fa1[i].w};
            const i32x8 b8 = {(int)fb0[j].x, (int)fb0[j].y, (int)fb0[j].z, (int)fb0[j].w,
                              (int)fb1[j].x, (int)fb1[j].y, (int)fb1[j].z, (int)fb1[j].w};
            acc[i][j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, b8, acc[i][j], 0, 0, 0, 127,
                                                                        0, 127);
          }
      }
    } else {
    // fragments double buffered, except in the fused-epilogue variants of the 128 x 128 tile: their
    // row-sum registers and epilogue operands do not fit next to 64 accumulators + two fragment
    // sets under the 128-VGPR budget (the spills cost more than the exposed LDS latency)
    constexpr int NFB = (EPI != 0 && TM * TN >= 4) ? 1 : 2;
    u32x4 fa[NFB][TM], fb[NFB][TN];
    if constexpr (NFB == 2) read_frags(As, Bs, 0, fa[0], fb[0]);
#pragma unroll
    for (int ks = 0; ks < KSTEPS; ++ks) {
      if constexpr (NFB == 2) {
        if (ks + 1 < KSTEPS) read_frags(As, Bs, ks + 1, fa[(ks + 1) & 1], fb[(ks + 1) & 1]);
      } else {
        read_frags(As, Bs, ks, fa[0], fb[0]);
      }
      __builtin_amdgcn_sched_barrier(0);   // keep the prefetch ahead of this k-step's MFMAs
      if constexpr (LNF) {
#pragma unroll
        for (int i = 0; i < TM; ++i) {
          const u32x4 a4 = fa[ks & (NFB - 1)][i];
          if constexpr (sizeof(T) == 2) {
            // (explicit lanes: indexing the vector inside an unrolled loop was miscompiled to lane 0)
            const sdmi_bf16x2 ones = __builtin_bit_cast(sdmi_bf16x2, 0x3F803F80u);
            const bf16x8 a8 = __builtin_bit_cast(bf16x8, a4);
            const sdmi_bf16x2 v0 = {a8[0], a8[1]}, v1 = {a8[2], a8[3]}, v2 = {a8[4], a8[5]},
                              v3 = {a8[6], a8[7]};
            sx[i] = __builtin_amdgcn_fdot2_f32_bf16(v0, ones, sx[i], false);
            sxx[i] = __builtin_amdgcn_fdot2_f32_bf16(v0, v0, sxx[i], false);
            sx[i] = __builtin_amdgcn_fdot2_f32_bf16(v1, ones, sx[i], false);
            sxx[i] = __builtin_amdgcn_fdot2_f32_bf16(v1, v1, sxx[i], false);
            sx[i] = __builtin_amdgcn_fdot2_f32_bf16(v2, ones, sx[i], false);
            sxx[i] = __builtin_amdgcn_fdot2_f32_bf16(v2, v2, sxx[i], false);
            sx[i] = __builtin_amdgcn_fdot2_f32_bf16(v3, ones, sx[i], false);
            sxx[i] = __builtin_amdgcn_fdot2_f32_bf16(v3, v3, sxx[i], false);
          } else {
            const float f0 = __uint_as_float(a4.x), f1 = __uint_as_float(a4.y),
                        f2 = __uint_as_float(a4.z), f3 = __uint_as_float(a4.w);
            sx[i] += (f0 + f1) + (f2 + f3);
            sxx[i] = fmaf(f0, f0, fmaf(f1, f1, fmaf(f2, f2, fmaf(f3, f3, sxx[i]))));
          }
        }
      }
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          const u32x4 a4 = fa[ks & (NFB - 1)][i], b4 = fb[ks & (NFB - 1)][j];
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
  }

  if constexpr (EPI == 0)
    wave_epilogue<TM, TN>(p, acc, m0 + wm * WTM, n0 + wn * WTN, zb, hw_shift, lane, by);
  else
    fused_epilogue<TM, TN, EPI>(p, acc, sx, sxx, m0 + wm * WTM, n0 + wn * (GEGLU ? WTN / 2 : WTN), lane, zb);
  }  // tiles of this workgroup
}

}  // namespace
