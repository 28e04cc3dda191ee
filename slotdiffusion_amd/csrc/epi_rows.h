// Row-major epilogue of a wave's 64 x 64 accumulator block (round 5; igemm_halo.h; conv3x3_c64_kernel keeps an un-instantiated ROWS form).
//
// The column-per-lane epilogue of rounds 1 - 4 (wave_epilogue_fast) stores 2 bytes per lane: a store instruction
// writes two 64-byte row segments, 64 instructions per 64 x 64 block -- measured 6.5 us per 256 x 128 output tile when
// nothing else runs on the CU (profiles/r05_pp_ablation.txt), 20 % of a 3x3 128 -> 128 convolution at 32^2.  Its
// direct row-per-lane alternative (8-byte stores of 4 consecutive columns) touches 32 cache lines per instruction and
// measured slower.  Here the kernels run their MFMAs with the operands SWAPPED (weights as the A
// operand), so an accumulator holds C^T:
//     lane l:  output row    m = mw0 + 32 i + (l & 31)
//     reg  r:  output column n = nw0 + 32 j + 8 (r >> 2) + 4 (l >> 5) + (r & 3)        -- four consecutive columns
// and the block goes through a WAVE-PRIVATE LDS patch: each lane rounds its 4-column groups to bf16 and writes them
// with `ds_write_b64` at [row][column] (pitch = row + 16 B: 16-byte aligned rows, two-way conflicts at worst), the wave
// reads the patch back as whole rows -- lane l: row 8 q + (l >> 3), 16-byte chunk l & 7 -- and stores with
// `buffer_store_dwordx4`: an instruction writes eight full 128-byte row segments.  No barrier (the patch is private to
// the wave; LDS operations of a wave complete in order), two rounds of 32 rows through a 4.6 KB patch.
// alpha / bias / per-image row vector / residual / activation happen in registers BEFORE the (single) rounding, in
// wave_epilogue_fast's order of operations: results are bit-identical to it.
#pragma once
#include "igemm_body.h"

namespace {

// NJ = 32-column blocks per wave (2: 64 x 64 wave blocks, 1: 64 x 32)
template <int NJ> struct EpiRows {
  static constexpr int PITCH = NJ * 64 + 16;           // bytes per staged row (NJ * 32 bf16 + 16 B)
  static constexpr int PATCH = 32 * PITCH;             // per wave
};
constexpr int EPI_ROWS_PATCH = EpiRows<2>::PATCH;

// interior 64 x 64 block, bf16 out, 16-byte aligned rows, one image per 64 rows when a per-image vector is added
template <int NJ = 2>
__device__ __forceinline__ bool epilogue_rows_ok(const SdmiGemmArgs& p, int mw0, int nw0, int hw_shift) {
  return p.split_k <= 1 && p.out_dtype == SDMI_BF16 && !p.bias_m && !p.gn_part && p.osy == 0 && mw0 + 64 <= p.M &&
         nw0 + 32 * NJ <= p.N && (p.ldc & 7) == 0 && ((uintptr_t)p.out & 15) == 0 && (!p.rowvec || hw_shift >= 6) &&
         (!p.residual || ((p.ldr & 3) == 0 && ((uintptr_t)p.residual & 7) == 0)) &&
         (!p.bias || ((uintptr_t)p.bias & 15) == 0) && (!p.rowvec || (((uintptr_t)p.rowvec & 15) == 0 && (p.ldrv & 3) == 0)) &&
         (long long)p.M * (p.ldc > p.ldr ? p.ldc : p.ldr) < (1ll << 30);
}

// acc[i][j]: C^T layout above.  patch: this wave's EPI_ROWS_PATCH bytes of LDS.  Ends with every load of its own
// retired (explicit vmcnt(0) behind them), so no compiler-inserted wait leaks into the caller's K loop.
template <int NJ>
__device__ __forceinline__ void wave_epilogue_rows(const SdmiGemmArgs& p, f32x16 (&acc)[2][NJ], int mw0, int nw0,
                                                   int hw_shift, int lane, char* patch) {
  constexpr int EPI_ROWS_PITCH = EpiRows<NJ>::PITCH;
  constexpr int CPR = 4 * NJ, RPI = 64 / CPR, NQ = 32 / RPI;     // 16-byte chunks per row, rows per store instruction
  const int mwu = __builtin_amdgcn_readfirstlane(mw0), nwu = __builtin_amdgcn_readfirstlane(nw0);
  const int ml = lane & 31, hh = lane >> 5;
  // ---- per-column terms: bias[n] (+ rowvec[b][n]: the wave's 64 rows lie in one image), four consecutive n per load
  f32x4 add[NJ][4];
#pragma unroll
  for (int j = 0; j < NJ; ++j)
#pragma unroll
    for (int g = 0; g < 4; ++g) add[j][g] = f32x4{0.f, 0.f, 0.f, 0.f};
  if (p.bias) {
    const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc((void*)p.bias, 0, 0x7fffffff, 0x00020000);
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int g = 0; g < 4; ++g)
        add[j][g] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsB, hh * 16, (nwu + 32 * j + 8 * g) * 4, 0));
  }
  if (p.rowvec) {
    const int b = mwu >> hw_shift;
    const __amdgpu_buffer_rsrc_t rsV =
        __builtin_amdgcn_make_buffer_rsrc((void*)(p.rowvec + (long long)b * p.ldrv), 0, 0x7fffffff, 0x00020000);
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f32x4 rv = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsV, hh * 16, (nwu + 32 * j + 8 * g) * 4, 0));
        // (wave_epilogue_fast: rv + bias, then + residual)
        add[j][g] = rv + add[j][g];
      }
  }
  // ---- residual: 8-byte loads of the lane's 4-column groups (row m, columns n .. n + 3), all in flight together
  typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
  u32x2 res[2][NJ][4];
  const bool has_res = p.residual != nullptr;
  if (has_res) {
    const __amdgpu_buffer_rsrc_t rsR = __builtin_amdgcn_make_buffer_rsrc((void*)p.residual, 0, 0x7fffffff, 0x00020000);
    const int vo = (ml * p.ldr + 4 * hh) * 2;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int g = 0; g < 4; ++g)
          res[i][j][g] = __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(
                                                       rsR, vo, ((mwu + 32 * i) * p.ldr + nwu + 32 * j + 8 * g) * 2, 0));
  }
  __builtin_amdgcn_s_waitcnt(0x0F70);          // vmcnt(0): the loads above (and the prefetched K tiles in front of them)
  const __amdgpu_buffer_rsrc_t rsO = __builtin_amdgcn_make_buffer_rsrc((void*)p.out, 0, 0x7fffffff, 0x00020000);
  const int st_vo = ((lane / CPR) * p.ldc + (lane % CPR) * 8) * 2;
  char* wr = patch + ml * EPI_ROWS_PITCH + hh * 8;
  const char* rd = patch + (lane / CPR) * EPI_ROWS_PITCH + (lane % CPR) * 16;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        float v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = acc[i][j][4 * g + k] * p.alpha + add[j][g][k];
        if (has_res) {
          const u32x2 rr = res[i][j][g];
          v[0] = acc[i][j][4 * g + 0] * p.alpha + (add[j][g][0] + __uint_as_float(rr.x << 16));
          v[1] = acc[i][j][4 * g + 1] * p.alpha + (add[j][g][1] + __uint_as_float(rr.x & 0xffff0000u));
          v[2] = acc[i][j][4 * g + 2] * p.alpha + (add[j][g][2] + __uint_as_float(rr.y << 16));
          v[3] = acc[i][j][4 * g + 3] * p.alpha + (add[j][g][3] + __uint_as_float(rr.y & 0xffff0000u));
        }
        if (p.act == SDMI_ACT_SILU) {
#pragma unroll
          for (int k = 0; k < 4; ++k) v[k] = act_apply<true>(v[k], SDMI_ACT_SILU);
        } else if (p.act == SDMI_ACT_RELU) {
#pragma unroll
          for (int k = 0; k < 4; ++k) v[k] = fmaxf(v[k], 0.f);
        } else if (p.act == SDMI_ACT_GELU) {
#pragma unroll
          for (int k = 0; k < 4; ++k) v[k] = act_apply<true>(v[k], SDMI_ACT_GELU);
        }
        u32x2 pk;
        pk.x = f32x2_to_bf16x2(v[0], v[1]);
        pk.y = f32x2_to_bf16x2(v[2], v[3]);
        *reinterpret_cast<u32x2*>(wr + j * 64 + g * 16) = pk;
      }
    // the wave's own writes complete in order before its reads; reads complete before the next round's writes
    u32x4 rows[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) rows[q] = *reinterpret_cast<const u32x4*>(rd + q * RPI * EPI_ROWS_PITCH);
#pragma unroll
    for (int q = 0; q < NQ; ++q)
      __builtin_amdgcn_raw_buffer_store_b128(rows[q], rsO, st_vo, ((mwu + 32 * i + RPI * q) * p.ldc + nwu) * 2, 0);
  }
}

// Fallback for the C^T layout (edge blocks, fp32 out, split-K partials, sub-sampled placement ...): plain per-element
// code, the arithmetic of wave_epilogue's generic path (inlined: a call would move the kernel arguments and the accumulators to
// scratch memory and make every value derived from them divergent).  Rare by construction (the callers' tiles are interior at the
// shapes that take these kernels).
template <int NJ>
__device__ __forceinline__ void wave_epilogue_rows_generic(const SdmiGemmArgs& p, f32x16 (&acc)[2][NJ], int mw0, int nw0,
                                                        int hw_shift, int lane) {
  const int ml = lane & 31, hh = lane >> 5;
  const int HoWo = p.Ho * p.Wo;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int m = mw0 + 32 * i + ml;
    if (m >= p.M) continue;
    const int b = hw_shift >= 0 ? (m >> hw_shift) : (m / HoWo);
    long long orow = (long long)m;
    if (p.osy > 0) {
      const int rem = m - b * HoWo, oy = rem / p.Wo, ox = rem - oy * p.Wo;
      orow = ((long long)b * p.oH + oy * p.osy + p.ooy) * p.oW + ox * p.osx + p.oox;
    }
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int n = nw0 + 32 * j + 8 * (r >> 2) + 4 * hh + (r & 3);
        if (n >= p.N) continue;
        float a = 0.f;
        if (p.bias) a = p.bias_m ? p.bias[m] : p.bias[n];
        if (p.rowvec) a += p.rowvec[(long long)b * p.ldrv + n];
        const long long o = orow * p.ldc + n;
        if (p.residual) {
          const long long ro = p.osy > 0 ? o : (long long)m * p.ldr + n;
          a += p.out_dtype == SDMI_BF16 ? bf16_to_f32(((const bf16_t*)p.residual)[ro]) : ((const float*)p.residual)[ro];
        }
        float v = acc[i][j][r] * p.alpha + a;
        v = p.out_dtype == SDMI_BF16 ? act_apply<true>(v, p.act) : act_apply(v, p.act);
        if (p.out_dtype == SDMI_BF16) ((bf16_t*)p.out)[o] = f32_to_bf16(v);
        else ((float*)p.out)[o] = v;
      }
  }
}

}  // namespace
