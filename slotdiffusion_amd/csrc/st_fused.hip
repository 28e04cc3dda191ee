// Image-resident fused SpatialTransformer block (bf16 inference; sdmi.h: sdmi_st_block).
//
// Reference: video_based/models/unet/attention.py:297-308 (SpatialTransformer: GroupNorm -> proj_in ->
// BasicTransformerBlock -> proj_out + x) and 247-251 (BasicTransformerBlock: self-attention, slot
// cross-attention, GEGLU feed-forward, each behind a LayerNorm and with a residual add).
//
// Everything in the block is local to a token row or to an image, so a workgroup that owns 64 token rows of
// one image needs no other workgroup -- except for the keys / values of its image's self-attention.  The
// block is therefore TWO launches instead of 10-13:
//   phase A  GroupNorm (statistics of the whole image recomputed by each of its S/64 workgroups: the image
//            is 32-128 KB and L2 resident) -> proj_in -> tok;  LayerNorm-fold -> q | k | v
//   phase B  self-attention of the 64 rows over the image's S keys -> to_out + tok -> LayerNorm-fold ->
//            folded slot cross-attention (per-image weights, softmax over each head's 7 slots) -> + ->
//            LayerNorm-fold -> GEGLU feed-forward, hidden chunk by hidden chunk, accumulated straight into
//            the merged (ff.net.2 ; proj_out) output -> + x
// Activations of the 64 rows stay in LDS (bf16, as the next GEMM's operand) and registers (fp32 residual
// stream); only tok, q|k|v and the block output touch HBM.
//
// GEMM core: out^T[n][m] = W[n][:] . act[m][:] on v_mfma_f32_16x16x32_bf16 with W as the A operand, so a lane's
// four accumulator values are FOUR CONSECUTIVE OUTPUT COLUMNS of ONE token row (lane & 15): residual adds,
// LayerNorm-fold terms, GEGLU and the 8-wide slot softmax are lane-local, and the result goes back to LDS as
// one ds_write_b64 in the next GEMM's operand layout.  The eight waves of the workgroup are the same program
// (no loader waves): wave w owns output columns [w*C/8, (w+1)*C/8) -- NSL = C/128 slices of 16 -- for ALL 64
// rows, and streams exactly the weight rows it multiplies through a PRIVATE ring of 2 KB units
// (`buffer_load_dwordx4 ... lds`, counted `s_waitcnt vmcnt`): no barrier orders weight traffic -- the waves run
// free inside a GEMM phase --, barriers exist only where an LDS activation buffer changes hands (~2 per phase).  The weights are pre-packed
// on the host (kern.WeightBank.st_stream_*) into per-wave streams of units in consumption order, each unit
// the exact XOR-swizzled LDS image of 16 weight rows x 64 k, so the fetch side is `base + g * 2048`.
#include "st_core.h"

namespace {

// ---------------------------------------------------------------------------------------------------------
// phase A: GroupNorm -> proj_in -> tok ; LayerNorm-fold -> q | k | v
// ---------------------------------------------------------------------------------------------------------
template <int C, int TT>
__global__ __launch_bounds__(512, 2) void st_block_a_kernel(SdmiStBlockArgs p) {
  typedef StGeom<C, TT> G;
  constexpr int ROWS = G::ROWS;
  constexpr int NSL = G::NSL, KT = G::KT, D = G::D, PITCH = G::PITCH;
  extern __shared__ __attribute__((aligned(16))) char smem_[];
  lds_char* const smem = (lds_char*)smem_;
  lds_char* const Y = smem + G::Y_OFF;
  float* const red = (float*)(smem_ + G::Y_OFF + G::Y_BYTES);          // GroupNorm partials (the chunk buffer's region)
  const int tid = threadIdx.x, lane = tid & 63, l15 = lane & 15, lg = lane >> 4;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wgs_per_img = p.S / ROWS;
  // block b runs on XCD b % 8 (observed placement; speed only): every XCD gets a contiguous range of workgroup
  // ids, so the S / ROWS workgroups of an image share one L2 (they all read the image's x / K / V)
  // ff_split = 2: a PAIR of workgroups per ROWS token rows -- both run the GroupNorm and proj_in (tok is written twice with
  // the same bytes), the first takes the q pass, the second k and v: 2 C^2 / 3 C^2 of weights per workgroup instead of 4 C^2
  const int vid0 = st_xcd_id((int)blockIdx.x, (int)gridDim.x);
  const int ffs = p.ff_split == 2 ? 2 : 1;
  const int half = ffs == 2 ? (vid0 & 1) : 0;
  const int vid = ffs == 2 ? (vid0 >> 1) : vid0;
  const int b = vid / wgs_per_img, rb = vid - b * wgs_per_img;
  const long long row0 = (long long)b * p.S + rb * ROWS;     // first token row of this workgroup
  const int pass0 = half ? 1 : 0, pass1 = (ffs == 2 && !half) ? 1 : 3;

  StRing<D> rg;
  rg.rs_sh = __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)p.wstream_a + (long long)w * G::UA * ST_UNIT), 0,
                                               G::UA * ST_UNIT, 0x00020000);
  rg.rs_img = rg.rs_sh;
  rg.g_iss = 0; rg.n1 = G::UA; rg.n_img = 0; rg.total = G::UA; rg.pos_iss = 0; rg.pos_con = 0;
  if (ffs == 2) {
    rg.total = half ? 3 * KT * NSL : 2 * KT * NSL;
    if (half) { rg.jump_at = KT * NSL; rg.jump = KT * NSL; }      // (skips the q units)
  }
  rg.ring = smem + G::RING_OFF + w * D * ST_UNIT;
  rg.voff = lane * 16;
#pragma unroll
  for (int i = 0; i < D; ++i) rg.issue_one();               // weights in flight under the GroupNorm
  STA_STAMP(0);

  // ---- GroupNorm statistics of the image (32 groups): thread -> (row slot, 16-byte vector column)
  constexpr int VPR = C / 8;                    // vectors per row
  constexpr int RPP = 512 / VPR;                // rows per pass
  constexpr int GS4 = C / 32 / 4;               // half-vectors (4 channels) per group
  {
    const bf16_t* xi = (const bf16_t*)p.x + (long long)b * p.S * C;
    float s0 = 0.f, s1 = 0.f, q0 = 0.f, q1 = 0.f;
    const int vr = tid / VPR, vc = tid - vr * VPR;
    if (vr < RPP) {
      // (eight rows in flight: one load per iteration behind its own wait cost a memory latency per row)
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
      *reinterpret_cast<f32x4*>(red + (vr * VPR + vc) * 4) = v;      // [row slot][vector][{s, q} x 2 halves]
    }
    ST_BARRIER();
    if (tid < 32) {
      float s = 0.f, q = 0.f;
      for (int r = 0; r < RPP; ++r)
        for (int h = 0; h < GS4; ++h) {
          const int hv = tid * GS4 + h;           // half-vector index along C
          const float* e = red + ((r * VPR + (hv >> 1)) * 4 + (hv & 1) * 2);
          s += e[0];
          q += e[1];
        }
      const float n = (float)(p.S * (C / 32));
      const float mean = s / n;
      const float var = fmaxf(q / n - mean * mean, 0.f);
      red[512 * 4 + tid * 2] = mean;
      red[512 * 4 + tid * 2 + 1] = rsqrtf(var + p.gn_eps);
    }
    ST_BARRIER();
  STA_STAMP(1);
  }
  // ---- normalise this workgroup's 64 rows into the operand buffer
  {
    const float* st = red + 512 * 4;
    const bf16_t* xr = (const bf16_t*)p.x + row0 * C;
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
      { const uint4 pk = pack16<bf16_t>(f);
        *reinterpret_cast<__attribute__((address_space(3))) u32x4*>(Y + r * PITCH + phys * 16) = u32x4{pk.x, pk.y, pk.z, pk.w}; }
    }
  }
  ST_BARRIER();
  STA_STAMP(2);

  // lane constants of the GEMM core
  int yaddr[4], woff[2];
#pragma unroll
  for (int j = 0; j < 4; ++j) yaddr[j] = l15 * PITCH + ((((4 * j + lg) ^ l15) & 15) * 16);
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) woff[ks] = l15 * 128 + (((4 * ks + lg) ^ ((l15 >> 1) & 7)) * 16);
  float sx[TT], sxx[TT];
#pragma unroll
  for (int tt = 0; tt < TT; ++tt) sx[tt] = sxx[tt] = 0.f;

  // Epilogue operands are fetched BEFORE the GEMM they follow: a vector load issued behind the weight DMAs
  // would make its consumer wait for every DMA in front of it (vmcnt retires in order) -- a ring drain.
  // ---- proj_in: tok = gn(x) Win^T + bin
  f32x4 acc[NSL][TT], ev0[NSL], ev1[NSL];
#pragma unroll
  for (int s = 0; s < NSL; ++s) {
    ev0[s] = st_vec4(p.vec_a + (w * NSL + s) * 16, lg);
#pragma unroll
    for (int tt = 0; tt < TT; ++tt) acc[s][tt] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
#pragma unroll
  for (int kt = 0; kt < KT; ++kt) st_gemm_step<D, NSL, false, 0, TT>(rg, Y, yaddr, kt, 16 * PITCH, woff, acc, sx, sxx);
  ST_BARRIER();                                      // every wave is done reading gn(x)
  STA_STAMP(3);
  {
    bf16_t* tok = (bf16_t*)p.tok + row0 * C;
#pragma unroll
    for (int s = 0; s < NSL; ++s) {
      const int n0 = (w * NSL + s) * 16 + 4 * lg;
      const f32x4 bi = ev0[s];
      const int c = n0 >> 3;
#pragma unroll
      for (int tt = 0; tt < TT; ++tt) {
        const int r = tt * 16 + l15;
        uint2 o;
        o.x = st_pack2(acc[s][tt][0] + bi[0], acc[s][tt][1] + bi[1]);
        o.y = st_pack2(acc[s][tt][2] + bi[2], acc[s][tt][3] + bi[3]);
        *reinterpret_cast<uint2*>(tok + (long long)r * C + n0) = o;
        const int phys = (c & ~15) | ((c ^ r) & 15);
        *reinterpret_cast<__attribute__((address_space(3))) u32x2*>(Y + r * PITCH + phys * 16 + (lg & 1) * 8) = u32x2{o.x, o.y};
      }
    }
  }
  ST_BARRIER();
  STA_STAMP(4);

  // ---- q | k | v = LayerNorm(tok) W^T through the fold: three passes of N = C over the same operand
  float mean[TT], rstd[TT];
#pragma unroll 1
  for (int pass = pass0; pass < pass1; ++pass) {
    const float* colsum = p.vec_a + C + pass * C;
    const float* bias = p.vec_a + 4 * C + pass * C;
#pragma unroll
    for (int s = 0; s < NSL; ++s) {
      ev0[s] = st_vec4(colsum + (w * NSL + s) * 16, lg);
      ev1[s] = st_vec4(bias + (w * NSL + s) * 16, lg);
#pragma unroll
      for (int tt = 0; tt < TT; ++tt) acc[s][tt] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    // (the NSL * 4 row stores of the epilogue before this pass are younger than every DMA in flight)
    if (pass == pass0) {
      st_gemm_step<D, NSL, true, NSL * TT, TT>(rg, Y, yaddr, 0, 16 * PITCH, woff, acc, sx, sxx);
#pragma unroll
      for (int kt = 1; kt < KT; ++kt) st_gemm_step<D, NSL, true, 0, TT>(rg, Y, yaddr, kt, 16 * PITCH, woff, acc, sx, sxx);
      st_ln_stats<TT>(sx, sxx, 1.f / (float)C, p.ln_eps, mean, rstd);
    } else {
      st_gemm_step<D, NSL, false, NSL * TT, TT>(rg, Y, yaddr, 0, 16 * PITCH, woff, acc, sx, sxx);
#pragma unroll
      for (int kt = 1; kt < KT; ++kt) st_gemm_step<D, NSL, false, 0, TT>(rg, Y, yaddr, kt, 16 * PITCH, woff, acc, sx, sxx);
    }
    bf16_t* qkv = (bf16_t*)p.qkv + row0 * 3 * C + pass * C;
#pragma unroll
    for (int s = 0; s < NSL; ++s) {
      const int n0 = (w * NSL + s) * 16 + 4 * lg;
      const f32x4 cs = ev0[s];
      const f32x4 bi = ev1[s];
#pragma unroll
      for (int tt = 0; tt < TT; ++tt) {
        const int r = tt * 16 + l15;
        float v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = rstd[tt] * (acc[s][tt][j] - mean[tt] * cs[j]) + bi[j];
        uint2 o;
        o.x = st_pack2(v[0], v[1]);
        o.y = st_pack2(v[2], v[3]);
        *reinterpret_cast<uint2*>(qkv + (long long)r * 3 * C + n0) = o;
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // no DMA may outlive the workgroup's LDS
  STA_STAMP(5);
}

// ---------------------------------------------------------------------------------------------------------
// phase B
// ---------------------------------------------------------------------------------------------------------
template <int C, int TT>
__global__ __launch_bounds__(512, 2) void st_block_b_kernel(SdmiStBlockArgs p) {
  typedef StGeom<C, TT> G;
  constexpr int ROWS = G::ROWS;
  constexpr int NSL = G::NSL, KT = G::KT, D = G::D, PITCH = G::PITCH, HEADS = G::HEADS, R = G::R;
  extern __shared__ __attribute__((aligned(16))) char smem_[];
  lds_char* const smem = (lds_char*)smem_;
  lds_char* const Y = smem + G::Y_OFF;
  lds_char* const Gb = smem + G::Y_OFF + G::Y_BYTES;
  const int tid = threadIdx.x, lane = tid & 63, l15 = lane & 15, lg = lane >> 4;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wgs_per_img = p.S / ROWS;
  // block b runs on XCD b % 8 (observed placement; speed only): every XCD gets a contiguous range of workgroup
  // ids, so the S / ROWS workgroups of an image share one L2 (they all read the image's x / K / V)
  const int vid0 = st_xcd_id((int)blockIdx.x, (int)gridDim.x);
  const int ffs = p.ff_split == 2 ? 2 : 1;
  const int half = ffs == 2 ? (vid0 & 1) : 0;            // (the two workgroups of a pair: neighbours on one XCD)
  const int vid = ffs == 2 ? (vid0 >> 1) : vid0;
  const int b = vid / wgs_per_img, rb = vid - b * wgs_per_img;
  const long long row0 = (long long)b * p.S + rb * ROWS;
  const int S = p.S;
  ST_STAMP(0);

  // =========================== self-attention: 64 queries x HEADS heads over S keys ===========================
  // (attention.hip's transposed matrix-core formulation: a wave owns 32 queries of one head; four heads per round)
  unsigned opack[HEADS / 4][8];
  {
    // 64 rows: two 32-query halves x four heads per round; 32 rows: waves 0-3 take the four heads, waves 4-7 only
    // stage and keep the barriers
    const int hs = TT == 4 ? (w >> 1) : (w & 3), qh = TT == 4 ? (w & 1) : 0;
    const bool att_active = TT == 4 || w < 4;
    const int ql = lane & 31, hh = lane >> 5;
    const bf16_t* qkv_img = (const bf16_t*)p.qkv + (long long)b * S * 3 * C;
    const int head_bytes = S * (ST_KP + ST_VP);
    const float sc2 = p.attn_scale * 1.4426950408889634f;
    const int g4 = lane >> 4, t16 = lane & 15;
    // K / V of `hb` heads are staged at a time: all of them when they fit (S = 64: one staging pass for the block),
    // four otherwise; the loads of a pass are in flight together (a load per iteration behind its own wait cost a
    // memory latency each: 27 of the kernel's 72 us at S = 256).
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
      const int ppr = hb * 8;                              // 16-byte pieces per key row: [K of hb heads | V of hb heads]
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
        const int per_thread = S * hb / 64;                // pieces per thread of this pass
        if (per_thread % 16 == 0) stage(ri * 4, std::integral_constant<int, 16>());
        else if (per_thread % 12 == 0) stage(ri * 4, std::integral_constant<int, 12>());
        else stage(ri * 4, std::integral_constant<int, 4>());
        __syncthreads();
        if (ri == 0) ST_STAMP(8);
      }
      const int hl = (all_heads ? ri * 4 : 0) + hs;
      const lds_char* Ks = smem + hl * head_bytes;
      const lds_char* Vs = Ks + S * ST_KP;
      const lds_char* kfrag = Ks + ql * ST_KP + hh * 16;
      const lds_char* vfrag = Vs + (4 * hh + (t16 >> 2)) * ST_VP + ((g4 & 1) * 16 + (t16 & 3) * 4) * 2;
      const int rd = ri;
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
        opack[rd][2 * j] = st_pack2(o[4 * j] * inv, o[4 * j + 1] * inv);
        opack[rd][2 * j + 1] = st_pack2(o[4 * j + 2] * inv, o[4 * j + 3] * inv);
      }
    }
    ST_STAMP(9);
    __syncthreads();                      // the staging region becomes operand buffers + rings
    ST_STAMP(10);
  }

  // the token residual of the first epilogue: fetched and RETIRED before any weight DMA is issued (st_vec4's note)
  uint2 rsd[NSL][TT];
  {
    const bf16_t* tok = (const bf16_t*)p.tok + row0 * C;
#pragma unroll
    for (int s = 0; s < NSL; ++s)
#pragma unroll
      for (int tt = 0; tt < TT; ++tt)
        rsd[s][tt] = *reinterpret_cast<const uint2*>(tok + (long long)(tt * 16 + l15) * C + (w * NSL + s) * 16 + 4 * lg);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  ST_STAMP(11);
  // the second-dispatched half of the workgroup loses issue arbitration to the first on every SIMD and arrives last
  // at each hand-off barrier (the FF loop's wave 0 spent 22 % of its time waiting there): static priority for it
  if (w >= 4) __builtin_amdgcn_s_setprio(1);
  // ================================ weight stream + ring of this wave ================================
  StRing<D> rg;
  rg.rs_sh = __builtin_amdgcn_make_buffer_rsrc(
      (void*)((const char*)p.wstream_b + (long long)w * (G::UB1 + G::UB2) * ST_UNIT), 0, (G::UB1 + G::UB2) * ST_UNIT, 0x00020000);
  rg.rs_img = __builtin_amdgcn_make_buffer_rsrc(
      (void*)((const char*)p.wstream_img + ((long long)b * 8 + w) * G::UIMG * ST_UNIT), 0, G::UIMG * ST_UNIT, 0x00020000);
  rg.g_iss = 0; rg.n1 = G::UB1; rg.n_img = G::UIMG; rg.total = G::UB1 + G::UIMG + G::UB2;
  constexpr int UCH = KT * 2 + 2 * NSL;                 // units of one hidden chunk
  if (ffs == 2) {                                        // this workgroup's half of the hidden chunks
    rg.total -= (G::NHC / 2) * UCH;
    if (half) { rg.jump_at = G::UB1 + KT * NSL; rg.jump = (G::NHC / 2) * UCH; }
  }
  rg.pos_iss = 0; rg.pos_con = 0;
  rg.ring = smem + G::RING_OFF + w * D * ST_UNIT;
  rg.voff = lane * 16;
#pragma unroll
  for (int i = 0; i < D; ++i) rg.issue_one();
  {
    const int hs = TT == 4 ? (w >> 1) : (w & 3), qh = TT == 4 ? (w & 1) : 0;
    const int ql = lane & 31, hh = lane >> 5;
    // attention output -> operand buffer: row qh * 32 + ql, channels h * 32 + 8 j + 4 hh .. + 4
    const int r = qh * 32 + ql;
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
      }
    }
  }
  ST_BARRIER();                           // attention output complete in Y
  ST_STAMP(1);

  int yaddr[4], gaddr[4], woff[2];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int sw = (((4 * j + lg) ^ l15) & 15) * 16;
    yaddr[j] = l15 * PITCH + sw;
    gaddr[j] = l15 * 256 + sw;
  }
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) woff[ks] = l15 * 128 + (((4 * ks + lg) ^ ((l15 >> 1) & 7)) * 16);
  float sx[TT], sxx[TT], mean[TT], rstd[TT];
  const float* vb = p.vec_b;              // [bo | bo2 | colsum_ff (8C) | bias_ff (8C) | bias_out]  fp32
  const float* vi = p.vec_img + (long long)b * 256;   // per image: [colsum_q (128) | bias_q (128)]

  // residual stream of this wave's columns, fp32: res[s][tt][j] = row 16 tt + l15, column (w NSL + s) 16 + 4 lg + j
  f32x4 res[NSL][TT], acc[NSL][TT];
  auto zero_acc = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int s = 0; s < NSL; ++s)
#pragma unroll
      for (int tt = 0; tt < TT; ++tt) acc[s][tt] = f32x4{0.f, 0.f, 0.f, 0.f};
  };
  auto res_to_y = [&]() __attribute__((always_inline)) {      // bf16 copy of the residual stream = next GEMM's operand
#pragma unroll
    for (int s = 0; s < NSL; ++s) {
      const int c = ((w * NSL + s) * 16 + 4 * lg) >> 3;
#pragma unroll
      for (int tt = 0; tt < TT; ++tt) {
        const int r = tt * 16 + l15;
        const int phys = (c & ~15) | ((c ^ r) & 15);
        uint2 o;
        o.x = st_pack2(res[s][tt][0], res[s][tt][1]);
        o.y = st_pack2(res[s][tt][2], res[s][tt][3]);
        *reinterpret_cast<__attribute__((address_space(3))) u32x2*>(Y + r * PITCH + phys * 16 + (lg & 1) * 8) = u32x2{o.x, o.y};
      }
    }
  };

  // (epilogue operands are fetched BEFORE the GEMM they follow -- see phase A)
  ST_STAMP(2);
  // ---- attn1.to_out + tok  -> x1
  f32x4 ev0[NSL];
#pragma unroll
  for (int s = 0; s < NSL; ++s) ev0[s] = st_vec4(vb + (w * NSL + s) * 16, lg);
  zero_acc();
#pragma unroll
  for (int kt = 0; kt < KT; ++kt) st_gemm_step<D, NSL, false, 0, TT>(rg, Y, yaddr, kt, 16 * PITCH, woff, acc, sx, sxx);
  {
#pragma unroll
    for (int s = 0; s < NSL; ++s) {
      const f32x4 bi = ev0[s];
#pragma unroll
      for (int tt = 0; tt < TT; ++tt) {
        const uint2 t2 = rsd[s][tt];
        res[s][tt][0] = acc[s][tt][0] + bi[0] + __uint_as_float(t2.x << 16);
        res[s][tt][1] = acc[s][tt][1] + bi[1] + __uint_as_float(t2.x & 0xffff0000u);
        res[s][tt][2] = acc[s][tt][2] + bi[2] + __uint_as_float(t2.y << 16);
        res[s][tt][3] = acc[s][tt][3] + bi[3] + __uint_as_float(t2.y & 0xffff0000u);
      }
    }
  }
  ST_BARRIER();
  res_to_y();
  ST_BARRIER();

  ST_STAMP(3);
  // ---- folded slot cross-attention: P = softmax8(LN-fold(x1) Wq[b]^T) ; x2 = P W2[b]^T + bo2 + x1
  {
    f32x4 sc[1][TT];
    const int n0 = w * 16 + 4 * lg;                   // column of the padded score matrix
    const f32x4 cs = st_vec4(vi + w * 16, lg);
    const f32x4 bi = st_vec4(vi + 128 + w * 16, lg);
#pragma unroll
    for (int s = 0; s < NSL; ++s) ev0[s] = st_vec4(vb + C + (w * NSL + s) * 16, lg);
#pragma unroll
    for (int tt = 0; tt < TT; ++tt) {
      sc[0][tt] = f32x4{0.f, 0.f, 0.f, 0.f};
      sx[tt] = sxx[tt] = 0.f;
    }
#pragma unroll
    for (int kt = 0; kt < KT; ++kt) st_gemm_step<D, 1, true, 0, TT>(rg, Y, yaddr, kt, 16 * PITCH, woff, sc, sx, sxx);
    st_ln_stats<TT>(sx, sxx, 1.f / (float)C, p.ln_eps, mean, rstd);
    const int c = n0 >> 3;
    // score groups: 8 columns per head (two lane groups of four, lg ^ 1) up to 8 slots; from 9 slots (C = 256: 8 heads
    // x 16 = the 128 padded columns, a wave's 16-column slice is ONE head) 16 columns = all four lane groups of a row
    const bool wide = p.slots > 8;
    const int Rr = wide ? HEADS * 16 : R;
#pragma unroll
    for (int tt = 0; tt < TT; ++tt) {
      float v[4];
      float mx = -INFINITY;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const bool ok = n0 < Rr && (4 * (wide ? lg : (lg & 1)) + j) < p.slots;
        v[j] = ok ? rstd[tt] * (sc[0][tt][j] - mean[tt] * cs[j]) + bi[j] : -INFINITY;
        mx = fmaxf(mx, v[j]);
      }
      mx = fmaxf(mx, __shfl_xor(mx, 16, 64));          // the group's other four columns: lane ^ 16
      if (wide) mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
      float sm = 0.f;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        v[j] = n0 < Rr ? __expf(v[j] - mx) : 0.f;
        sm += v[j];
      }
      sm += __shfl_xor(sm, 16, 64);
      if (wide) sm += __shfl_xor(sm, 32, 64);
      const float inv = n0 < Rr ? 1.f / sm : 0.f;
      const int r = tt * 16 + l15;
      const int phys = (c & ~15) | ((c ^ r) & 15);
      uint2 o;
      o.x = st_pack2(v[0] * inv, v[1] * inv);
      o.y = st_pack2(v[2] * inv, v[3] * inv);
      *reinterpret_cast<__attribute__((address_space(3))) u32x2*>(Gb + r * 256 + phys * 16 + (lg & 1) * 8) = u32x2{o.x, o.y};
    }
  }
  ST_BARRIER();
  zero_acc();
#pragma unroll
  for (int kt = 0; kt < 2; ++kt) st_gemm_step<D, NSL, false, 0, TT>(rg, Gb, gaddr, kt, 16 * 256, woff, acc, sx, sxx);
#pragma unroll
  for (int s = 0; s < NSL; ++s) {
    const f32x4 bi = ev0[s];
#pragma unroll
    for (int tt = 0; tt < TT; ++tt)
#pragma unroll
      for (int j = 0; j < 4; ++j) res[s][tt][j] += acc[s][tt][j] + bi[j];
  }
  ST_BARRIER();                                         // (Y readers of the score GEMM are long done; Gb readers too)
  res_to_y();
  ST_BARRIER();

  ST_STAMP(4);
  // ---- merged (ff.net.2 ; proj_out): out = [g | x2] [Wpo Wff | Wpo]^T + b' + x.  First the x2 part (also
  //      yields the LayerNorm-fold statistics of x2 for the feed-forward), then the hidden chunks.
#pragma unroll
  for (int tt = 0; tt < TT; ++tt) sx[tt] = sxx[tt] = 0.f;
  zero_acc();
#pragma unroll
  for (int kt = 0; kt < KT; ++kt) st_gemm_step<D, NSL, true, 0, TT>(rg, Y, yaddr, kt, 16 * PITCH, woff, acc, sx, sxx);
  st_ln_stats<TT>(sx, sxx, 1.f / (float)C, p.ln_eps, mean, rstd);
  if (half) zero_acc();                                  // (the x2 Wpo^T term belongs to the first workgroup of the pair)
  const float* cs_ff = vb + 2 * C;
  const float* bi_ff = vb + 10 * C;
  ST_TL_DECL
  ST_TL_BEGIN
  const int hc0 = half * (G::NHC / 2), hc1 = ffs == 2 ? hc0 + G::NHC / 2 : G::NHC;
#pragma unroll 1
  for (int hc = hc0; hc < hc1; ++hc) {
    f32x4 vg[2][TT];                                     // [value | gate] columns hc * 128 + w * 16 + 4 lg ..
    const f32x4 csv = st_vec4(cs_ff + hc * 128 + w * 16, lg);
    const f32x4 csg = st_vec4(cs_ff + 4 * C + hc * 128 + w * 16, lg);
    const f32x4 biv = st_vec4(bi_ff + hc * 128 + w * 16, lg);
    const f32x4 big = st_vec4(bi_ff + 4 * C + hc * 128 + w * 16, lg);
#pragma unroll
    for (int tt = 0; tt < TT; ++tt) vg[0][tt] = vg[1][tt] = f32x4{0.f, 0.f, 0.f, 0.f};
    ST_TL_LAP(3);
#pragma unroll
    for (int kt = 0; kt < KT; ++kt) st_gemm_step<D, 2, false, 0, TT>(rg, Y, yaddr, kt, 16 * PITCH, woff, vg, sx, sxx);
    ST_TL_LAP(0);
    lds_char* gb = Gb + (G::GBUF == 2 ? (hc & 1) * G::G_BYTES : 0);
    if (G::GBUF == 1) ST_BARRIER();                     // the previous chunk's readers are done
    {
      const int c = (w * 16 + 4 * lg) >> 3;
#pragma unroll
      for (int tt = 0; tt < TT; ++tt) {
        float y[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float v = rstd[tt] * (vg[0][tt][j] - mean[tt] * csv[j]) + biv[j];
          const float g = rstd[tt] * (vg[1][tt][j] - mean[tt] * csg[j]) + big[j];
          y[j] = v * act_apply<true>(g, SDMI_ACT_GELU);
        }
        const int r = tt * 16 + l15;
        const int phys = (c & ~15) | ((c ^ r) & 15);
        uint2 o;
        o.x = st_pack2(y[0], y[1]);
        o.y = st_pack2(y[2], y[3]);
        *reinterpret_cast<__attribute__((address_space(3))) u32x2*>(gb + r * 256 + phys * 16 + (lg & 1) * 8) = u32x2{o.x, o.y};
      }
    }
    ST_TL_LAP(1);
    ST_BARRIER();
    ST_TL_LAP(2);
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) st_gemm_step<D, NSL, false, 0, TT>(rg, gb, gaddr, kt, 16 * 256, woff, acc, sx, sxx);
  }
  ST_STAMP(5);
  ST_TL_FLUSH;
  // ---- + b' + x  -> out   (only dummy re-fetches are in flight now: drain them, then ordinary loads are safe)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (ffs == 2) {          // this workgroup's fp32 partial; the pair is summed (+ b' + x) by the kernel behind
    float* pp = p.part + ((long long)half * p.B * p.S + row0) * C;
#pragma unroll
    for (int s = 0; s < NSL; ++s)
#pragma unroll
      for (int tt = 0; tt < TT; ++tt)
        *reinterpret_cast<f32x4*>(pp + (long long)(tt * 16 + l15) * C + (w * NSL + s) * 16 + 4 * lg) = acc[s][tt];
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    return;
  }
  {
    const bf16_t* xr = (const bf16_t*)p.x + row0 * C;
#pragma unroll
    for (int s = 0; s < NSL; ++s) {
      ev0[s] = st_vec4(vb + 18 * C + (w * NSL + s) * 16, lg);
#pragma unroll
      for (int tt = 0; tt < TT; ++tt)
        rsd[s][tt] = *reinterpret_cast<const uint2*>(xr + (long long)(tt * 16 + l15) * C + (w * NSL + s) * 16 + 4 * lg);
    }
  }
  {
    bf16_t* outp = (bf16_t*)p.out + row0 * C;
#pragma unroll
    for (int s = 0; s < NSL; ++s) {
      const int n0 = (w * NSL + s) * 16 + 4 * lg;
      const f32x4 bi = ev0[s];
#pragma unroll
      for (int tt = 0; tt < TT; ++tt) {
        const long long o_ = (long long)(tt * 16 + l15) * C + n0;
        const uint2 x2 = rsd[s][tt];
        uint2 o;
        o.x = st_pack2(acc[s][tt][0] + bi[0] + __uint_as_float(x2.x << 16),
                       acc[s][tt][1] + bi[1] + __uint_as_float(x2.x & 0xffff0000u));
        o.y = st_pack2(acc[s][tt][2] + bi[2] + __uint_as_float(x2.y << 16),
                       acc[s][tt][3] + bi[3] + __uint_as_float(x2.y & 0xffff0000u));
        *reinterpret_cast<uint2*>(outp + o_) = o;
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  ST_STAMP(6);
}

template <int C, int TT>
int st_launch(const SdmiStBlockArgs& a, hipStream_t st) {
  typedef StGeom<C, TT> G;
  const int heads_bytes = a.S * (ST_KP + ST_VP);
  const int att = (G::HEADS * heads_bytes <= 160 * 1024 ? G::HEADS : 4) * heads_bytes;
  const int smem_b = att > G::SMEM_GEMM ? att : G::SMEM_GEMM;
  const int grid = a.B * (a.S / G::ROWS);
  if (a.phase == 0 || a.phase == 1) {
    SDMI_OPTIN_LDS((st_block_a_kernel<C, TT>), G::SMEM_GEMM, "st_block (phase A)");
    hipLaunchKernelGGL((st_block_a_kernel<C, TT>), dim3(grid * (a.ff_split == 2 ? 2 : 1)), dim3(512), G::SMEM_GEMM, st, a);
    const int rc = sdmi_check_launch("st_block (phase A)");
    if (rc) return rc;
  }
  if (a.phase == 0 || a.phase == 2) {
    SDMI_OPTIN_LDS((st_block_b_kernel<C, TT>), 160 * 1024, "st_block (phase B)");
    hipLaunchKernelGGL((st_block_b_kernel<C, TT>), dim3(grid * (a.ff_split == 2 ? 2 : 1)), dim3(512), smem_b, st, a);
    return sdmi_check_launch("st_block (phase B)");
  }
  return SDMI_OK;
}

}  // namespace

extern "C" int sdmi_st_block(const SdmiStBlockArgs* a, void* stream) {
  SDMI_REQUIRE(a && a->x && a->tok && a->qkv && a->out, "null pointer");
  SDMI_REQUIRE(a->wstream_a && a->vec_a && a->wstream_b && a->wstream_img && a->vec_b && a->vec_img, "null stream");
  SDMI_REQUIRE(a->C == 256 || a->C == 384, "C must be 256 or 384");
  SDMI_REQUIRE(a->rows == 0 || a->rows == 64 || a->rows == 32, "rows per workgroup: 64 (0) or 32");
  const int rows = a->rows ? a->rows : 64;
  SDMI_REQUIRE(a->S >= rows && a->S % rows == 0 && a->S % 32 == 0 && 4 * a->S * (ST_KP + ST_VP) <= 160 * 1024,
               "S must be a multiple of the rows per workgroup and at most 256 tokens per image");
  SDMI_REQUIRE(a->slots >= 1 && (a->slots <= 8 || (a->slots <= 16 && a->C == 256)),
               "1..8 slots (C = 256: up to 16, in 16-column score groups)");
  SDMI_REQUIRE(a->phase >= 0 && a->phase <= 2, "phase: 0 = both, 1 = A, 2 = B");
  SDMI_REQUIRE(a->ff_split == 0 || a->ff_split == 1 || (a->ff_split == 2 && a->part), "ff_split: 0 / 1, or 2 with a partials buffer");
  hipStream_t st = (hipStream_t)stream;
  if (rows == 64) return a->C == 256 ? st_launch<256, 4>(*a, st) : st_launch<384, 4>(*a, st);
  return a->C == 256 ? st_launch<256, 2>(*a, st) : st_launch<384, 2>(*a, st);
}
