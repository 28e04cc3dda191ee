// VQ nearest-code search (include/sdmi.h: sdmi_vq_nearest; reference vqvae/quantize.py:85-94).
//
// Integer output => bit-exact contract.  The reference evaluates, in fp32 on the CPU,
//     d[r][j] = (|z_r|^2 + |e_j|^2) - 2 * (z_r . e_j),      argmin_j (first minimum)
// with |.|^2 = (x0*x0 + x1*x1) + x2*x2 (separately rounded products) and the dot product as the
// FMA chain fma(z2,e2, fma(z1,e1, z0*e0)) (probed against torch 2.10 CPU; tests/test_vq.py keeps
// the probe).  This file is compiled with -ffp-contract=off so exactly those roundings happen.
// The codebook lives in LDS as PAIRS of codes -- {e0a,e0b, e1a,e1b, e2a,e2b, |ea|^2,|eb|^2} = 32 bytes per pair,
// 64 KiB for 4096 codes -- so that two ds_read_b128 feed the packed-fp32 VALU (v_pk_mul / v_pk_fma / v_pk_add:
// two codes per instruction, each lane-half rounding exactly like the scalar instruction).  The scan is VALU-bound
// (65536 latents x 4096 codes per sampling step), so it only tracks the minimum DISTANCE per chunk of 64 codes
// (v_min3_f32: half an instruction per code) and remembers the first chunk that lowers it; the index -- the
// reference's "first minimum" -- is recovered by re-evaluating that one chunk.  (zz + |e|^2) - 2 * dot is formed
// as fma(-2, dot, zz + |e|^2): 2 * dot is exact, so the single rounding is the same one.
// Two threads share a latent, each scanning half of the chunks.  Measured at [65536] x [4096]: 90 us (scalar scan with
// per-code index bookkeeping) -> 79.5 us: the VALU work fell to a third, the LDS pipe became the bound.  This kernel now
// serves few latents (< 512) and large codebooks; the sampler's shape takes vq_reg_kernel below (46 us).
#include "common.h"

namespace {

typedef float vq_f2 __attribute__((ext_vector_type(2)));

// channel c of latent row zr (optionally the linear combination of sdmi.h: z2), scaled
__device__ __forceinline__ float vq_chan(const SdmiVqArgs& p, const float* zr, const float* z2r, int c) {
  float v = zr[c];
  if (z2r) {
    const float t0 = p.zc0 * v, t1 = p.zc1 * z2r[c];
    v = t0 + t1;
    if (p.zdiv != 0.f) v = v / p.zdiv;
  }
  return v * p.scale;
}
constexpr int VQ_CH = 32;            // pairs per chunk

__device__ __forceinline__ vq_f2 vq_dist2(const float* cbp, vq_f2 z0, vq_f2 z1, vq_f2 z2, vq_f2 zz) {
  const f32x4 a = *reinterpret_cast<const f32x4*>(cbp);
  const f32x4 b = *reinterpret_cast<const f32x4*>(cbp + 4);
  const vq_f2 e0 = {a[0], a[1]}, e1 = {a[2], a[3]}, e2 = {b[0], b[1]}, e3 = {b[2], b[3]};
  vq_f2 dot = z0 * e0;
  dot = __builtin_elementwise_fma(z1, e1, dot);
  dot = __builtin_elementwise_fma(z2, e2, dot);
  const vq_f2 t = zz + e3;
  const vq_f2 m2 = {-2.f, -2.f};
  return __builtin_elementwise_fma(m2, dot, t);
}

__device__ __forceinline__ float vq_dist1(float z0, float z1, float z2, float zz, float a, float b, float c, float n2) {
  const float dot = fmaf(z2, c, fmaf(z1, b, z0 * a));
  return fmaf(-2.f, dot, zz + n2);
}

// Non-finite latents (NaN / Inf components, or |z|^2 overflowing): the packed scan's fminf drops NaN distances, so no
// code "attains the minimum" and the index would stay at its sentinel.  These rows take a sequential scan with
// torch.argmin's semantics (vqvae/quantize.py:90): the first NaN distance if there is one, else the first minimum.
__device__ __noinline__ int vq_scan_nonfinite(const SdmiVqArgs& p, float z0, float z1, float z2, float zz) {
  float best = INFINITY;
  int bi = 0;
  for (int j = 0; j < p.n_codes; ++j) {
    const float a = p.codebook[j * 3 + 0], b = p.codebook[j * 3 + 1], c = p.codebook[j * 3 + 2];
    const float t0 = a * a, t1 = b * b, t2 = c * c;
    // (the reference's separately rounded 2 * dot: equal to the scan's fma(-2, dot, .) except where 2 * dot overflows,
    // which only these rows can reach -- inf - inf = NaN there, +-inf from the fused form)
    const float dot = fmaf(z2, c, fmaf(z1, b, z0 * a));
    const float two = 2.f * dot;
    const float d = (zz + ((t0 + t1) + t2)) - two;
    if (d != d) return j;
    if (d < best) { best = d; bi = j; }
  }
  return bi;
}

__global__ __launch_bounds__(256) void vq_kernel(SdmiVqArgs p) {
  extern __shared__ __attribute__((aligned(16))) float cb[];  // [pairs padded to chunks][8]
  const int n_pairs = (p.n_codes + 1) / 2;
  const int n_chunks = (n_pairs + VQ_CH - 1) / VQ_CH;
  for (int j = threadIdx.x; j < n_chunks * VQ_CH * 2; j += blockDim.x) {
    float e0 = 0.f, e1 = 0.f, e2 = 0.f, n2 = INFINITY;          // pad codes: distance +inf, never the minimum
    if (j < p.n_codes) {
      e0 = p.codebook[j * 3 + 0]; e1 = p.codebook[j * 3 + 1]; e2 = p.codebook[j * 3 + 2];
      const float s0 = e0 * e0, s1 = e1 * e1, s2 = e2 * e2;
      n2 = (s0 + s1) + s2;
    }
    float* o = cb + (j >> 1) * 8 + (j & 1);
    o[0] = e0; o[2] = e1; o[4] = e2; o[6] = n2;
  }
  __syncthreads();
  // partial results merge on (distance, index): the smallest distance, on ties the smallest index -- the
  // reference's "first minimum" of the sequential scan
  const int r = blockIdx.x * (blockDim.x / 2) + (threadIdx.x >> 1);
  const int half = threadIdx.x & 1;
  const bool live = r < p.R;
  const float* zr = p.z + (long long)(live ? r : 0) * p.ldz;
  const float* z2r = p.z2 ? p.z2 + (long long)(live ? r : 0) * p.ldz : nullptr;
  const float z0 = vq_chan(p, zr, z2r, 0), z1 = vq_chan(p, zr, z2r, 1), z2 = vq_chan(p, zr, z2r, 2);
  const float q0 = z0 * z0, q1 = z1 * z1, q2 = z2 * z2;
  const float zz = (q0 + q1) + q2;
  const vq_f2 z0v = {z0, z0}, z1v = {z1, z1}, z2v = {z2, z2}, zzv = {zz, zz};
  const int c_half = (n_chunks + 1) / 2;
  const int c0 = half * c_half, c1 = min(n_chunks, c0 + c_half);
  float best = INFINITY;
  int bc = c0;
  for (int c = c0; c < c1; ++c) {
    const float* base = cb + c * VQ_CH * 8;
    float m0 = INFINITY, m1 = INFINITY;
#pragma unroll
    for (int u = 0; u < VQ_CH; u += 2) {
      const vq_f2 da = vq_dist2(base + u * 8, z0v, z1v, z2v, zzv);
      const vq_f2 db = vq_dist2(base + u * 8 + 8, z0v, z1v, z2v, zzv);
      m0 = fminf(fminf(m0, da[0]), da[1]);
      m1 = fminf(fminf(m1, db[0]), db[1]);
    }
    const float m = fminf(m0, m1);
    if (m < best) { best = m; bc = c; }
  }
  // the first code of chunk bc that attains `best` (same instructions -> the same bits)
  int bi = 0x7fffffff;
  {
    const float* base = cb + bc * VQ_CH * 8;
    for (int u = VQ_CH - 1; u >= 0; --u) {
      const vq_f2 d = vq_dist2(base + u * 8, z0v, z1v, z2v, zzv);
      const int j = (bc * VQ_CH + u) * 2;
      if (d[1] == best) bi = j + 1;
      if (d[0] == best) bi = j;
    }
  }
  {
    const float ob = __shfl_xor(best, 1, 64);
    const int oi = __shfl_xor(bi, 1, 64);
    if (ob < best || (ob == best && oi < bi)) { best = ob; bi = oi; }
  }
  if (!live || half) return;
  if (!(zz < INFINITY)) bi = vq_scan_nonfinite(p, z0, z1, z2, zz);
  if (p.idx) p.idx[r] = bi;
  if (p.zq) {
    float* o = p.zq + (long long)r * p.ldz;
    const float* e = cb + (bi >> 1) * 8 + (bi & 1);
    const float c0_ = z0 + (e[0] - z0), c1_ = z1 + (e[2] - z1), c2_ = z2 + (e[4] - z2);
    o[0] = c0_ / p.scale; o[1] = c1_ / p.scale; o[2] = c2_ / p.scale;
    for (int c = 3; c < p.ldz; ++c) o[c] = 0.f;
  }
}

// ---- codes in registers ---------------------------------------------------------------------------------------
// The LDS scan above is bound by the LDS pipe (every lane reads the same 32 bytes: 2 ds_read_b128 per pair of codes
// per wave = 78 us at [65536] x [4096]).  Here the CODES are the lane-private operand: wave w of a workgroup keeps
// codes [w * 128 NP, (w + 1) * 128 NP) in registers -- lane l holds NP pairs, pair q = codes base + 128 q + 2 l + {0,1}
// -- and the workgroup's 64 latents (one per lane, broadcast with v_readlane) stream past as wave-uniform scalars.
// Per latent and wave: NP x 5 packed operations, the lane minimum (v_min3), the wave minimum (six DPP steps), and
// the ballot of the lanes that attain it; nothing else -- no index bookkeeping in the hot loop.  Afterwards lane t
// owns latent t: it takes the wave with the smallest minimum (lower wave = lower indices on ties) and re-evaluates
// the 2 NP codes of each lane in that wave's ballot (one lane unless distances tie exactly) for the FIRST index that
// attains the minimum.  Same instruction sequence per distance everywhere -> the same bits.
constexpr int VQ_LAT = 64;           // latents per workgroup (one per lane)

// min over the wave, valid in lane 63 (row_ror: every lane of a 16-lane row gets the row's min; row_bcast: rows 1, 3
// take row 0 / 2's, then rows 2, 3 take row 1's).  VALU write -> DPP read of the same VGPR needs two wait states.
__device__ __forceinline__ float vq_wave_min(float v) {
  asm volatile(
      "s_nop 1\n\tv_min_f32_dpp %0, %0, %0 row_ror:1 row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\tv_min_f32_dpp %0, %0, %0 row_ror:2 row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\tv_min_f32_dpp %0, %0, %0 row_ror:4 row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\tv_min_f32_dpp %0, %0, %0 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\tv_min_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
      "s_nop 1\n\tv_min_f32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
      "s_nop 1"
      : "+v"(v));
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}


template <int NP>
__global__ __launch_bounds__(256) void vq_reg_kernel(SdmiVqArgs p) {
  __shared__ float res_d[4][VQ_LAT];
  __shared__ unsigned long long res_b[4][VQ_LAT];
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r0 = blockIdx.x * VQ_LAT;
  const int cbase = w * 128 * NP;
  vq_f2 e0[NP], e1[NP], e2[NP], e3[NP];
#pragma unroll
  for (int q = 0; q < NP; ++q) {
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int j = cbase + q * 128 + lane * 2 + k;
      float a = 0.f, b = 0.f, c = 0.f, n2 = INFINITY;              // pad codes: distance +inf
      if (j < p.n_codes) {
        a = p.codebook[j * 3 + 0]; b = p.codebook[j * 3 + 1]; c = p.codebook[j * 3 + 2];
        const float s0 = a * a, s1 = b * b, s2 = c * c;
        n2 = (s0 + s1) + s2;
      }
      e0[q][k] = a; e1[q][k] = b; e2[q][k] = c; e3[q][k] = n2;
    }
  }
  // every wave holds the workgroup's latents, one per lane
  const int r = r0 + lane;
  const float* zr = p.z + (long long)(r < p.R ? r : 0) * p.ldz;
  const float* z2r = p.z2 ? p.z2 + (long long)(r < p.R ? r : 0) * p.ldz : nullptr;
  const float z0 = vq_chan(p, zr, z2r, 0), z1 = vq_chan(p, zr, z2r, 1), z2 = vq_chan(p, zr, z2r, 2);
  const float zq0 = z0 * z0, zq1 = z1 * z1, zq2 = z2 * z2;
  const float zz = (zq0 + zq1) + zq2;
  float keep_d = INFINITY;
  unsigned keep_lo = 0, keep_hi = 0;
  const vq_f2 m2 = {-2.f, -2.f};
#pragma unroll 2
  for (int i = 0; i < VQ_LAT; ++i) {
    const float s0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(z0), i));
    const float s1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(z1), i));
    const float s2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(z2), i));
    const float sz = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(zz), i));
    const vq_f2 z0v = {s0, s0}, z1v = {s1, s1}, z2v = {s2, s2}, zzv = {sz, sz};
    float ma = INFINITY, mb = INFINITY;
#pragma unroll
    for (int q = 0; q < NP; ++q) {
      vq_f2 dot = z0v * e0[q];
      dot = __builtin_elementwise_fma(z1v, e1[q], dot);
      dot = __builtin_elementwise_fma(z2v, e2[q], dot);
      const vq_f2 d = __builtin_elementwise_fma(m2, dot, zzv + e3[q]);
      if (q & 1) mb = fminf(fminf(mb, d[0]), d[1]);
      else ma = fminf(fminf(ma, d[0]), d[1]);
    }
    const float ml = fminf(ma, mb);
    const float m = vq_wave_min(ml);
    const unsigned long long hit = __ballot(ml == m);
    if (lane == i) { keep_d = m; keep_lo = (unsigned)hit; keep_hi = (unsigned)(hit >> 32); }
  }
  res_d[w][lane] = keep_d;
  res_b[w][lane] = ((unsigned long long)keep_hi << 32) | keep_lo;
  __syncthreads();
  if (tid >= VQ_LAT || r >= p.R) return;
  int wb = 0;
  float best = res_d[0][tid];
#pragma unroll
  for (int v = 1; v < 4; ++v)
    if (res_d[v][tid] < best) { best = res_d[v][tid]; wb = v; }
  unsigned long long mask = res_b[wb][tid];
  int bi = 0x7fffffff;
  while (mask) {
    const int L = __ffsll((long long)mask) - 1;
    mask &= mask - 1;
    for (int q = 0; q < NP; ++q) {
      for (int k = 0; k < 2; ++k) {
        const int j = wb * 128 * NP + q * 128 + L * 2 + k;
        if (j < p.n_codes && j < bi) {
          const float a = p.codebook[j * 3 + 0], b = p.codebook[j * 3 + 1], c = p.codebook[j * 3 + 2];
          const float t0 = a * a, t1 = b * b, t2 = c * c;
          if (vq_dist1(z0, z1, z2, zz, a, b, c, (t0 + t1) + t2) == best) bi = j;
        }
      }
    }
  }
  if (!(zz < INFINITY)) bi = vq_scan_nonfinite(p, z0, z1, z2, zz);
  if (p.idx) p.idx[r] = bi;
  if (p.zq) {
    float* o = p.zq + (long long)r * p.ldz;
    const float c0_ = z0 + (p.codebook[bi * 3 + 0] - z0), c1_ = z1 + (p.codebook[bi * 3 + 1] - z1),
                c2_ = z2 + (p.codebook[bi * 3 + 2] - z2);
    o[0] = c0_ / p.scale; o[1] = c1_ / p.scale; o[2] = c2_ / p.scale;
    for (int c = 3; c < p.ldz; ++c) o[c] = 0.f;
  }
}

}  // namespace

extern "C" int sdmi_vq_nearest(const SdmiVqArgs* a, void* stream) {
  SDMI_REQUIRE(a && a->z && a->codebook && (a->idx || a->zq), "null pointer");
  SDMI_REQUIRE(a->dim == 3, "embed_dim must be 3 (every LDM config)");
  if (a->n_codes <= 4 * 128 * 16 && a->R >= 512) {
    const dim3 grid((a->R + VQ_LAT - 1) / VQ_LAT);
    if (a->n_codes <= 4 * 128 * 4) hipLaunchKernelGGL(vq_reg_kernel<4>, grid, dim3(256), 0, (hipStream_t)stream, *a);
    else if (a->n_codes <= 4 * 128 * 8) hipLaunchKernelGGL(vq_reg_kernel<8>, grid, dim3(256), 0, (hipStream_t)stream, *a);
    else hipLaunchKernelGGL(vq_reg_kernel<16>, grid, dim3(256), 0, (hipStream_t)stream, *a);
    return sdmi_check_launch("vq_nearest (codes in registers)");
  }
  const int n_chunks = ((a->n_codes + 1) / 2 + VQ_CH - 1) / VQ_CH;
  const int smem = n_chunks * VQ_CH * 32;
  SDMI_REQUIRE(a->ldz >= 3 && a->n_codes >= 1 && smem <= 160 * 1024, "bad shape");
  SDMI_OPTIN_LDS(vq_kernel, 160 * 1024, "vq_nearest");
  hipLaunchKernelGGL(vq_kernel, dim3((a->R + 127) / 128), dim3(256), smem, (hipStream_t)stream, *a);
  return sdmi_check_launch("vq_nearest");
}
