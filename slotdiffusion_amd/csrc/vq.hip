// VQ nearest-code search (include/sdmi.h: sdmi_vq_nearest; reference vqvae/quantize.py:85-94).
//
// Integer output => bit-exact contract.  The reference evaluates, in fp32 on the CPU,
//     d[r][j] = (|z_r|^2 + |e_j|^2) - 2 * (z_r . e_j),      argmin_j (first minimum)
// with |.|^2 = (x0*x0 + x1*x1) + x2*x2 (separately rounded products) and the dot product as the
// FMA chain fma(z2,e2, fma(z1,e1, z0*e0)) (probed against torch 2.10 CPU; tests/test_vq.py keeps
// the probe).  This file is compiled with -ffp-contract=off so exactly those roundings happen.
// The codebook (n_codes x {e0,e1,e2,|e|^2} = 64 KiB for 4096 codes) lives in LDS; two threads share
// a latent, each scanning half of the codes with ds_read_b128 (two distinct addresses per wave).
#include "common.h"

namespace {

__global__ __launch_bounds__(256) void vq_kernel(SdmiVqArgs p) {
  extern __shared__ __attribute__((aligned(16))) float cb[];  // [n_codes][4]
  for (int j = threadIdx.x; j < p.n_codes; j += blockDim.x) {
    const float e0 = p.codebook[j * 3 + 0], e1 = p.codebook[j * 3 + 1], e2 = p.codebook[j * 3 + 2];
    const float s0 = e0 * e0, s1 = e1 * e1, s2 = e2 * e2;
    cb[j * 4 + 0] = e0; cb[j * 4 + 1] = e1; cb[j * 4 + 2] = e2;
    cb[j * 4 + 3] = (s0 + s1) + s2;
  }
  __syncthreads();
  // two threads per latent (each scans half of the codes, four independent running minima for
  // instruction-level parallelism); partial results merge on (distance, index) -- the smallest
  // distance, on ties the smallest index: the reference's "first minimum" of the sequential scan
  const int r = blockIdx.x * (blockDim.x / 2) + (threadIdx.x >> 1);
  const int half = threadIdx.x & 1;
  const bool live = r < p.R;
  const float* zr = p.z + (long long)(live ? r : 0) * p.ldz;
  const float z0 = zr[0] * p.scale, z1 = zr[1] * p.scale, z2 = zr[2] * p.scale;
  const float q0 = z0 * z0, q1 = z1 * z1, q2 = z2 * z2;
  const float zz = (q0 + q1) + q2;
  const int per = (p.n_codes + 1) / 2;
  const int j0 = half * per, j1 = min(p.n_codes, j0 + per);
  float bd[4] = {INFINITY, INFINITY, INFINITY, INFINITY};
  int bj[4] = {0, 0, 0, 0};
  int j = j0;
  for (; j + 3 < j1; j += 4) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const f32x4 e = *reinterpret_cast<const f32x4*>(cb + (j + u) * 4);
      const float dot = fmaf(z2, e[2], fmaf(z1, e[1], z0 * e[0]));
      const float d = (zz + e[3]) - 2.f * dot;
      if (d < bd[u]) { bd[u] = d; bj[u] = j + u; }
    }
  }
  for (; j < j1; ++j) {
    const f32x4 e = *reinterpret_cast<const f32x4*>(cb + j * 4);
    const float dot = fmaf(z2, e[2], fmaf(z1, e[1], z0 * e[0]));
    const float d = (zz + e[3]) - 2.f * dot;
    if (d < bd[0] || (d == bd[0] && j < bj[0])) { bd[0] = d; bj[0] = j; }
  }
  float best = bd[0];
  int bi = bj[0];
#pragma unroll
  for (int u = 1; u < 4; ++u)
    if (bd[u] < best || (bd[u] == best && bj[u] < bi)) { best = bd[u]; bi = bj[u]; }
  {
    const float ob = __shfl_xor(best, 1, 64);
    const int oi = __shfl_xor(bi, 1, 64);
    if (ob < best || (ob == best && oi < bi)) { best = ob; bi = oi; }
  }
  if (!live || half) return;
  if (p.idx) p.idx[r] = bi;
  if (p.zq) {
    float* o = p.zq + (long long)r * p.ldz;
    const float c0 = z0 + (cb[bi * 4 + 0] - z0), c1 = z1 + (cb[bi * 4 + 1] - z1),
                c2 = z2 + (cb[bi * 4 + 2] - z2);
    o[0] = c0 / p.scale; o[1] = c1 / p.scale; o[2] = c2 / p.scale;
    for (int c = 3; c < p.ldz; ++c) o[c] = 0.f;
  }
}

}  // namespace

extern "C" int sdmi_vq_nearest(const SdmiVqArgs* a, void* stream) {
  SDMI_REQUIRE(a && a->z && a->codebook && (a->idx || a->zq), "null pointer");
  SDMI_REQUIRE(a->dim == 3, "embed_dim must be 3 (every LDM config)");
  SDMI_REQUIRE(a->ldz >= 3 && a->n_codes >= 1 && a->n_codes * 16 <= 160 * 1024, "bad shape");
  SDMI_OPTIN_LDS(vq_kernel, 160 * 1024, "vq_nearest");
  hipLaunchKernelGGL(vq_kernel, dim3((a->R + 127) / 128), dim3(256), a->n_codes * 16,
                     (hipStream_t)stream, *a);
  return sdmi_check_launch("vq_nearest");
}
