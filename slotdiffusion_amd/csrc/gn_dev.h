// Device helpers of the second-form single-pass GroupNorm kernels (forward: norm.hip, backward: norm_bwd.hip).
#pragma once
__device__ __forceinline__ float dpp_ror_add(float v, int ctrl_sel) {
  // v + (v rotated right by 1 / 2 / 4 / 8 lanes inside each row of 16 lanes)
  int r;
  const int iv = __float_as_int(v);
  switch (ctrl_sel) {
    case 1: r = __builtin_amdgcn_update_dpp(0, iv, 0x121, 0xf, 0xf, false); break;
    case 2: r = __builtin_amdgcn_update_dpp(0, iv, 0x122, 0xf, 0xf, false); break;
    case 4: r = __builtin_amdgcn_update_dpp(0, iv, 0x124, 0xf, 0xf, false); break;
    default: r = __builtin_amdgcn_update_dpp(0, iv, 0x128, 0xf, 0xf, false); break;
  }
  return v + __int_as_float(r);
}
// all-reduce over the lanes of a wave that share (lane mod CVp); CVp a power of two < 64 (wave-uniform)
__device__ __forceinline__ float col_allreduce(float v, int CVp) {
  if (CVp <= 1) v = dpp_ror_add(v, 1);
  if (CVp <= 2) v = dpp_ror_add(v, 2);
  if (CVp <= 4) v = dpp_ror_add(v, 4);
  if (CVp <= 8) v = dpp_ror_add(v, 8);
  if (CVp <= 16) v += __shfl_xor(v, 16, 64);
  if (CVp <= 32) v += __shfl_xor(v, 32, 64);
  return v;
}
template <int VEC>
__device__ __forceinline__ void load_fvec(const float* q, float* out) {
  if ((reinterpret_cast<uintptr_t>(q) & 15) == 0) {
#pragma unroll
    for (int h = 0; h < VEC / 4; ++h) {
      const float4 v = *reinterpret_cast<const float4*>(q + 4 * h);
      out[4 * h] = v.x; out[4 * h + 1] = v.y; out[4 * h + 2] = v.z; out[4 * h + 3] = v.w;
    }
  } else {
#pragma unroll
    for (int j = 0; j < VEC; ++j) out[j] = q[j];
  }
}


// The same all-reduce on fp64 values (group sums): the two 32-bit halves travel separately, the add is fp64.
__device__ __forceinline__ double dpp_ror_add_d(double v, int ctrl_sel) {
  const long long iv = __double_as_longlong(v);
  const int lo = (int)iv, hi = (int)(iv >> 32);
  int rl, rh;
  switch (ctrl_sel) {
    case 1: rl = __builtin_amdgcn_update_dpp(0, lo, 0x121, 0xf, 0xf, false); rh = __builtin_amdgcn_update_dpp(0, hi, 0x121, 0xf, 0xf, false); break;
    case 2: rl = __builtin_amdgcn_update_dpp(0, lo, 0x122, 0xf, 0xf, false); rh = __builtin_amdgcn_update_dpp(0, hi, 0x122, 0xf, 0xf, false); break;
    case 4: rl = __builtin_amdgcn_update_dpp(0, lo, 0x124, 0xf, 0xf, false); rh = __builtin_amdgcn_update_dpp(0, hi, 0x124, 0xf, 0xf, false); break;
    default: rl = __builtin_amdgcn_update_dpp(0, lo, 0x128, 0xf, 0xf, false); rh = __builtin_amdgcn_update_dpp(0, hi, 0x128, 0xf, 0xf, false); break;
  }
  return v + __longlong_as_double(((long long)rh << 32) | (unsigned int)rl);
}
__device__ __forceinline__ double col_allreduce(double v, int CVp) {
  if (CVp <= 1) v = dpp_ror_add_d(v, 1);
  if (CVp <= 2) v = dpp_ror_add_d(v, 2);
  if (CVp <= 4) v = dpp_ror_add_d(v, 4);
  if (CVp <= 8) v = dpp_ror_add_d(v, 8);
  if (CVp <= 16) v += __shfl_xor(v, 16, 64);
  if (CVp <= 32) v += __shfl_xor(v, 32, 64);
  return v;
}
