// Probe of ds_read_b64_tr_b16 lane semantics on gfx950 (build: hipcc --offload-arch=gfx950).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short v4s __attribute__((ext_vector_type(4)));
#define LDSP(p) ((__attribute__((address_space(3))) v4s*)(p))
__global__ void k(short* out, int mode) {
  __shared__ short lds[8192];
  for (int i = threadIdx.x; i < 8192; i += 64) lds[i] = i;
  __syncthreads();
  const int l = threadIdx.x, g = l >> 4, t = l & 15;
  v4s v;
  if (mode == 0) {
    v = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDSP(lds + l * 4));
  } else {
    // row-major image, pitch 160 elements: group g reads rows [g*4, g*4+4), cols [0,16)
    const int PITCH = 160;
    v = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDSP(lds + (g * 4 + t / 4) * PITCH + (t % 4) * 4));
  }
  for (int j = 0; j < 4; ++j) out[l * 4 + j] = v[j];
}
int main() {
  short* d; hipMalloc(&d, 64 * 4 * 2);
  short h[256];
  for (int mode = 0; mode < 2; ++mode) {
    k<<<1, 64>>>(d, mode);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("mode %d\n", mode);
    for (int l = 0; l < 64; ++l) printf("lane %2d: %5d %5d %5d %5d\n", l, h[l*4], h[l*4+1], h[l*4+2], h[l*4+3]);
  }
  return 0;
}
