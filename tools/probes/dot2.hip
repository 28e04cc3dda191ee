// probe: v_dot2c_f32_bf16 semantics on gfx950 (sum and sum of squares of packed bf16 pairs)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <cstring>
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
__global__ void k(const unsigned* x, float* y, int n) {
  float sx = 0.f, sxx = 0.f;
  const bf16x2 ones = __builtin_bit_cast(bf16x2, 0x3F803F80u);
  for (int i = 0; i < n; ++i) {
    bf16x2 a = __builtin_bit_cast(bf16x2, x[i]);
    sx = __builtin_amdgcn_fdot2_f32_bf16(a, ones, sx, false);
    sxx = __builtin_amdgcn_fdot2_f32_bf16(a, a, sxx, false);
  }
  y[0] = sx; y[1] = sxx;
}
static unsigned short f2b(float f) { unsigned u; std::memcpy(&u, &f, 4); u += 0x7fff + ((u >> 16) & 1); return u >> 16; }
static float b2f(unsigned short b) { unsigned u = (unsigned)b << 16; float f; std::memcpy(&f, &u, 4); return f; }
int main() {
  const int n = 256;
  unsigned h[n]; double sx = 0, sxx = 0;
  srand(1);
  for (int i = 0; i < n; ++i) {
    float a = (rand() / (float)RAND_MAX - 0.3f) * 3.f, b = (rand() / (float)RAND_MAX - 0.3f) * 3.f;
    unsigned short ba = f2b(a), bb = f2b(b);
    h[i] = ba | ((unsigned)bb << 16);
    sx += b2f(ba) + b2f(bb); sxx += (double)b2f(ba) * b2f(ba) + (double)b2f(bb) * b2f(bb);
  }
  unsigned* dx; float* dy; float out[2];
  hipMalloc(&dx, sizeof(h)); hipMalloc(&dy, 8);
  hipMemcpy(dx, h, sizeof(h), hipMemcpyHostToDevice);
  k<<<1, 1>>>(dx, dy, n);
  hipMemcpy(out, dy, 8, hipMemcpyDeviceToHost);
  printf("sum  gpu %.6f exact %.6f\nsumsq gpu %.6f exact %.6f\n", out[0], sx, out[1], sxx);
  return 0;
}
