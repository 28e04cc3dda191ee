// Probe: LDS write throughput per CU by instruction width / address pattern (all waves of a CU
// hammer the LDS; shader cycles per wave-instruction from s_memtime).
// build: hipcc --offload-arch=gfx950 -O3 tools/probes/ldswrite.hip -o ldswrite.bin
// Measured on MI355X (round 1), ticks per wave-instruction as seen by ONE wave / aggregate B/tick/CU:
//   4 waves: b128 52.0 / 78.8   b64 24-26.5 / 77-85   b32 16.0 / 63.9      (any address pattern)
//   8 waves: b128 52.0 / 157.5  b64 27.0 / 151.6      b32 20.0 / 102.3
// -> a wave's ds_write_b128 issues every 52 cycles regardless of contention: the per-CU rate
//    scales with the number of writing waves (no LDS write-bandwidth wall at 8 waves).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2;

template <int W, int PAT>   // W bytes per lane; PAT 0: contiguous lane*W, 1: rows of 128 B pitch 144, 2: rows of 128 B pitch 128
__global__ __launch_bounds__(512) void k(unsigned long long* out, int iters) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int off;
  if (PAT == 0) off = lane * W;
  else {
    const int per_row = 128 / W, row = lane / per_row, c = lane % per_row;
    off = row * (PAT == 1 ? 144 : 128) + c * W;
  }
  const unsigned p = wave * 16384 + off;     // dynamic LDS starts at byte 0
  u32x4 v = {(unsigned)lane, 1u, 2u, 3u};
  u32x2 v2 = {v[0], v[1]};
  if (threadIdx.x == 9999) smem[0] = 1;
  __syncthreads();
  unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const unsigned q = p + i * 1152;
      if (W == 16) asm volatile("ds_write_b128 %0, %1" ::"v"(q), "v"(v) : "memory");
      else if (W == 8) asm volatile("ds_write_b64 %0, %1" ::"v"(q), "v"(v2) : "memory");
      else asm volatile("ds_write_b32 %0, %1" ::"v"(q), "v"(v[0]) : "memory");
    }
  }
  __builtin_amdgcn_s_waitcnt(0);
  unsigned long long t1 = __builtin_readcyclecounter();
  if (lane == 0) out[blockIdx.x * 8 + wave] = t1 - t0;
}

template <int W, int PAT>
void run(const char* name, int nthreads) {
  unsigned long long* out;
  hipMalloc(&out, 256 * 8 * 8);
  const int iters = 2000;
  hipFuncSetAttribute((const void*)k<W, PAT>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  k<W, PAT><<<256, nthreads, 140 * 1024>>>(out, iters);
  hipDeviceSynchronize();
  k<W, PAT><<<256, nthreads, 140 * 1024>>>(out, iters);
  unsigned long long h[8];
  hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
  const int nw = nthreads / 64;
  double cyc = (double)h[0] / (iters * 8.0);           // cycles per instr per wave (with nw waves contending)
  // s_memtime counts at 100 MHz constant? print raw too
  printf("%-28s waves=%d  ticks/instr/wave %7.2f  -> per-CU: %6.2f ticks per wave-instr, %6.1f B/tick\n", name, nw, cyc,
         cyc / nw, 64.0 * W * nw / cyc);
  hipFree(out);
}

int main() {
  for (int nt : {256, 512}) {
    run<16, 0>("b128 contiguous", nt);
    run<16, 1>("b128 rows pitch144", nt);
    run<16, 2>("b128 rows pitch128", nt);
    run<8, 0>("b64 contiguous", nt);
    run<8, 1>("b64 rows pitch144", nt);
    run<4, 0>("b32 contiguous", nt);
    run<4, 1>("b32 rows pitch144", nt);
  }
  return 0;
}
