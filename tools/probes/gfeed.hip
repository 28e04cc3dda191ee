// Probe: global->LDS feed rate per CU by access pattern (no MFMA, no readers): 4 loader waves per
// WG, 2 WGs per CU, two register sets of loads in flight (like igemm's loaders), ds_write_b128.
//   PAT 0: 1 KB contiguous per wave-instr
//   PAT 1: 8 rows x 128 B per wave-instr, row pitch `pitch` bytes, K-tile kt reads 128-B chunk (kt % (pitch/128)) of the rows
// Measured on MI355X (round 1), 512 workgroups x 400 K tiles of 32 KB, 2 workgroups per CU:
//   L2-resident footprint (16 MB): 24-27 TB/s = 94-106 GB/s/CU for contiguous 1 KB pieces and for
//   128-byte rows at pitch 256 ... 8192 B alike (no channel camping);
//   HBM streaming (1 GB footprint, contiguous): 6.5 TB/s = 25.5 GB/s/CU;
//   many workgroups on the SAME rows (weights-like sharing, pitch 4608-9344): 12-18 TB/s.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

template <int PAT>
__global__ __launch_bounds__(256) void feed(const char* __restrict__ src, float* out, int iters, int pitch, long long ws_mask) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  // per WG tile: 256 rows x 128 B = 32 KB per K tile; thread t: rows t/8 + 32*i, chunk t%8
  u32x4 r0[8], r1[8];
  const int cpr = pitch / 128;            // chunks per row
  auto load = [&](u32x4 (&r)[8], int kt) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      long long off;
      if (PAT == 0) off = ((long long)blockIdx.x * 64 + kt) * 32768 + i * 4096 + tid * 16;
      else {
        const int row = tid / 8 + 32 * i;
        const int tap = kt / cpr, ch = kt % cpr;
        const int nb = max(1, 512 * 128 / pitch);   // distinct WG bases: constant 16 MB footprint
        off = ((long long)(blockIdx.x % nb) * 256 + (row + tap * 37) % 256) * pitch + ch * 128 + (tid % 8) * 16;
      }
      r[i] = *reinterpret_cast<const u32x4*>(src + (off & ws_mask));
    }
  };
  auto store = [&](const u32x4 (&r)[8], int st) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < 8; ++i) *reinterpret_cast<u32x4*>(smem + st * 32768 + i * 4096 + tid * 16) = r[i];
  };
  load(r0, 0);
  load(r1, 1);
  for (int kt = 0; kt < iters; kt += 2) {
    store(r0, 0);
    load(r0, kt + 2);
    __syncthreads();
    store(r1, 1);
    load(r1, kt + 3);
    __syncthreads();
  }
  out[blockIdx.x * 256 + tid] = smem[tid * 4] + (float)r0[0][0] + (float)r1[0][0];
}

template <int PAT>
void run(const char* name, const char* src, float* out, int pitch, long long mask) {
  const int iters = 400, grid = 512;
  hipFuncSetAttribute((const void*)feed<PAT>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  feed<PAT><<<grid, 256, 65536>>>(src, out, iters, pitch, mask);
  hipEventRecord(a);
  for (int r = 0; r < 5; ++r) feed<PAT><<<grid, 256, 65536>>>(src, out, iters, pitch, mask);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b); ms /= 5;
  double bytes = (double)grid * iters * 32768;
  printf("%-34s ws=%5lld MB  %7.3f ms  %6.2f TB/s  %6.1f GB/s/CU\n", name, (mask + 1) >> 20, ms, bytes / ms * 1e-9, bytes / ms * 1e-6 / 256);
}

int main() {
  char* src; float* out;
  const long long N = 1ll << 30;
  hipMalloc(&src, N); hipMemset(src, 0, N);
  hipMalloc(&out, 512 * 256 * 4);
  const long long mask = (1ll << 30) - 1;
  run<0>("contiguous 1 KB pieces", src, out, 128, (32ll << 20) - 1);
  for (int pitch : {512, 1152 * 2, 2304 * 2, 2304 * 2 + 128, 4608 * 2, 4608 * 2 + 128, 3456 * 2, 576 * 2, 256, 1024, 2048, 4096, 8192}) {
    char name[64];
    snprintf(name, sizeof name, "rows 128 B, pitch %d", pitch);
    run<1>(name, src, out, pitch, mask);
  }
  return 0;
}
