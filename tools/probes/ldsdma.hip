// Probe: cost of filling LDS with global_load_lds_dwordx4 (LDS-DMA) vs global_load + ds_write_b128
// while 4 other waves of the workgroup read fragments with ds_read_b128 and issue MFMAs.
// build: hipcc --offload-arch=gfx950 -O3 tools/probes/ldsdma.hip -o /tmp/ldsdma
// Measured on MI355X (round 1), ns per K-tile iteration of a CU, 2 workgroups per CU (1 per CU):
//   readers only (16 ds_read_b128 + 16 MFMA per wave) 530 (340)    MFMA only 433 (227)    LDS reads only 301
//   fill only: LDS-DMA 426 (387), load + ds_write_b128 560 (460)
//   fill + readers, VALU address arithmetic in the loaders: DMA 880-920 (590), ds_write 880-905 (515)
//   LDS-DMA + MFMA only: 800 (526) -- the two ADD UP; LDS-DMA + LDS reads only: 440-527 (overlap)
//   sleep-loader + MFMA 380, DMA + sleep-reader 396 (1 per CU): the barrier structure overlaps fine
//   LDS-DMA with SCALAR-only loaders (buffer_load ... lds, fixed voffset, soffset walk):
//       + MFMA only 547 (394), + full readers 545-555 (396)  -> MFMA bound
//   register-staged buffer loads (scalar addressing) + ds_write_b128 + readers: 749
//   s_setprio(3) in the loaders: no change.  Shader clock 2.2-2.4 GHz in every mode.
// => VALU instructions of a loader wave serialise with the MFMAs of the wave next to it on the SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) const void g_void;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

// MODE 0: DMA   1: load + ds_write_b128   2: no fill (readers only)   3: DMA, no readers  4: ds_write, no readers
// 5: DMA + MFMA-only readers (no LDS reads)   6: DMA + LDS-read-only readers (no MFMA)   7: no fill, MFMA only   8: no fill, LDS reads only
__device__ unsigned long long clk[2];
template <int MODE>
__global__ __launch_bounds__(512) void probe(const u32x4* __restrict__ src, float* __restrict__ out,
                                             int iters, int src_mask) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int STAGE = 32768;     // 256 rows x 128 B
  const int tid = threadIdx.x;
  if (tid >= 256) {
    const int lt = tid - 256, lane = lt & 63, lw = lt >> 6;
#ifdef PRIO
    __builtin_amdgcn_s_setprio(3);
#endif
    if (MODE == 9) { for (int it = 0; it < iters; ++it) { for (int q = 0; q < 7; ++q) __builtin_amdgcn_s_sleep(2); __syncthreads(); } return; }
    if (MODE == 11 || MODE == 12 || MODE == 13) {
      // SALU-only loop: buffer resource + loop-invariant voffset, the advance goes through soffset
      __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, 0x7fffffff, 0x00020000);
      const int voff = (((blockIdx.x * 2048 + lw * 512) & src_mask) + lane) * 16;
      for (int it = 0; it < iters; ++it) {
        char* st = smem + (it & 1) * STAGE + lw * 8192;
        const int so = ((it * 512) & src_mask) * 16;
#define BL(i) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void*)(st + i * 1024), 16, voff, so, i * 1024, 0)
        BL(0); BL(1); BL(2); BL(3); BL(4); BL(5); BL(6); BL(7);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
      }
      return;
    }
    if (MODE == 14) {
      // register-staged, SALU-only addressing: buffer_load -> VGPR -> ds_write_b128
      __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, 0x7fffffff, 0x00020000);
      const int voff = (((blockIdx.x * 2048 + lw * 512) & src_mask) + lane) * 16;
      for (int it = 0; it < iters; ++it) {
        char* st = smem + (it & 1) * STAGE + lw * 8192 + lane * 16;
        const int so = ((it * 512) & src_mask) * 16;
        u32x4 r[8];
#define BR(i) r[i] = __builtin_amdgcn_raw_buffer_load_b128(rs, voff + i * 1024, so, 0)
        BR(0); BR(1); BR(2); BR(3); BR(4); BR(5); BR(6); BR(7);
#pragma unroll
        for (int i = 0; i < 8; ++i) *reinterpret_cast<u32x4*>(st + i * 1024) = r[i];
        __syncthreads();
      }
      return;
    }
    if (MODE == 2 || MODE == 7 || MODE == 8) { for (int it = 0; it < iters; ++it) __syncthreads(); return; }
    // each loader wave fills 8 KB per stage: 8 pieces of 1 KB
    const u32x4* g = src + ((blockIdx.x * 2048 + lw * 512 + lane) & src_mask);
    for (int it = 0; it < iters; ++it) {
      char* st = smem + (it & 1) * STAGE + lw * 8192;
      if (MODE == 0 || MODE == 3 || MODE == 5 || MODE == 6 || MODE == 10) {
#pragma unroll
        for (int i = 0; i < 8; ++i)
          __builtin_amdgcn_global_load_lds((g_void*)(g + ((it * 8 + i) * 64 & src_mask)), (lds_void*)(st + i * 1024), 16, 0, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      } else {
        u32x4 r[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) r[i] = g[(it * 8 + i) * 64 & src_mask];
#pragma unroll
        for (int i = 0; i < 8; ++i) *reinterpret_cast<u32x4*>(st + i * 1024 + lane * 16) = r[i];
      }
      __syncthreads();
    }
    return;
  }
  const int lane = tid & 63, wave = tid >> 6;
  const unsigned long long c0 = __builtin_readcyclecounter(), w0 = wall_clock64();
  f32x16 acc[2][2];
  for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  const bool readers = MODE <= 2 || (MODE >= 5 && MODE != 10 && MODE != 13);
  constexpr bool DO_READ = !(MODE == 5 || MODE == 7 || MODE == 9 || MODE == 11), DO_MFMA = !(MODE == 6 || MODE == 8);
  u32x4 keep = {0u, 0u, 0u, 0u};
  for (int it = 0; it < iters; ++it) {
    __syncthreads();
    if (MODE == 10) { for (int q = 0; q < 4; ++q) __builtin_amdgcn_s_sleep(2); }
    if (readers) {
      const char* base = smem + ((it + 1) & 1) * STAGE;
      const int R = lane & 31;
      const char* As = base + ((wave >> 1) * 64 + R) * 128;
      const char* Bs = base + 16384 + ((wave & 1) * 64 + R) * 128;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const int c = ks * 2 + (lane >> 5);
        const int sw = ((c ^ (R >> 1)) & 7) * 16;
        u32x4 fa[2], fb[2];
        if (DO_READ) {
          fa[0] = *reinterpret_cast<const u32x4*>(As + sw);
          fa[1] = *reinterpret_cast<const u32x4*>(As + 32 * 128 + sw);
          fb[0] = *reinterpret_cast<const u32x4*>(Bs + sw);
          fb[1] = *reinterpret_cast<const u32x4*>(Bs + 32 * 128 + sw);
        } else {
          fa[0] = fa[1] = fb[0] = fb[1] = keep;
        }
        if (DO_MFMA) {
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fa[i]),
                                                                 __builtin_bit_cast(bf16x8, fb[j]), acc[i][j], 0, 0, 0);
        } else {
          keep = keep ^ fa[0] ^ fa[1] ^ fb[0] ^ fb[1];
        }
      }
    }
  }
  float s = (float)keep[0];
  for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) s += acc[i][j][r];
  out[blockIdx.x * 256 + tid] = s;
  if (blockIdx.x == 0 && tid == 0) { clk[0] = __builtin_readcyclecounter() - c0; clk[1] = wall_clock64() - w0; }
}

template <int MODE>
float run(const u32x4* src, float* out, int iters, int mask, int grid) {
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  hipFuncSetAttribute((const void*)probe<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  probe<MODE><<<grid, 512, 65536>>>(src, out, iters, mask);
  hipEventRecord(a);
  for (int r = 0; r < 5; ++r) probe<MODE><<<grid, 512, 65536>>>(src, out, iters, mask);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  unsigned long long h[2]; hipMemcpyFromSymbol(h, HIP_SYMBOL(clk), sizeof h);
  printf("   [mode %d: %llu shader ticks / %llu wall ticks(100MHz) = %.0f MHz]\n", MODE, h[0], h[1], h[1] ? 100.0 * h[0] / h[1] : 0.0);
  return ms / 5;
}

int main() {
  const int n = 1 << 22;   // 64 MB of uint4; mask selects the working set
  u32x4* src; float* out;
  hipMalloc(&src, (size_t)n * 16); hipMemset(src, 0, (size_t)n * 16);
  hipMalloc(&out, 512 * 256 * 4 * 8);
  const int iters = 2000, grid = GRID;
  const char* names[] = {"dma+readers", "dswrite+readers", "readers only", "dma only", "dswrite only", "dma+mfma-only", "dma+ldsread-only", "mfma only", "ldsread only", "sleep-loader+mfma", "dma+sleep-reader", "bufdma(salu)+mfma-only", "bufdma(salu)+readers", "bufdma(salu) only", "bufload(salu)+dswrite+readers"};
  for (int mask_bits : {14}) {   // 256 KB (L2), 4 MB, 32 MB working sets
    int mask = (1 << mask_bits) - 1;
    float t[15] = {run<0>(src, out, iters, mask, grid), run<1>(src, out, iters, mask, grid), run<2>(src, out, iters, mask, grid),
                  run<3>(src, out, iters, mask, grid), run<4>(src, out, iters, mask, grid), run<5>(src, out, iters, mask, grid),
                  run<6>(src, out, iters, mask, grid), run<7>(src, out, iters, mask, grid), run<8>(src, out, iters, mask, grid), run<9>(src, out, iters, mask, grid), run<10>(src, out, iters, mask, grid), run<11>(src, out, iters, mask, grid), run<12>(src, out, iters, mask, grid), run<13>(src, out, iters, mask, grid), run<14>(src, out, iters, mask, grid)};
    for (int m = 0; m < 15; ++m)
      printf("ws=%6d KB  %-16s %8.3f ms  %7.1f ns/iter/WG-pair  (32 KB per iter per WG: %6.1f B/clk/CU @2.0GHz eq)\n",
             (1 << mask_bits) * 16 / 1024, names[m], t[m], t[m] * 1e6 / iters, 2 * 32768.0 / (t[m] * 1e6 / iters * 2.0));
  }
  return 0;
}
