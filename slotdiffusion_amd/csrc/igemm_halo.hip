// Halo-staged 3x3 implicit GEMM (igemm_halo.h: 256 x 128 tiles, the activation patch of a tile staged once per 64-channel
// chunk) -- its own translation unit: sdmi_igemm's dispatch (igemm.hip) calls sdmi_launch_halo for the shapes that take it.
#include "igemm_halo.h"

int sdmi_launch_halo(const SdmiGemmArgs& p, int logw, int nj, int hw_shift, hipStream_t st, int n_cu) {
  (void)nj;           // (256 x 64 tiles lost to 256 x 128 and to conv3x3_c64_kernel: igemm.hip's dispatch note)
  if (logw == 4) return launch_halo<4, 2>(p, hw_shift, st, n_cu);
  if (logw == 5) return launch_halo<5, 2>(p, hw_shift, st, n_cu);
  return launch_halo<6, 2>(p, hw_shift, st, n_cu);
}
