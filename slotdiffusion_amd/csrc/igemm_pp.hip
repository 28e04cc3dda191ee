// 256 x 128 ping-pong implicit GEMMs (igemm_pp.h, igemm_halo.h) -- their own translation unit: sdmi_igemm's dispatch
// (igemm.hip) calls sdmi_launch_pp / sdmi_launch_halo for the shapes that take them.
#include "igemm_pp.h"
#include "igemm_halo.h"

int sdmi_launch_pp(const SdmiGemmArgs& p, bool is1x1, int hw_shift, hipStream_t st, int n_cu) {
  if (p.a2) return is1x1 ? launch_pp<1, true>(p, hw_shift, st, n_cu) : launch_pp<2, true>(p, hw_shift, st, n_cu);
  return is1x1 ? launch_pp<1>(p, hw_shift, st, n_cu) : launch_pp<2>(p, hw_shift, st, n_cu);
}

int sdmi_launch_halo(const SdmiGemmArgs& p, int logw, int nj, int hw_shift, hipStream_t st, int n_cu) {
  if (nj == 1) {
    if (logw == 4) return launch_halo<4, 1>(p, hw_shift, st, n_cu);
    if (logw == 5) return launch_halo<5, 1>(p, hw_shift, st, n_cu);
    return launch_halo<6, 1>(p, hw_shift, st, n_cu);
  }
  if (logw == 4) return launch_halo<4, 2>(p, hw_shift, st, n_cu);
  if (logw == 5) return launch_halo<5, 2>(p, hw_shift, st, n_cu);
  return launch_halo<6, 2>(p, hw_shift, st, n_cu);
}
