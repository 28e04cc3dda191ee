// Geometry of the single-pass GroupNorm kernels (forward and backward share it).
#pragma once
#include <stdlib.h>

// Geometry of the single-pass kernels, shared by forward and backward: workgroup width T and channel
// split S (whole groups per chunk).  These kernels are latency chains (load slab -> reduce -> LDS
// combine -> apply -> store), so what pays is parallelism, not line-perfect coalescing (sweep on
// MI355X, profiles/r02_gn_geometry_sweep.txt): candidates keep row segments >= 64 bytes and >= 8 KB
// of slab per workgroup; among them take the largest split whose slab needs <= 4 vectors per
// thread with the narrowest workgroup that achieves it, else the fewest vectors per thread.
struct GnGeom { int T, S, need; };   // need = vectors per thread the slab takes
// min_slab: smallest slab (bytes of 16-byte vectors) a channel chunk may shrink to -- the partials-source form of the
// forward kernel reads 4 x splits the bytes per element and wants the read spread over more workgroups.
static inline GnGeom gn_pick(int B, int HW, int C, int groups, int vec, int nv_of_T[3], int min_slab = 8192) {
  constexpr int env_T = 0, env_S = 0;       // (the sweep that set the rule below: profiles/r02_gn_geometry_sweep.txt)
  (void)B;
  const int cvt = C / vec;
  static const int Ts[3] = {256, 512, 1024};
  GnGeom best{0, 0, 0};
  long long best_score = -1;
  for (int S = 1; S <= groups && S <= 32; S <<= 1) {
    if (groups % S || cvt % S) break;
    if (env_S && env_S != S) continue;
    const int cv = cvt / S;
    if (S > 1 && cv * 16 < 64) break;                          // row segments shorter than 64 B
    if (S > 1 && (long long)HW * cv * 16 < min_slab) break;    // < 8 KB of slab per workgroup
    int cvp = 1;
    while (cvp < cv) cvp <<= 1;
    for (int ti = 0; ti < 3; ++ti) {
      const int T = Ts[ti], nv = nv_of_T[ti];
      if (nv <= 0 || (env_T && env_T != T) || cvp > T) continue;
      const int R = T / cvp;
      const int need = (HW + R - 1) / R;
      if (need > nv) continue;                                  // slab does not fit
      // need <= 4: larger S first, then narrower T; otherwise fewer vectors per thread first
      const long long score = need <= 4 ? 1000000 + S * 100 - ti : 1000 - need * 10 + S;
      if (score > best_score) { best_score = score; best = GnGeom{T, S, need}; }
    }
  }
  return best;
}
