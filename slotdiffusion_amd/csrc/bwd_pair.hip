// Data gradient AND weight gradient of one conv / linear layer in ONE launch (include/sdmi.h:
// sdmi_bwd_pair).  Reference semantics: the autograd backward of unet.py:271-285 (ResBlock convolutions),
// attention.py:182-206, 247-251 (projection / feed-forward Linear layers), slot_attention.py:57-106.
//
// Both gradients read the same dY right after the backward chain produced it:
//     dX[m][c] = sum_n dY[m][n] W[n][c]        (implicit GEMM over the flipped operand: igemm_body.h)
//     dW[n][k] = sum_m dY[m][n] A[m][k]        (contraction over the B*Ho*Wo rows:        wgrad_body.h)
// Launched separately, the weight gradient sat on a side HIP stream: one fork and one join edge per layer
// in the captured train-step graph (~250 forks x ~16 us) plus a fold launch for its
// M-split partials.  Here one grid hosts three kinds of 512-thread workgroups, in block-index order:
//   [0, n_fold)                 fold the M-split partials of the PREVIOUS layer's weight gradient into the
//                               gradient arena (they were complete when that launch ended; no fence, no
//                               atomics, deterministic split order) -- short, so they leave first;
//   [n_fold, n_fold + n_wgrad)  weight-gradient tiles of THIS layer, (split, 128 x 128 output tile) each,
//                               one deep contraction per workgroup; bias-gradient tiles trail each split;
//   [d_begin, d_begin+n_dgrad)  persistent data-gradient workgroups walking the dX tiles (d_begin is a
//                               multiple of 8: block b still runs on XCD b % 8, so igemm's XCD-aware tile
//                               order holds).
// The weight-gradient workgroups fill the CUs a small-M data gradient leaves idle; with M split so that a
// weight-gradient workgroup and a data-gradient workgroup walk about the same number of 64-deep MFMA
// steps, both kinds end together.  Register budget 128 (two workgroups per CU, as both bodies had alone);
// LDS = max of the two double-buffered images = 80 KB: exactly two workgroups per 160 KB CU.
#include "common.h"
#include "igemm_body.h"
#include "wgrad_body.h"

namespace {

struct PairGeom {
  int n_fold, n_wgrad, d_begin, n_dgrad;
  int w_tiles_n, w_tiles_k, w_per_split, w_mps;
  int d_tiles_m, d_tiles_n, d_ktps, d_hw_shift;
};

constexpr int pair_wg_lds(int wt) { return 2 * 64 * (wt * 2 + 64 + wt * 2 + 64); }   // wgrad_tr_body<WT, WT>: 80 / 48 KB

// WT: output tile of the weight gradient, 128 (two workgroups per CU next to 128 x 128 data-gradient tiles)
// or 64 for small layers: four times the workgroups per dW element, each a quarter of the MFMA work per
// 64-row step, 48 KB of LDS (three workgroups per CU next to 64 x 64 data-gradient tiles).
template <int BM, int BN, int BKB, int MODE, int WT>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(4, 4))) void bwd_pair_kernel(
    SdmiGemmArgs d, SdmiWgradArgs w, SdmiWgradArgs f, PairGeom g) {
  const int b = (int)blockIdx.x;
  if (b < g.n_fold) {
    wgrad_reduce_body<512>(f, b, g.n_fold);
    return;
  }
  if (b < g.n_fold + g.n_wgrad) {
    // XCD-aware order (round 4): block b runs on XCD b % 8, so consecutive indices used to spread the ~40 dW tiles
    // of ONE M split -- which all read the same rows of x and dY -- over all eight L2s (the 400 MB per launch the
    // round-3 counters showed against ~100 MB of operands).  Each XCD now owns a contiguous range of (split, tile)
    // indices: a split's rows are fetched into one or two L2s.  Same work per workgroup, bit-identical results.
    const int j = b - g.n_fold;
    const int xc = j & 7, q8 = g.n_wgrad >> 3, r8 = g.n_wgrad & 7;
    const int idx = (xc < r8 ? xc * (q8 + 1) : r8 * (q8 + 1) + (xc - r8) * q8) + (j >> 3);
    const int split = idx / g.w_per_split;
    wgrad_tr_body<WT, WT, MODE>(w, g.w_tiles_n, g.w_tiles_k, g.w_mps, idx - split * g.w_per_split, split);
    return;
  }
  if (b < g.d_begin) return;          // padding up to a multiple of 8
  igemm_body<bf16_t, BM, BN, BKB, MODE>(d, g.d_tiles_m, g.d_tiles_n, g.d_ktps, g.d_hw_shift, b - g.d_begin,
                                        g.n_dgrad, 0);
}

template <int BM, int BN, int BKB, int MODE, int WT>
int launch_pair(const SdmiGemmArgs& d, const SdmiWgradArgs& w, const SdmiWgradArgs* f, int dgrad_cap,
                int hw_shift, hipStream_t st) {
  constexpr int BK = BKB / 2;
  constexpr int d_lds = 2 * (BM + BN) * (BKB + 16);
  constexpr int smem = d_lds > pair_wg_lds(WT) ? d_lds : pair_wg_lds(WT);
  auto kern = bwd_pair_kernel<BM, BN, BKB, MODE, WT>;
  SDMI_OPTIN_LDS(kern, smem, "bwd_pair");
  PairGeom g;
  SdmiWgradArgs fz = {};
  if (f) {
    fz = *f;
    const long long v4 = (long long)f->N * f->K / 4 * wgrad_fold_lanes(f->splits);
    long long nf = (v4 + 512 * 4 - 1) / (512 * 4);           // ~4 output vectors per thread
    g.n_fold = (int)(nf < 1 ? 1 : (nf > 128 ? 128 : nf));
  } else {
    g.n_fold = 0;
  }
  g.w_tiles_n = (w.N + WT - 1) / WT;
  g.w_tiles_k = (w.K + WT - 1) / WT;
  g.w_per_split = g.w_tiles_n * g.w_tiles_k + (w.dbias ? g.w_tiles_n : 0);
  int mps = (w.M + w.splits - 1) / w.splits;
  g.w_mps = (mps + 63) / 64 * 64;
  g.n_wgrad = g.w_per_split * w.splits;
  g.d_begin = (g.n_fold + g.n_wgrad + 7) & ~7;
  g.d_tiles_m = (d.M + BM - 1) / BM;
  g.d_tiles_n = (d.N + BN - 1) / BN;
  g.d_ktps = (d.K + BK - 1) / BK;
  g.d_hw_shift = hw_shift;
  const int tiles = g.d_tiles_m * g.d_tiles_n;
  constexpr int SLOTS = smem <= 49152 ? 768 : 512;       // resident workgroups (LDS: 3 or 2 per CU)
  int cap = dgrad_cap > 0 ? dgrad_cap : SLOTS - g.d_begin;
  cap = cap < 64 ? 64 : (cap > SLOTS ? SLOTS : cap);
  cap &= ~7;                             // a virtual block id keeps its XCD
  g.n_dgrad = tiles <= cap ? tiles : cap;
  SdmiGemmArgs q = d;
  q.split_k = 1;
  hipLaunchKernelGGL(kern, dim3(g.d_begin + g.n_dgrad), dim3(512), smem, st, q, w, fz, g);
  return sdmi_check_launch("bwd_pair");
}

}  // namespace

extern "C" int sdmi_bwd_pair(const SdmiBwdPairArgs* a, void* stream) {
  SDMI_REQUIRE(a && a->dgrad && a->wgrad, "null pointer");
  const SdmiGemmArgs& d = *(const SdmiGemmArgs*)a->dgrad;
  const SdmiWgradArgs& w = *(const SdmiWgradArgs*)a->wgrad;
  const SdmiWgradArgs* f = (const SdmiWgradArgs*)a->fold;
  SDMI_REQUIRE(d.a && d.w && d.out && w.a && w.dy && w.dw, "null operand");
  SDMI_REQUIRE(d.dtype == SDMI_BF16 && d.out_dtype == SDMI_BF16 && w.dtype == SDMI_BF16, "bf16 operands only");
  // ---- the data gradient: a 1x1 / linear problem or a plain stride-1 convolution over dY
  const bool d1x1 = d.KH == 1 && d.KW == 1 && d.stride == 1 && d.pad_t == 0 && d.pad_l == 0 && !d.ups && d.zins <= 1;
  const bool dplain = !d1x1 && !d.ups && d.zins <= 1 && d.stride == 1;
  SDMI_REQUIRE(d1x1 || dplain, "data gradient: 1x1 or plain stride-1 convolution");
  SDMI_REQUIRE(d.M > 0 && d.N > 64 && d.K > 0 && d.K == d.KH * d.KW * d.Cin && d.M == d.B * d.Ho * d.Wo, "bad dgrad geometry");
  SDMI_REQUIRE(d.Cin % 8 == 0 && d.lda % 8 == 0 && d.ldw % 8 == 0 && ((uintptr_t)d.a & 15) == 0 && ((uintptr_t)d.w & 15) == 0,
               "dgrad operands must keep 16-byte vectors");
  SDMI_REQUIRE(!d.a2 && !d.ln_colsum && !d.geglu && !d.softmax8 && !d.out2 && d.osy == 0 && !(d.batch > 1) && !d.rowvec &&
               !d.bias_m && d.split_k <= 1, "data gradient: plain epilogue (alpha, bias, residual) only");
  const long long a_bytes = ((long long)d.B * d.H * d.W + (long long)(d.KH + 1) * d.W) * d.lda * 2;
  SDMI_REQUIRE(a_bytes < (1ll << 31) && (long long)d.N * d.ldw * 2 < (1ll << 31), "dgrad operands beyond 31-bit offsets");
  // ---- the weight gradient: 128 x 128 output tiles, same loader class as the data gradient
  const bool w1x1 = w.KH == 1 && w.KW == 1 && w.stride == 1 && w.pad_t == 0 && w.pad_l == 0 && !w.ups && w.H == w.Ho &&
                    w.W == w.Wo;
  const bool wlin = !w1x1 && !w.ups && w.stride == 1 && w.H == w.Ho && w.W == w.Wo;
  SDMI_REQUIRE((d1x1 && w1x1) || (dplain && wlin), "both gradients must be 1x1, or both a stride-1 same-size convolution");
  SDMI_REQUIRE(w.N > 64 && w.K > 64 && w.K == w.KH * w.KW * w.Cin && w.M == w.B * w.Ho * w.Wo && w.Cin % 8 == 0 &&
               w.lda % 8 == 0 && w.ldy % 8 == 0, "bad wgrad geometry");
  SDMI_REQUIRE(w.splits >= 1 && (w.splits == 1 || w.workspace), "splits > 1 need a workspace");
  const long long mps = ((long long)w.M + w.splits - 1) / w.splits + 64;
  const long long ld = w.lda > w.ldy ? w.lda : w.ldy;
  SDMI_REQUIRE((mps + (long long)(w.KH + 1) * w.W + 64) * ld * 2 < (1ll << 31), "wgrad split beyond 31-bit offsets");
  if (f) SDMI_REQUIRE(f->dw && f->workspace && f->splits > 1 && f->N > 0 && f->K > 0 && ((long long)f->N * f->K) % 4 == 0 &&
                      f->dw != w.dw && (!f->dbias || f->dbias != w.dbias), "bad fold problem (or it targets this layer's dW)");
  int hw_shift = -1;
  {
    const int hw = d.Ho * d.Wo;
    if (hw > 0 && (hw & (hw - 1)) == 0) {
      hw_shift = 0;
      while ((1 << hw_shift) < hw) ++hw_shift;
    }
  }
  // tile shape of the data gradient: igemm's rule (128 x 128 when that gives >= 192 tiles, shallow-K 1x1
  // problems and everything smaller on 64 x 64); no split-K here -- the weight-gradient workgroups fill the chip
  const long long t128 = (long long)((d.M + 127) / 128) * ((d.N + 127) / 128);
  bool big = t128 >= 192;
  const int kbytes = d.K * 2;
  if (d1x1 && kbytes <= 512) big = false;
  const bool wide = kbytes >= 512;
  const int bk = (wide ? 128 : 64) / 2;
  if (!d1x1) SDMI_REQUIRE(d.KH * d.KW <= 32 && d.Cin % bk == 0, "convolution data gradient: K tiles must lie inside one filter tap");
  hipStream_t st = (hipStream_t)stream;
  SDMI_REQUIRE(a->wgrad_tile == 0 || a->wgrad_tile == 128 || (a->wgrad_tile == 64 && !big),
               "wgrad_tile: 128, or 64 next to 64 x 64 data-gradient tiles");
#define PAIR_GO(BM, BN, BKB, WT)                                                                       \
  return d1x1 ? launch_pair<BM, BN, BKB, 1, WT>(d, w, f, a->dgrad_cap, hw_shift, st)                    \
              : launch_pair<BM, BN, BKB, 2, WT>(d, w, f, a->dgrad_cap, hw_shift, st)
  if (big) {
    if (wide) PAIR_GO(128, 128, 128, 128);
    PAIR_GO(128, 128, 64, 128);
  }
  if (a->wgrad_tile == 64) {
    if (wide) PAIR_GO(64, 64, 128, 64);
    PAIR_GO(64, 64, 64, 64);
  }
  if (wide) PAIR_GO(64, 64, 128, 128);
  PAIR_GO(64, 64, 64, 128);
#undef PAIR_GO
}
