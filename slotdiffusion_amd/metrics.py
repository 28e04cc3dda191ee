"""Evaluation metrics with the pixel work on the GPU.

Host-side mirror of video_based/models/eval_utils.py (img_based twin identical): ARI / FG-ARI
(119-186), Hungarian mIoU (238-263, 293-320), mBO (266-290, 323-333), MSE / PSNR (75-92).  The
per-pixel part -- the (gt id, pred id) contingency table of every image, the per-image squared
error -- runs in HIP kernels (sdmi_contingency, sdmi_sqerr_rows); what is left is arithmetic on a
[B, C, K] table of exact integers, done with the reference's own float32 formulas on the host, and
scipy's linear_sum_assignment for the matching, exactly as the reference does.

Not covered: SSIM / LPIPS (skimage / lpips networks), bounding-box AP/AR (torchvision ops),
postproc_mask.
"""
import numpy as np
import torch

from . import _lib

__all__ = ['contingency', 'adjusted_rand_index', 'ARI_metric', 'fARI_metric', 'miou_metric',
           'fmiou_metric', 'mbo_metric', 'mse_metric', 'psnr_metric']


def _need_gpu(t):
    if not t.is_cuda:
        raise RuntimeError('slotdiffusion_amd.metrics runs on the GPU (libsdmi); got a CPU tensor')


def contingency(true_ids, pred_ids, num_true=None, num_pred=None):
    """Integer ids [B, ...] (any trailing shape, e.g. [B,H,W] or [B,T,H,W]) -> int32 table
    [B, C, K] on the device; C / K default to max id + 1 over the batch (the one_hot widths)."""
    _need_gpu(true_ids)
    _need_gpu(pred_ids)
    assert true_ids.shape == pred_ids.shape
    assert 'int' in str(true_ids.dtype) and 'int' in str(pred_ids.dtype)
    B = true_ids.shape[0]
    g = true_ids.reshape(B, -1).to(torch.int32).contiguous()
    q = pred_ids.reshape(B, -1).to(torch.int32).contiguous()
    C = int(num_true if num_true is not None else int(g.max()) + 1)
    K = int(num_pred if num_pred is not None else int(q.max()) + 1)
    counts = torch.zeros((B, C, K), dtype=torch.int32, device=g.device)
    _lib.call('sdmi_contingency', torch.cuda.current_stream().cuda_stream, gt=g.data_ptr(),
              pred=q.data_ptr(), counts=counts.data_ptr(), B=B, P=g.shape[1], Kg=C, Kp=K)
    return counts


def _ari_from_table(tab):
    """Adjusted Rand index per image from the float32 contingency table [B, C, K], with the float32
    operation order of the reference (eval_utils.py:157-176) so that the scores are identical:
    pairs(v) = sum v*(v-1) over cells / row sums / column sums; expected = rows*cols / total pairs;
    score = (cells - expected) / ((rows + cols)/2 - expected), 1 where that denominator is 0."""
    pairs = lambda v, dims: torch.sum(v * (v - 1), dim=dims)
    row_tot, col_tot = tab.sum(-1), tab.sum(-2)
    n_pix = row_tot.sum(1)
    same_cell = pairs(tab, [1, 2])
    same_row, same_col = pairs(row_tot, 1), pairs(col_tot, 1)
    chance = same_row * same_col / torch.clamp(n_pix * (n_pix - 1), min=1)
    gap = (same_row + same_col) / 2 - chance
    score = (same_cell - chance) / gap
    return torch.where(gap != 0, score, torch.ones_like(score))


def adjusted_rand_index(true_ids, pred_ids, ignore_background=False):
    """-> float32 [B] (CPU), eval_utils.py:119-176."""
    N = contingency(true_ids, pred_ids).cpu().float()
    if ignore_background:
        N = N[:, 1:]
    return _ari_from_table(N)


def ARI_metric(x, y):
    return adjusted_rand_index(x, y, ignore_background=False).mean().item()


def fARI_metric(x, y):
    return adjusted_rand_index(x, y, ignore_background=True).mean().item()


def _iou_tables(gt_mask, pred_mask):
    """Per-image (intersect [N_i, M_i] float32) with the per-image one_hot widths of the
    reference (N_i = max gt id of the image + 1, M_i likewise)."""
    T = contingency(gt_mask, pred_mask).cpu()
    out = []
    for b in range(T.shape[0]):
        t = T[b]
        rows = torch.nonzero(t.sum(1)).flatten()
        cols = torch.nonzero(t.sum(0)).flatten()
        n = int(rows.max()) + 1 if len(rows) else 1
        m = int(cols.max()) + 1 if len(cols) else 1
        out.append(t[:n, :m].float())
    return out


def _iou(inter, ignore_background):
    true_cnt = inter.sum(1)                      # pixels per gt id (one_hot column sums)
    pred_cnt = inter.sum(0)
    if ignore_background:
        inter, true_cnt = inter[1:], true_cnt[1:]
    union = true_cnt[:, None] + pred_cnt[None] - inter
    return (inter / (union + 1e-8)).numpy()


def _hungarian_miou(inter, ignore_background):
    from scipy.optimize import linear_sum_assignment
    if inter.shape[0] == 1 and ignore_background:          # GT holds only the background id
        return np.nan
    iou = _iou(inter, ignore_background)
    N, M = iou.shape
    row_ind, col_ind = linear_sum_assignment(iou, maximize=True)
    if M >= N:
        return iou[row_ind, col_ind].mean()
    return iou[row_ind, col_ind].sum() / float(N)


def miou_metric(gt_mask, pred_mask, ignore_background=False):
    ious = [_hungarian_miou(t, ignore_background) for t in _iou_tables(gt_mask, pred_mask)]
    if all(np.isnan(v) for v in ious):
        return np.nan
    return np.nanmean(ious)


def fmiou_metric(gt_mask, pred_mask):
    return miou_metric(gt_mask, pred_mask, ignore_background=True)


def mbo_metric(gt_mask, pred_mask):
    mbos = []
    for t in _iou_tables(gt_mask, pred_mask):
        if t.shape[0] == 1:
            mbos.append(np.nan)
        else:
            mbos.append(_iou(t, True).max(1).mean())
    return np.nanmean(mbos)


def _sqerr(x, y, mode=0):
    _need_gpu(x)
    _need_gpu(y)
    assert x.shape == y.shape
    B = x.shape[0]
    xf = x.reshape(B, -1).float().contiguous()
    yf = y.reshape(B, -1).float().contiguous()
    n = xf.shape[1]
    nchunk = max(1, min(64, n // 4096))
    part = torch.empty((B, nchunk), dtype=torch.float64, device=x.device)
    _lib.call('sdmi_sqerr_rows', torch.cuda.current_stream().cuda_stream, x=xf.data_ptr(),
              y=yf.data_ptr(), partial=part.data_ptr(), B=B, n=n, nchunk=nchunk, mode=mode)
    return part.cpu().sum(1).numpy(), n


def mse_metric(x, y):
    """x/y [B,3,H,W] in [0,1]: squared error summed over an image, mean over the batch."""
    se, _ = _sqerr(x, y)
    return float(se.mean())


def psnr_metric(x, y):
    """skimage.metrics.peak_signal_noise_ratio(data_range=1) per image, mean over the batch:
    10 * log10(1 / mean squared error)."""
    se, n = _sqerr(x, y)
    return float(np.mean(10.0 * np.log10(1.0 / (se / n))))


def ssim_metric(x, y):
    """eval_utils.ssim_metric (eval_utils.py:91-106): x / y [B,3,H,W] in [0,1] -> mean over images of
    skimage's structural_similarity(x*255, y*255, channel_axis=0, gaussian_weights=True, sigma=1.5,
    use_sample_covariance=False, data_range=255).  skimage is absent offline: the published algorithm
    is restated (sdmi_ssim); parity with skimage itself is unpinned."""
    x, y = torch.as_tensor(x), torch.as_tensor(y)
    _need_gpu(x)
    _need_gpu(y)
    assert x.shape == y.shape and x.dim() == 4
    B, Cc, H, W = x.shape
    xf = (x.float() * 255.0).contiguous()
    yf = (y.float() * 255.0).contiguous()
    out = torch.empty((B * Cc,), dtype=torch.float64, device=x.device)
    _lib.call('sdmi_ssim', torch.cuda.current_stream().cuda_stream, x=xf.data_ptr(), y=yf.data_ptr(),
              out=out.data_ptr(), P=B * Cc, H=H, W=W, data_range=255.0)
    return float(out.view(B, Cc).mean(1).mean().cpu())        # mean over channels, then over images


def shuffle_slots(slots):
    """Compositional generation (video_based/test_comp_gen.py:25-31): slot i of every item is taken
    from the item i places further along the batch, so each new scene mixes objects of N scenes.
    slots [B,T,N,C] or [B,N,C] -> a new tensor of the same shape (the input is left untouched)."""
    nd = slots.dim()
    assert nd in (3, 4)
    B, N = slots.shape[0], slots.shape[-2]
    shift = torch.arange(N, device=slots.device)
    shift = torch.where(shift < B, shift, torch.zeros_like(shift))     # slots[i:] is empty for i >= B
    src = (torch.arange(B, device=slots.device)[:, None] + shift[None]) % B
    slot_ix = torch.arange(N, device=slots.device)[None].expand(B, N)
    if nd == 3:
        return slots[src, slot_ix]
    return slots[src, :, slot_ix].permute(0, 2, 1, 3).contiguous()
