"""Kernel providers used by engine.py.

`Kern`      -- inference: direct libsdmi launches (no autograd bookkeeping, fused epilogues).
`KernGrad`  -- training: the same operations as `torch.autograd.Function`s whose backward passes are
               libsdmi kernels too (dgrad = the implicit-GEMM kernel on a flipped operand, wgrad,
               GroupNorm/LayerNorm/attention backward ...).  Parameter gradients are written by
               the kernels straight into the model's flat fp32 gradient arena (`p.grad` views),
               so autograd only routes activation gradients.

Both resolve weights through a WeightBank: fp32 operands are views of the master arena, bf16
operands views of the bf16 shadow arena (same offsets); only channel-padded and non-adjacent
fused operands are materialised.
"""
import contextlib
import os

import torch

from . import _lib, ops, policy
from ._lib import call

_DT = {torch.float32: _lib.F32, torch.bfloat16: _lib.BF16}


def _p(t):
    return 0 if t is None else t.data_ptr()


def _st():
    return torch.cuda.current_stream().cuda_stream


_DEBUG = False            # (set by hand: prints the norm of every gradient the backward kernels produce)


def _dbg(tag, **tensors):
    if _DEBUG:
        msg = ' '.join(f'{k}={float(v.float().norm()):.4e}' if v is not None else f'{k}=None'
                       for k, v in tensors.items())
        print(f'[bwd] {tag}: {msg}', flush=True)


FP8_ACT_SCALE = 8.0        # activations -> e4m3fn: +-56 representable, 2^-9 resolution near zero
_LAZY_CAT = bool(policy.flag('LAZY_CAT'))
_GEGLU_FUSE = bool(policy.flag('GEGLU_FUSE'))     # training: GEGLU in the FF GEMM's epilogue


class CatPair:
    """Two NHWC tensors standing for their channel concatenation (inference: read in place)."""

    def __init__(self, a, b):
        self.a, self.b = a, b
        self.shape = tuple(a.shape[:-1]) + (a.shape[-1] + b.shape[-1],)
        self.dtype, self.device = a.dtype, a.device

    def dim(self):
        return self.a.dim()

    def materialize(self):
        return ops.concat_channels(self.a, self.b)


_RES_MERGE = bool(policy.flag('RES_MERGE'))
_FF_MERGE = bool(policy.flag('FF_MERGE'))
_CROSS_FOLD = bool(policy.flag('CROSS_FOLD'))
_CROSS_MAX_SLOTS = 16      # most slots the folded cross-attention path takes (16-row groups per head from 9 slots)
_LN_FOLD = bool(policy.flag('LN_FOLD'))
_ST_FUSED = bool(policy.flag('ST_FUSED'))     # fused SpatialTransformer block (sdmi_st_block)
_ST_FF_SPLIT = bool(policy.flag('ST_FF_SPLIT'))   # ... its feed-forward over workgroup pairs where the grid is half the chip
_WGRAD_HALO = bool(policy.flag('WGRAD_HALO'))    # direct 3x3 weight gradient on channel pairs (wgrad3x3_halo_kernel)
_CROSS_ONE = 64            # folded slot cross-attention as ONE launch (sdmi_cross_fold) up to this many tokens per image
_UPS_PARITY = bool(policy.flag('UPS_PARITY'))   # upsample convolutions as four 2x2 parity convolutions in one launch
# the fused inference block only when its grid (one workgroup per 64 / 32 token rows) fills a good part of the chip: at
# B = 64 the 8^2 level gives 64 workgroups that each stream the block's 4 MB of weights -- 119 us against 108 us for the
# per-layer launches (tests set these module variables)
_ST_MIN_WGS = 128
# token rows per workgroup: 0 = 64 when that gives >= 192 workgroups, else 32; 64 / 32 force one
_ST_ROWS = 0


def _copy_group(items):
    """[(src ptr, dst ptr, bytes)] -> sdmi_copy_group launches of up to 32 copies each."""
    import ctypes
    Item = _lib.CSTRUCT['SdmiCopyItem']
    for c0 in range(0, len(items), 32):
        chunk = items[c0:c0 + 32]
        arr = (Item * len(chunk))()
        for a, (sp, dp, nb) in zip(arr, chunk):
            a.src, a.dst, a.bytes = sp, dp, nb
        call('sdmi_copy_group', _st(), items=ctypes.addressof(arr), n=len(chunk))



# ------------------------------------------------------------------------------------------
# weight streams of the fused SpatialTransformer block (csrc/st_fused.hip): per-wave sequences of 2 KB
# units = the XOR-swizzled LDS image of 16 weight rows x 64 k, in the order the kernel multiplies them
# ------------------------------------------------------------------------------------------
def _st_unit_index(entries):
    """entries[w] = [(element offset of the matrix, row pitch, first row, first k), ...] per unit, 8 waves
    -> int64 [8 * U * 1024] gather index into the flat source (unit byte b of wave w, unit u at
    ((w * U + u) * 1024 + b / 2)): physical 16-byte chunk p of row r holds logical chunk p ^ ((r >> 1) & 7)."""
    import numpy as np
    r = np.arange(16, dtype=np.int64)[:, None, None]
    p_ = np.arange(8, dtype=np.int64)[None, :, None]
    e = np.arange(8, dtype=np.int64)[None, None, :]
    col = (p_ ^ ((r >> 1) & 7)) * 8 + e                       # [16, 8, 8]
    out = []
    for ent in entries:
        for off, ld, row0, k0 in ent:
            out.append((off + (row0 + r) * ld + k0 + col).reshape(-1))
    return torch.from_numpy(np.concatenate(out))


def st_geometry(C):
    NSL, KT = C // 128, C // 64
    return dict(NSL=NSL, KT=KT, NHC=C // 32, R=C // 32 * 8, UA=4 * KT * NSL, UB1=KT * NSL,
                UIMG=KT + 2 * NSL, UB2=KT * NSL + (C // 32) * (2 * KT + 2 * NSL))


def st_index_a(C):
    """Phase A source = [W_in (C x C) | W_q' | W_k' | W_v' (C x C each)] flat."""
    g = st_geometry(C)
    ent = []
    for w in range(8):
        e = []
        for m in range(4):                                  # proj_in, q, k, v
            for kt in range(g['KT']):
                for s_ in range(g['NSL']):
                    e.append((m * C * C, C, (w * g['NSL'] + s_) * 16, kt * 64))
        ent.append(e)
    return _st_unit_index(ent)


def st_index_b(C):
    """Phase B shared source = [W_o (C x C) | W_po (C x C) | W_1' (8C x C) | W_m = W_po W_ff (C x 4C)] flat."""
    g = st_geometry(C)
    NSL, KT = g['NSL'], g['KT']
    o_po, o_1, o_m = C * C, 2 * C * C, 2 * C * C + 8 * C * C
    ent = []
    for w in range(8):
        e = []
        for kt in range(KT):                                # attn1.to_out
            for s_ in range(NSL):
                e.append((0, C, (w * NSL + s_) * 16, kt * 64))
        for kt in range(KT):                                # x2 @ W_po^T
            for s_ in range(NSL):
                e.append((o_po, C, (w * NSL + s_) * 16, kt * 64))
        for hc in range(g['NHC']):
            for kt in range(KT):                            # value | gate rows of hidden chunk hc
                e.append((o_1, C, hc * 128 + w * 16, kt * 64))
                e.append((o_1, C, 4 * C + hc * 128 + w * 16, kt * 64))
            for k2 in range(2):                             # the chunk's 128 k of the merged output weight
                for s_ in range(NSL):
                    e.append((o_m, 4 * C, (w * NSL + s_) * 16, hc * 128 + k2 * 64))
        ent.append(e)
    return _st_unit_index(ent)


def st_index_img(C):
    """Per-image source = [Wq[b] padded to 128 rows (128 x C) | W2[b] with K padded to 128 (C x 128)] flat."""
    g = st_geometry(C)
    ent = []
    for w in range(8):
        e = [(0, C, w * 16, kt * 64) for kt in range(g['KT'])]
        for k2 in range(2):
            for s_ in range(g['NSL']):
                e.append((128 * C, 128, (w * g['NSL'] + s_) * 16, k2 * 64))
        ent.append(e)
    return _st_unit_index(ent)


def st_train_units(C):
    """Unit order of the training kernels' weight streams (csrc/st_train.hip), per wave: lists of
    (matrix key, first row, first k, transposed).  Forward: a = [proj_in | q | k | v], b = [to_out | q2 | to_out2 | per
    hidden chunk: ff1 value rows, ff1 gate rows, ff2 k-chunk | proj_out].  Backward (rows / k of the TRANSPOSED
    matrices): b1 = [proj_out^T | per hidden chunk: ff2^T rows, ff1^T value k-chunk, gate k-chunk | to_out2^T],
    b2 = [q2^T | to_out^T], ba = [q^T | k^T | v^T | proj_in^T]."""
    NSL, KT, NHC = C // 128, C // 64, C // 32
    out = dict(a=[], b=[], b1=[], b2=[], ba=[])
    for w in range(8):
        full = lambda m, tr: [(m, (w * NSL + s_) * 16, kt * 64, tr) for kt in range(KT) for s_ in range(NSL)]
        a = [u for m in ('in', 'q', 'k', 'v') for u in full(m, False)]
        b = [u for m in ('o', 'q2', 'o2') for u in full(m, False)]
        b1 = full('po', True)
        for hc in range(NHC):
            for kt in range(KT):
                b.append(('ff1', hc * 128 + w * 16, kt * 64, False))
                b.append(('ff1', 4 * C + hc * 128 + w * 16, kt * 64, False))
            for k2 in range(2):
                for s_ in range(NSL):
                    b.append(('ff2', (w * NSL + s_) * 16, hc * 128 + k2 * 64, False))
            b1 += [('ff2', hc * 128 + w * 16, kt * 64, True) for kt in range(KT)]
            for part in range(2):
                b1 += [('ff1', (w * NSL + s_) * 16, part * 4 * C + hc * 128 + k2 * 64, True)
                       for k2 in range(2) for s_ in range(NSL)]
        b += full('po', False)
        b1 += full('o2', True)
        out['a'].append(a)
        out['b'].append(b)
        out['b1'].append(b1)
        out['b2'].append(full('q2', True) + full('o', True))
        out['ba'].append([u for m in ('q', 'k', 'v', 'in') for u in full(m, True)])
    return out


def ups_parity_split(w):
    """Nearest-2x upsampling followed by a 3x3 convolution (unet.py:108-121) reads only 2 x 2 DISTINCT input pixels
    per output pixel: output parity (py, px) is a 2 x 2 convolution of the low-resolution input whose taps are sums
    of the 3 x 3 taps that fall on the same input pixel -- rows: parity 0 -> {w0 | w1 + w2} at offsets (-1, 0)
    (pad top 1), parity 1 -> {w0 + w1 | w2} at offsets (0, +1) (pad bottom 1); columns alike.  16 instead of 36
    multiply-adds per output pixel and input channel.  w [Cout, Cin, 3, 3] fp32 -> {(py, px): [Cout, 2*2*Cin] fp32}
    (tap-major K, the layout sdmi_igemm takes)."""
    co, ci = w.shape[0], w.shape[1]
    taps = {0: ((0,), (1, 2)), 1: ((0, 1), (2,))}
    out = {}
    for py in (0, 1):
        for px in (0, 1):
            we = torch.zeros((co, 2, 2, ci), dtype=torch.float32, device=w.device)
            for r in (0, 1):
                for c in (0, 1):
                    for ky in taps[py][r]:
                        for kx in taps[px][c]:
                            we[:, r, c, :] += w[:, :, ky, kx]
            out[py, px] = we.reshape(co, 4 * ci)
    return out


class WeightBank:
    """name(s) -> GEMM operand [N, K] in the requested dtype (K padded to the vector width)."""

    def __init__(self, model, dtype):
        self.model = model
        self.t = model.tensors()
        self.dtype = dtype
        self.cache = {}
        # autograd anchor: parameters reach the kernels by name, so parameterised Functions take
        # this requires-grad dummy to make their outputs part of the graph
        self.anchor = torch.zeros(1, device=model.arena().device, requires_grad=True)
        # a second anchor for the denoiser's Functions: `autograd.grad(loss, [slots, anchor_dec])`
        # then runs exactly the denoiser's backward (and stops at the slots), so that its gradient
        # range can be all-reduced while the slot encoder's backward is still running
        self.anchor_dec = torch.zeros(1, device=model.arena().device, requires_grad=True)
        # weight-gradient GEMMs are off the backward critical path (nothing downstream reads them
        # before the optimiser): they run on a second HIP stream, concurrently with the dgrad
        # chain, and are joined when the autograd pass ends.  Their operands are kept alive until
        # the join (HBM is plentiful), so no allocator stream bookkeeping is needed.
        self.overlap_wgrad = bool(policy.flag('WGRAD_STREAM'))
        self.n_side = 4
        self._sides = []
        self._pending, self._side_of = [], {}
        self._join_queued = False
        # grouped weight gradients (sdmi_wgrad_group): the fused SpatialTransformer backward queues a block's problems
        # and launches them together (StBlockFn.backward_fused)
        self._wq, self._wq_keep = [], []
        self.defer_colsum = bool(policy.flag('DEFER_COLSUM'))
        # data gradient + weight gradient of a layer in ONE launch on the main stream (sdmi_bwd_pair):
        # no side-stream fork / join per layer, the M-split partials of a layer are folded by extra
        # workgroups of the NEXT layer's launch (`_pfold`: the pending fold), the last one at the join
        self.pair_bwd = bool(policy.flag('BWD_PAIR'))
        # geometry of a pair launch (round-3 sweeps, profiles/r03_pair_sweep.txt): resident workgroups (2 per CU), of
        # which walk dX tiles, 64-row steps per weight-gradient workgroup, 64 x 64 dW tiles below this many of them
        self.pair_slots, self.pair_dgrad, self.pair_min_steps, self.pair_wt64 = 512, 256, 8, 96
        self._pfold = None
        # gradient destinations a side-stream launch of this backward wrote: a main-stream writer of the same
        # destination (pair launch, pending fold) waits for that stream first
        self._side_dst = {}
        # dgrad operands: key -> [buffer, epoch, (stable source view | None, geometry)]
        self._wd, self._wd_epoch, self._wd_stale, self._wd_table = {}, 0, False, None
        # e4m3fn operands with device-side scales (training in the fp8 configuration): name -> [bytes, inv view, epoch,
        # source view]; one slot of _w8_amax / _w8_inv per operand, in registration order
        self._w8, self._w8_epoch, self._w8_stale, self._w8_table = {}, 0, False, None
        self._w8_amax = self._w8_inv = None
        # unit streams of the fused SpatialTransformer training kernels: block -> dict(wa, wb, descs, epoch)
        self._stp, self._stp_epoch, self._stp_stale, self._stp_table = {}, 0, False, None

    def anchor_for(self, names):
        n = names if isinstance(names, str) else names[0]
        return self.anchor_dec if n.startswith('dm_decoder') else self.anchor

    def side_index(self, key):
        """Side stream of a parameter (by name): a parameter used several times per step (Slot
        Attention iterations, the per-frame predictor) accumulates its gradient in stream order on
        ONE stream; different parameters are spread round-robin."""
        i = self._side_of.get(key)
        if i is None:
            i = self._side_of[key] = len(self._side_of) % self.n_side
        return i

    def side_stream(self, key=None):
        if not self.overlap_wgrad:
            return None
        # a few streams: most weight-gradient launches are small (<= 128 workgroups + their
        # reduction) and several of them fit the chip next to the dgrad chain
        if not self._sides:
            self._sides = [torch.cuda.Stream() for _ in range(self.n_side)]
        return self._sides[self.side_index(key)]

    def ensure_join(self):
        if not self._join_queued:
            self._join_queued = True
            torch.autograd.Variable._execution_engine.queue_callback(self.join)

    def defer(self, *tensors):
        self._pending.append(tensors)
        self.ensure_join()

    # ---- grouped weight gradients (sdmi_wgrad_group) ----------------------------------------------
    WQ_TARGET = 512.0          # workgroups of a grouped launch, shared by the problems in proportion to their work

    def flush_wgrad(self):
        """Launch the queued bf16 1x1 / linear weight-gradient problems (`_wq`: (sdmi_wgrad fields, 128 x 128 tiles, 64-row
        steps)) as ONE sdmi_wgrad_group call on a side stream; M-splits chosen here.  -> the stream it ran on."""
        if not self._wq:
            return None
        q, keep = self._wq, self._wq_keep
        self._wq, self._wq_keep = [], []
        # M-splits: ~512 workgroups over the group, shared in proportion to each problem's work
        # (tiles x 64-row steps), every split keeping >= 8 steps
        work = [t * st for _, t, st in q]
        total = float(sum(work))
        splits = []
        for (kw, t, st), w in zip(q, work):
            want = max(1, int(round(self.WQ_TARGET * w / total / t)))
            splits.append(max(1, min(want, st // 8, 64)))
        ws_len = sum(sp * (kw['N'] * kw['K'] + kw['N']) for (kw, _, _), sp in zip(q, splits) if sp > 1)
        dev = keep[0].device
        ws = torch.empty((max(ws_len, 1),), dtype=torch.float32, device=dev)
        Arr = _lib.CSTRUCT['SdmiWgradArgs'] * len(q)
        arr = Arr()
        off = 0
        flops = 0.0
        for a, (kw, _, _), sp in zip(arr, q, splits):
            for k, v in kw.items():
                setattr(a, k, v)
            a.splits = sp
            if sp > 1:
                a.workspace = ws.data_ptr() + 4 * off
                off += sp * (kw['N'] * kw['K'] + kw['N'])
            flops += 2.0 * kw['M'] * kw['N'] * kw['K']
        self._wq_flushes = getattr(self, '_wq_flushes', 0) + 1
        side = self.side_stream(f'wgrad-group-{self._wq_flushes % self.n_side}')
        if side is not None:
            ev = torch.cuda.Event()
            ev.record()
            side.wait_event(ev)
        with (torch.cuda.stream(side) if side is not None else contextlib.nullcontext()):
            import ctypes
            call('sdmi_wgrad_group', _st(), problems=ctypes.addressof(arr), n=len(q),
                 _meta=dict(flops=flops))
        self._pending.append(tuple(keep) + (ws,))
        return side

    # ---- deferred dgamma / dbeta folds of the normalisation backward passes ---------------------
    def queue_colsum(self, partial, nblk, C, out0, out1):
        self._cq = getattr(self, '_cq', [])
        self._cq.append((partial, nblk, C, out0, out1))
        self.ensure_join()

    def flush_colsum(self):
        q = getattr(self, '_cq', [])
        if not q:
            return
        self._cq = []
        import ctypes
        Item = _lib.CSTRUCT['SdmiColsumItem']
        # a parameter used several times per step (Slot Attention iterations, per-frame modules)
        # has several items with the same destination: they go to successive launches (the
        # workgroups of one launch read-modify-write their destinations concurrently)
        chunks = []
        for it in q:
            for ch in chunks:
                if len(ch[0]) < 64 and _p(it[3]) not in ch[1]:
                    break
            else:
                ch = ([], set())
                chunks.append(ch)
            ch[0].append(it)
            ch[1].add(_p(it[3]))
        for chunk, _ in chunks:
            arr = (Item * len(chunk))()
            for a, (partial, nblk, C, o0, o1) in zip(arr, chunk):
                a.partial, a.out0, a.out1, a.nblk, a.C = _p(partial), _p(o0), _p(o1), nblk, C
            call('sdmi_colsum_group', _st(), items=ctypes.addressof(arr), n=len(chunk))
        # (the partial buffers die with q: the launches above are ordered before their reuse)

    # ---- fused data + weight gradient launches --------------------------------------------------
    def flush_pending_fold(self, dst_ptrs=None):
        """Fold the pending M-split partials now (own launch).  dst_ptrs given: only when the pending
        fold targets one of these gradient buffers (a parameter used again before its fold ran)."""
        pf = self._pfold
        if pf is None:
            return
        if dst_ptrs is not None and pf[0]['dw'] not in dst_ptrs and (not pf[0]['dbias'] or pf[0]['dbias'] not in dst_ptrs):
            return
        self._pfold = None
        self._after_side_writers(pf[0]['dw'], pf[0]['dbias'])
        import ctypes
        arr = (_lib.CSTRUCT['SdmiWgradArgs'] * 1)()
        for k, v in pf[0].items():
            setattr(arr[0], k, v)
        call('sdmi_wgrad_fold_group', _st(), problems=ctypes.addressof(arr), n=1)

    def _after_side_writers(self, *ptrs):
        """Main stream waits for side streams that have launches into any of these destinations in flight."""
        for p_ in ptrs:
            side = self._side_dst.pop(p_, None) if p_ else None
            if side is not None:
                torch.cuda.current_stream().wait_stream(side)

    def pair_launch(self, dkw, wkw, keep):
        """One sdmi_bwd_pair launch: dkw / wkw are the sdmi_igemm / sdmi_wgrad fields of the layer's data
        and weight gradient; M-split chosen here (the weight-gradient workgroups take what the data
        gradient leaves of `pair_slots` resident workgroups, >= pair_min_steps 64-row steps each)."""
        import ctypes
        M, N, K = wkw['M'], wkw['N'], wkw['K']
        t128 = ((dkw['M'] + 127) // 128) * ((dkw['N'] + 127) // 128)
        shallow = dkw['KH'] == 1 and dkw['KW'] == 1 and dkw['K'] * 2 <= 512
        big = t128 >= 192 and not shallow
        tiles_d = t128 if big else ((dkw['M'] + 63) // 64) * ((dkw['N'] + 63) // 64)
        steps = (M + 63) // 64

        def plan(wt, slots, min_steps):
            tn = (N + wt - 1) // wt
            tiles_w = tn * ((K + wt - 1) // wt) + (tn if wkw['dbias'] else 0)
            n_d = min(tiles_d, self.pair_dgrad)
            return tiles_w, n_d, max(1, min((slots - n_d) // tiles_w, steps // min_steps, 256))
        wt = 128
        tiles_w, n_d, splits = plan(128, self.pair_slots, self.pair_min_steps)
        if not big and self.pair_wt64 and tiles_w * splits < self.pair_wt64:
            # a small layer: 64 x 64 weight-gradient tiles -- four times the workgroups, three per CU
            wt = 64
            tiles_w, n_d, splits = plan(64, self.pair_slots * 3 // 2, max(2, self.pair_min_steps // 2))
        self.flush_pending_fold((wkw['dw'], wkw['dbias']))      # this parameter again: fold first
        self._after_side_writers(wkw['dw'], wkw['dbias'])
        ws = None
        if splits > 1:
            ws = torch.empty((splits * (N * K + N),), dtype=torch.float32, device=keep[0].device)
        S = _lib.CSTRUCT
        d, w, f = S['SdmiGemmArgs'](), S['SdmiWgradArgs'](), S['SdmiWgradArgs']()
        for k, v in dkw.items():
            setattr(d, k, v)
        for k, v in wkw.items():
            setattr(w, k, v)
        w.splits, w.workspace, w.defer_fold = splits, _p(ws), 1
        pf, self._pfold = self._pfold, None
        if pf is not None:
            self._after_side_writers(pf[0]['dw'], pf[0]['dbias'])
            for k, v in pf[0].items():
                setattr(f, k, v)
        call('sdmi_bwd_pair', _st(), dgrad=ctypes.addressof(d), wgrad=ctypes.addressof(w),
             fold=(ctypes.addressof(f) if pf is not None else 0), dgrad_cap=((n_d + 7) // 8 * 8), wgrad_tile=wt,
             _meta=dict(flops=2.0 * dkw['M'] * dkw['N'] * dkw['K'] + 2.0 * M * N * K,
                        flops_dgrad=2.0 * dkw['M'] * dkw['N'] * dkw['K'],
                        bytes=float(2 * (dkw['M'] * dkw['K'] // (dkw['KH'] * dkw['KW']) + dkw['N'] * dkw['K']
                                         + dkw['M'] * dkw['N'] * (2 if dkw.get('residual', 0) else 1)
                                         + M * wkw['Cin']) + 4 * N * K)))
        if splits > 1:            # folded by the next pair launch (or at the join)
            self._pfold = (dict(dw=wkw['dw'], dbias=wkw['dbias'], workspace=_p(ws), N=N, K=K, splits=splits,
                                accumulate=wkw['accumulate']), ws)
            self.ensure_join()

    def join(self):
        self.flush_pending_fold()
        self.flush_wgrad()
        self.flush_colsum()
        for side in self._sides:
            torch.cuda.current_stream().wait_stream(side)
        self._pending.clear()
        self._side_dst.clear()
        self._join_queued = False

    def invalidate(self):
        """The master weights changed (optimiser step / checkpoint load): derived operands of
        TRAINABLE parameters are dropped; those of frozen ones (VQ-VAE, DINO ViT) stay valid."""
        fz = self._frozen_names()
        keep = {}
        for key, val in self.cache.items():
            names = key[0] if (isinstance(key, tuple) and isinstance(key[0], tuple)) else \
                (key[1] if (isinstance(key, tuple) and key[0] == 'bias') else None)
            if names is not None and all(n in fz for n in names):
                keep[key] = val
        self.cache = keep
        self._wd_stale = True
        self._w8_stale = True
        self._stp_stale = True

    def _frozen_names(self):
        fz = getattr(self, '_fz', None)
        if fz is None:
            fz = self._fz = {p.name for p in getattr(self.model, '_spec', []) if not p.trainable}
        return fz

    def f(self, name):
        return self.t[name]

    def _flat(self, name, dtype):
        """Operand view straight out of the (master | shadow) arena; None if it needs padding."""
        p = self.t[name]
        vec = ops.vec_of(dtype)
        if p.dim() == 4:
            co, ci, kh, kw = p.shape
            if ci % vec:
                return None
            shape = (co, kh * kw * ci)
        else:
            if p.shape[-1] % vec:
                return None
            shape = (p.shape[0], p.shape[1]) if p.dim() == 2 else (1, p.numel())
        if dtype == torch.float32:
            src = self.model.arena_slice(name)
        else:
            src = self.model.arena_slice(name, self.model.shadow_arena())
        return src.view(shape)

    def w(self, names, dtype=None):
        dtype = dtype or self.dtype
        if isinstance(names, str):
            names = (names,)
        key = (names, dtype)
        if key in self.cache:
            return self.cache[key]
        parts = [self._flat(n, dtype) for n in names]
        out = None
        if all(p is not None for p in parts):
            adjacent = all(parts[i].data_ptr() + parts[i].numel() * parts[i].element_size() ==
                           parts[i + 1].data_ptr() for i in range(len(parts) - 1))
            if len(parts) == 1:
                out = parts[0]
            elif adjacent:                                 # fused operand = one arena slice
                k = parts[0].shape[1]
                n = sum(p.shape[0] for p in parts)
                base = self.model.arena() if dtype == torch.float32 else self.model.shadow_arena()
                o = self.model._offsets[names[0]][0]
                out = base[o:o + n * k].view(n, k)
        if out is None and all(p is not None for p in parts) and len({p.shape[1] for p in parts}) == 1:
            # operand views exist but are not adjacent (the 22 time-embedding projections): gather them
            # with ONE grouped copy instead of a cast + a copy per part
            out = torch.empty((sum(p.shape[0] for p in parts), parts[0].shape[1]), dtype=dtype,
                              device=parts[0].device)
            o, items = 0, []
            for p in parts:
                items.append((_p(p), _p(out[o:]), p.numel() * p.element_size()))
                o += p.shape[0]
            _copy_group(items)
        if out is None:
            mats = []
            vec = ops.vec_of(dtype)
            for nme in names:
                p = self.t[nme]
                if p.dim() == 4:
                    co, ci, kh, kw = p.shape
                    flat = self.model.arena_slice(nme).view(co * kh * kw, ci)
                    cpad = (ci + vec - 1) // vec * vec
                    mats.append(ops.cast2d(flat, dtype, cols=ci, ldd=cpad).view(co, kh * kw * cpad))
                else:
                    k = p.shape[1]
                    kpad = (k + vec - 1) // vec * vec
                    mats.append(ops.cast2d(self.model.arena_slice(nme).view(p.shape[0], k), dtype,
                                           cols=k, ldd=kpad))
            if len(mats) == 1:
                out = mats[0]
            else:
                out = torch.empty((sum(m.shape[0] for m in mats), mats[0].shape[1]), dtype=dtype,
                                  device=mats[0].device)
                o = 0
                for m in mats:
                    ops.cast2d(m, dtype, out=out[o:o + m.shape[0]])
                    o += m.shape[0]
        self.cache[key] = out
        return out

    def w8(self, name):
        """e4m3fn operand of a conv / linear weight with its per-tensor scale: (bytes [N][K],
        1 / scale), scale = 448 / amax.  Weight preparation (inference: once per weight update); the
        amax read-back synchronises, so the first use must precede any graph capture (the samplers'
        warm-up pass does)."""
        key = ('fp8', name)
        if key not in self.cache:
            p = self.t[name]
            flat = self._flat(name, torch.float32)
            if flat is None:
                raise _lib.SdmiError(f'{name}: fp8 operands need 16-byte input-channel rows')
            amax = float(p.detach().abs().max())
            scale = 448.0 / max(amax, 1e-12)
            self.cache[key] = (ops.quant_fp8(flat.float().contiguous(), scale), 1.0 / scale)
        return self.cache[key]

    W8_SLOTS = 1024

    def w8_dev(self, name):
        """e4m3fn operand of a TRAINABLE conv / linear weight with a scale the device derives from the tensor's own
        amax: (bytes [N][K], inv = one-element fp32 view holding amax / 448 -- SdmiGemmArgs.alpha_dev).  The operands
        live in persistent buffers; after an optimiser step (invalidate()) the first request re-quantises ALL of
        them from the fp32 master arena with one sdmi_fp8_quant_group call (memset + amax launch + quantise launch,
        descriptor table on the device, no read-back) -- part of the captured train step, so every replay
        quantises the weights it multiplies with."""
        if self._w8_stale:
            self._requant_all()
        ent = self._w8.get(name)
        if ent is not None and ent[2] == self._w8_epoch:
            return ent[0], ent[1]
        flat = self._flat(name, torch.float32)
        if flat is None or flat.numel() % 16:
            raise _lib.SdmiError(f'{name}: fp8 operands need 16-byte input-channel rows')
        if self._w8_amax is None:
            self._w8_amax = torch.zeros(self.W8_SLOTS, dtype=torch.int32, device=flat.device)
            self._w8_inv = torch.zeros(self.W8_SLOTS, dtype=torch.float32, device=flat.device)
        if ent is None:
            idx = len(self._w8)
            if idx >= self.W8_SLOTS:
                raise _lib.SdmiError('fp8 operand table full')
            ent = [torch.empty(flat.shape, dtype=torch.uint8, device=flat.device), self._w8_inv[idx:idx + 1],
                   self._w8_epoch, flat, idx]
            self._w8[name] = ent
            self._w8_table = None             # new member: rebuild the descriptor table at the next re-quantisation
        # (first use: a one-entry table -- eager only, it copies the table to the device)
        if torch.cuda.is_current_stream_capturing():
            raise _lib.SdmiError(f'{name}: first fp8 use of a weight inside a graph capture (its descriptor table is '
                                 'a host-to-device copy): run the warm-up passes before capturing')
        Desc = _lib.CSTRUCT['SdmiFp8Desc']
        arr = (Desc * 1)()
        blocks = self._fill_desc8(arr[0], ent, 0)
        dev = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(flat.device)
        idx = ent[4]
        call('sdmi_fp8_quant_group', _st(), descs=_p(dev), n_desc=1, src_dtype=_lib.F32, total_blocks=blocks,
             amax_bits=_p(self._w8_amax[idx:]), inv_scale=_p(self._w8_inv[idx:]))
        ent[2] = self._w8_epoch
        return ent[0], ent[1]

    @staticmethod
    def _fill_desc8(d, e, blk):
        d.src, d.dst, d.n, d.block_begin = e[3].data_ptr(), e[0].data_ptr(), e[3].numel(), blk
        return blk + (e[3].numel() + 4095) // 4096

    def _requant_all(self):
        self._w8_stale = False
        if not self._w8:
            return
        self._w8_epoch += 1
        items = sorted(self._w8.values(), key=lambda e: e[4])         # slot order = descriptor order
        if self._w8_table is None:
            if torch.cuda.is_current_stream_capturing():
                raise _lib.SdmiError('fp8 descriptor table rebuilt inside a graph capture (a new fp8 operand was '
                                     'registered after the warm-up passes)')
            Desc = _lib.CSTRUCT['SdmiFp8Desc']
            arr = (Desc * len(items))()
            blk = 0
            for d, e in zip(arr, items):
                blk = self._fill_desc8(d, e, blk)
            dev = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(items[0][0].device)
            self._w8_table = (dev, len(items), blk)
        tab = self._w8_table
        call('sdmi_fp8_quant_group', _st(), descs=_p(tab[0]), n_desc=tab[1], src_dtype=_lib.F32, total_blocks=tab[2],
             amax_bits=_p(self._w8_amax), inv_scale=_p(self._w8_inv))
        for e in items:
            e[2] = self._w8_epoch

    def fp8_train_ok(self, x, wnames, geom):
        """Training forward in the fp8 configuration (models.set_compute_dtype('fp8')): the denoiser's 3x3
        convolutions multiply e4m3fn operands -- the layers Kern.fp8_ok names at inference; the backward pass keeps
        bf16 operands (the saved activation and the bf16 shadow weights)."""
        kh, kw, stride, pad, ups = geom
        return (getattr(self.model, 'fp8_unet', False) and isinstance(wnames, str) and x.dim() == 4
                and x.dtype == torch.bfloat16 and kh * kw > 1 and not ups and x.shape[-1] % 16 == 0
                and x.shape[-1] >= 64 and wnames.startswith(self.model.fp8_prefix))

    def conv_skip_weights(self, n, dtype):
        """[W_out3 (3x3, tap-major K) | W_skip (1x1)] along K and the summed bias of ResBlock `n`: its second
        convolution and its skip convolution as ONE implicit GEMM (sdmi.h: a2 / a3).  Cached."""
        key = ('convskip', n, dtype)
        if key not in self.cache:
            with torch.no_grad():
                w2 = self.w(n + '.out_layers.3.weight', dtype)
                ws = self.w(n + '.skip_connection.weight', dtype)
                w = torch.cat([w2.reshape(w2.shape[0], -1), ws.reshape(ws.shape[0], -1)], 1).contiguous()
                b = (self.t[n + '.out_layers.3.bias'].float() + self.t[n + '.skip_connection.bias'].float())
                self.cache[key] = (w, b.contiguous())
        return self.cache[key]

    def ups_parity_weights(self, wname, dtype):
        """{(py, px): [Cout][2][2][Cin] in `dtype`} of an upsample convolution (ups_parity_split).  Cached."""
        key = ('upsparity', wname, dtype)
        if key not in self.cache:
            with torch.no_grad():
                parts = ups_parity_split(self.t[wname].float())
                w4 = torch.stack([parts[py, px] for py in (0, 1) for px in (0, 1)]).to(dtype).contiguous()
                self.cache[key] = {(z >> 1, z & 1): w4[z] for z in range(4)}       # views of one [4][Cout][4 Cin] tensor
        return self.cache[key]

    def ffout_proj_weights(self, t, n, dtype):
        """Feed-forward output and proj_out of a SpatialTransformer are two linear layers with only a
        residual add in between (attention.py:250-251, 305-308):
            out = (g Wff^T + bff + tres) Wpo^T + bpo + xres = [g | tres] [Wpo Wff | Wpo]^T + (Wpo bff + bpo) + xres
        -> combined weight [C, 4C + C] in `dtype` and fp32 bias (weight preparation, cached)."""
        key = ('ffoutproj', t, dtype)
        if key not in self.cache:
            with torch.no_grad():
                wff, bff = self.t[t + '.ff.net.2.weight'].float(), self.t[t + '.ff.net.2.bias'].float()
                wpo = self.t[n + '.proj_out.weight'].float().reshape(wff.shape[0], -1)
                bpo = self.t[n + '.proj_out.bias'].float()
                w = torch.cat([wpo @ wff, wpo], 1).to(dtype).contiguous()
                self.cache[key] = (w, (wpo @ bff + bpo).contiguous())
        return self.cache[key]

    def cross_fold_weights(self, t, dtype):
        """Weight-side operands of the folded slot cross-attention of transformer block `t`
        (engine.UNetRunner.cross_fold): (W_q * gamma_norm2)^T [C_k, C_d], (W_q beta_norm2) [1, C_d],
        ones [1, C_k] in `dtype`.  Weight preparation, cached until the weights change."""
        key = ('crossfold', t, dtype)
        if key not in self.cache:
            with torch.no_grad():
                g, be = self.t[t + '.norm2.weight'].float(), self.t[t + '.norm2.bias'].float()
                wq = self.t[t + '.attn2.to_q.weight'].float()
                self.cache[key] = ((wq * g).t().contiguous().to(dtype).unsqueeze(0),
                                   (wq @ be).to(dtype).view(1, 1, -1).contiguous(),
                                   torch.ones((1, 1, wq.shape[1]), dtype=dtype, device=wq.device))
        return self.cache[key]

    def ln_folded(self, ln_name, wnames, bnames, dtype):
        """Operands of a LayerNorm folded into the linear layer behind it (sdmi.h: ln_colsum):
        W' = W * gamma in `dtype`, colsum[n] = sum_k W'[n][k] of the rounded W', bias' = W beta + b.
        Weight preparation, once per weight update (cached like the bf16 operand shadows)."""
        if isinstance(wnames, str):
            wnames = (wnames,)
        key = ('lnfold', ln_name, wnames, dtype)
        if key not in self.cache:
            with torch.no_grad():
                g, be = self.t[ln_name + '.weight'].float(), self.t[ln_name + '.bias'].float()
                W = torch.cat([self.t[n].float().reshape(-1, g.numel()) for n in wnames])
                Wp = (W * g).to(dtype).contiguous()
                bias = W @ be
                if bnames is not None:
                    bias = bias + self.b(bnames).float()
                self.cache[key] = (Wp, Wp.float().sum(1).contiguous(), bias.contiguous())
        return self.cache[key]

    def cf_index(self, rows, cols, device):
        key = ('cf_index', rows, cols)
        if key not in self.cache:
            self.cache[key] = ops.cross_fold_pack_index(rows, cols).to(device)
        return self.cache[key]

    def st_index(self, which, C, device):
        key = ('st_index', which, C)
        if key not in self.cache:
            self.cache[key] = {'a': st_index_a, 'b': st_index_b, 'img': st_index_img}[which](C).to(device)
        return self.cache[key]

    def st_fused_weights(self, n, dtype):
        """Packed operands of the fused SpatialTransformer block `n` (sdmi.h: sdmi_st_block): weight streams of
        phase A / B and their fp32 epilogue vectors.  Same folds as ln_folded / ffout_proj_weights (LayerNorm
        gamma into the weights, beta into the bias; ff.net.2 and proj_out pre-multiplied).  Weight preparation,
        cached until the weights change."""
        key = ('st_fused', n, dtype)
        if key not in self.cache:
            with torch.no_grad():
                t = n + '.transformer_blocks.0'
                f = lambda k: self.t[k].float()
                C = f(n + '.proj_in.bias').numel()
                dev = self.t[n + '.proj_in.bias'].device
                w_in = f(n + '.proj_in.weight').reshape(C, C)
                g1, b1 = f(t + '.norm1.weight'), f(t + '.norm1.bias')
                wqkv = torch.cat([f(t + '.attn1.to_q.weight'), f(t + '.attn1.to_k.weight'), f(t + '.attn1.to_v.weight')])
                wqkv_p = (wqkv * g1).to(dtype)
                src_a = torch.cat([w_in.to(dtype).reshape(-1), wqkv_p.reshape(-1)])
                vec_a = torch.cat([f(n + '.proj_in.bias'), wqkv_p.float().sum(1), wqkv @ b1]).contiguous()
                wa = src_a[self.st_index('a', C, dev)].contiguous()
                wo, bo = f(t + '.attn1.to_out.0.weight'), f(t + '.attn1.to_out.0.bias')
                bo2 = f(t + '.attn2.to_out.0.bias')
                g3, b3 = f(t + '.norm3.weight'), f(t + '.norm3.bias')
                w1, bb1 = f(t + '.ff.net.0.proj.weight'), f(t + '.ff.net.0.proj.bias')
                w1_p = (w1 * g3).to(dtype)
                wff, bff = f(t + '.ff.net.2.weight'), f(t + '.ff.net.2.bias')
                wpo, bpo = f(n + '.proj_out.weight').reshape(C, C), f(n + '.proj_out.bias')
                src_b = torch.cat([wo.to(dtype).reshape(-1), wpo.to(dtype).reshape(-1), w1_p.reshape(-1),
                                   (wpo @ wff).to(dtype).reshape(-1)])
                wbs = src_b[self.st_index('b', C, dev)].contiguous()
                vec_b = torch.cat([bo, bo2, w1_p.float().sum(1), w1 @ b3 + bb1, wpo @ bff + bpo]).contiguous()
                self.cache[key] = dict(wa=wa, va=vec_a, wb=wbs, vb=vec_b, C=C)
        return self.cache[key]

    # ---- weight streams of the fused SpatialTransformer TRAINING kernels (sdmi.h: sdmi_st_train_fwd, sdmi_st_pack)
    def st_train_mats(self, n):
        """{key: (bf16 operand view [N][K], K)} of block `n` straight out of the shadow arena."""
        t = n + '.transformer_blocks.0'
        names = {'in': n + '.proj_in.weight', 'q': t + '.attn1.to_q.weight', 'k': t + '.attn1.to_k.weight',
                 'v': t + '.attn1.to_v.weight', 'o': t + '.attn1.to_out.0.weight', 'q2': t + '.attn2.to_q.weight',
                 'o2': t + '.attn2.to_out.0.weight', 'ff1': t + '.ff.net.0.proj.weight', 'ff2': t + '.ff.net.2.weight',
                 'po': n + '.proj_out.weight'}
        out = {}
        for k, nm in names.items():
            v = self._flat(nm, torch.bfloat16)
            assert v is not None and v.is_contiguous()
            out[k] = v
        return out

    def _st_descs(self, units, mats, dst):
        """numpy descriptor table (SdmiStPackDesc) of one stream: `units` per wave, destination tensor `dst`."""
        import numpy as np
        flat = [u for wave in units for u in wave]
        arr = np.zeros(len(flat), dtype=np.dtype([('src', '<u8'), ('dst', '<u8'), ('rs', '<i4'), ('cs', '<i4')]))
        base = dst.data_ptr()
        for i, (key, r0, k0, tr) in enumerate(flat):
            m = mats[key]
            ld = m.shape[1]
            if tr:      # unit rows walk the matrix's columns: element (r, k) = M[k0 + k][r0 + r]
                arr[i] = (m.data_ptr() + 2 * (k0 * ld + r0), base + i * 2048, 1, ld)
            else:
                arr[i] = (m.data_ptr() + 2 * (r0 * ld + k0), base + i * 2048, ld, 1)
        return arr

    ST_STREAMS = ('a', 'b', 'b1', 'b2', 'ba')

    def st_train_streams(self, n):
        """dict(wa, wb, wb1, wb2, wba) -- bf16 unit streams of block `n` for sdmi_st_train_fwd / _bwd.  The streams live in
        persistent buffers; after an optimiser step (invalidate()) the first request re-packs the streams of ALL
        registered blocks with one sdmi_st_pack launch (descriptor table on the device: part of the captured train step)."""
        if self._stp_stale:
            self._st_pack_all()
        ent = self._stp.get(n)
        if ent is not None and ent['epoch'] == self._stp_epoch and ent['shadow'] == self.model.shadow_arena().data_ptr():
            return ent
        if torch.cuda.is_current_stream_capturing():
            raise _lib.SdmiError(f'{n}: first use of a fused training block inside a graph capture (its descriptor '
                                 'table is a host-to-device copy): run the warm-up passes before capturing')
        import numpy as np
        mats = self.st_train_mats(n)
        C = mats['in'].shape[0]
        dev = mats['in'].device
        units = st_train_units(C)
        if ent is None:
            ent = {'w' + k: torch.empty((sum(len(w) for w in units[k]) * 1024,), dtype=torch.bfloat16, device=dev)
                   for k in self.ST_STREAMS}
        ent['descs'] = np.concatenate([self._st_descs(units[k], mats, ent['w' + k]) for k in self.ST_STREAMS])
        ent['epoch'], ent['shadow'], ent['C'] = self._stp_epoch, self.model.shadow_arena().data_ptr(), C
        self._stp[n] = ent
        self._stp_table = None                      # new member: rebuild the device table at the next re-pack
        tab = torch.from_numpy(ent['descs'].view(np.uint8).copy()).to(dev)
        call('sdmi_st_pack', _st(), descs=_p(tab), n_units=len(ent['descs']))
        ent['_tab'] = tab                            # (kept alive past the launch)
        return ent

    def _st_pack_all(self):
        self._stp_stale = False
        if not self._stp:
            return
        self._stp_epoch += 1
        if self._stp_table is None:
            if torch.cuda.is_current_stream_capturing():
                raise _lib.SdmiError('stream descriptor table rebuilt inside a graph capture')
            import numpy as np
            arr = np.concatenate([e['descs'] for e in self._stp.values()])
            dev = next(iter(self._stp.values()))['wa'].device
            self._stp_table = (torch.from_numpy(arr.view(np.uint8).copy()).to(dev), len(arr))
        call('sdmi_st_pack', _st(), descs=_p(self._stp_table[0]), n_units=self._stp_table[1])
        for e in self._stp.values():
            e['epoch'] = self._stp_epoch

    def b(self, names):
        """fp32 bias vector, fused across names (view when adjacent)."""
        if names is None:
            return None
        if isinstance(names, str):
            return self.t[names]
        key = ('bias', names)
        if key in self.cache:
            return self.cache[key]
        parts = [self.t[n] for n in names]
        out = torch.empty((sum(p.numel() for p in parts),), dtype=torch.float32,
                          device=parts[0].device)
        if all(p.dtype == torch.float32 and p.is_contiguous() for p in parts):
            o, items = 0, []
            for p in parts:
                items.append((_p(p), _p(out[o:]), p.numel() * 4))
                o += p.numel()
            _copy_group(items)
            self.cache[key] = out
            return out
        o = 0
        for p in parts:
            ops.cast2d(p.view(-1, 1), torch.float32, out=out[o:o + p.numel()].view(-1, 1))
            o += p.numel()
        self.cache[key] = out
        return out

    def wd(self, names, dtype, kh, kw, cin_x):
        """dgrad operand of the (fused) forward operand: [Cin][kh'][kw'][CoutPad], taps flipped.

        The operands live in persistent buffers; after an optimiser step (invalidate()) the first
        request re-packs ALL of them with one sdmi_pack_dgrad_batch launch (descriptor table on
        the device) instead of one small launch per layer.  Operands whose forward operand is a
        materialised copy (no fixed arena address) are packed individually."""
        key = ('dgrad', names if not isinstance(names, str) else (names,), dtype)
        if self._wd_stale:
            self._repack_all()
        ent = self._wd.get(key)
        if ent is not None and ent[1] == self._wd_epoch:
            return ent[0]
        w = self.w(names, dtype)
        n = w.shape[0]
        vec = ops.vec_of(dtype)
        npad = (n + vec - 1) // vec * vec
        dst = ent[0] if ent is not None else \
            (torch.zeros if npad != n else torch.empty)((cin_x * kh * kw, npad), dtype=dtype,
                                                        device=w.device)
        call('sdmi_pack_dgrad', _st(), src=_p(w), dst=_p(dst), dtype=_DT[dtype], Cout=n, KH=kh,
             KW=kw, Cin=cin_x, CoutPad=npad)
        stable = self._in_arena(w)
        self._wd[key] = [dst, self._wd_epoch, (w if stable else None, n, kh, kw, cin_x, npad)]
        if stable:
            self._wd_table = None          # new member: rebuild the descriptor table
        return dst

    @staticmethod
    def _fill_desc(d, e, blk):
        w, n, kh, kw, cin, npad = e[2][:6]
        d.src, d.dst = w.data_ptr(), e[0].data_ptr()
        d.Cout, d.KH, d.KW, d.Cin, d.CoutPad, d.block_begin = n, kh, kw, cin, npad, blk
        taps = kh * kw
        if len(e[2]) > 6:                       # parity sub-filter: selected taps only
            d.kh0, d.kw0, d.kstep, d.nkh, d.nkw = e[2][6]
            taps = d.nkh * d.nkw
        return blk + taps * ((n + 63) // 64) * ((cin + 63) // 64)

    def wd_sub(self, names, dtype, kh, kw, cin_x, kh0, kw0, step):
        """Parity sub-filter of the dgrad operand (taps kh0::step, kw0::step of the flipped filter,
        [Cin][nkh][nkw][CoutPad]) for the data gradient of a strided convolution; kept and re-packed
        with the other dgrad operands (one sdmi_pack_dgrad_batch launch per optimiser step)."""
        names = names if not isinstance(names, str) else (names,)
        key = ('dgradsub', names, dtype, kh0, kw0, step)
        if self._wd_stale:
            self._repack_all()
        ent = self._wd.get(key)
        if ent is not None and ent[1] == self._wd_epoch:
            return ent[0]
        w = self.w(names, dtype)
        n = w.shape[0]
        vec = ops.vec_of(dtype)
        npad = (n + vec - 1) // vec * vec
        nkh, nkw = len(range(kh0, kh, step)), len(range(kw0, kw, step))
        dst = ent[0] if ent is not None else \
            (torch.zeros if npad != n else torch.empty)((cin_x * nkh * nkw, npad), dtype=dtype, device=w.device)
        stable = self._in_arena(w)
        e = [dst, self._wd_epoch, (w if stable else None, n, kh, kw, cin_x, npad, (kh0, kw0, step, nkh, nkw))]
        # (first use / unstable operand: a one-entry table -- eager only, it copies the table to the device)
        Desc = _lib.CSTRUCT['SdmiPackDesc']
        arr = (Desc * 1)()
        blocks = self._fill_desc(arr[0], [dst, 0, (w,) + e[2][1:]], 0)
        dev = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(w.device)
        call('sdmi_pack_dgrad_batch', _st(), descs=_p(dev), n_desc=1, dtype=_DT[dtype], total_blocks=blocks)
        self._wd[key] = e
        if stable:
            self._wd_table = None
        return dst

    def _in_arena(self, t):
        m = self.model
        for a in (m.arena(), m.shadow_arena() if m.compute_dtype != torch.float32 else None):
            if a is not None and a.data_ptr() <= t.data_ptr() < a.data_ptr() + a.numel() * a.element_size():
                return True
        return False

    def _repack_all(self):
        self._wd_stale = False
        self._wd_epoch += 1
        for dtype in {k[2] for k, e in self._wd.items() if e[2][0] is not None}:
            items = [e for k, e in self._wd.items() if k[2] == dtype and e[2][0] is not None]   # full + sub-filters
            tab = self._wd_table.get(dtype) if self._wd_table else None
            if tab is None:
                Desc = _lib.CSTRUCT['SdmiPackDesc']
                arr = (Desc * len(items))()
                blk = 0
                for d, e in zip(arr, items):
                    blk = self._fill_desc(d, e, blk)
                dev = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(
                    items[0][0].device)
                tab = (dev, len(items), blk)
                self._wd_table = self._wd_table or {}
                self._wd_table[dtype] = tab
            call('sdmi_pack_dgrad_batch', _st(), descs=_p(tab[0]), n_desc=tab[1], dtype=_DT[dtype],
                 total_blocks=tab[2])
            for e in items:
                e[1] = self._wd_epoch


class Kern:
    """Inference provider."""
    training = False

    def __init__(self, wb):
        self.wb = wb

    def conv(self, x, wname, bname=None, *, kh=3, kw=3, stride=1, pad=(1, 1, 1, 1), ups=False,
             rowvec=None, residual=None, out_dtype=None, ldc=None, zero_pad=True):
        if isinstance(x, CatPair):
            if kh == 1 and kw == 1 and stride == 1 and not ups:
                return ops.conv2d(x.a, self.wb.w(wname, x.dtype), self.wb.b(bname), kh=1, kw=1, pad=pad,
                                  rowvec=rowvec, residual=residual, out_dtype=out_dtype, ldc=ldc, x2=x.b)
            x = x.materialize()
        if ups and _UPS_PARITY and kh == 3 and kw == 3 and stride == 1 and pad == (1, 1, 1, 1) and \
                x.dtype == torch.bfloat16 and x.shape[-1] % 64 == 0 and rowvec is None and residual is None and \
                out_dtype is None and ldc is None:
            # nearest-2x upsample + 3x3 convolution = four 2x2 convolutions of the low-resolution input, one per
            # output parity, written interleaved (WeightBank.ups_parity_weights): 2.25x fewer multiply-adds
            B, H, W_, _ = x.shape
            ws = self.wb.ups_parity_weights(wname, x.dtype)
            out = torch.empty((B, 2 * H, 2 * W_, ws[0, 0].shape[0]), dtype=x.dtype, device=x.device)
            bias = self.wb.b(bname)
            # all four parities in one launch (sdmi.h: parity4; the filters are one [4][N][4 Cin] tensor)
            w4 = next(iter(ws.values()))._base
            return ops.conv2d(x, w4, bias, kh=2, kw=2, out=out, parity4=True)
        if x.dtype == torch.uint8 or self.fp8_ok(x, wname, kh * kw, ups):
            # BASELINE "fp8 MFMA UNet": e4m3fn operands (activations at a fixed scale -- written by
            # the GroupNorm in front when there is one -- weights at 448 / amax), fp32 accumulation,
            # the scales undone in the epilogue's alpha
            w8, inv = self.wb.w8(wname)
            x8 = x if x.dtype == torch.uint8 else ops.quant_fp8(x, FP8_ACT_SCALE)
            return ops.conv2d(x8, w8, self.wb.b(bname), kh=kh, kw=kw,
                              stride=stride, pad=pad, rowvec=rowvec, residual=residual,
                              out_dtype=out_dtype or self.wb.dtype, ldc=ldc, alpha=inv / FP8_ACT_SCALE)
        return ops.conv2d(x, self.wb.w(wname, x.dtype), self.wb.b(bname), kh=kh, kw=kw,
                          stride=stride, pad=pad, ups=ups, rowvec=rowvec, residual=residual,
                          out_dtype=out_dtype, ldc=ldc, zero_pad=zero_pad)

    def fp8_ok(self, x, wname, taps, ups):
        """fp8 operands for the UNet's 3x3 convolutions (bf16 storage path, >= 64 input channels in
        16-byte rows); everything else -- 1x1 / linear layers, attention, the 3 / 4-channel
        in / out convolutions, the VQ-VAE and the slot encoder -- stays bf16."""
        return (getattr(self.wb.model, 'fp8_unet', False) and x.dtype == torch.bfloat16 and taps > 1
                and not ups and x.shape[-1] % 16 == 0 and x.shape[-1] >= 64
                and wname.startswith(self.wb.model.fp8_prefix))

    def linear(self, x, wnames, bnames=None, *, act=None, residual=None, out_dtype=None):
        return ops.linear(x, self.wb.w(wnames, x.dtype), self.wb.b(bnames), act=act,
                          residual=residual, out_dtype=out_dtype)

    def gn(self, x, name, *, eps, act=None, residual=None, dropout=None, for_conv=None, rowsum_of=None):
        if isinstance(x, CatPair):
            f8 = FP8_ACT_SCALE if (for_conv and self.fp8_ok(x, for_conv, 9, False)) else None
            return ops.group_norm(x.a, self.wb.f(name + '.weight'), self.wb.f(name + '.bias'), eps=eps,
                                  act=act, residual=residual, fp8_scale=f8, x2=x.b)
        return self._gn(x, name, eps=eps, act=act, residual=residual, dropout=dropout, for_conv=for_conv)

    def _gn(self, x, name, *, eps, act=None, residual=None, dropout=None, for_conv=None):
        """dropout = site name: training-mode dropout behind the activation (no-op at inference).
        for_conv = weight name of the 3x3 convolution that is the output's only reader: in the fp8
        configuration the norm writes that convolution's e4m3fn operand directly."""
        f8 = FP8_ACT_SCALE if (for_conv and self.fp8_ok(x, for_conv, 9, False)) else None
        return ops.group_norm(x, self.wb.f(name + '.weight'), self.wb.f(name + '.bias'), eps=eps,
                              act=act, residual=residual, fp8_scale=f8)

    def linear_multi(self, x, wname_list):
        """Several bias-free projections of one input -> tuple of outputs."""
        return tuple(self.linear(x, n) for n in wname_list)

    # fan-out forms: (result, alias(es) of x for x's other consumers) -- plain x at inference
    def gn_fan(self, x, name, *, eps, act=None, residual=None, n_alias=1, for_conv=None):
        return (self.gn(x, name, eps=eps, act=act, residual=residual, for_conv=for_conv),) + (x,) * n_alias

    def ln_fan(self, x, name):
        return self.ln(x, name), x

    def ln_linear_fan(self, x, ln_name, wnames, bnames=None, *, act=None, geglu=False, eps=1e-5):
        """linear(LayerNorm(x)) (-> GEGLU) in ONE launch at inference: the norm is folded into the
        GEMM (sdmi.h: ln_colsum), the gated activation into its epilogue.  -> (out, x for the
        residual branch).  bf16 (throughput) path only: the fp32 path keeps the reference's kernel
        sequence for the parity tests.  SDMI_LN_FOLD=0: the unfused sequence everywhere."""
        # (the folded GEMM costs ~1.45x a plain one on 128 x 128 tiles but saves a dependent launch per site: it wins end
        # to end at every size -- 94.75 vs 95.6 / 95.8 ms per sampling pass, round 3)
        if not _LN_FOLD or x.dtype != torch.bfloat16:
            h = self.linear(self.ln(x, ln_name, eps), wnames, bnames, act=act)
            return (self.geglu(h) if geglu else h), x
        w, colsum, bias = self.wb.ln_folded(ln_name, wnames, bnames, x.dtype)
        return ops.linear(x, w, bias, act=act, ln_colsum=colsum, ln_eps=eps, geglu=geglu), x

    def conv_fan(self, x, wname, bname=None, **kw):
        return self.conv(x, wname, bname, **kw), x

    def linear_fan(self, x, wnames, bnames=None):
        return self.linear(x, wnames, bnames), x

    def deconv(self, x, wname, bname, *, k, stride, pad, act='relu'):
        return deconv_forward(self.wb, x, wname, bname, k, stride, pad, act)

    def broadcast_pos(self, x, pos, dtype):
        return BroadcastPosFn.forward(None, x, pos, dtype)

    def sa_combine(self, o, B, N):
        BN, H, W_, ld = o.shape
        recon = torch.empty((B, H, W_, 4), dtype=torch.float32, device=o.device)
        masks = torch.empty((B, N, H * W_), dtype=torch.float32, device=o.device)
        call('sdmi_sa_combine', _st(), o=_p(o), recon=_p(recon), masks=_p(masks), dtype=_DT[o.dtype],
             B=B, N=N, HW=H * W_, ldo=ld)
        return recon, masks

    def ln(self, x, name, eps=1e-5):
        return ops.layer_norm(x, self.wb.f(name + '.weight'), self.wb.f(name + '.bias'), eps=eps)

    def attn_self(self, qkv, heads, head_dim=32):
        C = heads * head_dim
        fits = ops.ATTN_MFMA_MAX_KV if (qkv.dtype == torch.bfloat16 and head_dim == 32) else ops.ATTN_LDS_MAX_KV
        if qkv.shape[1] > fits:          # e.g. the 785 tokens x head_dim 64 of the DINO ViT
            return ops.attention_long(qkv[..., :C], qkv[..., C:2 * C], qkv[..., 2 * C:], heads, head_dim)
        return ops.attention(qkv[..., :C], qkv[..., C:2 * C], qkv[..., 2 * C:], heads,
                             head_dim=head_dim)

    def attn_cross(self, q, kv, heads):
        C = heads * 32
        return ops.attention(q, kv[..., :C], kv[..., C:], heads)

    def res_tail(self, h, n, skip):
        """out = out_layers.3(h) + (skip_connection(skip) if the block changes its channel count else skip).
        bf16 inference: the 3x3 convolution and the 1x1 skip convolution are ONE implicit GEMM over
        [im2col(h) | skip] (or [im2col(h) | a | b] for a skip that is a lazy concat)."""
        has_skip = (n + '.skip_connection.weight') in self.wb.t
        if has_skip and _RES_MERGE and torch.is_tensor(h) and h.dtype == torch.bfloat16 and h.shape[-1] % 64 == 0:
            srcs = (skip.a, skip.b) if isinstance(skip, CatPair) else (skip,)
            if all(t.shape[-1] % 64 == 0 and t.dtype == h.dtype for t in srcs):
                w, b = self.wb.conv_skip_weights(n, h.dtype)
                return ops.conv2d(h, w, b, kh=3, kw=3, pad=(1, 1, 1, 1), x2=srcs[0],
                                  x3=(srcs[1] if len(srcs) > 1 else None))
        if has_skip:
            skip = self.conv(skip, n + '.skip_connection.weight', n + '.skip_connection.bias', kh=1, kw=1,
                             pad=(0, 0, 0, 0))
        elif hasattr(skip, 'materialize'):             # (a lazy concat that keeps its channel count)
            skip = skip.materialize()
        return self.conv(h, n + '.out_layers.3.weight', n + '.out_layers.3.bias', residual=skip)

    def ff_out_proj(self, g, tres, xres, t, n):
        """ff.net.2 (+ token residual) followed by proj_out (+ block residual): ONE GEMM over the two
        sources [g | tres] with pre-multiplied weights at bf16 inference, the two launches otherwise."""
        if _FF_MERGE and g.dtype == torch.bfloat16 and g.shape[-1] % 64 == 0 and tres.shape[-1] % 64 == 0:
            w, b = self.wb.ffout_proj_weights(t, n, g.dtype)
            B, HW, C = tres.shape
            out = ops.conv2d(g.view(B, HW, 1, g.shape[-1]), w, b, kh=1, kw=1, pad=(0, 0, 0, 0),
                             residual=xres.view(B, HW, 1, C), x2=tres.view(B, HW, 1, C))
            return out.view(B, HW, C)
        tok = self.linear(g, t + '.ff.net.2.weight', t + '.ff.net.2.bias', residual=tres)
        return self.linear(tok, n + '.proj_out.weight', n + '.proj_out.bias', residual=xres)

    def cross_prepare(self, kv, t, heads):
        """Once per sampling call (slots and weights are fixed over the NFEs): the slot keys (7 of the image models in
        8-row groups per head; the video models' 11 / 15 in 16-row groups, savi_diffusion.py:143-144) folded
        into the query projection, the values into the output projection -- per image
          Wq[b] = (scale K_b restricted per head) (W_q gamma)   [heads*8, C]   (+ LayerNorm-fold terms)
          W2[b] = W_o (V_b restricted per head)^T               [C, heads*8]
        so that slot cross-attention is two small per-image GEMMs per evaluation (cross_block)."""
        if not (_CROSS_FOLD and kv.dtype == torch.bfloat16 and kv.shape[1] <= _CROSS_MAX_SLOTS):
            return None
        wt, tb, ones = self.wb.cross_fold_weights(t, kv.dtype)
        B, C = kv.shape[0], kv.shape[-1] // 2
        gw = 8 if kv.shape[1] <= 8 else 16             # score columns per head (softmax group width)
        R = heads * gw
        kexp, vexp = ops.expand_heads(kv, heads, float(C // heads) ** -0.5, gw)
        # (the fused block pads the score matrix to 128 columns: 16-row groups fit C = 256 -- 8 heads -- only)
        fused = _ST_FUSED and C in (256, 384) and R <= 128
        if fused:
            # the fused block (sdmi_st_block) takes the same operands padded to 128 score columns, as a unit
            # stream: Wq[b] / W2[b] are produced straight into the padded storage (pads stay zero)
            # (allocated per call: the fold dict hands out VIEWS of these -- a cached buffer would be overwritten by the
            # next context of the same batch size (cond / uncond, two live generators) while an earlier fold is alive)
            src = torch.zeros((B, 128 * C + C * 128), dtype=kv.dtype, device=kv.device)
            vec = torch.zeros((B, 256), dtype=torch.float32, device=kv.device)
            wq = src[:, :R * C].view(B, R, C)
            w2 = src[:, 128 * C:].view(B, C, 128)[:, :, :R]
            colsum, biasq = vec[:, :R].unsqueeze(-1), vec[:, 128:128 + R].unsqueeze(-1)
        else:
            wq = torch.empty((B, R, C), dtype=kv.dtype, device=kv.device)
            w2 = torch.empty((B, C, R), dtype=kv.dtype, device=kv.device)
            colsum = torch.empty((B, R, 1), dtype=torch.float32, device=kv.device)
            biasq = torch.empty((B, R, 1), dtype=torch.float32, device=kv.device)
        ops.bmm_nt(kexp, wt, wq)
        ops.bmm_nt(wq, ones, colsum)
        ops.bmm_nt(kexp, tb, biasq)
        wo = self.wb.w(t + '.attn2.to_out.0.weight', kv.dtype)
        ops.bmm_nt(wo.view(1, C, C).expand(B, -1, -1), vexp, w2)
        fold = dict(wq=wq, colsum=colsum.squeeze(-1), biasq=biasq.squeeze(-1), w2=w2, slots=kv.shape[1])
        if fused:
            fold['st_img'] = ops.gather_rows(src, self.wb.st_index('img', C, kv.device))
            fold['st_vec'] = vec
        return fold

    def cross_block(self, tok, t, kvp, heads):
        """norm2 -> slot cross-attention -> to_out + residual of transformer block `t` -> new tok."""
        fold = kvp.get('fold') if isinstance(kvp, dict) else None
        if fold is not None:
            B, HW, C = tok.shape
            # few tokens per image (the 4^2 / 8^2 levels): the layer is a per-image weight stream -- one launch, one
            # workgroup per 16 tokens (sdmi_cross_fold) instead of two batched GEMMs on mostly empty 64 x 64 tiles
            if _CROSS_ONE and HW % 16 == 0 and HW <= _CROSS_ONE and tok.is_contiguous() and fold['slots'] <= 8 and \
                    (C, fold['wq'].shape[1]) in ops.CROSS_FOLD_SHAPES:
                if 'cf_wq' not in fold:          # once per sampling call: the operands in MFMA-fragment order
                    R = fold['wq'].shape[1]
                    iq = self.wb.cf_index(R, C, tok.device)
                    i2 = self.wb.cf_index(C, R, tok.device)
                    fold['cf_wq'] = ops.gather_rows(fold['wq'].reshape(B, R * C), iq)
                    fold['cf_w2'] = ops.gather_rows(fold['w2'].reshape(B, C * R), i2)
                return ops.cross_fold(tok, fold['cf_wq'], fold['colsum'], fold['biasq'], fold['cf_w2'],
                                      self.wb.b(t + '.attn2.to_out.0.bias'), 1e-5, fold['slots'], packed=True)
            P = ops.cross_scores(tok, fold['wq'], fold['colsum'], fold['biasq'], 1e-5, fold['slots'])
            return ops.bmm_nt(P, fold['w2'], torch.empty_like(tok), bias=self.wb.b(t + '.attn2.to_out.0.bias'),
                              residual=tok)
        kv = kvp['kv'] if isinstance(kvp, dict) else kvp
        q, tres = self.ln_linear_fan(tok, t + '.norm2', t + '.attn2.to_q.weight')
        a = self.attn_cross(q, kv, heads)
        return self.linear(a, t + '.attn2.to_out.0.weight', t + '.attn2.to_out.0.bias', residual=tres)

    def st_fused(self, x, n, heads, kvp):
        """The whole SpatialTransformer block `n` in two launches (sdmi.h: sdmi_st_block) -- bf16 inference,
        C = 256 / 384, 64 | tokens per image <= 256, folded slot cross-attention (<= 7 slots; C = 256: <= 16).  None when the
        block does not qualify (the caller runs the per-layer launches)."""
        fold = kvp.get('fold') if isinstance(kvp, dict) else None
        B, H, W, C = x.shape
        S = H * W
        if not (_ST_FUSED and fold is not None and 'st_img' in fold and x.dtype == torch.bfloat16 and
                C in (256, 384) and heads * 32 == C and S % 64 == 0 and S <= 256 and x.is_contiguous()):
            return None
        rows = _ST_ROWS or (64 if B * S // 64 >= 192 else 32)
        if B * S // rows < _ST_MIN_WGS:
            return None
        wts = self.wb.st_fused_weights(n, x.dtype)
        tok = torch.empty((B, S, C), dtype=x.dtype, device=x.device)
        qkv = torch.empty((B, S, 3 * C), dtype=x.dtype, device=x.device)
        out = torch.empty_like(x)
        # algorithmic work of the block (bench.py's roofline legs): its GEMMs + the two attention contractions
        g = st_geometry(C)
        flops = 2.0 * B * S * (C * C * 4 + C * C + 2 * 128 * C + 8 * C * C + 5 * C * C) + 4.0 * B * S * S * C
        wbytes = 2.0 * (wts['wa'].numel() + wts['wb'].numel()) + 2.0 * fold['st_img'].numel()
        # a grid that leaves half the chip idle (64 images at 8^2: 128 workgroups of 32 rows): the feed-forward of phase B
        # split over workgroup PAIRS, each streaming half of the hidden chunks; the pair's fp32 partials are summed (+ bias
        # + x) by the kernel behind like a split-K convolution's second stage (ops.defer_splitk: the next GroupNorm's
        # prologue, or sdmi_splitk_finish in front of any other reader)
        split = 2 if (_ST_FF_SPLIT and ops._DEFER[0] and B * S // rows <= 128) else 1
        part = torch.empty((2 * B * S * C,), dtype=torch.float32, device=x.device) if split == 2 else None
        call('sdmi_st_block', _st(), x=_p(x), tok=_p(tok), qkv=_p(qkv), out=_p(out),
             gn_gamma=_p(self.wb.f(n + '.norm.weight')), gn_beta=_p(self.wb.f(n + '.norm.bias')),
             wstream_a=_p(wts['wa']), vec_a=_p(wts['va']), wstream_b=_p(wts['wb']), vec_b=_p(wts['vb']),
             wstream_img=_p(fold['st_img']), vec_img=_p(fold['st_vec']), B=B, S=S, C=C, slots=fold['slots'],
             phase=0, gn_eps=1e-6, ln_eps=1e-5, attn_scale=32.0 ** -0.5, rows=rows, part=_p(part), ff_split=split,
             _meta=dict(flops=flops, bytes=2.0 * B * S * C * 2 + wbytes))
        if split == 2:
            bias = wts['vb'][18 * C:19 * C]
            kw = dict(out=_p(out), workspace=_p(part), bias=_p(bias), residual=_p(x), ldr=C, dtype=_lib.BF16,
                      out_dtype=_lib.BF16, M=B * S, N=C, ldc=C, B=B, H=H, W=W, Ho=H, Wo=W, alpha=1.0, act=0, batch=1)
            ops._PENDING[out.data_ptr()] = ops._PendingSplit(out, kw, 2, (part, bias, None, x, 1.0))
        return out

    def geglu(self, h):
        return ops.geglu(h)

    def add_pos(self, x, pos):
        return ops.add_pos(x, pos)

    def rowvec_slices(self, rowvecs, bounds):
        """[B, total] -> {i: rowvecs[:, off:off+c]} (strided views; the kernels take the row pitch)."""
        return [rowvecs[:, off:off + c] for off, c in bounds]

    def concat(self, a, b):
        """Channel concatenation for the UNet's skip connections.  At inference the two readers of the
        result -- the ResBlock's first GroupNorm and its 1x1 skip convolution -- take the two tensors
        in place (CatPair: sdmi.h x2 / a2), so nothing is materialised."""
        if _LAZY_CAT and a.dim() == 4 and a.shape[-1] % 64 == 0 and b.shape[-1] % 64 == 0:
            return CatPair(a, b)
        return ops.concat_channels(a, b)

    def cast(self, x, dtype):
        return x if x.dtype == dtype else ops.act(x, None, dtype)

    def cast_pad(self, x, dtype, cols, ldd):
        """First `cols` channels of x [..., ld] -> dtype [..., ldd] (zero padded)."""
        return ops.cast2d(x, dtype, cols=cols, ldd=ldd)

    def dropout(self, x, site='unet'):
        return x

    def linear_drop_res(self, x, wname, bname, res):
        """res + dropout(linear(x)); without dropout the residual add is the GEMM epilogue."""
        return self.linear(x, wname, bname, residual=res)

    def slot_attention(self, kv, init, name, iters, eps):
        wb = self.wb
        D = kv.shape[-1] // 2
        P = dict(lnq_g=wb.f(f'{name}.project_q.0.weight'), lnq_b=wb.f(f'{name}.project_q.0.bias'),
                 wq=wb.f(f'{name}.project_q.1.weight'), w_ih=wb.f(f'{name}.gru.weight_ih'),
                 w_hh=wb.f(f'{name}.gru.weight_hh'), b_ih=wb.f(f'{name}.gru.bias_ih'),
                 b_hh=wb.f(f'{name}.gru.bias_hh'), lnm_g=wb.f(f'{name}.mlp.0.weight'),
                 lnm_b=wb.f(f'{name}.mlp.0.bias'), w1=wb.f(f'{name}.mlp.1.weight'),
                 b1=wb.f(f'{name}.mlp.1.bias'), w2=wb.f(f'{name}.mlp.3.weight'),
                 b2=wb.f(f'{name}.mlp.3.bias'))
        return ops.slot_attention(kv[..., :D], kv[..., D:], init, P, iters=iters, eps=eps)


# ==========================================================================================
# training provider: autograd Functions with libsdmi backward kernels
# ==========================================================================================
def _grads_of(wb, names):
    """Contiguous gradient-arena destination for (fused) params, or None if not adjacent."""
    m = wb.model
    g = m.grad_arena()
    if isinstance(names, str):
        names = (names,)
    offs = [m._offsets[n] for n in names]
    for i in range(len(offs) - 1):
        if offs[i][0] + offs[i][1] != offs[i + 1][0]:
            return None
    return g[offs[0][0]:offs[-1][0] + offs[-1][1]]


class _RowvecSink:
    """Gradient matrix [B, total] shared by the column slices RowvecSplitFn hands out: every consumer
    writes its slice (rowgroup_sum with an output pitch); allocated at the first write of a backward."""

    def __init__(self, B, total):
        self.B, self.total, self.buf = B, total, None
        self.filled = set()        # offsets whose slice a GroupNorm backward has written already

    def buffer(self, device):
        if self.buf is None:
            self.buf = torch.empty((self.B, self.total), dtype=torch.float32, device=device)
        return self.buf


class RowvecSplitFn(torch.autograd.Function):
    """rowvecs [B, total] -> its column slices (one per ResBlock: the time-embedding rows).  Plain
    slicing costs three framework kernels per slice in backward (zeros, copy, accumulate): here the
    consumers' backward kernels write straight into one shared gradient matrix, which this backward
    just returns.  The slices must tile [0, total) and each be consumed exactly once."""

    @staticmethod
    def forward(ctx, rowvecs, bounds):
        B, total = rowvecs.shape
        assert bounds[0][0] == 0 and bounds[-1][0] + bounds[-1][1] == total
        sink = _RowvecSink(B, total)
        ctx.sink = sink
        outs = []
        for off, c in bounds:
            v = rowvecs[:, off:off + c]
            outs.append(v)
        ctx.set_materialize_grads(False)
        ctx.n = len(bounds)
        return tuple(outs)

    @staticmethod
    def backward(ctx, *grads):
        sink = ctx.sink
        assert sink.buf is not None and all(g is not None for g in grads), 'every slice needs its consumer'
        buf, sink.buf = sink.buf, None
        sink.filled.clear()
        return buf, None


_CUS = {}


def _n_cus(device):
    """Compute units of `device` (cached)."""
    key = str(device)
    if key not in _CUS:
        _CUS[key] = int(torch.cuda.get_device_properties(device).multi_processor_count)
    return _CUS[key]


class GemmFn(torch.autograd.Function):
    """out = conv/linear(x, W) + bias (+ rowvec[b]) (+ residual).  Activation-free."""

    @staticmethod
    def forward(ctx, x, rowvec, residual, anchor, wb, wnames, bnames, geom, out_dtype, ldc, n_alias=0):
        """n_alias = 1: also returns an alias of x for x's OTHER consumer (identity / skip branch);
        its gradient comes back into this backward and is summed inside the dgrad kernel's epilogue
        instead of by an autograd accumulation kernel."""
        kh, kw, stride, pad, ups = geom
        w = wb.w(wnames, x.dtype)
        b = wb.b(bnames)
        x8 = getattr(x, '_sdmi_fp8', None)           # e4m3fn copy written by the GroupNorm in front (KernGrad.gn)
        if x8 is not None and x8.shape == x.shape and wb.fp8_train_ok(x, wnames, geom):
            # e4m3fn operands in the forward GEMM: activations at the fixed scale of the inference path -- only where a
            # GroupNorm (+ SiLU) bounds them and wrote the operand itself: the stride-2 downsample convolutions read
            # the raw residual stream and keep bf16 in training -- weights at 448 / amax with amax taken on the device
            # this step (WeightBank.w8_dev); fp32 accumulation, both scales undone in the epilogue.  x stays bf16 for
            # the backward pass.
            w8, inv = wb.w8_dev(wnames)
            out = ops.conv2d(x8, w8, b, kh=kh, kw=kw, stride=stride, pad=pad,
                             rowvec=rowvec, residual=residual, out_dtype=out_dtype or x.dtype, ldc=ldc,
                             alpha=1.0 / FP8_ACT_SCALE, alpha_dev=inv)
        elif x.dim() == 4 and (kh, kw) != (0, 0):
            out = ops.conv2d(x, w, b, kh=kh, kw=kw, stride=stride, pad=pad, ups=ups, rowvec=rowvec,
                             residual=residual, out_dtype=out_dtype, ldc=ldc)
        else:
            out = ops.linear(x, w, b, residual=residual, out_dtype=out_dtype)
        ctx.save_for_backward(x)
        ctx.cfg = (wb, wnames, bnames, geom, rowvec is not None, residual is not None,
                   rowvec.shape if rowvec is not None else None)
        ctx.rv_sink = getattr(rowvec, '_sdmi_sink', None)      # (RowvecSplitFn: shared gradient matrix)
        ctx.set_materialize_grads(False)
        if n_alias:
            return out, x.view_as(x)
        return out

    @staticmethod
    def backward(ctx, dy, dalias=None):
        (x,) = ctx.saved_tensors
        if dalias is not None:
            dalias = dalias.contiguous()
        wb, wnames, bnames, geom, has_rv, has_res, rv_shape = ctx.cfg
        dx, dy, (N, ldy, B, Ho, Wo, dt) = GemmFn.core(wb, x, dy, wnames, bnames, geom,
                                                       ctx.needs_input_grad[0], dalias)
        drv = None
        if has_rv and ctx.needs_input_grad[1]:
            if ctx.rv_sink is not None:        # a column slice of the shared [B, total] gradient matrix
                sink, off = ctx.rv_sink
                drv = sink.buffer(x.device)[:, off:off + N]
                if off not in sink.filled:     # (else: the GroupNorm behind this conv summed its dx already)
                    call('sdmi_rowgroup_sum', _st(), x=_p(dy), out=_p(drv), dtype=_DT[dt], groups=B,
                         rows_per=Ho * Wo, N=N, ldx=ldy, ldo=drv.stride(0))
            else:
                drv = torch.empty(rv_shape, dtype=torch.float32, device=x.device)
                call('sdmi_rowgroup_sum', _st(), x=_p(dy), out=_p(drv), dtype=_DT[dt], groups=B,
                     rows_per=Ho * Wo, N=N, ldx=ldy)
        dres = None
        if has_res and ctx.needs_input_grad[2]:
            dres = dy if dy.shape[-1] == N else None
            assert dres is not None
        _dbg(f'gemm {wnames if isinstance(wnames, str) else wnames[0]}', dy=dy, dx=dx, drv=drv)
        return dx, drv, dres, None, None, None, None, None, None, None, None

    @staticmethod
    def core(wb, x, dy, wnames, bnames, geom, need_dx, dalias=None):
        """Weight / bias gradient (queued or launched) and data gradient of one conv / linear.
        -> (dx or None, dy in the compute dtype, (N, ldy, B, Ho, Wo, dt))."""
        kh, kw, stride, pad, ups = geom
        is_conv = x.dim() == 4 and (kh, kw) != (0, 0)
        if not is_conv:
            kh = kw = 1
            stride, pad, ups = 1, (0, 0, 0, 0), False
        dt = x.dtype
        vec = ops.vec_of(dt)
        w = wb.w(wnames, dt)
        N = w.shape[0]
        dy = dy.contiguous()
        # bring dY to the compute dtype with a vector-multiple row pitch
        if dy.dtype != dt or dy.shape[-1] % vec:
            npad = (N + vec - 1) // vec * vec
            dy = ops.cast2d(dy, dt, cols=N, ldd=npad)
        ldy = dy.shape[-1]
        Cin = x.shape[-1]
        if is_conv:
            B, H, W_, _ = x.shape
            Ho, Wo = dy.shape[1], dy.shape[2]
        else:
            B, H, W_, Ho, Wo = x.numel() // Cin, 1, 1, 1, 1
        M = B * Ho * Wo
        K = kh * kw * Cin
        # ---- weight / bias gradients straight into the gradient arena
        names = (wnames,) if isinstance(wnames, str) else tuple(wnames)
        dst = _grads_of(wb, names)
        k_true = wb.t[names[0]].numel() // wb.t[names[0]].shape[0]
        direct = dst is not None and k_true == K
        tiles = ((N + 127) // 128) * ((K + 127) // 128)
        # M is split so that ~0.75 workgroups per CU exist (the launches share the chip with the
        # dgrad chain and three more weight-gradient streams; sweep 128 ... 768 on one box: 192-256
        # best, 512 is +0.7 ... +1.8 ms per step) while each still walks >= 8 m-steps
        # (64 rows per step in the bf16 kernel, 32 in the fp32 one); short contractions take one
        # launch straight into the gradient arena
        mt = 64 if dt == torch.bfloat16 else 32
        splits = max(1, min((192 + tiles - 1) // tiles, M // (8 * mt), 512))
        if M <= 16 * mt:
            splits = 1
        if (dt == torch.bfloat16 and is_conv and kh == 3 and kw == 3 and stride == 1 and not ups and Cin == 64
                and N == 64 and tuple(pad) == (1, 1, 1, 1) and W_ % 64 == 0 and H % 4 == 0
                and B * (H // 4) * (W_ // 64) >= 2 * _n_cus(x.device)):
            # the direct 3x3 kernel (wgrad.hip: wgrad3x3_c64_kernel): one persistent workgroup per CU and slot
            splits = _n_cus(x.device)
        elif (dt == torch.bfloat16 and is_conv and kh == 3 and kw == 3 and stride == 1 and not ups and _WGRAD_HALO
                and tuple(pad) == (1, 1, 1, 1) and W_ in (16, 32, 64) and Cin % 64 == 0 and N % 64 == 0 and Ho == H and Wo == W_
                and (H * W_) % 256 == 0 and ((H * W_) & (H * W_ - 1)) == 0):
            # the direct 3x3 kernel on (64 output x 64 input channel) pairs (wgrad.hip: wgrad3x3_halo_kernel): one
            # persistent workgroup per CU -- slots x pairs fills the chip
            pairs = (N // 64) * (Cin // 64)
            halo_splits = min(_n_cus(x.device) // pairs, M // 256)
            if halo_splits >= 2 and pairs * halo_splits >= 128:        # (the kernel writes partials: never one split)
                splits = halo_splits
        bdst = _grads_of(wb, bnames) if bnames is not None else None
        lda = Cin if is_conv else x.stride(-2)
        # ---- data + weight gradient in ONE launch (sdmi_bwd_pair) when both are of the same loader
        # class: 1x1 / linear, or a stride-1 same-size convolution
        same = is_conv and stride == 1 and not ups and Ho == H and Wo == W_
        one = kh == 1 and kw == 1 and stride == 1 and not ups and pad[0] == 0 and pad[2] == 0
        kd = kh * kw * ldy                                   # contraction depth of the data gradient
        bk = 64 if kd * 2 >= 512 else 32
        pair = (wb.pair_bwd and need_dx and dt == torch.bfloat16 and direct and N > 64 and K > 64 and Cin > 64
                and (bnames is None or bdst is not None) and Cin % 8 == 0 and ldy % 8 == 0
                and (one or (same and kh * kw <= 32 and ldy % bk == 0
                             and pad[0] == pad[1] == (kh - 1) // 2 and pad[2] == pad[3] == (kw - 1) // 2))
                and (dalias is None or dalias.shape[-1] == Cin)
                and (M + (kh + 1) * W_ + 128) * max(lda, ldy, Cin) * 2 < (1 << 31) and N * kd * 2 < (1 << 31))
        if pair:
            wd = wb.wd(wnames, dt, kh, kw, Cin)
            dx = torch.empty(x.shape if not is_conv else (B, H, W_, Cin), dtype=dt, device=x.device)
            wb.pair_launch(
                dict(a=_p(dy), w=_p(wd), out=_p(dx), dtype=_DT[dt], out_dtype=_DT[dt], M=M, N=Cin, K=kd,
                     lda=ldy, ldw=kd, ldc=Cin, B=B, H=Ho, W=Wo, Cin=ldy, Ho=H, Wo=W_, KH=kh, KW=kw, stride=1,
                     pad_t=kh - 1 - pad[0], pad_l=kw - 1 - pad[2], ups=0, act=0, alpha=1.0, split_k=1, batch=1,
                     residual=_p(dalias), ldr=Cin),
                dict(a=_p(x), dy=_p(dy), dw=_p(dst), dbias=_p(bdst), dtype=_DT[dt], M=M, N=N, K=K, lda=lda,
                     ldy=ldy, B=B, H=H, W=W_, Cin=Cin, Ho=Ho, Wo=Wo, KH=kh, KW=kw, stride=stride,
                     pad_t=pad[0], pad_l=pad[2], ups=0, accumulate=1),
                (x, dy, dx))
            return dx, dy, (N, ldy, B, Ho, Wo, dt)
        wb.flush_pending_fold((_p(dst), _p(bdst)))     # (a pending fold into this parameter goes first)
        side = wb.side_stream(names[0])
        if side is not None:
            ev = torch.cuda.Event()
            ev.record()
            side.wait_event(ev)          # dy is ready
        with (torch.cuda.stream(side) if side is not None else contextlib.nullcontext()):
            GemmFn._wgrad(wb, x, dy, dt, names, bnames, dst, direct, splits, M, N, K, Cin, ldy, B,
                          H, W_, Ho, Wo, kh, kw, stride, pad, ups, is_conv)
        if side is not None:
            wb.defer(x, dy)
            wb._side_dst[_p(dst)] = side
            if bdst is not None:
                wb._side_dst[_p(bdst)] = side
        # ---- data gradient: the forward kernel on the flipped operand
        dx = None
        if need_dx:
            wd = wb.wd(wnames, dt, kh, kw, Cin)
            if is_conv and stride > 1 and not ups:
                dx, dalias = GemmFn._dgrad_strided(
                    dy, wd, B, H, W_, Ho, Wo, Cin, ldy, kh, kw, stride, pad, dt, dalias,
                    sub_of=lambda a0, b0: wb.wd_sub(wnames, dt, kh, kw, Cin, a0, b0, stride))
            elif is_conv:
                Hs, Ws = (2 * H, 2 * W_) if ups else (H, W_)
                out = torch.empty((B, Hs, Ws, Cin), dtype=dt, device=x.device)
                ws = None if stride > 1 else ops.splitk_workspace(B * Hs * Ws, Cin, kh * kw * ldy,
                                                                  dy.element_size(), x.device)
                call('sdmi_igemm', _st(), a=_p(dy), w=_p(wd), out=_p(out), dtype=_DT[dt], workspace=_p(ws),
                     out_dtype=_DT[dt], M=B * Hs * Ws, N=Cin, K=kh * kw * ldy, lda=ldy,
                     ldw=kh * kw * ldy, ldc=Cin, B=B, H=Ho, W=Wo, Cin=ldy, Ho=Hs, Wo=Ws, KH=kh,
                     KW=kw, stride=1, pad_t=kh - 1 - pad[0], pad_l=kw - 1 - pad[2], ups=0, act=0,
                     alpha=1.0, split_k=(0 if ws is not None else 1), batch=1,
                     zins=(stride if stride > 1 else 0), residual=(_p(dalias) if (dalias is not None and not ups) else 0), ldr=Cin)
                if dalias is not None and not ups:
                    dalias = None
                if ups:
                    dx = torch.empty_like(x)
                    call('sdmi_pool2x2_sum', _st(), x=_p(out), y=_p(dx), dtype=_DT[dt], B=B, H=H,
                         W=W_, C=Cin)
                else:
                    dx = out
            else:
                dx = ops.linear(dy.view(-1, ldy), wd,
                                residual=(dalias.view(-1, Cin) if dalias is not None else None)).view(x.shape)
                dalias = None
            if dalias is not None:          # (no epilogue to fold it into: nearest-x2 / partial parity cover)
                call('sdmi_add', _st(), x=_p(dx), z=_p(dalias), y=_p(dx), dtype=_DT[dt], n=dx.numel())
        elif dalias is not None:
            dx = dalias
        return dx, dy, (N, ldy, B, Ho, Wo, dt)

    @staticmethod
    def _dgrad_strided(dy, wd, B, H, W_, Ho, Wo, Cin, ldy, kh, kw, s, pad, dt, extra=None, sub_of=None):
        """Data gradient of a stride-s convolution as s*s plain stride-1 convolutions over dy, one
        per input-pixel parity (py, px): only the filter taps kh = (py + pad_t) mod s (+ s, ...)
        reach that parity, so each launch uses its sub-filter (a strided slice of the flipped
        operand) and writes its pixels interleaved into dx (igemm's sub-sampled output placement).
        No multiply-adds on inserted zeros: 1/s^2 of the work of the zero-insertion form."""
        w4 = wd.view(Cin, kh, kw, ldy)
        full = all(len(range((p + pd) % s, k, s)) > 0
                   for k, pd in ((kh, pad[0]), (kw, pad[2])) for p in range(s))
        dx = torch.empty((B, H, W_, Cin), dtype=dt, device=dy.device)
        if not full:
            ops.zero_(dx)
        # `extra` (gradient of x's other consumer) rides in the epilogue when every pixel is written
        fuse = extra is not None and full
        for py in range(s):
            ay = (py + pad[0]) % s
            nky = len(range(ay, kh, s))
            hs = len(range(py, H, s))
            for px in range(s):
                ax = (px + pad[2]) % s
                nkx = len(range(ax, kw, s))
                ws_ = len(range(px, W_, s))
                if nky == 0 or nkx == 0 or hs == 0 or ws_ == 0:
                    continue
                sub = sub_of((kh - 1 - ay) % s, (kw - 1 - ax) % s) if sub_of is not None else \
                    w4[:, (kh - 1 - ay) % s::s, (kw - 1 - ax) % s::s, :].contiguous()
                K = nky * nkx * ldy
                call('sdmi_igemm', _st(), a=_p(dy), w=_p(sub), out=_p(dx), dtype=_DT[dt],
                     out_dtype=_DT[dt], M=B * hs * ws_, N=Cin, K=K, lda=ldy, ldw=K, ldc=Cin, B=B,
                     H=Ho, W=Wo, Cin=ldy, Ho=hs, Wo=ws_, KH=nky, KW=nkx, stride=1,
                     pad_t=(nky - 1) - (py + pad[0] - ay) // s, pad_l=(nkx - 1) - (px + pad[2] - ax) // s,
                     ups=0, act=0, alpha=1.0, split_k=1, batch=1, oH=H, oW=W_, osy=s, osx=s, ooy=py,
                     oox=px, residual=(_p(extra) if fuse else 0), ldr=Cin)
        return dx, (None if fuse else extra)

    @staticmethod
    def _wgrad(wb, x, dy, dt, names, bnames, dst, direct, splits, M, N, K, Cin, ldy, B, H, W_, Ho,
               Wo, kh, kw, stride, pad, ups, is_conv):
        """Weight / bias gradient launches (on whatever stream is current)."""
        ws = torch.empty((splits * (N * K + N),), dtype=torch.float32, device=x.device)
        # (temporaries of non-direct destinations are overwritten by the kernel, then added to the arena)
        acc = 1 if direct else 0
        dwbuf = dst if direct else torch.empty((N, K), dtype=torch.float32, device=x.device)
        btmp, stage_bias = None, False
        if bnames is not None:
            bdst = _grads_of(wb, bnames)
            if direct and bdst is not None:
                btmp = bdst
            else:
                stage_bias = True
                btmp = torch.empty((N,), dtype=torch.float32, device=x.device)
                if acc:
                    ops.zero_(btmp)
        call('sdmi_wgrad', _st(), a=_p(x), dy=_p(dy), dw=_p(dwbuf), dbias=_p(btmp),
             workspace=_p(ws), dtype=_DT[dt], M=M, N=N, K=K,
             lda=(Cin if is_conv else x.stride(-2)), ldy=ldy, B=B, H=H, W=W_,
             Cin=Cin, Ho=Ho, Wo=Wo, KH=kh, KW=kw, stride=stride, pad_t=pad[0], pad_l=pad[2],
             ups=int(ups), splits=splits, accumulate=acc)
        # non-direct destinations (channel-padded Cin, non-adjacent fused parameters) ACCUMULATE like
        # the direct path does: a parameter used several times per step (per-frame modules,
        # gradient accumulation over micro-batches) keeps every contribution
        if not direct:
            import ctypes
            g = wb.model.grad_arena()
            taps = kh * kw
            o, segs, keep = 0, [], []
            for nme in names:
                off, cnt = wb.model._offsets[nme]
                rows = wb.t[nme].shape[0]
                ci_true = cnt // (rows * taps)
                src = dwbuf[o:o + rows]
                if ci_true != Cin:                     # channel-padded Cin: [rows*taps, Cin(pad)] -> true Cin
                    src = ops.cast2d(src.reshape(rows * taps, Cin), torch.float32, cols=ci_true, ldd=ci_true)
                    keep.append(src)
                segs.append((_p(src), _p(g[off:]), cnt))
                o += rows
            Item = _lib.CSTRUCT['SdmiScatterItem']
            for c0 in range(0, len(segs), 32):         # one launch per 32 destinations
                chunk = segs[c0:c0 + 32]
                arr = (Item * len(chunk))()
                for a_, (sp, dp, cnt) in zip(arr, chunk):
                    a_.src, a_.dst, a_.count = sp, dp, cnt
                call('sdmi_scatter_add', _st(), items=ctypes.addressof(arr), n=len(chunk))
        if stage_bias:
            import ctypes
            g = wb.model.grad_arena()
            o, segs = 0, []
            for nme in (bnames if not isinstance(bnames, str) else (bnames,)):
                off, cnt = wb.model._offsets[nme]
                segs.append((_p(btmp[o:]), _p(g[off:]), cnt))
                o += cnt
            Item = _lib.CSTRUCT['SdmiScatterItem']
            for c0 in range(0, len(segs), 32):
                chunk = segs[c0:c0 + 32]
                arr = (Item * len(chunk))()
                for a_, (sp, dp, cnt) in zip(arr, chunk):
                    a_.src, a_.dst, a_.count = sp, dp, cnt
                call('sdmi_scatter_add', _st(), items=ctypes.addressof(arr), n=len(chunk))


class MultiLinearFn(torch.autograd.Function):
    """outs[i] = x @ W_i^T for several (fused) weights sharing ONE input -- the UNet's 16
    cross-attention K/V projections of the slot context (attention.py:187-189).  The data gradients
    are chained through the GEMM epilogue (dx_i = dy_i W_i + dx_{i-1}), so the shared input
    receives one tensor instead of 16 autograd accumulations."""

    @staticmethod
    def forward(ctx, x, anchor, wb, wname_list):
        ctx.save_for_backward(x)
        ctx.cfg = (wb, wname_list)
        ctx.set_materialize_grads(False)
        return tuple(ops.linear(x, wb.w(n, x.dtype)) for n in wname_list)

    @staticmethod
    def backward(ctx, *dys):
        (x,) = ctx.saved_tensors
        wb, wname_list = ctx.cfg
        geom = (0, 0, 1, (0, 0, 0, 0), False)
        dx = None
        for n, dy in zip(wname_list, dys):
            if dy is None:
                continue
            d, _, _ = GemmFn.core(wb, x, dy, n, None, geom, ctx.needs_input_grad[0], dx)
            dx = d if d is not None else dx
        return dx, None, None, None


class GroupNormFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, residual, anchor, wb, name, eps, act, n_alias=0, drop=None, rsink=None, f8=None):
        """rsink = (sink, offset): x = conv(.) + a per-image row vector whose gradient -- the pixel sums
        of this norm's dx -- goes into that slice of the shared row-vector gradient matrix (written
        by the backward kernel itself: no separate reduction launch).
        drop = (p, seed, seed_dev): dropout fused behind the activation (the mask is regenerated
        from the seed in backward).  n_alias aliases of x are returned next to y for x's other consumers (ResBlock skip,
        UNet skip-concat, SpatialTransformer residual): their gradients come back into this
        backward and are summed by the GroupNorm backward kernel (dextra0/1) -- no separate
        accumulation kernels.
        f8 = [scale]: the fp8 configuration -- the launch also writes the e4m3fn operand of the convolution behind
        the norm (appended to the list; KernGrad.gn hands it to GemmFn.forward on the output tensor)."""
        gamma, beta = wb.f(name + '.weight'), wb.f(name + '.bias')
        y, stats = ops.group_norm(x, gamma, beta, eps=eps, act=act, residual=residual,
                                  return_stats=True, drop=drop, fp8_scale=(f8[0] if f8 else None),
                                  fp8_also=(f8 if f8 else None))
        ctx.save_for_backward(x, stats, residual)
        ctx.cfg = (wb, name, act)
        ctx.drop = drop if (drop is not None and drop[0] > 0.0) else None
        ctx.rsink = rsink
        ctx.set_materialize_grads(False)
        if n_alias:
            return (y,) + tuple(x.view_as(x) for _ in range(n_alias))
        return y

    @staticmethod
    def backward(ctx, dy, *dal):
        x, stats, residual = ctx.saved_tensors
        wb, name, act = ctx.cfg
        extras = [d.contiguous() for d in dal if d is not None]
        assert len(extras) <= 2 and all(e.dtype == x.dtype for e in extras)
        if dy is None:                     # y unused: only the aliases carry gradient
            dx = extras[0]
            for e in extras[1:]:
                dx = AddFn.apply(dx, e)
            return dx, None, None, None, None, None, None, None, None, None, None
        dy = dy.contiguous()
        B, C = x.shape[0], x.shape[-1]
        HW = x.numel() // (B * C)
        G = 32
        nsplit = max(1, min(16, HW // 64))
        partial = torch.empty((B * nsplit * C * 2 + B * G * 2,), dtype=torch.float32,
                              device=x.device)
        dx = torch.empty_like(x)
        dres = torch.empty_like(x) if residual is not None else None
        dg = _grads_of(wb, name + '.weight')
        db = _grads_of(wb, name + '.bias')
        geo = dict(dtype=_DT[x.dtype], B=B, HW=HW, C=C, groups=G, nsplit=nsplit)
        defer = wb.defer_colsum
        if defer:          # the dbeta / dgamma folds of the whole step run in a few grouped launches
            wb.queue_colsum(partial, _lib.query('sdmi_groupnorm_bwd_entries', **geo), C, db, dg)
        rs_kw, rs_view = {}, None
        if ctx.rsink is not None and not extras:
            sink, off = ctx.rsink
            rs_view = sink.buffer(x.device)[:, off:off + C]
            if _lib.query('sdmi_groupnorm_bwd_fused', **geo):
                rs_kw = dict(dxsum=_p(rs_view), ld_dxsum=rs_view.stride(0))
        call('sdmi_groupnorm_bwd', _st(), x=_p(x), dy=_p(dy), dx=_p(dx), gamma=_p(wb.f(name + '.weight')),
             beta=_p(wb.f(name + '.bias')), stats=_p(stats), dgamma=_p(dg), dbeta=_p(db),
             partial=_p(partial), defer_colsum=int(defer), **geo, **rs_kw,
             act=_lib.ACT[act], residual=_p(residual), dresidual=_p(dres), accumulate=1,
             dextra0=(_p(extras[0]) if extras else 0), dextra1=(_p(extras[1]) if len(extras) > 1 else 0),
             **(dict(drop_p=float(ctx.drop[0]), drop_seed=int(ctx.drop[1]), drop_seed_dev=_p(ctx.drop[2]))
                if ctx.drop else {}))
        if rs_view is not None:
            if not rs_kw:              # two-pass geometry: the sums take their own launch after all
                call('sdmi_rowgroup_sum', _st(), x=_p(dx), out=_p(rs_view), dtype=_DT[dx.dtype], groups=B,
                     rows_per=HW, N=C, ldx=C, ldo=rs_view.stride(0))
            ctx.rsink[0].filled.add(ctx.rsink[1])
        _dbg(f'gn {name}', dy=dy, dx=dx, dres=dres)
        return dx, dres, None, None, None, None, None, None, None, None, None


class LayerNormFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, anchor, wb, name, n_alias=0):
        C = x.shape[-1]
        stats = torch.empty((x.numel() // C, 2), dtype=torch.float32, device=x.device)
        y = ops.layer_norm(x, wb.f(name + '.weight'), wb.f(name + '.bias'), stats=stats)
        ctx.save_for_backward(x, stats)
        ctx.cfg = (wb, name)
        ctx.set_materialize_grads(False)
        if n_alias:                        # alias of x for the residual branch (see GroupNormFn)
            return y, x.view_as(x)
        return y

    @staticmethod
    def backward(ctx, dy, dalias=None):
        x, stats = ctx.saved_tensors
        wb, name = ctx.cfg
        if dalias is not None:
            dalias = dalias.contiguous()
        if dy is None:
            return dalias, None, None, None, None
        return LayerNormFn.bwd_core(wb, name, x, stats, dy.contiguous(), dalias), None, None, None, None

    @staticmethod
    def bwd_core(wb, name, x, stats, dy, dalias=None):
        """dx of y = LayerNorm(x) (+ dalias: the gradient of x's other consumer, summed in the kernel); dgamma / dbeta
        into the gradient arena (deferred column sums)."""
        C = x.shape[-1]
        rows = x.numel() // C
        nblk = max(1, min(512, rows // 16))
        partial = torch.empty((nblk * C * 2,), dtype=torch.float32, device=x.device)
        dx = torch.empty_like(x)
        dg, db = _grads_of(wb, name + '.weight'), _grads_of(wb, name + '.bias')
        defer = wb.defer_colsum
        if defer:
            wb.queue_colsum(partial, nblk, C, dg, db)
        call('sdmi_layernorm_bwd', _st(), x=_p(x), dy=_p(dy), dx=_p(dx),
             gamma=_p(wb.f(name + '.weight')), stats=_p(stats), dgamma=_p(dg), dbeta=_p(db),
             partial=_p(partial), dtype=_DT[x.dtype], rows=rows, C=C, nblk=nblk, accumulate=1,
             dextra=_p(dalias), defer_colsum=int(defer))
        _dbg(f'ln {name}', dy=dy, dx=dx)
        return dx


class AttnFn(torch.autograd.Function):
    """self: qkv fused [B,S,3C]; cross: q [B,S,C] + kv [B,N,2C]."""

    @staticmethod
    def forward(ctx, q_or_qkv, kv, heads, head_dim=32):
        C = heads * head_dim
        if kv is None:
            q, k, v = q_or_qkv[..., :C], q_or_qkv[..., C:2 * C], q_or_qkv[..., 2 * C:]
        else:
            q, k, v = q_or_qkv, kv[..., :C], kv[..., C:]
        B, Sq = q.shape[0], q.shape[1]
        lse = torch.empty((B, heads, Sq), dtype=torch.float32, device=q.device)
        out = ops.attention(q, k, v, heads, lse=lse, head_dim=head_dim)
        ctx.save_for_backward(q_or_qkv, kv, out, lse)
        ctx.heads, ctx.hd = heads, head_dim
        return out

    @staticmethod
    def backward(ctx, dout):
        a, kv, out, lse = ctx.saved_tensors
        da, dkv = AttnFn.bwd_core(a, kv, out, lse, dout.contiguous(), ctx.heads, ctx.hd)
        return da, dkv, None, None

    @staticmethod
    def bwd_core(a, kv, out, lse, dout, heads, hd=32):
        """(d q|k|v, None) of self-attention over the fused a = q|k|v, or (dq, d k|v) of cross-attention."""
        C = heads * hd
        da = torch.empty_like(a)
        dkv = torch.empty_like(kv) if kv is not None else None
        if kv is None:
            q, k, v = a[..., :C], a[..., C:2 * C], a[..., 2 * C:]
            dq, dk, dv = da[..., :C], da[..., C:2 * C], da[..., 2 * C:]
        else:
            q, k, v = a, kv[..., :C], kv[..., C:]
            dq, dk, dv = da, dkv[..., :C], dkv[..., C:]
        B, Sq, Skv = q.shape[0], q.shape[1], k.shape[1]
        call('sdmi_attention_bwd', _st(), q=_p(q), k=_p(k), v=_p(v), out=_p(out), dout=_p(dout),
             lse=_p(lse), dq=_p(dq), dk=_p(dk), dv=_p(dv), dtype=_DT[q.dtype], B=B, heads=heads,
             Sq=Sq, Skv=Skv, ldq=q.stride(1), ldk=k.stride(1), ldv=v.stride(1), ldo=out.stride(1),
             scale=hd ** -0.5, head_dim=hd)
        _dbg(f'attn Sq={Sq} Skv={Skv}', dout=dout, da=da, dkv=dkv)
        return da, dkv


class LongAttnFn(torch.autograd.Function):
    """Self-attention over sequences beyond the LDS-resident kernels (ops.attention_long) with its
    backward as batched GEMMs per head:  dP = dO v^T,  dv = P^T dO,  dS = softmax'(P, dP),
    dq = dS k,  dk = dS^T q  (the score / probability matrices live in HBM).  qkv [B,S,3C]."""

    @staticmethod
    def forward(ctx, qkv, heads, head_dim):
        C = heads * head_dim
        assert qkv.shape[1] % ops.vec_of(qkv.dtype) == 0, 'sequence length must keep 16-byte rows'
        out, P = ops.attention_long(qkv[..., :C], qkv[..., C:2 * C], qkv[..., 2 * C:], heads, head_dim,
                                    keep_p=True)
        ctx.save_for_backward(qkv, P)
        ctx.heads, ctx.hd = heads, head_dim
        return out

    @staticmethod
    def backward(ctx, dout):
        qkv, P = ctx.saved_tensors
        heads, hd = ctx.heads, ctx.hd
        C = heads * hd
        B, S, _ = qkv.shape
        dout = dout.contiguous()
        dqkv = torch.empty_like(qkv)
        dp = torch.empty((B, S, S), dtype=qkv.dtype, device=qkv.device)
        for h in range(heads):
            q, k, v = (qkv[..., j * C + h * hd:j * C + (h + 1) * hd] for j in range(3))
            dq, dk, dv = (dqkv[..., j * C + h * hd:j * C + (h + 1) * hd] for j in range(3))
            do = dout[..., h * hd:(h + 1) * hd]
            p = P[h]
            ops.bmm_nt(do, v, dp)                                            # dP = dO v^T
            ops.bmm_nt(ops.transpose2d(p), ops.transpose2d(do), dv)          # dv = P^T dO
            ops.softmax_rows_bwd_(p, dp, scale=float(hd) ** -0.5)            # dp <- dS
            ops.bmm_nt(dp, ops.transpose2d(k), dq)                           # dq = dS k
            ops.bmm_nt(ops.transpose2d(dp), ops.transpose2d(q), dk)          # dk = dS^T q
        return dqkv, None, None


class GegluFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, h):
        ctx.save_for_backward(h)
        return ops.geglu(h)

    @staticmethod
    def backward(ctx, dy):
        (h,) = ctx.saved_tensors
        dy = dy.contiguous()
        dh = torch.empty_like(h)
        C = h.shape[-1] // 2
        call('sdmi_geglu_bwd', _st(), h=_p(h), dy=_p(dy), dh=_p(dh), dtype=_DT[h.dtype],
             rows=h.numel() // (2 * C), C=C)
        return dh


class GegluLinearFn(torch.autograd.Function):
    """y = GEGLU(x W^T + b) as ONE GEMM launch whose epilogue stores the pre-activation h as well
    (sdmi.h: geglu + out2); backward = geglu_bwd on the kept h, then the linear layer's gradients."""

    @staticmethod
    def forward(ctx, x, anchor, wb, wname, bname):
        w = wb.w(wname, x.dtype)
        h = torch.empty(x.shape[:-1] + (w.shape[0],), dtype=x.dtype, device=x.device)
        y = ops.linear(x, w, wb.b(bname), geglu=True, out2=h)
        ctx.save_for_backward(x, h)
        ctx.cfg = (wb, wname, bname)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, h = ctx.saved_tensors
        wb, wname, bname = ctx.cfg
        dy = dy.contiguous()
        dh = torch.empty_like(h)
        C = h.shape[-1] // 2
        call('sdmi_geglu_bwd', _st(), h=_p(h), dy=_p(dy), dh=_p(dh), dtype=_DT[h.dtype],
             rows=h.numel() // (2 * C), C=C)
        dx = GemmFn.core(wb, x, dh, wname, bname, (0, 0, 1, (0, 0, 0, 0), False), ctx.needs_input_grad[0])[0]
        return dx, None, None, None, None


def _gn_bwd_plain(wb, name, x, stats, dy, extras=()):
    """dx of y = GroupNorm(x) (no activation / residual / dropout) + the gradients of x's other consumers."""
    B, C = x.shape[0], x.shape[-1]
    HW = x.numel() // (B * C)
    G = 32
    nsplit = max(1, min(16, HW // 64))
    partial = torch.empty((B * nsplit * C * 2 + B * G * 2,), dtype=torch.float32, device=x.device)
    dx = torch.empty_like(x)
    dg, db = _grads_of(wb, name + '.weight'), _grads_of(wb, name + '.bias')
    geo = dict(dtype=_DT[x.dtype], B=B, HW=HW, C=C, groups=G, nsplit=nsplit)
    defer = wb.defer_colsum
    if defer:
        wb.queue_colsum(partial, _lib.query('sdmi_groupnorm_bwd_entries', **geo), C, db, dg)
    call('sdmi_groupnorm_bwd', _st(), x=_p(x), dy=_p(dy), dx=_p(dx), gamma=_p(wb.f(name + '.weight')),
         beta=_p(wb.f(name + '.bias')), stats=_p(stats), dgamma=_p(dg), dbeta=_p(db), partial=_p(partial),
         defer_colsum=int(defer), **geo, act=0, residual=0, dresidual=0, accumulate=1,
         dextra0=(_p(extras[0]) if extras else 0), dextra1=(_p(extras[1]) if len(extras) > 1 else 0))
    return dx


_ST_TRAIN = bool(policy.flag('ST_TRAIN'))        # fused training form of the SpatialTransformer block
# ... when its grid has at least this many workgroups (tests set it to 0).  At B = 64 the 16^2 level has 256 workgroups of
# 64 rows, the 8^2 level 128 of 32 rows: fusing the 8^2 blocks as well LOSES 0.1 - 0.2 ms per step (half the chip streams
# the block's weights while the other half idles; 26.41 / 26.41 ms with them fused, 26.33 / 26.22 ms per-layer, same box)
_ST_TRAIN_MIN_WGS = 200
_ST_TRAIN_BWD = bool(policy.flag('ST_TRAIN_BWD'))
_ST_WGRAD_GROUP = bool(policy.flag('ST_WGRAD_GROUP'))   # the block's weight gradients as two grouped launches    # ... and its backward data path (sdmi_st_train_bwd)


class StBlockFn(torch.autograd.Function):
    """A whole SpatialTransformer block of the denoiser in training (attention.py:297-308, 247-251, 182-206, 44-65):
    forward = TWO launches (sdmi.h: sdmi_st_train_fwd) that also store what the backward pass reads.
    x [B,H,W,C] bf16, kv [B,N,2C] bf16 (attn2.to_k | to_v of the slots) -> out [B,H,W,C]."""

    capture = None          # tests set a dict here: the backward pass leaves its intermediate gradients in it
    SAVED = ('hgn', 'gn_stats', 'tok', 'n1', 'st1', 'qkv', 'a1', 'lse1', 'x1', 'n2', 'st2', 'q2', 'a2', 'lse2', 'x2', 'n3',
             'st3', 'h', 'g', 'x3')

    @staticmethod
    def run_forward(wb, x, kv, n, heads, rows):
        """-> (out, {saved tensors}).  Plain launch (also what the tests drive)."""
        B, H, W, C = x.shape
        S = H * W
        dev, bf, f32 = x.device, torch.bfloat16, torch.float32
        e = lambda *sh: torch.empty(sh, dtype=bf, device=dev)
        f = lambda *sh: torch.empty(sh, dtype=f32, device=dev)
        sv = dict(hgn=e(B, S, C), gn_stats=f(B, 32, 2), tok=e(B, S, C), n1=e(B, S, C), st1=f(B * S, 2), qkv=e(B, S, 3 * C),
                  a1=e(B, S, C), lse1=f(B, heads, S), x1=e(B, S, C), n2=e(B, S, C), st2=f(B * S, 2), q2=e(B, S, C),
                  a2=e(B, S, C), lse2=f(B, heads, S), x2=e(B, S, C), n3=e(B, S, C), st3=f(B * S, 2), h=e(B, S, 8 * C),
                  g=e(B, S, 4 * C), x3=e(B, S, C))
        out = torch.empty_like(x)
        st = wb.st_train_streams(n)
        t = n + '.transformer_blocks.0'
        F_ = lambda k: _p(wb.f(k))
        flops = 2.0 * B * S * C * C * 20 + 4.0 * B * S * S * C + 4.0 * B * S * kv.shape[1] * C
        call('sdmi_st_train_fwd', _st(), x=_p(x), out=_p(out), **{k: _p(v) for k, v in sv.items()},
             kv2=_p(kv), ldkv=kv.stride(1), wstream_a=_p(st['wa']), wstream_b=_p(st['wb']),
             gn_gamma=F_(n + '.norm.weight'), gn_beta=F_(n + '.norm.bias'), b_in=F_(n + '.proj_in.bias'),
             ln1_g=F_(t + '.norm1.weight'), ln1_b=F_(t + '.norm1.bias'), b_o=F_(t + '.attn1.to_out.0.bias'),
             ln2_g=F_(t + '.norm2.weight'), ln2_b=F_(t + '.norm2.bias'), b_o2=F_(t + '.attn2.to_out.0.bias'),
             ln3_g=F_(t + '.norm3.weight'), ln3_b=F_(t + '.norm3.bias'), b_ff1=F_(t + '.ff.net.0.proj.bias'),
             b_ff2=F_(t + '.ff.net.2.bias'), b_po=F_(n + '.proj_out.bias'), B=B, S=S, C=C, slots=kv.shape[1], phase=0,
             rows=rows, gn_eps=1e-6, ln_eps=1e-5, attn_scale=32.0 ** -0.5,
             _meta=dict(flops=flops, bytes=2.0 * B * S * C * 23 + 2.0 * 36.5 * C * C))
        return out, sv

    @staticmethod
    def forward(ctx, x, kv, anchor, wb, n, heads, rows):
        out, sv = StBlockFn.run_forward(wb, x, kv, n, heads, rows)
        ctx.save_for_backward(x, kv, *[sv[k] for k in StBlockFn.SAVED])
        ctx.cfg = (wb, n, heads, rows)
        return out

    @staticmethod
    def backward(ctx, dout):
        x, kv = ctx.saved_tensors[:2]
        sv = dict(zip(StBlockFn.SAVED, ctx.saved_tensors[2:]))
        wb, n, heads, rows = ctx.cfg
        if _ST_TRAIN_BWD:
            dx, dkv = StBlockFn.backward_fused(wb, n, heads, x, kv, sv, dout.contiguous(), rows)
        else:
            dx, dkv = StBlockFn.backward_layers(wb, n, heads, x, kv, sv, dout.contiguous())
        return dx, dkv, None, None, None, None, None

    @staticmethod
    def backward_fused(wb, n, heads, x, kv, sv, dout, rows):
        """The block's backward data path as three fused launches (sdmi.h: sdmi_st_train_bwd) around the two attention
        backward kernels and the GroupNorm backward; the weight / bias gradients of the eight linear layers are
        stand-alone launches on side streams as their dY tensors appear; LayerNorm dgamma / dbeta from the per-workgroup
        column sums the fused launches leave (deferred grouped fold)."""
        t = n + '.transformer_blocks.0'
        lin = (0, 0, 1, (0, 0, 0, 0), False)
        B, S, C = sv['tok'].shape
        dev = x.device
        do = dout.view(B, S, C)
        nwg = B * S // rows
        st = wb.st_train_streams(n)
        e = lambda *sh: torch.empty(sh, dtype=torch.bfloat16, device=dev)
        part = lambda: torch.empty((nwg * C * 2,), dtype=torch.float32, device=dev)
        grp = []                      # (weight-gradient problems of this block for ONE grouped launch)

        def wgrad(a, dy, wn, bn):
            names = (wn,) if isinstance(wn, str) else tuple(wn)
            dst, bdst = _grads_of(wb, names), (_grads_of(wb, bn) if bn is not None else None)
            if not _ST_WGRAD_GROUP or dst is None or (bn is not None and bdst is None):
                GemmFn.core(wb, a, dy, wn, bn, lin, False)
                return
            M, K, N = a.numel() // a.shape[-1], a.shape[-1], dy.shape[-1]
            kw = dict(a=_p(a), dy=_p(dy), dw=_p(dst), dbias=_p(bdst), dtype=_lib.BF16, M=M, N=N, K=K, lda=K, ldy=N, B=M,
                      H=1, W=1, Cin=K, Ho=1, Wo=1, KH=1, KW=1, stride=1, pad_t=0, pad_l=0, ups=0, accumulate=1)
            tn = (N + 127) // 128
            grp.append((kw, tn * ((K + 127) // 128) + (tn if bn is not None else 0), (M + 63) // 64, a, dy))

        def flush_group():
            if not grp:
                return
            wb.flush_wgrad()                       # (nothing of another layer rides in this launch)
            for kw, tiles, steps, a, dy in grp:
                wb.flush_pending_fold((kw['dw'], kw['dbias']))
                wb._after_side_writers(kw['dw'], kw['dbias'])
                wb._wq.append((kw, tiles, steps))
                wb._wq_keep.extend((a, dy))
            side = wb.flush_wgrad()
            for kw, *_ in grp:
                for d in (kw['dw'], kw['dbias']):
                    if d and side is not None:
                        wb._side_dst[d] = side
            del grp[:]
            wb.ensure_join()
        F_ = lambda k: _p(wb.f(k))
        geo = dict(B=B, S=S, C=C, rows=rows)
        rf = 2.0 * B * S * C * C
        # ---- phase B1
        dx3, dh, dx2, da2, p3 = e(B, S, C), e(B, S, 8 * C), e(B, S, C), e(B, S, C), part()
        call('sdmi_st_train_bwd', _st(), phase=1, dout=_p(do), h=_p(sv['h']), x2=_p(sv['x2']), st3=_p(sv['st3']),
             ln3_g=F_(t + '.norm3.weight'), dx3=_p(dx3), dh=_p(dh), dx2=_p(dx2), da2=_p(da2), ln3_part=_p(p3),
             wstream_b1=_p(st['wb1']), **geo, _meta=dict(flops=14.0 * rf, bytes=2.0 * B * S * C * 22 + 28.0 * C * C))
        wb.queue_colsum(p3, nwg, C, _grads_of(wb, t + '.norm3.weight'), _grads_of(wb, t + '.norm3.bias'))
        wgrad(sv['x3'], do, n + '.proj_out.weight', n + '.proj_out.bias')
        wgrad(sv['g'], dx3, t + '.ff.net.2.weight', t + '.ff.net.2.bias')
        wgrad(sv['n3'], dh, t + '.ff.net.0.proj.weight', t + '.ff.net.0.proj.bias')
        wgrad(sv['a2'], dx2, t + '.attn2.to_out.0.weight', t + '.attn2.to_out.0.bias')
        flush_group()
        dq2, dkv = AttnFn.bwd_core(sv['q2'], kv, sv['a2'], sv['lse2'], da2, heads)
        # ---- phase B2
        dx1, da1, p2 = e(B, S, C), e(B, S, C), part()
        call('sdmi_st_train_bwd', _st(), phase=2, dq2=_p(dq2), x1=_p(sv['x1']), st2=_p(sv['st2']),
             ln2_g=F_(t + '.norm2.weight'), dx2=_p(dx2), dx1=_p(dx1), da1=_p(da1), ln2_part=_p(p2),
             wstream_b2=_p(st['wb2']), **geo, _meta=dict(flops=2.0 * rf, bytes=2.0 * B * S * C * 6 + 4.0 * C * C))
        wb.queue_colsum(p2, nwg, C, _grads_of(wb, t + '.norm2.weight'), _grads_of(wb, t + '.norm2.bias'))
        wgrad(sv['n2'], dq2, t + '.attn2.to_q.weight', None)
        wgrad(sv['a1'], dx1, t + '.attn1.to_out.0.weight', t + '.attn1.to_out.0.bias')
        dqkv, _ = AttnFn.bwd_core(sv['qkv'], None, sv['a1'], sv['lse1'], da1, heads)
        # ---- phase A
        dtok, dhgn, p1 = e(B, S, C), e(B, S, C), part()
        call('sdmi_st_train_bwd', _st(), phase=3, dqkv=_p(dqkv), tok=_p(sv['tok']), st1=_p(sv['st1']),
             ln1_g=F_(t + '.norm1.weight'), dx1=_p(dx1), dtok=_p(dtok), dhgn=_p(dhgn), ln1_part=_p(p1),
             wstream_a=_p(st['wba']), **geo, _meta=dict(flops=4.0 * rf, bytes=2.0 * B * S * C * 8 + 8.0 * C * C))
        wb.queue_colsum(p1, nwg, C, _grads_of(wb, t + '.norm1.weight'), _grads_of(wb, t + '.norm1.bias'))
        wgrad(sv['n1'], dqkv, (t + '.attn1.to_q.weight', t + '.attn1.to_k.weight', t + '.attn1.to_v.weight'), None)
        wgrad(sv['hgn'], dtok, n + '.proj_in.weight', n + '.proj_in.bias')
        flush_group()
        dx = _gn_bwd_plain(wb, n + '.norm', x.view(B, S, C), sv['gn_stats'], dhgn, (do,))
        if StBlockFn.capture is not None:        # (tests: the intermediate gradients of this form)
            StBlockFn.capture.update(dx3=dx3, dh=dh, dx2=dx2, da2=da2, dq2=dq2, dx1=dx1, da1=da1, dqkv=dqkv, dtok=dtok,
                                     dhgn=dhgn)
        return dx.view_as(x), dkv

    @staticmethod
    def backward_layers(wb, n, heads, x, kv, sv, dout):
        """The block's backward pass as per-layer launches (pair launches of the linear layers, LayerNorm / attention /
        GEGLU / GroupNorm backward kernels) on the tensors the fused forward stored."""
        t = n + '.transformer_blocks.0'
        lin = (0, 0, 1, (0, 0, 0, 0), False)
        B, S, C = sv['tok'].shape
        do = dout.view(B, S, C)
        core = lambda a, dy, wn, bn, dal=None: GemmFn.core(wb, a, dy, wn, bn, lin, True, dal)[0]
        dx3 = core(sv['x3'], do, n + '.proj_out.weight', n + '.proj_out.bias')
        dg = core(sv['g'], dx3, t + '.ff.net.2.weight', t + '.ff.net.2.bias')
        h = sv['h']
        dh = torch.empty_like(h)
        call('sdmi_geglu_bwd', _st(), h=_p(h), dy=_p(dg), dh=_p(dh), dtype=_DT[h.dtype], rows=B * S, C=4 * C)
        dn3 = core(sv['n3'], dh, t + '.ff.net.0.proj.weight', t + '.ff.net.0.proj.bias')
        dx2 = LayerNormFn.bwd_core(wb, t + '.norm3', sv['x2'], sv['st3'], dn3, dx3)
        da2 = core(sv['a2'], dx2, t + '.attn2.to_out.0.weight', t + '.attn2.to_out.0.bias')
        dq2, dkv = AttnFn.bwd_core(sv['q2'], kv, sv['a2'], sv['lse2'], da2, heads)
        dn2 = core(sv['n2'], dq2, t + '.attn2.to_q.weight', None)
        dx1 = LayerNormFn.bwd_core(wb, t + '.norm2', sv['x1'], sv['st2'], dn2, dx2)
        da1 = core(sv['a1'], dx1, t + '.attn1.to_out.0.weight', t + '.attn1.to_out.0.bias')
        dqkv, _ = AttnFn.bwd_core(sv['qkv'], None, sv['a1'], sv['lse1'], da1, heads)
        dn1 = core(sv['n1'], dqkv, (t + '.attn1.to_q.weight', t + '.attn1.to_k.weight', t + '.attn1.to_v.weight'), None)
        dtok = LayerNormFn.bwd_core(wb, t + '.norm1', sv['tok'], sv['st1'], dn1, dx1)
        dhgn = core(sv['hgn'], dtok, n + '.proj_in.weight', n + '.proj_in.bias')
        dx = _gn_bwd_plain(wb, n + '.norm', x.view(B, S, C), sv['gn_stats'], dhgn, (do,))
        if StBlockFn.capture is not None:
            StBlockFn.capture.update(dx3=dx3, dh=dh, dx2=dx2, da2=da2, dq2=dq2, dx1=dx1, da1=da1, dqkv=dqkv, dtok=dtok,
                                     dhgn=dhgn)
        return dx.view_as(x), dkv


class ActFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, kind):
        ctx.save_for_backward(x)
        ctx.kind = kind
        return ops.act(x, kind)

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        dy = dy.contiguous()
        dx = torch.empty_like(x)
        call('sdmi_act_bwd', _st(), x=_p(x), dy=_p(dy), dx=_p(dx), dtype=_DT[x.dtype],
             act=_lib.ACT[ctx.kind], n=x.numel())
        return dx, None


class CastFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, dtype):
        ctx.src = x.dtype
        return ops.act(x, None, dtype)

    @staticmethod
    def backward(ctx, dy):
        return ops.act(dy.contiguous(), None, ctx.src), None


class CastPadFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, dtype, cols, ldd):
        ctx.src, ctx.ld, ctx.cols = x.dtype, x.shape[-1], cols
        return ops.cast2d(x, dtype, cols=cols, ldd=ldd)

    @staticmethod
    def backward(ctx, dy):
        return ops.cast2d(dy.contiguous(), ctx.src, cols=ctx.cols, ldd=ctx.ld), None, None, None


class AddPosFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, pos):
        ctx.pshape = pos.shape
        return ops.add_pos(x, pos)

    @staticmethod
    def backward(ctx, dy):
        dy = dy.contiguous()
        dpos = None
        if ctx.needs_input_grad[1]:
            B = dy.shape[0]
            per = dy.numel() // B
            dpos = torch.empty(ctx.pshape, dtype=torch.float32, device=dy.device)
            call('sdmi_rowgroup_sum', _st(), x=_p(dy), out=_p(dpos), dtype=_DT[dy.dtype], groups=1,
                 rows_per=B, N=per, ldx=per)
        return dy, dpos


class ConcatFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b):
        ctx.ca, ctx.cb = a.shape[-1], b.shape[-1]
        return ops.concat_channels(a, b)

    @staticmethod
    def backward(ctx, dy):
        dy = dy.contiguous()
        da = torch.empty(dy.shape[:-1] + (ctx.ca,), dtype=dy.dtype, device=dy.device)
        db = torch.empty(dy.shape[:-1] + (ctx.cb,), dtype=dy.dtype, device=dy.device)
        call('sdmi_split_channels', _st(), y=_p(dy), a=_p(da), b=_p(db), dtype=_DT[dy.dtype],
             rows=dy.numel() // (ctx.ca + ctx.cb), Ca=ctx.ca, Cb=ctx.cb)
        return da, db


class SaAttendFn(torch.autograd.Function):
    """One Slot Attention iteration's streaming pass: (kv, q) -> (updates, attn)."""

    @staticmethod
    def forward(ctx, kv, q, eps, chain=False):
        """chain=True: also returns an alias of kv for the NEXT iteration; the gradient that comes
        back through it is the buffer this iteration's dk / dv are accumulated into."""
        B, M, D2 = kv.shape
        D = D2 // 2
        N = q.shape[1]
        attn = torch.empty((B, M, N), dtype=torch.float32, device=kv.device)
        upd = torch.empty((B, N, D), dtype=torch.float32, device=kv.device)
        den = torch.empty((B, N), dtype=torch.float32, device=kv.device)
        q = q.contiguous()
        ws = torch.empty((B * ((M + 63) // 64) * N * (D + 1),), dtype=torch.float32,
                         device=kv.device)
        call('sdmi_sa_attend_fwd', _st(), k=_p(kv), v=_p(kv[..., D:]), q=_p(q), attn=_p(attn),
             upd=_p(upd), den=_p(den), dtype=_DT[kv.dtype], B=B, M=M, N=N, D=D, ldkv=D2, eps=eps,
             scale=D ** -0.5, workspace=_p(ws))
        ctx.save_for_backward(kv, q, attn, upd, den)
        ctx.eps = eps
        ctx.mark_non_differentiable(attn)
        ctx.set_materialize_grads(False)
        if chain:
            return upd, attn, kv.view_as(kv)
        return upd, attn

    @staticmethod
    def backward(ctx, dupd, _dattn, dkv_next=None):
        kv, q, attn, upd, den = ctx.saved_tensors
        B, M, D2 = kv.shape
        D = D2 // 2
        N = q.shape[1]
        dupd = dupd.contiguous()
        dq = torch.empty_like(q)
        acc = dkv_next is not None and dkv_next.is_contiguous()
        dkv = dkv_next if acc else torch.empty_like(kv)
        ws = torch.empty((B * ((M + 63) // 64) * N * D,), dtype=torch.float32, device=kv.device)
        call('sdmi_sa_attend_bwd', _st(), k=_p(kv), v=_p(kv[..., D:]), q=_p(q), attn=_p(attn),
             upd=_p(upd), den=_p(den), dupd=_p(dupd), dq=_p(dq), dk=_p(dkv), dv=_p(dkv[..., D:]),
             dtype=_DT[kv.dtype], B=B, M=M, N=N, D=D, ldkv=D2, eps=ctx.eps, scale=D ** -0.5,
             workspace=_p(ws), accumulate=int(acc))
        if dkv_next is not None and not acc:
            dkv = AddFn.apply(dkv, dkv_next)
        _dbg('sa_attend', dupd=dupd, dkv=dkv, dq=dq)
        return dkv, dq, None, None


class GruGatesFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, gi, gh, h):
        R, D = h.shape
        out = torch.empty_like(h)
        call('sdmi_gru_gates', _st(), gi=_p(gi), gh=_p(gh), h=_p(h), hout=_p(out), R=R, D=D)
        ctx.save_for_backward(gi, gh, h)
        return out

    @staticmethod
    def backward(ctx, dout):
        gi, gh, h = ctx.saved_tensors
        R, D = h.shape
        dout = dout.contiguous()
        dgi, dgh, dh = torch.empty_like(gi), torch.empty_like(gh), torch.empty_like(h)
        call('sdmi_gru_gates_bwd', _st(), gi=_p(gi), gh=_p(gh), h=_p(h), dhout=_p(dout),
             dgi=_p(dgi), dgh=_p(dgh), dh=_p(dh), R=R, D=D)
        return dgi, dgh, dh


class AddFn(torch.autograd.Function):
    """y = a + b on our add kernel (both grads are dy)."""

    @staticmethod
    def forward(ctx, a, b):
        y = torch.empty_like(a)
        call('sdmi_add', _st(), x=_p(a), z=_p(b.contiguous()), y=_p(y), dtype=_DT[a.dtype], n=a.numel())
        return y

    @staticmethod
    def backward(ctx, dy):
        return dy, dy


class StackTimeFn(torch.autograd.Function):
    """out[:, t] = xs[t] -- torch.stack(xs, 1) of the per-frame slots / masks on the strided copy kernel (no framework
    kernel on the clip encode); backward hands every frame its slice of the gradient."""

    @staticmethod
    def forward(ctx, *xs):
        T, B = len(xs), xs[0].shape[0]
        n = xs[0].numel() // B
        out = torch.empty((B, T) + tuple(xs[0].shape[1:]), dtype=xs[0].dtype, device=xs[0].device)
        es, dt = out.element_size(), _DT[out.dtype]
        for t, x in enumerate(xs):
            x = x.contiguous()
            call('sdmi_cast2d', _st(), src=_p(x), dst=out.data_ptr() + t * n * es, src_dtype=dt, dst_dtype=dt, rows=B, cols=n,
                 lds=n, ldd=T * n, zpad=0)
        ctx.geo = (T, B, n, tuple(xs[0].shape))
        return out

    @staticmethod
    def backward(ctx, dout):
        T, B, n, shape = ctx.geo
        dout = dout.contiguous()
        es, dt = dout.element_size(), _DT[dout.dtype]
        outs = []
        for t in range(T):
            g = torch.empty(shape, dtype=dout.dtype, device=dout.device)
            call('sdmi_cast2d', _st(), src=dout.data_ptr() + t * n * es, dst=_p(g), src_dtype=dt, dst_dtype=dt, rows=B,
                 cols=n, lds=T * n, ldd=n, zpad=0)
            outs.append(g)
        return tuple(outs)


class VaeAttnFn(torch.autograd.Function):
    """Core of the VQ-VAE AttnBlock (vqvae/modules.py:130-150): one head of width C over S tokens,
    qkv [B,S,3C] -> o [B,S,C], as batched GEMMs (S = q k^T, P = softmax(S / sqrt(C)), o = P v) with
    batched transposes where a product contracts over a leading dimension."""

    @staticmethod
    def forward(ctx, qkv):
        B, S, C3 = qkv.shape
        C = C3 // 3
        q, k, v = qkv[..., :C], qkv[..., C:2 * C], qkv[..., 2 * C:]
        sc = torch.empty((B, S, S), dtype=qkv.dtype, device=qkv.device)
        ops.bmm_nt(q, k, sc)
        ops.softmax_rows_(sc, scale=float(C) ** -0.5)
        o = torch.empty((B, S, C), dtype=qkv.dtype, device=qkv.device)
        ops.bmm_nt(sc, ops.transpose2d(v), o)
        ctx.save_for_backward(qkv, sc)
        return o

    @staticmethod
    def backward(ctx, do):
        qkv, p = ctx.saved_tensors
        B, S, C3 = qkv.shape
        C = C3 // 3
        q, k, v = qkv[..., :C], qkv[..., C:2 * C], qkv[..., 2 * C:]
        do = do.contiguous()
        dqkv = torch.empty_like(qkv)
        dp = torch.empty_like(p)
        ops.bmm_nt(do, v, dp)                                   # dP = dO v^T
        ops.bmm_nt(ops.transpose2d(p), ops.transpose2d(do), dqkv[..., 2 * C:])     # dv = P^T dO
        ops.softmax_rows_bwd_(p, dp, scale=float(C) ** -0.5)    # dp <- dS
        ops.bmm_nt(dp, ops.transpose2d(k), dqkv[..., :C])       # dq = dS k
        ops.bmm_nt(ops.transpose2d(dp), ops.transpose2d(q), dqkv[..., C:2 * C])    # dk = dS^T q
        return dqkv


class VqFn(torch.autograd.Function):
    """VectorQuantizer2.forward (quantize.py:80-123): nearest code with the straight-through
    estimator and the legacy commitment loss.  z [..,4] fp32 NHWC (3 channels used) ->
    (zq, quant_loss, idx); the codebook gradient is accumulated straight into `dcode`."""

    @staticmethod
    def forward(ctx, z, anchor, codebook, dcode, beta):
        idx, zq = ops.vq_nearest(z, codebook)
        dim = codebook.shape[1]
        ql = (ops.mse(zq, z) * (z.shape[-1] / float(dim)) * (1.0 + beta)).reshape(())
        ctx.save_for_backward(z, zq, idx)
        ctx.dcode, ctx.beta, ctx.dim = dcode, beta, dim
        ctx.mark_non_differentiable(idx)
        return zq, ql, idx

    @staticmethod
    def backward(ctx, dzq, dql, _didx):
        z, zq, idx = ctx.saved_tensors
        dz = torch.empty_like(z)
        g = dql.reshape(1).float().contiguous() if dql is not None else \
            torch.zeros(1, dtype=torch.float32, device=z.device)
        dzq = dzq.contiguous() if dzq is not None else None
        call('sdmi_vq_bwd', _st(), z=_p(z), zq=_p(zq), dzq=_p(dzq), dz=_p(dz), dcode=_p(ctx.dcode),
             idx=_p(idx), g=_p(g), R=z.numel() // z.shape[-1], dim=ctx.dim, ldz=z.shape[-1],
             beta=ctx.beta)
        return dz, None, None, None, None


class MseFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred, target, scale, l1=False):
        val, dpred = ops.mse(pred, target, want_grad=True, gscale=scale, l1=l1, oscale=scale)
        ctx.save_for_backward(dpred)
        return val.reshape(())

    @staticmethod
    def backward(ctx, g):
        (dpred,) = ctx.saved_tensors
        out = torch.empty_like(dpred)                  # dpred * g, g a device scalar
        call('sdmi_scale_dev', _st(), x=_p(dpred), y=_p(out), s=_p(g.reshape(1).float().contiguous()),
             dtype=_DT[dpred.dtype], n=dpred.numel())
        return out, None, None, None


class DropoutFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, p, seed, seed_dev):
        ctx.p, ctx.seed, ctx.seed_dev = p, seed, seed_dev
        y = torch.empty_like(x)
        call('sdmi_dropout', _st(), x=_p(x), y=_p(y), dtype=_DT[x.dtype], n=x.numel(), p=p, seed=seed,
             seed_dev=_p(seed_dev))
        return y

    @staticmethod
    def backward(ctx, dy):
        dy = dy.contiguous()
        dx = torch.empty_like(dy)
        call('sdmi_dropout', _st(), x=_p(dy), y=_p(dx), dtype=_DT[dy.dtype], n=dy.numel(), p=ctx.p,
             seed=ctx.seed, seed_dev=_p(ctx.seed_dev))
        return dx, None, None, None



# ------------------------------------------------------------------------------------------
# plain-SA decoder pieces (img_based/models/slot_attention.py:343-364)
# ------------------------------------------------------------------------------------------
def _deconv_geom(x, w, k, stride, pad):
    B, H, W_, Cin = x.shape
    Cout = w.shape[1] // (k * k)
    Ho = (H - 1) * stride - 2 * pad + k + (stride - 1)
    Wo = (W_ - 1) * stride - 2 * pad + k + (stride - 1)
    return B, H, W_, Cin, Cout, Ho, Wo


def deconv_forward(wb, x, wname, bname, k, stride, pad, act):
    """ConvTranspose2d(k, stride, padding=pad, output_padding=stride-1) on NHWC.  The parameter
    [Cin, Cout, k, k] lies in the arena as [Cin][k][k][Cout] = the weight of the stride-`stride`
    convolution this layer is the data gradient of, so the forward pass is sdmi_igemm on the
    flipped/transposed operand with virtual zero insertion (exactly GemmFn's dgrad launch)."""
    dt = x.dtype
    w = wb.w(wname, dt)                           # [Cin][k*k*Cout]
    B, H, W_, Cin, Cout, Ho, Wo = _deconv_geom(x, w, k, stride, pad)
    wd = wb.wd(wname, dt, k, k, Cout)             # [Cout][k'][k'][Cin]
    y = torch.empty((B, Ho, Wo, Cout), dtype=dt, device=x.device)
    call('sdmi_igemm', _st(), a=_p(x), w=_p(wd), out=_p(y), bias=_p(wb.b(bname)), dtype=_DT[dt],
         out_dtype=_DT[dt], M=B * Ho * Wo, N=Cout, K=k * k * Cin, lda=Cin, ldw=k * k * Cin, ldc=Cout,
         B=B, H=H, W=W_, Cin=Cin, Ho=Ho, Wo=Wo, KH=k, KW=k, stride=1, pad_t=k - 1 - pad,
         pad_l=k - 1 - pad, ups=0, act=_lib.ACT[act], alpha=1.0, split_k=1, batch=1,
         zins=(stride if stride > 1 else 0))
    return y


class DeconvFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, anchor, wb, wname, bname, k, stride, pad, act):
        y = deconv_forward(wb, x, wname, bname, k, stride, pad, act)
        ctx.save_for_backward(x, y)
        ctx.cfg = (wb, wname, bname, k, stride, pad, act)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, y = ctx.saved_tensors
        wb, wname, bname, k, stride, pad, act = ctx.cfg
        dt = x.dtype
        dy = dy.contiguous()
        w = wb.w(wname, dt)
        B, H, W_, Cin, Cout, Ho, Wo = _deconv_geom(x, w, k, stride, pad)
        if act:                                   # relu: act'(z) from the output
            dz = torch.empty_like(dy)
            call('sdmi_act_bwd', _st(), x=_p(y), dy=_p(dy), dx=_p(dz), dtype=_DT[dt],
                 act=_lib.ACT[act], n=dy.numel())
        else:
            dz = dy
        # data gradient = the strided convolution itself
        dx = ops.conv2d(dz, w, None, kh=k, kw=k, stride=stride, pad=(pad, pad, pad, pad)) \
            if ctx.needs_input_grad[0] else None
        # weight gradient: roles swapped (A = dz gathered with the stride, "dY" = x)
        M, N, K = B * H * W_, Cin, k * k * Cout
        mt = 64 if dt == torch.bfloat16 else 32
        tiles = ((N + 127) // 128) * ((K + 127) // 128)
        splits = max(1, min((192 + tiles - 1) // tiles, M // (8 * mt), 512))
        ws = torch.empty((splits * (N * K + N),), dtype=torch.float32, device=x.device)
        call('sdmi_wgrad', _st(), a=_p(dz), dy=_p(x), dw=_p(_grads_of(wb, wname)), dbias=0,
             workspace=_p(ws), dtype=_DT[dt], M=M, N=N, K=K, lda=Cout, ldy=Cin, B=B, H=Ho, W=Wo,
             Cin=Cout, Ho=H, Wo=W_, KH=k, KW=k, stride=stride, pad_t=pad, pad_l=pad, ups=0,
             splits=splits, accumulate=1)
        # bias gradient: column sums of dz in 1024-row chunks, folded into the arena
        rows = B * Ho * Wo
        chunk = 1024 if rows % 1024 == 0 else rows
        part = torch.empty((rows // chunk, Cout), dtype=torch.float32, device=x.device)
        call('sdmi_rowgroup_sum', _st(), x=_p(dz), out=_p(part), dtype=_DT[dt], groups=rows // chunk,
             rows_per=chunk, N=Cout, ldx=Cout)
        gb = _grads_of(wb, bname)
        tot = torch.empty((1, Cout), dtype=torch.float32, device=x.device)
        call('sdmi_rowgroup_sum', _st(), x=_p(part), out=_p(tot), dtype=_lib.F32, groups=1,
             rows_per=rows // chunk, N=Cout, ldx=Cout)
        call('sdmi_add', _st(), x=_p(gb), z=_p(tot), y=_p(gb), dtype=_lib.F32, n=Cout)
        return dx, None, None, None, None, None, None, None, None


class BroadcastPosFn(torch.autograd.Function):
    """[G, C] slot vectors -> [G, R, C] (+ position embedding [R, C])."""

    @staticmethod
    def forward(ctx, x, pos, dtype):
        G, C = x.shape
        R = pos.shape[0]
        y = torch.empty((G, R, C), dtype=dtype, device=x.device)
        call('sdmi_broadcast_pos', _st(), x=_p(x.contiguous()), pos=_p(pos.contiguous()), y=_p(y),
             dtype=_DT[dtype], G=G, R=R, C=C)
        return y

    @staticmethod
    def backward(ctx, dy):
        dy = dy.contiguous()
        G, R, C = dy.shape
        dx = torch.empty((G, C), dtype=torch.float32, device=dy.device)
        call('sdmi_rowgroup_sum', _st(), x=_p(dy), out=_p(dx), dtype=_DT[dy.dtype], groups=G,
             rows_per=R, N=C, ldx=C)
        dpos = torch.empty((R, C), dtype=torch.float32, device=dy.device)
        call('sdmi_rowgroup_sum', _st(), x=_p(dy), out=_p(dpos), dtype=_DT[dy.dtype], groups=1,
             rows_per=G, N=R * C, ldx=R * C)
        return dx, dpos, None


class SaCombineFn(torch.autograd.Function):
    """o [B*N, H, W, ld] (rgb + alpha logit) -> recon [B, H, W, 4] fp32, masks [B, N, H*W] fp32."""

    @staticmethod
    def forward(ctx, o, B, N):
        BN, H, W_, ld = o.shape
        recon = torch.empty((B, H, W_, 4), dtype=torch.float32, device=o.device)
        masks = torch.empty((B, N, H * W_), dtype=torch.float32, device=o.device)
        call('sdmi_sa_combine', _st(), o=_p(o), recon=_p(recon), masks=_p(masks), dtype=_DT[o.dtype],
             B=B, N=N, HW=H * W_, ldo=ld)
        ctx.save_for_backward(o, masks)
        ctx.mark_non_differentiable(masks)
        ctx.set_materialize_grads(False)
        return recon, masks

    @staticmethod
    def backward(ctx, drecon, _dm):
        o, masks = ctx.saved_tensors
        B, N = masks.shape[0], masks.shape[1]
        dout = torch.empty_like(o)
        call('sdmi_sa_combine_bwd', _st(), o=_p(o), masks=_p(masks), drecon=_p(drecon.contiguous()),
             dout=_p(dout), dtype=_DT[o.dtype], B=B, N=N, HW=masks.shape[2], ldo=o.shape[-1])
        return dout, None, None


class NhwcToNchwFn(torch.autograd.Function):
    """fp32 [B, H, W, Cpad] -> [B, C, H, W] with gradient (user-facing recon_img)."""

    @staticmethod
    def forward(ctx, x, C):
        ctx.cpad = x.shape[-1]
        return ops.nhwc_to_nchw(x, C)

    @staticmethod
    def backward(ctx, dy):
        return ops.nchw_to_nhwc(dy.contiguous(), torch.float32, ctx.cpad), None


class KernGrad(Kern):
    """Training provider (autograd)."""
    training = True
    dropout_p = 0.0
    _drop_ctr = 0

    def __init__(self, wb):
        super().__init__(wb)
        # run seed and data-parallel rank: ranks must not share dropout masks
        rank = int(os.environ.get('RANK', 0))
        self.seed = ((int(getattr(wb.model, 'seed', 0)) * 1000003 + rank * 7919 + 1) & 0xffffffff)

    def _p_drop(self, site):
        attr = 'train_dropout' if site == 'unet' else 'pred_dropout'
        return float(getattr(self.wb.model, attr, 0.0))

    def linear_drop_res(self, x, wname, bname, res):
        if self._p_drop('pred') <= 0.0:
            return self.linear(x, wname, bname, residual=res)
        return AddFn.apply(res, self.dropout(self.linear(x, wname, bname), site='pred'))

    def dropout(self, x, site='unet'):
        p = self._p_drop(site)
        if p <= 0.0:
            return x
        # site counter restarts every step (`begin_step`), the per-step variation comes from the
        # device word `model.step_seed` so that a captured HIP graph draws new masks on replay
        self._drop_ctr += 1
        return DropoutFn.apply(x, p, (self.seed << 20) + self._drop_ctr,
                               getattr(self.wb.model, 'step_seed', None))

    def begin_step(self):
        """Start of a training step (eager or under graph capture): advance the device-side seed word
        (a captured add, so replays draw new masks too) and restart the per-step site counter."""
        m = self.wb.model
        dev = m.arena().device
        if getattr(m, 'step_seed', None) is None or m.step_seed.device != dev:
            m.step_seed = torch.zeros(1, dtype=torch.int64, device=dev)
        call('sdmi_counters_inc', _st(), seed=_p(m.step_seed))
        self._drop_ctr = 0

    def conv(self, x, wname, bname=None, *, kh=3, kw=3, stride=1, pad=(1, 1, 1, 1), ups=False,
             rowvec=None, residual=None, out_dtype=None, ldc=None):
        return GemmFn.apply(x, rowvec, residual, self.wb.anchor_for(wname), self.wb, wname, bname,
                            (kh, kw, stride, pad, ups), out_dtype, ldc)

    def linear(self, x, wnames, bnames=None, *, act=None, residual=None, out_dtype=None):
        y = GemmFn.apply(x, None, residual, self.wb.anchor_for(wnames), self.wb, wnames, bnames,
                         (0, 0, 1, (0, 0, 0, 0), False), out_dtype, None)
        return ActFn.apply(y, act) if act else y

    def gn(self, x, name, *, eps, act=None, residual=None, dropout=None, for_conv=None, rowsum_of=None):
        drop = None
        if dropout is not None:
            p = self._p_drop(dropout)
            if p > 0.0:          # fused behind the activation: same seed scheme as DropoutFn
                self._drop_ctr += 1
                drop = (p, (self.seed << 20) + self._drop_ctr, getattr(self.wb.model, 'step_seed', None))
        f8 = self._f8_for(x, for_conv)
        y = GroupNormFn.apply(x, residual, self.wb.anchor_for(name), self.wb, name, eps, act, 0, drop,
                              getattr(rowsum_of, '_sdmi_sink', None), f8)
        if f8 is not None and len(f8) > 1:
            y._sdmi_fp8 = f8[1]
        return y

    def _f8_for(self, x, for_conv):
        """[scale] when the norm's only reader is a convolution that multiplies e4m3fn operands (fp8 configuration):
        the norm writes that operand next to its bf16 output -- no quantisation launch in front of the GEMM."""
        return [FP8_ACT_SCALE] if (for_conv and self.fp8_ok(x, for_conv, 9, False)) else None

    def st_train(self, x, n, heads, kv):
        """The SpatialTransformer block `n` through the fused training kernels (StBlockFn), or None when the block
        does not qualify (the caller runs the per-layer launches): bf16, C = 256 / 384, 32 | tokens per image <= 256,
        <= 16 slots, a grid that fills a good part of the chip."""
        B, H, W, C = x.shape
        S = H * W
        if not (_ST_TRAIN and torch.is_tensor(kv) and x.dtype == torch.bfloat16 and kv.dtype == torch.bfloat16 and
                C in (256, 384) and heads * 32 == C and S % 32 == 0 and S <= 256 and kv.shape[1] <= 16 and
                kv.shape[-1] == 2 * C and kv.stride(-1) == 1 and x.is_contiguous()):
            return None
        rows = _ST_ROWS or (64 if (S % 64 == 0 and B * S // 64 >= 192) else 32)
        if S % rows or B * S // rows < _ST_TRAIN_MIN_WGS:
            return None
        return StBlockFn.apply(x, kv, self.wb.anchor_for(n), self.wb, n, heads, rows)

    def ff_out_proj(self, g, tres, xres, t, n):
        tok = self.linear(g, t + '.ff.net.2.weight', t + '.ff.net.2.bias', residual=tres)
        return self.linear(tok, n + '.proj_out.weight', n + '.proj_out.bias', residual=xres)

    def res_tail(self, h, n, skip):
        if (n + '.skip_connection.weight') in self.wb.t:
            skip = self.conv(skip, n + '.skip_connection.weight', n + '.skip_connection.bias', kh=1, kw=1,
                             pad=(0, 0, 0, 0))
        return self.conv(h, n + '.out_layers.3.weight', n + '.out_layers.3.bias', residual=skip)

    def rowvec_slices(self, rowvecs, bounds):
        if not rowvecs.requires_grad:
            return [rowvecs[:, off:off + c] for off, c in bounds]
        outs = RowvecSplitFn.apply(rowvecs, tuple(bounds))
        sink = outs[0].grad_fn.sink if hasattr(outs[0].grad_fn, 'sink') else None
        for v, (off, c) in zip(outs, bounds):
            v._sdmi_sink = (sink, off)
        return list(outs)

    def gn_fan(self, x, name, *, eps, act=None, residual=None, n_alias=1, for_conv=None):
        f8 = self._f8_for(x, for_conv)
        outs = GroupNormFn.apply(x, residual, self.wb.anchor_for(name), self.wb, name, eps, act, n_alias, None, None, f8)
        if f8 is not None and len(f8) > 1:
            outs[0]._sdmi_fp8 = f8[1]
        return outs

    def linear_fan(self, x, wnames, bnames=None):
        return GemmFn.apply(x, None, None, self.wb.anchor_for(wnames), self.wb, wnames, bnames,
                            (0, 0, 1, (0, 0, 0, 0), False), None, None, 1)

    def linear_multi(self, x, wname_list):
        wname_list = tuple(wname_list)
        return MultiLinearFn.apply(x, self.wb.anchor_for(wname_list[0]), self.wb, wname_list)

    def ln_fan(self, x, name):
        return LayerNormFn.apply(x, self.wb.anchor_for(name), self.wb, name, 1)

    def ln_linear_fan(self, x, ln_name, wnames, bnames=None, *, act=None, geglu=False, eps=1e-5):
        n, xres = self.ln_fan(x, ln_name)          # (training keeps the normalised rows for wgrad)
        if geglu and _GEGLU_FUSE and act is None and isinstance(wnames, str) and n.dtype == torch.bfloat16 \
                and n.shape[-1] % 64 == 0:
            return GegluLinearFn.apply(n, self.wb.anchor_for(wnames), self.wb, wnames, bnames), xres
        h = self.linear(n, wnames, bnames, act=act)
        return (self.geglu(h) if geglu else h), xres

    def conv_fan(self, x, wname, bname=None, *, kh=3, kw=3, stride=1, pad=(1, 1, 1, 1), ups=False,
                 rowvec=None, residual=None, out_dtype=None, ldc=None):
        return GemmFn.apply(x, rowvec, residual, self.wb.anchor_for(wname), self.wb, wname, bname,
                            (kh, kw, stride, pad, ups), out_dtype, ldc, 1)

    def deconv(self, x, wname, bname, *, k, stride, pad, act='relu'):
        return DeconvFn.apply(x, self.wb.anchor_for(wname), self.wb, wname, bname, k, stride, pad, act)

    def broadcast_pos(self, x, pos, dtype):
        return BroadcastPosFn.apply(x, pos, dtype)

    def sa_combine(self, o, B, N):
        return SaCombineFn.apply(o, B, N)

    def ln(self, x, name):
        return LayerNormFn.apply(x, self.wb.anchor_for(name), self.wb, name)

    def attn_self(self, qkv, heads, head_dim=32):
        fits = ops.ATTN_MFMA_MAX_KV if (qkv.dtype == torch.bfloat16 and head_dim == 32) else ops.ATTN_LDS_MAX_KV
        if qkv.shape[1] > fits:
            return LongAttnFn.apply(qkv, heads, head_dim)
        return AttnFn.apply(qkv, None, heads, head_dim)

    def attn_cross(self, q, kv, heads):
        return AttnFn.apply(q, kv, heads)

    def vae_attn_core(self, qkv):
        return VaeAttnFn.apply(qkv)

    def geglu(self, h):
        return GegluFn.apply(h)

    def add_pos(self, x, pos):
        return AddPosFn.apply(x, pos)

    def concat(self, a, b):
        return ConcatFn.apply(a, b)

    def cast(self, x, dtype):
        return x if x.dtype == dtype else CastFn.apply(x, dtype)

    def cast_pad(self, x, dtype, cols, ldd):
        return CastPadFn.apply(x, dtype, cols, ldd)

    def slot_attention(self, kv, init, name, iters, eps):
        """Unfused training form of SlotAttentionWMask.forward (sa_diffusion.py:40-68)."""
        B, M, D2 = kv.shape
        D = D2 // 2
        if init.dim() == 2:
            init = init.unsqueeze(0).expand(B, -1, -1)
        slots = init.contiguous()
        N = slots.shape[1]
        seg = None
        # every tensor with several consumers hands out aliases (ln_fan / linear_fan / the kv chain):
        # the consumers' gradients meet inside a backward kernel, not in accumulation kernels
        for it in range(iters):
            nq, s_alias = self.ln_fan(slots, f'{name}.project_q.0')
            q = self.linear(nq, f'{name}.project_q.1.weight')
            if it + 1 < iters:
                upd, attn, kv = SaAttendFn.apply(kv, q, eps, True)
            else:
                upd, attn = SaAttendFn.apply(kv, q, eps)
            seg = attn
            gi = self.linear(upd.reshape(B * N, D), f'{name}.gru.weight_ih', f'{name}.gru.bias_ih')
            gh, prev = self.linear_fan(s_alias.reshape(B * N, D), f'{name}.gru.weight_hh',
                                       f'{name}.gru.bias_hh')
            h = GruGatesFn.apply(gi, gh, prev)
            nh, h_alias = self.ln_fan(h, f'{name}.mlp.0')
            hid = self.linear(nh, f'{name}.mlp.1.weight', f'{name}.mlp.1.bias', act='relu')
            slots = self.linear(hid, f'{name}.mlp.3.weight', f'{name}.mlp.3.bias',
                                residual=h_alias).view(B, N, D)
        return slots, seg
