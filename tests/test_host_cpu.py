"""CPU-side checks: C-ABI exports, host logic (DPM plan, model registry/keys), loud failure
without a GPU.  No kernel is launched here."""
import pytest
import torch

from slotdiffusion_amd import _lib, dpm, module
from tests import common as C


def test_library_exports_every_declared_symbol():
    structs, funcs, enums = _lib.parse_header()
    assert len(funcs) >= 25 and 'sdmi_igemm' in funcs and 'sdmi_slot_attention' in funcs
    L = _lib.lib()                       # raises if the .so or any declared symbol is missing
    assert L.sdmi_version() == 100
    for name in funcs:
        assert hasattr(L, name), name


def test_integration_doc_struct_mirror_matches_header():
    """The hand-written ctypes mirror of SdmiGemmArgs in INTEGRATION.md lists the header's fields in order."""
    import ctypes
    import os
    from slotdiffusion_amd import _lib
    src = open(os.path.join(os.path.dirname(__file__), '..', 'INTEGRATION.md')).read()
    i = src.index('class SdmiGemmArgs(ctypes.Structure)')
    ns = {}
    exec('import ctypes\n' + src[i:src.index('# (a hand-written mirror', i)], ns)
    gen, doc = _lib.CSTRUCT['SdmiGemmArgs'], ns['SdmiGemmArgs']
    assert [f[0] for f in doc._fields_] == [f[0] for f in gen._fields_]
    assert ctypes.sizeof(doc) == ctypes.sizeof(gen)


def test_ops_refuse_cpu_tensors():
    from slotdiffusion_amd import ops
    x = torch.zeros(1, 4, 4, 8)
    with pytest.raises(_lib.SdmiError):
        ops.group_norm(x, torch.ones(8), torch.zeros(8), eps=1e-5, groups=2)


def test_invalid_arguments_are_rejected_without_launch():
    # validation happens before any HIP call, so this is safe on a GPU-less host
    with pytest.raises(_lib.SdmiError):
        _lib.call('sdmi_igemm', None, a=16, w=16, out=16, dtype=7, M=1, N=1, K=1)
    assert b'dtype' in _lib.lib().sdmi_last_error()
    # null operands, empty problems, misaligned / inconsistent geometry: every entry point returns an
    # error code (negative) with a message instead of launching
    bad = [
        ('sdmi_igemm', dict(a=0, w=16, out=16, dtype=_lib.BF16, M=1, N=1, K=8)),                 # null A
        ('sdmi_igemm', dict(a=16, w=16, out=16, dtype=_lib.BF16, out_dtype=_lib.BF16, M=0, N=8, K=8)),   # empty
        ('sdmi_igemm', dict(a=16, w=16, out=16, dtype=_lib.BF16, out_dtype=_lib.BF16, M=4, N=8, K=16,
                            KH=1, KW=1, Cin=8, B=4, Ho=1, Wo=1, lda=8, ldw=16)),                 # K != KH*KW*Cin
        ('sdmi_igemm', dict(a=18, w=16, out=16, dtype=_lib.BF16, out_dtype=_lib.BF16, M=4, N=8, K=8,
                            KH=1, KW=1, Cin=8, B=4, Ho=1, Wo=1, lda=8, ldw=8)),                  # unaligned A
        ('sdmi_igemm', dict(a=16, w=16, out=16, dtype=_lib.BF16, out_dtype=_lib.BF16, M=4, N=8, K=8,
                            KH=1, KW=1, Cin=8, B=4, Ho=1, Wo=1, lda=8, ldw=8, osy=2)),           # osy without oH/oW
        ('sdmi_wgrad', dict(a=0, dy=16, dw=16, dtype=_lib.BF16, M=8, N=8, K=8)),
        ('sdmi_groupnorm', dict(x=16, y=16, gamma=16, beta=16, stats=16, partial=16, dtype=_lib.BF16,
                                B=1, HW=4, C=12, groups=32, nsplit=1)),                          # C not a vector multiple
        ('sdmi_contingency', dict(gt=16, pred=16, counts=16, B=1, P=4, Kg=100, Kp=100)),         # table too large
        ('sdmi_transpose2d', dict(src=16, dst=16, dtype=9, Z=1, R=4, C=4, lds=4, ldd=4)),
        ('sdmi_vq_bwd', dict(z=16, zq=16, dz=16, dcode=16, idx=16, R=0, dim=3, ldz=4)),
    ]
    for fname, kw in bad:
        with pytest.raises(_lib.SdmiError):
            _lib.call(fname, None, **kw)
        assert len(_lib.lib().sdmi_last_error()) > 0, fname


def test_dpm_plan_matches_oracle_exactly():
    from oracle import slotdiff_oracle as O
    betas = torch.tensor(module.ddpm_schedule(1000, 'linear', 0.0015, 0.0195)['betas'],
                         dtype=torch.float32)
    plan = dpm.build_plan(betas, 20, 3)
    G = C.load_golden()
    assert torch.equal(plan['outer'], G['dpm_outer'])
    assert plan['orders'] == G['dpm_orders'].tolist()
    ns = O.NoiseScheduleDiscrete(betas)
    outer, orders = O.dpm_orders_and_timesteps(20, 3, 1.0, 1e-3)
    f = lambda v: float(v.reshape(-1)[0])
    nfe = 0
    for i, od in enumerate(orders):
        c = O.dpm_step_coeffs(ns, outer[i], outer[i + 1], od)
        st = plan['steps'][i]
        assert st['order'] == od and len(st['evals']) == od
        nfe += od
        assert st['evals'][0]['sigma'] == f(ns.std(c['s'])) and st['evals'][0]['alpha'] == f(ns.alpha(c['s']))
        assert st['evals'][0]['t_input'] == f((c['s'] - 1e-3) * 1000.)
        assert st['final']['c0'] == f(c['sigma_t'] / c['sigma_s'])
        assert st['final']['c1'] == f(-(c['alpha_t'] * c['phi_1']))
        if od >= 2:
            assert st['evals'][1]['t'] == f(c['s1'])
            assert st['to_s1']['c0'] == f(c['sigma_s1'] / c['sigma_s'])
            assert st['to_s1']['c1'] == f(-(c['alpha_s1'] * c['phi_11']))
        if od == 2:
            assert st['final']['c2'] == f(-(0.5 / c['r1']) * (c['alpha_t'] * c['phi_1']))
        if od == 3:
            assert st['evals'][2]['t'] == f(c['s2'])
            assert st['to_s2']['c2'] == f(c['r2'] / c['r1'] * (c['alpha_s2'] * c['phi_22']))
            assert st['final']['c2'] == f((1. / c['r2']) * (c['alpha_t'] * c['phi_2']))
    assert nfe == 20 and len(dpm.plan_t_inputs(plan)) == 20


def test_model_registry_and_checkpoint_keys():
    from slotdiffusion_amd.models import SADiffusion, build_model
    cfg = C.clevrtex_cfg()

    class P:
        model = 'SADiffusion'
    for k, v in cfg.items():
        setattr(P, k, v)
    m = build_model(P)
    assert isinstance(m, SADiffusion)
    ref = C.load_keys()['img_based/SADiffusion/clevrtex-7slot']
    sd = m.state_dict()
    assert [(k, list(v.shape)) for k, v in sd.items()] == [(k, s) for k, s, _ in ref['state']]
    frozen = {n for n, p in m.named_parameters() if not p.requires_grad}
    assert frozen == set(ref['frozen'])
    # optimizer grouping contract: 'dm_decoder' in the parameter name (method.py:307-313)
    assert any('dm_decoder' in n for n, _ in m.named_parameters())
    w = dict(m.named_parameters())['dm_decoder.model.diffusion_model.input_blocks.1.0.in_layers.2.weight']
    assert w.is_contiguous(memory_format=torch.channels_last)
    # round trip through a reference-layout (NCHW-contiguous) state dict
    sd2 = {k: v.clone().contiguous() for k, v in sd.items()}
    m.load_state_dict(sd2)
    assert dict(m.named_parameters())[
        'encoder.conv1.weight'].is_contiguous(memory_format=torch.channels_last)
    # schedule buffers equal the oracle's recipe
    assert torch.equal(m.dm_decoder.betas, C.oracle_weights(cfg)['dm_decoder.betas'])


def test_registry_modules_expose_the_plugin_api():
    """scripts/train.py:97-100 imports `slotdiffusion.<task>` and uses build_dataset / build_model /
    build_method; our registry modules keep that surface."""
    import importlib
    import inspect
    for task in ('img_based', 'video_based'):
        mod = importlib.import_module(f'slotdiffusion_amd.{task}')
        for fn in ('build_dataset', 'build_model', 'build_method'):
            assert callable(getattr(mod, fn))
        sig = inspect.signature(mod.build_method)
        assert any(p.kind == p.VAR_KEYWORD for p in sig.parameters.values())
    from slotdiffusion_amd.method import Method
    assert list(inspect.signature(Method.__init__).parameters)[1:] == [
        'model', 'datamodule', 'params', 'ckp_path', 'local_rank', 'use_ddp', 'use_fp16']
    from tests import common as C
    from slotdiffusion_amd.img_based import build_model
    m = build_model(C.make_params('SA'))
    assert type(m).__name__ == 'SA' and len(list(m.parameters())) == 83


def test_lr_schedule_closed_form():
    """FusedAdam.lr_scale = nerv's CosineAnnealingWarmupRestarts(first_cycle_steps=total, min_lr=0,
    warmup_steps=pct*total as a float) stepped after the optimiser: update k runs at the factor of
    k-1 completed steps (0 for the first update).  Closed form at it = 0, warmup-1, warmup, mid,
    total (img_based/method.py:275-285)."""
    import math
    from slotdiffusion_amd.optim import FusedAdam

    class Fake:
        def arena_ranges(self):
            return 8, 16, 16

        def arena(self):
            return torch.zeros(16)

    total, pct = 1000, 0.05
    opt = FusedAdam(Fake(), lr=1e-4, dec_lr=2e-4, total_steps=total, warmup_pct=pct)
    w = pct * total
    assert opt.warmup == w and isinstance(opt.warmup, float)
    assert opt.lr_scale(0) == 0.0
    assert abs(opt.lr_scale(int(w) - 1) - (w - 1) / w) < 1e-12
    assert abs(opt.lr_scale(int(w)) - 1.0) < 1e-12
    mid = int(w) + (total - int(w)) // 2
    ref = 0.5 * (1 + math.cos(math.pi * (mid - w) / (total - w)))
    assert abs(opt.lr_scale(mid) - ref) < 1e-12 and abs(ref - 0.5) < 2e-3
    assert abs(opt.lr_scale(total)) < 1e-12
    # fractional warm-up length (pct * total not an integer) stays a float
    opt2 = FusedAdam(Fake(), lr=1.0, total_steps=30, warmup_pct=0.05)
    assert opt2.warmup == 1.5 and abs(opt2.lr_scale(1) - 1 / 1.5) < 1e-12 and opt2.lr_scale(2) < 1.0
    # the device lr pair follows the schedule: before update 1 it is (0, 0)
    opt.set_lr_for_next_step()
    assert opt.lr_dev.tolist() == [0.0, 0.0]
    opt.step_count = int(w)
    opt.set_lr_for_next_step()
    assert abs(opt.lr_dev[0].item() - 1e-4) < 1e-10 and abs(opt.lr_dev[1].item() - 2e-4) < 1e-10
    # no schedule configured: constant factor
    assert FusedAdam(Fake(), lr=1.0).lr_scale(123) == 1.0
    # base-method floor (img_based/method.py:69-85: min_lr = lr / 100 for SA / SAVi / VQ-VAE stage 1):
    # update 1 runs at lr/100, the warm-up rises linearly from it, the cosine decays back to it
    opt3 = FusedAdam(Fake(), lr=4e-4, total_steps=total, warmup_pct=pct, min_lr_ratio=0.01)
    assert abs(opt3.lr_scale(0) - 0.01) < 1e-15
    assert abs(opt3.lr_scale(int(w) - 1) - (0.01 + 0.99 * (w - 1) / w)) < 1e-12
    assert abs(opt3.lr_scale(int(w)) - 1.0) < 1e-12
    assert abs(opt3.lr_scale(mid) - (0.01 + 0.99 * ref)) < 1e-12
    assert abs(opt3.lr_scale(total) - 0.01) < 1e-12
    opt3.set_lr_for_next_step()
    assert abs(opt3.lr_dev[0].item() - 4e-6) < 1e-12
    # a resumed optimiser must continue the same schedule
    sd = opt3.state_dict()
    assert sd['min_lr_ratio'] == 0.01
    # ... like the reference (the trainer loads the scheduler state and continues): a run configured with another
    # length / floor adopts the checkpoint's schedule, with a warning
    o4 = FusedAdam(Fake(), lr=4e-4, total_steps=total + 1, warmup_pct=pct, min_lr_ratio=0.0)
    with pytest.warns(UserWarning):
        o4.load_state_dict(sd)
    assert o4.total_steps == total and o4.min_lr_ratio == 0.01 and abs(o4.lr_scale(total) - 0.01) < 1e-12
    # a checkpoint from before the floor was recorded keeps THIS run's floor (the schedule triple is one unit: length
    # and warm-up come from the checkpoint, a missing key does not silently zero the floor)
    import warnings
    old_sd = {k: v for k, v in sd.items() if k != 'min_lr_ratio'}
    o5 = FusedAdam(Fake(), lr=4e-4, total_steps=total, warmup_pct=pct, min_lr_ratio=0.01)
    with warnings.catch_warnings():
        warnings.simplefilter('error')                  # identical schedule: nothing to report
        o5.load_state_dict(old_sd)
    assert o5.min_lr_ratio == 0.01 and o5.total_steps == total and o5.warmup == w
    # a checkpoint written WITHOUT a schedule leaves this run's whole triple alone (warm-up is not dropped) ...
    none_sd = dict(sd, total_steps=None, warmup=0.0)
    o6 = FusedAdam(Fake(), lr=4e-4, total_steps=total, warmup_pct=pct, min_lr_ratio=0.01)
    with pytest.warns(UserWarning):
        o6.load_state_dict(none_sd)
    assert (o6.total_steps, o6.warmup, o6.min_lr_ratio) == (total, w, 0.01)
    # ... and a run configured without one that loads a scheduled checkpoint adopts it, with a warning
    o7 = FusedAdam(Fake(), lr=4e-4)
    with pytest.warns(UserWarning):
        o7.load_state_dict(sd)
    assert (o7.total_steps, o7.warmup, o7.min_lr_ratio) == (total, w, 0.01)


def test_method_schedule_floor_follows_the_reference_method():
    """Method._configure_optimizers: SADiffusion anneals to 0 (img_based/method.py:277-283), SA / VQ-VAE
    stage 1 run on the base method's schedule with min_lr = lr / 100 (img_based/method.py:69-85)."""
    from tests import common as C
    from slotdiffusion_amd import method as M

    class FakeModel:
        def arena_ranges(self):
            return 8, 16, 16

        def arena(self):
            return torch.zeros(16)

    class DM:
        def __len__(self):
            return 10
    import slotdiffusion_amd.models as models
    for name, floor in (('SADiffusion', 0.0), ('SA', 0.01), ('VQVAE', 0.01)):
        P = C.make_params(name)
        fm = FakeModel()
        fm.__class__ = type('Fake' + name, (FakeModel, getattr(models, name)), {})   # isinstance dispatch only
        meth = M.Method.__new__(M.Method)
        meth.model, meth.datamodule, meth.params = fm, DM(), P
        opt = meth._configure_optimizers()
        assert opt.min_lr_ratio == floor, name
        assert abs(opt.lr_scale(0) - floor) < 1e-15


def test_runtime_knob_is_set_by_entry_points_not_by_import(monkeypatch):
    """Importing the package leaves the process environment alone; configure_runtime() (bench.py,
    Method, GraphedTrainStep) applies the HIP-graph queue pool setting unless the user exported one."""
    import importlib
    import os
    import slotdiffusion_amd
    monkeypatch.delenv('DEBUG_HIP_FORCE_GRAPH_QUEUES', raising=False)
    importlib.reload(slotdiffusion_amd)
    assert 'DEBUG_HIP_FORCE_GRAPH_QUEUES' not in os.environ
    assert slotdiffusion_amd.configure_runtime() == '2'
    monkeypatch.setenv('DEBUG_HIP_FORCE_GRAPH_QUEUES', '4')
    assert slotdiffusion_amd.configure_runtime() == '4'


def test_policy_table_is_the_only_kernel_selection_switchboard(monkeypatch):
    """Every switch has a default and a description; SDMI_<NAME> overrides it; no other module of the package reads an
    SDMI_* environment name for kernel selection (the few remaining names are run configuration, not kernel choice)."""
    import os
    import re
    from slotdiffusion_amd import policy
    assert all(isinstance(v[0], int) and isinstance(v[1], str) and v[1] for v in policy.SWITCHES.values())
    assert policy.flag('ST_TRAIN') == policy.SWITCHES['ST_TRAIN'][0]
    monkeypatch.setenv('SDMI_ST_TRAIN', '0')
    assert policy.flag('ST_TRAIN') == 0
    allowed = {'SDMI_LIBPATH', 'SDMI_CALL_LOG', 'SDMI_GRAPH', 'SDMI_DTYPE', 'SDMI_GRAD_BF16', 'SDMI_BUILD_FORCE'}
    root = os.path.dirname(policy.__file__)
    for dirpath, _, files in os.walk(root):
        for f in files:
            if f.endswith('.py') and f != 'policy.py':
                names = set(re.findall(r"environ\.get\('(SDMI_[A-Z0-9_]+)'", open(os.path.join(dirpath, f)).read()))
                assert names <= allowed, (f, names - allowed)
    assert len(policy.SWITCHES) + len(allowed) <= 25


def test_fused_training_stream_geometry_matches_the_kernels():
    """Unit counts of the weight streams (kern.st_train_units) = what csrc/st_train.hip consumes per wave (StTrGeom /
    StTrBwdGeom: UA, UB, UB1, UB2), every unit inside its matrix, forward units row-major and backward units transposed."""
    from slotdiffusion_amd import kern
    for C in (256, 384):
        NSL, KT, NHC = C // 128, C // 64, C // 32
        u = kern.st_train_units(C)
        want = dict(a=4 * KT * NSL, b=4 * KT * NSL + NHC * (2 * KT + 2 * NSL), b1=2 * KT * NSL + NHC * (KT + 4 * NSL),
                    b2=2 * KT * NSL, ba=4 * KT * NSL)
        shapes = {'in': (C, C), 'q': (C, C), 'k': (C, C), 'v': (C, C), 'o': (C, C), 'q2': (C, C), 'o2': (C, C),
                  'ff1': (8 * C, C), 'ff2': (C, 4 * C), 'po': (C, C)}
        for k, n in want.items():
            assert len(u[k]) == 8 and all(len(w) == n for w in u[k]), (C, k)
            for w in u[k]:
                for key, r0, k0, tr in w:
                    rows, cols = shapes[key]
                    assert tr == (k in ('b1', 'b2', 'ba'))
                    if tr:      # unit rows walk the matrix's columns
                        assert r0 + 16 <= cols and k0 + 64 <= rows and r0 % 16 == 0 and k0 % 64 == 0
                    else:
                        assert r0 + 16 <= rows and k0 + 64 <= cols and r0 % 16 == 0 and k0 % 64 == 0
        # every weight element is streamed exactly once per direction
        for keys in (('a', 'b'), ('b1', 'b2', 'ba')):
            seen = {}
            for k in keys:
                for w in u[k]:
                    for key, r0, k0, tr in w:
                        seen[(key, r0, k0)] = seen.get((key, r0, k0), 0) + 1
            assert set(seen.values()) == {1}
            assert sum(16 * 64 for _ in seen) == sum(r * c for r, c in shapes.values())
