"""ctypes binding of libsdmi.so, generated from include/sdmi.h at import time.

The argument structs and the exported function list are parsed from the C header, so the Python
side can never drift from the ABI.  There is NO fallback: if the shared library is missing or a
declared symbol cannot be resolved, importing the compute path raises.
"""
import ctypes
import os
import re

_HERE = os.path.dirname(os.path.abspath(__file__))
HEADER = os.path.join(_HERE, '..', 'include', 'sdmi.h')
LIBPATH = os.environ.get('SDMI_LIBPATH') or os.path.join(_HERE, 'libsdmi.so')   # override: kernel experiments

_CT = {
    'int': ctypes.c_int, 'float': ctypes.c_float, 'long long': ctypes.c_longlong,
    'double': ctypes.c_double, 'unsigned': ctypes.c_uint,
}


def _strip_comments(src):
    return re.sub(r'/\*.*?\*/', '', src, flags=re.S)


def parse_header(path=HEADER):
    """-> (structs {name: [(field, ctype)]}, functions {name: arg struct name or None}, enums)."""
    src = _strip_comments(open(path).read())
    structs = {}
    for body, name in re.findall(r'typedef\s+struct\s*\{(.*?)\}\s*(\w+)\s*;', src, flags=re.S):
        fields = []
        for decl in body.split(';'):
            decl = ' '.join(decl.split())
            if not decl:
                continue
            m = re.match(r'^(const\s+)?(void|float|double|int|long long|unsigned)\s*(.*)$', decl)
            assert m, f'cannot parse field declaration {decl!r} in {name}'
            base, rest = m.group(2), m.group(3)
            for item in rest.split(','):
                item = item.strip()
                is_ptr = item.startswith('*')
                fname = item.lstrip('* ').strip()
                if is_ptr or base == 'void':
                    fields.append((fname, ctypes.c_void_p))
                else:
                    fields.append((fname, _CT[base]))
        structs[name] = fields
    funcs = {}
    for ret, fname, args in re.findall(r'\b(int|const char\*)\s+(sdmi_\w+)\s*\(([^)]*)\)\s*;', src):
        m = re.match(r'\s*const\s+(\w+)\s*\*', args)
        funcs[fname] = m.group(1) if m else None
    enums = {}
    for body in re.findall(r'enum\s*\{(.*?)\}\s*;', src, flags=re.S):
        for item in body.split(','):
            if '=' in item:
                k, v = item.split('=')
                enums[k.strip()] = int(v.strip())
    return structs, funcs, enums


STRUCTS, FUNCS, ENUMS = parse_header()
F32, BF16, FP8 = ENUMS['SDMI_F32'], ENUMS['SDMI_BF16'], ENUMS['SDMI_FP8']
ACT = {None: 0, 'none': 0, 'relu': ENUMS['SDMI_ACT_RELU'], 'silu': ENUMS['SDMI_ACT_SILU'],
       'gelu': ENUMS['SDMI_ACT_GELU']}


def _make_struct(name, fields):
    return type(name, (ctypes.Structure,), {'_fields_': fields})


CSTRUCT = {n: _make_struct(n, f) for n, f in STRUCTS.items()}

_lib = None


class SdmiError(RuntimeError):
    pass


def lib():
    """Load libsdmi.so (built by slotdiffusion_amd.csrc.build); fail loudly if absent."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIBPATH):
            raise SdmiError(
                f'{LIBPATH} not found: the HIP extension is required (no fallback path). '
                'Build it with `python -m slotdiffusion_amd.csrc.build`.')
        # torch first: libsdmi.so must bind to the libamdhip64 torch ships and initialises (loaded
        # the other way round the system runtime under /opt/rocm is picked up and kernel launches
        # fail with "no ROCm-capable device")
        import torch  # noqa: F401
        L = ctypes.CDLL(LIBPATH)
        for fname, sname in FUNCS.items():
            fn = getattr(L, fname)          # AttributeError if a declared symbol is missing
            if fname == 'sdmi_last_error':
                fn.restype = ctypes.c_char_p
                fn.argtypes = []
            elif sname is None:
                fn.restype = ctypes.c_int
                fn.argtypes = []
            else:
                fn.restype = ctypes.c_int
                fn.argtypes = [ctypes.POINTER(CSTRUCT[sname]), ctypes.c_void_p]
        _lib = L
    return _lib


class KernelTimer:
    """Optional per-call HIP-event timing (bench.py's roofline leg).  Events are recorded on the
    stream the kernels are launched on (torch's current stream), so the durations are those of
    the launches themselves.  Enable with `with KernelTimer() as kt: ...; kt.summary()`."""
    active = None

    def __init__(self):
        self.records = []

    def __enter__(self):
        KernelTimer.active = self
        return self

    def __exit__(self, *a):
        KernelTimer.active = None

    def summary(self):
        import torch
        torch.cuda.synchronize()
        out = {}
        for name, e0, e1, meta in self.records:
            d = out.setdefault(name, dict(calls=0, ms=0.0, flops=0.0, bytes=0.0))
            d['calls'] += 1
            d['ms'] += e0.elapsed_time(e1)
            for k, v in meta.items():               # flops / bytes (+ entry-specific splits of them)
                if isinstance(v, (int, float)):
                    d[k] = d.get(k, 0.0) + v
        return out


def _meta(fname, kw):
    """Algorithmic work of one launch (bench.py's roofline legs): flops of the GEMM-shaped entry
    points; algorithmic HBM bytes (every operand read once, every result written once)."""
    elt = {BF16: 2, FP8: 1}.get(kw.get('dtype', F32), 4)
    if fname == 'sdmi_igemm':
        b = max(1, kw.get('batch', 1))
        oelt = 2 if kw.get('out_dtype', F32) == BF16 else 4
        src = kw['B'] * kw['H'] * kw['W'] * kw['Cin']          # the image, not its im2col expansion
        wts = kw['N'] * kw['K'] * (b if kw.get('sw', 0) else 1)
        out = kw['M'] * kw['N'] * (2 if kw.get('residual', 0) else 1)
        return dict(flops=2.0 * kw['M'] * kw['N'] * kw['K'] * b,
                    bytes=float(b * src * elt + wts * elt + b * out * oelt), fp8=int(elt == 1))
    if fname == 'sdmi_wgrad':          # x and dY read once, dW written once (fp32)
        return dict(flops=2.0 * kw['M'] * kw['N'] * kw['K'],
                    bytes=float((kw['B'] * kw['H'] * kw['W'] * kw['Cin'] + kw['M'] * kw['N']) * elt + kw['N'] * kw['K'] * 4))
    if fname == 'sdmi_groupnorm':          # x in, y out (+ residual in)
        n = kw['B'] * kw['HW'] * kw['C']
        return dict(bytes=float(n * elt * (3 if kw.get('residual', 0) else 2)))
    if fname == 'sdmi_groupnorm_bwd':      # x, dy in; dx out (+ residual-branch gradient, extras)
        n = kw['B'] * kw['HW'] * kw['C']
        k = 3 + (1 if kw.get('residual', 0) else 0) + (1 if kw.get('dresidual', 0) else 0) + \
            (1 if kw.get('dextra0', 0) else 0) + (1 if kw.get('dextra1', 0) else 0)
        return dict(bytes=float(n * elt * k))
    return {}


# ops.py: finishes split-K launches whose second stage was left to the next kernel before any OTHER kernel runs
pre_call = None


# SDMI_CALL_LOG=path (measurement only; bench.py's PMC child passes): one line per C-ABI call -- entry point, algorithmic
# bytes, flops -- in launch order, so that the profiler's per-dispatch counters can be set against the algorithmic
# bytes of the launch that produced them
_CALL_LOG = None


def _log_call(fname, kw, meta):
    global _CALL_LOG
    if _CALL_LOG is None:
        path = os.environ.get('SDMI_CALL_LOG')
        _CALL_LOG = open(path, 'a') if path else False
    if _CALL_LOG:
        m = meta if meta is not None else _meta(fname, kw)
        _CALL_LOG.write(f"{fname}\t{m.get('bytes', 0.0):.0f}\t{m.get('flops', 0.0):.0f}\n")
        _CALL_LOG.flush()


def call(fname, stream, **kw):
    """Invoke `fname` with its argument struct filled from keyword args (missing fields = 0).
    `_meta` (optional dict: flops / bytes of the launch) only feeds KernelTimer."""
    if pre_call is not None:
        pre_call()
    meta = kw.pop('_meta', None)
    if _CALL_LOG is not False:
        _log_call(fname, kw, meta)
    kt = KernelTimer.active
    if kt is not None:
        import torch
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        _call(fname, stream, **kw)
        e1.record()
        kt.records.append((fname, e0, e1, meta if meta is not None else _meta(fname, kw)))
        return
    _call(fname, stream, **kw)


def query(fname, **kw):
    """Call a no-launch query entry point and return its integer result."""
    L = lib()
    sname = FUNCS[fname]
    args = CSTRUCT[sname]()
    for k, v in kw.items():
        setattr(args, k, v)
    return int(getattr(L, fname)(ctypes.byref(args), None))


def _call(fname, stream, **kw):
    L = lib()
    sname = FUNCS[fname]
    args = CSTRUCT[sname]()
    names = {f for f, _ in STRUCTS[sname]}
    for k, v in kw.items():
        if k not in names:
            raise KeyError(f'{sname} has no field {k}')
        setattr(args, k, v)
    rc = getattr(L, fname)(ctypes.byref(args), stream)
    if rc != 0:
        raise SdmiError(f'{fname} failed ({rc}): {L.sdmi_last_error().decode()}')
