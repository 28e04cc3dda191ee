"""Build libsdmi.so (hipcc, gfx950) in-tree.  `python -m slotdiffusion_amd.csrc.build`."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, '..', 'libsdmi.so')
SOURCES = ['api.cpp', 'igemm.hip', 'igemm_halo.hip', 'wgrad.hip', 'bwd_pair.hip', 'norm.hip', 'norm_bwd.hip', 'attention.hip',
           'attention_bwd.hip', 'st_fused.hip', 'st_train.hip', 'cross_fold.hip', 'slot_attn.hip', 'slot_attn_train.hip', 'bwd_misc.hip',
           'elementwise.hip', 'vq.hip', 'metrics.hip', 'vae_train.hip']
EXTRA = {'vq.hip': ['-ffp-contract=off'], 'elementwise.hip': ['-ffp-contract=off'],
         # no packed-fp32 VALU ops in the fused training kernels (st_train.hip's build note: not repeatable on MI355X)
         'st_train.hip': ['-fno-slp-vectorize']}
HIPCC = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-Wall', '-Wno-unused-function']


def _obj(src):
    return os.path.join(HERE, '_build', os.path.splitext(src)[0] + '.o')


def _headers(src):
    """csrc headers a source includes, transitively (`#include "x.h"` lines), sorted."""
    import re
    seen, todo = set(), [src]
    while todo:
        f = todo.pop()
        for h in re.findall(r'^\s*#\s*include\s+"([^"/]+\.h)"', open(os.path.join(HERE, f)).read(), flags=re.M):
            if h not in seen and os.path.exists(os.path.join(HERE, h)):
                seen.add(h)
                todo.append(h)
    return sorted(os.path.join(HERE, h) for h in seen)


def _digest(src):
    """sha256 over the source, the csrc headers it includes (transitively), the C ABI header and the compile flags: an
    object file is reused only for byte-identical inputs (mtimes do not survive a snapshot copy)."""
    import hashlib
    h = hashlib.sha256()
    deps = [os.path.join(HERE, src), os.path.join(HERE, '..', '..', 'include', 'sdmi.h')] + _headers(src)
    for d in deps:
        h.update(d.encode())
        h.update(open(d, 'rb').read())
    h.update(' '.join(FLAGS + EXTRA.get(src, [])).encode())
    return h.hexdigest()


def _stale(src, obj):
    stamp = obj + '.sha256'
    return not (os.path.exists(obj) and os.path.exists(stamp) and open(stamp).read().strip() == _digest(src))


LOG = []        # (source, 'compiled' | 'reused') of the last build() call


def _compile(src):
    obj = _obj(src)
    if not _stale(src, obj):
        LOG.append((src, 'reused'))
        return obj
    cmd = [HIPCC] + FLAGS + EXTRA.get(src, []) + ['-x', 'hip', '-c', os.path.join(HERE, src), '-o', obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f'hipcc failed for {src}:\n{r.stderr[-6000:]}')
    if r.stderr.strip():
        sys.stderr.write(r.stderr[-3000:])
    with open(obj + '.sha256', 'w') as f:
        f.write(_digest(src))
    LOG.append((src, 'compiled'))
    return obj


def build(force=False):
    """Compile what changed (content hashes, not mtimes), link, and write _build/BUILD_INFO.json: which
    sources this call compiled and which objects it reused, so a build record can be read from the tree."""
    import json
    import time
    os.makedirs(os.path.join(HERE, '_build'), exist_ok=True)
    force = force or os.environ.get('SDMI_BUILD_FORCE', '0') != '0'
    if force:
        for s in SOURCES:
            for f in (_obj(s), _obj(s) + '.sha256'):
                if os.path.exists(f):
                    os.remove(f)
    del LOG[:]
    with ThreadPoolExecutor(max_workers=8) as ex:
        objs = list(ex.map(_compile, SOURCES))
    relinked = False
    if not os.path.exists(OUT) or any(what == 'compiled' for _, what in LOG) or \
            any(os.path.getmtime(o) > os.path.getmtime(OUT) for o in objs):
        cmd = [HIPCC, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', OUT] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f'link failed:\n{r.stderr[-4000:]}')
        relinked = True
    info = os.path.join(HERE, '_build', 'BUILD_INFO.json')
    hist = []
    try:
        hist = json.load(open(info)).get('history', [])
    except (OSError, ValueError):
        pass
    hist.append({'time': time.strftime('%Y-%m-%d %H:%M:%S'), 'relinked': relinked,
                 'compiled': sorted(s for s, what in LOG if what == 'compiled'),
                 'reused': sorted(s for s, what in LOG if what == 'reused')})
    with open(info, 'w') as f:
        json.dump({'hipcc': HIPCC, 'flags': FLAGS, 'history': hist[-20:]}, f, indent=1)
    return os.path.abspath(OUT)


if __name__ == '__main__':
    print(build(force='--force' in sys.argv))
