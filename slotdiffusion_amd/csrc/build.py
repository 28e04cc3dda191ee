"""Build libsdmi.so (hipcc, gfx950) in-tree.  `python -m slotdiffusion_amd.csrc.build`."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, '..', 'libsdmi.so')
SOURCES = ['api.cpp', 'igemm.hip', 'wgrad.hip', 'bwd_pair.hip', 'norm.hip', 'norm_bwd.hip', 'attention.hip',
           'attention_bwd.hip', 'slot_attn.hip', 'slot_attn_train.hip', 'bwd_misc.hip',
           'elementwise.hip', 'vq.hip', 'metrics.hip', 'vae_train.hip']
EXTRA = {'vq.hip': ['-ffp-contract=off'], 'elementwise.hip': ['-ffp-contract=off']}
HIPCC = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-Wall', '-Wno-unused-function']


def _obj(src):
    return os.path.join(HERE, '_build', os.path.splitext(src)[0] + '.o')


def _stale(src, obj):
    if not os.path.exists(obj):
        return True
    deps = [os.path.join(HERE, src), os.path.join(HERE, '..', '..', 'include', 'sdmi.h')] + \
        [os.path.join(HERE, h) for h in os.listdir(HERE) if h.endswith('.h')]
    return any(os.path.getmtime(d) > os.path.getmtime(obj) for d in deps)


def _compile(src):
    obj = _obj(src)
    if not _stale(src, obj):
        return obj
    cmd = [HIPCC] + FLAGS + EXTRA.get(src, []) + ['-x', 'hip', '-c', os.path.join(HERE, src), '-o', obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f'hipcc failed for {src}:\n{r.stderr[-6000:]}')
    if r.stderr.strip():
        sys.stderr.write(r.stderr[-3000:])
    return obj


def build(force=False):
    os.makedirs(os.path.join(HERE, '_build'), exist_ok=True)
    if force:
        for s in SOURCES:
            if os.path.exists(_obj(s)):
                os.remove(_obj(s))
    with ThreadPoolExecutor(max_workers=8) as ex:
        objs = list(ex.map(_compile, SOURCES))
    if not os.path.exists(OUT) or any(os.path.getmtime(o) > os.path.getmtime(OUT) for o in objs):
        cmd = [HIPCC, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', OUT] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f'link failed:\n{r.stderr[-4000:]}')
    return os.path.abspath(OUT)


if __name__ == '__main__':
    print(build(force='--force' in sys.argv))
