"""Dump the VALUES of every in-scope reference config to tests/golden/configs/*.json (build
container only: reads /root/reference).  The files are loaded unmodified through
slotdiffusion_amd.compat.load_params (the BaseParams shim); only plain data is stored."""
import glob
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, '..'))
from slotdiffusion_amd import compat   # noqa: E402

REF = '/root/reference/slotdiffusion'
OUT = os.path.join(HERE, '..', 'tests', 'golden', 'configs')
PATTERNS = ['img_based/configs/sa/*.py', 'img_based/configs/sa_ldm/*.py',
            'video_based/configs/savi/*.py', 'video_based/configs/savi_ldm/*.py']


def plain(v):
    if isinstance(v, dict):
        return {k: plain(x) for k, x in v.items()}
    if isinstance(v, (list, tuple)):
        return [plain(x) for x in v]
    if isinstance(v, (int, float, str, bool)) or v is None:
        return v
    return repr(v)


def main():
    os.makedirs(OUT, exist_ok=True)
    n = 0
    for pat in PATTERNS:
        for path in sorted(glob.glob(os.path.join(REF, pat))):
            task = pat.split('/')[0]
            P = compat.load_params(path)
            d = plain(P.to_dict())
            d['_task'] = task
            d['_source'] = os.path.relpath(path, '/root/reference')
            name = os.path.basename(path)[:-3]
            with open(os.path.join(OUT, f'{task}__{name}.json'), 'w') as f:
                json.dump(d, f, indent=1, sort_keys=True)
            n += 1
            print(task, name, d.get('model'))
    print(n, 'configs')


if __name__ == '__main__':
    main()
