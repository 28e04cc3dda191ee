"""Per-kernel register / scratch / occupancy table of one HIP source (hipcc -Rpass-analysis=kernel-resource-usage).
usage: python tools/resource_usage.py igemm.hip [substring filter]"""
import os
import re
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, '..', 'slotdiffusion_amd', 'csrc')
src = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ''
extra = sys.argv[3:]
r = subprocess.run(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC',
                    '-Rpass-analysis=kernel-resource-usage', '-x', 'hip', '-c', os.path.join(CSRC, src),
                    '-o', '/dev/null'] + extra, capture_output=True, text=True)
if r.returncode:
    sys.stderr.write(r.stderr[-4000:])
    sys.exit(1)
blocks = re.split(r'remark: [^\n]*Function Name: ', r.stderr)[1:]


def field(b, k):
    m = re.search(re.escape(k) + r': (\d+)', b)
    return m.group(1) if m else '?'


for b in blocks:
    name = b.split('\n')[0].split(' [')[0]
    dn = subprocess.run(['c++filt', name], capture_output=True, text=True).stdout.strip()
    dn = dn.replace('(anonymous namespace)::', '').replace('void ', '')
    dn = re.sub(r'\(.*', '', dn)
    if flt and flt not in dn:
        continue
    print(f"{dn[:84]:84s} vgpr {field(b, 'VGPRs'):>3s} agpr {field(b, 'AGPRs'):>3s} sgpr {field(b, 'SGPRs'):>3s} "
          f"scratch {field(b, 'ScratchSize [bytes/lane]'):>4s} occ {field(b, 'Occupancy [waves/SIMD]')} "
          f"lds {field(b, 'LDS Size [bytes/block]')}")
