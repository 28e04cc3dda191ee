import sys, torch, ctypes
sys.path.insert(0, '/root/repo')
from slotdiffusion_amd import _lib
st = torch.cuda.current_stream().cuda_stream
Item = _lib.CSTRUCT['SdmiColsumItem']
print('Item size', ctypes.sizeof(Item), [f[0] for f in Item._fields_])
items = []
refs = []
g = torch.Generator().manual_seed(0)
for nblk, C in ((64, 128), (128, 384), (5, 40), (1024, 64)):
    part = torch.randn(nblk, C, 2, generator=g).cuda()
    o0 = torch.randn(C, generator=g).cuda(); o1 = torch.randn(C, generator=g).cuda()
    refs.append((o0.clone() + part[..., 0].sum(0), o1.clone() + part[..., 1].sum(0), o0, o1))
    items.append((part, nblk, C, o0, o1))
arr = (Item * len(items))()
for a, (part, nblk, C, o0, o1) in zip(arr, items):
    a.partial, a.out0, a.out1, a.nblk, a.C = part.data_ptr(), o0.data_ptr(), o1.data_ptr(), nblk, C
_lib.call('sdmi_colsum_group', st, items=ctypes.addressof(arr), n=len(items))
torch.cuda.synchronize()
for r0, r1, o0, o1 in refs:
    print(float((o0 - r0).abs().max()), float((o1 - r1).abs().max()))
