"""Stand-alone timing of the direct 3x3 weight gradient on channel pairs (wgrad3x3_halo_kernel) and its fold, against
the implicit-GEMM kernel (SDMI_WGRAD_HALO=0): run under `rocprofv3 --kernel-trace --stats` (tools/exp/wgrad_halo_time.sh)."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
from slotdiffusion_amd import _lib
dev = 'cuda'
halo = os.environ.get('SDMI_WGRAD_HALO', '1') != '0'
for B, H, C, N in ((64, 32, 128, 128), (64, 32, 256, 128), (64, 32, 256, 256), (64, 64, 128, 128), (64, 16, 256, 256)):
    M, K = B * H * H, 9 * C
    pairs = (N // 64) * (C // 64)
    splits = max(1, min(256 // pairs, M // 256)) if halo else max(1, min((192 + (N // 128) * (K // 128) - 1) // ((N // 128) * (K // 128)), M // 512))
    x = torch.randn(B, H, H, C, device=dev).bfloat16()
    dy = torch.randn(B, H, H, N, device=dev).bfloat16()
    ws = torch.empty(splits * (N * K + N), device=dev)
    dw = torch.zeros(N, K, device=dev)
    db = torch.zeros(N, device=dev)
    for it in range(5):
        _lib.call('sdmi_wgrad', torch.cuda.current_stream().cuda_stream, a=x.data_ptr(), dy=dy.data_ptr(), dw=dw.data_ptr(),
                  dbias=db.data_ptr(), workspace=ws.data_ptr(), dtype=_lib.BF16, M=M, N=N, K=K, lda=C, ldy=N, B=B, H=H, W=H,
                  Cin=C, Ho=H, Wo=H, KH=3, KW=3, stride=1, pad_t=1, pad_l=1, ups=0, splits=splits, accumulate=0)
    torch.cuda.synchronize()
    print(f'B={B} H={H} C={C} N={N} splits={splits} gflop={2.0 * M * N * K / 1e9:.1f}', flush=True)
