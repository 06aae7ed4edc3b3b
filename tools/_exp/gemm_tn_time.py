"""dW = X^T dY at the GNN shapes: library vs oea_gemm_tn_f32"""
import sys
import time

import torch

sys.path.insert(0, '.')
from openea_amd import ops  # noqa: E402


def bench(fn, reps=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / reps * 1e3


dev = torch.device('cuda:0')
m = 200000
g = torch.Generator(device=dev).manual_seed(1)
for k1, k2 in ((500, 400), (400, 400), (400, 300), (300, 400), (300, 300), (500, 500), (512, 512)):
    x = torch.randn(m, k1, device=dev, generator=g)
    dy = torch.randn(m, k2, device=dev, generator=g)
    flops = 2.0 * m * k1 * k2
    t_lib = bench(lambda: torch.mm(x.t(), dy))
    t_own = bench(lambda: ops.gemm_tn(x, dy))
    err = (ops.gemm_tn(x, dy) - x.t() @ dy).abs().max().item()
    print(f'M={m} {k1}x{k2}: lib {t_lib:.3f} ms ({flops / t_lib / 1e9:.1f} TF)  own {t_own:.3f} ms ({flops / t_own / 1e9:.1f} TF)  '
          f'max |diff| {err:.2e}', flush=True)
