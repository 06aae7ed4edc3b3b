"""AliNet's dense layers at the EN-DE-100K shape: library fp32 GEMM against this repo's NT tile pipeline (oea_sim_matrix,
inner metric).  Prints ms and TFLOP/s per shape for  Y = X W (NN),  dX = dY W^T (NT),  dW = X^T dY (TN)."""
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


def main():
    dev = torch.device('cuda:0')
    m = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
    g = torch.Generator(device=dev).manual_seed(1)
    for k, n in ((500, 400), (400, 300), (400, 400), (300, 300), (500, 500)):
        x = torch.randn(m, k, device=dev, generator=g)
        w = torch.randn(k, n, device=dev, generator=g) * 0.05
        dy = torch.randn(m, n, device=dev, generator=g)
        wt = w.t().contiguous()
        flops = 2.0 * m * k * n
        y_lib = x @ w
        y_own = ops.sim_matrix(x, wt, k)
        err = (y_lib - y_own).abs().max().item() / y_lib.abs().max().item()
        dx_lib = dy @ wt
        dx_own = ops.sim_matrix(dy, w, n)
        err2 = (dx_lib - dx_own).abs().max().item() / dx_lib.abs().max().item()
        t_nn = bench(lambda: torch.mm(x, w))
        t_nn_own = bench(lambda: ops.sim_matrix(x, wt, k))
        t_nt = bench(lambda: torch.mm(dy, wt))
        t_nt_own = bench(lambda: ops.sim_matrix(dy, w, n))
        t_tn = bench(lambda: torch.mm(x.t(), dy))
        print(f'M={m} K={k} N={n}: NN lib {t_nn:.3f} ms ({flops / t_nn / 1e9:.1f} TF) own {t_nn_own:.3f} ms '
              f'({flops / t_nn_own / 1e9:.1f} TF) rel err {err:.1e} | NT lib {t_nt:.3f} own {t_nt_own:.3f} err {err2:.1e} | '
              f'TN lib {t_tn:.3f} ms ({flops / t_tn / 1e9:.1f} TF)', flush=True)


if __name__ == '__main__':
    main()
