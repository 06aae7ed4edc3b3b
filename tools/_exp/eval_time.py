"""whole-call time of the inner-product evaluation (bf16 prefilter) at 70,000^2: d = 100 / 300 plain and with CSLS 10; d = 1,200 plain"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from openea_amd import ops
from openea_amd.modules.finding.alignment import greedy_alignment_device
ops.lib()
rng = np.random.RandomState(0)
n = 70000
for d, csls_list in ((100, (0, 10)), (300, (0, 10)), (1200, (0,))):
    e1 = rng.standard_normal((n, d)).astype(np.float32); e1 /= np.linalg.norm(e1, axis=1, keepdims=True)
    e2 = (e1 + 0.9 * rng.standard_normal((n, d)).astype(np.float32) / np.sqrt(d)).astype(np.float32); e2 /= np.linalg.norm(e2, axis=1, keepdims=True)
    t1, t2 = ops.to_table(e1), ops.to_table(e2)
    for csls in csls_list:
        r = greedy_alignment_device(t1, t2, d, [1, 5, 10], "inner", False, csls)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            r = greedy_alignment_device(t1, t2, d, [1, 5, 10], "inner", False, csls)
        torch.cuda.synchronize()
        print("eval 70000^2 x %d csls=%d: %.2f ms  hits %s rank_sum %d" % (d, csls, (time.perf_counter() - t0) / 5 * 1e3, [int(x) for x in r[2]], int(r[3])), flush=True)
    del t1, t2
