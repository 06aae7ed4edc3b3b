"""attention operator at the EN-DE-100K AliNet shape (d = 400, 4.1 M two-hop edges): forward / backward under the row grouping,
and the model epoch under 'row' and 'reorder'"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import bench
from openea_amd import ops
dev = torch.device("cuda", 0)
a, kgs, init_s, ms_epoch = bench._build_alinet(torch, ops, dev, epochs=3)
g2 = a.adj[1]
n, d = kgs.entities_num, a.args.layer_dims[1]
for tag, g in (("runs", g2), ("row", bench._row_grouped(g2, dev))):
    fwd, bwd = bench._attn_closures(torch, ops, dev, g, n, d)
    print("grouping %-5s d=%d nnz=%d groups=%d: fwd %.3f ms, bwd %.3f ms" % (tag, d, g2.nnz, len(g.seg_row_host),
          bench._timed_events(torch, fwd, 10), bench._timed_events(torch, bwd, 10)))
print("epoch (runs): %.2f ms" % ms_epoch)
del a
torch.cuda.empty_cache()
for grp in ("row", "reorder"):
    a2, _, _, ms = bench._build_alinet(torch, ops, dev, grouping=grp, epochs=3)
    print("epoch (%s): %.2f ms" % (grp, ms))
    del a2
    torch.cuda.empty_cache()
