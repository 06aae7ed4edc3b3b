"""neighbour refresh at the 100K shape on TRAINED tables (clustered embeddings): time per refresh and the rows the list
paths send to their fallback (OEA_TOPK_DEBUG=1), for the symmetric / general / strip paths (OEA_TOPK_SYM, OEA_TOPK_LISTS)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402
from openea_amd import ops  # noqa: E402
from openea_amd.models.trainer import refresh_neighbours  # noqa: E402

ops.lib()
dev = torch.device("cuda", 0)
steps = int(os.environ.get("STEPS", "400"))
wl = bench.Workload(torch, ops, "EN-FR-100K-V1", 100, 20000, 10, 0.98, dev)
wl.epochs.run_steps(wl.trainer, steps)
torch.cuda.synchronize()
for rep in range(3):
    t0 = time.perf_counter()
    refresh_neighbours(wl.ent, wl.kgs.kg1.entities_list, wl.k1)
    torch.cuda.synchronize()
    print("refresh after %d steps: %.2f ms (SYM=%s LISTS=%s)" % (steps, (time.perf_counter() - t0) * 1e3, os.environ.get("OEA_TOPK_SYM", "1"),
                                                               os.environ.get("OEA_TOPK_LISTS", "1")), flush=True)
