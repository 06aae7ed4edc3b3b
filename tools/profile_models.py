#!/usr/bin/env python
"""Wall-clock of whole-model epochs of the GNN approaches on synthetic KG pairs of the BASELINE shapes
(device-synchronised): host-side init (adjacency builders), then per-epoch time with validation off.
    python tools/profile_models.py [15K|100K] [GCN_Align,AliNet,RDGCN]"""
import contextlib
import io
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import openea_amd.approaches as approaches  # noqa: E402
import openea_amd.models.trans as trans  # noqa: E402
from openea_amd.modules.load.synth import make_kgs  # noqa: E402
from openea_amd.run.default_args import get_args  # noqa: E402

SHAPE = {"15K": {"GCN_Align": "D-W-15K-V2", "AliNet": "EN-FR-15K-V1", "RDGCN": "EN-FR-15K-V1", "MTransE": "EN-FR-15K-V1",
                 "AlignE": "EN-FR-15K-V1", "BootEA": "EN-FR-15K-V1", "BootEA_TransH": "EN-FR-15K-V1",
                 "BootEA_RotatE": "EN-FR-15K-V1", "TransE": "EN-FR-15K-V1", "TransH": "EN-FR-15K-V1", "TransD": "EN-FR-15K-V1"},
         "100K": {"GCN_Align": "EN-FR-100K-V1", "AliNet": "EN-DE-100K-V1", "RDGCN": "EN-FR-100K-V2", "MTransE": "EN-FR-100K-V1",
                  "AlignE": "EN-FR-100K-V1", "BootEA": "EN-FR-100K-V1", "BootEA_TransH": "EN-FR-100K-V1",
                  "BootEA_RotatE": "EN-FR-100K-V1", "TransE": "EN-FR-100K-V1", "TransH": "EN-FR-100K-V1", "TransD": "EN-FR-100K-V1"}}
MODE = {"AlignE": "swapping", "BootEA": "swapping", "BootEA_TransH": "swapping", "BootEA_RotatE": "swapping", "TransE": "sharing",
        "TransH": "sharing", "TransD": "sharing"}


def main():
    scale = sys.argv[1] if len(sys.argv) > 1 else "15K"
    names = (sys.argv[2] if len(sys.argv) > 2 else "GCN_Align,AliNet,RDGCN").split(",")
    epochs = 20 if any(n in MODE or n == "MTransE" for n in names) else 5
    for name in names:
        shape = SHAPE[scale][name]
        t0 = time.time()
        kgs = make_kgs(shape, mode=MODE.get(name, "mapping"), seed=0)
        t_data = time.time() - t0
        m = (getattr(approaches, name, None) or getattr(trans, name))()
        m.set_args(get_args(name, scale=scale, output="/tmp/oea_prof/", training_data="synthetic/%s/" % shape, dataset_division="f/",
                            max_epoch=epochs, start_valid=10 ** 6, eval_freq=10 ** 6, sub_epoch=10))
        m.set_kgs(kgs)
        m.args.random_name_init = True              # RDGCN: no word-vector file on the box
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            t0 = time.time()
            m.init()
            torch.cuda.synchronize()
            t_init = time.time() - t0
            m.args.max_epoch = 10 if name.startswith("BootEA") else 1
            m.run()                                     # first epoch(s): lazy allocations / first-use costs
            torch.cuda.synchronize()
            m.args.max_epoch = epochs
            t0 = time.time()
            m.run()
            torch.cuda.synchronize()
            t_run = time.time() - t0
        print("%-10s %-14s entities %7d  synth %.1f s  init (host graph builders + upload) %.2f s  epoch %.2f ms"
              % (name, shape, kgs.entities_num, t_data, t_init, t_run / epochs * 1e3), flush=True)


if __name__ == "__main__":
    main()
