"""ms per epoch of one approach at a synthetic BASELINE shape: python tools/_exp/epoch_time.py NAME 15K|100K [epochs]"""
import contextlib
import io
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tools.profile_models import SHAPE  # noqa: E402
import openea_amd.approaches as approaches  # noqa: E402
from openea_amd.modules.load.synth import make_kgs  # noqa: E402
from openea_amd.run.default_args import get_args  # noqa: E402


def main():
    name, scale = sys.argv[1], sys.argv[2]
    epochs = int(sys.argv[3]) if len(sys.argv) > 3 else 20
    tag = " ".join("%s=%s" % (k, v) for k, v in sorted(os.environ.items()) if k.startswith("OEA_"))
    kgs = make_kgs(SHAPE[scale][name], mode="mapping", seed=0)
    m = getattr(approaches, name)()
    m.set_args(get_args(name, scale=scale, output="/tmp/oea_prof/", training_data="synthetic/x/", dataset_division="f/", max_epoch=1,
                        start_valid=10 ** 6, eval_freq=10 ** 6))
    m.set_kgs(kgs)
    m.args.random_name_init = True
    with contextlib.redirect_stdout(io.StringIO()):
        m.init()
        m.run()
        torch.cuda.synchronize()
        m.args.max_epoch = epochs
        t0 = time.time()
        m.run()
        torch.cuda.synchronize()
    print("%s %s [%s] %d epochs: %.2f ms/epoch" % (name, scale, tag, epochs, (time.time() - t0) / epochs * 1e3), flush=True)


if __name__ == "__main__":
    main()
