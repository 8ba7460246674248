"""Diagnostic: are eager and graph-replayed fit() runs from the same seeds bit-identical, with and without the
one-rank data-parallel rehearsal set-up (tools/rccl_one_rank_rehearsal.py)?  MODE = dp | single"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, ".."))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

import rccl_one_rank_rehearsal as R  # noqa: E402
from vegans.engine import parallel  # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else "dp"
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29531")
torch.cuda.set_device(0)
if mode == "dp":
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    sharded = parallel._Sharded
    parallel._dist_on = lambda: True
    parallel._Sharded = lambda world, rank: sharded(2, 0)
    os.environ["VEGANS_DP_CAPTURE_COLLECTIVES"] = "0"
rows = 512 if mode == "dp" else 256
_, pg = R.run(6, True, rows)
_, pe = R.run(6, False, rows)
R.compare("%s: graph vs eager" % mode, pg, pe)
_, pe2 = R.run(6, False, rows)
R.compare("%s: eager vs eager again" % mode, pe, pe2)
_, pg2 = R.run(6, True, rows)
R.compare("%s: graph vs graph again" % mode, pg, pg2)
