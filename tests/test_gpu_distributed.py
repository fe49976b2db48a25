"""GPU: the collectives of the sharded pipeline over RCCL itself (world of one, CV_FORCE_DIST=1) -- what the world_size-2 gloo
tests cannot see: RCCL moves device memory only, so the host-side result arrays of `process_images_sharded` (probabilities,
quadrangles, masks) must be staged through the rank's GPU and come back as host arrays."""
from __future__ import annotations

import os
import socket
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
pytestmark = pytest.mark.gpu

SCRIPT = r"""
import sys, numpy as np, torch
sys.path.insert(0, "ROOT/chessvision-3lc_amd"); sys.path.insert(0, "ROOT"); sys.path.insert(0, "ROOT/tests")
from chessvision import distributed as cvd
from test_distributed_cpu import _FakeVision
rank, world, device = cvd.init_process_group()
assert cvd.backend_name() == "nccl" and world == 1 and device.type == "cuda"
host = torch.arange(12, dtype=torch.float32).reshape(4, 3)
got = cvd.all_gather_rows(host)                               # host tensor in, host tensor out, RCCL in between
assert got.device.type == "cpu" and torch.equal(got, host)
dev = cvd.all_gather_rows(host.to(device))
assert dev.device == device and torch.equal(dev.cpu(), host)
assert cvd.stats_over_ranks(2.5, device) == {"min": 2.5, "mean": 2.5, "max": 2.5} and cvd.count_ranks(device) == 1
photos = [np.full((4, 4, 3), i, np.uint8) for i in range(5)]
tm = {}
res = cvd.process_images_sharded(_FakeVision(), photos, timings=tm)          # gather path forced although world == 1
assert [int(r.board_extraction.binary_mask[0, 0]) for r in res] == list(range(5)) and tm["gather_s"] > 0
assert res[3].position is None and res[4].position.fen == "local-4"
cvd.barrier(device); cvd.shutdown()
print("RCCL_HOST_STAGING_OK")
"""


def test_host_arrays_travel_through_rccl():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "LOCAL_WORLD_SIZE", "CV_DIST_BACKEND")}
    with socket.socket() as sock:                            # a free port of this box, not a fixed one that another job may hold
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    env.update(CV_FORCE_DIST="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    out = subprocess.run([sys.executable, "-c", SCRIPT.replace("ROOT", str(ROOT))], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "RCCL_HOST_STAGING_OK" in out.stdout, out.stderr[-3000:]
