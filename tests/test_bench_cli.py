"""bench.py command line: the multi-GPU self-launch and the JSON contract.

CPU: `python bench.py --gpus 2` from a bare shell must neither hang nor exec -- without two devices it exits with a clear
message before any launcher starts.  GPU (one device): the RCCL code path (process-group init, the weight broadcast, an
all-reduce, the barriers) runs with world size 1 under CV_FORCE_DIST=1, and the line carries every block the contract names."""
from __future__ import annotations

import json
import os
import subprocess
import sys
from pathlib import Path

import pytest
import torch

ROOT = Path(__file__).resolve().parent.parent


def test_gpus_gt_visible_devices_is_a_clean_exit():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    want = torch.cuda.device_count() + 2
    out = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", str(want)], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 2
    assert f"--gpus {want}" in out.stderr and "device(s) visible" in out.stderr
    assert out.stdout.strip() == ""


def test_relaunch_command_is_torchrun_on_localhost(monkeypatch):
    """The child command line (no GPU needed): one rank per GPU, rendezvous on 127.0.0.1, the original arguments passed on."""
    sys.path.insert(0, str(ROOT))
    import bench

    seen = {}

    class _Done:
        returncode = 7

    def fake_run(cmd, env=None):
        seen["cmd"], seen["env"] = cmd, env
        return _Done()

    monkeypatch.setattr(bench.subprocess, "run", fake_run)
    monkeypatch.setattr(bench, "visible_gpus", lambda: 8)

    class _Args:
        gpus = 4

    rc = bench.relaunch_under_torchrun(_Args(), ["--gpus", "4", "--steps", "3"])
    assert rc == 7
    cmd = seen["cmd"]
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nproc-per-node=4" in cmd and cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[-4:] == ["--gpus", "4", "--steps", "3"] and cmd[-5].endswith("bench.py")
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


def test_parent_counts_gpus_without_the_hip_runtime(monkeypatch):
    """The parent of `bench.py --gpus N` only launches children: it reads the visibility variables (or the KFD topology) instead of
    asking torch, which would bring the HIP runtime up in a process that never computes (VERDICT r04 'weak' 10)."""
    sys.path.insert(0, str(ROOT))
    import bench

    def boom():
        raise AssertionError("torch.cuda.device_count() must not be needed when the visibility variables are set")

    monkeypatch.setattr(bench.torch.cuda, "device_count", boom)
    for var in ("HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES"):
        monkeypatch.delenv(var, raising=False)
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "0,1,2,3")
    assert bench.visible_gpus() == 4
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "")
    assert bench.visible_gpus() == 0


@pytest.mark.gpu
def test_bench_line_contract_with_rccl_world_of_one():
    import socket

    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    env = dict(os.environ, CV_FORCE_DIST="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--boards", "64", "--steps", "1", "--warmup", "1",
                          "--no-cpu-baseline", "--no-extras"], env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    line = json.loads(out.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 1 and line["rccl_ranks_seen"] == 1 and line["value"] > 0
    assert line["unit"] == "boards/sec" and line["scaling"] == "weak" and line["vs_baseline"] is None and line["dtype"] == "f16x3"
    roof = line["roofline"]
    assert roof["bound"] == "mfma" and roof["unit"] == "TFLOP/s" and abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-3
    assert roof["launches_per_step"] == 40 and "traffic" in roof and "traffic_source" in roof     # 21 UNet (first two convs fused) + 19 ResNet
    dom = roof["dominant"]                                  # the instantiation a rocprofv3 kernel trace shows by template name
    assert dom["kernel"].startswith("conv3x3_halo_kernel<split_t,64,16x16") and dom["launches_per_step"] >= 10
    assert abs(dom["frac"] - dom["achieved"] / roof["peak"]) < 1e-3 and 0 < dom["share_of_conv_time"] <= 1
    assert sum(k["launches"] for k in roof["by_kernel"].values()) == roof["launches_per_step"]
    assert line["sharding"]["gathered_in_order"] is True and line["host_threads_per_rank"] >= 1
    hbm = line["roofline_hbm"]
    assert {"stem7x7+maxpool (mfma)", "head_avgpool_fc"} <= set(hbm)          # (the input packing is fused into the first conv)
    for blk in hbm.values():
        assert blk["bound"] == "hbm" and blk["peak"] == 8000.0 and 0 < blk["frac"] < 1


@pytest.mark.gpu
def test_two_ranks_on_one_gpu_over_gloo_run_the_sharded_bench():
    """The N > 1 code path of bench.py on a one-GPU box: RCCL refuses two ranks per device, so CV_DIST_BACKEND=gloo carries the
    collectives (host buffers) while both ranks compute on the same MI355X.  `python bench.py --gpus 2` from a bare shell starts
    its own torchrun child (never an exec), rank 0 generates the weights and both ranks receive them by broadcast, the global
    batch of 2 x 256 boards (BASELINE configs[4]'s per-rank shape) is sharded r::2, rank 1 leaves through the barrier / shutdown
    branch, the sharded host pipeline runs on both ranks at configs[4]'s precision (f16x3 UNet + the classifier's fp16 mode), the
    range calibration is synchronised from rank 0, and exactly one JSON line reaches stdout."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "LOCAL_WORLD_SIZE", "CV_FORCE_DIST")}
    env["CV_DIST_BACKEND"] = "gloo"
    out = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--boards", "256", "--steps", "1", "--warmup", "1",
                          "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=2400)
    assert out.returncode == 0, out.stderr[-4000:]
    lines = [ln for ln in out.stdout.strip().splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["rccl_ranks_seen"] == 2 and line["dist_backend"] == "gloo"
    assert line["config"]["global_boards_per_step"] == 512 and line["scaling"] == "weak" and line["value"] > 0
    assert line["sharding"] == {"global_boards": 512, "rule": "rank r owns boards r::world", "gathered_in_order": True}
    assert line["calibration_sync"]["identical_across_ranks"] is True and line["init_s"]["max"] > 0
    assert line["config"]["init_s_max_over_ranks"] == line["init_s"]["max"] and line["config"]["rccl_ranks_seen"] == 2
    ncpu = os.cpu_count() or 1
    assert 1 <= line["host_threads_per_rank"] <= max(1, min(32, ncpu // 2))
    e2e = line["pipeline_e2e_ranks"]
    assert e2e["global_boards"] == 512 and e2e["fens_on_rank0"] == 512
    assert e2e["precision"] == "f16x3+f16r" and e2e["calibration_identical_across_ranks"] is True
    assert line["config"]["pipeline_e2e_ranks_precision"] == "f16x3+f16r"
    assert line["config"]["pipeline_e2e_ranks_boards_per_sec_min"] == round(e2e["boards_per_sec_per_rank"]["min"], 1)
    r = e2e["boards_per_sec_per_rank"]
    assert 0 < r["min"] <= r["mean"] <= r["max"] and e2e["boards_per_sec_whole_job"] > 0


@pytest.mark.gpu
def test_four_ranks_on_one_gpu_over_gloo():
    """World size 4 (BASELINE's 1/2/4/8 curve has no hardware here): four ranks share the one MI355X over gloo, 4 x 32 boards sharded
    r::4; the rank-major gather re-interleaves to board order, every rank takes part in the collectives, rank 0 holds all 128 FENs."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "LOCAL_WORLD_SIZE", "CV_FORCE_DIST")}
    env["CV_DIST_BACKEND"] = "gloo"
    out = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "4", "--boards", "32", "--steps", "1", "--warmup", "1",
                          "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=2400)
    assert out.returncode == 0, out.stderr[-4000:]
    lines = [ln for ln in out.stdout.strip().splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    line = json.loads(lines[0])
    assert line["n_gpus"] == 4 and line["rccl_ranks_seen"] == 4 and line["config"]["global_boards_per_step"] == 128
    assert line["sharding"]["gathered_in_order"] is True and line["calibration_sync"]["identical_across_ranks"] is True
    e2e = line["pipeline_e2e_ranks"]
    assert e2e["global_boards"] == 128 and e2e["fens_on_rank0"] == 128 and e2e["calibration_identical_across_ranks"] is True


@pytest.mark.gpu
def test_eight_ranks_on_one_gpu_run_configs4_global_shape():
    """BASELINE configs[4] at its REAL global shape, functionally, on the one GPU there is: `CV_DIST_BACKEND=gloo python bench.py --gpus 8
    --boards 256` = 2048 boards sharded r::8 over eight ranks that share the MI355X (RCCL refuses that; gloo carries the collectives),
    f16x3 UNet + fp16 classifier in the sharded host pipeline.  Asserted: one JSON line; all 2048 FENs on rank 0 and in the caller's
    order (rank 0 recomputes photos of every other rank's shard and compares); identical calibration on all ranks; `init_s` and
    `gather_s` per rank in the line; the device-wide memory peak (eight workspaces on one device) below 288 GB; start-up of the slowest
    of eight ranks within 1.5x of... the fastest (the ranks pack and calibrate side by side: a serialised start-up would show as 8x).
    No scaling claim: eight ranks time-share one GPU."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "LOCAL_WORLD_SIZE", "CV_FORCE_DIST")}
    env["CV_DIST_BACKEND"] = "gloo"
    out = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "8", "--boards", "256", "--steps", "1", "--warmup", "1",
                          "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=3000)
    assert out.returncode == 0, out.stderr[-4000:]
    lines = [ln for ln in out.stdout.strip().splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    line = json.loads(lines[0])
    assert line["n_gpus"] == 8 and line["rccl_ranks_seen"] == 8 and line["dist_backend"] == "gloo"
    assert line["config"]["global_boards_per_step"] == 2048 and line["sharding"]["gathered_in_order"] is True
    assert line["calibration_sync"]["identical_across_ranks"] is True
    assert len(line["init_s_per_rank"]) == 8 and min(line["init_s_per_rank"]) > 0
    assert max(line["init_s_per_rank"]) <= 1.5 * min(line["init_s_per_rank"]), line["init_s_per_rank"]     # side by side, not one after the other
    e2e = line["pipeline_e2e_ranks"]
    assert e2e["global_boards"] == 2048 and e2e["fens_on_rank0"] == 2048 and e2e["precision"] == "f16x3+f16r"
    assert e2e["order_checked_on_rank0"] == 14 and e2e["order_mismatches"] == 0
    assert e2e["calibration_identical_across_ranks"] is True
    assert len(e2e["gather_s_per_rank"]) == 8 and len(e2e["shard_s_per_rank"]) == 8 and max(e2e["gather_s_per_rank"]) > 0
    mem = line["device_memory"]
    assert 20.0 < mem["peak_used_gb_max_over_ranks"] < 288.0 and mem["device_total_gb"] > 250.0, mem
    assert line["config"]["device_memory_peak_used_gb"] == mem["peak_used_gb_max_over_ranks"]
    print("configs[4] on one device:", {"init_s": line["init_s"], "peak_gb": mem["peak_used_gb_max_over_ranks"], "gather_s": e2e["gather_s"],
                                        "boards_per_sec_whole_job": e2e["boards_per_sec_whole_job"]})
