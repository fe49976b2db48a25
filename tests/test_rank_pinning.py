"""CPU: topology-aware rank pinning on fake sysfs trees.  No GPU, no process is re-pinned here: the readers
take a sysfs root, the planner is a pure function."""
from __future__ import annotations

import os
from pathlib import Path

from chessvision import distributed as cvd


def _fake_sysfs(root: Path, sockets: int, cores_per_socket: int, smt: int, gpus_per_socket: int, numbering: str = "cores_then_siblings",
                local_cpulist: bool = True, numa_file: bool = True) -> dict:
    """A `sockets` x `cores_per_socket` x `smt` host with `gpus_per_socket` GPUs behind every socket.  CPU numbering
    "cores_then_siblings": thread t of (socket s, core c) is CPU t * sockets * cores + s * cores + c (what Linux shows on most
    two-socket EPYC hosts); "siblings_adjacent": (s * cores + c) * smt + t.  KFD nodes: the CPU nodes first, then the GPUs."""
    n_cores = sockets * cores_per_socket

    def cpu_of(s, c, t):
        return t * n_cores + s * cores_per_socket + c if numbering == "cores_then_siblings" else (s * cores_per_socket + c) * smt + t

    node_cpus = {s: sorted(cpu_of(s, c, t) for c in range(cores_per_socket) for t in range(smt)) for s in range(sockets)}
    for s in range(sockets):
        for c in range(cores_per_socket):
            sib = sorted(cpu_of(s, c, t) for t in range(smt))
            for cpu in sib:
                d = root / f"devices/system/cpu/cpu{cpu}/topology"
                d.mkdir(parents=True, exist_ok=True)
                (d / "thread_siblings_list").write_text(",".join(str(v) for v in sib) + "\n")
        nd = root / f"devices/system/node/node{s}"
        nd.mkdir(parents=True, exist_ok=True)
        (nd / "cpulist").write_text(",".join(str(v) for v in node_cpus[s]) + "\n")
    (root / "devices/system/cpu/cpufreq").mkdir(parents=True, exist_ok=True)      # a non-cpuN entry, as on real hosts
    nid = 0
    for s in range(sockets):                                                       # CPU nodes of the KFD topology
        d = root / f"class/kfd/kfd/topology/nodes/{nid}"
        d.mkdir(parents=True, exist_ok=True)
        (d / "properties").write_text("cpu_cores_count 64\nsimd_count 0\ndrm_render_minor 0\n")
        nid += 1
    gpu_home = []
    for s in range(sockets):
        for g in range(gpus_per_socket):
            minor = 128 + len(gpu_home)
            d = root / f"class/kfd/kfd/topology/nodes/{nid}"
            d.mkdir(parents=True, exist_ok=True)
            (d / "properties").write_text(f"cpu_cores_count 0\nsimd_count 1024\ndrm_render_minor {minor}\n")
            dev = root / f"class/drm/renderD{minor}/device"
            dev.mkdir(parents=True, exist_ok=True)
            if local_cpulist:
                (dev / "local_cpulist").write_text(",".join(str(v) for v in node_cpus[s]) + "\n")
            if numa_file:
                (dev / "numa_node").write_text(f"{s}\n")
            gpu_home.append(s)
            nid += 1
    return {"node_cpus": node_cpus, "gpu_home": gpu_home, "n_cpus": n_cores * smt}


def _cores(plan_row, sysfs_root: Path):
    return {tuple(cvd.parse_cpulist((sysfs_root / f"devices/system/cpu/cpu{c}/topology/thread_siblings_list").read_text())) for c in plan_row}


def test_parse_cpulist():
    assert cvd.parse_cpulist("0-3,8,10-11\n") == [0, 1, 2, 3, 8, 10, 11]
    assert cvd.parse_cpulist("") == [] and cvd.parse_cpulist("a-b") == [] and cvd.parse_cpulist(" 5 ") == [5]


def test_two_sockets_four_gpus_each_smt2_every_rank_on_its_gpus_node_and_no_shared_core(tmp_path, monkeypatch):
    for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        monkeypatch.delenv(var, raising=False)
    for numbering in ("cores_then_siblings", "siblings_adjacent"):
        root = tmp_path / numbering
        host = _fake_sysfs(root, sockets=2, cores_per_socket=16, smt=2, gpus_per_socket=4, numbering=numbering)
        topo = cvd.read_gpu_topology(str(root))
        assert topo is not None and len(topo["gpu_cpus"]) == 8 and len(topo["siblings"]) == 64
        plan = cvd.plan_rank_cpus(range(host["n_cpus"]), 8, topo["gpu_cpus"], topo["siblings"])
        assert plan is not None and len(plan) == 8
        all_cores = []
        for r, row in enumerate(plan):
            assert len(row) == 8                                                   # 16 cores / 4 ranks = 4 cores x 2 threads
            assert set(row) <= set(host["node_cpus"][host["gpu_home"][r]]), (numbering, r, row)     # on its GPU's socket
            cores = _cores(row, root)
            assert len(cores) == 4 and all(set(core) <= set(row) for core in cores)                 # whole cores: siblings stay together
            all_cores.extend(cores)
        assert len(all_cores) == len(set(all_cores)) == 32                         # no two ranks share a physical core
        assert sorted(c for row in plan for c in row) == list(range(64))           # and no CPU is left idle
    # what the topology-blind blocks did on the cores-then-siblings host (the defect this replaces): ranks 2-3 on the wrong socket,
    # ranks r and r + 4 on the same physical cores
    host = _fake_sysfs(tmp_path / "blind", 2, 16, 2, 4)
    blind = cvd.plan_rank_cpus(range(64), 8)
    assert not set(blind[2]) <= set(host["node_cpus"][0]) and _cores(blind[0], tmp_path / "blind") == _cores(blind[4], tmp_path / "blind")


def test_numa_node_file_alone_and_restricted_affinity(tmp_path, monkeypatch):
    for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        monkeypatch.delenv(var, raising=False)
    host = _fake_sysfs(tmp_path, 2, 8, 2, 2, local_cpulist=False)                  # only numa_node -> node<k>/cpulist
    topo = cvd.read_gpu_topology(str(tmp_path))
    assert topo["gpu_cpus"] == [host["node_cpus"][0]] * 2 + [host["node_cpus"][1]] * 2
    allowed = [c for c in range(32) if c not in (0, 16)]                           # a cgroup took core 0 (CPUs 0 and 16) away
    plan = cvd.plan_rank_cpus(allowed, 4, topo["gpu_cpus"], topo["siblings"])
    assert all(set(row) <= set(allowed) for row in plan)
    assert [len(row) for row in plan] == [6, 6, 8, 8]                              # socket 0: 7 cores / 2 ranks = 3 each; socket 1: 4 each
    assert set(plan[0]) | set(plan[1]) <= set(host["node_cpus"][0]) and set(plan[2]) | set(plan[3]) <= set(host["node_cpus"][1])
    # more ranks than GPUs (gloo tests: ranks share devices): rank r computes on device r % n_devices and shares that node
    plan8 = cvd.plan_rank_cpus(range(32), 8, topo["gpu_cpus"], topo["siblings"])
    for r, row in enumerate(plan8):
        assert set(row) <= set(host["node_cpus"][host["gpu_home"][r % 4]]) and len(row) == 4
    assert len({c for row in plan8 for c in row}) == 32


def test_visible_devices_reorder_the_homes_and_unknown_syntax_falls_back(tmp_path, monkeypatch):
    host = _fake_sysfs(tmp_path, 2, 8, 2, 2)
    for var in ("ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        monkeypatch.delenv(var, raising=False)
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "3,0")
    topo = cvd.read_gpu_topology(str(tmp_path))
    assert topo["gpu_cpus"] == [host["node_cpus"][1], host["node_cpus"][0]]
    monkeypatch.setenv("ROCR_VISIBLE_DEVICES", "1,2,3")                            # ROCR filters first, HIP indexes into what is left
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "2,0")
    topo = cvd.read_gpu_topology(str(tmp_path))
    assert topo["gpu_cpus"] == [host["node_cpus"][1], host["node_cpus"][0]]        # physical 3, physical 1
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "GPU-deadbeef")
    assert cvd.read_gpu_topology(str(tmp_path)) is None
    assert cvd.read_gpu_topology(str(tmp_path / "nothing_here")) is None           # a container without /sys/class/kfd


def test_without_topology_the_plan_is_the_contiguous_blocks_and_too_few_cpus_pin_nothing():
    assert cvd.plan_rank_cpus(range(8), 2) == [[0, 1, 2, 3], [4, 5, 6, 7]]
    assert cvd.plan_rank_cpus([3, 5], 4) is None
    assert cvd.plan_rank_cpus(range(8), 2, [[], []], {}) == [[0, 1, 2, 3], [4, 5, 6, 7]]    # GPUs without a home share everything
    # fewer cores than ranks on a node: no plan (the caller falls back to the blocks)
    assert cvd.plan_rank_cpus(range(4), 4, [[0, 1, 2, 3]] * 4, {0: (0, 1), 1: (0, 1), 2: (2, 3), 3: (2, 3)}) is None


def test_pin_rank_cpus_applies_the_plan_of_the_fake_tree_in_a_child_process(tmp_path):
    """End to end through the environment (CV_SYSFS_ROOT, LOCAL_RANK, LOCAL_WORLD_SIZE) in a child process, so the test runner keeps
    its own affinity: the child's mask after pin_rank_cpus() is its rank's row of the plan, restricted to the CPUs it was allowed."""
    import subprocess
    import sys
    allowed = sorted(os.sched_getaffinity(0))
    if len(allowed) < 4:
        import pytest
        pytest.skip("needs 4 CPUs")
    # a fake host whose two 'sockets' are the two halves of the CPUs this process may use, SMT pairs = neighbours
    half = len(allowed) // 2 // 2 * 2
    root = tmp_path
    socks = [allowed[:half], allowed[half:2 * half]]
    for s, cpus in enumerate(socks):
        for i in range(0, len(cpus), 2):
            for c in cpus[i:i + 2]:
                d = root / f"devices/system/cpu/cpu{c}/topology"
                d.mkdir(parents=True, exist_ok=True)
                (d / "thread_siblings_list").write_text(f"{cpus[i]},{cpus[i + 1]}\n")
    for k in range(2):
        d = root / f"class/kfd/kfd/topology/nodes/{k}"
        d.mkdir(parents=True, exist_ok=True)
        (d / "properties").write_text(f"simd_count 1024\ndrm_render_minor {128 + k}\n")
        dev = root / f"class/drm/renderD{128 + k}/device"
        dev.mkdir(parents=True, exist_ok=True)
        (dev / "local_cpulist").write_text(",".join(str(c) for c in socks[1 - k]) + "\n")      # GPU 0 hangs off the SECOND half
    code = ("import os, sys; sys.path.insert(0, %r); from chessvision import distributed as d; "
            "print(sorted(d.pin_rank_cpus() or []), sorted(os.sched_getaffinity(0)), d.host_threads())" % str(Path(cvd.__file__).resolve().parent.parent))
    for lr in range(2):
        env = dict(os.environ, CV_SYSFS_ROOT=str(root), LOCAL_RANK=str(lr), LOCAL_WORLD_SIZE="2", WORLD_SIZE="2", RANK=str(lr))
        for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES", "CV_PIN_RANK_CPUS", "CV_PIN_TOPOLOGY"):
            env.pop(var, None)
        out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
        assert out.returncode == 0, out.stderr[-2000:]
        want = sorted(socks[1 - lr])
        assert out.stdout.strip() == f"{want} {want} {min(32, len(want))}", (lr, out.stdout, want)
