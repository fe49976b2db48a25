#!/usr/bin/env python3
"""bench.py -- boards/sec of the ChessVision CNN hot path on MI355X (BASELINE.json metric).

One "step" = one pass of the hot path over one batch of synthetic boards, resident in HBM:
    UNet(3->1) forward on B 256x256 images  +  ResNet-18 forward on 64*B 64x64 squares
(BASELINE.json configs[3]/[4]: the end-to-end batch of 256 boards per GPU; the classical-CV stages between
the two models are SURVEY.md section 8(f) "next" rows and are not inside the timed region).

    python bench.py --gpus N --steps K --warmup W [--dtype f32|f16] [--boards B]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One process per GPU; weights are generated on rank 0 and replicated with ONE RCCL broadcast; boards are
sharded (each rank owns B boards, no data-path collective) => "scaling": "weak".  Rank 0 prints one JSON line.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT / "chessvision-3lc_amd"))
sys.path.insert(0, str(ROOT))

import torch  # noqa: E402

# MI355X dense MFMA peaks (MI355X_MICROARCH.md, chip table): f32-input MFMA 157.3 TF, f16 MFMA ~2500 TF.
# f16x3 computes every algorithmic MAC with THREE f16 MFMA products (hi*hi + hi*lo + lo*hi), so the dense peak of
# that arithmetic type is 2500 / 3 algorithmic TFLOP/s; the line also carries the fraction of the raw f16 peak.
PEAK_TFLOPS = {"f32": 157.3, "f16": 2500.0, "f16x3": 2500.0 / 3.0}
MFMA_PER_MAC = {"f32": 1, "f16": 1, "f16x3": 3}


def pmc_traffic(dtype, unet_chunk, resnet_chunk, unet_launches, resnet_launches):
    """Per-launch HBM bytes of the conv family from the committed rocprofv3 PMC passes (profiles/r01_pmc_traffic.json:
    separate --pmc FETCH_SIZE / WRITE_SIZE runs of the same chunk sizes, gfx950-corrected); None if not collected."""
    path = ROOT / "profiles" / "r01_pmc_traffic.json"
    if not path.exists() or (unet_chunk, resnet_chunk) != (64, 16384):
        return None
    t = json.load(open(path))
    u, r = t.get(f"{dtype}_unet"), t.get(f"{dtype}_resnet18")
    if not u or not r:
        return None
    total = u["hbm_bytes_per_launch"] * unet_launches + r["hbm_bytes_per_launch"] * resnet_launches
    return total / max(1, unet_launches + resnet_launches)


def conv_algorithmic_bytes(dtype, unet_chunk, resnet_chunk):
    """Compulsory HBM bytes of the conv launches of one UNet chunk and one ResNet-18 chunk: every layer's input
    (+ residual), output and weights moved once at the engine's storage width (f32 4 B, f16 2 B, f16x3 4 B = hi+lo)."""
    esz = {"f32": 4, "f16": 2, "f16x3": 4}[dtype]
    def conv(n, hw_out, cin, cout, k=3, stride=1, res=False, out_ch=None):
        px = n * hw_out * hw_out
        inp = px * (stride * stride if k == 3 else 1) * cin                  # a strided 1x1 reads every other pixel only
        return (inp + px * (cout if out_ch is None else out_ch) * (2 if res else 1) + k * k * cin * cout) * esz
    u, n = [], unet_chunk
    u += [conv(n, 256, 8, 64), conv(n, 256, 64, 64)]                                   # 3 input channels stored as 8
    u[-1] += n * 128 * 128 * 64 * esz                                                  # + the fused 2x2 max-pool's output
    for hw, c in ((128, 64), (64, 128), (32, 256), (16, 512)):
        u += [conv(n, hw, c, 2 * c), conv(n, hw, 2 * c, 2 * c)]
        if hw > 16:
            u[-1] += n * (hw // 2) ** 2 * 2 * c * esz
    for hw, c in ((32, 1024), (64, 512), (128, 256), (256, 128)):                      # hw = output of the transposed conv
        u.append((n * (hw // 2) ** 2 * c + n * hw * hw * (c // 2) + 4 * c * (c // 2)) * esz)
        last = hw == 256
        u += [conv(n, hw, c, c // 2), conv(n, hw, c // 2, c // 2, out_ch=1 if last else None)]   # fused 1x1 head: 1 channel out
    r, n = [], resnet_chunk
    r += [conv(n, 16, 64, 64), conv(n, 16, 64, 64, res=True)] * 2
    for hw, c in ((8, 64), (4, 128), (2, 256)):
        r += [conv(n, hw, c, 2 * c, stride=2), conv(n, hw, c, 2 * c, k=1, stride=2), conv(n, hw, 2 * c, 2 * c, res=True),
              conv(n, hw, 2 * c, 2 * c), conv(n, hw, 2 * c, 2 * c, res=True)]
    return u, r


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def cpu_baseline(x_cpu: torch.Tensor, sq_cpu: torch.Tensor, gpu_logits: torch.Tensor, gpu_cls: torch.Tensor,
                 budget_s: float = 12.0, max_boards: int = 64):
    """Time the oracle (torch-CPU restatement = the arithmetic the reference runs) the way the reference drives
    it -- UNet batch 1 + classifier batch 64 per board (core.py:215-220,236-241) -- on a bounded sample, and
    use the same outputs to check the GPU results.  This is the ONLY place bench.py touches oracle/."""
    from oracle import synth

    unet, resnet = synth.make_unet(1), synth.make_resnet(2)
    threads = torch.get_num_threads()
    unet = unet.to(memory_format=torch.channels_last)            # core.py:89
    errs_u, errs_r = [], []
    with torch.no_grad():
        unet(x_cpu[:1]); resnet(sq_cpu[:64])                      # warm-up
        t0 = time.perf_counter()
        done = 0
        while done < min(max_boards, x_cpu.shape[0]):
            lo = unet(x_cpu[done:done + 1])
            cl = resnet(sq_cpu[done * 64:(done + 1) * 64])
            errs_u.append(float((lo - gpu_logits[done:done + 1]).abs().max()))
            errs_r.append(float((cl - gpu_cls[done * 64:(done + 1) * 64]).abs().max()))
            done += 1
            if time.perf_counter() - t0 >= budget_s:
                break
        dt = time.perf_counter() - t0
    return ({"value": round(done / dt, 3), "unit": "boards/sec", "cores": threads, "kind": "port",
             "sample": f"{done} boards, reference-style loop (UNet b=1 + ResNet-18 b=64 per board), torch {torch.__version__} CPU fp32, "
                       f"{os.cpu_count()} logical CPUs"},
            {"unet_logit_max_abs_err": max(errs_u), "resnet_logit_max_abs_err": max(errs_r), "boards_checked": done})


def pipeline_e2e(dtype: str, n_boards: int = 256):
    """BASELINE configs[3] end to end through the public API: 512x512 BGR photos on the HOST -> ChessVision.process_images
    (H2D, resize, UNet, mask D2H, C++ contour, warp+split, ResNet-18, softmax, FEN).  Random-init weights never draw a
    quadrangle, so fallback_quad routes every board through the classifier; reported beside the
    headline, never as `value`."""
    import tempfile

    import numpy as np

    from chessvision import ChessVision, synthetic

    with tempfile.TemporaryDirectory() as d:
        pe, pc = synthetic.save_checkpoints(d)
        cv = ChessVision(board_extractor_weights=str(pe), classifier_weights=str(pc), precision=dtype)
        rng = np.random.default_rng(0)
        images = [rng.integers(0, 256, (512, 512, 3), dtype=np.uint8) for _ in range(n_boards)]
        cv.process_images(images[:80], fallback_quad=True)                # warm-up (lazy model init, pinned buffers)
        t0 = time.perf_counter()
        res = cv.process_images(images, fallback_quad=True)
        dt = time.perf_counter() - t0
    found = sum(r.position is not None for r in res)
    return {"boards_per_sec": round(n_boards / dt, 1), "boards": n_boards, "classified": found,
            "note": "host images in, FEN out, one host thread software-pipelined against the GPU in jobs of 64 boards; includes PCIe, "
                    "the C++ contour stage and Python post-processing"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--dtype", default=os.environ.get("CV_BENCH_DTYPE", "f16x3"), choices=["f32", "f16", "f16x3"])
    ap.add_argument("--boards", type=int, default=256, help="boards per GPU per step")
    ap.add_argument("--unet-chunk", type=int, default=64)
    ap.add_argument("--resnet-chunk", type=int, default=16384)
    ap.add_argument("--overlap", type=int, default=int(os.environ.get("CV_BENCH_OVERLAP", "0")),
                    help="1: enqueue the UNet pass and the ResNet pass of a step on two HIP streams (they are independent)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-budget", type=float, default=12.0)
    args = ap.parse_args()

    from chessvision import distributed as cvd
    from chessvision import synthetic
    from chessvision.hip_backend import HipEngine

    rank, world, device = cvd.init_process_group()
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    if device.type != "cuda":
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")

    # ---- weights: rank 0 generates, one RCCL broadcast replicates ----
    uspec, rspec = synthetic.unet_spec(False), synthetic.resnet18_spec()
    usd = cvd.broadcast_state_dict(synthetic.unet_state_dict(1) if rank == 0 else None, uspec, device)
    rsd = cvd.broadcast_state_dict(synthetic.resnet18_state_dict(2) if rank == 0 else None, rspec, device)
    eng = HipEngine(device, precision=args.dtype, unet_chunk=args.unet_chunk, resnet_chunk=args.resnet_chunk)
    eng.load_unet(usd)
    eng.load_resnet18(rsd)

    # ---- synthetic inputs, resident in HBM before the timed region ----
    B = args.boards
    gen = torch.Generator(device=device)
    gen.manual_seed(1234 + rank)
    x = (torch.randint(0, 256, (B, 3, 256, 256), dtype=torch.uint8, device=device, generator=gen).float() / 255)
    sq = torch.randint(0, 256, (B * 64, 1, 64, 64), dtype=torch.uint8, device=device, generator=gen).float()
    sq /= 255.0

    streams = [torch.cuda.Stream(device), torch.cuda.Stream(device)] if args.overlap else None

    def step():
        if streams is None:
            return eng.unet_forward(x), eng.resnet18_forward(sq)
        cur = torch.cuda.current_stream(device)
        for st in streams:
            st.wait_stream(cur)
        with torch.cuda.stream(streams[0]):
            a = eng.unet_forward(x)
        with torch.cuda.stream(streams[1]):
            b = eng.resnet18_forward(sq)
        for st in streams:
            cur.wait_stream(st)
        return a, b

    for _ in range(args.warmup):
        step()
    cvd.barrier(device)
    torch.cuda.synchronize(device)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    torch.cuda.synchronize(device)
    cvd.barrier(device)
    elapsed = time.perf_counter() - t0
    elapsed = cvd.max_over_ranks(elapsed, device)
    ms_per_step = elapsed / args.steps * 1e3
    value = world * B * args.steps / elapsed

    if rank != 0:
        cvd.barrier(device)                              # leave together with rank 0 (it still profiles and reports)
        cvd.shutdown()
        return

    # ---- roofline of the dominant kernel (the implicit-GEMM conv family), HIP events on the launch stream ----
    conv_ms = conv_n = 0
    conv_flop = 0.0
    all_ms = 0.0
    table = {}
    launches = {}
    per_model = {}
    for model, inp in (("unet", x), ("resnet18", sq)):
        c_ms, c_n, a_ms, entries = eng.profile(model, inp, iters=1)
        conv_ms += c_ms; conv_n += c_n; all_ms += a_ms
        launches[model] = c_n
        per_model[model] = [c_ms, 0.0]
        for e in entries:
            if e["conv"]:
                conv_flop += 2.0 * e["macs"]
                per_model[model][1] += 2.0 * e["macs"]
            t = table.setdefault(f"{model}:{e['name']}", [0.0, 0.0])
            t[0] += e["ms"]; t[1] += 2.0 * e["macs"]
    achieved = conv_flop / (conv_ms * 1e-3) / 1e12
    peak = PEAK_TFLOPS[args.dtype]
    for k, (ms, fl) in table.items():
        log(f"  {k:52s} {ms:9.3f} ms {fl / (ms * 1e-3) / 1e12 if ms > 0 else 0:8.1f} TFLOP/s")
    macs_board = eng.model_macs("unet") + 64 * eng.model_macs("resnet18")
    log(f"  step: {ms_per_step:.2f} ms; event-timed kernels {all_ms:.2f} ms; conv family {conv_ms:.2f} ms over {conv_n} launches")

    # effective chunk sizes from the launch counts (the f32 engine halves the classifier chunk to stay under 4 GiB per tensor)
    n_unet_layers, n_resnet_layers = 22, 19
    assert launches["unet"] % n_unet_layers == 0 and launches["resnet18"] % n_resnet_layers == 0, launches
    eff_unet = B // (launches["unet"] // n_unet_layers)
    eff_resnet = B * 64 // (launches["resnet18"] // n_resnet_layers)
    ub, rb = conv_algorithmic_bytes(args.dtype, eff_unet, eff_resnet)
    assert len(ub) == n_unet_layers and len(rb) == n_resnet_layers
    alg_bytes = (sum(ub) * launches["unet"] / len(ub) + sum(rb) * launches["resnet18"] / len(rb)) / conv_n   # per launch, as traffic

    result = {
        "metric": "boards/sec (UNet 256x256 seg + 64-sq classify)",
        "value": round(value, 2), "unit": "boards/sec",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": args.dtype, "data": "synthetic",
        "config": {"workload": "e2e-cnn b=256/GPU: UNet(3->1, transposed-conv) 256x256 on 256 boards + ResNet-18(1ch,13cls) on "
                               "16384 64x64 squares per step per GPU (BASELINE configs[3]; configs[4] shape at 8 GPUs); inputs resident in HBM",
                   "boards_per_gpu": B, "global_boards_per_step": world * B, "unet_chunk": eff_unet,
                   "resnet_chunk": eff_resnet, "parallelism": f"replicas x{world}, boards sharded, weights RCCL-broadcast once",
                   "gflop_per_board": round(2 * macs_board / 1e9, 3)},
        "e2e_tflops": round(2 * macs_board * value / 1e12, 2),
        "roofline": {"bound": "mfma", "achieved": round(achieved, 2), "peak": round(peak, 1), "unit": "TFLOP/s",
                     "frac": round(achieved / peak, 4),
                     "traffic": pmc_traffic(args.dtype, args.unet_chunk, args.resnet_chunk, launches["unet"], launches["resnet18"]),
                     "algorithmic_bytes": alg_bytes,
                     "kernel": "cv::conv_igemm_kernel + cv::conv3x3_halo_kernel (the conv family, all instantiations)", "launches_per_step": conv_n,
                     "avg_launch_ms": round(conv_ms / max(conv_n, 1), 4), "algorithmic_gflop_per_step": round(conv_flop / 1e9, 2),
                     "by_model": {m: {"achieved": round(fl / (ms * 1e-3) / 1e12, 2), "frac": round(fl / (ms * 1e-3) / 1e12 / peak, 4)}
                                  for m, (ms, fl) in per_model.items()},       # north_star: >= 50 % on the UNet conv stages
                     "mfma_products_per_mac": MFMA_PER_MAC[args.dtype],
                     "mfma_issued_tflops": round(achieved * MFMA_PER_MAC[args.dtype], 2),
                     "frac_of_raw_mfma_peak": round(achieved * MFMA_PER_MAC[args.dtype] / (157.3 if args.dtype == "f32" else 2500.0), 4),
                     "traffic_unit": "HBM bytes per conv launch (rocprofv3 PMC, profiles/r01_pmc_traffic.json); algorithmic_bytes = compulsory bytes per launch"},
    }
    if world == 1 and not args.no_cpu_baseline:
        nchk = min(B, 64)
        base, parity = cpu_baseline(x[:nchk].cpu(), sq[:nchk * 64].cpu(), out[0][:nchk].cpu(), out[1][:nchk * 64].cpu(),
                                    budget_s=args.cpu_budget)
        result["cpu_baseline"] = base
        result["parity_vs_oracle"] = parity
        try:
            result["pipeline_e2e"] = pipeline_e2e(args.dtype)
        except Exception as exc:                                          # extra figure only; never hides the headline
            result["pipeline_e2e"] = {"error": repr(exc)}
    print(json.dumps(result), flush=True)
    cvd.barrier(device)
    cvd.shutdown()


if __name__ == "__main__":
    main()
