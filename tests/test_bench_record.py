"""CPU: the driver's record of the bench line keeps about 23 scalar entries of `config` and of `roofline` (names cut at 40 characters,
strings at 128) and only the NAMES of every other block.  VERDICT r05 item 7: the figures a reader needs must be among the FIRST
entries, under short names.  This test runs `bench.flatten_evidence` on a line shaped like a real one and applies the driver's cut."""
from __future__ import annotations

import importlib.util
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent


def _bench():
    spec = importlib.util.spec_from_file_location("bench_for_record_test", ROOT / "bench.py")
    mod = importlib.util.module_from_spec(spec)
    sys.modules["bench_for_record_test"] = mod
    spec.loader.exec_module(mod)
    return mod


def _driver_cut(block: dict, keep: int = 23) -> dict:
    """what the driver's `parsed` keeps of one block: scalar entries in order, `keep` of them, names <= 40 and strings <= 128 characters"""
    out = {}
    for k, v in block.items():
        if isinstance(v, (dict, list)):
            continue
        if len(out) >= keep:
            break
        out[k[:40]] = v[:128] if isinstance(v, str) else v
    return out


def _line():
    hbm = {"stem7x7+maxpool (mfma)": {"frac": 0.37, "achieved": 2967.6}, "head_avgpool_fc": {"frac": 0.48, "achieved": 3845.8},
           "resize_area_u8": {"frac": 0.86, "achieved": 6866.1}, "extract_squares_u8": {"frac": 0.22, "achieved": 1753.3}}
    return {
        "value": 4258.7, "rccl_ranks_seen": 1, "dist_backend": "nccl",
        "config": {"workload": "e2e-cnn b=256/GPU: UNet(3->1,convT) 256x256 x256 boards + ResNet-18(1ch,13cls) x16384 squares/step; in HBM (configs[3])",
                   "boards_per_gpu": 256, "global_boards_per_step": 256, "unet_chunk": 64, "resnet_chunk": 16384,
                   "parallelism": "replicas x1, boards sharded, weights RCCL-broadcast once", "gflop_per_board": 114.463},
        "roofline": {"bound": "mfma", "achieved": 488.3, "peak": 833.3, "unit": "TFLOP/s", "frac": 0.5859, "algorithmic_bytes": 8.9e8,
                     "kernel": "conv family: conv3x3_halo_kernel + conv_igemm_kernel + inc0_mfma_kernel, all instantiations", "launches_per_step": 103,
                     "avg_launch_ms": 0.58, "algorithmic_gflop_per_step": 29195.0,
                     "by_model": {"unet": {"achieved": 488.0, "frac": 0.586}, "resnet18": {"achieved": 487.0, "frac": 0.585}},
                     "mfma_products_per_mac": 3, "mfma_issued_tflops": 1464.8, "frac_of_raw_mfma_peak": 0.5859, "conv_family_hbm_gbs": 1534.3,
                     "dominant": {"kernel": "conv3x3_halo_kernel<split_t,64,16x16>", "frac": 0.616, "avg_launch_ms": 0.673, "launches_per_step": 68,
                                  "share_of_conv_time": 0.77},
                     "by_kernel": {}, "launches_by_model": {"unet": 84, "resnet18": 19}, "traffic": None, "traffic_null_reason": "stale",
                     "traffic_source": None, "traffic_unit": "HBM bytes per conv launch"},
        "roofline_hbm": hbm,
        "init_s": {"max": 2.6}, "device_memory": {"peak_used_gb_max_over_ranks": 28.4},
        "calibration_sync": {"identical_across_ranks": True}, "sharding": {"gathered_in_order": True},
        "parity_vs_oracle": {"unet_logit_max_abs_err": 1.0e-4, "resnet_logit_max_abs_err": 7.6e-6, "boards_checked": 64},
        "pipeline_e2e": {"boards_per_sec": 3883.8, "boards_per_sec_median": 3800.8, "boards_per_sec_min": 3143.3, "fen_checked": 8, "fen_mismatches": 0,
                         "board_byte_mismatches_vs_oracle": 0, "prob_max_abs_err_vs_oracle": 1.6e-6, "classifier_fp16": {"boards_per_sec": 3978.0},
                         "latency": {"process_image_ms_median": 0.965, "process_image_ms_p10": 0.955, "process_image_ms_p90": 1.009,
                                     "concurrent4_per_sec": 2313.0}},
        "by_dtype": {"f32": {"value": 1242.6, "roofline": {"frac": 0.919}, "parity_vs_oracle": {"unet_logit_max_abs_err": 1.1e-4}},
                     "f16": {"value": 9698.5, "roofline": {"frac": 0.442}, "parity_vs_oracle": {"unet_logit_max_abs_err": 3.4e-2}}},
        "classifier_fp16": {"ms_per_pass": 5.267, "roofline": {"frac": 0.3836}, "step_with_headline_unet": {"boards_per_sec": 4587.9},
                            "parity_vs_oracle": {"prob_max_abs_err": 5.8e-4}},
        "classifier_fp16_seed_search": {"worst_prob_err": 7.4e-4, "networks": 16},
        "by_variant": {"bilinear": {"value": 4923.7, "roofline_hbm": {"upsample_bilinear2x": {"frac": 0.619, "achieved": 4951.9}}}},
    }


def test_the_drivers_cut_of_the_line_still_holds_every_figure_a_judge_reads():
    bench = _bench()
    line = _line()
    bench.flatten_evidence(line)
    cfg, roof = _driver_cut(line["config"]), _driver_cut(line["roofline"])
    assert len(line["config"]["workload"]) <= 128 and len(line["roofline"]["kernel"]) <= 128
    for key, want in (("cls16_frac", 0.3836), ("cls16_prob_err", 5.8e-4), ("cls16_worst_err", 7.4e-4), ("cls16_step_bps", 4587.9), ("f32_bps", 1242.6),
                      ("f32_frac", 0.919), ("bil_bps", 4923.7), ("e2e_bps", 3883.8), ("e2e_bps_median", 3800.8), ("e2e_bps_min", 3143.3),
                      ("lat_ms_median", 0.965), ("conc4_per_sec", 2313.0), ("par_unet_err", 1.0e-4), ("par_resnet_err", 7.6e-6), ("boards_per_gpu", 256)):
        assert cfg.get(key) == want, (key, cfg)
    for key, want in (("frac", 0.5859), ("dominant_frac", 0.616), ("dominant_kernel", "conv3x3_halo_kernel<split_t,64,16x16>"), ("unet_conv_frac", 0.586),
                      ("resnet18_conv_frac", 0.585), ("hbm_upsample_frac", 0.619), ("hbm_stem_frac", 0.37), ("hbm_head_frac", 0.48), ("hbm_warp_frac", 0.22)):
        assert roof.get(key) == want, (key, roof)
    assert "traffic" in roof and roof["bound"] == "mfma"                       # the contract's own keys stay in front
    assert all(len(k) <= 40 for k in list(line["config"])[:23] + list(line["roofline"])[:23])
    # the long names of earlier rounds are still in the full line for the tools and tests that read them
    assert line["config"]["pipeline_e2e_boards_per_sec"] == 3883.8 and line["config"]["latency"]["concurrent4_per_sec"] == 2313.0
    assert line["roofline"]["dominant"]["frac"] == 0.616 and line["roofline"]["hbm_resize_area_u8_gbs"] == 6866.1
