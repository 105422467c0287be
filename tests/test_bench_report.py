"""Host logic of bench.py that needs no GPU: the multi-GPU part of the bench line (exchange_report) on synthetic per-rank numbers, and the
selection of the counter files the line may quote (recorded_traffic / recorded_fft_bound: only files stamped with the running build's source
hash, per precision)."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
import bench  # noqa: E402


def test_exchange_report_hidden_and_exposed_exchange():
    """rs[rank] = [ms per step, pack, Legendre, FFT, exchange, bytes sent per step, exchanges per step].  8 tasks of TCo1279: 7 peers,
    8 exchanges per step (2 directions x 4 batches), 2 x 7 x 0.96 GB per step."""
    base = [52.0, 1.5, 22.0, 20.0, 25.0, 13.44e9, 8.0]
    rs = np.array([base] * 8, dtype=float)
    rs[3, 0], rs[3, 3] = 55.0, 23.0  # the slowest rank
    out = bench.exchange_report(rs, 8, 8, 1, 20)
    assert out["exchange_ms_per_step"] == 25.0
    # slowest rank: phases 1.5 + 22 + 23 + 25 = 71.5 against a 55 ms step: 16.5 of the 25 ms of exchange are hidden
    assert abs(out["overlap_frac"] - 16.5 / 25.0) < 1e-12
    rp = out["rank_phase_ms_per_step"]
    assert rp["ms_per_step"] == {"min": 52.0, "max": 55.0, "mean": 52.375} and rp["fft"]["max"] == 23.0 and rp["legendre_mfma"]["min"] == 22.0
    ex = out["exchange"]
    assert ex["links_used_per_rank"] == 7 and ex["slowest_rank"] == 3 and ex["exchanges_per_step"] == 8.0
    assert abs(ex["bytes_per_link_per_step"] - 13.44e9 / 7) < 1 and abs(ex["bytes_per_link_and_direction_per_exchange"] - 13.44e9 / 8 / 7) < 1
    assert abs(ex["achieved_GBps_per_link"] - 13.44 / 7 / 25e-3) < 1e-6  # 76.8 GB/s: a link at its peak
    # fully exposed exchange (sequential: the step is the sum of the phases) and a job without exchange time
    seq = np.array([[68.5, 1.5, 22.0, 20.0, 25.0, 1e9, 2.0]] * 2)
    assert bench.exchange_report(seq, 2, 2, 1, 5)["overlap_frac"] == 0.0
    none = np.array([[40.0, 1.0, 20.0, 19.0, 0.0, 0.0, 0.0]] * 2)
    assert bench.exchange_report(none, 2, 1, 2, 5)["overlap_frac"] is None
    json.dumps(out)  # the block goes into the JSON line as it is


def test_counter_files_are_quoted_only_for_the_running_build(tmp_path, monkeypatch):
    """profiles/*_pmc_traffic[_fp32].json and *_pmc_fft[_fp32].json carry the source hash of the build they were taken on; a line of another
    build, workload or precision must not quote them."""
    prof = tmp_path / "profiles"
    prof.mkdir()
    k64 = {"emi_f64::k_leg_inv": {"hbm_bytes_per_launch": 2.0e11}, "emi_f64::k_leg_dir": {"hbm_bytes_per_launch": 4.0e11}}
    k32 = {"emi_f32::k_leg_inv": {"hbm_bytes_per_launch": 0.8e11}, "emi_f32::k_leg_dir": {"hbm_bytes_per_launch": 1.6e11}}
    (prof / "x_pmc_traffic.json").write_text(json.dumps({"source_hash": "abc", "kernels": k64}))
    (prof / "x_pmc_traffic_fp32.json").write_text(json.dumps({"source_hash": "abc", "kernels": k32}))
    fb = {"source_hash": "abc", "simd_issue_share": {"any": 0.8}, "wave_life_share": {"wait_any": 0.4}, "fft_ms_per_pair": 140.0}
    (prof / "x_pmc_fft.json").write_text(json.dumps(dict(fb, precision=8)))
    (prof / "x_pmc_fft_fp32.json").write_text(json.dumps(dict(fb, precision=4, fft_ms_per_pair=101.0)))
    monkeypatch.setattr(bench, "__file__", str(tmp_path / "bench.py"))
    t, src = bench.recorded_traffic(1279, 137, 10, 8, 1, "abc")
    assert t == 3.0e11 and src == "x_pmc_traffic.json"
    t, src = bench.recorded_traffic(1279, 137, 10, 4, 1, "abc")
    assert t == 1.2e11 and src.startswith("x_pmc_traffic_fp32.json") and "upper bound" in src
    assert bench.recorded_traffic(1279, 137, 10, 8, 1, "other build")[0] is None
    assert bench.recorded_traffic(399, 137, 4, 8, 1, "abc")[0] is None and bench.recorded_traffic(1279, 137, 10, 8, 8, "abc")[0] is None
    assert bench.recorded_fft_bound(1279, 137, 10, 8, 1, "abc")["fft_ms_per_pair_under_profiler"] == 140.0
    assert bench.recorded_fft_bound(1279, 137, 10, 4, 1, "abc")["fft_ms_per_pair_under_profiler"] == 101.0
    assert bench.recorded_fft_bound(1279, 137, 10, 8, 1, "zzz")["bound"] is None
    assert bench.recorded_fft_bound(399, 137, 4, 8, 1, "abc") is None
