"""bench.py's N > 1 flow on CPU: two gloo ranks, the library replaced by bench.py's own --mock stand-in
(no kernel runs, nothing is measured).  Checks the plumbing the 8-GPU run depends on - the weighted
re-deal agreed by all_reduce, the row gather, the max-over-ranks timing - and the shape of the JSON
line, including the fields that explain a multi-GPU result."""
import json
import os
import socket
import subprocess
import sys

from conftest import ROOT


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run(n, extra=()):
    cmd = [sys.executable]
    if n > 1:
        cmd += ["-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
                "--master-addr", "127.0.0.1", "--master-port", str(_free_port())]
    cmd += [os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "2", "--warmup", "2", "--width", "320",
            "--spp", "16", "--backend", "gloo", "--mock", *extra]
    res = subprocess.run(cmd, check=True, timeout=300, cwd=ROOT, stdout=subprocess.PIPE)
    lines = [l for l in res.stdout.decode().splitlines() if l.startswith("{")]
    assert len(lines) == 1                      # rank 0 prints ONE line
    return json.loads(lines[0])


def test_two_rank_json_shape():
    out = _run(2, ["--n1-value", "100.0"])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline"):
        assert key in out
    assert out["mock"] is True and out["n_gpus"] == 2 and out["steps"] == 2 and out["scaling"] == "strong"
    assert out["backend"] == "gloo" and out["dist_world"] == 2
    assert len(out["kernel_ms_per_rank"]) == 2 and all(k > 0 for k in out["kernel_ms_per_rank"])
    assert len(out["gather_ms_per_rank"]) == 2
    assert sum(out["rows_per_rank_initial"]) == 180 and sum(out["rows_per_rank"]) == 180
    assert out["rows_per_rank_initial"] == [90, 90]
    assert out["config"]["rows_per_rank"] == out["rows_per_rank"]
    assert abs(out["efficiency_vs_n1"] - out["value"] / 200.0) < 1e-3
    assert "cpu_baseline" not in out and "builds" not in out          # N = 1 legs only
    r = out["roofline"]
    # the roofline is a fraction of a physical peak: executed matrix flop against the dense f16 peak
    assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and r["peak"] == 2500.0
    assert 0.0 < r["frac"] <= 1.0 and abs(r["frac"] - r["achieved"] / 2500.0) < 1e-3
    assert r["tiles_per_scene"] == 16                                                        # 488 spheres = 16 tiles
    assert r["matrix_flop_per_launch"] == r["wave_scans_per_launch"] * r["tiles_multiplied_per_wave_scan"] * 4 * 32768
    assert 0 < r["tiles_multiplied_per_wave_scan"] <= 16
    assert r["traffic"] is None and r["traffic_ratio"] is None
    for key in ("valu_issue_busy", "mfma_busy", "lane_occupation", "vector_instr_per_wave_scan", "algorithmic_vs_vector_peak"):
        assert key in out                                              # flat keys, null without a counter pass
    assert out["valu_issue_busy"] is None and out["algorithmic_vs_vector_peak"] > 0
    assert out["config"]["chunk_spp"] == 8 and "parity_bar" in out["config"] and out["config"]["kernel"] == "exact"
    assert "gather_verified" in out


def test_one_rank_mock_line():
    out = _run(1)
    assert out["n_gpus"] == 1 and "backend" not in out and out["config"]["rows_per_rank"] == [180]
