"""bench.py keeps the driver's contract: ONE JSON line with the agreed keys, BASELINE.json's metric,
a roofline object and a cpu_baseline object; without a GPU it fails loudly instead of measuring a
fallback."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, timeout):
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True,
                          timeout=timeout, cwd=ROOT)


def test_bench_refuses_to_run_without_a_gpu():
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    out = _run(["--workload", "c2_n1008", "--steps", "1", "--warmup", "0"], 300)
    assert out.returncode != 0
    assert not any(line.startswith("{") for line in out.stdout.splitlines())   # no number from a fallback


@pytest.mark.gpu
def test_bench_json_contract(gpu):
    out = _run(["--workload", "c2_n1008", "--steps", "2", "--warmup", "1"], 900)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [line for line in out.stdout.splitlines() if line.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    base = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    assert d["metric"] == base["metric"] and d["unit"] == "syndromes/s"
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1
    assert d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["dtype"] == "f64" and d["data"] == "synthetic" and "workload" in d["config"] and "model" not in d["config"]
    assert d["value"] > 0 and d["ms_per_step"] > 0
    assert abs(d["value"] - d["config"]["global_batch"] / (d["ms_per_step"] * 1e-3)) <= 1e-6 * d["value"]
    r = d["roofline"]
    assert r["bound"] in ("hbm", "mfma", "infinity_cache") and r["unit"] in ("GB/s", "TFLOP/s") and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9 and "traffic" in r
    c = d["cpu_baseline"]
    assert c["kind"] in ("reference", "port") and c["cores"] >= 1 and c["value"] > 0 and c["unit"] == "syndromes/s" and c["sample"]
    assert c["gpu_matches_oracle_on_sample"] is True and c["reference_faithful_1_thread"]["cores"] == 1
    # N = 1 runs the multi-GPU path's own code (sharding.batchdecode_sharded) in its degenerate form
    assert d["config"]["mode"] == "scatter" and d["exchange"]["scatter_bytes_per_peer"] == 0
