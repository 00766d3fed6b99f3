"""bench.py's default line on the device: the contract's keys, the instrumented roofline pass and the HBM table must come out
(a change in an op's calling convention that the accounting hooks do not follow would otherwise only show at round end)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_bench_default_line_has_roofline_and_hbm_table():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--no-cpu-baseline"],
                       capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
    d = json.loads(line)
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config", "roofline"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["value"] > 0 and d["config"]["workload"].startswith("fsf_nuscenes_10sweep")
    roof = d["roofline"]
    assert roof["bound"] == "mfma" and 0 < roof["frac"] < 1 and roof["peak"] > roof["achieved"] > 0
    # the in-situ table (what is quoted) and, under `debug`, the cache-warm isolated replay it replaced as the headline of this block
    replay = roof["debug"]["hbm_isolated_replay"]
    for k in ("linear_norm_act", "seg_reduce", "sir_input", "rows_to_planes"):
        assert replay[k]["ms_per_step"] > 0 and 0 < replay[k]["frac_of_hbm_peak"] < 1
    assert "hbm" not in roof and "frac_of_fp32_pipe_peak" not in json.dumps(roof)
    for k in ("linear_norm_act", "seg_reduce", "sir_input"):
        assert 0 < roof["hbm_in_situ"][k]["frac_of_hbm_peak"] < 1
