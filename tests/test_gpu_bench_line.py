"""The driver's bench line: `python bench.py --gpus 1 --steps K --warmup W` prints ONE JSON line whose `roofline` and
`cpu_baseline` objects carry BOTH halves of BASELINE.json's metric (batched env.step board-steps/s + beam node-expansions/s at
width 20 / depth 30) as scalars -- the driver's record keeps those two objects -- and whose last key repeats the headline."""
import json
import os
import subprocess
import sys

import pytest

from conftest import REPO

pytestmark = pytest.mark.gpu


def test_bench_line_carries_both_halves_of_the_metric():
    out = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "1", "--steps", "20", "--warmup", "5",
                          "--no-extra", "--cpu-seconds", "2"], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, "exactly one JSON line on stdout"
    r = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline", "cpu_baseline"):
        assert k in r, k
    assert r["steps"] == 20 and r["warmup"] == 5 and r["n_gpus"] == 1 and r["scaling"] == "weak" and r["vs_baseline"] is None
    assert "workload" in r["config"] and "model" not in r["config"]
    rf, cb = r["roofline"], r["cpu_baseline"]
    # frac / achieved / peak are the HBM-algorithmic figures SURVEY 8(d) defines; `bound` names whichever limit is nearer, and the
    # VALU-issue block beside it carries the instruction count, the issue model and the counter-based busy fraction
    assert rf["bound"] in ("hbm", "valu_issue") and rf["peak"] == 8000.0 and rf["unit"] == "GB/s" and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-12
    assert 0.2 < rf["frac"] < 1.0 and abs(r["value"] - 1048576 / (r["ms_per_step"] * 1e-3)) / r["value"] < 1e-6
    vi = rf["valu_issue"]
    assert vi["bound"] == "valu_issue" and 300 < vi["valu_instructions_per_board"] < 420 and 0.3 < vi["valu_busy_frac_by_counter"] <= 1.05
    assert abs(vi["valu_busy_frac_by_counter"] - vi["valu_busy_us_per_step_by_counter"] / rf["kernel_us"]) < 1e-9
    assert (rf["bound"] == "valu_issue") == (vi["valu_busy_frac_by_counter"] > rf["frac"])
    # the launch form: two independent sub-batch chains per step by default, equal to the single launch, which is timed beside it
    assert r["config"]["chains"] == 2 and "sub-batch" in r["config"]["launch"] and rf["chains_equal_single_launch"] is True
    sl = rf["single_launch"]
    assert 0.2 < sl["frac"] < 1.0 and sl["kernel_us"] > 0 and sl["kernel_us_plain_launches"] > 0
    # config 4's env kernel has a roofline block of its own
    rr = r["rollout"]["roofline"]
    assert rr["kernel"] == "rollout_step_kernel" and rr["algorithmic_bytes_per_launch"] == 65536 * 132 and 0.02 < rr["frac"] < 1.0
    assert abs(rr["frac"] - rr["achieved"] / rr["peak"]) < 1e-12
    # the beam half, where the record keeps it: flat scalars (and the nested copies)
    assert rf["beam_value"] == r["beam"]["value"] == rf["beam"]["value"] and rf["beam_value"] > 1e10
    assert rf["beam_ms_per_batch_decision"] > 0 and 0.3 < rf["beam_valu_issue_frac"] < 1.0 and 1500 < rf["beam_expansions_per_decision"] < 1900
    assert r["beam"]["value"] <= r["beam"]["value_best_of_3_batches"]                      # `value` is the mean of the batches
    assert rf["evaluation_seconds"] == r["evaluation"]["seconds"] and rf["evaluation_same_games_without_helpers"] is True
    assert rf["evaluation_moves"] > 5_000_000 and r["evaluation"]["same_games_with_action_stream"] is True
    assert cb["kind"] == "port" and cb["cores"] >= 1 and cb["value"] > 1e5 and cb["beam_value"] == r["beam"]["cpu_baseline"]["value"] > 1e5
    assert cb["beam_one_thread_value"] > 1e5 and cb["config1_drop_in_steps_per_s"] > 100 and cb["config1_reference_style_python_steps_per_s"] > 100
    assert list(r)[-1] == "headline" and r["headline"]["beam_expansions_per_s"] == r["beam"]["value"]
    assert r["headline"]["board_steps_per_s"] == r["value"] and r["headline"]["evaluation_seconds"] == r["evaluation"]["seconds"]
