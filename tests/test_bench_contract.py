"""bench.py: the parts of the driver contract that can be checked without a GPU (the bench itself needs a HIP device and says so),
the CPU-baseline worker on a small budget, the core count under a cgroup quota, and the JSON line of a GPU run (marked gpu)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def test_refuses_to_run_without_a_hip_device():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a HIP device is present")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "0"], capture_output=True, text=True, timeout=300)
    assert p.returncode == 2 and "no CPU fallback" in p.stderr


def test_available_cores_respects_affinity_and_quota():
    cores, logical = bench._available_cores()
    assert 1 <= cores <= logical
    assert cores <= len(os.sched_getaffinity(0))


def test_cpu_worker_counts_oracle_steps():
    """The timed leg of cpu_baseline: one oracle environment for a fraction of a second (Avoiding: random policy)."""
    from d3il_amd.controllers.offline_ik import offline_ik
    from d3il_amd.kinematics import UrdfChain
    from d3il_amd.model import blob as blob_mod
    js = blob_mod.load_json("avoiding")
    c, tc = js["controller"], js["task_const"]
    q, _, _ = offline_ik(UrdfChain(js["urdf_chain"]), c["default_qpos"], list(tc["init_end_eff_pos"]) + list(tc["init_end_eff_quat"]),
                         np.array(c["joint_pos_min"]), np.array(c["joint_pos_max"]))
    n, t = bench._cpu_worker("avoiding", bytes(blob_mod.pack(js)), q, None, 0.3, 1)
    assert n >= 10 and 0.25 < t < 5.0


def test_algorithmic_bytes_follow_the_survey():
    # 2 S + A + O + F with S = the f64 state column + flags / counter words
    assert bench.ALG_BYTES["avoiding"] == 2 * (42 * 8 + 8) + 56 + 8 + 4 == 756
    assert bench.ALG_BYTES["pushing"] == 1212 and bench.ALG_BYTES["stacking"] == 1212 and bench.ALG_BYTES["sorting"] == 2 * (129 * 8 + 8) + 116


@pytest.mark.gpu
def test_json_line_of_a_short_run():
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "5", "--warmup", "2", "--preroll", "10", "--envs", "512"],
                       capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-2000:]
    line = json.loads(p.stdout.strip().splitlines()[-1])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
                "roofline", "cpu_baseline"):
        assert key in line, key
    assert line["metric"] == "env-steps/s" and line["n_gpus"] == 1 and line["steps"] == 5 and line["scaling"] == "weak" and line["dtype"] == "f64"
    assert line["vs_baseline"] is None and "workload" in line["config"] and line["config"]["finite"]
    r, c = line["roofline"], line["cpu_baseline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12 and r["kernel_ms"] > 0
    assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 0 and c["single_core_value"] > 0
    assert abs(line["value"] - 512 * 5 / (line["ms_per_step"] * 5e-3)) / line["value"] < 1e-6
