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
    # B_alg = 2 S + A + O + F per env step - Avoiding 704 (S 328, A 28, O 16, F 4), Pushing 1152 (S 536, A 28, O 40, F 12),
    # Sorting-4 1620 (S 760, A 28, O 64, F 8), Stacking 1228 (S 552, A 32, O 80, F 12); `roofline.achieved` is built from THESE
    assert bench.ALG_BYTES == {"avoiding": 2 * 328 + 28 + 16 + 4, "pushing": 2 * 536 + 28 + 40 + 12, "sorting": 2 * 760 + 28 + 64 + 8,
                               "stacking": 2 * 552 + 32 + 80 + 12,
                               # Aligning (not in SURVEY 8d's list): the same formula on its nq 16 / nv 15 model, S = 8 (16 + 15) + 168 + 64, A 28, O 68 (17 f32), F 12
                               "aligning": 2 * 480 + 28 + 68 + 12,
                               # Inserting: nq 30 / nv 27, S = 8 (30 + 27) + 168 + 16, A 28, O 44 (11 f32), F 8
                               "inserting": 2 * 640 + 28 + 44 + 8}
    assert bench.ALG_BYTES == {"avoiding": 704, "pushing": 1152, "sorting": 1620, "stacking": 1228, "aligning": 1068, "inserting": 1360}
    # what the implementation's state column moves is reported next to it and is never smaller
    assert all(bench.IMPL_BYTES[k] >= bench.ALG_BYTES[k] for k in bench.ALG_BYTES)


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
    cfg = line["config"]
    assert cfg["sub_batches"] * cfg["envs_per_launch"] == 512 and cfg["sub_batches"] == 4      # the default: four sub-batches on four streams
    assert r["algorithmic_bytes_per_env_step"] == 704 and r["algorithmic_bytes_per_launch"] == 704 * cfg["envs_per_launch"] == 704 * r["envs_per_launch"]
    # `achieved` is per env step at chip level (comparable between sub-batch counts); the per-launch figure of one sub-batch stands next to it
    assert r["algorithmic_bytes_per_step"] == 704 * 512 and abs(r["achieved"] - 704 * 512 / (line["ms_per_step"] * 1e-3) / 1e9) / r["achieved"] < 1e-9
    assert abs(r["achieved_per_launch"] - r["algorithmic_bytes_per_launch"] / (r["kernel_ms"] * 1e-3) / 1e9) / r["achieved_per_launch"] < 1e-9
    assert abs(r["frac_per_launch"] - r["achieved_per_launch"] / r["peak"]) < 1e-12


@pytest.mark.gpu
def test_sub_batches_do_the_same_work_as_one_batch():
    """--sub-batches 4 (the default) and --sub-batches 1 step the same environments with the same policy stream (Philox counters and context ids go by the
    global environment index): the integer episode tally of a steady-state run is identical."""
    out = {}
    for sb in (1, 4):
        p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "40", "--warmup", "2", "--preroll", "255", "--envs", "512", "--no-cpu-baseline",
                            "--sub-batches", str(sb)], capture_output=True, text=True, timeout=900)
        assert p.returncode == 0, p.stderr[-2000:]
        line = json.loads(p.stdout.strip().splitlines()[-1])
        assert line["config"]["sub_batches"] == sb and line["config"]["finite"] and line["config"]["flagged_envs"]["solver_fail"] == 0
        out[sb] = (line["config"]["episodes_finished_all_ranks"], line["config"]["episodes_success_all_ranks"])
    assert out[1] == out[4] and out[1][0] > 0, out


@pytest.mark.gpu
def test_gpus_2_self_spawn_runs_two_ranks():
    """`python bench.py --gpus 2` without a launcher re-launches itself under torch.distributed.run.  On the one-GPU box
    both ranks are pinned to GPU 0 and rendezvous over gloo (RCCL needs distinct GPUs); shards, Philox offsets, the tally reduction and the
    max-over-ranks timing are the code path of the 8-GPU run."""
    env = dict(os.environ, D3IL_BENCH_FORCE_DEVICE="0", D3IL_DIST_BACKEND="gloo")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "6", "--warmup", "2", "--preroll", "270", "--envs", "256"],
                       capture_output=True, text=True, timeout=900, env=env)
    assert p.returncode == 0, p.stdout[-1500:] + p.stderr[-2500:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, "rank 0 prints ONE JSON line"
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["config"]["parallelism"] == "env-shard x2" and line["config"]["envs_per_gpu"] == 256
    assert abs(line["value"] - 2 * 256 * 6 / (line["ms_per_step"] * 6e-3)) / line["value"] < 1e-6
    assert "cpu_baseline" not in line                                  # rank 0 at N = 1 only
    assert line["config"]["episodes_finished_all_ranks"] >= line["config"]["episodes_finished_rank0"] > 0
    assert "torch.distributed" in line["config"]["metric_reduction"]


@pytest.mark.gpu
def test_gpus_2_under_nccl_reaches_rccl_or_exits_with_the_self_check_code():
    """Multi-GPU readiness without a node: `bench.py --gpus 2` under the nccl backend - the backend of the driver's 8-GPU run - with
    D3IL_ALLOW_REDUCTION_FALLBACK unset.  On a box with two GPUs the line must say that the library's own RCCL communicator saw both ranks (rccl_ranks == 2,
    reduction by d3il_reduce_metrics); on the one-GPU box both ranks land on device 0, RCCL refuses the duplicate GPU and every rank leaves with the documented
    exit code 3 and ONE message on rank 0 - no hang, no JSON line that looks like a scaling result."""
    import torch
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "D3IL_ALLOW_REDUCTION_FALLBACK", "D3IL_DIST_BACKEND", "D3IL_BENCH_FORCE_DEVICE"):
        env.pop(k, None)
    two = torch.cuda.device_count() >= 2
    if not two:
        env["D3IL_BENCH_FORCE_DEVICE"] = "0"
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "6", "--warmup", "2", "--preroll", "270", "--envs", "256"],
                       capture_output=True, text=True, timeout=900, env=env)
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    if two:
        assert p.returncode == 0, p.stdout[-1500:] + p.stderr[-2500:]
        line = json.loads(lines[-1])
        assert line["n_gpus"] == 2 and line["config"]["rccl_ranks"] == 2 and "d3il_reduce_metrics" in line["config"]["metric_reduction"]
    else:
        assert p.returncode != 0 and not lines, p.stdout[-1500:] + p.stderr[-2500:]
        assert "multi-GPU self-check failed" in p.stderr, p.stderr[-2500:]
