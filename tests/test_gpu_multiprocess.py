"""Multi-process run of the batched Sim classes on the GPU box: the rollout shards of 2 and 3 ranks (gloo, all on cuda:0) must
reduce to exactly the integer count tables - and hence the metrics - of the single-process run."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(world, port):
    worker = os.path.join(ROOT, "tests", "dist_sim_worker.py")
    if world == 1:
        cmd = [sys.executable, worker]
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
               "--master-port", str(port), worker]
    env = dict(os.environ)
    env.pop("RANK", None); env.pop("WORLD_SIZE", None)
    p = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    line = [l for l in p.stdout.splitlines() if l.startswith("RESULT ")][-1]
    return json.loads(line[len("RESULT "):])


def test_sim_metrics_do_not_depend_on_the_number_of_ranks():
    one = _run(1, 0)
    assert one["pushing"]["shard"] == [0, 18] and one["avoiding"]["shard"] == [0, 37] and one["sorting"]["shard"] == [0, 22]
    assert one["stacking"]["shard"] == [0, 10] and one["stacking"]["counts"][-3] >= 8        # the red box reaches the target zone on (almost) every rollout
    assert one["sorting"]["mode_hist"][112] > 0 and sum(one["sorting"]["mode_hist"]) == 22      # some scripted pushes deliver the red cube
    assert one["aligning"]["shard"] == [0, 14] and one["aligning"]["modes"] == [0, 1]
    for world, port in ((2, 29531), (3, 29532)):
        many = _run(world, port)
        assert many["pushing"]["shard"][0] == 0 and many["pushing"]["shard"][1] < 18      # rank 0 owns a proper shard
        assert many["pushing"]["counts"] == one["pushing"]["counts"]
        assert many["avoiding"]["counts"] == one["avoiding"]["counts"]
        assert many["stacking"]["counts"] == one["stacking"]["counts"] and many["stacking"]["shard"][1] < 10
        assert many["stacking"]["successes_1_box"] == one["stacking"]["successes_1_box"]
        assert many["sorting"]["counts"] == one["sorting"]["counts"] and many["sorting"]["mode_hist"] == one["sorting"]["mode_hist"]
        assert many["pushing"]["success_rate"] == one["pushing"]["success_rate"] and many["pushing"]["entropy"] == one["pushing"]["entropy"]
        assert many["avoiding"]["entropy"] == one["avoiding"]["entropy"]
        assert many["aligning"]["counts"] == one["aligning"]["counts"] and many["aligning"]["shard"][1] < 14 and abs(many["aligning"]["mean_distance"] - one["aligning"]["mean_distance"]) < 1e-12
        assert abs(many["pushing"]["mean_distance"] - one["pushing"]["mean_distance"]) < 1e-12
