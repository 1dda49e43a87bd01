"""`bench.py --gpus N` end to end on a box with ONE GPU (UZ_BENCH_ONE_DEVICE=1: every rank on device 0, barrier and max-over-ranks over gloo):
the N-rank code path -- contiguous DNM shards that take their pile-ups from the whole list's clusters, the staged pipeline per rank, barrier,
reduction, rank 0's one JSON line -- runs and agrees with itself.  It measures nothing (the ranks share the device); the 8-GPU curve needs a node."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("ranks", [2, 3])
def test_n_ranks_on_one_device(ranks, hip_lib):
    env = dict(os.environ, UZ_BENCH_ONE_DEVICE="1")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(ranks), "--dnms", "6000", "--sites", "1500000", "--steps", "2", "--warmup", "1",
                          "--no-cpu", "--feed-dnms", "0", "--no-config5"], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [x for x in out.stdout.splitlines() if x.startswith("{")]
    assert len(lines) == 1  # rank 0 alone prints
    j = json.loads(lines[0])
    assert j["n_gpus"] == ranks and j["scaling"] == "strong" and j["config"]["parallelism"] == "dnm-shard x%d, no collective" % ranks
    assert len(j["ms_per_step_by_rank"]) == ranks and all(x > 0 for x in j["ms_per_step_by_rank"])
    assert j["link"]["result_mismatches_vs_resident"] == 0
    assert abs(j["config"]["dnms_per_gpu"] - 6000 / ranks) <= 1
    assert j["value"] > 0 and j["cpu_baseline"] is None and j["feed"] is None
