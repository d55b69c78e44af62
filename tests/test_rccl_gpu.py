"""The N > 1 launch path and the RCCL entry points, proven on the one-GPU box.

The work happens in tests/rccl_child.py, started by conftest.pytest_configure before this process initialised the GPU; here its
report is read.  Reference: the per-partition launcher auto_train.py:22-51 and the merge scene/ptgs/merge_partitions.py:111-170.
"""
import json
import os
import time

import pytest

from conftest import RCCL_CHILD

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def report():
    p, out = RCCL_CHILD["proc"], RCCL_CHILD["out"]
    if p is None:
        pytest.skip("child not started (run with -m gpu on a GPU box)")
    t0 = time.time()
    while p.poll() is None and time.time() - t0 < 1100:
        time.sleep(1.0)
    assert p.poll() is not None, "tests/rccl_child.py did not finish"
    assert os.path.exists(out), open(out + ".log").read()[-4000:]
    return json.load(open(out))


def test_world_size_one_rccl_group_runs_the_merge_collectives(report):
    assert "rccl_error" not in report, report.get("rccl_error")
    assert report["backend"] == "nccl"
    assert report["gather_equal"] and report["merge_equal_numpy_unique"] and report["eval_merged_equal"]
    assert report["allreduce_max"] == 1.25
    assert report["eval_world_size"] == 1 and report["eval_counts"] == [5000]


def test_bench_starts_its_own_ranks(report):
    """`python bench.py --gpus 2` from a plain shell: two ranks through torch.distributed.run, ONE JSON line, whole-job value."""
    assert "bench_error" not in report, report.get("bench_error")
    assert report["bench_rc"] == 0, report["bench_stderr_tail"]
    assert report["bench_json_lines"] == 1
    b = report["bench"]
    assert b["n_gpus"] == 2 and b["rccl_world_size"] == 2 and b["rccl_ranks_seen"] == [0, 1]
    assert b["collective_backend"] in ("gloo", "nccl")  # gloo when two ranks share the box's single GPU
    assert b["merge"]["world_size"] == 2 and len(b["merge"]["rows_per_rank_after_crop"]) == 2
    assert b["scaling"] == "weak" and b["value"] > 0
