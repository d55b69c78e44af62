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

CHILD_WAIT_S = 420


def _log_tail(out):
    try:
        return open(out + ".log").read()[-4000:]
    except OSError as e:
        return f"(no log: {e})"


@pytest.fixture(scope="module")
def report():
    p, out = RCCL_CHILD["proc"], RCCL_CHILD["out"]
    if p is None:
        pytest.skip("child not started (run with -m gpu on a GPU box)")
    # the child has been running since pytest_configure (beside the first tests); it needs about a minute.  Well inside the driver's
    # 900 s limit for the whole GPU suite a hung child becomes a FAILING TEST with its log, not a killed run
    t0 = time.time()
    while p.poll() is None and time.time() - t0 < CHILD_WAIT_S:
        time.sleep(1.0)
    if p.poll() is None:
        import signal

        try:
            os.killpg(p.pid, signal.SIGKILL)  # the child leads its own process group (conftest: start_new_session), ranks included
        except OSError:
            p.kill()
        p.wait(timeout=30)
        pytest.fail(f"tests/rccl_child.py still running after {CHILD_WAIT_S} s; killed.  Its log:\n" + _log_tail(out))
    assert os.path.exists(out), _log_tail(out)
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
    # the line explains its own MAX-over-ranks time: per-rank elapsed / barrier waits / workload / dominant stage, who was slowest
    pr = b["per_rank"]
    assert [p["rank"] for p in pr] == [0, 1] and b["slowest_rank"] in (0, 1) and b["fastest_rank"] in (0, 1) and b["skew_ms"] >= 0.0
    for p in pr:
        assert p["ms_per_step"] > 0 and p["V_mean"] > 0 and p["R_mean"] > 0 and p["instances_listed_mean"] > 0
        assert p["render_backward_avg_ms"] > 0 and p["opening_barrier_wait_ms"] >= 0 and p["start_offset_ms"] >= 0
    assert abs(max(p["ms_per_step"] for p in pr) - b["ms_per_step"]) < 1e-3  # the headline IS the slowest rank's time
    assert pr[0]["R_mean"] != pr[1]["R_mean"]  # two different partitions
