import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


RCCL_CHILD = {"proc": None, "out": None}


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # tests/test_rccl_gpu.py needs two other programs run (bench.py starting its own ranks; a world-size-1 RCCL group).  A process
    # that has initialised the GPU must not start programs on this pool, so the child is started HERE, before anything in this
    # process touches HIP (device_count() does not), and the test only reads what it wrote.
    import subprocess
    import tempfile

    import torch

    expr = getattr(config.option, "markexpr", "") or ""
    if "not gpu" in expr or "gpu" not in expr or os.environ.get("GSR_NO_RCCL_CHILD") or torch.cuda.device_count() == 0:
        return
    out = os.path.join(tempfile.mkdtemp(prefix="gsr_rccl_"), "result.json")
    log = open(out + ".log", "w")
    RCCL_CHILD["proc"] = subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "rccl_child.py"), out], stdout=log,
                                          stderr=subprocess.STDOUT, start_new_session=True)
    RCCL_CHILD["out"] = out


def pytest_sessionfinish(session, exitstatus):
    """GPU sessions leave the worst |err| / tolerance of every gradient tensor (two-stage bar, tests/helpers.py) under gpurun_out/."""
    try:
        import json

        import helpers
        if helpers.WORST:
            os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
            json.dump(dict(worst_ratio=helpers.WORST, GRAD_COND=helpers.GRAD_COND, CHAIN_RTOL=helpers.CHAIN_RTOL, exitstatus=int(exitstatus)),
                      open(os.path.join(ROOT, "gpurun_out", "grad_worst.json"), "w"), indent=1, sort_keys=True)
    except Exception:  # noqa: BLE001 -- reporting only
        pass


def pytest_unconfigure(config):
    p = RCCL_CHILD["proc"]
    if p is not None and p.poll() is None:
        try:
            p.wait(timeout=60)
        except Exception:  # noqa: BLE001
            p.kill()


def pytest_collection_modifyitems(config, items):
    # GPU tests are skipped (not failed) when no device is present and -m gpu was not requested explicitly
    import torch

    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)
