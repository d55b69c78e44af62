"""Child process of tests/test_rccl_gpu.py (started from conftest.pytest_configure, BEFORE the pytest process touches the GPU:
a process that has initialised HIP must not start other programs on this pool).

1. `python bench.py --gpus 2 ...` from a plain environment (no RANK): bench.py must start its own two ranks through
   torch.distributed.run (the per-partition launcher of the reference, auto_train.py:22-51) and relay rank 0's JSON line.  On a
   one-GPU box the two ranks share the card, so bench.py falls back to gloo for the exchange and says so.
2. a world-size-1 process group on backend "nccl" (= RCCL on ROCm) in THIS process: the collectives of the merge step
   (scene/ptgs/merge_partitions.py:111-170) -- all_gather_into_tensor of the counts and of the padded [maxN, 59] blocks, the
   all_reduce of the evaluation sums, barrier, all_reduce(MAX) of the bench's timing -- run through librccl on device tensors.
Writes one JSON file with what it saw; the test asserts on it.
"""
import json
import os
import subprocess
import sys
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def bench_two_ranks(out):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--config", "cfg1", "--steps", "8", "--warmup", "2",
           "--spinup-ms", "0"]
    p = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    out["bench_rc"] = p.returncode
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    out["bench_json_lines"] = len(lines)
    out["bench_stderr_tail"] = p.stderr[-2000:]
    if lines:
        out["bench"] = json.loads(lines[-1])


def rccl_world_one(out):
    import datetime
    import socket

    import numpy as np
    import torch
    import torch.distributed as dist

    from based_on_point_cloud_partitions_amd import evaluate, merge

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev, timeout=datetime.timedelta(seconds=300))
    out["backend"] = dist.get_backend()
    rng = np.random.default_rng(3)
    block = rng.uniform(-1, 1, (5000, merge.ROW)).astype(np.float32)
    block[10, :3] = block[9, :3]  # an exact-xyz duplicate inside the partition
    dblock = torch.from_numpy(block).to(dev)
    rows, counts = merge.all_gather_blocks(dblock)  # two all_gather_into_tensor calls on device tensors
    out["gather_equal"] = bool(torch.equal(rows, dblock)) and counts == [5000]
    dist.barrier()
    t = torch.tensor([1.25], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    out["allreduce_max"] = float(t.item())
    merged = merge.merge_partitions(dblock, (None, None, None, None))
    _, idx = np.unique(block[:, :3], axis=0, return_index=True)
    out["merge_equal_numpy_unique"] = bool(np.array_equal(merged.cpu().numpy(), block[idx]))
    ev = evaluate.seamless_merge_and_evaluate(dblock, 0, [(None, None, None, None)], [], [], torch.zeros(3, device=dev))
    out["eval_world_size"], out["eval_counts"] = ev["world_size"], ev["counts"]
    out["eval_merged_equal"] = bool(np.array_equal(ev["merged"].cpu().numpy(), block[idx]))
    dist.barrier()
    dist.destroy_process_group()


def main():
    out = {}
    try:
        bench_two_ranks(out)  # first: this process has not touched the GPU yet
    except Exception:  # noqa: BLE001
        out["bench_error"] = traceback.format_exc()
    try:
        rccl_world_one(out)
    except Exception:  # noqa: BLE001
        out["rccl_error"] = traceback.format_exc()
    with open(sys.argv[1] + ".tmp", "w") as f:
        json.dump(out, f)
    os.replace(sys.argv[1] + ".tmp", sys.argv[1])


if __name__ == "__main__":
    main()
