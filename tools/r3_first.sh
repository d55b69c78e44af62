#!/bin/bash
# round-3 first GPU pass: full GPU suite, bench (with cpu baseline), a 2-rank gloo rehearsal of the N > 1 path on one GPU
out=gpurun_out/r3a
mkdir -p $out
timeout -k 10 1000 python -m pytest tests -m gpu -q --durations=15 > $out/pytest.log 2>&1; echo "pytest rc=$?"; tail -40 $out/pytest.log
timeout -k 10 300 python bench.py --steps 32 --warmup 8 > $out/bench.log 2>&1; echo "bench rc=$?"; tail -c 1500 $out/bench.log
GSR_DIST_BACKEND=gloo timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29711 bench.py --gpus 2 --steps 8 --warmup 2 --no-cpu-baseline > $out/bench_gloo2.log 2>&1; echo "gloo2 rc=$?"; tail -c 600 $out/bench_gloo2.log
