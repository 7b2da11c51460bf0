#!/bin/bash
# Exercise the multi-rank bench path (broadcast, global counts, 3-segment graphs, flat-arena all-reduce) with two
# processes sharing ONE GPU over gloo, with bench.py's DEFAULT flags (what the driver passes at N > 1; an earlier version of this
# script skipped the roofline / cpu-baseline legs and so never exercised the rank-0-only code that hung the job)
# processes sharing ONE GPU over gloo (RCCL refuses two ranks on one device; the exchange layer is backend agnostic).
set -e
export ADVMIL_DIST_BACKEND=gloo HSA_ENABLE_IPC_MODE_LEGACY=0
python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 \
    bench.py --gpus 2 --steps 4 --warmup 1 --patches 2048 --pool 8 --bags 4
# ... and with 4 and 8 ranks (the 16-bag step split 4 / 2 bags per rank in the strong leg): the row maps of every fused kernel above W = 2
python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29615 \
    bench.py --gpus 4 --steps 3 --warmup 1 --patches 1024 --pool 8 --bags 4 --no-cpu-baseline
python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29617 \
    bench.py --gpus 8 --steps 3 --warmup 1 --patches 1024 --pool 4 --bags 2 --no-cpu-baseline
# the stall watchdog: rank 1 stops before the first collective of the timed legs -> every rank must leave with a non-zero code within seconds
set +e
t0=$(date +%s)
ADVMIL_BENCH_TEST_STALL=1 ADVMIL_BENCH_STALL_S=5 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29613 \
    bench.py --gpus 2 --steps 4 --warmup 1 --patches 2048 --pool 8 --bags 4 --no-extras --no-roofline --no-cpu-baseline > /dev/null 2> /tmp/stall.err
rc=$?
echo "watchdog self-test: exit code $rc after $(( $(date +%s) - t0 )) s: $(grep -m1 'watchdog' /tmp/stall.err)"
[ $rc -ne 0 ]
