#!/bin/bash
# Soak of the single-rank forced-RCCL launch that tests/test_dist_gpu.py makes (round 3 saw a worker SIGABRT about once in
# ten launches; this runs it N times and keeps the return code + stderr tail of every failure).
#   scripts/dist_soak.sh [N=8] -> gpurun_out/dist_soak.txt
N=${1:-8}
OUT=gpurun_out/dist_soak.txt
mkdir -p gpurun_out
: > $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0 MEDT_FORCE_DIST=1
fail=0
for i in $(seq 1 $N); do
    port=$((29500 + RANDOM % 2000))
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port $port \
        bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-roofline > /tmp/soak_$i.out 2> /tmp/soak_$i.err
    rc=$?
    loss=$(grep -o '"final_loss": [0-9.e-]*' /tmp/soak_$i.out | tail -1)
    ms=$(grep -o '"ms_per_step": [0-9.]*' /tmp/soak_$i.out | tail -1)
    echo "run $i rc=$rc $ms $loss" >> $OUT
    if [ $rc -ne 0 ]; then fail=$((fail + 1)); echo "---- stderr tail of run $i" >> $OUT; tail -40 /tmp/soak_$i.err >> $OUT; fi
done
echo "failures: $fail of $N" >> $OUT
cat $OUT
