#!/bin/bash
# Weak-scaling sweep of the data-parallel training step on ONE node: bench.py at 1, 2, 4, 8 GPUs (4 images per GPU,
# BASELINE.json configs[2] -> configs[3]), one process per GPU over RCCL, exactly as the driver launches it.
#   scripts/scale.sh [STEPS=20] [WARMUP=5] [GPUS="1 2 4 8"]   -> gpurun_out/scale_N.json (one bench line per N) + a summary
# Scaling efficiency = value(N) / (N * value(1)); the flat-bucket all-reduce is ONE graph node per step.
set -u
STEPS=${1:-20}; WARMUP=${2:-5}; GPUS=${3:-"1 2 4 8"}
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
have=$(python -c "import torch; print(torch.cuda.device_count())")
base=""
for n in $GPUS; do
    if [ "$n" -gt "$have" ]; then echo "skip N=$n: only $have GPU(s) visible"; continue; fi
    port=$((29500 + RANDOM % 2000))
    extra="--no-cpu-baseline"; [ "$n" -gt 1 ] || extra=""
    if [ "$n" -eq 1 ]; then
        python bench.py --gpus 1 --steps $STEPS --warmup $WARMUP $extra > gpurun_out/scale_$n.json 2> gpurun_out/scale_$n.err
    else
        python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $port \
            bench.py --gpus $n --steps $STEPS --warmup $WARMUP > gpurun_out/scale_$n.json 2> gpurun_out/scale_$n.err
    fi
    rc=$?
    [ $rc -eq 0 ] || { echo "N=$n failed (rc=$rc)"; tail -20 gpurun_out/scale_$n.err; continue; }
    python - "$n" "${base:-0}" <<'PY'
import json, sys
n, base = int(sys.argv[1]), float(sys.argv[2])
line = [l for l in open(f"gpurun_out/scale_{n}.json") if l.startswith("{")][-1]
r = json.loads(line)
eff = f", efficiency {r['value'] / (n * base):.3f}" if base else ""
print(f"N={n}: {r['value']:.1f} images/s, {r['ms_per_step']:.3f} ms/step, collective_in_graph={r.get('collective_in_graph')}{eff}")
PY
    if [ "$n" -eq 1 ]; then base=$(python -c "import json;print(json.loads([l for l in open('gpurun_out/scale_1.json') if l.startswith('{')][-1])['value'])"); fi
done
