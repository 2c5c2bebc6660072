#!/bin/bash
# Multi-rank flow of bench_scenes.py on a one-GPU box (run through gpurun from the repository root): the ranks share GPU 0 and
# exchange host tensors over gloo, so the numbers are not a benchmark; the image checks are.
mkdir -p gpurun_out
run() { timeout -k 10 200 "$@" 2>>gpurun_out/rehearse_scenes.err | grep '^{' >> gpurun_out/rehearse_scenes.jsonl || exit 1; }
: > gpurun_out/rehearse_scenes.jsonl; : > gpurun_out/rehearse_scenes.err
L="python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1"
run python bench_scenes.py --scene final --nx 1920 --ny 1080 --spp 64
run $L --master-port 29512 bench_scenes.py --gpus 2 --shard tiles --rehearse-on-one-gpu --scene final --nx 1920 --ny 1080 --spp 64
run $L --master-port 29513 bench_scenes.py --gpus 2 --shard samples --rehearse-on-one-gpu --scene stress --spheres 65536 --nx 640 --ny 360 --spp 16
run $L --master-port 29514 bench_scenes.py --gpus 2 --shard tiles --rehearse-on-one-gpu --scene smoke --nx 1000 --ny 700 --spp 16
L3="python -m torch.distributed.run --nnodes=1 --nproc-per-node 3 --master-addr 127.0.0.1"
run $L3 --master-port 29515 bench_scenes.py --gpus 3 --shard tiles --rehearse-on-one-gpu --scene cornell --nx 500 --ny 300 --spp 32
python - <<'PY'
import json
for l in open('gpurun_out/rehearse_scenes.jsonl'):
    d = json.loads(l)
    print(d['scene'], d['nx'], d['ny'], d['spp'], d['sharding'], 'ms', d['ms_per_image'], 'image check', d.get('rehearsal_image_check'))
PY
