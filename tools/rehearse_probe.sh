run() { tag=$1; shift; env "$@" YV3_DIST_BACKEND=gloo timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 10 --warmup 3 $EXTRA 2>/dev/null | grep '^{' | python -c 'import sys,json; r=json.loads(sys.stdin.read()); print("'$tag'", r["value"], r["ms_per_step"], r.get("lanes"), r["stages_ms"]["convs"], r.get("lanes_calibration_ms"))'; }
EXTRA="--lanes 1" run lanes1 A=1
EXTRA="" run default A=1
EXTRA="" run hwq2 GPU_MAX_HW_QUEUES=2
EXTRA="--lanes 2" run forced2 A=1
