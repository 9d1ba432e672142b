#!/bin/bash
# the N = 2 launch path on a ONE-GPU box (two ranks share the device, gloo transport): the driver's command line with --backend gloo
O=$PWD/gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 5 --warmup 2 --backend gloo --no-cpu-baseline --extra 0 > $O/bench_gloo2_r04i.log 2>&1
tail -1 $O/bench_gloo2_r04i.log | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print({k:d[k] for k in ('value','n_gpus','ms_per_step','scaling')}); print(json.dumps(d.get('dp'))[:900])"
