#!/bin/bash
# bench.py's N > 1 code path on a 1-GPU box: N ranks on one GPU over gloo (MNR_BENCH_SHARE_GPU=1), weak (one submodule per rank) and
# strong (--submodules 8: the fixed 8-cell set dealt to the ranks) -- a code-path check, not a scaling measurement
export MNR_BENCH_SHARE_GPU=1 HSA_ENABLE_IPC_MODE_LEGACY=0
p=29610
for n in 2 4 8; do
  for extra in "" "--submodules 8"; do
    p=$((p + 1))
    timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $p bench.py --gpus $n --steps 4 --warmup 2 $extra --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('n=%d %-16s value %10.0f rays/s  ms_per_step %8.3f  scaling %s  parallelism %s' % (d['n_gpus'], '$extra', d['value'], d['ms_per_step'], d['scaling'], d['config'].get('parallelism')))"
  done
done
