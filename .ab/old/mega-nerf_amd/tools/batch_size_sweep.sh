#!/bin/bash
# rays/s of the training step against the batch size: the MLP launches are paid in quanta of 256 workgroups, so
# throughput is a sawtooth in the number of rays -- 1024 (the reference's default batch_size) sits just past a tooth
for r in 832 896 944 960 1024 1088 1184 1280 1536 1888 2048 4096; do
  python bench.py --gpus 1 --steps 10 --warmup 3 --rays $r --no-cpu-baseline --no-extras --no-config-sweep 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('rays %5d  bg %4d  step %7.3f ms  %8.0f rays/s' % ($r, d['config']['bg_rays_in_batch'], d['ms_per_step'], d['value']))"
done
