#!/bin/bash
# host enqueue time against step time for every bench configuration (a configuration is host-bound when the two are equal)
for a in "--mode train" "--mode eval" "--submodules 8" "--container 8 --mode eval" "--layer-dim 512" "--layer-dim 512 --mode eval" \
         "--layer-dim 512 --container 25 --mode eval" "--sh-deg 2" "--sh-deg 2 --mode eval" "--sh-deg 3" "--samples 256,512" "--samples 256,512 --mode eval" \
         "--mode eval --rays 65536"; do
  python bench.py --gpus 1 --steps 10 --warmup 3 $a --no-cpu-baseline --no-extras --no-config-sweep 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('%-46s step %8.3f ms   host enqueue %7.3f ms' % ('$a', d['ms_per_step'], d['host']['host_enqueue_ms_per_step']))"
done
