P='import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["roofline"]["frac"])'
timeout 600 python -m pytest tests/test_gpu_step.py -x -q -m gpu -k "w512 or wide" < /dev/null 2>&1 | tail -3
for rep in 1 2; do
for v in 128 256 512; do
  export MNR_WIDE_HEAD_BLOCKS=$v
  echo "head blocks=$v:"; timeout 200 python bench.py --mode train --layer-dim 512 --no-config-sweep --no-diag --no-cpu-baseline --no-extras < /dev/null 2>/dev/null | tail -1 | python -c "$P"
done; done
