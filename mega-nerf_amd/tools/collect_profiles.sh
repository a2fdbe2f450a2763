#!/bin/bash
# Collect the rocprofv3 evidence behind bench.py's roofline numbers (run on the MI355X box from the repo root):
#   kernel trace + stats of the default training bench and of the eval bench, then separate --pmc passes
#   (HBM counters in their own passes as MI355X_MICROARCH.md prescribes: FETCH_SIZE takes 3 TCC slots, WRITE_SIZE 2).
# Output: $OUT (default gpurun_out/r06_prof); summarise with  python mega-nerf_amd/tools/summarize_pmc.py $OUT profiles r06
set -u
OUT=${1:-$PWD/gpurun_out/r06_prof}
B=$PWD/bench.py
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
for mode in train eval; do
  timeout 300 rocprofv3 --kernel-trace --stats -d "$OUT/trace_$mode" -o t --output-format csv -- \
      python "$B" --mode $mode --steps 10 --warmup 3 --no-cpu-baseline --no-extras --no-diag > "$OUT/bench_${mode}_under_rocprof.json" 2> "$OUT/trace_$mode.err" < /dev/null
  timeout 300 rocprofv3 --pmc FETCH_SIZE GRBM_GUI_ACTIVE --kernel-trace -d "$OUT/pmc_fetch_$mode" -o p --output-format csv -- \
      python "$B" --mode $mode --steps 2 --warmup 2 --no-cpu-baseline --no-extras --no-diag > /dev/null 2> "$OUT/pmc_fetch_$mode.err" < /dev/null
  timeout 300 rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE --kernel-trace -d "$OUT/pmc_write_$mode" -o p --output-format csv -- \
      python "$B" --mode $mode --steps 2 --warmup 2 --no-cpu-baseline --no-extras --no-diag > /dev/null 2> "$OUT/pmc_write_$mode.err" < /dev/null
  timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA GRBM_GUI_ACTIVE \
      --kernel-trace -d "$OUT/pmc_sq_$mode" -o p --output-format csv -- \
      python "$B" --mode $mode --steps 2 --warmup 2 --no-cpu-baseline --no-extras --no-diag > /dev/null 2> "$OUT/pmc_sq_$mode.err" < /dev/null
done
# Building-shaped foreground (layer_dim 512): kernel trace + the SQ / HBM passes of the training step
W5="--layer-dim 512 --no-cpu-baseline --no-extras --no-diag"
timeout 300 rocprofv3 --kernel-trace --stats -d "$OUT/trace_w512" -o t --output-format csv -- \
    python "$B" $W5 --steps 10 --warmup 3 > "$OUT/bench_w512_under_rocprof.json" 2> "$OUT/trace_w512.err" < /dev/null
timeout 300 rocprofv3 --pmc FETCH_SIZE GRBM_GUI_ACTIVE --kernel-trace -d "$OUT/pmc_fetch_w512" -o p --output-format csv -- \
    python "$B" $W5 --steps 2 --warmup 2 > /dev/null 2> "$OUT/pmc_fetch_w512.err" < /dev/null
timeout 300 rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE --kernel-trace -d "$OUT/pmc_write_w512" -o p --output-format csv -- \
    python "$B" $W5 --steps 2 --warmup 2 > /dev/null 2> "$OUT/pmc_write_w512.err" < /dev/null
timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA GRBM_GUI_ACTIVE \
    --kernel-trace -d "$OUT/pmc_sq_w512" -o p --output-format csv -- \
    python "$B" $W5 --steps 2 --warmup 2 > /dev/null 2> "$OUT/pmc_sq_w512.err" < /dev/null
# opt-in split-precision kernels (k_mlp_fwd_h2 / k_mlp_bwd_h2): they run in the side measurements of the default train bench
SP="--mode train --steps 8 --warmup 2 --no-cpu-baseline --only-split-extras --no-diag"
timeout 300 rocprofv3 --kernel-trace --stats -d "$OUT/trace_split" -o t --output-format csv -- \
    python "$B" $SP > "$OUT/bench_split_under_rocprof.json" 2> "$OUT/trace_split.err" < /dev/null
timeout 300 rocprofv3 --pmc FETCH_SIZE GRBM_GUI_ACTIVE --kernel-trace -d "$OUT/pmc_fetch_split" -o p --output-format csv -- \
    python "$B" $SP > /dev/null 2> "$OUT/pmc_fetch_split.err" < /dev/null
timeout 300 rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE --kernel-trace -d "$OUT/pmc_write_split" -o p --output-format csv -- \
    python "$B" $SP > /dev/null 2> "$OUT/pmc_write_split.err" < /dev/null
timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA GRBM_GUI_ACTIVE \
    --kernel-trace -d "$OUT/pmc_sq_split" -o p --output-format csv -- \
    python "$B" $SP > /dev/null 2> "$OUT/pmc_sq_split.err" < /dev/null
# spherical-harmonics shape (configs/mega-nerf-sh-3: sh_deg 2): kernel trace of the one-call step incl. k_sh_head_bwd
timeout 300 rocprofv3 --kernel-trace --stats -d "$OUT/trace_sh2" -o t --output-format csv -- \
    python "$B" --sh-deg 2 --steps 10 --warmup 3 --no-cpu-baseline --no-extras --no-diag > "$OUT/bench_sh2_under_rocprof.json" 2> "$OUT/trace_sh2.err" < /dev/null
# routed 8-cell container through the one-call render (route -> all cells in one launch -> blend)
timeout 300 rocprofv3 --kernel-trace --stats -d "$OUT/trace_container8" -o t --output-format csv -- \
    python "$B" --mode eval --container 8 --steps 20 --warmup 3 --no-cpu-baseline --no-extras --no-diag > "$OUT/bench_container8_under_rocprof.json" 2> "$OUT/trace_container8.err" < /dev/null
# routed 25-cell container of 512-wide cells (k_mlp_fwd_pair in gather mode, XCD-contiguous workgroup order)
timeout 300 rocprofv3 --kernel-trace --stats -d "$OUT/trace_container25" -o t --output-format csv -- \
    python "$B" --mode eval --layer-dim 512 --container 25 --steps 10 --warmup 3 --no-cpu-baseline --no-extras --no-diag > "$OUT/bench_container25_under_rocprof.json" 2> "$OUT/trace_container25.err" < /dev/null
# keep the merge small: only the csv / json / err files travel back
find "$OUT" -type f ! -name "*.csv" ! -name "*.json" ! -name "*.err" -delete
ls "$OUT"
