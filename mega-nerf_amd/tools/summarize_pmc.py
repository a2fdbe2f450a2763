#!/usr/bin/env python3
"""rocprofv3 output of tools/collect_profiles.sh -> the committed evidence files of a round:

    python summarize_pmc.py gpurun_out/r02_prof profiles r02

writes  profiles/<round>_{train,eval}_kernel_stats.csv   (rocprofv3 --kernel-trace --stats: calls, total, average per kernel)
        profiles/<round>_bench_{train,eval}_under_rocprof.json
        profiles/<round>_pmc_summary.json   per roofline kernel: launches, median duration, HBM bytes per launch
                                            = (2 x FETCH_SIZE + WRITE_SIZE) KiB x 1024 (gfx950 reports half of wide coalesced
                                            reads in FETCH_SIZE: MI355X_MICROARCH.md, HBM section), MFMA-busy =
                                            SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs), parked fraction =
                                            SQ_WAIT_ANY / SQ_WAVE_CYCLES, L2 hit rate, and the git revision of the measured tree.
bench.py reads ``hbm_bytes_per_launch`` / ``mfma_busy`` of this file for ``roofline.traffic`` / ``roofline.mfma_busy``.
Each kernel's population is every launch of that symbol in the pass (the same population as its kernel-trace row); the
train-mode forward symbol is told from the eval one by its ``true`` template argument, the eval fine launch by its grid."""
import csv
import glob
import json
import shutil
import statistics
import subprocess
import sys
from collections import defaultdict
from pathlib import Path

KERNELS = {      # summary key -> (pass mode, predicate on (kernel name, grid size))
    'k_wgrad2': ('train', lambda n, g: ('k_wgrad2<' in n and 'true>' not in n) or 'k_wgrad2(' in n),
    'k_wgrad2_reduce': ('train', lambda n, g: 'k_wgrad2_reduce' in n),
    'k_mlp_fwd_multi_train': ('train', lambda n, g: 'k_mlp_fwd_multi' in n and ', true' in n),
    'k_mlp_bwd_multi': ('train', lambda n, g: 'k_mlp_bwd_multi' in n),
    'k_head_grads': ('train', lambda n, g: 'k_head_grads' in n),
    'k_mlp_fwd_multi_eval': ('eval', lambda n, g: 'k_mlp_fwd_multi' in n and ', false' in n),
    'k_mlp_fwd_multi_eval_fine': ('eval', lambda n, g: 'k_mlp_fwd_multi' in n and ', false' in n and g >= 2048 * 256),
    # Building-shaped foreground (layer_dim 512): tiled GEMMs + job-form weight gradients of the layer-by-layer path
    'k_tgemm_forward_w512': ('w512', lambda n, g: 'k_tgemm<false' in n),
    'k_mlp_fwd_pair_train_w512': ('w512', lambda n, g: 'k_mlp_fwd_pair' in n and 'true>' in n),      # round 4: the one-launch forward of W = 512 training
    'k_tgemm_data_gradient_w512': ('w512', lambda n, g: 'k_tgemm<true' in n),
    'k_wgrad2_jobs_w512': ('w512', lambda n, g: 'k_wgrad2<1' in n),
    # opt-in split-precision kernels (16-bit matrix pipe, hi/lo operands)
    'k_mlp_fwd_h2_train': ('split', lambda n, g: 'k_mlp_fwd_h2' in n and 'true>' in n),
    'k_mlp_fwd_h2_eval': ('split', lambda n, g: 'k_mlp_fwd_h2' in n and 'false>' in n),
    'k_mlp_bwd_h2': ('split', lambda n, g: 'k_mlp_bwd_h2' in n),
    'k_wgrad2_h2': ('split', lambda n, g: 'k_wgrad2<0, true>' in n),
}


def load_pass(d):
    """{dispatch id: (kernel name, grid, duration ns, {counter: value})} of one --pmc pass directory."""
    tr = {}
    for f in glob.glob(str(Path(d) / '**' / '*kernel_trace.csv'), recursive=True):
        for r in csv.DictReader(open(f)):
            tr[r['Dispatch_Id']] = [r['Kernel_Name'], int(r.get('Grid_Size', 0) or 0), int(r['End_Timestamp']) - int(r['Start_Timestamp']), {}]
    for f in glob.glob(str(Path(d) / '**' / '*counter_collection.csv'), recursive=True):
        for r in csv.DictReader(open(f)):
            e = tr.setdefault(r['Dispatch_Id'], [r['Kernel_Name'], int(r.get('Grid_Size', 0) or 0), 0, {}])
            e[3][r['Counter_Name']] = float(r['Counter_Value'])
            if not e[1]:
                e[1] = int(r.get('Grid_Size', 0) or 0)
    return tr


def med(xs):
    xs = list(xs)
    return statistics.median(xs) if xs else None


def main():
    src, dst, rnd = Path(sys.argv[1]), Path(sys.argv[2]), sys.argv[3]
    dst.mkdir(exist_ok=True)
    for mode in ('train', 'eval', 'w512', 'split', 'sh2', 'container8', 'container25'):
        for f in glob.glob(str(src / ('trace_' + mode) / '**' / '*kernel_stats.csv'), recursive=True):
            shutil.copy(f, dst / ('%s_%s_kernel_stats.csv' % (rnd, mode)))
        j = src / ('bench_%s_under_rocprof.json' % mode)
        if j.exists() and j.stat().st_size:
            shutil.copy(j, dst / ('%s_bench_%s_under_rocprof.json' % (rnd, mode)))
    passes = {(kind, mode): load_pass(src / ('pmc_%s_%s' % (kind, mode))) for kind in ('fetch', 'write', 'sq') for mode in ('train', 'eval', 'w512', 'split')}
    out = {}
    for key, (mode, pred) in KERNELS.items():
        e = {}
        pick = lambda kind: [v for v in passes[(kind, mode)].values() if pred(v[0], v[1])]      # noqa: E731
        f, w, q = pick('fetch'), pick('write'), pick('sq')
        if not (f or w or q):
            continue
        e['launches_in_pass'] = len(q) or len(f)
        e['median_duration_us_under_pmc'] = round(med(v[2] for v in (q or f)) / 1e3, 1)
        fetch, write = med(v[3].get('FETCH_SIZE') for v in f if 'FETCH_SIZE' in v[3]), med(v[3].get('WRITE_SIZE') for v in w if 'WRITE_SIZE' in v[3])
        if fetch is not None and write is not None:
            e['fetch_size_kib'], e['write_size_kib'] = round(fetch, 1), round(write, 1)
            e['hbm_bytes_per_launch'] = int((2 * fetch + write) * 1024)
        hit, miss = med(v[3].get('TCC_HIT_sum') for v in w if 'TCC_HIT_sum' in v[3]), med(v[3].get('TCC_MISS_sum') for v in w if 'TCC_MISS_sum' in v[3])
        if hit is not None and miss is not None and hit + miss > 0:
            e['l2_hit_rate'] = round(hit / (hit + miss), 4)
        rows = [v[3] for v in q if 'SQ_VALU_MFMA_BUSY_CYCLES' in v[3] and v[3].get('GRBM_GUI_ACTIVE')]
        if rows:
            e['mfma_busy'] = round(med(r['SQ_VALU_MFMA_BUSY_CYCLES'] / (1024 * r['GRBM_GUI_ACTIVE'] / 8) for r in rows), 4)
            e['parked_fraction_of_wave_cycles'] = round(med(r['SQ_WAIT_ANY'] / r['SQ_WAVE_CYCLES'] for r in rows if r.get('SQ_WAVE_CYCLES')), 4)
            e['valu_per_mfma_instruction'] = round(med(r['SQ_INSTS_VALU'] / r['SQ_INSTS_MFMA'] for r in rows if r.get('SQ_INSTS_MFMA')), 3) if any(r.get('SQ_INSTS_MFMA') for r in rows) else None
            e['cycles_per_xcd'] = int(med(r['GRBM_GUI_ACTIVE'] / 8 for r in rows))
            # clock the kernel ran at (the fp32-MFMA peak is priced at 2.4 GHz: frac of peak ~ mfma_busy x clock / 2.4)
            if e.get('median_duration_us_under_pmc'):
                e['approx_clock_ghz'] = round(e['cycles_per_xcd'] / e['median_duration_us_under_pmc'] / 1e3, 3)
        out[key] = e
    try:
        sha = subprocess.run(['git', 'rev-parse', 'HEAD'], capture_output=True, text=True, cwd=str(Path(__file__).resolve().parent)).stdout.strip()
    except Exception:
        sha = None
    import os
    sha = sha or os.environ.get('MNR_GIT_HEAD')        # (the GPU box has no .git: the caller passes the revision it snapshotted)
    out['_meta'] = {'git_head_when_summarised': sha, 'source': str(src), 'command': 'tools/collect_profiles.sh; tools/summarize_pmc.py %s %s %s' % (src, dst, rnd),
                    'formulae': 'hbm_bytes = (2 FETCH_SIZE + WRITE_SIZE) KiB * 1024; mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 * GRBM_GUI_ACTIVE / 8); medians over the launches of the pass'}
    (dst / ('%s_pmc_summary.json' % rnd)).write_text(json.dumps(out, indent=1) + '\n')
    print(json.dumps(out, indent=1))


if __name__ == '__main__':
    main()
