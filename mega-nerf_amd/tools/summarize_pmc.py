#!/usr/bin/env python3
"""Per-kernel means of rocprofv3 ``--pmc`` passes (one or more output directories) -> JSON, and the HBM bytes per launch
of the roofline kernels the way MI355X_MICROARCH.md prescribes: (2 x FETCH_SIZE + WRITE_SIZE) KiB (gfx950 reports half of
wide coalesced reads in FETCH_SIZE).  Usage: summarize_pmc.py OUT.json HBM.json DIR [DIR ...]"""
import csv
import json
import sys
from collections import defaultdict
from pathlib import Path


def main():
    out_path, hbm_path, dirs = sys.argv[1], sys.argv[2], sys.argv[3:]
    acc = defaultdict(lambda: defaultdict(list))
    for d in dirs:
        for f in Path(d).rglob('*counter_collection.csv'):
            with open(f) as fh:
                for row in csv.DictReader(fh):
                    name = row.get('Kernel_Name') or row.get('Kernel-Name') or ''
                    key = '{} grid={}'.format(name.split('(')[0][:110], row.get('Grid_Size', '?'))
                    acc[key][row['Counter_Name']].append(float(row['Counter_Value']))
    summary = {k: {c: {'mean': sum(v) / len(v), 'max': max(v), 'n': len(v)} for c, v in cs.items()} for k, cs in sorted(acc.items())}
    json.dump(summary, open(out_path, 'w'), indent=1)

    def traffic(pred):
        best = None
        for k, cs in summary.items():
            if pred(k) and 'FETCH_SIZE' in cs and 'WRITE_SIZE' in cs:
                b = (2 * cs['FETCH_SIZE']['mean'] + cs['WRITE_SIZE']['mean']) * 1024
                if best is None or b > best[0]:
                    best = (b, k)
        return best

    picks = {
        'k_wgrad_fg_bytes_per_launch': traffic(lambda k: 'k_wgrad' in k),
        'k_mlp_fwd_fg_fine_bytes_per_launch': traffic(lambda k: 'k_mlp_fwd' in k and 'Li3ELi12' not in k and '<3, 12' in k and 'false' in k),
        'k_mlp_fwd_train_fg_fine_bytes_per_launch': traffic(lambda k: 'k_mlp_fwd' in k and '<3, 12' in k and 'true' in k),
        'k_mlp_bwd_fg_fine_bytes_per_launch': traffic(lambda k: 'k_mlp_bwd' in k and '<3, 12' in k),
    }
    hbm = {'_note': 'HBM bytes per launch from rocprofv3 PMC passes: (2*FETCH_SIZE + WRITE_SIZE) KiB; FETCH_SIZE doubled per '
                    'MI355X_MICROARCH.md (gfx950 reports half of wide coalesced reads). bench.py --rays 1024; the largest-grid '
                    '(fine pass / dense fg) launch of each kernel.'}
    for k, v in picks.items():
        if v is not None:
            hbm[k] = int(v[0])
            hbm[k + '_source'] = v[1]
    json.dump(hbm, open(hbm_path, 'w'), indent=1)
    print(json.dumps(hbm, indent=1))


if __name__ == '__main__':
    main()
