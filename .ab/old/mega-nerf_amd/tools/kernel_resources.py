#!/usr/bin/env python3
"""Register / scratch budget of every gfx950 kernel in libmeganerf_hip.so, read from the code objects' metadata notes.

    python mega-nerf_amd/tools/kernel_resources.py [--lib PATH] [--scratch-only] [--json]

The shared library carries one offload bundle per translation unit; `llvm-objdump --offloading` unpacks them (into a temporary copy's
directory), `llvm-readelf --notes` prints each code object's `amdhsa.kernels` list.  Used by `tests/test_native_cpu.py` to keep the
kernels of the `mnr_train_step` / `mnr_render_fwd` path free of scratch (a spilled VGPR is a `scratch_store` / `scratch_load` pair in
the instruction stream and HBM write-back traffic; VERDICT round 4 found 51-182 of them in kernels DESIGN.md called spill-free).
"""
import argparse
import json
import os
import re
import shutil
import subprocess
import tempfile

import yaml

LLVM_BIN = os.environ.get('MNR_LLVM_BIN', '/opt/rocm/lib/llvm/bin')
DEFAULT_LIB = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'lib', 'libmeganerf_hip.so')

_FIELDS = ('.vgpr_count', '.agpr_count', '.sgpr_count', '.vgpr_spill_count', '.sgpr_spill_count', '.private_segment_fixed_size',
           '.group_segment_fixed_size', '.max_flat_workgroup_size')


def tools_available():
    return all(os.path.exists(os.path.join(LLVM_BIN, t)) for t in ('llvm-objdump', 'llvm-readelf'))


def _demangle(names):
    filt = shutil.which('c++filt') or shutil.which('llvm-cxxfilt')
    if not names or filt is None:
        return {}
    out = subprocess.run([filt], input='\n'.join(names), capture_output=True, text=True, check=True)
    return dict(zip(names, out.stdout.splitlines()))


def kernel_resources(lib=DEFAULT_LIB):
    """-> list of dicts {name (demangled), symbol, vgpr_count, vgpr_spill_count, private_segment_fixed_size, ...}."""
    kernels = []
    with tempfile.TemporaryDirectory(prefix='mnr_co_') as tmp:
        copy = os.path.join(tmp, os.path.basename(lib))
        shutil.copy(lib, copy)
        subprocess.run([os.path.join(LLVM_BIN, 'llvm-objdump'), '--offloading', copy], capture_output=True, check=True)
        for f in sorted(os.listdir(tmp)):
            if 'amdgcn' not in f:
                continue
            notes = subprocess.run([os.path.join(LLVM_BIN, 'llvm-readelf'), '--notes', os.path.join(tmp, f)],
                                   capture_output=True, text=True, check=True).stdout
            # the note body is a YAML document between "---" and "..."
            for doc in re.findall(r'^\s*---\n(.*?)^\.\.\.', notes, flags=re.S | re.M):
                meta = yaml.safe_load(doc) or {}
                for k in meta.get('amdhsa.kernels', []):
                    entry = {'symbol': k['.name']}
                    entry.update({f[1:]: int(k.get(f, 0)) for f in _FIELDS})
                    kernels.append(entry)
    kernels = [k for k in kernels if 'symbol' in k and 'vgpr_count' in k]
    names = _demangle([k['symbol'] for k in kernels])
    for k in kernels:
        k['name'] = names.get(k['symbol'], k['symbol'])
    return kernels


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--lib', default=DEFAULT_LIB)
    ap.add_argument('--scratch-only', action='store_true')
    ap.add_argument('--json', action='store_true')
    a = ap.parse_args()
    ks = kernel_resources(a.lib)
    if a.scratch_only:
        ks = [k for k in ks if k.get('private_segment_fixed_size', 0) or k.get('vgpr_spill_count', 0)]
    if a.json:
        print(json.dumps(ks, indent=1))
        return
    print(f'{len(ks)} kernels')
    for k in sorted(ks, key=lambda k: -k.get('private_segment_fixed_size', 0)):
        print(f"{k.get('vgpr_count', 0):4d} vgpr {k.get('agpr_count', 0):4d} agpr  spill {k.get('vgpr_spill_count', 0):5d}  "
              f"scratch {k.get('private_segment_fixed_size', 0):6d} B  lds {k.get('group_segment_fixed_size', 0):7d}  {k['name'][:150]}")


if __name__ == '__main__':
    main()
