"""CPU-side checks of the native library: it loads, exports every symbol declared in include/mnr_api.h,
and the packed-weight K ordering is a permutation of every nn.Linear's input columns (no GPU needed)."""
import ctypes as C
import re
from pathlib import Path

import pytest

from mega_nerf import _native as N

ROOT = Path(__file__).resolve().parent.parent


def test_library_exports_every_declared_symbol():
    hdr = (ROOT / 'include' / 'mnr_api.h').read_text()
    declared = set(re.findall(r'\b(mnr_[a-z0-9_]+)\s*\(', hdr))
    assert declared, 'no declarations parsed'
    lib = N.lib()
    for name in sorted(declared):
        assert hasattr(lib, name), name
    assert set(N.EXPORTS) == declared
    assert lib.mnr_version() == 1


def make_desc(xyz_dim=3, W=256, pos_dir_dim=4, app=48, rgb_dim=3, layers=8, skip=(4,), tile=0):
    d = N.ModelDesc()
    d.xyz_dim, d.pos_xyz_dim, d.pos_dir_dim, d.layers = xyz_dim, 12, pos_dir_dim, layers
    d.skip_mask = sum(1 << i for i in skip)
    d.layer_dim, d.appearance_dim, d.appearance_count, d.rgb_dim, d.sigma_activation = W, app, 10, rgb_dim, 1
    d.mfma_tile = tile
    return d


@pytest.mark.parametrize('kw', [dict(), dict(xyz_dim=4), dict(W=512), dict(W=512, xyz_dim=4), dict(W=64),
                                dict(tile=32), dict(tile=32, xyz_dim=4),
                                dict(pos_dir_dim=0, rgb_dim=27), dict(app=0), dict(app=0, pos_dir_dim=0)])
def test_layout_is_a_column_permutation(kw):
    d = make_desc(**kw)
    lib = N.lib()
    P = lib.mnr_layout_parts(C.byref(d))
    assert P == (2 if d.mfma_tile == 32 else 4)   # default tile: 16 samples per wave -> 4 lane-parts
    in_xyz = d.xyz_dim * (1 + 24)
    in_dir = 3 * (1 + 2 * d.pos_dir_dim) if d.pos_dir_dim else 0
    has_final = d.pos_dir_dim > 0 or d.appearance_dim > 0
    widths = [in_xyz if i == 0 else d.layer_dim + (in_xyz if (d.skip_mask >> i) & 1 else 0) for i in range(d.layers)]
    if has_final:
        widths += [d.layer_dim, d.layer_dim + in_dir + d.appearance_dim]
    for layer, width in enumerate(widths):
        steps = lib.mnr_layout_num_steps(C.byref(d), layer)
        assert steps > 0 and steps % 4 == 0
        cols = [lib.mnr_layout_src_col(C.byref(d), layer, s, p) for s in range(steps) for p in range(P)]
        real = [c for c in cols if c >= 0]
        assert sorted(real) == list(range(width)), (layer, width)
        assert all(c >= -1 for c in cols)
    assert lib.mnr_layout_num_steps(C.byref(d), len(widths)) < 0          # out of range -> error
    assert lib.mnr_packed_model_bytes(C.byref(d)) > 0


def test_unsupported_architecture_reports_error():
    d = make_desc(W=2048)
    assert N.lib().mnr_packed_model_bytes(C.byref(d)) == 0
    assert b'unsupported' in N.lib().mnr_last_error()


def test_product_path_refuses_cpu_tensors():
    import torch
    from mega_nerf.models.nerf import NeRF, ShiftedSoftplus
    from mega_nerf import ray_utils
    m = NeRF(12, 4, 8, [4], 256, 48, False, 10, 3, 3, ShiftedSoftplus())
    with pytest.raises(N.NativeError):
        m(torch.zeros(4, 7))
    with pytest.raises(N.NativeError):
        ray_utils.get_ray_directions(4, 4, 1., 1., 2., 2., True, torch.device('cpu'))
    with pytest.raises(Exception, match='Unexpected input shape'):
        m(torch.zeros(4, 5))


def test_ctypes_structures_match_the_c_header(tmp_path):
    """sizeof() of every structure in include/mnr_api.h, as a plain C compiler sees it, equals the ctypes mirror's
    (a field missing on either side would make the library read garbage)."""
    import ctypes
    import shutil
    import subprocess
    from mega_nerf import _native as N
    if shutil.which('gcc') is None:
        pytest.skip('no C compiler')
    names = {'mnr_model_desc': N.ModelDesc, 'mnr_mlp_io': N.MlpIO, 'mnr_composite_io': N.CompositeIO, 'mnr_model_grads': N.ModelGrads,
             'mnr_mlp_grad_io': N.MlpGradIO, 'mnr_composite_grad_io': N.CompositeGradIO, 'mnr_mlp_launch': N.MlpLaunch, 'mnr_mlp_cells_launch': N.MlpCellsLaunch,
             'mnr_mlp_grad_launch': N.MlpGradLaunch, 'mnr_wgrad_region': N.WgradRegion, 'mnr_step_model': N.StepModel,
             'mnr_step_cfg': N.StepCfg, 'mnr_step_layout': N.StepLayout, 'mnr_step_batch': N.StepBatch, 'mnr_step_randoms': N.StepRandoms, 'mnr_render_io': N.RenderIO}
    src = tmp_path / 'sizes.c'
    src.write_text('#include <stdio.h>\n#include "mnr_api.h"\nint main(void) {\n' +
                   ''.join('  printf("%s %%zu\\n", sizeof(%s));\n' % (n, n) for n in list(names) + ['mnr_mlp_cell']) + '  return 0;\n}\n')
    exe = tmp_path / 'sizes'
    subprocess.run(['gcc', '-std=c99', '-I', str(ROOT / 'include'), str(src), '-o', str(exe)], check=True)
    sizes = dict(line.split() for line in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.splitlines())
    for n, cls in names.items():
        assert int(sizes[n]) == ctypes.sizeof(cls), (n, sizes[n], ctypes.sizeof(cls))
    assert int(sizes['mnr_mlp_cell']) == 5 * 8            # MegaNeRF._routed packs cells as rows of five int64


# No kernel of the library may touch scratch (round 5: the one-call paths first, then every MLP kernel incl. the split-precision and the
# 512-wide pair kernels, then k_cluster_ratios).  A spilled VGPR is a scratch_store / scratch_load pair inside the instruction stream (each reload a vmcnt
# dependency) plus HBM write-back traffic; round 4 shipped 51-182 of them in kernels the design notes called spill-free.  The check
# reads the code objects' own metadata (tools/kernel_resources.py), so it runs on the CPU and spills cannot come back silently.
NO_SCRATCH_ALLOWED_AGPR_COPIES = ('k_mlp_fwd<mnr::MlpCfg<3, 12, 4, 48, 512', 'k_mlp_fwd<mnr::MlpCfg<4, 12, 4, 48, 512')


def test_no_kernel_touches_scratch():
    import importlib.util
    spec = importlib.util.spec_from_file_location('kernel_resources', ROOT / 'mega-nerf_amd' / 'tools' / 'kernel_resources.py')
    kr = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(kr)
    if not kr.tools_available():
        pytest.skip('llvm-objdump / llvm-readelf not found')
    ks = kr.kernel_resources(str(N.LIB_PATH))
    assert len(ks) > 100
    # EVERY kernel of the library: no private segment.  (A spill count with a zero-byte private segment = registers parked in AGPRs by the
    # one-wavefront-per-SIMD 512-wide comparison kernel, which owns all 512 registers: v_accvgpr moves, no memory traffic.)
    bad = [(k['name'][:120], k['vgpr_spill_count'], k['private_segment_fixed_size']) for k in ks
           if k['private_segment_fixed_size'] or (k['vgpr_spill_count'] and not any(p in k['name'] for p in NO_SCRATCH_ALLOWED_AGPR_COPIES))]
    assert not bad, bad
