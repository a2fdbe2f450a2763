#!/usr/bin/env python3
"""Golden fixtures for the merged-container format, produced with the REAL reference:

  container_ref.pt    TorchScript archive written exactly like scripts/merge_submodules.py:70-79 does
                      (torch.jit.script(MegaNeRFContainer(reference NeRF modules ...))) from seeded weights
  container_eval.npz  inputs + outputs of the reference's MegaNeRF wrappers over that archive (model_utils.py:22-29)

and a cross-check that is only possible here: the archive written by THIS repo's exporter is loaded through the
reference's own reader (get_nerf(container_path=...)) and must reproduce the same outputs.

Build container only (needs /root/reference)."""
import sys
import tempfile
from argparse import Namespace
from pathlib import Path

import numpy as np
import torch

HERE = Path(__file__).resolve().parent
ROOT = HERE.parent.parent
sys.dont_write_bytecode = True
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(HERE))

import common  # noqa: E402
from oracle.nerf_oracle import make_hparams  # noqa: E402

N_CELLS, WIDTH, COUNT = 2, 32, 6
f32 = np.float32


def case_hparams():
    return Namespace(**vars(make_hparams(coarse_samples=64, fine_samples=128, layer_dim=WIDTH, bg_layer_dim=WIDTH)))


def seeded_weights():
    hp = case_hparams()
    fcfg, bcfg = common.model_cfg(hp, 3, WIDTH), common.model_cfg(hp, 4, WIDTH)
    return hp, fcfg, bcfg, [common.make_weights(fcfg, COUNT, 7000 + i, sharpen=False) for i in range(N_CELLS)], \
        [common.make_weights(bcfg, COUNT, 7500 + i, sharpen=False) for i in range(N_CELLS)]


def centroid_metadata():
    return {'centroids': torch.tensor([[0., -0.3, 0.1], [0., 0.35, -0.2]]), 'grid_dim': [1, 2],
            'min_position': torch.tensor([-0.4, -0.6, -0.5]), 'max_position': torch.tensor([-0.1, 0.7, 0.4]), 'cluster_2d': False}


def inputs():
    rng = np.random.default_rng(77)
    B = 64
    fg_x = np.concatenate([rng.uniform(-.7, .7, (B, 3)), rng.standard_normal((B, 3)), rng.integers(0, COUNT, (B, 1))], 1).astype(f32)
    bg_x = np.concatenate([rng.uniform(-.7, .7, (B, 3)), rng.uniform(-1, 1, (B, 4)), rng.standard_normal((B, 3)),
                           rng.integers(0, COUNT, (B, 1))], 1).astype(f32)
    return fg_x, bg_x


def main():
    # --- everything below touches the reference; keep its package first on the path only inside this block
    sys.path.insert(0, '/root/reference')
    for k in [k for k in sys.modules if k == 'mega_nerf' or k.startswith('mega_nerf.')]:
        del sys.modules[k]
    from mega_nerf.models import model_utils as MU
    from mega_nerf.models.mega_nerf_container import MegaNeRFContainer
    hp, fcfg, bcfg, fw, bw = seeded_weights()
    meta = centroid_metadata()

    def ref_model(cfg, w):
        m = MU._get_single_nerf_inner(hp, COUNT, cfg.layer_dim, cfg.xyz_dim)
        m.load_state_dict({k: torch.from_numpy(v) for k, v in w.items()})
        return m

    container = MegaNeRFContainer([ref_model(fcfg, w) for w in fw], [ref_model(bcfg, w) for w in bw], meta['centroids'],
                                  torch.IntTensor(meta['grid_dim']), meta['min_position'], meta['max_position'], True, True,
                                  meta['cluster_2d'])
    torch.jit.save(torch.jit.script(container.eval()), str(HERE / 'container_ref.pt'))

    def evaluate(path):
        h = Namespace(**vars(hp))
        h.container_path = str(path)
        fg, bg = MU.get_nerf(h, COUNT).eval(), MU.get_bg_nerf(h, COUNT).eval()
        fg_x, bg_x = inputs()
        with torch.inference_mode():
            return dict(fg_x=fg_x, bg_x=bg_x, fg_out=fg(torch.from_numpy(fg_x)).numpy(), bg_out=bg(torch.from_numpy(bg_x)).numpy(),
                        fg_sigma=fg(torch.from_numpy(fg_x[:, :3]), sigma_only=True).numpy())

    ref_out = evaluate(HERE / 'container_ref.pt')
    np.savez_compressed(HERE / 'container_eval.npz', **ref_out)

    # --- this repo's exporter, read back by the reference
    ref_modules = {k: v for k, v in sys.modules.items() if k == 'mega_nerf' or k.startswith('mega_nerf.')}
    for k in ref_modules:
        del sys.modules[k]
    sys.path.remove('/root/reference')
    sys.path.insert(0, str(ROOT / 'mega-nerf_amd'))
    from mega_nerf.models.export import build_container, save_container
    from mega_nerf.models.model_utils import _get_single_nerf_inner

    def mine(cfg, w):
        m = _get_single_nerf_inner(hp, COUNT, cfg.layer_dim, cfg.xyz_dim)
        m.load_state_dict({k: torch.from_numpy(v) for k, v in w.items()})
        return m

    with tempfile.TemporaryDirectory() as tmp:
        out = Path(tmp) / 'mine.pt'
        save_container(build_container([mine(fcfg, w) for w in fw], [mine(bcfg, w) for w in bw], meta, True, True), out)
        for k in [k for k in sys.modules if k == 'mega_nerf' or k.startswith('mega_nerf.')]:
            del sys.modules[k]
        sys.modules.update(ref_modules)
        got = evaluate(out)
    for k in ('fg_out', 'bg_out', 'fg_sigma'):
        err = float(np.abs(got[k] - ref_out[k]).max())
        print(k, 'max |mine under the reference reader - reference archive| =', err)
        assert err < 2e-6, (k, err)
    print('wrote container_ref.pt', (HERE / 'container_ref.pt').stat().st_size // 1024, 'KiB')


if __name__ == '__main__':
    main()
