"""Model factories with the reference's names and behaviour (mega_nerf/models/model_utils.py:12-69)."""
from argparse import Namespace

import torch
from torch import nn
from torch.nn.modules.utils import consume_prefix_in_state_dict_if_present

from mega_nerf.models.cascade import Cascade
from mega_nerf.models.mega_nerf import MegaNeRF
from mega_nerf.models.nerf import NeRF, ShiftedSoftplus


def get_nerf(hparams: Namespace, appearance_count: int) -> nn.Module:
    return _get_nerf_inner(hparams, appearance_count, hparams.layer_dim, 3, 'model_state_dict')


def get_bg_nerf(hparams: Namespace, appearance_count: int) -> nn.Module:
    return _get_nerf_inner(hparams, appearance_count, hparams.bg_layer_dim, 4, 'bg_model_state_dict')


def nerf_from_scripted(sub) -> NeRF:
    """Rebuild a native NeRF from a (TorchScript or eager) reference submodule by reading its state_dict."""
    sd = {k: v for k, v in sub.state_dict().items()}
    n_layers = len({k.split('.')[1] for k in sd if k.startswith('xyz_encodings.')})
    W = sd['xyz_encodings.0.0.weight'].shape[0]
    in_xyz = sd['xyz_encodings.0.0.weight'].shape[1]
    skips = [i for i in range(1, n_layers) if sd['xyz_encodings.%d.0.weight' % i].shape[1] != W]
    app = sd['embedding_a.weight'].shape if 'embedding_a.weight' in sd else (0, 0)
    rgb_dim = sd['rgb.weight'].shape[0]
    affine = 'affine.weight' in sd
    in_dir = 0
    if 'dir_a_encoding.0.weight' in sd:
        in_dir = sd['dir_a_encoding.0.weight'].shape[1] - W - (app[1] if not affine else 0)
    pos_dir = (in_dir // 3 - 1) // 2 if in_dir > 0 else 0
    # in_xyz = xyz_dim * (1 + 2L): 3 * odd is odd, 4 * odd is even -> the parity decides xyz_dim
    xyz_dim = 3 if in_xyz % 2 else 4
    pos_xyz = (in_xyz // xyz_dim - 1) // 2
    act = getattr(sub, 'sigma_activation', None)
    act_name = getattr(act, 'original_name', type(act).__name__)
    softplus = 'ReLU' not in str(act_name)
    m = NeRF(pos_xyz, pos_dir, n_layers, skips, W, app[1], affine, app[0], rgb_dim, xyz_dim,
             ShiftedSoftplus() if softplus else nn.ReLU())
    m.load_state_dict(sd)
    return m


def _get_nerf_inner(hparams: Namespace, appearance_count: int, layer_dim: int, xyz_dim: int,
                    weight_key: str) -> nn.Module:
    if hparams.container_path is not None:
        container = torch.jit.load(hparams.container_path, map_location='cpu')
        prefix = 'sub_module_{}' if xyz_dim == 3 else 'bg_sub_module_{}'
        subs = [nerf_from_scripted(getattr(container, prefix.format(i))) for i in range(len(container.centroids))]
        return MegaNeRF(subs, container.centroids, hparams.boundary_margin, xyz_dim == 4, container.cluster_2d)
    elif hparams.use_cascade:
        nerf = Cascade(_get_single_nerf_inner(hparams, appearance_count, layer_dim, xyz_dim),
                       _get_single_nerf_inner(hparams, appearance_count, layer_dim, xyz_dim))
    elif hparams.train_mega_nerf is not None:
        meta = torch.load(hparams.train_mega_nerf, map_location='cpu', weights_only=False)
        centroids = meta['centroids']
        nerf = MegaNeRF([_get_single_nerf_inner(hparams, appearance_count, layer_dim, xyz_dim)
                         for _ in range(len(centroids))], centroids, 1, xyz_dim == 4, meta['cluster_2d'], True)
    else:
        nerf = _get_single_nerf_inner(hparams, appearance_count, layer_dim, xyz_dim)

    if hparams.ckpt_path is not None:
        state_dict = torch.load(hparams.ckpt_path, map_location='cpu', weights_only=False)[weight_key]
        consume_prefix_in_state_dict_if_present(state_dict, prefix='module.')
        merged = nerf.state_dict()
        merged.update(state_dict)
        nerf.load_state_dict(merged)
    return nerf


def _get_single_nerf_inner(hparams: Namespace, appearance_count: int, layer_dim: int, xyz_dim: int) -> nn.Module:
    rgb_dim = 3 * ((hparams.sh_deg + 1) ** 2) if hparams.sh_deg is not None else 3
    return NeRF(hparams.pos_xyz_dim, hparams.pos_dir_dim, hparams.layers, hparams.skip_layers, layer_dim,
                hparams.appearance_dim, hparams.affine_appearance, appearance_count, rgb_dim, xyz_dim,
                ShiftedSoftplus() if hparams.shifted_softplus else nn.ReLU())
