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


def _branch_of(xyz_dim: int) -> str:
    return 'fg' if xyz_dim == 3 else 'bg'


def _from_container(hparams: Namespace, xyz_dim: int) -> MegaNeRF:
    """Routed model over the cells of a merged TorchScript container (native modules rebuilt from the state_dicts)."""
    archive = torch.jit.load(hparams.container_path, map_location='cpu')
    stem = {'fg': 'sub_module_', 'bg': 'bg_sub_module_'}[_branch_of(xyz_dim)]
    cells = [nerf_from_scripted(getattr(archive, stem + str(i))) for i in range(archive.centroids.shape[0])]
    return MegaNeRF(cells, archive.centroids, hparams.boundary_margin, _branch_of(xyz_dim) == 'bg', archive.cluster_2d)


def _load_weights(model: nn.Module, ckpt_path: str, weight_key: str) -> None:
    state = torch.load(ckpt_path, map_location='cpu', weights_only=False)[weight_key]
    consume_prefix_in_state_dict_if_present(state, prefix='module.')       # checkpoints written under DDP
    full = model.state_dict()
    full.update(state)
    model.load_state_dict(full)


def _get_nerf_inner(hparams: Namespace, appearance_count: int, layer_dim: int, xyz_dim: int, weight_key: str) -> nn.Module:
    """Model selection of the reference (model_utils.py:20-54): container > cascade > jointly trained cells > single NeRF,
    then optional weights from ``--ckpt_path`` (never for a container: it carries its own)."""
    if hparams.container_path is not None:
        return _from_container(hparams, xyz_dim)

    def single() -> NeRF:
        return _get_single_nerf_inner(hparams, appearance_count, layer_dim, xyz_dim)

    if hparams.use_cascade:
        model: nn.Module = Cascade(single(), single())
    elif hparams.train_mega_nerf is not None:
        clustering = torch.load(hparams.train_mega_nerf, map_location='cpu', weights_only=False)
        model = MegaNeRF([single() for _ in clustering['centroids']], clustering['centroids'], 1, _branch_of(xyz_dim) == 'bg',
                         clustering['cluster_2d'], True)
    else:
        model = single()
    if hparams.ckpt_path is not None:
        _load_weights(model, hparams.ckpt_path, weight_key)
    return model


def _get_single_nerf_inner(hparams: Namespace, appearance_count: int, layer_dim: int, xyz_dim: int) -> NeRF:
    colour_outputs = 3 if hparams.sh_deg is None else 3 * (hparams.sh_deg + 1) ** 2
    density_activation = ShiftedSoftplus() if hparams.shifted_softplus else nn.ReLU()
    return NeRF(pos_xyz_dim=hparams.pos_xyz_dim, pos_dir_dim=hparams.pos_dir_dim, layers=hparams.layers,
                skip_layers=hparams.skip_layers, layer_dim=layer_dim, appearance_dim=hparams.appearance_dim,
                affine_appearance=hparams.affine_appearance, appearance_count=appearance_count, rgb_dim=colour_outputs,
                xyz_dim=xyz_dim, sigma_activation=density_activation)
