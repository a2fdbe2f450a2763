"""Device-resident training set (reference: mega_nerf/datasets/memory_dataset.py + dataset_utils.py).

MI355X-first: rays are generated on the GPU by the native ray kernels and the whole (rays, rgb, image index) set
stays in HBM (288 GB: 8 floats + 3 bytes + 4 bytes per pixel, i.e. ~40 GB for 10^9 pixels); batches are drawn with a
device-side permutation, so a training step never waits for a host-side DataLoader collation."""
from typing import Dict, List, Optional, Tuple

import torch
from torch.utils.data import Dataset

from mega_nerf.image_metadata import ImageMetadata
from mega_nerf.misc_utils import main_print, main_tqdm
from mega_nerf.ray_utils import get_ray_directions, get_rays


_UNIT: Dict[str, torch.Tensor] = {}


def unit_rgb(rgb_u8: torch.Tensor) -> torch.Tensor:
    """uint8 colours -> fp32 in [0, 1] with the values the reference's CPU ``x / 255.`` produces (device division by a
    scalar is a multiplication by the rounded reciprocal and differs by an ulp for half of the 256 levels): table lookup."""
    key = str(rgb_u8.device)
    if key not in _UNIT:
        _UNIT[key] = (torch.arange(256, dtype=torch.float32) / 255.).to(rgb_u8.device)
    return _UNIT[key][rgb_u8.long()]


def unit_table(device) -> torch.Tensor:
    """The 256-entry table behind :func:`unit_rgb` on ``device`` (``mnr_step_batch::u8_table``)."""
    probe = torch.zeros(1, dtype=torch.uint8, device=device)
    unit_rgb(probe)
    return _UNIT[str(probe.device)]


def get_rgb_index_mask(metadata: ImageMetadata) -> Optional[Tuple[torch.Tensor, torch.Tensor, Optional[torch.Tensor]]]:
    """Pixels of one image that take part in training (reference dataset_utils.py:8-39): validation images keep
    only their left half; cluster masks select the pixels of this submodule."""
    rgbs = metadata.load_image().view(-1, 3)
    keep = metadata.load_mask()
    H, W = metadata.H, metadata.W
    if metadata.is_val:
        if keep is None:
            keep = torch.ones(H, W, dtype=torch.bool)
        else:
            lost = int(keep[:, W // 2:].sum())                  # right-half pixels we are about to drop
            cand = torch.arange(H * W).view(H, W)[:, :W // 2][~keep[:, :W // 2]].reshape(-1)
            add = cand[torch.randperm(cand.shape[0])[:lost]]    # compensate with unmasked left-half pixels
            keep.view(-1)[add] = True
        keep[:, W // 2:] = False
    if keep is not None:
        if not bool(keep.any()):
            return None
        keep = keep.reshape(-1)
        rgbs = rgbs[keep]
    return rgbs, torch.full((rgbs.shape[0],), metadata.image_index, dtype=torch.int32), keep


class MemoryDataset(Dataset):
    def __init__(self, metadata_items: List[ImageMetadata], near: float, far: float, ray_altitude_range: List[float],
                 center_pixels: bool, device: torch.device):
        super().__init__()
        rgbs, rays, indices = [], [], []
        main_print('Loading data')
        for item in main_tqdm(metadata_items):
            data = get_rgb_index_mask(item)
            if data is None:
                continue
            image_rgbs, image_indices, keep = data
            dirs = get_ray_directions(item.W, item.H, item.intrinsics[0], item.intrinsics[1], item.intrinsics[2],
                                      item.intrinsics[3], center_pixels, device)
            image_rays = get_rays(dirs, item.c2w.to(device), near, far, ray_altitude_range).view(-1, 8)
            if keep is not None:
                image_rays = image_rays[keep.to(device)]
            rgbs.append(image_rgbs.to(device))
            rays.append(image_rays)
            indices.append(image_indices.to(device))
        main_print('Finished loading data')
        self._rgbs = torch.cat(rgbs)                     # uint8 (P, 3) on the device
        self._rays = torch.cat(rays)                     # fp32 (P, 8)
        self._img_indices = torch.cat(indices)           # int32 (P,)

    def __len__(self) -> int:
        return self._rgbs.shape[0]

    def __getitem__(self, idx) -> Dict[str, torch.Tensor]:
        return {'rgbs': unit_rgb(self._rgbs[idx]), 'rays': self._rays[idx], 'img_indices': self._img_indices[idx]}

    def batches(self, batch_size: int, generator: Optional[torch.Generator] = None):
        """One shuffled epoch of device-resident batches."""
        for sel in self.index_batches(batch_size, generator):
            yield {'rgbs': unit_rgb(self._rgbs[sel]), 'rays': self._rays[sel], 'img_indices': self._img_indices[sel]}

    def index_batches(self, batch_size: int, generator: Optional[torch.Generator] = None):
        """The same epoch as :meth:`batches`, as row selections (int64, on the device): a consumer that gathers by itself -- the one-call
        training step (``training.GatheredBatch``) -- pairs them with :meth:`gather_source`."""
        perm = torch.randperm(len(self), generator=generator).to(self._rays.device)
        for i in range(0, len(self), batch_size):
            yield perm[i:i + batch_size]

    def gather_source(self):
        """(rays [P, 8] fp32, img_indices [P] int32, rgbs [P, 3] uint8, u8 -> fp32 table): the resident arrays :meth:`index_batches` selects from."""
        return self._rays, self._img_indices, self._rgbs, unit_table(self._rays.device)
