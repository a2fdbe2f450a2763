"""One training / validation image: where its pixels live, its pose and intrinsics, and (optionally) the cell mask that
selects the pixels a submodule trains on.  Counterpart of the reference's mega_nerf/image_metadata.py; the mask file
format is the one create_cluster_masks.py writes (a ZIP archive holding a single torch-saved bool[H, W], :203-210)."""
import zipfile
from pathlib import Path
from typing import Optional

import numpy as np
import torch
import torch.nn.functional as F


class ImageMetadata:
    """Attributes (names used throughout the reference): image_path, c2w (3, 4), W, H, intrinsics (fx, fy, cx, cy),
    image_index (row of the appearance table), is_val."""

    def __init__(self, image_path: Path, c2w: torch.Tensor, W: int, H: int, intrinsics: torch.Tensor, image_index: int,
                 mask_path: Optional[Path], is_val: bool):
        self.image_path = image_path
        self.c2w = c2w
        self.W, self.H = W, H
        self.intrinsics = intrinsics
        self.image_index = image_index
        self.is_val = is_val
        self._mask_path = mask_path

    def load_image(self) -> torch.Tensor:
        """uint8 (H, W, 3), resampled (Lanczos) to the metadata's resolution when the file on disk is larger."""
        from PIL import Image
        with Image.open(self.image_path) as handle:
            picture = handle.convert('RGB')
            if picture.size != (self.W, self.H):
                picture = picture.resize((self.W, self.H), Image.LANCZOS)
            return torch.from_numpy(np.array(picture, dtype=np.uint8))

    def load_mask(self) -> Optional[torch.Tensor]:
        """bool (H, W) or None; masks stored at another resolution are resized with nearest-neighbour sampling."""
        if self._mask_path is None:
            return None
        with zipfile.ZipFile(self._mask_path) as archive:
            with archive.open(self._mask_path.name) as member:
                mask = torch.load(member, map_location='cpu')
        if mask.shape[0] != self.H or mask.shape[1] != self.W:
            mask = F.interpolate(mask.float()[None, None], size=(self.H, self.W))[0, 0].bool()
        return mask
