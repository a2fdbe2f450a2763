"""Per-image pose/intrinsics record + lazy image / cluster-mask loading (reference: mega_nerf/image_metadata.py).
Mask files are ZIP archives holding one torch-saved bool[H, W] (written by create_cluster_masks.py:203-210)."""
from pathlib import Path
from typing import Optional
from zipfile import ZipFile

import numpy as np
import torch


class ImageMetadata:
    def __init__(self, image_path: Path, c2w: torch.Tensor, W: int, H: int, intrinsics: torch.Tensor, image_index: int,
                 mask_path: Optional[Path], is_val: bool):
        self.image_path, self.c2w, self.W, self.H = image_path, c2w, W, H
        self.intrinsics, self.image_index, self._mask_path, self.is_val = intrinsics, image_index, mask_path, is_val

    def load_image(self) -> torch.Tensor:
        from PIL import Image
        img = Image.open(self.image_path).convert('RGB')
        if img.size != (self.W, self.H):
            img = img.resize((self.W, self.H), Image.LANCZOS)
        return torch.from_numpy(np.asarray(img).copy())          # uint8 (H, W, 3)

    def load_mask(self) -> Optional[torch.Tensor]:
        if self._mask_path is None:
            return None
        with ZipFile(self._mask_path) as zf, zf.open(self._mask_path.name) as f:
            keep = torch.load(f, map_location='cpu')
        if tuple(keep.shape) != (self.H, self.W):
            keep = torch.nn.functional.interpolate(keep[None, None].float(), size=(self.H, self.W)).bool()[0, 0]
        return keep
