#!/usr/bin/env python3
"""Golden chunk directory written by the REAL reference FilesystemDataset (torch CPU) on a tiny synthetic image set, plus
what its own loader returns for the first chunk; and the cross-check that the reference loads chunk directories written
by this repo's FilesystemDataset writer format (schema / dtypes / compression / metadata).

Build container only.  Work-arounds for importing the reference here: ``np.int`` (removed from numpy) is restored for the
annotation at filesystem_dataset.py:306, and nothing else is touched."""
import shutil
import sys
import tempfile
from pathlib import Path

import numpy as np
import torch
from PIL import Image

HERE = Path(__file__).resolve().parent
sys.dont_write_bytecode = True
sys.path.insert(0, '/root/reference')
np.int = int  # noqa
from mega_nerf.datasets.filesystem_dataset import FilesystemDataset  # noqa: E402  (reference)
from mega_nerf.image_metadata import ImageMetadata  # noqa: E402  (reference)

W, H, N_IMG = 16, 12, 5
f32 = np.float32


def scene(tmp: Path):
    rng = np.random.default_rng(5)
    items, raw = [], []
    for i in range(N_IMG):
        img = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
        path = tmp / 'img_{}.png'.format(i)
        Image.fromarray(img).save(path)
        yaw = 0.2 * i - 0.4
        c2w = np.array([[.6, 0, -.8, -.3 + 0.02 * i], [0.8 * np.sin(yaw), np.cos(yaw), 0.6 * np.sin(yaw), .1 * i - .2],
                        [.8 * np.cos(yaw), -np.sin(yaw), .6 * np.cos(yaw), .05]], f32)
        intr = torch.tensor([W * 0.9, W * 0.9, W / 2.0, H / 2.0])
        items.append(ImageMetadata(path, torch.from_numpy(c2w), W, H, intr, i, None, i == 3))      # image 3 is a val image
        raw.append((img, c2w))
    return items, raw


def main():
    out_dir = HERE / 'chunks_ref'
    if out_dir.exists():
        shutil.rmtree(out_dir)
    with tempfile.TemporaryDirectory() as t:
        tmp = Path(t)
        items, raw = scene(tmp)
        torch.manual_seed(3)
        ds = FilesystemDataset(items, 0.01, 2.0, [-0.5, 0.2], True, torch.device('cpu'), [tmp / 'chunks'], 3, 1, 400)
        ds.load_chunk()
        first = dict(chosen=Path(ds.get_state()).name, rgbs=ds._loaded_rgbs.numpy(), rays=ds._loaded_rays.numpy(),
                     img_indices=ds._loaded_img_indices.numpy())
        ds._chunk_load_executor.shutdown(wait=True)
        shutil.copytree(tmp / 'chunks', out_dir)
        np.savez_compressed(HERE / 'chunks_ref.npz', images=np.stack([r[0] for r in raw]), c2w=np.stack([r[1] for r in raw]), W=W, H=H,
                            intr=np.array([W * 0.9, W * 0.9, W / 2.0, H / 2.0], f32), near=0.01, far=2.0, alt=np.array([-0.5, 0.2], f32),
                            val_index=3, first_chosen=first['chosen'], first_rgbs=first['rgbs'], first_rays=first['rays'],
                            first_img_indices=first['img_indices'])
        print('reference wrote', sorted(p.name for p in out_dir.iterdir()), 'first chunk rows', first['rays'].shape[0])

        # the reference reading a directory in THIS repo's writer format (column names, dtypes, BROTLI, metadata.pt)
        import pyarrow as pa
        import pyarrow.parquet as pq
        mine = tmp / 'mine'
        mine.mkdir()
        rng = np.random.default_rng(0)
        rows = 37
        schema = pa.schema([('img_indices', pa.uint16()), ('rgbs_0', pa.uint8()), ('rgbs_1', pa.uint8()), ('rgbs_2', pa.uint8()),
                            ('pixel_indices', pa.int32())])
        cols = {'img_indices': rng.integers(0, N_IMG, rows).astype(np.uint16), 'rgbs_0': rng.integers(0, 256, rows).astype(np.uint8),
                'rgbs_1': rng.integers(0, 256, rows).astype(np.uint8), 'rgbs_2': rng.integers(0, 256, rows).astype(np.uint8),
                'pixel_indices': rng.integers(0, W * H, rows).astype(np.int32)}
        with pq.ParquetWriter(mine / '000000.parquet', schema, compression='BROTLI') as w:
            w.write_table(pa.table(cols, schema=schema))
        torch.save({'images': N_IMG, 'scale_factor': 1}, mine / 'metadata.pt')
        ds2 = FilesystemDataset(items, 0.01, 2.0, [-0.5, 0.2], True, torch.device('cpu'), [mine], 1, 1, 400)
        ds2.load_chunk()
        assert np.array_equal(ds2._loaded_img_indices.numpy(), cols['img_indices'].astype(np.int32))
        assert np.allclose(ds2._loaded_rgbs.numpy() * 255, np.stack([cols['rgbs_%d' % i] for i in range(3)], 1))
        ds2._chunk_load_executor.shutdown(wait=True)
        print('reference loaded a chunk directory in this repo\'s writer format: OK')


if __name__ == '__main__':
    main()
