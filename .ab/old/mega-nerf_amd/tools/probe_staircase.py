#!/usr/bin/env python3
"""Launch-quantisation staircase of the register-chained forward kernel: time of ONE inference launch of the default
foreground model against its workgroup count (64 rows per workgroup, 2 workgroups -- 8 wavefronts -- resident per CU, 256 CUs = 512
slots per round).  Prints one JSON line per count: ms, TFLOP/s, ms per 512-workgroup round, and the same for the tape-writing (training) form of the kernel."""
import json
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from mega_nerf.models.nerf import NeRF, ShiftedSoftplus   # noqa: E402


def main():
    dev = torch.device('cuda')
    torch.manual_seed(0)
    m = NeRF(12, 4, 8, [4], 256, 48, False, 100, 3, 3, ShiftedSoftplus()).to(dev).eval()
    fl = 2 * sum(p.numel() for n, p in m.named_parameters() if n.endswith('weight') and not n.startswith('embedding_a'))
    S = 64
    for wgs in (64, 128, 256, 320, 384, 512, 576, 768, 1024, 1093, 1280, 1536, 2048, 2186, 4096, 16384):
        B = wgs * 64
        n_rays = B // S
        xyz = torch.rand(B, 3, device=dev) * 2 - 1
        dirs = torch.nn.functional.normalize(torch.randn(n_rays, 3, device=dev), dim=-1)
        idx = torch.randint(0, 100, (n_rays,), device=dev).float()
        out = torch.empty(B, 4, device=dev)

        def fwd():
            with torch.no_grad():
                m.evaluate(xyz, 3, dirs, 3, idx, 1, S, B, out, None, False, -1)

        def fwd_train():          # the tape-writing form of the same kernel (k_mlp_fwd<.., true>: activations + sign-bit planes to HBM)
            m.train_eval(xyz, 3, dirs, 3, S, idx, 1, S, B, out, None, -1, None, 0)

        def median_ms(fn):
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            times = []
            for _ in range(10):
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                fn()
                b.record()
                torch.cuda.synchronize()
                times.append(a.elapsed_time(b))
            return sorted(times)[len(times) // 2]

        ms, ms_t = median_ms(fwd), median_ms(fwd_train)
        print(json.dumps({'workgroups': wgs, 'rounds_of_512': round(wgs / 512, 2), 'ms': round(ms, 4), 'tflops': round(fl * B / ms / 1e9, 1),
                          'ms_per_round': round(ms / max(1.0, wgs / 512), 4), 'tape_writing_ms': round(ms_t, 4),
                          'tape_writing_tflops': round(fl * B / ms_t / 1e9, 1)}), flush=True)


if __name__ == '__main__':
    main()
