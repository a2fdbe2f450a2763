#!/usr/bin/env python3
"""Per-architecture MLP throughput on the MI355X (diagnostics beside bench.py): fused register-chained kernel vs the
layer-by-layer GEMM path, forward and forward+backward, in TFLOP/s of algorithmic GEMM FLOPs (SURVEY 8d counting)."""
import argparse
import json
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from mega_nerf.models.nerf import NeRF, ShiftedSoftplus   # noqa: E402


def flops_per_sample(m: NeRF) -> int:
    mac = sum(p.numel() for n, p in m.named_parameters() if n.endswith('weight') and not n.startswith('embedding_a'))
    return 2 * mac


def timed(fn, reps):
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--rows', type=int, default=1024 * 192)
    ap.add_argument('--reps', type=int, default=5)
    ap.add_argument('--only', default='', help='comma-separated arch names (default: all)')
    a = ap.parse_args()
    dev = torch.device('cuda')
    torch.manual_seed(0)
    B, S = a.rows, 192
    n_rays = B // S
    xyz = torch.rand(B, 3, device=dev) * 2 - 1
    dirs = torch.nn.functional.normalize(torch.randn(n_rays, 3, device=dev), dim=-1)
    idx = torch.randint(0, 100, (n_rays,), device=dev).float()
    out = torch.empty(B, 4, device=dev)
    d_out = torch.randn(B, 4, device=dev)
    cases = [('w256', dict(layer_dim=256)), ('w512', dict(layer_dim=512)), ('w2048_noapp', dict(layer_dim=2048, appearance_dim=0)),
             ('sh2_w256', dict(layer_dim=256, pos_dir_dim=0, rgb_dim=27))]
    for name, kw in cases:
        if a.only and name not in a.only.split(','):
            continue
        W, app, pd, rgb = kw['layer_dim'], kw.get('appearance_dim', 48), kw.get('pos_dir_dim', 4), kw.get('rgb_dim', 3)
        m = NeRF(12, pd, 8, [4], W, app, False, 100, rgb, 3, ShiftedSoftplus()).to(dev)
        fl = flops_per_sample(m)
        sh = 2 if rgb > 3 else -1
        res = {'arch': name, 'rows': B, 'gflop_per_pass': round(fl * B / 1e9, 1)}
        q8 = pd > 0 and app == 0
        dq = torch.cat([xyz[:, -1:], dirs.repeat_interleave(S, 0)[:, :2]], 1).contiguous() if q8 else None

        def fwd(force_layerwise):
            args = (xyz, 3, dq, 3, None, 0, 1, B, out, None, False, sh) if q8 else \
                (xyz, 3, dirs if (pd > 0 or sh >= 0) else None, 3, idx if app > 0 else None, 1, S, B, out, None, False, sh)
            with torch.no_grad():
                if force_layerwise:
                    m._evaluate_layerwise(*args, None, 0)
                else:
                    m.prefer_wide_layerwise = False          # measure the register-chained kernel itself
                    m.evaluate(*args)
                    m.prefer_wide_layerwise = True

        if m.fused_supported():
            ms = timed(lambda: fwd(False), a.reps)
            res['fused_fwd_ms'], res['fused_fwd_tflops'] = round(ms, 3), round(fl * B / ms / 1e9, 1)
        ms = timed(lambda: fwd(True), a.reps)
        res['layerwise_fwd_ms'], res['layerwise_fwd_tflops'] = round(ms, 3), round(fl * B / ms / 1e9, 1)

        def train():
            grads = {k: torch.zeros_like(p) for k, p in m.named_parameters()}
            if q8:
                tape = m.train_eval(xyz, 3, dq, 3, 1, None, 0, 1, B, out, None, sh, None, 0)
            else:
                tape = m.train_eval(xyz, 3, dirs, 3, S, idx if app > 0 else None, 1, S, B, out, None, sh, None, 0,
                                    dirs if sh >= 0 else None, 3)
            tape.backward(d_out, 4, grads)

        ms = timed(train, max(2, a.reps // 2))
        res['train_path'] = 'fused' if m.fused_train_supported() else 'layerwise'
        res['fwd_bwd_ms'], res['fwd_bwd_tflops'] = round(ms, 3), round(3 * fl * B / ms / 1e9, 1)
        print(json.dumps(res), flush=True)
        del m


if __name__ == '__main__':
    main()
