"""Diagnostic: the gradient-tape planes of a split-precision step against the fp32 step's on the same inputs (plane by plane)."""
import sys
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/mega-nerf_amd'); sys.path.insert(0, '/root/repo/tests'); sys.path.insert(0, '/root/repo/tests/golden')
import numpy as np, torch
from argparse import Namespace
import common
from test_gpu_step import _randoms_of
from test_gpu_parity import T, native_models
from test_oracle_golden import load
from mega_nerf.training import FusedTrainStep
g = load('render_fgbg_train')
s = common.SCENE
sc, sr = T(s['sphere_center']), T(s['sphere_radius'])
n = g['rays'].shape[0]
planes = {}
for split in (False, True):
    hp, nerf, bg = native_models('render_fgbg_train')
    st = FusedTrainStep([(nerf, bg)], Namespace(**vars(hp)), sc, sr, n, split_precision=split)
    st([(T(g['rays']), T(g['idx'].astype(np.int32)), T(g['target']))], _randoms=[_randoms_of(g)], optimize=False)
    torch.cuda.synchronize()
    lay = st.layout
    wsf = st.workspace.view(torch.float32)
    fpr = nerf.tape_floats_per_row()
    rows = lay.tape_fg_rows
    W = 256
    gt = wsf[lay.gtape_fg_offset // 4:lay.gtape_fg_offset // 4 + rows * fpr]
    tp = wsf[lay.tape_fg_offset // 4:lay.tape_fg_offset // 4 + rows * fpr]
    used = n * 64                      # coarse rows of the foreground
    d = {}
    for l in range(8):
        d['dZ%d' % l] = gt[l * W * rows:(l + 1) * W * rows].view(rows, W)[:used].cpu().numpy().copy()
        d['act%d' % l] = tp[l * W * rows:(l + 1) * W * rows].view(rows, W)[:used].cpu().numpy().copy()
    d['dZfin'] = gt[8 * W * rows:9 * W * rows].view(rows, W)[:used].cpu().numpy().copy()
    d['dZdact'] = gt[9 * W * rows:9 * W * rows + 128 * rows].view(rows, 128)[:used].cpu().numpy().copy()
    doff = lay.gtape_fg_offset + (fpr * rows * 4 + 255) // 256 * 256
    d['dheads'] = wsf[doff // 4:doff // 4 + rows * 4].view(rows, 4)[:used].cpu().numpy().copy()
    planes[split] = d
    if split:
        img = st._packed[0][1].cpu().numpy().view(np.float16).reshape(-1, 4096, 8)      # [chunk][u4][8 halves]
        Wfin = nerf.xyz_encoding_final.weight.detach().cpu().numpy()
        wsig = nerf.sigma.weight.detach().cpu().numpy()[0]
for k in ['dZdact', 'dZfin'] + ['dZ%d' % l for l in range(7, -1, -1)] + ['act%d' % l for l in range(8)]:
    a, b = planes[False][k], planes[True][k]
    sc_ = np.abs(a).max()
    bad = ~np.isfinite(b)
    err = np.abs(np.where(bad, 0, b) - a)
    r, c = np.unravel_index(np.argmax(err), err.shape)
    print('%-7s scale %.2e  max err %.2e (rel %.1e) at row %d col %d: fp32 %.4e split %.4e; non-finite %d; rows with err>1e-3*scale: %d' % (
        k, sc_, err.max(), err.max() / max(sc_, 1e-30), r, c, a[r, c], b[r, c], bad.sum(), (err.max(1) > 1e-3 * sc_).sum()))
a, b = planes[False]['dZ7'], planes[True]['dZ7']
r = int(np.argmax(np.abs(b - a).max(1)))
print('worst row', r, 'fp32', a[r, :8], 'split', b[r, :8], 'ratio', (b[r, :8] / a[r, :8]))

# ---- the transposed image of `final` (chunks 4..7, two K-steps per chunk) against the weights
def hid_src(P, s_, p): return 4 * P * (s_ // 4) + 4 * p + s_ % 4
worst = 0.0; bad_frag = []
for S in range(8):
    ch, kc = 4 + S // 2, S % 2
    for ob in range(16):
        hi = img[ch, ((kc * 16 + ob) * 2) * 64:((kc * 16 + ob) * 2) * 64 + 64].astype(np.float64)        # [lane][8]
        lo = img[ch, ((kc * 16 + ob) * 2 + 1) * 64:((kc * 16 + ob) * 2 + 1) * 64 + 64].astype(np.float64)
        for lane in range(64):
            i, part = lane & 15, lane >> 4
            exp = np.array([Wfin[hid_src(4, 8 * S + j, part), 16 * ob + i] for j in range(8)], np.float64)
            e = np.abs(hi[lane] + lo[lane] - exp).max()
            if e > 1e-6: bad_frag.append((S, ob, lane, e))
            worst = max(worst, e)
print('final^T image: worst |hi+lo-w| %.3e, bad lanes %d' % (worst, len(bad_frag)), bad_frag[:10])
# ---- dZ7 predicted from the split step's own dZfin / dheads / act7
sp = planes[True]
fin, dh, act7, z7 = sp['dZfin'].astype(np.float64), sp['dheads'].astype(np.float64), sp['act7'], sp['dZ7'].astype(np.float64)
print('dheads split vs fp32', np.abs(sp['dheads'] - planes[False]['dheads']).max(), 'scale', np.abs(planes[False]['dheads']).max())
T1 = fin @ Wfin.astype(np.float64)
T2 = dh[:, 3:4] * wsig[None, :].astype(np.float64)
m = act7 > 0
pred = np.where(m, T1 + T2, 0)
sc7 = np.abs(pred).max()
e = np.abs(z7 - pred)
print('dZ7 split vs own prediction: max %.3e scale %.3e' % (np.nanmax(e), sc7))
colbad = (e > 1e-4 * sc7).sum(0)
print('bad entries per column (first 64):', colbad[:64])
print('bad entries per column%16:', [int(colbad[c::16].sum()) for c in range(16)])
rowbad = (e > 1e-4 * sc7).sum(1)
print('rows with a bad entry:', int((rowbad > 0).sum()), 'of', len(rowbad), '; by row%16:', [int((rowbad[c::16] > 0).sum()) for c in range(16)])
print('by (row//16)%8 (wave):', [int((rowbad.reshape(-1, 16).sum(1)[w::8] > 0).sum()) for w in range(8)])
only1 = np.abs(z7 - np.where(m, T1, 0)); only2 = np.abs(z7 - np.where(m, T2, 0)); dbl = np.abs(z7 - np.where(m, T1 + 2 * T2, 0))
print('|z7 - m*T1| max %.3e; |z7 - m*T2| max %.3e; |z7 - m*(T1+2T2)| %.3e' % (only1.max(), only2.max(), dbl.max()))
bm = e > 1e-4 * sc7
if bm.any():
    rr, cc = np.nonzero(bm)
    for r_, c_ in list(zip(rr, cc))[:12]:
        print('  row %d col %d: got %.4e pred %.4e T1 %.4e T2 %.4e ds %.3e' % (r_, c_, z7[r_, c_], pred[r_, c_], T1[r_, c_], T2[r_, c_], dh[r_, 3]))
    # least squares per bad entry set: z7 = a*T1 + b*T2
    A = np.stack([T1[bm], T2[bm]], 1); sol = np.linalg.lstsq(A, z7[bm], rcond=None)[0]
    print('fit over bad entries: z7 = %.4f*T1 + %.4f*T2' % tuple(sol))
print('PROBE_OK')
