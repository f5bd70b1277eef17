#!/usr/bin/env python
"""fp32 error of the weight gradient of a 3x3 convolution accumulated in a Winograd domain, against an fp64 direct
weight gradient, at the ODE block's shape (C = 256, 8x8 image, N samples): prices the F(4x4,3x3)-domain weight
gradient (dU_c = V_c^T Z_c with V = B^T d B the conv's own row operand and Z = A dz A^T; dW = G^T dU G) before any
kernel is written.  Pure numpy on the CPU.  Points (0, 1, -1, 1/2, -2, inf) as in csrc/wino4.h."""
import numpy as np

rng = np.random.default_rng(0)
C, H, W, N = 64, 8, 8, 128
x = np.maximum(rng.standard_normal((N, C, H, W)), 0).astype(np.float32)
dz = (rng.standard_normal((N, C, H, W)) / (C * H * W) ** 0.5).astype(np.float32)


def direct(x, dz, dt):
    x = x.astype(dt); dz = dz.astype(dt)
    xp = np.pad(x, ((0, 0), (0, 0), (1, 1), (1, 1)))
    out = np.zeros((C, C, 3, 3), dt)   # [co, ci, kh, kw]
    for kh in range(3):
        for kw in range(3):
            out[:, :, kh, kw] = np.einsum('nohw,nchw->oc', dz, xp[:, :, kh:kh + H, kw:kw + W])
    return out


def mats(m, dt):
    if m == 2:
        BT = np.array([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dt)
        G = np.array([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], dt)
        AT = np.array([[1, 1, 1, 0], [0, 1, -1, -1]], dt)
    else:
        BT = np.array([[1, -1.5, -2, 1.5, 1, 0], [0, -1, .5, 2.5, 1, 0], [0, 1, -2.5, .5, 1, 0], [0, -2, -1, 2, 1, 0],
                       [0, .5, -1, -.5, 1, 0], [0, 1, -1.5, -2, 1.5, 1]], dt)
        G = np.array([[1, 0, 0], [1 / 3, 1 / 3, 1 / 3], [-1 / 3, 1 / 3, -1 / 3], [-16 / 15, -8 / 15, -4 / 15],
                      [1 / 15, -2 / 15, 4 / 15], [0, 0, 1]], dt)
        AT = np.array([[1, 1, 1, 1, 1, 0], [0, 1, -1, .5, -2, 0], [0, 1, 1, .25, 4, 0], [0, 1, -1, .125, -8, 1]], dt)
    return BT, G, AT


def winograd_wgrad(x, dz, m, dt=np.float32, final=np.float32):
    BT, G, AT = mats(m, dt)
    a = m + 2
    xp = np.pad(x.astype(dt), ((0, 0), (0, 0), (1, 1), (1, 1)))
    dU = np.zeros((C, C, a, a), dt)
    for th in range(H // m):
        for tw in range(W // m):
            d = xp[:, :, th * m:th * m + a, tw * m:tw * m + a]
            V = np.einsum('ij,ncjk,lk->ncil', BT, d, BT).astype(dt)
            t = dz.astype(dt)[:, :, th * m:(th + 1) * m, tw * m:(tw + 1) * m]
            Z = np.einsum('ji,nojk,kl->noil', AT, t, AT).astype(dt)           # A dz A^T, A = AT^T: [n, o, a, a]
            dU += np.einsum('noil,ncil->ocil', Z, V).astype(dt)              # fp32 accumulation over samples (and tiles)
    Gf = G.astype(final)
    return np.einsum('ia,ocij,jb->ocab', Gf, dU.astype(final), Gf)           # G^T dU G


ref = direct(x, dz, np.float64)
scale = np.abs(ref).max()
for name, got in (('direct fp32', direct(x, dz, np.float32)), ('F(2x2,3x3) domain fp32', winograd_wgrad(x, dz, 2)),
                  ('F(4x4,3x3) domain fp32', winograd_wgrad(x, dz, 4)),
                  ('F(4x4,3x3) domain fp32, G^T dU G in fp64', winograd_wgrad(x, dz, 4, final=np.float64))):
    err = np.abs(got.astype(np.float64) - ref)
    print('%-44s max err / max|dW| = %.2e   rms err / rms dW = %.2e' % (name, err.max() / scale, np.sqrt((err ** 2).mean()) / np.sqrt((ref ** 2).mean())))
