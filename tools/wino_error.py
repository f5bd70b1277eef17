#!/usr/bin/env python
"""fp32 error of Winograd F(2x2,3x3) and F(4x4,3x3) against an fp64 direct convolution, at the ODE block's conv shape
(C = 256 input channels, 3x3, pad 1, 8x8 image; activations ~ relu(N(0,1)), weights ~ U(-1/48, 1/48) like the default
init).  Pure numpy on the CPU -- prices candidate (iii) of the round-1 review before any kernel is written:
dopri5's embedded error estimate at tol 1e-5 is ~1e-5 |y|, so the convolution noise must stay well below that.
F(4x4,3x3) is evaluated for the textbook interpolation points (0, +-1, +-2, inf) and for (0, 1, -1, 1/2, -2, inf),
the set csrc/wino4.h uses: 8.4e-6 against 3.2e-6 of max|y|."""
import numpy as np

rng = np.random.default_rng(0)
C, H, W, N = 256, 8, 8, 4
x = np.maximum(rng.standard_normal((N, C, H, W)), 0).astype(np.float32)
w = (rng.uniform(-1, 1, (C, C, 3, 3)) / 48).astype(np.float32)


def direct(x, w, dt):
    x = x.astype(dt); w = w.astype(dt)
    xp = np.pad(x, ((0, 0), (0, 0), (1, 1), (1, 1)))
    out = np.zeros((x.shape[0], w.shape[0], H, W), dt)
    for kh in range(3):
        for kw in range(3):
            out += np.einsum('nchw,oc->nohw', xp[:, :, kh:kh + H, kw:kw + W], w[:, :, kh, kw])
    return out


def winograd(x, w, m, dt=np.float32):
    if m == 2:
        BT = np.array([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dt)
        G = np.array([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], dt)
        AT = np.array([[1, 1, 1, 0], [0, 1, -1, -1]], dt)
    elif m == 4:    # textbook points 0, +-1, +-2, inf (Lavin & Gray)
        BT = np.array([[4, 0, -5, 0, 1, 0], [0, -4, -4, 1, 1, 0], [0, 4, -4, -1, 1, 0], [0, -2, -1, 2, 1, 0],
                       [0, 2, -1, -2, 1, 0], [0, 4, 0, -5, 0, 1]], dt)
        G = np.array([[1 / 4, 0, 0], [-1 / 6, -1 / 6, -1 / 6], [-1 / 6, 1 / 6, -1 / 6], [1 / 24, 1 / 12, 1 / 6],
                      [1 / 24, -1 / 12, 1 / 6], [0, 0, 1]], dt)
        AT = np.array([[1, 1, 1, 1, 1, 0], [0, 1, -1, 2, -2, 0], [0, 1, 1, 4, 4, 0], [0, 1, -1, 8, -8, 1]], dt)
    else:           # m == -4: points 0, 1, -1, 1/2, -2, inf (csrc/wino4.h)
        m = 4
        BT = np.array([[1, -1.5, -2, 1.5, 1, 0], [0, -1, .5, 2.5, 1, 0], [0, 1, -2.5, .5, 1, 0], [0, -2, -1, 2, 1, 0],
                       [0, .5, -1, -.5, 1, 0], [0, 1, -1.5, -2, 1.5, 1]], dt)
        G = np.array([[1, 0, 0], [1 / 3, 1 / 3, 1 / 3], [-1 / 3, 1 / 3, -1 / 3], [-16 / 15, -8 / 15, -4 / 15],
                      [1 / 15, -2 / 15, 4 / 15], [0, 0, 1]], dt)
        AT = np.array([[1, 1, 1, 1, 1, 0], [0, 1, -1, .5, -2, 0], [0, 1, 1, .25, 4, 0], [0, 1, -1, .125, -8, 1]], dt)
    a = m + 2
    x = x.astype(dt); w = w.astype(dt)
    U = np.einsum('ij,ocjk,lk->ocil', G, w, G).astype(dt)            # [o, c, a, a]
    xp = np.pad(x, ((0, 0), (0, 0), (1, 1), (1, 1)))
    out = np.zeros((x.shape[0], w.shape[0], H, W), dt)
    for th in range(H // m):
        for tw in range(W // m):
            d = xp[:, :, th * m:th * m + a, tw * m:tw * m + a]
            V = np.einsum('ij,ncjk,lk->ncil', BT, d, BT).astype(dt)
            M = np.einsum('ncil,ocil->noil', V, U).astype(dt)        # fp32 accumulate over channels
            Y = np.einsum('ij,nojk,lk->noil', AT, M, AT).astype(dt)
            out[:, :, th * m:(th + 1) * m, tw * m:(tw + 1) * m] = Y
    return out


ref = direct(x, w, np.float64)
scale = np.abs(ref).max()
for name, got in (('direct fp32', direct(x, w, np.float32)), ('F(2x2,3x3) fp32', winograd(x, w, 2)),
                  ('F(4x4,3x3) fp32, points 0 +-1 +-2', winograd(x, w, 4)), ('F(4x4,3x3) fp32, points 0 1 -1 1/2 -2', winograd(x, w, -4))):
    err = np.abs(got.astype(np.float64) - ref)
    print('%-38s max err / max|y| = %.2e   rms err / rms y = %.2e' % (name, err.max() / scale, np.sqrt((err ** 2).mean()) / np.sqrt((ref ** 2).mean())))
