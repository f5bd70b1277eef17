import numpy as np
rng = np.random.default_rng(0)
C, H, W, N = 256, 8, 8, 4
x = np.maximum(rng.standard_normal((N, C, H, W)), 0).astype(np.float32)
w = (rng.uniform(-1, 1, (C, C, 3, 3)) / 48).astype(np.float32)
BT = np.array([[1, -1.5, -2, 1.5, 1, 0], [0, -1, .5, 2.5, 1, 0], [0, 1, -2.5, .5, 1, 0], [0, -2, -1, 2, 1, 0],
               [0, .5, -1, -.5, 1, 0], [0, 1, -1.5, -2, 1.5, 1]], np.float64)
G = np.array([[1, 0, 0], [1 / 3, 1 / 3, 1 / 3], [-1 / 3, 1 / 3, -1 / 3], [-16 / 15, -8 / 15, -4 / 15],
              [1 / 15, -2 / 15, 4 / 15], [0, 0, 1]], np.float64)
AT = np.array([[1, 1, 1, 1, 1, 0], [0, 1, -1, .5, -2, 0], [0, 1, 1, .25, 4, 0], [0, 1, -1, .125, -8, 1]], np.float64)

def direct(x, w):
    x = x.astype(np.float64); w = w.astype(np.float64)
    xp = np.pad(x, ((0, 0), (0, 0), (1, 1), (1, 1)))
    out = np.zeros((x.shape[0], w.shape[0], H, W))
    for kh in range(3):
        for kw in range(3):
            out += np.einsum('nchw,oc->nohw', xp[:, :, kh:kh + H, kw:kw + W], w[:, :, kh, kw])
    return out

def bf16(v):
    u = v.astype(np.float32).view(np.uint32)
    r = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16) << 16
    return r.astype(np.uint32).view(np.float32)

def split_bf3(v):
    v = v.astype(np.float32)
    h = bf16(v); r = v - h; m = bf16(r); l = bf16(r - m)
    return h.astype(np.float64), m.astype(np.float64), l.astype(np.float64)

def split_f16(v, scale):
    v = (v.astype(np.float32) * np.float32(scale))
    h = v.astype(np.float16); r = v - h.astype(np.float32); l = r.astype(np.float16)
    assert np.isfinite(h).all()
    return h.astype(np.float64) / scale, l.astype(np.float64) / scale

def gemm(V, U, mode, sv=1.0, su=1.0, acc32=False):
    # V [n c i l], U [o c i l] -> M[n o i l]
    def mm(a, b):
        if acc32:
            return np.einsum('ncil,ocil->noil', a.astype(np.float32), b.astype(np.float32)).astype(np.float64)
        return np.einsum('ncil,ocil->noil', a, b)
    if mode == 'f32':
        return mm(V.astype(np.float64), U.astype(np.float64))
    if mode == 'bf3':
        vh, vm, vl = split_bf3(V); uh, um, ul = split_bf3(U)
        return mm(vl, uh) + mm(vh, ul) + mm(vm, um) + mm(vm, uh) + mm(vh, um) + mm(vh, uh)
    if mode in ('h2', 'h2x4'):
        vh, vl = split_f16(V, sv); uh, ul = split_f16(U, su)
        r = mm(vl, uh) + mm(vh, ul) + mm(vh, uh)
        if mode == 'h2x4': r = r + mm(vl, ul)
        return r
    if mode == 'bf2':   # 3 products of bf16
        vh, vm, vl = split_bf3(V); uh, um, ul = split_bf3(U)
        return mm(vm, uh) + mm(vh, um) + mm(vh, uh)

def winograd(x, w, mode, xs=1.0, **kw):
    U = np.einsum('ij,ocjk,lk->ocil', G, w.astype(np.float64), G).astype(np.float32)
    xp = np.pad(x, ((0, 0), (0, 0), (1, 1), (1, 1)))
    out = np.zeros((x.shape[0], w.shape[0], H, W))
    vmax = 0
    for th in range(2):
        for tw in range(2):
            d = xp[:, :, th * 4:th * 4 + 6, tw * 4:tw * 4 + 6]
            V = np.einsum('ij,ncjk,lk->ncil', BT.astype(np.float32), d, BT.astype(np.float32)).astype(np.float32)
            vmax = max(vmax, np.abs(V).max())
            M = gemm(V, U, mode, **kw).astype(np.float32)
            Y = np.einsum('ij,nojk,lk->noil', AT.astype(np.float32), M, AT.astype(np.float32)).astype(np.float32)
            out[:, :, th * 4:(th + 1) * 4, tw * 4:(tw + 1) * 4] = Y
    return out, vmax, np.abs(U).max()

for xs in (1.0, 1e-4, 1e3):
    xx = (x * np.float32(xs))
    ref = direct(xx, w); scale = np.abs(ref).max()
    print('input scale', xs)
    o, vmax, umax = winograd(xx, w, 'f32')
    sv = 2.0 ** np.floor(np.log2(16384 / vmax)); su = 2.0 ** np.floor(np.log2(16384 / umax))
    print('  vmax %.3g umax %.3g sv %g su %g' % (vmax, umax, sv, su))
    for mode, kw in (('f32', {}), ('bf3', {}), ('bf2', {}), ('h2', dict(sv=sv, su=su)), ('h2x4', dict(sv=sv, su=su)),
                     ('h2', dict(sv=sv / 1024, su=su / 1024)), ('h2', dict(sv=sv / 2**20, su=su)), ('f32', dict(acc32=True)), ('h2', dict(sv=sv, su=su, acc32=True))):
        got, _, _ = winograd(xx, w, mode, **kw)
        err = np.abs(got - ref)
        print('  %-6s %-40s max err/max|y| = %.2e  rms rel = %.2e' % (mode, str({k: (v if isinstance(v, bool) else float(np.log2(v))) for k, v in kw.items()}), err.max() / scale, np.sqrt((err ** 2).mean()) / np.sqrt((ref ** 2).mean())))
