"""CPU check of the Winograd F(4x4,3x3) transform matrices compiled into the HIP library (csrc/wino4.h): parsed out of
the header and verified, in fp64, against the identity they must satisfy -- the 2-D correlation of a 6x6 patch with a
3x3 filter equals A^T [ (G g G^T) * (B^T d B) ] A -- and against the interpolation points the header names."""
import os
import re

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _matrix(text, name):
    m = re.search(name + r'\[\d+\]\[\d+\]\s*=\s*\{(.*?)\};', text, re.S)
    assert m, name
    rows = re.findall(r'\{([^{}]*)\}', m.group(1))
    out = []
    for r in rows:
        vals = []
        for tok in r.split(','):
            tok = tok.strip().rstrip('f')
            if not tok:
                continue
            vals.append(float(eval(tok, {'__builtins__': {}})))      # entries like 1.0 / 3 and -16.0 / 15
        out.append(vals)
    return np.array(out, dtype=np.float64)


def test_header_matrices_satisfy_the_winograd_identity():
    text = open(os.path.join(ROOT, 'neural-ode-features_amd', 'csrc', 'wino4.h')).read()
    BT, AT, G = _matrix(text, 'W4_BT'), _matrix(text, 'W4_AT'), _matrix(text, 'W4_G')
    assert BT.shape == (6, 6) and AT.shape == (4, 6) and G.shape == (6, 3)
    rng = np.random.default_rng(0)
    for _ in range(5):
        d = rng.standard_normal((6, 6))
        g = rng.standard_normal((3, 3))
        U = G @ g @ G.T
        V = BT @ d @ BT.T
        Y = AT @ (U * V) @ AT.T
        ref = np.array([[sum(d[i + a, j + b] * g[a, b] for a in range(3) for b in range(3)) for j in range(4)]
                        for i in range(4)])
        assert np.abs(Y - ref).max() < 1e-12 * max(1.0, np.abs(ref).max())


def test_header_matrices_belong_to_the_points_the_header_names():
    """A^T evaluates at the points (0, 1, -1, 1/2, -2) and picks the leading coefficient for the point at infinity."""
    text = open(os.path.join(ROOT, 'neural-ode-features_amd', 'csrc', 'wino4.h')).read()
    AT = _matrix(text, 'W4_AT')
    pts = [0.0, 1.0, -1.0, 0.5, -2.0]
    for i in range(4):
        for j, p in enumerate(pts):
            assert abs(AT[i, j] - p ** i) < 1e-15
        assert AT[i, 5] == (1.0 if i == 3 else 0.0)
