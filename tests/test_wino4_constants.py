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


def test_weight_gradient_identity_in_the_f4x4_domain():
    """What k_w4_wgrad + k_theta_finalize compute (csrc/kernels_w4.hip): with V = B^T d B the conv's own row operand and
    Z = A dz A^T of the output cotangent, dW = G^T (sum over tiles V * Z) G is the weight gradient of the 3x3 correlation
    -- checked in fp64 on an 8x8 image (2x2 tiles with the zero halo the pipeline uses) against the direct sum."""
    text = open(os.path.join(ROOT, 'neural-ode-features_amd', 'csrc', 'wino4.h')).read()
    BT, AT, G = _matrix(text, 'W4_BT'), _matrix(text, 'W4_AT'), _matrix(text, 'W4_G')
    rng = np.random.default_rng(1)
    x = rng.standard_normal((8, 8))
    dz = rng.standard_normal((8, 8))
    xp = np.pad(x, 1)
    ref = np.array([[sum(dz[i, j] * xp[i + a, j + b] for i in range(8) for j in range(8)) for b in range(3)] for a in range(3)])
    dU = np.zeros((6, 6))
    for ty in range(2):
        for tx in range(2):
            d = xp[4 * ty:4 * ty + 6, 4 * tx:4 * tx + 6]
            V = BT @ d @ BT.T
            Z = AT.T @ dz[4 * ty:4 * ty + 4, 4 * tx:4 * tx + 4] @ AT
            dU += V * Z
    dW = G.T @ dU @ G
    assert np.abs(dW - ref).max() < 1e-11 * np.abs(ref).max()


def test_bf16_triples_are_exact_and_six_products_carry_fp32_accuracy():
    """The scheme of k_w4_gemm64b: every fp32 value is the EXACT sum of three bf16 values (round-to-nearest splits of the
    successive remainders), and the six products hh, hm, mh, mm, hl, lh reproduce an fp32 x fp32 product to ~2^-22: against
    an fp64 matrix product the error from the three dropped products is far below the fp32 accumulation rounding."""
    import torch
    gen = torch.Generator().manual_seed(0)

    def split3(x):
        h = x.to(torch.bfloat16)
        r = x - h.float()
        m = r.to(torch.bfloat16)
        s = r - m.float()
        lo = s.to(torch.bfloat16)
        return h.float(), m.float(), lo.float()

    A = torch.randn(256, 128, generator=gen) * 3
    B = (torch.rand(128, 64, generator=gen) * 2 - 1) / 48
    ah, am, al = split3(A)
    bh, bm, bl = split3(B)
    assert torch.equal(ah + am + al, A) and torch.equal(bh + bm + bl, B)
    ref = A.double() @ B.double()
    six = sum(p.double() @ q.double() for p, q in ((ah, bh), (ah, bm), (am, bh), (am, bm), (ah, bl), (al, bh)))
    assert float((six - ref).abs().max() / ref.abs().max()) < 2e-7 / 4          # truncation of the dropped terms alone
    three = sum(p.double() @ q.double() for p, q in ((ah, bh), (ah, bm), (am, bh)))
    assert float((three - ref).abs().max() / ref.abs().max()) > 1e-6            # ... which three products would not give
