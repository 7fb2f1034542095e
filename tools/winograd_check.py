#!/usr/bin/env python
"""Accuracy gate for VERDICT r2 item 8: would Winograd F(2x2, 5x5) in fp32 stay within 2x of the direct fp32
convolution's distance to a float64 reference?  (36 multiplies per 2x2 outputs instead of 100.)

Simulates one conv-LSTM gate convolution with the statistics of this network: the input is a LayerNorm-ed
tensor (unit variance) concatenated with a tanh-bounded hidden state, Glorot-uniform weights, K = 25 * Cin.
Filter transforms are done offline in float64 and rounded once (as a real implementation would); the input
transform, the 36 per-frequency channel contractions and the output transform run in float32.
Point sets tried: (0, +-1, +-2, inf) - the classic F(4,3) set - and (0, +-1, +-1/2, inf).
"""
import sys
import numpy as np


def toom_cook(points, m, r):
    """A^T [m, n], G [n, r], B^T [n, n] for y = A^T ((G g) * (B^T d)), n = m + r - 1, last point = infinity."""
    n = m + r - 1
    pts = list(points)
    assert len(pts) == n - 1

    def vander(cols):
        V = np.zeros((n, cols))
        for i, p in enumerate(pts):
            V[i] = [p ** k for k in range(cols)]
        V[n - 1, cols - 1] = 1.0
        return V
    V = vander(n)
    return vander(m).T.copy(), vander(r), np.linalg.inv(V).T.copy()


def run(points, Cin, Cout, H, W, seed=0):
    rs = np.random.RandomState(seed)
    m, r = 2, 5
    AT, G, BT = toom_cook(points, m, r)
    x = np.concatenate([rs.normal(0, 1, (Cin // 2, H + 4, W + 4)), np.tanh(rs.normal(0, 1, (Cin - Cin // 2, H + 4, W + 4)))])
    x[:, :2] = 0; x[:, -2:] = 0; x[:, :, :2] = 0; x[:, :, -2:] = 0          # zero padding
    x = x.astype(np.float32)
    lim = np.sqrt(6.0 / (25 * Cin + 25 * Cout))
    w = rs.uniform(-lim, lim, (Cout, Cin, 5, 5)).astype(np.float32)
    # float64 reference and float32 direct (tap-major accumulation like the MFMA kernel)
    ref = np.zeros((Cout, H, W))
    d32 = np.zeros((Cout, H, W), np.float32)
    for ky in range(5):
        for kx in range(5):
            patch = x[:, ky:ky + H, kx:kx + W]
            ref += np.einsum('oc,chw->ohw', w[:, :, ky, kx].astype(np.float64), patch.astype(np.float64))
            d32 += np.einsum('oc,chw->ohw', w[:, :, ky, kx], patch).astype(np.float32)
    # Winograd: filter transform in float64, rounded once
    U = np.einsum('ik,ockl,jl->ocij', G, w.astype(np.float64), G).astype(np.float32)      # [Cout, Cin, 6, 6]
    BT32, AT32 = BT.astype(np.float32), AT.astype(np.float32)
    out = np.zeros((Cout, H, W), np.float32)
    for ty in range(0, H, 2):
        tiles = np.stack([x[:, ty:ty + 6, tx:tx + 6] for tx in range(0, W, 2)], 0)       # [T, Cin, 6, 6]
        Vt = np.einsum('ik,tckl->tcil', BT32, tiles).astype(np.float32)
        Vt = np.einsum('tcil,jl->tcij', Vt, BT32).astype(np.float32)
        M = np.einsum('ocij,tcij->toij', U, Vt).astype(np.float32)
        Y = np.einsum('ai,toij->toaj', AT32, M).astype(np.float32)
        Y = np.einsum('toaj,bj->toab', Y, AT32).astype(np.float32)
        for i, tx in enumerate(range(0, W, 2)):
            out[:, ty:ty + 2, tx:tx + 2] = Y[i]
    scale = np.abs(ref).max()
    return np.abs(d32 - ref).max() / scale, np.abs(out - ref).max() / scale, \
        np.sqrt(np.mean((d32 - ref) ** 2)) / scale, np.sqrt(np.mean((out - ref) ** 2)) / scale


if __name__ == '__main__':
    print('layer-like shapes: Cin = Cx + Ch, 32 output channels of one gate, 16x16 pixels')
    for name, pts in (('0,+-1,+-2,inf', (0, 1, -1, 2, -2)), ('0,+-1,+-1/2,inf', (0, 1, -1, 0.5, -0.5))):
        for Cin in (64, 96, 192):
            e = [run(pts, Cin, 32, 16, 16, seed=s) for s in range(3)]
            dmax, wmax, drms, wrms = (float(np.mean([x[i] for x in e])) for i in range(4))
            print('points %-16s Cin %3d: direct fp32 max %.2e rms %.2e | winograd fp32 max %.2e rms %.2e | ratio max %.1f rms %.1f'
                  % (name, Cin, dmax, drms, wmax, wrms, wmax / dmax, wrms / drms))
