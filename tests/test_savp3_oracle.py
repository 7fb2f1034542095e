"""arch = 'savp3' (the published SAVP generator) on the CPU: the oracle's reading of the public cell against naive loops, the
product's tensor table against the oracle's and the library's, and the algebraic identities the engine's fused forms rest on
(conv + average pool == one stride-2 convolution with the box-filtered kernel; tile-concatenated conditioning == border-class
bias tables for the pooled conv, the up-sampled conv and the conv-LSTM; bilinear up-sampling == the four-tap formula)."""
import ctypes

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle.savp3_predictor import OracleSavp3, expected_shapes, bilinear_kernel, IN_EPS
from visual_foresight_amd import _lib
from visual_foresight_amd.video_prediction.cdna_arch import CdnaWeights
from visual_foresight_amd.video_prediction.savp3_arch import Savp3Config


def _vfconfig(cfg, max_batch=4):
    return _lib.VfConfig(cfg.height, cfg.width, cfg.adim, cfg.sdim, cfg.ndesig, cfg.n_context, cfg.sequence_length,
                         cfg.num_masks, max_batch, 0, 0, 1, 1, cfg.arch_id, cfg.zdim, cfg.layer_spec)


@pytest.mark.parametrize('H,W,spec,cells', [(32, 32, 0, 3), (64, 64, 0, 5), (48, 64, 0, 3), (64, 96, 0, 5), (128, 128, 0, 6),
                                            (128, 128, 64, 5)])
def test_tensor_table_matches_the_oracles_and_the_librarys(H, W, spec, cells):
    cfg = Savp3Config(height=H, width=W, adim=12, sdim=5, ndesig=2, sequence_length=5, layer_spec=spec)
    assert {k: tuple(v) for k, v in cfg.tensor_shapes().items()} == expected_shapes(cfg)
    assert sum(1 for k in cfg.tensor_shapes() if k.endswith('l/w')) == cells
    w = CdnaWeights.random(cfg, seed=0)
    lib = _lib.load_library()
    c = _vfconfig(cfg)
    assert lib.vf_weight_count(ctypes.byref(c)) == w.n_floats()
    assert lib.vf_macs_per_sample_step(ctypes.byref(c)) == pytest.approx(sum(cfg.macs_per_sample_step().values()), rel=1e-12)
    assert sum(cfg.executed_macs_per_sample_step().values()) < sum(cfg.macs_per_sample_step().values())


def test_paper_network_is_the_five_cell_table():
    """arXiv:1804.01523 appendix A at 64 x 64: conv-LSTMs of 32 / 64 / 128 / 64 / 32 channels at 32 / 16 / 8 / 16 / 32 pixels."""
    cfg = Savp3Config(height=64, width=64, adim=12)
    cells = [(C, ho) for i, kind, k, cin, C, rnn, _, (ho, wo) in cfg.layer_table() if rnn]
    assert cells == [(32, 32), (64, 16), (128, 8), (64, 16), (32, 32)]
    assert [k for _, _, k, *_ in cfg.layer_table()] == [5, 3, 3, 3, 3, 3]
    assert cfg.ncond == 4 + 5 + 8


def test_invalid_configurations_are_refused():
    lib = _lib.load_library()
    with pytest.raises(ValueError):
        Savp3Config(height=24, width=32)
    with pytest.raises(ValueError):
        Savp3Config(height=64, width=64, adim=4, zdim=8)
    with pytest.raises(ValueError):
        Savp3Config(height=72, width=64, layer_spec=128)
    for bad, msg in ((dict(num_masks=10), b'num_masks'), (dict(zdim=0), b'zdim'), (dict(precision=1), b'precision'),
                     (dict(height=24), b'32'), (dict(layer_spec=48), b'layer_spec')):
        kw = dict(height=64, width=64, adim=12, sdim=5, ndesig=1, n_context=2, sequence_length=5, num_masks=4, max_batch=4,
                  device=0, precision=0, ncam=1, n_draws=1, arch=3, zdim=8, layer_spec=0)
        kw.update(bad)
        c = _lib.VfConfig(*[kw[n] for n, _ in _lib.VfConfig._fields_])
        assert lib.vf_weight_count(ctypes.byref(c)) == 0 and msg in lib.vf_last_error(), bad
    c = _lib.VfConfig(64, 64, 4, 5, 1, 2, 5, 10, 4, 0, 0, 1, 1, 0, 8, 0)       # zdim belongs to arch 3
    assert lib.vf_weight_count(ctypes.byref(c)) == 0 and b'zdim' in lib.vf_last_error()
    with pytest.raises(ValueError):         # a network of another table is not this network
        OracleSavp3(CdnaWeights.random(Savp3Config(height=64, width=64, adim=12, layer_spec=32), seed=0).__class__(
            Savp3Config(height=64, width=64, adim=12), CdnaWeights.random(Savp3Config(height=64, width=64, adim=12, layer_spec=32),
                                                                           seed=0).tensors))


# ------------------------------------------------------------------ the identities behind the engine's fused forms
def _conv_same(x, w):          # x [1, Cin, H, W] float64, w [k, k, Cin, Cout]
    k = w.shape[0]
    return F.conv2d(x, torch.from_numpy(w).permute(3, 2, 0, 1).contiguous(), padding=(k - 1) // 2)


def _box(w):
    k = w.shape[0]
    out = np.zeros((k + 1, k + 1) + w.shape[2:])
    for a in range(2):
        for b in range(2):
            out[a:a + k, b:b + k] += 0.25 * w
    return out


@pytest.mark.parametrize('k,H,W', [(5, 16, 24), (3, 8, 12), (5, 8, 8)])
def test_conv_then_average_pool_is_one_stride2_conv_with_the_box_filtered_kernel(k, H, W):
    """vf_engine_savp3.inc, s3_box_filter: K'[u][v] = 1/4 sum_{a,b in {0,1}} K[u - a][v - b], stride 2, pad (k - 1) / 2 on
    the top / left (k - 1) / 2 + 1 ... on the other side (TensorFlow SAME of the stride-1 conv, no padding in the pool)."""
    rs = np.random.RandomState(k * H)
    x = torch.from_numpy(rs.normal(0, 1, (1, 3, H, W)))
    w = rs.normal(0, 0.3, (k, k, 3, 4))
    want = F.avg_pool2d(_conv_same(x, w), 2)
    p = (k - 1) // 2
    xp = F.pad(x, (p, p, p, p))
    got = F.conv2d(xp, torch.from_numpy(_box(w)).permute(3, 2, 0, 1).contiguous(), stride=2)
    np.testing.assert_allclose(got.numpy(), want.numpy(), rtol=1e-12, atol=1e-12)


def _cls5(y, H):
    return y if y < 2 else (y - (H - 5) if y >= H - 2 else 2)


def _table(t, f):              # t [KH, KH, C], f [5, KH] -> [5, 5, C]
    return np.einsum('yt,xs,tsc->yxc', f, f, t)


def _f_lstm():
    return np.array([[0, 0, 1, 1, 1], [0, 1, 1, 1, 1], [1, 1, 1, 1, 1], [1, 1, 1, 1, 0], [1, 1, 1, 0, 0]], float)


def _f_pool(k):
    K, p = k + 1, (k - 1) // 2
    f = np.ones((5, K))
    f[0, :p] = 0
    f[4, p + 2:] = 0
    return f


def _f_up():
    return np.array([[0, .75, 1], [.75, 1, 1], [1, 1, 1], [1, 1, .75], [1, .75, 0]])


def _upsample(x):
    C = x.shape[1]
    k = torch.from_numpy(bilinear_kernel()).view(1, 1, 4, 4).expand(C, 1, 4, 4).contiguous()
    return F.conv_transpose2d(x, k, stride=2, padding=1, groups=C)


@pytest.mark.parametrize('H,W', [(4, 4), (8, 12), (5, 9)])
def test_border_class_tables_equal_the_convolutions_over_the_tiled_vector(H, W):
    """cond3_item + the element-wise items (vf_savp3.h): for a spatially constant input the layer's output is
    table[class(y)][class(x)], table = sum_ty f[ry][ty] sum_tx f[rx][tx] (W . v)[ty][tx] - for the conv-LSTM's 5 x 5 conv,
    for conv + pool (as the fused stride-2 conv) and for bilinear up-sampling + 3 x 3 conv (output size 2H x 2W)."""
    rs = np.random.RandomState(H * W)
    nc, C = 17, 6
    v = rs.normal(0, 1, nc)
    tile = lambda h, w: torch.from_numpy(np.broadcast_to(v[:, None, None], (nc, h, w)).copy())[None]
    # conv-LSTM
    Wl = rs.normal(0, 0.2, (5, 5, nc, C))
    want = _conv_same(tile(H, W), Wl)[0].numpy()
    tab = _table(np.einsum('yxck,c->yxk', Wl, v), _f_lstm())
    got = np.stack([[tab[_cls5(y, H), _cls5(x, W)] for x in range(W)] for y in range(H)])
    np.testing.assert_allclose(got.transpose(2, 0, 1), want, rtol=1e-12, atol=1e-12)
    # conv k x k + pool on an input of 2H x 2W -> H x W
    for k in (5, 3):
        Wk = rs.normal(0, 0.2, (k, k, nc, C))
        want = F.avg_pool2d(_conv_same(tile(2 * H, 2 * W), Wk), 2)[0].numpy()
        tab = _table(np.einsum('yxck,c->yxk', _box(Wk), v), _f_pool(k))
        got = np.stack([[tab[_cls5(y, H), _cls5(x, W)] for x in range(W)] for y in range(H)])
        np.testing.assert_allclose(got.transpose(2, 0, 1), want, rtol=1e-12, atol=1e-12)
    # bilinear up-sampling + 3 x 3 conv, output 2H x 2W
    Wu = rs.normal(0, 0.2, (3, 3, nc, C))
    want = _conv_same(_upsample(tile(H, W)), Wu)[0].numpy()
    tab = _table(np.einsum('yxck,c->yxk', Wu, v), _f_up())
    got = np.stack([[tab[_cls5(y, 2 * H), _cls5(x, 2 * W)] for x in range(2 * W)] for y in range(2 * H)])
    np.testing.assert_allclose(got.transpose(2, 0, 1), want, rtol=1e-12, atol=1e-12)


def test_bilinear_upsampling_is_the_four_tap_formula():
    """upsample_item (vf_savp3.h): out(2i + a) = a ? .75 s[i] + .25 s[i+1] : .25 s[i-1] + .75 s[i], zero outside, per axis."""
    rs = np.random.RandomState(1)
    x = rs.normal(0, 1, (3, 5, 7))
    want = _upsample(torch.from_numpy(x)[None])[0].numpy()
    xp = np.pad(x, ((0, 0), (1, 1), (1, 1)))
    got = np.zeros((3, 10, 14))
    for Y in range(10):
        i, a = divmod(Y, 2)
        r = [(i - 1 + a, .75 if a else .25), (i + a, .25 if a else .75)]
        for X in range(14):
            j, b = divmod(X, 2)
            q = [(j - 1 + b, .75 if b else .25), (j + b, .25 if b else .75)]
            got[:, Y, X] = sum(wy * wx * xp[:, ry + 1, qx + 1] for ry, wy in r for qx, wx in q)
    np.testing.assert_allclose(got, want, rtol=1e-12, atol=1e-12)


# ------------------------------------------------------------------ the oracle's blocks against naive loops
def _tiny():
    cfg = Savp3Config(height=32, width=32, adim=6, sdim=3, zdim=2, ndesig=1, sequence_length=4)
    w = CdnaWeights.random(cfg, seed=4, bias_scale=0.1, ln_jitter=0.3)
    return cfg, w, OracleSavp3(w, torch.float64)


def test_instance_norm_and_cell_against_naive_numpy():
    cfg, w, ora = _tiny()
    rs = np.random.RandomState(0)
    x = rs.normal(0.3, 2.0, (2, 32, 5, 7))
    g, b = w.tensors['h0n/g'].astype(np.float64), w.tensors['h0n/b'].astype(np.float64)
    want = np.empty_like(x)
    for n in range(2):
        for c in range(32):
            m, v = x[n, c].mean(), x[n, c].var()
            want[n, c] = (x[n, c] - m) / np.sqrt(v + IN_EPS) * g[c] + b[c]
    np.testing.assert_allclose(ora._inorm(torch.from_numpy(x), 'h0n').numpy(), want, rtol=1e-10, atol=1e-10)
    # the cell: gates = IN(conv5x5([u | h])), split i, j, f, o; c = IN(c_prev sig(f + 1) + sig(i) tanh(j)); h = tanh(c) sig(o)
    C, nc = 32, cfg.ncond
    u = rs.normal(0, 1, (1, C + nc, 6, 6)); hp = rs.normal(0, 1, (1, C, 6, 6)); cp = rs.normal(0, 1, (1, C, 6, 6))
    h_new, (c_new, _) = ora._convlstm(torch.from_numpy(u), (torch.from_numpy(cp), torch.from_numpy(hp)), 0)
    Wl = w.tensors['h0l/w'].astype(np.float64)
    xin = np.pad(np.concatenate([u, hp], axis=1)[0], ((0, 0), (2, 2), (2, 2)))
    gates = np.zeros((4 * C, 6, 6))
    for y in range(6):
        for x_ in range(6):
            gates[:, y, x_] = np.einsum('yxc,yxck->k', xin[:, y:y + 5, x_:x_ + 5].transpose(1, 2, 0), Wl)
    lg, lb = w.tensors['h0lg/g'].astype(np.float64), w.tensors['h0lg/b'].astype(np.float64)
    gates = (gates - gates.mean(axis=(1, 2), keepdims=True)) / np.sqrt(gates.var(axis=(1, 2), keepdims=True) + IN_EPS)
    gates = gates * lg[:, None, None] + lb[:, None, None]
    sig = lambda a: 1 / (1 + np.exp(-a))
    i, j, f, o = gates[:C], gates[C:2 * C], gates[2 * C:3 * C], gates[3 * C:]
    c = cp[0] * sig(f + 1.0) + sig(i) * np.tanh(j)
    cg, cb = w.tensors['h0lc/g'].astype(np.float64), w.tensors['h0lc/b'].astype(np.float64)
    c = (c - c.mean(axis=(1, 2), keepdims=True)) / np.sqrt(c.var(axis=(1, 2), keepdims=True) + IN_EPS) * cg[:, None, None] + cb[:, None, None]
    np.testing.assert_allclose(c_new[0].numpy(), c, rtol=1e-9, atol=1e-9)
    np.testing.assert_allclose(h_new[0].numpy(), np.tanh(c) * sig(o), rtol=1e-9, atol=1e-9)


def test_rnn_z_is_a_basic_lstm_cell():
    cfg, w, ora = _tiny()
    rs = np.random.RandomState(1)
    z, c, h = rs.normal(0, 1, (3, 2)), rs.normal(0, 1, (3, 2)), rs.normal(0, 1, (3, 2))
    out, (c_new, h_new) = ora._rnn_z(torch.from_numpy(z), (torch.from_numpy(c), torch.from_numpy(h)))
    g = np.concatenate([z, h], axis=1) @ w.tensors['rnnz/w'].astype(np.float64) + w.tensors['rnnz/b'].astype(np.float64)
    sig = lambda a: 1 / (1 + np.exp(-a))
    i, j, f, o = np.split(g, 4, axis=1)
    cn = c * sig(f + 1.0) + sig(i) * np.tanh(j)
    np.testing.assert_allclose(c_new.numpy(), cn, rtol=1e-12)
    np.testing.assert_allclose(out.numpy(), np.tanh(cn) * sig(o), rtol=1e-12)


def test_compositing_follows_the_published_order_with_symmetric_warps_and_dependent_masks():
    """One-hot masks (zero mask kernel, +-80 biases) show every compositing layer on its own: [warp_0..3, previous, first,
    scratch]; the distributions take the previous distribution in the scratch slot; the warps read the symmetrically padded
    image; and with a non-zero mask kernel on the layer channels the masks depend on the layers."""
    cfg = Savp3Config(height=32, width=32, adim=6, sdim=3, zdim=2, ndesig=1, sequence_length=4)
    w = CdnaWeights.random(cfg, seed=2, bias_scale=0.05, ln_jitter=0.1)
    rs = np.random.RandomState(0)
    f32 = lambda a: torch.from_numpy(a.astype(np.float32))
    frame, first = f32(rs.uniform(0, 1, (1, 3, 32, 32))), f32(rs.uniform(0, 1, (1, 3, 32, 32)))
    distrib = f32(rs.uniform(0, 1, (1, 1, 32, 32))); distrib /= distrib.sum()
    first_d = f32(rs.uniform(0, 1, (1, 1, 32, 32))); first_d /= first_d.sum()
    state, action = torch.zeros(1, 3), f32(rs.normal(0, 0.1, (1, 6)))

    def step(tensors):
        ora = OracleSavp3(CdnaWeights(cfg, tensors))
        sizes = [(32, 16), (64, 8), (32, 16)]
        lstm = [(torch.zeros(1, C, r, r), torch.zeros(1, C, r, r)) for C, r in sizes]
        return ora, ora.step(frame, distrib, state, action, (lstm, (torch.zeros(1, 2), torch.zeros(1, 2))), first, first_d)

    outs, outs_d = [], []
    for hot in range(7):
        t = dict(w.tensors)
        t['masks/w'] = np.zeros_like(t['masks/w'])
        b = np.full(7, -80.0, np.float32); b[hot] = 80.0
        t['masks/b'] = b
        ora, (nf, nd, _, _) = step(t)
        outs.append(nf[0].numpy()); outs_d.append(nd[0].numpy())
    np.testing.assert_allclose(outs[4], frame[0].numpy(), atol=1e-6)
    np.testing.assert_allclose(outs[5], first[0].numpy(), atol=1e-6)
    np.testing.assert_allclose(outs_d[4], distrib[0].numpy(), rtol=1e-5, atol=1e-9)
    np.testing.assert_allclose(outs_d[5], first_d[0].numpy(), rtol=1e-5, atol=1e-9)
    np.testing.assert_allclose(outs_d[6], distrib[0].numpy(), rtol=1e-5, atol=1e-9)     # the scratch slot
    assert outs[6].min() >= 0 and outs[6].max() <= 1 and np.abs(outs[6] - frame[0].numpy()).max() > 0.1
    # warp k: a normalised 5 x 5 kernel over the SYMMETRICALLY padded previous frame - a constant image stays constant, and
    # the warp of the real frame equals the naive loop over np.pad(..., mode='symmetric')
    fr = frame[0].numpy()
    pad = np.pad(fr, ((0, 0), (2, 2), (2, 2)), mode='symmetric')
    for k in range(4):
        # recover kernel k from a delta image is not possible with symmetric padding at the border: use least squares on
        # the interior instead - the centre 20 x 20 pixels see an ordinary correlation
        A = np.stack([pad[0, 6 + dy:26 + dy, 6 + dx:26 + dx].ravel() for dy in range(5) for dx in range(5)], axis=1).astype(np.float64)
        kern = np.linalg.lstsq(A, outs[k][0, 6:26, 6:26].ravel().astype(np.float64), rcond=None)[0].reshape(5, 5)
        assert kern.min() > -1e-4 and abs(kern.sum() - 1) < 1e-4
        naive = np.zeros_like(fr)
        for dy in range(5):
            for dx in range(5):
                naive += kern[dy, dx] * pad[:, dy:dy + 32, dx:dx + 32]
        np.testing.assert_allclose(outs[k], naive, atol=2e-4)                 # (borders included: symmetric, not zero, padding)
    # dependent masks: the mask head sees the layers
    t = dict(w.tensors)
    _, (base, _, _, _) = step(t)
    t2 = dict(t)
    mw = t2['masks/w'].copy(); mw[:, :, 32:, :] = 0.0
    t2['masks/w'] = mw
    _, (indep, _, _, _) = step(t2)
    assert np.abs(base.numpy() - indep.numpy()).max() > 1e-4


def test_rollout_feeds_context_then_its_own_predictions():
    cfg, w, _ = _tiny()
    ora = OracleSavp3(w, torch.float32)
    rs = np.random.RandomState(3)
    H = W = 32
    ctx_f = rs.randint(0, 256, (3, 1, H, W, 3)).astype(np.uint8)
    d = np.zeros((2, 1, H, W, 1), np.float32); d[:, 0, 10, 12, 0] = 1
    acts = rs.normal(0, 0.1, (2, 2, 6))
    f, dd, s = ora.rollout(ctx_f, rs.normal(0, 0.05, (2, 6)), d, rs.normal(0, 0.1, (3, 3)), acts)
    assert f.shape == (2, 2, 1, H, W, 3) and dd.shape == (2, 2, 1, H, W, 1) and s.shape == (2, 2, 3)
    np.testing.assert_allclose(dd.sum(axis=(3, 4)), 1.0, atol=1e-5)
    assert f.min() >= 0.0 and f.max() <= 1.0
    # the first prediction does not depend on later actions; the second does
    acts2 = acts.copy(); acts2[:, 1] += 0.5
    f2, _, _ = ora.rollout(ctx_f, np.zeros((2, 6)) + 0.01, d, np.zeros((3, 3)), acts2)
    f1, _, _ = ora.rollout(ctx_f, np.zeros((2, 6)) + 0.01, d, np.zeros((3, 3)), acts)
    np.testing.assert_array_equal(f1[:, 0], f2[:, 0])
    assert np.abs(f1[:, 1] - f2[:, 1]).max() > 1e-6
