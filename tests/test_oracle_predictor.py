"""Pin the oracle's reading of the network spec (cdna_arch.py) with naive NumPy loops.

The network arithmetic has no reference source or golden vector (parity unpinned, see
oracle/cdna_predictor.py); these tests at least nail down that the torch-based oracle computes
the layer semantics the spec states - TF 'SAME' padding, the transposed-conv index rule, the
LayerNorm axes, the conv-LSTM cell, the CDNA kernel normalisation and the compositing rule.
"""
import numpy as np
import pytest
import torch

from oracle.cdna_predictor import OracleCdna
from visual_foresight_amd.video_prediction.cdna_arch import CdnaConfig, CdnaWeights, LN_EPS


def _oracle(H=16, W=16, nd=1, seed=1):
    cfg = CdnaConfig(height=H, width=W, ndesig=nd, sequence_length=4)
    w = CdnaWeights.random(cfg, seed=seed, bias_scale=0.1, ln_jitter=0.2)
    return OracleCdna(w, torch.float64), w


def test_oracle_constants_agree_with_the_spec():
    """The oracle carries its own copy of the architecture constants (it must not inherit a typo)."""
    from oracle import cdna_predictor as ocp, savp_predictor as osp
    from visual_foresight_amd.video_prediction import cdna_arch
    for mod in (ocp, osp):
        assert tuple(mod.LSTM_SIZES) == tuple(cdna_arch.LSTM_SIZES) == (32, 32, 64, 64, 128, 64, 32)
        assert mod.RELU_SHIFT == cdna_arch.RELU_SHIFT == 1e-12 and mod.DNA_KERN == cdna_arch.DNA_KERN == 5
    assert ocp.LN_EPS == cdna_arch.LN_EPS == 1e-12


def _nchw(x):
    return torch.from_numpy(np.ascontiguousarray(x.transpose(0, 3, 1, 2)))


def _nhwc(t):
    return t.permute(0, 2, 3, 1).numpy()


def naive_conv_same(x, w, b, stride):
    """x [B,H,W,Ci], w [kh,kw,Ci,Co]; TensorFlow SAME: pad_total = max((ceil(H/s)-1)*s + k - H, 0)."""
    B, H, W, Ci = x.shape
    kh, kw, _, Co = w.shape
    Ho, Wo = -(-H // stride), -(-W // stride)
    pt = max((Ho - 1) * stride + kh - H, 0) // 2
    pl = max((Wo - 1) * stride + kw - W, 0) // 2
    out = np.zeros((B, Ho, Wo, Co))
    for oy in range(Ho):
        for ox in range(Wo):
            acc = np.tile(b, (B, 1)).astype(np.float64)
            for ky in range(kh):
                for kx in range(kw):
                    iy, ix = oy * stride + ky - pt, ox * stride + kx - pl
                    if 0 <= iy < H and 0 <= ix < W:
                        acc += x[:, iy, ix, :] @ w[ky, kx]
            out[:, oy, ox] = acc
    return out


def test_conv_same_padding_and_stride():
    o, w = _oracle()
    rs = np.random.RandomState(0)
    x = rs.normal(size=(2, 16, 16, 3))
    got = _nhwc(o._conv(_nchw(x), 'enc0', 2))
    want = naive_conv_same(x, w.tensors['enc0/w'].astype(np.float64), w.tensors['enc0/b'].astype(np.float64), 2)
    np.testing.assert_allclose(got, want, atol=1e-12)
    x = rs.normal(size=(2, 8, 8, 32))
    got = _nhwc(o._conv(_nchw(x), 'enc1', 2))
    want = naive_conv_same(x, w.tensors['enc1/w'].astype(np.float64), w.tensors['enc1/b'].astype(np.float64), 2)
    np.testing.assert_allclose(got, want, atol=1e-12)


def test_transposed_conv_index_rule():
    o, w = _oracle()
    x = np.random.RandomState(1).normal(size=(2, 4, 4, 128))
    got = _nhwc(o._convt(_nchw(x), 'convt1'))
    wt, b = w.tensors['convt1/w'].astype(np.float64), w.tensors['convt1/b'].astype(np.float64)
    want = np.tile(b, (2, 8, 8, 1))
    for iy in range(4):
        for ix in range(4):
            for ky in range(3):
                for kx in range(3):
                    oy, ox = 2 * iy + ky, 2 * ix + kx
                    if oy < 8 and ox < 8:
                        want[:, oy, ox] += x[:, iy, ix] @ wt[ky, kx]
    np.testing.assert_allclose(got, want, atol=1e-12)


def test_layernorm_axes():
    o, w = _oracle()
    x = np.random.RandomState(2).normal(1.0, 2.0, size=(3, 8, 8, 32))
    got = _nhwc(o._ln(_nchw(x), 'ln2'))
    mean = x.reshape(3, -1).mean(1)[:, None, None, None]
    var = x.reshape(3, -1).var(1)[:, None, None, None]
    want = (x - mean) / np.sqrt(var + LN_EPS) * w.tensors['ln2/g'] + w.tensors['ln2/b']
    np.testing.assert_allclose(got, want, atol=1e-12)


def test_lstm_cell():
    o, w = _oracle()
    rs = np.random.RandomState(3)
    x, c, h = (rs.normal(size=(2, 8, 8, 32)) for _ in range(3))
    h_new, (c_new, _) = o._lstm(_nchw(x), (_nchw(c), _nchw(h)), 'lstm1', 32)
    gates = naive_conv_same(np.concatenate([x, h], -1), w.tensors['lstm1/w'].astype(np.float64),
                            w.tensors['lstm1/b'].astype(np.float64), 1)
    i, j, f, og = np.split(gates, 4, axis=-1)
    sig = lambda v: 1 / (1 + np.exp(-v))
    c_want = c * sig(f + 1.0) + sig(i) * np.tanh(j)
    np.testing.assert_allclose(_nhwc(c_new), c_want, atol=1e-12)
    np.testing.assert_allclose(_nhwc(h_new), np.tanh(c_want) * sig(og), atol=1e-12)


def test_step_compositing_rule():
    """Recompute frame' / distrib' of one step from the oracle's own intermediate tensors."""
    o, w = _oracle(H=16, W=16, nd=2)
    cfg = o.cfg
    rs = np.random.RandomState(4)
    B = 2
    frame = torch.from_numpy(rs.uniform(0, 1, (B, 3, 16, 16)))
    distrib = torch.from_numpy(rs.uniform(0, 1, (B, 2, 16, 16)))
    distrib = distrib / distrib.sum(dim=(2, 3), keepdim=True)
    state, action = torch.from_numpy(rs.normal(size=(B, 5))), torch.from_numpy(rs.normal(size=(B, 4)))
    sizes = [(8, 8)] * 2 + [(4, 4)] * 2 + [(2, 2)] + [(4, 4)] + [(8, 8)]
    lstm = [(torch.zeros(B, C, a, b, dtype=torch.float64),) * 2 for C, (a, b) in zip((32, 32, 64, 64, 128, 64, 32), sizes)]

    captured = {}
    orig_conv = o._conv

    def spy(x, name, stride=1):
        y = orig_conv(x, name, stride)
        if name in ('rgb', 'masks'):
            captured[name] = y
        return y
    o._conv = spy
    orig_ln = o._ln

    def spy_ln(x, name):
        y = orig_ln(x, name)
        if name == 'ln6':
            captured['h5'] = y
        return y
    o._ln = spy_ln
    nf, nd_, ns, _ = o.step(frame, distrib, state, action, lstm)

    K = cfg.num_masks
    masks = torch.softmax(captured['masks'], dim=1).numpy()
    scratch = torch.sigmoid(captured['rgb']).numpy()
    flat = captured['h5'].permute(0, 2, 3, 1).reshape(B, -1).numpy()
    kern = flat @ w.tensors['cdna/w'].astype(np.float64) + w.tensors['cdna/b']
    kern = np.maximum(kern - 1e-12, 0) + 1e-12
    kern = kern.reshape(B, 25, K)
    kern = kern / kern.sum(1, keepdims=True)

    def warp(img, k):           # img [B,C,H,W]; zero padded 5x5 correlation with kernel k
        pad = np.pad(img, ((0, 0), (0, 0), (2, 2), (2, 2)))
        out = np.zeros_like(img)
        for dy in range(5):
            for dx in range(5):
                out += pad[:, :, dy:dy + 16, dx:dx + 16] * kern[:, dy * 5 + dx, k][:, None, None, None]
        return out

    f, d = frame.numpy(), distrib.numpy()
    want_f = masks[:, 0:1] * f + masks[:, 1:2] * scratch
    want_d = masks[:, 0:1] * d
    for k in range(K - 1):
        want_f += masks[:, k + 2:k + 3] * warp(f, k)
        want_d += masks[:, k + 2:k + 3] * warp(d, k)
    want_d /= want_d.sum(axis=(2, 3), keepdims=True)
    np.testing.assert_allclose(nf.numpy(), want_f, atol=1e-12)
    np.testing.assert_allclose(nd_.numpy(), want_d, atol=1e-12)
    sa = np.concatenate([action.numpy(), state.numpy()], 1)
    np.testing.assert_allclose(ns.numpy(), sa @ w.tensors['state/w'].astype(np.float64) + w.tensors['state/b'], atol=1e-12)


def test_rollout_context_semantics():
    """Last n_context frames/states are used, /255; n_context-1 executed actions are prepended."""
    o, w = _oracle(H=16, W=16)
    rs = np.random.RandomState(6)
    frames = rs.randint(0, 256, (5, 1, 16, 16, 3)).astype(np.uint8)
    states = rs.normal(size=(5, 5))
    ctx_actions = rs.normal(size=(4, 4))
    d = np.zeros((2, 1, 16, 16, 1), np.float32)
    d[:, 0, 3, 4, 0] = 1
    acts = rs.normal(0, 0.1, (3, 2, 4))
    a = o.rollout(frames, ctx_actions, d, states, acts)
    b = o.rollout(frames[-2:], ctx_actions[-1:], d, states[-2:], acts)
    for x, y in zip(a, b):
        np.testing.assert_array_equal(x, y)
    assert a[0].shape == (3, 2, 1, 16, 16, 3) and a[1].shape == (3, 2, 1, 16, 16, 1) and a[2].shape == (3, 2, 5)
    np.testing.assert_allclose(a[1].sum(axis=(3, 4)), 1.0, atol=1e-12)
    assert a[0].min() >= 0 and a[0].max() <= 1


# ---------------------------------------------------------------------------------------- SAVP-class generator
def test_savp_step_wiring_and_compositing_rule():
    """Pin oracle/savp_predictor.py's reading of savp_arch.py: the extra encoder / decoder scale around the core
    (recomputed layer by layer from the oracle's own captured tensors with the naive loops above) and the
    first-frame compositing rule."""
    from oracle.savp_predictor import OracleSavp
    from visual_foresight_amd.video_prediction.savp_arch import SavpConfig
    H = W = 32
    cfg = SavpConfig(height=H, width=W, adim=6, ndesig=2, sequence_length=4)
    w = CdnaWeights.random(cfg, seed=7, bias_scale=0.1, ln_jitter=0.2)
    o = OracleSavp(w, torch.float64)
    rs = np.random.RandomState(11)
    B, K = 2, cfg.num_masks
    frame = torch.from_numpy(rs.uniform(0, 1, (B, 3, H, W)))
    first = torch.from_numpy(rs.uniform(0, 1, (B, 3, H, W)))
    distrib = torch.from_numpy(rs.uniform(0, 1, (B, 2, H, W)))
    distrib = distrib / distrib.sum(dim=(2, 3), keepdim=True)
    dfirst = torch.from_numpy(rs.uniform(0, 1, (B, 2, H, W)))
    dfirst = dfirst / dfirst.sum(dim=(2, 3), keepdim=True)
    state, action = torch.from_numpy(rs.normal(size=(B, 5))), torch.from_numpy(rs.normal(size=(B, 6)))
    lstm = [(torch.zeros(B, C, a, b, dtype=torch.float64),) * 2 for C, (a, b) in zip((32, 32, 64, 64, 128, 64, 32), o.core_sizes())]
    assert o.core_sizes() == [(8, 8)] * 2 + [(4, 4)] * 2 + [(2, 2)] + [(4, 4)] + [(8, 8)]

    cap = {}
    orig_conv, orig_convt, orig_ln = o._conv, o._convt, o._ln

    def spy_conv(x, name, stride=1):
        y = orig_conv(x, name, stride)
        cap[name] = (x, y)
        return y

    def spy_convt(x, name):
        y = orig_convt(x, name)
        cap[name] = (x, y)
        return y

    def spy_ln(x, name):
        y = orig_ln(x, name)
        cap[name] = (x, y)
        return y
    o._conv, o._convt, o._ln = spy_conv, spy_convt, spy_ln
    nf, nd_, ns, _ = o.step(frame, distrib, state, action, lstm, first, dfirst)

    T64 = lambda name: w.tensors[name].astype(np.float64)
    relu = lambda v: np.maximum(v, 0)
    # extra encoder scale: enc00 = relu(LNa(conv5x5/2(frame))) feeds enc0 = relu(LN1(conv5x5/2(enc00)))
    x00 = naive_conv_same(_nhwc(frame), T64('enc00/w'), T64('enc00/b'), 2)
    assert x00.shape == (B, H // 2, W // 2, 16)
    np.testing.assert_allclose(_nhwc(cap['enc00'][1]), x00, atol=1e-12)
    enc00 = relu(_nhwc(cap['lna'][1]))
    np.testing.assert_allclose(_nhwc(cap['enc0'][0]), enc00, atol=1e-12)
    np.testing.assert_allclose(_nhwc(cap['enc0'][1]), naive_conv_same(enc00, T64('enc0/w'), T64('enc0/b'), 2), atol=1e-11)
    # extra decoder scale: enc7 = relu(LNb(convT(concat[enc6, enc00]))) with enc6 = relu(LN9(convT3(...)))
    enc6 = relu(_nhwc(cap['ln9'][1]))
    np.testing.assert_allclose(_nhwc(cap['convt4'][0]), np.concatenate([enc6, enc00], -1), atol=1e-12)
    assert cap['convt4'][1].shape == (B, 32, H, W) and cap['convt3'][1].shape == (B, 32, H // 2, W // 2)
    np.testing.assert_allclose(_nhwc(cap['rgb'][0]), relu(_nhwc(cap['lnb'][1])), atol=1e-12)
    # the conditioning vector [action (+ latent), state] is tiled into enc3's input
    np.testing.assert_allclose(_nhwc(cap['enc3'][0])[:, 1, 0, 64:], np.concatenate([action.numpy(), state.numpy()], 1), atol=0)

    masks = torch.softmax(cap['masks'][1], dim=1).numpy()
    scratch = torch.sigmoid(cap['rgb'][1]).numpy()
    flat = cap['ln6'][1].permute(0, 2, 3, 1).reshape(B, -1).numpy()
    kern = flat @ T64('cdna/w') + T64('cdna/b')
    kern = np.maximum(kern - 1e-12, 0) + 1e-12
    kern = kern.reshape(B, 25, K)
    kern = kern / kern.sum(1, keepdims=True)

    def warp(img, k):
        pad = np.pad(img, ((0, 0), (0, 0), (2, 2), (2, 2)))
        out = np.zeros_like(img)
        for dy in range(5):
            for dx in range(5):
                out += pad[:, :, dy:dy + H, dx:dx + W] * kern[:, dy * 5 + dx, k][:, None, None, None]
        return out

    f, d = frame.numpy(), distrib.numpy()
    want_f = masks[:, 0:1] * f + masks[:, 1:2] * scratch + masks[:, 2:3] * first.numpy()
    want_d = masks[:, 0:1] * d + masks[:, 2:3] * dfirst.numpy()
    for k in range(K - 2):
        want_f += masks[:, k + 3:k + 4] * warp(f, k)
        want_d += masks[:, k + 3:k + 4] * warp(d, k)
    want_d /= want_d.sum(axis=(2, 3), keepdims=True)
    np.testing.assert_allclose(nf.numpy(), want_f, atol=1e-12)
    np.testing.assert_allclose(nd_.numpy(), want_d, atol=1e-12)


def test_savp_rollout_uses_first_context_frame():
    """The compositing's first-frame layer is context frame 0 of the last n_context frames, for every step."""
    from oracle.savp_predictor import OracleSavp
    from visual_foresight_amd.video_prediction.savp_arch import SavpConfig
    cfg = SavpConfig(height=32, width=32, adim=6, sequence_length=4)
    w = CdnaWeights.random(cfg, seed=3, bias_scale=0.1, ln_jitter=0.2)
    o = OracleSavp(w, torch.float64)
    rs = np.random.RandomState(2)
    frames = rs.randint(0, 256, (4, 1, 32, 32, 3)).astype(np.uint8)
    d = np.zeros((2, 1, 32, 32, 1), np.float32)
    d[0, 0, 3, 4, 0] = 1
    d[1, 0, 20, 9, 0] = 1
    args = (rs.normal(size=(3, 6)), d, rs.normal(size=(4, 5)), rs.normal(0, 0.1, (2, 2, 6)))
    a = o.rollout(frames, *args)
    b = o.rollout(frames[-2:], args[0][-1:], args[1], args[2][-2:], args[3])
    for x, y in zip(a, b):
        np.testing.assert_array_equal(x, y)
    seen = []
    orig = o.step

    def spy(*s_args):
        seen.append((s_args[5], s_args[6]))
        return orig(*s_args)
    o.step = spy
    o.rollout(frames, *args)
    want_f = torch.from_numpy(frames[-2, 0].astype(np.float32) / 255.).to(torch.float64).permute(2, 0, 1)
    for ff, dd in seen:
        assert torch.equal(ff[1], want_f) and float(dd[0, 0, 3, 4]) == 1.0
    np.testing.assert_allclose(a[1].sum(axis=(3, 4)), 1.0, atol=1e-12)


def test_public_decoder_table_agrees_with_the_oracles_and_the_librarys():
    """``CdnaConfig(decoder='public')``: the decoder widths of the public ``prediction_model.py`` (transposed convs keep their
    input's width: convt2 96 -> 96, convt3 64 -> 64; lstm7 K = 25 * 128; heads read 64 channels).  The product's table, the
    oracle's own statement of it and the library's (``vf_config.layer_spec = 1`` with ``arch = 0``) agree; 1.76 GMAC per
    sample-step against 1.63 for the SURVEY a14 table."""
    import ctypes
    from oracle.cdna_predictor import expected_shapes
    from visual_foresight_amd import _lib
    from visual_foresight_amd.video_prediction.cdna_arch import CdnaConfig, CdnaWeights
    cfg = CdnaConfig(height=64, width=64, decoder='public')
    shp = cfg.tensor_shapes()
    assert {k: tuple(v) for k, v in shp.items()} == expected_shapes(cfg)
    assert shp['convt2/w'] == (3, 3, 96, 96) and shp['lstm7/w'] == (5, 5, 128, 128) and shp['convt3/w'] == (3, 3, 64, 64)
    assert shp['rgb/w'] == (1, 1, 64, 3) and shp['masks/w'] == (1, 1, 64, 11) and shp['ln9/g'] == (64,)
    lib = _lib.load_library()
    c = _lib.VfConfig(64, 64, 4, 5, 1, 2, 15, 10, 4, 0, 0, 1, 1, 0, 0, cfg.layer_spec)
    assert cfg.layer_spec == 1 and lib.vf_weight_count(ctypes.byref(c)) == CdnaWeights.random(cfg, seed=0).n_floats()
    macs = sum(cfg.macs_per_sample_step().values())
    assert lib.vf_macs_per_sample_step(ctypes.byref(c)) == macs and 1.76e9 < macs < 1.77e9
    bad = _lib.VfConfig(64, 64, 4, 5, 1, 2, 15, 10, 4, 0, 1, 1, 1, 0, 0, 1)        # exact fp32 only
    assert lib.vf_weight_count(ctypes.byref(bad)) == 0 and b'public decoder' in lib.vf_last_error()
    bad = _lib.VfConfig(128, 128, 12, 5, 1, 2, 15, 10, 4, 0, 0, 1, 1, 1, 0, 1)     # arch 1 has no second table
    assert lib.vf_weight_count(ctypes.byref(bad)) == 0 and b'layer_spec' in lib.vf_last_error()
    with pytest.raises(ValueError):
        CdnaConfig(decoder='other')
