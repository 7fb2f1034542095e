"""Architecture table, seeded initialisation and weight-file format of the CDNA predictor.

The reference holds no network code: the action-conditioned CDNA conv-LSTM video predictor
lives in un-vendored third-party packages.  This module is
therefore the *normative* description of the network this repo implements - written from
Finn, Goodfellow & Levine 2016 (arXiv:1605.07157, section 3 + the public ``prediction_model``
layer list) with designated-pixel distribution propagation per Finn & Levine 2017 /
Ebert et al. 2018 (arXiv:1812.00568) - following the layer table of SURVEY.md row a14.

Per time step, for every candidate sample (NHWC, float32, "SAME" padding as TensorFlow
defines it, ``H x W`` input with H, W multiples of 8)::

    enc0  = relu(LN1(conv5x5/2(frame, 3->32)))                         H/2
    h1    = LN2(lstm1(enc0, 32));  h2 = LN3(lstm2(h1, 32))
    enc1  = relu(conv3x3/2(h2, 32->32))                                H/4
    h3    = LN4(lstm3(enc1, 64));  h4 = LN5(lstm4(h3, 64))
    enc2  = relu(conv3x3/2(h4, 64->64))                                H/8
    enc3  = relu(conv1x1(concat[enc2, tile(action, state)], ->64))
    h5    = LN6(lstm5(enc3, 128))
    enc4  = relu(convT3x3*2(h5, 128->128))                             H/4
    h6    = LN7(lstm6(enc4, 64))
    enc5  = relu(convT3x3*2(concat[h6, enc1], 96->64))                 H/2
    h7    = LN8(lstm7(enc5, 32))
    enc6  = relu(LN9(convT3x3*2(concat[h7, enc0], 64->32)))            H
    scratch = sigmoid(conv1x1(enc6, ->3))
    masks   = softmax_c(conv1x1(enc6, ->K+1))                          K = num_masks = 10
    kern    = normalise(relu(FC(flatten(h5), ->5*5*K) - 1e-12) + 1e-12)     per-sample 5x5 kernels
    warp_k  = depthwise 5x5 correlation of the previous frame with kern[..., k]
    frame'  = masks_0 * frame + masks_1 * scratch + sum_{k=0..K-2} masks_{k+2} * warp_k
    distr'  = normalise_hw(masks_0 * distr + sum_{k=0..K-2} masks_{k+2} * warp_k(distr))
    state'  = FC(concat[action, state], ->sdim)

(layer list ``[scratch, warp_0..warp_{K-1}]`` zipped against ``masks[1:]`` as in the public
CDNA implementation, which leaves the last kernel unused; the scratch image carries no
designated-pixel mass.)  conv-LSTM cell: ``gates = conv5x5(concat[x, h], ->4C)``, split
``i, j, f, o``; ``c' = c * sigmoid(f + 1) + sigmoid(i) * tanh(j)``; ``h' = tanh(c') * sigmoid(o)``.
LayerNorm normalises over (H, W, C) of one sample with per-channel gain/offset, eps 1e-12.
The first ``n_context`` steps are fed the ground-truth context (frames, distributions,
states); later steps feed back the cell's own predictions.

Canonical tensor layouts (what ``weights.bin`` stores, all float32, C-order):
conv / transposed conv ``[kh, kw, cin, cout]``; FC ``[in, out]``; vectors ``[n]``.
For a transposed conv, output pixel ``(2*iy + ky, 2*ix + kx)`` accumulates
``in[iy, ix, ci] * w[ky, kx, ci, co]`` (outputs beyond ``2*H_in`` are cropped).
"""
import json
import os
from collections import OrderedDict

import numpy as np

RELU_SHIFT = 1e-12
LN_EPS = 1e-12
DNA_KERN = 5
LSTM_SIZES = (32, 32, 64, 64, 128, 64, 32)


class CdnaConfig(object):
    """Static shape of one predictor instance."""

    def __init__(self, height=64, width=64, adim=4, sdim=5, ndesig=1, n_context=2,
                 sequence_length=15, num_masks=10, ncam=1, decoder='survey'):
        if decoder not in ('survey', 'public'):
            raise ValueError("decoder must be 'survey' or 'public', got %r" % (decoder,))
        self.decoder = decoder
        if height % 8 or width % 8:
            raise ValueError('image size must be a multiple of 8, got %dx%d' % (height, width))
        if ncam != 1:
            raise NotImplementedError('multi-view predictors are instantiated one per view')
        self.height, self.width = int(height), int(width)
        self.adim, self.sdim, self.ndesig = int(adim), int(sdim), int(ndesig)
        self.n_context, self.sequence_length = int(n_context), int(sequence_length)
        self.num_masks = int(num_masks)
        self.ncam = 1

    @property
    def horizon(self):
        """Number of predicted frames T = sequence_length - n_context."""
        return self.sequence_length - self.n_context

    arch = 'cdna'           # architecture tag (manifest, vf_config.arch): 'cdna' here, 'savp' in savp_arch.py
    arch_id = 0

    @property
    def layer_spec(self):
        """``vf_config.layer_spec`` of this table: 1 = the public decoder widths (arch 'cdna' only)."""
        return 1 if self.decoder == 'public' else 0

    def as_dict(self):
        d = dict(height=self.height, width=self.width, adim=self.adim, sdim=self.sdim,
                 ndesig=self.ndesig, n_context=self.n_context,
                 sequence_length=self.sequence_length, num_masks=self.num_masks)
        if self.decoder != 'survey':
            d['decoder'] = self.decoder
        return d

    def tensor_shapes(self):
        """Ordered name -> shape table of every learned tensor of this architecture."""
        return tensor_shapes(self)

    def macs_per_sample_step(self):
        return macs_per_sample_step(self)


def tensor_shapes(cfg):
    """Ordered name -> shape table of every learned tensor."""
    L = LSTM_SIZES
    a = cfg.adim + cfg.sdim
    K = cfg.num_masks
    fc_in = (cfg.height // 8) * (cfg.width // 8) * L[4]
    t = OrderedDict()

    def conv(name, kh, kw, cin, cout):
        t[name + '/w'] = (kh, kw, cin, cout)
        t[name + '/b'] = (cout,)

    def ln(name, c):
        t[name + '/g'] = (c,)
        t[name + '/b'] = (c,)

    conv('enc0', 5, 5, 3, 32);              ln('ln1', 32)
    conv('lstm1', 5, 5, 32 + L[0], 4 * L[0]); ln('ln2', L[0])
    conv('lstm2', 5, 5, L[0] + L[1], 4 * L[1]); ln('ln3', L[1])
    conv('enc1', 3, 3, L[1], L[1])
    conv('lstm3', 5, 5, L[1] + L[2], 4 * L[2]); ln('ln4', L[2])
    conv('lstm4', 5, 5, L[2] + L[3], 4 * L[3]); ln('ln5', L[3])
    conv('enc2', 3, 3, L[3], L[3])
    conv('enc3', 1, 1, L[3] + a, L[3])
    conv('lstm5', 5, 5, L[3] + L[4], 4 * L[4]); ln('ln6', L[4])
    conv('convt1', 3, 3, L[4], L[4])
    conv('lstm6', 5, 5, L[4] + L[5], 4 * L[5]); ln('ln7', L[5])
    # decoder widths: 'survey' = SURVEY.md row a14 (convt2 96 -> 64, convt3 64 -> 32); 'public' = the public
    # ``prediction_model.py`` of arXiv:1605.07157, whose transposed convs keep the width of their (concatenated) input
    # (``conv2d_transpose(hidden6, hidden6.get_shape()[3], ...)``): convt2 96 -> 96, convt3 64 -> 64, so lstm7 convolves
    # 96 + 32 channels and the 1 x 1 heads read 64.  A checkpoint has one or the other.
    pub = getattr(cfg, 'decoder', 'survey') == 'public'
    c_t2, c_top = (L[5] + L[1], L[6] + 32) if pub else (L[5], 32)
    conv('convt2', 3, 3, L[5] + L[1], c_t2)
    conv('lstm7', 5, 5, c_t2 + L[6], 4 * L[6]); ln('ln8', L[6])
    conv('convt3', 3, 3, L[6] + 32, c_top);  ln('ln9', c_top)
    conv('rgb', 1, 1, c_top, 3)
    conv('masks', 1, 1, c_top, K + 1)
    t['cdna/w'] = (fc_in, DNA_KERN * DNA_KERN * K)
    t['cdna/b'] = (DNA_KERN * DNA_KERN * K,)
    t['state/w'] = (a, cfg.sdim)
    t['state/b'] = (cfg.sdim,)
    return t


def macs_per_sample_step(cfg):
    """Algorithmic multiply-accumulates of one cell evaluation for one sample.

    ``MAC = sum_layers H_out * W_out * k^2 * C_in * C_out`` (transposed conv:
    ``H_in * W_in * k^2 * C_in * C_out``), convs + FCs + the CDNA warps only
    (SURVEY.md 8d).  64x64, ndesig=1: 1.63e9.
    """
    H, W = cfg.height, cfg.width
    shp = tensor_shapes(cfg)
    res = {'enc0': (H // 2, W // 2), 'lstm1': (H // 2, W // 2), 'lstm2': (H // 2, W // 2),
           'enc1': (H // 4, W // 4), 'lstm3': (H // 4, W // 4), 'lstm4': (H // 4, W // 4),
           'enc2': (H // 8, W // 8), 'enc3': (H // 8, W // 8), 'lstm5': (H // 8, W // 8),
           'convt1': (H // 8, W // 8), 'lstm6': (H // 4, W // 4), 'convt2': (H // 4, W // 4),
           'lstm7': (H // 2, W // 2), 'convt3': (H // 2, W // 2), 'rgb': (H, W), 'masks': (H, W)}
    out = OrderedDict()
    for name, (h, w) in res.items():
        kh, kw, cin, cout = shp[name + '/w']
        out[name] = h * w * kh * kw * cin * cout
    out['cdna_fc'] = shp['cdna/w'][0] * shp['cdna/w'][1]
    out['warp_frame'] = H * W * DNA_KERN * DNA_KERN * 3 * cfg.num_masks
    out['warp_distrib'] = H * W * DNA_KERN * DNA_KERN * cfg.ndesig * cfg.num_masks
    out['state_fc'] = shp['state/w'][0] * shp['state/w'][1]
    return out


class CdnaWeights(object):
    """Named float32 tensors in canonical layout + (de)serialisation."""

    def __init__(self, cfg, tensors):
        self.cfg = cfg
        want = cfg.tensor_shapes()
        if list(tensors.keys()) != list(want.keys()):
            raise ValueError('tensor set does not match the architecture table')
        for name, shape in want.items():
            if tuple(tensors[name].shape) != tuple(shape):
                raise ValueError('%s: shape %s, expected %s' % (name, tensors[name].shape, shape))
        self.tensors = OrderedDict((k, np.ascontiguousarray(v, dtype=np.float32))
                                   for k, v in tensors.items())

    # ------------------------------------------------------------------ initialisation
    @classmethod
    def random(cls, cfg, seed=0, bias_scale=0.0, ln_jitter=0.0):
        """Seeded Glorot-uniform weights; zero bias and unit LayerNorm unless jitter is asked for.

        Uses the legacy ``RandomState`` stream, which is stable across NumPy versions, so the
        same seed gives the same network on the build container and on the GPU box.
        """
        rs = np.random.RandomState(seed)
        tensors = OrderedDict()
        for name, shape in cfg.tensor_shapes().items():
            kind = name.split('/')[1]
            if kind == 'w':
                if len(shape) == 4:
                    fan_in = shape[0] * shape[1] * shape[2]
                    fan_out = shape[0] * shape[1] * shape[3]
                else:
                    fan_in, fan_out = shape
                lim = np.sqrt(6.0 / (fan_in + fan_out))
                tensors[name] = rs.uniform(-lim, lim, shape).astype(np.float32)
            elif kind == 'g':
                tensors[name] = (1.0 + ln_jitter * rs.uniform(-1, 1, shape)).astype(np.float32)
            else:   # conv / FC bias or LayerNorm offset
                scale = ln_jitter if name.startswith('ln') else bias_scale     # ('lna'/'lnb' included)
                tensors[name] = (scale * rs.uniform(-1, 1, shape)).astype(np.float32)
        return cls(cfg, tensors)

    # ------------------------------------------------------------------ file format
    def save(self, model_dir):
        """``model_dir/manifest.json`` + ``model_dir/weights.bin`` (flat little-endian float32)."""
        os.makedirs(model_dir, exist_ok=True)
        manifest = {'format': 'vf-cdna-v1', 'arch': self.cfg.arch, 'config': self.cfg.as_dict(), 'tensors': []}
        offset = 0
        with open(os.path.join(model_dir, 'weights.bin'), 'wb') as f:
            for name, arr in self.tensors.items():
                manifest['tensors'].append({'name': name, 'shape': list(arr.shape), 'offset': offset})
                f.write(arr.astype('<f4').tobytes())
                offset += arr.size
        manifest['n_floats'] = offset
        with open(os.path.join(model_dir, 'manifest.json'), 'w') as f:
            json.dump(manifest, f, indent=1)

    @classmethod
    def load(cls, model_dir, cfg=None):
        with open(os.path.join(model_dir, 'manifest.json')) as f:
            manifest = json.load(f)
        if manifest.get('format') != 'vf-cdna-v1':
            raise ValueError('unknown weight file format %r' % manifest.get('format'))
        arch = manifest.get('arch', 'cdna')
        if arch == 'cdna':
            file_cfg = CdnaConfig(**manifest['config'])
        elif arch in ('savp', 'savp2'):
            from visual_foresight_amd.video_prediction.savp_arch import SavpConfig, Savp2Config
            file_cfg = (SavpConfig if arch == 'savp' else Savp2Config)(**manifest['config'])
        elif arch == 'savp3':
            from visual_foresight_amd.video_prediction.savp3_arch import Savp3Config
            file_cfg = Savp3Config(**manifest['config'])
        else:
            raise ValueError('unknown architecture %r in %s' % (arch, model_dir))
        if cfg is not None and cfg.arch != arch:
            raise ValueError('checkpoint architecture %r does not match requested %r' % (arch, cfg.arch))
        if cfg is not None:
            mine, theirs = cfg.as_dict(), file_cfg.as_dict()
            for k in ('height', 'width', 'adim', 'sdim', 'num_masks') + (('zdim', 'layer_spec') if arch == 'savp3' else ()) + \
                    (('decoder',) if 'decoder' in mine or 'decoder' in theirs else ()):
                mine.setdefault('decoder', 'survey'); theirs.setdefault('decoder', 'survey')
                if mine[k] != theirs[k]:
                    raise ValueError('checkpoint %s=%r does not match requested %r' % (k, theirs[k], mine[k]))
            file_cfg = cfg      # ndesig / sequence_length are run-time choices, not weights
        blob = np.fromfile(os.path.join(model_dir, 'weights.bin'), dtype='<f4')
        if blob.size != manifest['n_floats']:
            raise ValueError('weights.bin holds %d floats, manifest says %d' % (blob.size, manifest['n_floats']))
        tensors = OrderedDict()
        for ent in manifest['tensors']:
            n = int(np.prod(ent['shape']))
            tensors[ent['name']] = blob[ent['offset']:ent['offset'] + n].reshape(ent['shape'])
        return cls(file_cfg, tensors)

    def n_floats(self):
        return sum(v.size for v in self.tensors.values())
