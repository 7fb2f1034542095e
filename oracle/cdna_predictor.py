"""ORACLE (test infrastructure): CPU restatement of the CDNA conv-LSTM video predictor.

PARITY UNPINNED.  The network arithmetic is NOT part of ``/root/reference``: it lives in the
un-vendored, un-pinned third-party packages ``video_prediction`` (``febert/video_prediction-1``
branch ``dev``; call sites ``visual_mpc/video_prediction/vpred_model_interface.py:2,53-58,73-88``)
and ``robonet`` (``visual_mpc/policy/cem_controllers/pixel_cost_controller.py:11-12``), on
TensorFlow 1.6 (``requirements.txt:18``) - none installable here, and the reference holds no
golden vectors for it.  This file restates the published algorithm (Finn, Goodfellow & Levine
2016, arXiv:1605.07157; designated-pixel propagation per Ebert et al. 2018, arXiv:1812.00568)
exactly as ``visual_foresight_amd/video_prediction/cdna_arch.py`` specifies it, in plain
PyTorch CPU ops (float32, or float64 to measure rounding), and is what the HIP kernels are
checked against.  The boundary semantics it follows ARE in the reference:
  * last ``n_context`` frames, ``astype(float32)/255``      <- ``video_prediction/pred_util.py:4-7``
  * batch-1 context tiled over the sample batch             <- ``video_prediction/setup_predictor.py:40-44``
  * actions ``[B, seq_len-1, adim]`` = context actions + plan <- ``setup_predictor.py:105-106``
  * outputs ``[B, T, ncam, H, W, C]`` with a camera axis     <- ``vpred_model_interface.py:78,88``

Only tests/, ``__graft_entry__.smoke()`` and bench.py's cpu_baseline leg may import this.
"""
import numpy as np
import torch
import torch.nn.functional as F

# The oracle states its constants itself (a typo in the product's table must show up as a test failure,
# not be inherited); tests/test_oracle_predictor.py checks they agree with cdna_arch.py.
LSTM_SIZES = (32, 32, 64, 64, 128, 64, 32)     # hidden channels of lstm1..lstm7 (arXiv:1605.07157 fig. 3)
RELU_SHIFT = 1e-12                              # CDNA kernels: relu(x - shift) + shift before normalising
LN_EPS = 1e-12                                  # layer-norm variance epsilon
DNA_KERN = 5                                    # CDNA kernel size


def _same_pad(x, k, stride):
    """TensorFlow 'SAME' padding for an NCHW tensor (extra pixel goes bottom/right)."""
    H, W = x.shape[-2:]
    ph = max((-(-H // stride) - 1) * stride + k - H, 0)
    pw = max((-(-W // stride) - 1) * stride + k - W, 0)
    return F.pad(x, (pw // 2, pw - pw // 2, ph // 2, ph - ph // 2))


def expected_shapes(cfg):
    """The oracle's OWN reading of the layer table (arXiv:1605.07157 section 3 / SURVEY row a14): name -> shape.
    A network handed to the oracle must have exactly these tensors, so a wrong layer size in the product's
    table (cdna_arch.py) cannot hide behind weights generated from that same table."""
    a, K = cfg.adim + cfg.sdim, cfg.num_masks
    h8, w8 = cfg.height // 8, cfg.width // 8
    if getattr(cfg, 'decoder', 'survey') == 'public':
        return _expected_shapes_public(a, K, h8, w8, cfg.sdim)
    t = {'enc0/w': (5, 5, 3, 32), 'lstm1/w': (5, 5, 64, 128), 'lstm2/w': (5, 5, 64, 128), 'enc1/w': (3, 3, 32, 32),
         'lstm3/w': (5, 5, 96, 256), 'lstm4/w': (5, 5, 128, 256), 'enc2/w': (3, 3, 64, 64), 'enc3/w': (1, 1, 64 + a, 64),
         'lstm5/w': (5, 5, 192, 512), 'convt1/w': (3, 3, 128, 128), 'lstm6/w': (5, 5, 192, 256),
         'convt2/w': (3, 3, 96, 64), 'lstm7/w': (5, 5, 96, 128), 'convt3/w': (3, 3, 64, 32), 'rgb/w': (1, 1, 32, 3),
         'masks/w': (1, 1, 32, K + 1), 'cdna/w': (h8 * w8 * 128, 25 * K), 'state/w': (a, cfg.sdim)}
    for name in list(t):
        t[name[:-2] + '/b'] = (t[name][-1],)
    for i, c in enumerate((32, 32, 32, 64, 64, 128, 64, 32, 32)):
        t['ln%d/g' % (i + 1)] = t['ln%d/b' % (i + 1)] = (c,)
    return t


def _expected_shapes_public(a, K, h8, w8, sdim):
    """The decoder as the PUBLIC ``prediction_model.py`` of arXiv:1605.07157 builds it: every ``conv2d_transpose`` keeps the
    channel count of its input - ``enc4 = convT(hidden5, 128)``, ``enc5 = convT(concat[hidden6, enc1] = 96, 96)``, ``enc6 =
    convT(concat[hidden7, enc0] = 64, 64)`` - so ``lstm7`` convolves 96 + 32 channels and the two 1 x 1 heads read 64."""
    t = {'enc0/w': (5, 5, 3, 32), 'lstm1/w': (5, 5, 64, 128), 'lstm2/w': (5, 5, 64, 128), 'enc1/w': (3, 3, 32, 32),
         'lstm3/w': (5, 5, 96, 256), 'lstm4/w': (5, 5, 128, 256), 'enc2/w': (3, 3, 64, 64), 'enc3/w': (1, 1, 64 + a, 64),
         'lstm5/w': (5, 5, 192, 512), 'convt1/w': (3, 3, 128, 128), 'lstm6/w': (5, 5, 192, 256),
         'convt2/w': (3, 3, 96, 96), 'lstm7/w': (5, 5, 128, 128), 'convt3/w': (3, 3, 64, 64), 'rgb/w': (1, 1, 64, 3),
         'masks/w': (1, 1, 64, K + 1), 'cdna/w': (h8 * w8 * 128, 25 * K), 'state/w': (a, sdim)}
    for name in list(t):
        t[name[:-2] + '/b'] = (t[name][-1],)
    for i, c in enumerate((32, 32, 32, 64, 64, 128, 64, 32, 64)):
        t['ln%d/g' % (i + 1)] = t['ln%d/b' % (i + 1)] = (c,)
    return t


class OracleCdna(object):
    expected_shapes = staticmethod(expected_shapes)

    def __init__(self, weights, dtype=torch.float32, threads=None):
        self.cfg = weights.cfg
        self.dtype = dtype
        want = self.expected_shapes(self.cfg)
        got = {k: tuple(v.shape) for k, v in weights.tensors.items()}
        if got != want:
            diff = sorted(k for k in set(got) | set(want) if got.get(k) != want.get(k))
            raise ValueError('network does not match the oracle\'s layer table: %s' % diff[:6])
        if threads:
            torch.set_num_threads(threads)
        self.p = {k: torch.from_numpy(np.array(v)).to(dtype) for k, v in weights.tensors.items()}

    # ------------------------------------------------------------------ layers
    def _conv(self, x, name, stride=1):
        w = self.p[name + '/w'].permute(3, 2, 0, 1).contiguous()        # HWIO -> OIHW
        return F.conv2d(_same_pad(x, w.shape[-1], stride), w, self.p[name + '/b'], stride=stride)

    def _convt(self, x, name):
        w = self.p[name + '/w'].permute(2, 3, 0, 1).contiguous()        # [kh,kw,ci,co] -> [ci,co,kh,kw]
        y = F.conv_transpose2d(x, w, self.p[name + '/b'], stride=2)     # out[2i+k]; size 2N+1
        return y[:, :, :2 * x.shape[2], :2 * x.shape[3]]

    def _ln(self, x, name):
        mean = x.mean(dim=(1, 2, 3), keepdim=True)
        var = ((x - mean) ** 2).mean(dim=(1, 2, 3), keepdim=True)
        y = (x - mean) / torch.sqrt(var + LN_EPS)
        return y * self.p[name + '/g'].view(1, -1, 1, 1) + self.p[name + '/b'].view(1, -1, 1, 1)

    def _lstm(self, x, state, name, C):
        c, h = state
        gates = self._conv(torch.cat([x, h], dim=1), name)
        i, j, f, o = torch.split(gates, C, dim=1)
        c_new = c * torch.sigmoid(f + 1.0) + torch.sigmoid(i) * torch.tanh(j)
        h_new = torch.tanh(c_new) * torch.sigmoid(o)
        return h_new, (c_new, h_new)

    # ------------------------------------------------------------------ one cell evaluation
    def step(self, frame, distrib, state_vec, action, lstm_states):
        """frame [B,3,H,W], distrib [B,nd,H,W], state_vec [B,sdim], action [B,adim]."""
        cfg, L = self.cfg, LSTM_SIZES
        B = frame.shape[0]
        K = cfg.num_masks
        new_states = [None] * 7

        enc0 = F.relu(self._ln(self._conv(frame, 'enc0', 2), 'ln1'))
        h1, new_states[0] = self._lstm(enc0, lstm_states[0], 'lstm1', L[0]); h1 = self._ln(h1, 'ln2')
        h2, new_states[1] = self._lstm(h1, lstm_states[1], 'lstm2', L[1]);   h2 = self._ln(h2, 'ln3')
        enc1 = F.relu(self._conv(h2, 'enc1', 2))
        h3, new_states[2] = self._lstm(enc1, lstm_states[2], 'lstm3', L[2]); h3 = self._ln(h3, 'ln4')
        h4, new_states[3] = self._lstm(h3, lstm_states[3], 'lstm4', L[3]);   h4 = self._ln(h4, 'ln5')
        enc2 = F.relu(self._conv(h4, 'enc2', 2))

        sa = torch.cat([action, state_vec], dim=1)
        smear = sa.view(B, -1, 1, 1).expand(B, sa.shape[1], enc2.shape[2], enc2.shape[3])
        enc3 = F.relu(self._conv(torch.cat([enc2, smear], dim=1), 'enc3'))
        h5, new_states[4] = self._lstm(enc3, lstm_states[4], 'lstm5', L[4]); h5 = self._ln(h5, 'ln6')
        enc4 = F.relu(self._convt(h5, 'convt1'))
        h6, new_states[5] = self._lstm(enc4, lstm_states[5], 'lstm6', L[5]); h6 = self._ln(h6, 'ln7')
        enc5 = F.relu(self._convt(torch.cat([h6, enc1], dim=1), 'convt2'))
        h7, new_states[6] = self._lstm(enc5, lstm_states[6], 'lstm7', L[6]); h7 = self._ln(h7, 'ln8')
        enc6 = F.relu(self._ln(self._convt(torch.cat([h7, enc0], dim=1), 'convt3'), 'ln9'))

        scratch = torch.sigmoid(self._conv(enc6, 'rgb'))
        masks = torch.softmax(self._conv(enc6, 'masks'), dim=1)              # [B, K+1, H, W]

        # per-sample CDNA kernels from the NHWC-flattened bottleneck
        flat = h5.permute(0, 2, 3, 1).reshape(B, -1)
        kern = flat @ self.p['cdna/w'] + self.p['cdna/b']
        kern = F.relu(kern - RELU_SHIFT) + RELU_SHIFT
        kern = kern.view(B, DNA_KERN * DNA_KERN, K)
        kern = kern / kern.sum(dim=1, keepdim=True)
        kern = kern.permute(0, 2, 1).reshape(B, K, DNA_KERN, DNA_KERN)       # [B, K, 5, 5]

        def warp(img):      # img [B, C, H, W] -> [B, K, C, H, W]; correlation, zero padded
            Bc, C, H, W = img.shape
            x = _same_pad(img, DNA_KERN, 1).reshape(1, Bc * C, H + 4, W + 4)
            w = kern.repeat_interleave(C, dim=0).reshape(Bc * C * K, 1, DNA_KERN, DNA_KERN)
            y = F.conv2d(x, w, groups=Bc * C)                                 # [1, B*C*K, H, W]
            return y.view(Bc, C, K, H, W).permute(0, 2, 1, 3, 4)

        wf = warp(frame)
        next_frame = masks[:, 0:1] * frame + masks[:, 1:2] * scratch
        for k in range(K - 1):
            next_frame = next_frame + masks[:, k + 2:k + 3] * wf[:, k]

        wd = warp(distrib)
        next_distrib = masks[:, 0:1] * distrib
        for k in range(K - 1):
            next_distrib = next_distrib + masks[:, k + 2:k + 3] * wd[:, k]
        next_distrib = next_distrib / next_distrib.sum(dim=(2, 3), keepdim=True)

        next_state = sa @ self.p['state/w'] + self.p['state/b']
        return next_frame, next_distrib, next_state, new_states

    # ------------------------------------------------------------------ whole rollout
    def rollout(self, ctx_frames_u8, ctx_actions, ctx_distrib, ctx_states, actions):
        """Predict T frames for every action sequence.

        ctx_frames_u8 [>=n_context, 1, H, W, 3] uint8 (history; the last n_context are used),
        ctx_actions   [>=n_context-1, adim], ctx_distrib [n_context, 1, H, W, nd] float32,
        ctx_states    [>=n_context, sdim], actions [M, T, adim]
        -> frames [M, T, 1, H, W, 3], distrib [M, T, 1, H, W, nd], states [M, T, sdim] (numpy)
        """
        cfg, dt = self.cfg, self.dtype
        nc = cfg.n_context
        M, T = actions.shape[:2]
        H, W = cfg.height, cfg.width
        frames = np.asarray(ctx_frames_u8)[-nc:, 0].astype(np.float32) / 255.
        frames = torch.from_numpy(frames).to(dt).permute(0, 3, 1, 2)                    # [nc,3,H,W]
        distr = torch.from_numpy(np.asarray(ctx_distrib, dtype=np.float32)[-nc:, 0]).to(dt).permute(0, 3, 1, 2)
        states = torch.from_numpy(np.asarray(ctx_states, dtype=np.float64)[-nc:]).to(dt)
        acts = torch.from_numpy(np.asarray(actions, dtype=np.float64)).to(dt)
        if nc > 1:
            ca = torch.from_numpy(np.asarray(ctx_actions, dtype=np.float64)[-(nc - 1):]).to(dt)
            acts = torch.cat([ca[None].expand(M, nc - 1, cfg.adim), acts], dim=1)     # [M, T+nc-1, adim]

        sizes = [(H // 2, W // 2)] * 2 + [(H // 4, W // 4)] * 2 + [(H // 8, W // 8)] + \
                [(H // 4, W // 4)] + [(H // 2, W // 2)]
        lstm = [(torch.zeros(M, C, h, w, dtype=dt), torch.zeros(M, C, h, w, dtype=dt))
                for C, (h, w) in zip(LSTM_SIZES, sizes)]

        out_f, out_d, out_s = [], [], []
        gen_f = gen_d = gen_s = None
        for s in range(T + nc - 1):
            if s < nc:
                f_in = frames[s][None].expand(M, 3, H, W)
                d_in = distr[s][None].expand(M, cfg.ndesig, H, W)
                s_in = states[s][None].expand(M, cfg.sdim)
            else:
                f_in, d_in, s_in = gen_f, gen_d, gen_s
            gen_f, gen_d, gen_s, lstm = self.step(f_in, d_in, s_in, acts[:, s], lstm)
            if s >= nc - 1:
                out_f.append(gen_f); out_d.append(gen_d); out_s.append(gen_s)
        frames_out = torch.stack(out_f, 1).permute(0, 1, 3, 4, 2)[:, :, None]     # [M,T,1,H,W,3]
        distr_out = torch.stack(out_d, 1).permute(0, 1, 3, 4, 2)[:, :, None]
        return (frames_out.contiguous().numpy(), distr_out.contiguous().numpy(),
                torch.stack(out_s, 1).numpy())
