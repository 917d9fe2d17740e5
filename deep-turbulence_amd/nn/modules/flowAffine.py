"""Conditional affine coupling layers on the HIP path.  API mirror of the reference's
nn/modules/flowAffine.py (AffineCouplingLayer :20-109, LSTMAffineCouplingLayer :111-236)."""
import torch
import torch.nn as nn

import tmg_hip as H
import tmg_ops as ops
from nn.modules.convLSTM import ResidLSTMBlock
from nn.modules.denseBlock import NoNormDenseBlock
from nn.modules.flowUtils import Conv2dZeros


def _channels(in_features, cond_features):
    if in_features % 2 == 0:
        return in_features // 2 + cond_features, in_features
    # odd channel counts (reference :43-46) never occur in TMGlow: squeezed widths are multiples of 4
    raise NotImplementedError("odd channel counts are not on the TM-Glow path (reference flowAffine.py:43-46)")


class AffineCouplingLayer(nn.Module):
    """h = ZeroConv(relu(Dense2(cat(x1, cond)))); x2 <- (x2 + h_even) * exp(2 softsign(h_odd))
    (forward, reference :73-83) or x2 / s - shift (reverse, :98-109); log-det has the same sign both ways.

    The channel concatenations are never materialised: the conv kernels read x1, cond and the two
    growth channels as separate segments, with the ReLU applied on the way into LDS."""

    def __init__(self, in_features, cond_features):
        super().__init__()
        in_channels, out_channels = _channels(in_features, cond_features)
        self.coupling_nn = nn.Sequential()
        self.coupling_nn.add_module('dense_block', NoNormDenseBlock(2, in_channels, growth_rate=1, drop_rate=0., bottleneck=False))
        self.coupling_nn.add_module('zero_conv', Conv2dZeros(in_channels + 2, out_channels))

    def run(self, xn, condn, reverse):
        db, zc = self.coupling_nn.dense_block, self.coupling_nn.zero_conv
        if zc.logscale_factor != 1:  # never the case in TMGlow; keep the un-fused composition for it
            ch = xn.shape[3] // 2
            t = db.run([xn[..., :ch], condn])
            t = t[:2] + [torch.cat(t[2:], 3)]
            return ops.AffineFn.apply(zc.run(t, relu_in=True), xn, reverse)
        return ops.CouplingTailFn.apply(xn, condn, db.denselayer1.conv1.weight, db.denselayer2.conv1.weight, zc.conv.weight,
                                        zc.conv.bias, zc.scale, reverse, 0)

    def forward(self, x, cond):
        y, ld = self.run(H.nhwc(x), H.nhwc(cond), False)
        return H.nchw(y), ld

    def reverse(self, y, cond):
        x, ld = self.run(H.nhwc(y), H.nhwc(cond), True)
        return H.nchw(x), ld


class LSTMAffineCouplingLayer(nn.Module):
    """As AffineCouplingLayer with a residual ConvLSTM block in front of the dense layers
    (reference :161-236).  Returns (y, logdet, (h_next, c_next))."""

    def __init__(self, in_features, cond_features, rec_features):
        super().__init__()
        in_channels, out_channels = _channels(in_features, cond_features)
        self.resid_lstm = ResidLSTMBlock(in_channels, rec_features, in_channels, kernel_size=(3, 3))
        self.dense_nn = nn.Sequential()
        self.dense_nn.add_module('dense_block', NoNormDenseBlock(2, in_channels, growth_rate=1, drop_rate=0., bottleneck=False))
        self.out_conv = nn.Sequential()
        self.out_conv.add_module('zero_conv', Conv2dZeros(in_channels + 2, out_channels))

    def run(self, xn, condn, state, reverse, pad=0):
        """pad > 0: xn is in the zero-padded layout [x1 | 0 x pad | x2 | 0 x pad] of LSTMFLowBlock (channel halves that are not a
        multiple of 4); the weights get matching zero rows / columns / output channels, so every conv of the block sees
        float4-addressable segments and the padding channels of the result stay exactly zero."""
        chp = xn.shape[3] // 2
        db, zc = self.dense_nn.dense_block, self.out_conv.zero_conv
        if xn.requires_grad and torch.is_grad_enabled():
            xn, x1 = ops.LeadingChannelsFn.apply(xn, chp)     # the slice's gradient joins the whole tensor's in place
        else:
            x1 = xn[..., :chp]
        if pad == 0:
            # (the tail below masks the gradient it returns for `out` by [out > 0]: the out-conv's own ReLU mask pass is redundant)
            out, h_next, c_next = self.resid_lstm.run([x1, condn], state, out_grad_premasked=True)
            y, ld = ops.CouplingTailFn.apply(xn, out, db.denselayer1.conv1.weight, db.denselayer2.conv1.weight, zc.conv.weight,
                                             zc.conv.bias, zc.scale, reverse, 1)
            return y, ld, (h_next, c_next)
        ch, dev = chp - pad, xn.device
        z = lambda *shape: torch.zeros(shape, device=dev, dtype=torch.float32)  # noqa: E731
        cell, oc = self.resid_lstm.convLSTM, self.resid_lstm.out_seq.LSTM_out_conv
        cin = oc.weight.shape[0]                       # ch + cond channels: width of the block's feature map
        src = [cell.conv.weight, oc.weight, oc.bias, db.denselayer1.conv1.weight, db.denselayer2.conv1.weight, zc.conv.weight, zc.conv.bias]

        def build():
            ins = lambda w, at: torch.cat([w[:, :at], z(w.shape[0], pad, 3, 3), w[:, at:]], 1)  # noqa: E731  zero input rows at `at`
            gw = ins(cell.conv.weight, ch)
            ow = torch.cat([ins(oc.weight, ch), z(pad, oc.weight.shape[1] + pad, 3, 3)], 0)      # feature map widened by `pad` zero channels
            ob = torch.cat([oc.bias, z(pad)])
            w1 = torch.cat([db.denselayer1.conv1.weight, z(1, pad, 3, 3)], 1)
            w2 = ins(db.denselayer2.conv1.weight, cin)                                             # before the d1 row
            wz = torch.cat([ins(zc.conv.weight, cin), z(2 * pad, zc.conv.weight.shape[1] + pad, 3, 3)], 0)
            bz = torch.cat([zc.conv.bias, z(2 * pad)])
            return [gw, ow, ob, w1, w2, wz, bz]

        # parameter-sized functions of the weights alone: one evaluation per BPTT window (ops.DerivedCache, gradient-sink proxies)
        cache = self.__dict__.get('_derived')
        if cache is None:
            cache = self.__dict__['_derived'] = ops.DerivedCache()
        gw, ow, ob, w1, w2, wz, bz = cache.get('pad', src, (int(ch), int(pad)), build, proxies=True)
        if state is None:
            h_cur, c_cur = z(*xn.shape[:3], cell.hidden_dim), None
        else:
            h_cur, c_cur = state
        h_next, c_next = ops.ConvLSTMCellFn.apply(gw, cell.conv.bias, h_cur, c_cur, x1, condn)
        out = ops.conv([x1, condn, h_next], ow, ob, relu_out=True, _grad_premasked=True)
        y, ld = ops.CouplingTailFn.apply(xn, out, w1, w2, wz, bz, zc.scale, reverse, 1)
        return y, ld, (h_next, c_next)

    def _call(self, x, cond, rec_states, reverse):
        st = None if rec_states is None else (H.nhwc(rec_states[0]), H.nhwc(rec_states[1]))
        y, ld, (h, c) = self.run(H.nhwc(x), H.nhwc(cond), st, reverse)
        return H.nchw(y), ld, (H.nchw(h), H.nchw(c))

    def forward(self, x, cond, rec_states=None):
        return self._call(x, cond, rec_states, False)

    def reverse(self, y, cond, rec_states=None):
        return self._call(y, cond, rec_states, True)
