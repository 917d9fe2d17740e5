"""Flow blocks on the HIP path.  API mirror of the reference's nn/modules/flowLSTMBlock.py
(AffineCouplingBlock :24-86, UnNormedAffineCouplingBlock :88-146, LSTMCouplingBlock :148-218,
LSTMFLowBlock :220-361)."""
import os

import torch
import torch.nn as nn

import tmg_hip as H
import tmg_ops as ops
from nn.modules.actNorm import ActNorm
from nn.modules.flowAffine import AffineCouplingLayer, LSTMAffineCouplingLayer
from nn.modules.flowUtils import CheckerSqueeze, Split, Squeeze
from nn.modules.glowConv import InvertibleConv1x1, InvertibleConv1x1LU


def _make_conv(in_features, train_sampling, LUdecompose):
    cls = InvertibleConv1x1LU if LUdecompose else InvertibleConv1x1
    return cls(in_features, train_sampling=train_sampling)


class _BlockBase(nn.Module):
    """Shared mechanics of the three block kinds: the ActNorm (if any) and the invertible 1x1 conv are
    ONE channel-mix launch.  Forward order is norm -> conv:  W (a*x + b) = (W diag a) x + W b;
    reverse order is conv -> norm^-1:  (W y - b) / a = (diag(1/a) W) y - b/a.  The O(C^2) folding of
    the parameters is ordinary torch autograd, the [B,H,W,C] contraction is the MFMA 1x1 kernel."""

    def _mix_params(self, reverse):
        norm = getattr(self, 'norm', None)
        if isinstance(self.conv, InvertibleConv1x1LU):
            W = self.conv.matrix(reverse)
        else:
            W = self.conv._matrix((not reverse) if self.conv.train_sampling else reverse)
        if norm is None:
            return W, None
        s, t = norm.folded(reverse)
        if reverse:
            return s.unsqueeze(1) * W, t
        return W * s.unsqueeze(0), W @ t

    def _mix_logdet(self, xn):
        hw = xn.shape[1] * xn.shape[2]
        if isinstance(self.conv, InvertibleConv1x1LU):
            ld = self.conv.log_s.sum() * hw
            ld = -ld if self.conv.train_sampling else ld
        else:
            raise NotImplementedError("TMGlow always builds LU blocks (tmGlow.py:366); plain blocks run via the stand-alone modules")
        norm = getattr(self, 'norm', None)
        if norm is not None:
            ld = ld + norm.weight.abs().log().sum() * hw
        return ld

    def _mix(self, xn, reverse, mix=None):
        W, b = self._mix_params(reverse) if mix is None else mix
        c = W.shape[0]
        if ops.mix_precision() != "f32":
            return ops.MixFn.apply(xn, W, b)
        return ops.conv([xn], W.reshape(c, c, 1, 1), b, ksize=1)


class AffineCouplingBlock(_BlockBase):
    def __init__(self, in_features, cond_features, train_sampling=True, LUdecompose=False):
        super().__init__()
        self.norm = ActNorm(in_features)
        self.conv = _make_conv(in_features, train_sampling, LUdecompose)
        self.coupling = AffineCouplingLayer(in_features, cond_features)

    def run(self, xn, condn, reverse, mix=None):
        """mix: optional pre-folded (W, b) of this block's ActNorm + 1x1 conv (LSTMFLowBlock builds them for all its
        layers in one batched pass and also accounts for their log-dets); None -> fold here and add the log-det."""
        if not isinstance(self.conv, InvertibleConv1x1LU):
            return _run_unfused(self, xn, condn, None, reverse)[:2]
        extra = self._mix_logdet(xn) if mix is None else None      # (None: the caller accounts for the mix log-dets of the level)
        if reverse:
            t, ld = self.coupling.run(xn, condn, True)
            return self._mix(t, True, mix), (ld if extra is None else ld + extra)
        y, ld = self.coupling.run(self._mix(xn, False, mix), condn, False)
        return y, (ld if extra is None else ld + extra)

    def forward(self, x, cond):
        y, ld = self.run(H.nhwc(x), H.nhwc(cond), False)
        return H.nchw(y), ld

    def reverse(self, y, cond):
        x, ld = self.run(H.nhwc(y), H.nhwc(cond), True)
        return H.nchw(x), ld


class UnNormedAffineCouplingBlock(AffineCouplingBlock):
    """First block of every level: no ActNorm (reference :101-146)."""

    def __init__(self, in_features, cond_features, train_sampling=True, LUdecompose=False):
        nn.Module.__init__(self)
        self.conv = _make_conv(in_features, train_sampling, LUdecompose)
        self.coupling = AffineCouplingLayer(in_features, cond_features)


class LSTMCouplingBlock(_BlockBase):
    """Last block of every level (reference :148-218).  `norm2` is dead in the reference (:170, never
    called); it is kept so the state_dict matches, and it never receives a gradient."""

    def __init__(self, in_features, cond_features, rec_features, train_sampling=True, LUdecompose=False):
        super().__init__()
        self.norm = ActNorm(in_features)
        self.norm2 = ActNorm(in_features)
        self.conv = _make_conv(in_features, train_sampling, LUdecompose)
        self.coupling = LSTMAffineCouplingLayer(in_features, cond_features, rec_features)

    def run(self, xn, condn, state, reverse, mix=None, pad=0):
        """pad > 0: xn and mix are in the zero-padded channel layout of LSTMFLowBlock._pad_x (then mix is given)."""
        if not isinstance(self.conv, InvertibleConv1x1LU):
            return _run_unfused(self, xn, condn, state, reverse)
        extra = self._mix_logdet(xn) if mix is None else None      # (None: the caller accounts for the mix log-dets of the level)
        if reverse:
            t, ld, st = self.coupling.run(xn, condn, state, True, pad=pad)
            return self._mix(t, True, mix), (ld if extra is None else ld + extra), st
        y, ld, st = self.coupling.run(self._mix(xn, False, mix), condn, state, False, pad=pad)
        return y, (ld if extra is None else ld + extra), st

    def _call(self, x, cond, rec_states, reverse):
        st = None if rec_states is None else (H.nhwc(rec_states[0]), H.nhwc(rec_states[1]))
        y, ld, (h, c) = self.run(H.nhwc(x), H.nhwc(cond), st, reverse)
        return H.nchw(y), ld, (H.nchw(h), H.nchw(c))

    def forward(self, x, cond, rec_states=None):
        return self._call(x, cond, rec_states, False)

    def reverse(self, y, cond, rec_states=None):
        return self._call(y, cond, rec_states, True)


def _run_unfused(block, xn, condn, state, reverse):
    """Blocks built with the plain (non-LU) 1x1 conv: chain the stand-alone modules."""
    is_lstm = isinstance(block, LSTMCouplingBlock)
    norm = getattr(block, 'norm', None)
    x = H.nchw(xn)
    cond = H.nchw(condn)
    st_out = None
    if reverse:
        if is_lstm:
            st = None if state is None else (H.nchw(state[0]), H.nchw(state[1]))
            x, ld, st_out = block.coupling.reverse(x, cond, st)
        else:
            x, ld = block.coupling.reverse(x, cond)
        x, ld2 = block.conv.reverse(x)
        ld = ld + ld2
        if norm is not None:
            x, ld3 = norm.reverse(x)
            ld = ld + ld3
    else:
        ld = 0.
        if norm is not None:
            x, ld = norm(x)
        x, ld2 = block.conv(x)
        ld = ld + ld2
        if is_lstm:
            st = None if state is None else (H.nchw(state[0]), H.nchw(state[1]))
            x, ld3, st_out = block.coupling(x, cond, st)
        else:
            x, ld3 = block.coupling(x, cond)
        ld = ld + ld3
    if st_out is not None:
        st_out = (H.nhwc(st_out[0]), H.nhwc(st_out[1]))
    return H.nhwc(x), ld, st_out


class LSTMFLowBlock(nn.Module):
    """One flow level: squeeze -> K coupling blocks (first un-normed, last LSTM) -> split
    (reference :220-361).  Module names `revlayers.affine_layer{i}` as in the reference."""

    def __init__(self, in_features, cond_features, rec_features, n_layers, factor=2, LUdecompose=True, train_sampling=False,
                 do_split=True, squeeze_type=0):
        super().__init__()
        self.do_split = do_split
        self.n_layers = n_layers
        self.squeeze = CheckerSqueeze(factor) if squeeze_type == 0 else Squeeze(factor)
        in_features = in_features * factor ** 2
        self.revlayers = nn.Sequential()
        for i in range(n_layers - 1):
            cls = UnNormedAffineCouplingBlock if i == 0 else AffineCouplingBlock
            self.revlayers.add_module('affine_layer{}'.format(i + 1),
                                      cls(in_features, cond_features, LUdecompose=LUdecompose, train_sampling=train_sampling))
        self.revlayers.add_module('affine_layer{}'.format(n_layers),
                                  LSTMCouplingBlock(in_features, cond_features, rec_features, LUdecompose=LUdecompose,
                                                    train_sampling=train_sampling))
        if do_split:
            self.split = Split(in_features)

    def _all_lu(self):
        """Every layer of the level has a PLU-parameterised 1x1 conv of the same orientation (what _level_mix folds; TMGlow always
        builds such levels, tmGlow.py:366)."""
        convs = [l.conv for l in self.revlayers._modules.values()]
        return all(isinstance(c, InvertibleConv1x1LU) and c.train_sampling == convs[0].train_sampling for c in convs)

    def _level_mix(self, reverse, hw):
        """Fold ActNorm + PLU 1x1 conv of ALL K layers of this level in one batched pass of O(K C^3) torch ops
        (parameter-side bookkeeping, reference glowConv.py:151-174 + actNorm.py:66-83) instead of K separate ones.
        Returns ([K,C,C] matrices, [K,C] biases, scalar log-det of all K mixes) or None if a layer is not LU."""
        layers = list(self.revlayers._modules.values())
        convs = [l.conv for l in layers]
        if not all(isinstance(c, InvertibleConv1x1LU) and c.train_sampling == convs[0].train_sampling for c in convs):
            return None
        ts = convs[0].train_sampling
        # (tmg_lu_fold_* keep per-channel vectors in fixed 256-entry LDS arrays: wider levels take the batched torch fold below)
        if ((reverse if ts else not reverse) and convs[0].l.is_cuda and convs[0].l.shape[0] <= 256
                and os.environ.get("TMG_NO_LU_FOLD_KERNEL") is None):
            return self._level_mix_hip(layers, convs, reverse, hw)
        st = lambda name: torch.stack([getattr(c, name) for c in convs])  # noqa: E731
        eye = convs[0].eye
        logs = st('log_s')
        lower = st('l') * st('l_mask') + eye
        upper = st('u') * st('u_mask') + torch.diag_embed(logs.exp() * st('sign_s')) + 0.01 * eye
        P = st('p')
        ts = convs[0].train_sampling
        use_weight = reverse if ts else not reverse
        if use_weight:
            W = P @ (lower @ upper)
            with torch.no_grad():  # the reference's W-cache key (glowConv.py:157,209-212), all layers in one launch
                torch._foreach_copy_([c.log_s_old for c in convs], [c.log_s for c in convs])
        else:
            from nn.modules.glowConv import _TriInv
            W = _TriInv.apply(upper, True) @ (_TriInv.apply(lower, False) @ P.transpose(1, 2))
        ones = torch.ones_like(logs[0])
        a = torch.stack([l.norm.weight.view(-1) if hasattr(l, 'norm') else ones for l in layers])
        b = torch.stack([l.norm.bias.view(-1) if hasattr(l, 'norm') else torch.zeros_like(ones) for l in layers])
        if reverse:      # (W y - b) / a
            Wm = W / a.unsqueeze(2)
            bm = -b / a
        else:            # W (a x + b)
            Wm = W * a.unsqueeze(1)
            bm = (W @ b.unsqueeze(2)).squeeze(2)
        ld = logs.sum() * hw
        ld = (-ld if ts else ld) + a.abs().log().sum() * hw
        return Wm, bm, ld

    def _level_mix_hip(self, layers, convs, reverse, hw):
        """The W = P L U direction of _level_mix through tmg_lu_fold_fwd / _bwd (two launches; the parameters are read in place
        through a cached device pointer table, P as a cached row permutation)."""
        params = []
        for l, c in zip(layers, convs):
            nm = getattr(l, 'norm', None)       # a block without ActNorm: null pointers (scale 1, shift 0)
            params += [c.l, c.u, c.log_s, nm.weight if nm is not None else None, nm.bias if nm is not None else None]
        # the cache holds VALUES derived from the buffers p / sign_s (row permutations, the sign stack): load_state_dict copies into
        # them in place, so the key carries their version counters beside the addresses
        key = (tuple(t.data_ptr() if t is not None else 0 for t in params)
               + tuple(v for c in convs for v in (c.p.data_ptr(), c.p._version, c.sign_s.data_ptr(), c.sign_s._version)))
        cache = getattr(self, '_lu_fold_cache', None)
        if cache is None or cache[0] != key:
            dev = convs[0].l.device
            tab = torch.tensor([[t.data_ptr() if t is not None else 0 for t in params[5 * k:5 * k + 5]] for k in range(len(convs))],
                               dtype=torch.int64).to(dev)
            P = torch.stack([c.p for c in convs])
            perm = P.argmax(dim=2).to(torch.int32).contiguous()            # P[i, perm[i]] = 1
            iperm = P.argmax(dim=1).to(torch.int32).contiguous()           # P[iperm[r], r] = 1
            sign_s = torch.stack([c.sign_s for c in convs]).contiguous()
            cache = (key, tab, sign_s, perm, iperm)
            self._lu_fold_cache = cache
        _, tab, sign_s, perm, iperm = cache
        ts = convs[0].train_sampling
        with torch.no_grad():  # the reference's W-cache key (glowConv.py:157,209-212), all layers in one launch
            torch._foreach_copy_([c.log_s_old for c in convs], [c.log_s for c in convs])
        K, C = sign_s.shape
        meta = (tab, sign_s, perm, iperm, 1 if reverse else 0, -1.0 if ts else 1.0, float(hw), K, C)
        Wm, bm, ld, Wh, bh, Wt, bt = ops.LevelMixFoldFn.apply(meta, *params)
        return Wm, bm, ld.view(()), (Wh, bh, Wt, bt)

    def _cache(self):
        c = self.__dict__.get('_derived')
        if c is None:
            c = self.__dict__['_derived'] = ops.DerivedCache()     # (plain attribute: not a buffer, never part of the state_dict)
        return c

    def _mix_params(self):
        out = self.__dict__.get('_mix_param_list')     # (module attribute walks are slow: ~1 us each, 160 per level and call)
        if out is not None and len(out[1]) == len(self.revlayers._modules) and all(a is b for a, b in zip(out[1], self.revlayers._modules.values())):
            return out[0]
        out = []
        for l in self.revlayers._modules.values():
            c = l.conv
            # (not log_s_old: the fold itself writes that buffer - the reference's W-cache key, glowConv.py:157)
            out += [c.l, c.u, c.log_s, c.p, c.sign_s] if isinstance(c, InvertibleConv1x1LU) else list(c.parameters())
            if hasattr(l, 'norm'):
                out += [l.norm.weight, l.norm.bias]
        self.__dict__['_mix_param_list'] = (out, list(self.revlayers._modules.values()))
        return out

    def _level_mix_cached(self, reverse, hw, ch, pad):
        """(_level_mix(reverse, hw), padded Wm, padded bm) - evaluated once per BPTT window (ops.DerivedCache): the T time-steps of a
        window share the folded mixes, autograd sums their T gradients and runs the fold's backward launch once."""
        def build():
            lm = self._level_mix(reverse, hw)
            if lm is None:
                return (None, None, None)
            Wm, bm = self._pad_mix(lm[0], lm[1], ch, pad) if pad else (lm[0], lm[1])
            return (lm, Wm, bm)
        return self._cache().get(('mix', bool(reverse)), self._mix_params(), (int(hw), int(ch), int(pad)), build)

    def _fusable(self, lm, xn):
        """The level-fused node needs LU blocks throughout (lm) and >= 1 non-LSTM layer.  Halves that are not 16-byte aligned
        (3-channel fields: C = 12 on the first level) run the same node on zero-padded channels, see _level_call."""
        if lm is None or self.n_layers < 2 or os.environ.get("TMG_NO_LEVEL_FUSION"):
            return False
        layers = list(self.revlayers._modules.values())[:-1]
        return xn.shape[3] % 2 == 0 and all(l.coupling.coupling_nn.zero_conv.logscale_factor == 1 for l in layers)

    # ---- zero-padded channel layout --------------------------------------------------------------------------------
    # When the channel half ch = C/2 is not a multiple of 4 (the reference's own data sets have 3 output channels: C = 12,
    # ch = 6 on the first, largest level), every kernel of the fast path - float4 staging, the fused coupling launch, grouped
    # weight gradients, the lean MFMA staging of the ConvLSTM convs - would be off.  The level then runs on the layout
    # [x1 | 0.. | x2 | 0..] with ch rounded up to a multiple of 4: weights get zero rows / columns for the padding channels
    # (plain differentiable torch ops on parameter-sized tensors), so the padding channels stay exactly zero through every
    # layer (shift = 0, r = 0 => scale 1, log-det 0) and un-padding the result gives the un-padded computation.
    @staticmethod
    def _pad_x(xn, ch, pad):
        return ops.PadHalvesFn.apply(xn, ch, pad, True)       # one launch, zeros included (round 3: two fills + a cat)

    @staticmethod
    def _unpad_x(xp, ch, pad):
        return ops.PadHalvesFn.apply(xp, ch, pad, False)

    @staticmethod
    def _pad_mix(W, b, ch, pad):
        """[K, C, C] / [K, C] channel-mix parameters -> padded layout (zero rows / columns / biases for the padding channels)."""
        K, dev, chp = W.shape[0], W.device, ch + pad
        idx = torch.cat([torch.arange(ch, device=dev), torch.arange(chp, chp + ch, device=dev)])
        ar = torch.arange(K, device=dev)
        Wp = torch.zeros((K, 2 * chp, 2 * chp), device=dev).index_put((ar[:, None, None], idx[None, :, None], idx[None, None, :]), W)
        bp = torch.zeros((K, 2 * chp), device=dev).index_put((ar[:, None], idx[None, :]), b)
        return Wp, bp

    def _level_call(self, xn, condn, Wm, bm, layers, reverse, ch, pad):
        """All non-LSTM coupling layers of the level as one LevelCouplingFn node; xn, Wm, bm already padded when pad > 0."""
        wts = self._tail_weights(layers)
        if pad == 0:
            return ops.LevelCouplingFn.apply(xn, condn, Wm, bm, reverse, *wts)
        NL, dev = len(layers), xn.device

        def build():
            z = lambda *shape: torch.zeros(shape, device=dev, dtype=torch.float32)  # noqa: E731
            w1 = torch.stack(wts[0::5])                     # [NL, 1, ch + Cc, 3, 3]
            w2 = torch.stack(wts[1::5])                     # [NL, 1, ch + Cc + 1, 3, 3]
            wz = torch.stack(wts[2::5])                     # [NL, C, ch + Cc + 2, 3, 3]
            bz = torch.stack(wts[3::5])                     # [NL, C]
            ins = lambda w: torch.cat([w[:, :, :ch], z(NL, w.shape[1], pad, 3, 3), w[:, :, ch:]], 2)  # noqa: E731  zero rows after x1's
            w1p, w2p = ins(w1), ins(w2)
            wzp = torch.cat([ins(wz), z(NL, 2 * pad, wz.shape[2] + pad, 3, 3)], 1)   # padding pairs (shift, r) = zero output channels
            bzp = torch.cat([bz, z(NL, 2 * pad)], 1)
            out = []
            for k in range(NL):
                out += [w1p[k], w2p[k], wzp[k], bzp[k]]
            return out

        # the padded copies are parameter-sized functions of the weights alone: one evaluation per BPTT window, gradients summed over
        # the window's time-steps before they go back through the padding ops (ops.DerivedCache, proxies of the gradient sink)
        padded = self._cache().get(('tail', len(layers)), [w for i, w in enumerate(wts) if i % 5 != 4], (int(ch), int(pad)), build, proxies=True)
        wp = []
        for k in range(NL):
            wp += padded[4 * k:4 * k + 4] + [wts[5 * k + 4]]
        return ops.LevelCouplingFn.apply(xn, condn, Wm, bm, reverse, *wp)

    @staticmethod
    def _tail_weights(layers):
        out = []
        for l in layers:
            nn_ = l.coupling.coupling_nn
            out += [nn_.dense_block.denselayer1.conv1.weight, nn_.dense_block.denselayer2.conv1.weight, nn_.zero_conv.conv.weight,
                    nn_.zero_conv.conv.bias, nn_.zero_conv.scale]
        return out

    def _squeeze_nhwc(self, xn, to_small):
        if isinstance(self.squeeze, CheckerSqueeze):
            return ops.CheckerFn.apply(xn, to_small)
        x = H.nchw(xn)
        return H.nhwc(self.squeeze(x) if to_small else self.squeeze.reverse(x))

    def forward(self, x, cond, rec_states, return_eps=False):
        xn, condn = self._squeeze_nhwc(H.nhwc(x), True), H.nhwc(cond)
        st = None if rec_states is None else (H.nhwc(rec_states[0]), H.nhwc(rec_states[1]))
        layers = list(self.revlayers._modules.values())
        ch = xn.shape[3] // 2
        pad = (-ch) % 4 if (self._all_lu() and xn.shape[3] % 2 == 0 and not os.environ.get("TMG_NO_LEVEL_FUSION")) else 0
        lm, Wm, bm = self._level_mix_cached(False, xn.shape[1] * xn.shape[2], ch, pad)
        lds = [] if lm is None else [lm[2]]          # log-det terms of the level, summed by ONE launch at the end (ops.sum_logdet)
        out_states = []
        fused = self._fusable(lm, xn)
        if pad:
            xn = self._pad_x(xn, ch, pad)
        sp = lm[3] if (fused and not pad and len(lm) > 3) else None      # (head, tail) views of the fold node: no autograd slicing
        if fused:
            Wh, bh = (sp[0], sp[1]) if sp else (Wm[:-1], bm[:-1])
            xn, dld = self._level_call(xn, condn, Wh, bh, layers[:-1], False, ch, pad)
            lds.append(dld)
        for i, layer in enumerate(layers):
            if fused and i < self.n_layers - 1:
                continue
            mix = None if lm is None else ((sp[2], sp[3]) if sp else (Wm[i], bm[i]))
            if i == self.n_layers - 1:
                xn, dld, so = layer.run(xn, condn, st, False, mix, pad=pad)
                out_states = (H.nchw(so[0]), H.nchw(so[1]))
            else:
                xn, dld = layer.run(xn, condn, False, mix)
            lds.append(dld)
        if pad:
            xn = self._unpad_x(xn, ch, pad)
        B, dev = xn.shape[0], xn.device
        if self.do_split:
            z1, lp, eps = self.split(H.nchw(xn), return_eps=return_eps)
            return z1, ops.sum_logdet(lds + [lp], B, dev), out_states, eps
        return H.nchw(xn), ops.sum_logdet(lds, B, dev), out_states, None

    def reverse(self, y, cond, rec_states, eps=None, rng=None):
        """rng: (nonce, site) of the in-kernel latent draw when eps is None (TMGlow.sample hands one nonce to all levels)."""
        lds = []                                     # log-det terms of the level, summed by ONE launch at the end (ops.sum_logdet)
        out_states = []
        if self.do_split:
            y, lp = self.split.reverse(y, eps, rng=rng)
            lds.append(lp)
        yn, condn = H.nhwc(y), H.nhwc(cond)
        st = None if rec_states is None else (H.nhwc(rec_states[0]), H.nhwc(rec_states[1]))
        layers = list(self.revlayers._modules.values())
        ch = yn.shape[3] // 2
        pad = (-ch) % 4 if (self._all_lu() and yn.shape[3] % 2 == 0 and not os.environ.get("TMG_NO_LEVEL_FUSION")) else 0
        lm, Wm, bm = self._level_mix_cached(True, yn.shape[1] * yn.shape[2], ch, pad)
        if lm is not None:
            lds.append(lm[2])
        fused = self._fusable(lm, yn)
        if pad:
            yn = self._pad_x(yn, ch, pad)
        sp = lm[3] if (fused and not pad and len(lm) > 3) else None      # (head, tail) views of the fold node: no autograd slicing
        for i in range(len(layers) - 1, -1, -1):
            mix = None if lm is None else ((sp[2], sp[3]) if (sp and i == self.n_layers - 1) else (Wm[i], bm[i]))
            if i == self.n_layers - 1:
                yn, dld, so = layers[i].run(yn, condn, st, True, mix, pad=pad)
                out_states = (H.nchw(so[0]), H.nchw(so[1]))
            elif fused:
                Wh, bh = (sp[0], sp[1]) if sp else (Wm[:-1], bm[:-1])
                yn, dld = self._level_call(yn, condn, Wh, bh, layers[:-1], True, ch, pad)
                lds.append(dld)
                break
            else:
                yn, dld = layers[i].run(yn, condn, True, mix)
            lds.append(dld)
        if pad:
            yn = self._unpad_x(yn, ch, pad)
        return H.nchw(self._squeeze_nhwc(yn, False)), ops.sum_logdet(lds, yn.shape[0], yn.device), out_states
