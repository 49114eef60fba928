"""Block-level autograd Functions with hand-written backward passes.

The reference expresses BasicBlock (archs.py:205-241) and SPADE (normalization.py:67-122) as
chains of stock torch ops and lets autograd add the gradients of tensors that are consumed
twice (block input -> conv1 and shortcut; SPADE input -> x2map and the modulation).  Here a
whole block is ONE autograd node: the backward pass is an explicit kernel schedule in which
those gradient sums ride the dgrad kernel's residual epilogue, the ReLU masks are folded into
the batch-norm backward kernels, and `torch.cat` never materialises (two-pointer convs).
"""
import ctypes as C

import torch

from . import ops
from ._lib import ACT_NONE, ACT_RELU, ConvDesc, call, ptr, stream_ptr
from .ops import (_act_bwd, _bn_bwd_impl, _bn_fwd_impl, _channel_sum, _conv_dgrad_impl, _conv_fwd_impl, _conv_wgrad_impl,
                  _ld, new_nhwc, to_nhwc)


class _BasicBlockFn(torch.autograd.Function):
    """relu(bn1(conv3x3(x))) -> bn2(conv3x3(.)) -> (+ conv1x1(x) | + x) -> relu, x = cat(x1, x2)."""

    @staticmethod
    def forward(ctx, x1, x2, w1, g1, b1, rm1, rv1, w2, g2, b2, rm2, rv2, wsc, eps1, mom1, eps2, mom2, stride, var_mode, group):
        x1 = to_nhwc(x1)
        x2 = to_nhwc(x2) if x2 is not None else None
        # batch-norm statistics ride the producing conv's epilogue where its kernel has one (part = None otherwise)
        c1, part1 = _conv_fwd_impl(x1, x2, w1, None, stride, 1, ACT_NONE, 0.0, want_bn=True)
        # relu(bn1(c1)) is applied on conv2's INPUT where its kernel can (the split-operand k32 tiles): y1 is never written, the
        # backward's weight gradient reads c1 through the same transform and the ReLU mask is recomputed from c1 as before
        y1 = None
        c2 = None
        if ops.BN_FUSE_INPUT and part1 is not None:
            _, st1, cnt1 = _bn_fwd_impl(c1, g1, b1, rm1, rv1, None, eps1, mom1, ACT_RELU, 0.0, var_mode, group, part=part1, apply=False)
            r = _conv_fwd_impl(c1, None, w2, None, 1, 1, ACT_NONE, 0.0, want_bn=True, in_affine=(st1[2], st1[3], ACT_RELU, 0.0))
            if r is not None:
                c2, part2 = r
            else:
                with ops._hbm('bn_fwd', 8.0 * c1.numel()):
                    y1 = ops._bn_apply(c1, st1, None, ACT_RELU, 0.0)
        else:
            y1, st1, cnt1 = _bn_fwd_impl(c1, g1, b1, rm1, rv1, None, eps1, mom1, ACT_RELU, 0.0, var_mode, group, part=part1)
        if c2 is None:
            c2, part2 = _conv_fwd_impl(y1, None, w2, None, 1, 1, ACT_NONE, 0.0, want_bn=True)
        if wsc is not None:
            sc = _conv_fwd_impl(x1, x2, wsc, None, stride, 0, ACT_NONE, 0.0)
        else:
            if x2 is not None:
                raise ValueError('identity shortcut with a two-tensor input')
            sc = x1
        out, st2, cnt2 = _bn_fwd_impl(c2, g2, b2, rm2, rv2, sc, eps2, mom2, ACT_RELU, 0.0, var_mode, group, part=part2)
        ctx.save_for_backward(x1, x2, c1, y1, c2, out, w1, g1, w2, g2, wsc, st1, st2)      # y1 = None on the fused route
        ctx.cfg = (stride, group, cnt1, cnt2)
        return out

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, dout):
        x1, x2, c1, y1, c2, out, w1, g1, w2, g2, wsc, st1, st2 = ctx.saved_tensors
        stride, group, cnt1, cnt2 = ctx.cfg
        dout = to_nhwc(dout)
        n, ca, h, w = x1.shape
        cb = x2.shape[1] if x2 is not None else 0
        need_x1, need_x2 = ctx.needs_input_grad[0], (x2 is not None and ctx.needs_input_grad[1])
        # bn2 backward; dres = dout masked by the final ReLU = gradient of the shortcut branch
        dc2, g, dg2, db2 = _bn_bwd_impl(c2, out, dout, g2, st2, ACT_RELU, 0.0, group, cnt2, want_dres=True)
        if y1 is None:                                   # fused route: conv2 read relu(bn1(c1)) through its input transform
            dw2 = _conv_wgrad_impl(c1, None, dc2, w2.shape, 1, 1, in_affine=(st1[2], st1[3], ACT_RELU, 0.0))
            if dw2 is None:                              # no weight-gradient kernel with the transform for this shape: write y1 now
                with ops._hbm('bn_fwd', 8.0 * c1.numel()):
                    y1w = ops._bn_apply(c1, st1, None, ACT_RELU, 0.0)
                dw2 = _conv_wgrad_impl(y1w, None, dc2, w2.shape, 1, 1)
                del y1w
        else:
            dw2 = _conv_wgrad_impl(y1, None, dc2, w2.shape, 1, 1)
        # bn1's backward sums ride the epilogue of the input gradient that produces d(relu(bn1(c1))) where its kernel has one
        # (ssg_conv_desc.bwd_x: the k32 tiles): the reduce pass over (dy1, c1) goes, the apply pass reads the already masked gradient
        r = _conv_dgrad_impl(dc2, w2, 1, 1, c1.shape[2], c1.shape[3], 0, c1.shape[1], bwd_stats=(c1, st1, ACT_RELU, 0.0)) \
            if ops.BN_BWD_EPILOGUE else None
        if r is not None:
            dc1, dg1, db1 = ops._bn_bwd_from_partials(c1, r[0], r[1], g1, st1, group, cnt1)
        else:
            dy1 = _conv_dgrad_impl(dc2, w2, 1, 1, c1.shape[2], c1.shape[3], 0, c1.shape[1])
            dc1, _, dg1, db1 = _bn_bwd_impl(c1, y1, dy1, g1, st1, ACT_RELU, 0.0, group, cnt1, want_dres=False, had_res=False)
        dw1 = _conv_wgrad_impl(x1, x2, dc1, w1.shape, stride, 1)
        dwsc = _conv_wgrad_impl(x1, x2, g, wsc.shape, stride, 0) if wsc is not None else None
        dx1 = dx2 = None
        if stride != 1 and (need_x1 or need_x2):
            raise NotImplementedError('strided BasicBlock input gradient')
        if need_x1:
            part = _conv_dgrad_impl(g, wsc, 1, 0, h, w, 0, ca) if wsc is not None else g
            dx1 = _conv_dgrad_impl(dc1, w1, 1, 1, h, w, 0, ca, res=part)
        if need_x2:
            part = _conv_dgrad_impl(g, wsc, 1, 0, h, w, ca, ca + cb)
            dx2 = _conv_dgrad_impl(dc1, w1, 1, 1, h, w, ca, ca + cb, res=part)
        return (dx1, dx2, dw1, dg1, db1, None, None, dw2, dg2, db2, None, None, dwsc,
                None, None, None, None, None, None, None)


def basic_block(x1, x2, conv1, bn1, conv2, bn2, shortcut_conv, group=None):
    """Fused training-mode BasicBlock over nn.Conv2d / nn.BatchNorm2d parameter holders."""
    for bn in (bn1, bn2):
        # the reference's synchronised branch never touches num_batches_tracked (batchnorm.py:57-80); its stock branch does
        if bn.track_running_stats and bn.num_batches_tracked is not None and not ops._synced(group):
            bn.num_batches_tracked.add_(1)
    var_mode = getattr(bn1, '_ssg_var_mode', 1 if group is not None else 0)
    return _BasicBlockFn.apply(x1, x2, conv1.weight, bn1.weight, bn1.bias, bn1.running_mean, bn1.running_var,
                               conv2.weight, bn2.weight, bn2.bias, bn2.running_mean, bn2.running_var,
                               shortcut_conv.weight if shortcut_conv is not None else None,
                               float(bn1.eps), float(bn1.momentum), float(bn2.eps), float(bn2.momentum),
                               int(conv1.stride[0]), var_mode, group)


def _spade_cat(wg, bg, wb, bb):
    """gamma and beta convs as ONE conv nhidden -> 2C: weights / biases concatenated along the output channels.  Cached on
    the gamma weight for one (storage, version, optimizer epoch) of the four tensors, so an eval loop concatenates once."""
    stamp = tuple((t.data_ptr(), t._version) for t in (wg, bg, wb, bb)) + (ops._WEIGHT_EPOCH[0],)
    hit = wg.__dict__.get('_ssg_gb_cat')
    if hit is not None and hit[0] == stamp:
        return hit[1], hit[2]
    with torch.no_grad():
        w = torch.cat([wg.detach(), wb.detach()], 0).contiguous()
        b = torch.cat([bg.detach(), bb.detach()], 0).contiguous()
    try:
        wg._ssg_gb_cat = (stamp, w, b)
    except Exception:
        pass
    return w, b


def _spade_fused_fwd(x, a, wgb, bgb, pad, out):
    """gamma|beta conv + modulation in one kernel (ssg_spade_conv_modulate_f32) where the shape allows it (4- or 8-channel `a`,
    3x3, >= 65536 pixels: the 512^2 and 256^2 levels at batch 16).  Writes `out`, returns gamma (needed by the backward) or None."""
    o, i, kh, kw = wgb.shape
    n, c, h, w = x.shape
    cin = ops.pad4(a.shape[1])
    if kh != 3 or kw != 3 or pad != 1 or cin not in (4, 8) or o != 2 * c or c % 4:
        return None
    taps = ops._taps_fwd(kh, kw, pad)
    wpk, kp, kmode = ops._pack(wgb, 0, taps, cin, cin)
    d = ConvDesc()
    d.in1 = a.data_ptr(); d.C1 = cin; d.ld1 = _ld(a); d.in2 = None; d.C2 = 0; d.ld2 = 0
    d.N, d.H, d.W = n, h, w
    d.w = wpk.data_ptr(); d.Kp = kp; d.kmode = kmode
    d.bias = bgb.data_ptr(); d.res = None; d.ldr = 0
    d.out = out.data_ptr(); d.Cout = o; d.ldo = _ld(out)
    d.GH, d.GW, d.OH, d.OW = h, w, h, w
    d.in_sy = d.in_sx = d.out_sy = d.out_sx = 1
    d.out_oy = d.out_ox = 0
    ops._fill_taps(d, taps)
    d.act = ACT_NONE; d.slope = 0.0; d.bnpart = None; d.ws = None; d.ws_bytes = 0
    if not call('ssg_spade_conv_modulate_ok', C.byref(d)):
        return None
    gamma = new_nhwc(n, c, h, w, x.device)
    with ops._Timed('thin32_cin_kernel<spade>' if ops.PROFILE is not None else None, 2.0 * n * h * w * o * a.shape[1] * 9):
        call('ssg_spade_conv_modulate_f32', C.byref(d), ptr(x), _ld(x), ptr(gamma), _ld(gamma), stream_ptr())
    return gamma


class _SpadeFn(torch.autograd.Function):
    """Self-conditioned SPADE: out = x*(1+gamma(a)) + beta(a), a = relu(shared(x2map(x))).
    gamma and beta come from one conv nhidden -> 2C (concatenated weights) writing gamma|beta pixel rows; its input
    gradient is then one launch with K = 9*2C and its weight gradient one launch whose halves are d(gamma), d(beta)."""

    @staticmethod
    def forward(ctx, x, wx, bx, ws, bs, wg, bg, wb, bb, pad):
        x = to_nhwc(x)
        n, c, h, w = x.shape
        seg = _conv_fwd_impl(x, None, wx, bx, 1, pad, ACT_NONE, 0.0)
        a = _conv_fwd_impl(seg, None, ws, bs, 1, pad, ACT_RELU, 0.0)
        wgb, bgb = _spade_cat(wg, bg, wb, bb)
        out = new_nhwc(n, c, h, w, x.device)
        gb = _spade_fused_fwd(x, a, wgb, bgb, pad, out)           # gamma only ([n, c, h, w]) when the fused kernel took it
        if gb is None:
            gb = _conv_fwd_impl(a, None, wgb, bgb, 1, pad, ACT_NONE, 0.0)
            with ops._hbm('spade_modulate_fwd', 16.0 * n * h * w * c):           # x, gamma, beta in; out (3 reads + 1 write)
                call('ssg_spade_modulate_fwd_f32', ptr(x), _ld(x), ptr(gb), _ld(gb), n * h * w, c, ptr(out), _ld(out), stream_ptr())
        ctx.save_for_backward(x, seg, a, gb, wx, ws, wgb)
        ctx.pad = pad
        return out

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, dout):
        x, seg, a, gb, wx, ws, wgb = ctx.saved_tensors
        pad = ctx.pad
        dout = to_nhwc(dout)
        n, c, h, w = x.shape
        dxm = new_nhwc(n, c, h, w, x.device)
        dgb = new_nhwc(n, 2 * c, h, w, x.device)
        # modulate backward and the gamma / beta bias gradients (column sums of dgb) in one pass over the data
        scratch = ops._ws(call('ssg_bn_workspace_bytes', n * h * w, c), x.device)
        sums = torch.empty(2 * c, dtype=torch.float64, device=x.device)
        # x, gamma (|beta where the conv wrote both), dout in; dx, d(gamma), d(beta) out
        with ops._hbm('spade_modulate_bwd', 4.0 * n * h * w * c * (3 + 3)):
            call('ssg_spade_modulate_bwd_sums_f32', ptr(x), _ld(x), ptr(gb), _ld(gb), ptr(dout), _ld(dout), n * h * w, c,
                 ptr(dxm), _ld(dxm), ptr(dgb), _ld(dgb), ptr(sums), ptr(scratch), stream_ptr())
        dwgb = _conv_wgrad_impl(a, None, dgb, wgb.shape, 1, pad)
        dbias_gb = sums.float()
        nh = a.shape[1]
        da = _conv_dgrad_impl(dgb, wgb, 1, pad, h, w, 0, nh)
        da = _act_bwd(a, da, ACT_RELU, 0.0)
        dws = _conv_wgrad_impl(seg, None, da, ws.shape, 1, pad)
        dbs = _channel_sum(da, nh)
        dseg = _conv_dgrad_impl(da, ws, 1, pad, h, w, 0, seg.shape[1])
        dwx = _conv_wgrad_impl(x, None, dseg, wx.shape, 1, pad)
        dbx = _channel_sum(dseg, seg.shape[1])
        dx = _conv_dgrad_impl(dseg, wx, 1, pad, h, w, 0, c, res=dxm) if ctx.needs_input_grad[0] else None
        return dx, dwx, dbx, dws, dbs, dwgb[:c], dbias_gb[:c], dwgb[c:], dbias_gb[c:], None


def spade_self(x, x2map, shared, gamma, beta):
    """Fused SPADE(x, x) over the four nn.Conv2d parameter holders."""
    return _SpadeFn.apply(x, x2map.weight, x2map.bias, shared.weight, shared.bias, gamma.weight, gamma.bias,
                          beta.weight, beta.bias, int(x2map.padding[0]))
