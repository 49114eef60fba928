"""bf16 tensor family for the EfficientNet MBConv path (BASELINE config 4: "EfficientNet-B4 encoder, 3x1024x1024 tiles,
bf16 MFMA").  Activations live in HBM as bf16 channels_last tensors (C % 8 == 0: 16-byte pixel rows); parameters, batch-norm
statistics, squeeze-excite gates, weight gradients and optimizer state stay fp32.  The pointwise convolutions
run on v_mfma_f32_32x32x16_bf16 (csrc/gemm_bf16.hip), the HBM-bound pieces on the bf16 instantiations of the fp32 kernels
(8-byte loads, fp32 arithmetic, fp64 statistics).  As everywhere in this package there is no fallback: a CPU tensor or a
missing library raises.

Reference ops restated here: efficientnet_pytorch/model.py:18-99 (MBConvBlock pieces), utils.py:37-48 (swish backward)."""
import ctypes as C

import torch

from . import _lib, ops
from ._lib import ACT_NONE, call, ptr, stream_ptr

ACT_SWISH = 3          # SSG_ACT_SWISH
BF16 = torch.bfloat16


def new_bf16(n, c, h, w, device):
    if c % 8:
        raise ValueError('bf16 tensors need C %% 8 == 0 (16-byte pixel rows); got C=%d' % c)
    return torch.empty((n, c, h, w), dtype=BF16, device=device, memory_format=torch.channels_last)


def _ld(x):
    """Pixel stride of a bf16 NHWC tensor, after checking it is one the kernels can take."""
    if x.dtype != BF16 or x.dim() != 4:
        raise _lib.HipLibraryError('internal: expected a 4-d bf16 tensor, got %s %s' % (x.dtype, tuple(x.shape)))
    n, c, h, w = x.shape
    if c % 8 or not x.is_contiguous(memory_format=torch.channels_last) or x.data_ptr() % 16:
        raise _lib.HipLibraryError('internal: bf16 tensor is not dense channels_last with C %% 8 == 0: shape %s stride %s' % (tuple(x.shape), x.stride()))
    return c


def as_bf16(x):
    """bf16 channels_last view/copy of a gradient tensor autograd hands back (already ours in the common case)."""
    _lib.require_gpu(x)
    if x.dtype == BF16 and x.dim() == 4 and x.shape[1] % 8 == 0 and x.is_contiguous(memory_format=torch.channels_last) and x.data_ptr() % 16 == 0:
        return x
    if x.dtype == BF16:                         # expanded / non-dense gradient (e.g. from .sum()): materialise it
        out = new_bf16(*x.shape, device=x.device)
        out.copy_(x)
        return out
    return to_bf16(x)


# ----------------------------------------------------------------------------- dtype boundary
class _ToBF16(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        x = ops.to_nhwc(x)
        n, c, h, w = x.shape
        y = new_bf16(n, c, h, w, x.device)
        call('ssg_convert_f32_to_bf16', ptr(x), ops._ld(x), n * h * w, c, ptr(y), c, stream_ptr())
        return y

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g):
        return _to_f32_impl(as_bf16(g))


def _to_f32_impl(x):
    n, c, h, w = x.shape
    y = ops.new_nhwc(n, c, h, w, x.device)
    call('ssg_convert_bf16_to_f32', ptr(x), _ld(x), n * h * w, c, ptr(y), ops._ld(y), stream_ptr())
    return y


class _ToF32(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        return _to_f32_impl(x)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g):
        g = ops.to_nhwc(g)
        n, c, h, w = g.shape
        y = new_bf16(n, c, h, w, g.device)
        call('ssg_convert_f32_to_bf16', ptr(g), ops._ld(g), n * h * w, c, ptr(y), c, stream_ptr())
        return y


def to_bf16(x):
    """fp32 NHWC tensor -> bf16 (differentiable)."""
    _lib.require_gpu(x)
    return _ToBF16.apply(x) if x.requires_grad else _ToBF16.forward(None, x)


def to_f32(x):
    """bf16 tensor -> fp32 NHWC-with-stride tensor (differentiable)."""
    _lib.require_gpu(x)
    return _ToF32.apply(x) if x.requires_grad else _to_f32_impl(x)


# ----------------------------------------------------------------------------- dense k x k conv of the image (the stem)
class _ConvThin(torch.autograd.Function):
    """Conv2d(Cin <= 4 -> Cout, k x k, stride 1 / 2, no bias) of the fp32 input image into a bf16 tensor (model.py:162,206: the stem;
    csrc/conv_stem_bf16.hip): operands rounded to bf16, fp32 accumulation."""

    @staticmethod
    def forward(ctx, x, weight, stride, pad):
        x = ops.to_nhwc(x)
        n, cin, h, w = x.shape
        o, i, kh, kw = weight.shape
        if i != cin or cin > 4 or o % 8:
            raise ValueError('conv_thin: needs Cin <= 4 and Cout %% 8 == 0 (weight %s, input channels %d)' % (tuple(weight.shape), cin))
        pt, pb, pl, pr = ops._pad4(pad)
        oh, ow = ops._out_hw(h, w, kh, kw, stride, pad)
        y = new_bf16(n, o, oh, ow, x.device)
        wc = weight.detach().contiguous()
        call('ssg_conv2d_thin_bf16', ptr(x), n, h, w, ops._ld(x), ptr(wc), o, cin, kh, kw, stride, pt, pl, oh, ow, ptr(y), o, stream_ptr())
        ctx.save_for_backward(x, wc)
        ctx.cfg = (stride, pt, pl, oh, ow)
        return y

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, dy):
        x, wc = ctx.saved_tensors
        stride, pt, pl, oh, ow = ctx.cfg
        dy = as_bf16(dy)
        n, cin, h, w = x.shape
        o, _, kh, kw = wc.shape
        dw = dx = None
        if ctx.needs_input_grad[1]:
            dw = torch.empty_like(wc)
            ws = ops._ws(call('ssg_conv2d_thin_bf16_wgrad_workspace_bytes', n, oh, ow, o, kh, kw), x.device)
            call('ssg_conv2d_thin_bf16_wgrad', ptr(x), n, h, w, ops._ld(x), ptr(dy), o, o, cin, kh, kw, stride, pt, pl, oh, ow, ptr(dw), ptr(ws),
                 stream_ptr())
        if ctx.needs_input_grad[0]:
            dx = ops.new_nhwc(n, cin, h, w, x.device)
            call('ssg_conv2d_thin_bf16_dgrad', ptr(dy), o, n, h, w, ptr(wc), o, cin, kh, kw, stride, pt, pl, oh, ow, ptr(dx), ops._ld(dx), stream_ptr())
        return dx, dw, None, None


def conv_thin(x, weight, stride=1, pad=0):
    """Dense conv of a <= 4-channel fp32 image into the bf16 family (the EfficientNet stem)."""
    _lib.require_gpu(x)
    return _ConvThin.apply(x, weight, int(stride), pad)


# ----------------------------------------------------------------------------- pointwise conv = GEMM on the bf16 MFMA
def _pack(weight, transpose):
    """fp32 [O, I, 1, 1] parameter -> bf16 [rows_pad][Kp] operand (cached on the parameter like ops._pack)."""
    o, i = weight.shape[0], weight.shape[1]
    rows, cols = (i, o) if transpose else (o, i)
    rows_pad, kp = (rows + 127) // 128 * 128, (cols + 31) // 32 * 32
    stamp = (weight.data_ptr(), weight._version, ops._WEIGHT_EPOCH[0])
    cache = weight.__dict__.get('_ssg_pack_bf16')
    if cache is None or cache[0] != stamp:
        cache = (stamp, {})
        try:
            weight._ssg_pack_bf16 = cache
        except Exception:
            pass
    hit = cache[1].get(transpose)
    if hit is None:
        hit = torch.empty((rows_pad, kp), dtype=BF16, device=weight.device)
        wc = weight.detach().contiguous()
        call('ssg_pack_weights_bf16', ptr(wc), o, i, int(transpose), rows_pad, kp, ptr(hit), stream_ptr())
        cache[1][transpose] = hit
    return hit, kp


PROFILE_SHAPES = False  # append the GEMM shape to the label
PROFILE = None          # tools/bench_b4.py: list receiving (label, flops, bytes, start event, end event) per GEMM launch


def _timed(label, flops, nbytes, fn):
    if PROFILE is None:
        return fn()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record(torch.cuda.current_stream())
    r = fn()
    e1.record(torch.cuda.current_stream())
    PROFILE.append((label, flops, nbytes, e0, e1))
    return r


class _Conv1x1(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight):
        ldx = _ld(x)
        n, c, h, w = x.shape
        o = weight.shape[0]
        if weight.shape[1] != c or weight.shape[2:] != (1, 1) or o % 8:
            raise ValueError('conv1x1_bf16: weight %s does not fit input C=%d (Cout %% 8 == 0 required)' % (tuple(weight.shape), c))
        wp, kp = _pack(weight, 0)
        y = new_bf16(n, o, h, w, x.device)
        p = n * h * w
        _timed('gemm_bf16_kernel fwd' + (' P%d K%d N%d' % (p, c, o) if PROFILE_SHAPES else ''), 2.0 * p * c * o, 2.0 * p * (c + o) + 2.0 * c * o,
               lambda: call('ssg_gemm_bf16', ptr(x), p, c, ldx, ptr(wp), kp, o, None, 0, ptr(y), o, stream_ptr()))
        ctx.save_for_backward(x, weight)
        return y

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        dy = as_bf16(dy)
        n, c, h, w = x.shape
        o = weight.shape[0]
        p = n * h * w
        dx = dw = None
        if ctx.needs_input_grad[0]:
            wt, kpt = _pack(weight, 1)
            dx = new_bf16(n, c, h, w, x.device)
            _timed('gemm_bf16_kernel dgrad' + (' P%d K%d N%d' % (p, o, c) if PROFILE_SHAPES else ''), 2.0 * p * c * o, 2.0 * p * (c + o) + 2.0 * c * o,
                   lambda: call('ssg_gemm_bf16', ptr(dy), p, o, o, ptr(wt), kpt, c, None, 0, ptr(dx), c, stream_ptr()))
        if ctx.needs_input_grad[1]:
            nbytes = call('ssg_gemm_wgrad_bf16_workspace_bytes', p, o, c)
            ws = ops._ws(nbytes, x.device)
            dw = torch.empty((o, c, 1, 1), dtype=torch.float32, device=x.device)
            _timed('gemm_wgrad_bf16_kernel' + (' P%d M%d N%d' % (p, o, c) if PROFILE_SHAPES else ''), 2.0 * p * c * o, 2.0 * p * (c + o) + 4.0 * c * o,
                   lambda: call('ssg_gemm_wgrad_bf16', ptr(dy), o, ptr(x), c, p, o, c, ptr(dw), ptr(ws), nbytes, stream_ptr()))
        return dx, dw


def conv1x1(x, weight):
    """F.conv2d(x, weight) for a [O, I, 1, 1] fp32 parameter on a bf16 tensor (bias-free: model.py:40,56)."""
    _lib.require_gpu(x)
    return _Conv1x1.apply(x, weight)


# ----------------------------------------------------------------------------- batch norm (+ swish | + residual)
class _BNAct(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, running_mean, running_var, res, eps, momentum, act, var_mode, group):
        ldx = _ld(x)
        n, c, h, w = x.shape
        p = n * h * w
        dev = x.device
        if res is not None:
            _ld(res)
        # synchronised statistics (BASELINE config 4 "on 8 x MI355X"): exactly ops._bn_fwd_impl -- the fp64 [sum, sum^2, pixel
        # count] vector is all-reduced between the two stages, the reference's sync formula clamp(var, eps)^-1/2 (var_mode 1)
        synced = ops._synced(group)
        ws = ops._ws(call('ssg_bn_workspace_bytes', p, c), dev)
        stats = torch.empty((4, c), dtype=torch.float32, device=dev)
        if not synced and ops.BN_FUSED_FINALIZE:
            fin = ops.bn_fin(weight, bias, eps, momentum, var_mode, running_mean, running_var, stats)
            call('ssg_bn_stats_finalize_bf16', ptr(x), p, c, ldx, C.byref(fin), ptr(ws), stream_ptr())
        else:
            sums = torch.empty(2 * c + 1, dtype=torch.float64, device=dev)
            call('ssg_bn_stats_bf16', ptr(x), p, c, ldx, ptr(sums), int(synced), ptr(ws), stream_ptr())
            if synced:
                ops._timed_all_reduce('sync_bn_fwd', sums, group)
            call('ssg_bn_finalize_f32', ptr(sums), 0.0 if synced else float(p), c, ptr(weight), ptr(bias), eps, momentum, var_mode,
                 ptr(running_mean), ptr(running_var), ptr(stats[0]), ptr(stats[1]), ptr(stats[2]), ptr(stats[3]), stream_ptr())
        if running_mean is not None or running_var is not None:
            ops._STATS_EPOCH[0] += 1
        y = new_bf16(n, c, h, w, dev)
        call('ssg_bn_apply_bf16', ptr(x), p, c, ldx, ptr(stats[2]), ptr(stats[3]), ptr(res), c if res is not None else 0, act, 0.0,
             ptr(y), c, stream_ptr())
        ctx.save_for_backward(x, weight, stats)
        ctx.cfg = (act, res is not None, group if synced else None)
        return y

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, dy):
        x, weight, stats = ctx.saved_tensors
        act, has_res, group = ctx.cfg
        synced = group is not None
        dy = as_bf16(dy)
        n, c, h, w = x.shape
        p = n * h * w
        dev = x.device
        ws = ops._ws(call('ssg_bn_workspace_bytes', p, c), dev)
        sums = torch.empty(2 * c + 1, dtype=torch.float64, device=dev)
        call('ssg_bn_bwd_reduce_bf16', ptr(x), None, ptr(dy), p, c, c, 0, c, ptr(stats[0]), ptr(stats[1]), ptr(stats[2]), ptr(stats[3]),
             act, 0.0, ptr(sums), int(synced), ptr(ws), stream_ptr())
        local = sums.clone() if synced else sums          # this rank's weight / bias gradients; data-parallel averages them later
        if synced:
            ops._timed_all_reduce('sync_bn_bwd', sums, group)
        dwb = torch.empty((2, c), dtype=torch.float32, device=dev)
        dx = new_bf16(n, c, h, w, dev)
        call('ssg_bn_bwd_apply_bf16', ptr(x), None, ptr(dy), p, c, c, 0, c, ptr(stats[0]), ptr(stats[1]), ptr(weight), ptr(stats[2]), ptr(stats[3]),
             ptr(sums), 0.0 if synced else float(p), act, 0.0, ptr(dx), c, None, 0, ptr(dwb[0]), ptr(dwb[1]), stream_ptr())
        if synced:
            dwb = torch.stack([local[c:2 * c], local[:c]]).float()
        dres = dy if (has_res and ctx.needs_input_grad[5]) else None         # no activation after the residual add (model.py:93-97)
        return dx, dwb[0], dwb[1], None, None, dres, None, None, None, None, None


class _Affine(torch.autograd.Function):
    """Eval-mode batch norm on a bf16 tensor: y = act(x*scale + shift (+res))."""

    @staticmethod
    def forward(ctx, x, scale, shift, res, act):
        n, c, h, w = x.shape
        y = new_bf16(n, c, h, w, x.device)
        call('ssg_bn_apply_bf16', ptr(x), n * h * w, c, _ld(x), ptr(scale), ptr(shift), ptr(res), c if res is not None else 0, act, 0.0,
             ptr(y), c, stream_ptr())
        return y

    @staticmethod
    def backward(ctx, g):
        raise NotImplementedError('eval-mode batch norm is inference-only in ssunet-gan_amd')


def batch_norm_act(x, bn, res=None, act=ACT_NONE):
    """nn.BatchNorm2d `bn` on a bf16 tensor, optionally fused with swish (act=ACT_SWISH) or a residual add."""
    _lib.require_gpu(x)
    if act == ACT_SWISH and res is not None:
        raise ValueError('swish with a residual is not a pattern of the MBConv block')
    if bn.training or not bn.track_running_stats:
        if bn.momentum is None:
            raise NotImplementedError('cumulative-average batch norm (momentum=None)')
        group = getattr(bn, '_ssg_sync_group', None)          # dp.convert_sync_batchnorm
        if bn.track_running_stats and bn.num_batches_tracked is not None and not ops._synced(group):
            bn.num_batches_tracked.add_(1)           # the reference's synchronised branch never counts batches (batchnorm.py:57-80)
        var_mode = getattr(bn, '_ssg_var_mode', 1 if group is not None else 0)
        return _BNAct.apply(x, bn.weight, bn.bias, bn.running_mean if bn.track_running_stats else None,
                            bn.running_var if bn.track_running_stats else None, res, float(bn.eps), float(bn.momentum), int(act),
                            int(var_mode), group)
    with torch.no_grad():
        scale = torch.rsqrt(bn.running_var + bn.eps) * bn.weight
        shift = bn.bias - bn.running_mean * scale
    return _Affine.apply(x, scale.contiguous(), shift.contiguous(), res, int(act))


# ----------------------------------------------------------------------------- depthwise conv
class _DwConv(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, stride, pad):
        ldx = _ld(x)
        n, c, h, w = x.shape
        if weight.shape[0] != c or weight.shape[1] != 1:
            raise ValueError('dwconv_bf16: depthwise weight [C,1,KH,KW] expected, got %s for C=%d' % (tuple(weight.shape), c))
        kh, kw = weight.shape[2:]
        pt, pb, pl, pr = ops._pad4(pad)
        oh, ow = ops._out_hw(h, w, kh, kw, stride, pad)
        y = new_bf16(n, c, oh, ow, x.device)
        wc = weight.contiguous()
        call('ssg_dwconv2d_fwd_bf16', ptr(x), n, h, w, c, ldx, ptr(wc), None, kh, kw, stride, pt, pl, oh, ow, ptr(y), c, stream_ptr())
        ctx.save_for_backward(x, wc)
        ctx.cfg = (stride, pt, pl, oh, ow)
        return y

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, dy):
        x, wc = ctx.saved_tensors
        stride, pt, pl, oh, ow = ctx.cfg
        dy = as_bf16(dy)
        n, c, h, w = x.shape
        kh, kw = wc.shape[2:]
        dx = dw = None
        if ctx.needs_input_grad[0]:
            dx = new_bf16(n, c, h, w, x.device)
            call('ssg_dwconv2d_dgrad_bf16', ptr(dy), c, n, h, w, c, ptr(wc), kh, kw, stride, pt, pl, oh, ow, ptr(dx), c, stream_ptr())
        if ctx.needs_input_grad[1]:
            dw = torch.empty_like(wc)
            ws = ops._ws(call('ssg_dwconv2d_wgrad_workspace_bytes', n, oh, ow, c, kh, kw), x.device)
            call('ssg_dwconv2d_wgrad_bf16', ptr(x), n, h, w, c, c, ptr(dy), c, kh, kw, stride, pt, pl, oh, ow, ptr(dw), ptr(ws), stream_ptr())
        return dx, dw, None, None


def dwconv2d(x, weight, stride=1, padding=0):
    _lib.require_gpu(x)
    pad = tuple(int(v) for v in padding) if isinstance(padding, (tuple, list)) else int(padding)
    return _DwConv.apply(x, weight, int(stride), pad)


# ----------------------------------------------------------------------------- squeeze-excite pieces
class _GlobalAvgPool(torch.autograd.Function):
    """F.adaptive_avg_pool2d(x, 1) of a bf16 tensor -> fp32 [N, C, 1, 1] (the tiny SE convs run in fp32)."""

    @staticmethod
    def forward(ctx, x):
        ldx = _ld(x)
        n, c, h, w = x.shape
        y = ops.new_nhwc(n, c, 1, 1, x.device)
        ws = ops._ws(call('ssg_sample_channel_sum_workspace_bytes', n, h * w, c), x.device)
        call('ssg_sample_channel_sum_bf16', ptr(x), ldx, None, 0, n, h * w, c, 1.0 / (h * w), ptr(y), ptr(ws), stream_ptr())
        ctx.cfg = (n, c, h, w)
        return y

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, dy):
        n, c, h, w = ctx.cfg
        dy = ops.to_nhwc(dy)
        dx = new_bf16(n, c, h, w, dy.device)
        call('ssg_broadcast_rows_bf16', ptr(dy), n, h * w, c, 1.0 / (h * w), ptr(dx), c, stream_ptr())
        return dx


def global_avgpool(x):
    _lib.require_gpu(x)
    return _GlobalAvgPool.apply(x)


class _ChannelScale(torch.autograd.Function):
    """x * s with x bf16 and s an fp32 [N, C, 1, 1] gate (SE gate, drop-connect mask)."""

    @staticmethod
    def forward(ctx, x, s):
        ldx = _ld(x)
        s = ops.to_nhwc(s)
        n, c, h, w = x.shape
        if tuple(s.shape) != (n, c, 1, 1):
            raise ValueError('channel_scale_bf16: gate %s does not match %s' % (tuple(s.shape), tuple(x.shape)))
        y = new_bf16(n, c, h, w, x.device)
        call('ssg_channel_scale_fwd_bf16', ptr(x), ldx, ptr(s), n, h * w, c, ptr(y), c, stream_ptr())
        ctx.save_for_backward(x, s)
        return y

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, dy):
        x, s = ctx.saved_tensors
        dy = as_bf16(dy)
        n, c, h, w = x.shape
        dx = ds = None
        if ctx.needs_input_grad[0]:
            dx = new_bf16(n, c, h, w, x.device)
            call('ssg_channel_scale_fwd_bf16', ptr(dy), c, ptr(s), n, h * w, c, ptr(dx), c, stream_ptr())
        if ctx.needs_input_grad[1]:
            ds = ops.new_nhwc(n, c, 1, 1, x.device)
            ws = ops._ws(call('ssg_sample_channel_sum_workspace_bytes', n, h * w, c), x.device)
            call('ssg_sample_channel_sum_bf16', ptr(dy), c, ptr(x), c, n, h * w, c, 1.0, ptr(ds), ptr(ws), stream_ptr())
        return dx, ds


def channel_scale(x, s):
    _lib.require_gpu(x)
    return _ChannelScale.apply(x, s)


class _Add(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b):
        n, c, h, w = a.shape
        if a.shape != b.shape:
            raise ValueError('add_bf16: shape mismatch')
        y = new_bf16(n, c, h, w, a.device)
        call('ssg_add_bf16', ptr(a), _ld(a), ptr(b), _ld(b), n * h * w, c, ptr(y), c, stream_ptr())
        return y

    @staticmethod
    def backward(ctx, g):
        return g, g


def add(a, b):
    _lib.require_gpu(a)
    return _Add.apply(a, b)
