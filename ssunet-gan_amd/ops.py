"""Autograd ops over the gfx950 C-ABI (include/ssunet_hip.h).  Every op here launches a
hand-written HIP kernel; none falls back to a stock torch compute op, and all of them raise
if the tensors are not on a HIP device or the shared library is not built.

Tensor convention ("NHWC-with-stride"): an activation is a torch tensor of logical shape
[N, C, H, W] whose memory is [N, H, W, ld] with ld = C rounded up to 4 (pad channels are
zero).  For C % 4 == 0 this is exactly torch.channels_last, so tensors stay ordinary torch
tensors at the nn.Module boundary while the kernels see coalesced channel rows.
"""
import ctypes as C
import math

import torch
import torch.distributed as dist

from . import _lib
from ._lib import ACT_LRELU, ACT_NONE, ACT_RELU, ConvDesc, WgradDesc, call, ptr, stream_ptr

__all__ = ['conv2d', 'batch_norm_act', 'max_pool2x2', 'max_pool2x2_skip', 'max_unpool2x2', 'upsample2x_bilinear', 'upsample2x_nearest',
           'spade_modulate', 'adaptive_avgpool_flat', 'linear', 'seg_loss', 'bce_with_logits_const', 'nan_to_zero_',
           'to_nhwc', 'new_nhwc', 'bump_weight_epoch', 'dwconv2d', 'swish', 'sigmoid', 'gaussian', 'mul', 'global_avgpool',
           'channel_scale', 'spectral_norm_weight', 'conv2d_sn']


def pad4(c):
    return (c + 3) // 4 * 4


# ----------------------------------------------------------------------------- layout helpers
def new_nhwc(n, c, h, w, device, ld=None, zero=False):
    """Fresh [n, c, h, w] tensor over an [n, h, w, ld] buffer.  Built with set_() on the buffer's
    storage so the result is a base tensor, not an autograd view (callers may modify it in place,
    e.g. train_seg_gan.py:190), while the storage still covers the trailing pad lanes."""
    ld = pad4(c) if ld is None else ld
    buf = (torch.zeros if zero else torch.empty)(n * h * w * ld, device=device, dtype=torch.float32)
    return torch.empty(0, device=device, dtype=torch.float32).set_(
        buf.untyped_storage(), 0, (n, c, h, w), (h * w * ld, 1, w * ld, ld))


def nhwc_ld(x):
    """Pixel stride ld if x is NHWC-with-stride (and usable by the kernels), else None."""
    if x.dim() != 4 or x.dtype != torch.float32:
        return None
    n, c, h, w = x.shape
    s = x.stride()
    if w > 1:
        ld = s[3]
    elif h > 1:
        ld = s[2]
    elif n > 1:
        ld = s[0]
    else:
        ld = pad4(c)
    if ld % 4 or ld < c or (c > 1 and s[1] != 1):
        return None
    if (w > 1 and s[3] != ld) or (h > 1 and s[2] != w * ld) or (n > 1 and s[0] != h * w * ld):
        return None
    if c % 4 and ld != pad4(c):          # padded tensors must be our own (zero pad lanes)
        return None
    if x.data_ptr() % 16:
        return None
    return ld


def to_nhwc(x):
    """Return x in NHWC-with-stride form (no copy if it already is)."""
    _lib.require_gpu(x)
    if nhwc_ld(x) is not None:
        return x
    if x.dtype != torch.float32:
        raise TypeError('ssunet-gan_amd ops are fp32; got %s' % x.dtype)
    src = x.detach()
    if not src.is_contiguous():
        src = src.contiguous()
    n, c, h, w = src.shape
    out = new_nhwc(n, c, h, w, x.device)
    call('ssg_nchw_to_nhwc_f32', ptr(src), n, c, h, w, ptr(out), pad4(c), stream_ptr())
    return out


class _ToNHWC(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        return to_nhwc(x)

    @staticmethod
    def backward(ctx, g):
        return g


def as_nhwc(x):
    """Differentiable layout normalisation used at module boundaries."""
    if nhwc_ld(x) is not None:
        return x
    return _ToNHWC.apply(x) if x.requires_grad else to_nhwc(x)


def _ld(x):
    ld = nhwc_ld(x)
    if ld is None:
        raise _lib.HipLibraryError('internal: tensor is not NHWC-with-stride: shape %s stride %s' % (tuple(x.shape), x.stride()))
    return ld


def _ws(nbytes, device):
    return torch.empty((max(int(nbytes), 16) + 15) // 16 * 2, dtype=torch.float64, device=device)


# ----------------------------------------------------------------------------- weight packing cache
_WEIGHT_EPOCH = [0]


def bump_weight_epoch():
    """Called by the fused optimizer (which writes parameters through raw pointers, invisible
    to tensor version counters) to invalidate packed-weight caches."""
    _WEIGHT_EPOCH[0] += 1


def _pad4(pad):
    """int (symmetric) or (top, bottom, left, right) -> (pt, pb, pl, pr)."""
    if isinstance(pad, (tuple, list)):
        pt, pb, pl, pr = [int(v) for v in pad]
        return pt, pb, pl, pr
    return int(pad), int(pad), int(pad), int(pad)


def _taps_fwd(kh, kw, pad):
    pt, _, pl, _ = _pad4(pad)
    return [(ky, kx, ky - pt, kx - pl) for ky in range(kh) for kx in range(kw)]


def _pack(weight, transpose, taps, cred_pad, c1_for_mode, sigma=None):
    """Pack an OIHW weight for the given tap list; returns (tensor[R, Kp], Kp, kmode).  `sigma` (device scalar): pack
    weight / sigma instead (spectral norm; sigma changes every training forward, so this pack is not cached)."""
    o, i, kh, kw = weight.shape
    nt = len(taps)
    kmode = 0 if (cred_pad % 16 == 0 and c1_for_mode % 16 == 0) else 1
    kp = nt * cred_pad if kmode == 0 else (nt * cred_pad + 15) // 16 * 16
    rows = i if transpose else o
    if sigma is not None:
        out = torch.empty((rows, kp), device=weight.device, dtype=torch.float32)
        ky = (C.c_int * nt)(*[t[0] for t in taps]); kx = (C.c_int * nt)(*[t[1] for t in taps])
        wc = weight.detach().contiguous()
        call('ssg_pack_weights_scaled_f32', ptr(wc), o, i, kh, kw, int(transpose), nt, ky, kx, kmode, cred_pad, kp, ptr(sigma), ptr(out), stream_ptr())
        return out, kp, kmode
    # The cache lives ON the parameter object (never keyed by address alone: a freed tensor's
    # address can be reused by another weight).  Entries are valid for one (storage address,
    # autograd version, optimizer epoch) of that parameter.
    stamp = (weight.data_ptr(), weight._version, _WEIGHT_EPOCH[0])
    cache = weight.__dict__.get('_ssg_pack')
    if cache is None or cache[0] != stamp:
        cache = (stamp, {})
        try:
            weight._ssg_pack = cache
        except Exception:                        # exotic tensor subclasses: just do not cache
            pass
    key = (transpose, tuple(taps), cred_pad, kmode)
    hit = cache[1].get(key)
    if hit is not None:
        return hit, kp, kmode
    out = torch.empty((rows, kp), device=weight.device, dtype=torch.float32)
    ky = (C.c_int * nt)(*[t[0] for t in taps])
    kx = (C.c_int * nt)(*[t[1] for t in taps])
    call('ssg_pack_weights_f32', ptr(weight), o, i, kh, kw, int(transpose), nt, ky, kx, kmode, cred_pad, kp, ptr(out), stream_ptr())
    cache[1][key] = out
    return out, kp, kmode


def _fill_taps(desc, taps):
    desc.ntaps = len(taps)
    for t, (_, _, dy, dx) in enumerate(taps):
        desc.dy[t] = dy
        desc.dx[t] = dx


# Optional per-launch timing (bench.py): a list that receives (kernel label, algorithmic FLOPs,
# start event, end event) for every MFMA conv launch, recorded on the launch stream.
PROFILE = None
PROFILE_SHAPES = False      # debug: append the launch geometry to the label
# bench.py's north-star sub-metrics (BASELINE.json: "MFMA roofline on the 3x3 encoder convs", "achieved HBM GB/s on the
# memory-bound upsample/BN stages"): while PROFILE is a list, PROFILE_HBM receives (stage, algorithmic bytes per SURVEY.md 8(d),
# start event, end event) for every launch group of the memory-bound stages, and every MFMA record carries the ROLE the model
# code set around the launch (archs.UNet_R_SS_v2 tags the forward 3x3 convs of its six encoder BasicBlocks 'encoder_3x3').
PROFILE_HBM = None
PROFILE_COMM = None         # (kind, bytes, start event, end event) per small collective on the compute stream (sync-BN statistics)
_ROLE = [None]


def _timed_all_reduce(kind, t, group):
    """dist.all_reduce(SUM) of a sync-BN statistics vector, with HIP events on the launch stream while bench.py profiles: the
    collective is stream-ordered on the compute stream's critical path, so event time = what it costs the step."""
    if PROFILE_COMM is None:
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
        return
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record(torch.cuda.current_stream())
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    e1.record(torch.cuda.current_stream())
    PROFILE_COMM.append((kind, t.numel() * t.element_size(), e0, e1))


class role(object):
    """with ops.role('encoder_3x3'): ...   -- labels the forward conv launches issued inside (profiling only)."""

    def __init__(self, name):
        self.name = name

    def __enter__(self):
        self.prev = _ROLE[0]; _ROLE[0] = self.name

    def __exit__(self, *a):
        _ROLE[0] = self.prev


class _hbm(object):
    """HIP events on the launch stream around one memory-bound stage (no-op unless bench.py profiles)."""

    def __init__(self, stage, nbytes):
        self.rec = PROFILE_HBM is not None
        if self.rec:
            self.stage, self.nbytes = stage, float(nbytes)
            self.e0 = torch.cuda.Event(enable_timing=True); self.e1 = torch.cuda.Event(enable_timing=True)

    def __enter__(self):
        if self.rec:
            self.e0.record(torch.cuda.current_stream())

    def __exit__(self, *a):
        if self.rec:
            self.e1.record(torch.cuda.current_stream())
            PROFILE_HBM.append((self.stage, self.nbytes, self.e0, self.e1))


_CONV_LABELS = {0: 'conv_igemm_kernel<128,128>', 1: 'conv_igemm_kernel<256,64>', 2: 'conv_igemm_kernel<256,32>',
                20: 'conv_igemm_dma_kernel<128,128>', 21: 'conv_igemm_dma_kernel<256,64>', 22: 'conv_igemm_dma_kernel<128,64>',
                30: 'conv_igemm_halo_kernel<128,128>', 31: 'conv_igemm_halo_kernel<256,64>', 32: 'conv_igemm_halo_kernel<128,64>',
                33: 'conv_igemm_halo16_kernel<128,128>', 34: 'conv_igemm_halo16_kernel<128,64>',
                10: 'thin_small_cout_kernel',
                12: 'thin4_cin_kernel', 13: 'thin4_cout_kernel', 14: 'tiny4_kernel', 15: 'thin32_cin_kernel', 16: 'conv1x1_k64_kernel'}
_WGRAD_LABELS = {0: 'wgrad_kernel<128,128>', 1: 'wgrad_kernel<128,64>', 2: 'wgrad_kernel<128,32>',
                 20: 'wgrad_dma_kernel<128,128>', 21: 'wgrad_dma_kernel<128,64>',
                 30: 'wgrad_halo_kernel<32,128>', 31: 'wgrad_halo_kernel<64,64>',
                 40: 'wgrad_halo_x3_kernel<32,128>', 41: 'wgrad_halo_x3_kernel<64,64>', 60: 'wgrad_k32_kernel<64,64>',
                 50: 'wgrad_dma_x3_kernel<128,128>', 51: 'wgrad_dma_x3_kernel<128,64>',
                 15: 'wgrad4_kernel<thin_cout>', 16: 'wgrad4_kernel<thin_cin>', 17: 'wgrad_tiny4_kernel', 18: 'wgrad32_cin_kernel'}


class _Timed(object):
    def __init__(self, label, flops, tag=None):
        self.rec = PROFILE is not None
        if self.rec:
            self.label, self.flops, self.tag = label, flops, tag
            self.e0 = torch.cuda.Event(enable_timing=True); self.e1 = torch.cuda.Event(enable_timing=True)

    def __enter__(self):
        if self.rec:
            self.e0.record(torch.cuda.current_stream())

    def __exit__(self, *a):
        if self.rec:
            self.e1.record(torch.cuda.current_stream())
            PROFILE.append((self.label, self.flops, self.e0, self.e1, self.tag))


_DECLINED = object()      # _conv_launch: the library has no kernel with the fused input transform for this launch (nothing was launched)


def _conv_launch(x1, x2, wpk, kp, kmode, row0, cout, bias, res, act, slope, taps, n, h, w, gh, gw, oh, ow,
                 in_s, out_s, out_oy, out_ox, out, want_bn=False, tag=None, parity_merge=False, in_affine=None, bwd_stats=None):
    """Returns None, or -- with want_bn and a kernel that has the statistics epilogue -- the fp64 tensor [rows, 2, cout] of
    per-tile (sum, sum of squares) of the conv output (ssg_conv_desc.bnpart).  parity_merge: the nine taps are the four parity
    classes of a 3x3 stride-2 input gradient (ssg_conv_desc.parity_merge); returns False, with nothing launched, when the library
    has no merged kernel for this shape.  in_affine = (scale[C1], shift[C1], act, slope): the launch convolves act(x1 * scale +
    shift) (ssg_conv_desc.in_scale: a batch-norm apply that is never materialised); returns _DECLINED, with nothing launched, when
    the kernel this shape maps to has no such transform."""
    d = ConvDesc()
    d.in1 = x1.data_ptr(); d.C1 = pad4(x1.shape[1]); d.ld1 = _ld(x1)
    if x2 is not None:
        d.in2 = x2.data_ptr(); d.C2 = pad4(x2.shape[1]); d.ld2 = _ld(x2)
    else:
        d.in2 = None; d.C2 = 0; d.ld2 = 0
    d.N, d.H, d.W = n, h, w
    d.w = wpk.data_ptr() + row0 * kp * 4; d.Kp = kp; d.kmode = kmode
    d.bias = bias.data_ptr() if bias is not None else None
    if res is not None:
        d.res = res.data_ptr(); d.ldr = _ld(res)
    else:
        d.res = None; d.ldr = 0
    d.out = out.data_ptr(); d.Cout = cout; d.ldo = _ld(out)
    d.GH, d.GW, d.OH, d.OW = gh, gw, oh, ow
    d.in_sy = d.in_sx = in_s
    d.out_sy = d.out_sx = out_s
    d.out_oy, d.out_ox = out_oy, out_ox
    _fill_taps(d, taps)
    d.act = act; d.slope = slope
    d.bnpart = None
    d.ws = None; d.ws_bytes = 0
    d.w_split = None
    d.parity_merge = 1 if parity_merge else 0
    d.in_scale = None; d.in_shift = None; d.in_act = ACT_NONE; d.in_slope = 0.0
    d.bwd_x = None; d.bwd_ldx = 0; d.bwd_scale = None; d.bwd_shift = None; d.bwd_mean = None; d.bwd_act = ACT_NONE; d.bwd_slope = 0.0
    split = None
    if parity_merge and not (MFMA_SPLIT and kmode == 0 and call('ssg_conv2d_split_bn', C.byref(d)) == 64):
        return False
    if MFMA_SPLIT and kmode == 0:      # decided first: the split-operand kernel has its own tile geometry (bnpart rows)
        bn = call('ssg_conv2d_split_bn', C.byref(d))
        if bn:
            split = _split_pack(wpk, row0, cout, kp, 1064 if bn == 2064 else bn)      # 2064: the 16-row tile reads the 64-column k32 pack
            d.w_split = split.data_ptr()
    if in_affine is not None:
        d.in_scale = in_affine[0].data_ptr(); d.in_shift = in_affine[1].data_ptr(); d.in_act = int(in_affine[2]); d.in_slope = float(in_affine[3])
        if split is None or not call('ssg_conv2d_in_affine_ok', C.byref(d)):
            return _DECLINED
        bn = call('ssg_conv2d_split_bn', C.byref(d))      # with the transform a 16-row launch becomes a 4-row one (same pack)
    if bwd_stats is not None:
        # (x, stats rows [mean, invstd, scale, shift], act, slope): the launch produces d(act(bn(x))), masks it and writes the batch-norm
        # backward sums as per-tile rows (ssg_conv_desc.bwd_x); _DECLINED, with nothing launched, where the kernel has no such epilogue
        bx, bst, bact, bslope = bwd_stats
        d.bwd_x = bx.data_ptr(); d.bwd_ldx = _ld(bx); d.bwd_scale = bst[2].data_ptr(); d.bwd_shift = bst[3].data_ptr()
        d.bwd_mean = bst[0].data_ptr(); d.bwd_act = int(bact); d.bwd_slope = float(bslope)
        if split is None or not call('ssg_conv2d_bwd_stats_ok', C.byref(d)):
            return _DECLINED
        want_bn = True
    part = None
    if want_bn and BN_EPILOGUE:
        rows = call('ssg_conv2d_bnpart_rows', C.byref(d))
        if rows > 0:
            part = torch.empty((rows, 2, cout), dtype=torch.float64, device=out.device)
            d.bnpart = part.data_ptr()
    if bwd_stats is not None and part is None:
        return _DECLINED
    ws = None
    if part is None and split is None:  # split-K (small pixel grids with a long reduction): the kernel needs a workspace
        need = call('ssg_conv2d_workspace_bytes', C.byref(d))
        if need > 0:
            ws = _ws(need, out.device)
            d.ws = ws.data_ptr(); d.ws_bytes = need
    cred = x1.shape[1] + (x2.shape[1] if x2 is not None else 0)
    label = None
    if PROFILE is not None:
        label = _CONV_LABELS.get(call('ssg_conv2d_kernel_id', C.byref(d)), '?') + ('+splitk' if ws is not None else '')
        if parity_merge:
            label = 'conv_igemm_halo_x3_kernel<128,64,4,1,true>'
        elif split is not None and bn >= 1000:
            label = {1128: 'conv_halo_k32_kernel<8,128>', 1064: 'conv_halo_k32_kernel<4,64>', 2064: 'conv_halo_k32_kernel<16,64>',
                     1016: 'conv_halo_k32_kernel<8,16>', 1032: 'conv_halo_k32_kernel<8,32>'}[bn]
        elif split is not None:
            if 'halo' in label:
                label = label.replace('conv_igemm_halo_kernel', 'conv_igemm_halo_x3_kernel').replace('<256,64>', '<128,64>')
            else:
                label = 'conv_igemm_dma_x3_kernel<%d>' % bn
        if PROFILE_SHAPES:
            label += ' n%d %dx%d cin%d cout%d taps%d s%d/%d' % (n, gh, gw, cred, cout, len(taps), in_s, out_s)
    with _Timed(label, 2.0 * n * gh * gw * cout * cred * len(taps), tag):
        call('ssg_conv2d_f32', C.byref(d), stream_ptr())
    return part


# SSG_BN_EPILOGUE=0: batch-norm statistics from their own pass over the conv output instead of the conv epilogue (A/B switch)
import os as _os
BN_EPILOGUE = _os.environ.get('SSG_BN_EPILOGUE', '1') != '0'
# The dense 3x3 unit-stride convs and their input gradients multiply on the bf16 matrix pipe with every fp32 operand split into
# three bf16 terms (csrc/conv_igemm_halo_x3.hip: the error against fp64 is that of the fp32-MFMA kernel, tests/test_split_gpu.py;
# 16/6 of the fp32 MFMA rate).  SSG_MFMA_SPLIT=0 keeps them on v_mfma_f32_32x32x2_f32 (conv_igemm_halo.hip).
MFMA_SPLIT = _os.environ.get('SSG_MFMA_SPLIT', '1') == '1'
# SSG_BN_FUSE_INPUT=1: relu(bn1(conv1(x))) of a residual block is applied on conv2's input (and on the x operand of conv2's weight
# gradient) instead of being written out -- same bits (tests/test_blocks_gpu.py), one activation less to keep, 1.0 ms less in the
# batch-norm passes of the 16 x 512^2 step and 1.0-2.5 ms MORE in the k32 kernels that take the transform (issue-bound: the extra
# vector work on the operand path is not hidden), same-box A/B 195.5-195.9 vs 195.8-197.4 ms.  Off by default.
BN_FUSE_INPUT = _os.environ.get('SSG_BN_FUSE_INPUT', '0') == '1'
# SSG_PARITY_MERGE=0: the input gradient of a 3x3 stride-2 conv as four launches (one per output parity class) again
PARITY_MERGE = _os.environ.get('SSG_PARITY_MERGE', '1') != '0'


def _split_pack(wpk, row0, rows, kp, bn):
    """bf16x3 split of rows [row0, row0 + rows) of a packed fp32 weight matrix for column tile `bn` (cached on the packed
    tensor, which itself lives in the parameter's pack cache: both go stale together)."""
    cache = wpk.__dict__.get('_ssg_split')
    if cache is None:
        cache = {}
        try:
            wpk._ssg_split = cache
        except Exception:
            pass
    key = (row0, rows, bn)
    hit = cache.get(key)
    if hit is None:
        nbytes = call('ssg_pack_weights_split_bytes', rows, kp, bn)
        hit = torch.empty(nbytes // 2, dtype=torch.int16, device=wpk.device)
        call('ssg_pack_weights_split_bf16x3', wpk.data_ptr() + row0 * kp * 4, rows, kp, bn, ptr(hit), stream_ptr())
        cache[key] = hit
    return hit


def _out_size(h, k, s, p):
    return (h + 2 * p - k) // s + 1


def _out_hw(h, w, kh, kw, s, pad):
    pt, pb, pl, pr = _pad4(pad)
    return (h + pt + pb - kh) // s + 1, (w + pl + pr - kw) // s + 1


def _conv_fwd_impl(x1, x2, weight, bias, stride, pad, act, slope, res=None, out=None, wscale=None, want_bn=False, in_affine=None):
    """Returns y, or (y, part) with want_bn: part = per-tile batch-norm partial sums of y from the conv epilogue, or None when
    the kernel this shape maps to has none (the batch norm then runs its own statistics pass).  in_affine: see _conv_launch; the
    result is None (nothing launched) when the library declines."""
    o, i, kh, kw = weight.shape
    n, c1, h, w = x1.shape
    c2 = x2.shape[1] if x2 is not None else 0
    if x2 is not None and c1 % 4:
        raise ValueError('two-input conv needs C1 %% 4 == 0 (got %d)' % c1)
    if c1 + c2 != i:
        raise ValueError('conv: input channels %d+%d != weight in-channels %d' % (c1, c2, i))
    cred_pad = pad4(c1) + pad4(c2)
    taps = _taps_fwd(kh, kw, pad)
    wpk, kp, kmode = _pack(weight, 0, taps, cred_pad, pad4(c1), sigma=wscale)
    oh, ow = _out_hw(h, w, kh, kw, stride, pad)
    if out is None:
        out = new_nhwc(n, o, oh, ow, x1.device)
    part = _conv_launch(x1, x2, wpk, kp, kmode, 0, o, bias, res, act, slope, taps, n, h, w, oh, ow, oh, ow, stride, 1, 0, 0, out,
                        want_bn=want_bn and res is None and act == ACT_NONE,
                        tag=_ROLE[0] if (kh == 3 and kw == 3 and _ROLE[0] is not None) else None, in_affine=in_affine)
    if part is _DECLINED:
        return None
    return (out, part) if want_bn else out


def _conv_dgrad_impl(dy, weight, stride, pad, h, w, c_lo, c_hi, res=None, wscale=None, bwd_stats=None):
    """Input gradient for input channels [c_lo, c_hi) -> NHWC tensor [N, c_hi-c_lo, h, w].
    `res` (same shape) is added in the epilogue: gradient accumulation without an extra pass.
    bwd_stats = (x, stats, act, slope): the gradient is that of act(bn(x)); returns (g, part) -- the MASKED gradient and the per-tile rows
    (sum g, sum g * (x - mean)) of the batch-norm backward from the epilogue (ssg_conv_desc.bwd_x) -- or None, with nothing launched,
    where the kernel this launch maps to has no such epilogue."""
    o, i, kh, kw = weight.shape
    n, _, oh, ow = dy.shape
    cred_pad = pad4(o)
    pt, _, pl, _ = _pad4(pad)
    dx = new_nhwc(n, c_hi - c_lo, h, w, dy.device)
    if stride == 1:
        taps = [(ky, kx, pt - ky, pl - kx) for ky in range(kh) for kx in range(kw)]
        wpk, kp, kmode = _pack(weight, 1, taps, cred_pad, cred_pad, sigma=wscale)
        part = _conv_launch(dy, None, wpk, kp, kmode, c_lo, c_hi - c_lo, None, res, ACT_NONE, 0.0, taps, n, oh, ow, h, w, h, w, 1, 1, 0, 0, dx,
                            bwd_stats=bwd_stats)
        if bwd_stats is not None:
            return None if part is _DECLINED else (dx, part)
        return dx
    if bwd_stats is not None:
        return None
    if res is not None:
        raise NotImplementedError('strided dgrad with fused accumulation')
    s = stride
    classes = []
    for py in range(s):
        for px in range(s):
            taps = [(ky, kx, (py + pt - ky) // s, (px + pl - kx) // s) for ky in range(kh) for kx in range(kw)
                    if (py + pt - ky) % s == 0 and (px + pl - kx) % s == 0]
            classes.append((py, px, taps))
    # eligibility of the merged launch is decided BEFORE its 9-tap pack is built: the library declines dy narrower
    # than 17 pixels, channel counts that are not whole 64-column tiles and tensors beyond 32-bit byte offsets
    # (ssg_conv_halo_x3_parity_ok); with spectral norm a declined pack would be an uncached launch + allocation per backward
    if (PARITY_MERGE and s == 2 and kh == 3 and kw == 3 and (pt, pl) == (1, 1) and MFMA_SPLIT and ow >= 17
            and (c_hi - c_lo) % 64 == 0 and dy.numel() * 4 <= 0xfffffff0 and _os.environ.get('SSG_X3_PARITY', '1') != '0'):
        # the four classes read the same 2x2 neighbourhood of dy: one launch, nine tap steps, four accumulator sets
        # (conv_igemm_halo_x3_kernel<..., PARITY>); falls through to the per-class launches where the library declines
        merged = [t for _, _, taps in classes for t in taps]
        wpk, kp, kmode = _pack(weight, 1, merged, cred_pad, cred_pad, sigma=wscale)
        if _conv_launch(dy, None, wpk, kp, kmode, c_lo, c_hi - c_lo, None, None, ACT_NONE, 0.0, merged, n, oh, ow,
                        (h + 1) // 2, (w + 1) // 2, h, w, 1, s, 0, 0, dx, parity_merge=True) is not False:
            return dx
    if any(len(t) == 0 for _, _, t in classes):
        dx.zero_()
    for py, px, taps in classes:
        gh, gw = (h - py + s - 1) // s, (w - px + s - 1) // s
        if not taps or gh <= 0 or gw <= 0:
            continue
        wpk, kp, kmode = _pack(weight, 1, taps, cred_pad, cred_pad, sigma=wscale)
        _conv_launch(dy, None, wpk, kp, kmode, c_lo, c_hi - c_lo, None, None, ACT_NONE, 0.0, taps, n, oh, ow, gh, gw, h, w, 1, s, py, px, dx)
    return dx


def _conv_wgrad_impl(x1, x2, dy, weight_shape, stride, pad, in_affine=None):
    """in_affine = (scale, shift, act, slope): the x operand is act(x1 * scale + shift) (ssg_wgrad_desc.in_scale); returns None,
    with nothing launched, when the kernel this shape maps to has no such transform."""
    o, i, kh, kw = weight_shape
    n, c1, h, w = x1.shape
    _, _, oh, ow = dy.shape
    d = WgradDesc()
    d.in1 = x1.data_ptr(); d.C1 = pad4(c1); d.ld1 = _ld(x1)
    if x2 is not None:
        d.in2 = x2.data_ptr(); d.C2 = pad4(x2.shape[1]); d.ld2 = _ld(x2)
    else:
        d.in2 = None; d.C2 = 0; d.ld2 = 0
    d.N, d.H, d.W = n, h, w
    d.dout = dy.data_ptr(); d.Cout = o; d.ldd = _ld(dy); d.GH, d.GW = oh, ow
    d.in_sy = d.in_sx = stride
    taps = _taps_fwd(kh, kw, pad)
    _fill_taps(d, taps)
    for t, (ky, kx, _, _) in enumerate(taps):
        d.ky[t] = ky; d.kx[t] = kx
    d.KH, d.KW, d.Cin_real = kh, kw, i
    dw = torch.empty((o, i, kh, kw), device=dy.device, dtype=torch.float32)
    d.dw_oihw = dw.data_ptr()
    d.ws = None; d.ws_bytes = 0
    d.flags = 1 if MFMA_SPLIT else 0
    d.in_scale = None; d.in_shift = None; d.in_act = ACT_NONE; d.in_slope = 0.0
    if in_affine is not None:
        d.in_scale = in_affine[0].data_ptr(); d.in_shift = in_affine[1].data_ptr(); d.in_act = int(in_affine[2]); d.in_slope = float(in_affine[3])
        if not call('ssg_conv2d_wgrad_in_affine_ok', C.byref(d)):
            return None
    nbytes = call('ssg_conv2d_wgrad_workspace_bytes', C.byref(d))
    ws = _ws(nbytes, dy.device)
    d.ws = ws.data_ptr(); d.ws_bytes = ws.numel() * 8
    label = None
    if PROFILE is not None:
        label = _WGRAD_LABELS.get(call('ssg_conv2d_wgrad_kernel_id', C.byref(d)), '?')
        if PROFILE_SHAPES:
            label += ' n%d %dx%d cin%d cout%d k%d s%d' % (n, oh, ow, i, o, kh, stride)
    with _Timed(label, 2.0 * n * oh * ow * o * i * kh * kw):
        call('ssg_conv2d_wgrad_f32', C.byref(d), stream_ptr())
    return dw


def _channel_sum(x, c):
    n, _, h, w = x.shape
    p = n * h * w
    out = torch.empty(c, device=x.device, dtype=torch.float32)
    ws = _ws(call('ssg_bn_workspace_bytes', p, c), x.device)
    call('ssg_channel_sum_f32', ptr(x), p, c, _ld(x), ptr(out), ptr(ws), stream_ptr())
    return out


def _act_bwd(y, dy, act, slope):
    n, c, h, w = y.shape
    dx = new_nhwc(n, c, h, w, y.device)
    call('ssg_act_bwd_f32', ptr(y), _ld(y), ptr(dy), _ld(dy), n * h * w, pad4(c), act, slope, ptr(dx), _ld(dx), stream_ptr())
    return dx


class _Conv2d(torch.autograd.Function):
    """F.conv2d (+bias, +activation, optional second input = channel concat) on MFMA."""

    @staticmethod
    def forward(ctx, x1, x2, weight, bias, stride, pad, act, slope, res=None, want_bn=False):
        ctx.set_materialize_grads(False)       # no zero tensor for the (non-differentiable) statistics output's gradient
        x1 = to_nhwc(x1)
        x2 = to_nhwc(x2) if x2 is not None else None
        res = to_nhwc(res) if res is not None else None
        y, part = _conv_fwd_impl(x1, x2, weight, bias, stride, pad, act, slope, res=res, want_bn=True) if want_bn else \
            (_conv_fwd_impl(x1, x2, weight, bias, stride, pad, act, slope, res=res), None)
        ctx.cfg = (stride, pad, act, slope)
        ctx.save_for_backward(x1, x2, weight, y if act != ACT_NONE else None)
        ctx.has_bias = bias is not None
        ctx.has_res = res is not None
        if part is None:
            part = torch.empty(0, dtype=torch.float64, device=y.device)
        ctx.mark_non_differentiable(part)
        return y, part

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, dy, _=None):
        if dy is None:
            return (None,) * 10
        x1, x2, weight, y = ctx.saved_tensors
        stride, pad, act, slope = ctx.cfg
        dy = to_nhwc(dy)
        if act != ACT_NONE:
            dy = _act_bwd(y, dy, act, slope)
        n, c1, h, w = x1.shape
        dx1 = dx2 = dw = db = None
        if ctx.needs_input_grad[0]:
            dx1 = _conv_dgrad_impl(dy, weight, stride, pad, h, w, 0, c1)
        if x2 is not None and ctx.needs_input_grad[1]:
            dx2 = _conv_dgrad_impl(dy, weight, stride, pad, h, w, c1, c1 + x2.shape[1])
        if ctx.needs_input_grad[2]:
            dw = _conv_wgrad_impl(x1, x2, dy, weight.shape, stride, pad)
        if ctx.has_bias and ctx.needs_input_grad[3]:
            db = _channel_sum(dy, weight.shape[0])
        dres = dy if (ctx.has_res and ctx.needs_input_grad[8]) else None      # residual joins before the activation
        return dx1, dx2, dw, db, None, None, None, None, dres, None


def conv2d(x, weight, bias=None, stride=1, padding=0, act=ACT_NONE, slope=0.0, x2=None, res=None, bn_stats=False):
    """F.conv2d on the HIP kernels.  `padding`: int or (top, bottom, left, right) (TF-"same" static
    padding of the EfficientNet convs is asymmetric); `x2`: second tensor concatenated after x;
    `res`: tensor added in the epilogue before the activation (act(conv(x) + bias + res)).
    `bn_stats=True`: returns (y, part) -- `part` are the per-tile (sum, sum of squares) of y from the conv epilogue (empty when
    this shape's kernel has none); hand it to batch_norm_act(y, bn, stats_part=part) to skip its statistics pass."""
    _lib.require_gpu(x)
    pad = tuple(int(v) for v in padding) if isinstance(padding, (tuple, list)) else int(padding)
    y, part = _Conv2d.apply(x, x2, weight, bias, int(stride), pad, int(act), float(slope), res, bool(bn_stats))
    return (y, part) if bn_stats else y


# ----------------------------------------------------------------------------- linear (as 1x1 conv over a 1 x N "image")
class _Linear(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, act, slope):
        n, k = x.shape
        o = weight.shape[0]
        if k % 4 or x.stride(1) != 1 or x.stride(0) % 4 or x.data_ptr() % 16:
            x = _pad_rows(x)
        xi = x.as_strided((1, k, 1, n), (n * x.stride(0), 1, n * x.stride(0), x.stride(0)))
        if weight.is_contiguous() and weight.data_ptr() % 16 == 0 and k % 4 == 0:
            # skinny GEMM that streams the [O][K] parameter once (HBM-bound); no packed copy
            y = new_nhwc(1, o, 1, n, x.device)
            nbytes = call('ssg_linear_fwd_workspace_bytes', n, k, o)
            ws = _ws(nbytes, x.device)
            call('ssg_linear_fwd_f32', ptr(xi), n, k, x.stride(0), ptr(weight), o, ptr(bias) if bias is not None else None,
                 act, slope, ptr(y), _ld(y), ptr(ws), nbytes, stream_ptr())
        else:
            y = _conv_fwd_impl(xi, None, weight.view(o, k, 1, 1), bias, 1, 0, act, slope)      # [1, o, 1, n]
        ld = _ld(y)
        y2 = torch.empty(0, device=y.device, dtype=torch.float32).set_(y.untyped_storage(), y.storage_offset(), (n, o), (ld, 1))
        ctx.save_for_backward(xi, weight, y if act != ACT_NONE else None)
        ctx.cfg = (act, slope, n, k, o, bias is not None)
        return y2

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, dy2):
        xi, weight, y = ctx.saved_tensors
        act, slope, n, k, o, has_bias = ctx.cfg
        if dy2.stride(1) != 1 or dy2.stride(0) % 4 or dy2.stride(0) < pad4(o) or dy2.data_ptr() % 16 or (o % 4 and dy2.stride(0) != pad4(o)):
            dy2 = _pad_rows(dy2)
        ldd = dy2.stride(0)
        dy = dy2.as_strided((1, o, 1, n), (n * ldd, 1, n * ldd, ldd))
        if act != ACT_NONE:
            dy = _act_bwd(y, dy, act, slope)
        w4 = weight.view(o, k, 1, 1)
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            d = _conv_dgrad_impl(dy, w4, 1, 0, 1, n, 0, k)
            dx = d.as_strided((n, k), (_ld(d), 1))
        if ctx.needs_input_grad[1]:
            if k % 4 == 0 and xi.data_ptr() % 16 == 0 and xi.stride(3) % 4 == 0:
                dw = torch.empty((o, k), device=dy.device, dtype=torch.float32)
                call('ssg_linear_wgrad_f32', ptr(xi), n, k, xi.stride(3), ptr(dy), o, dy.stride(3), ptr(dw), stream_ptr())
            else:
                dw = _conv_wgrad_impl(xi, None, dy, (o, k, 1, 1), 1, 0).view(o, k)
        if has_bias and ctx.needs_input_grad[2]:
            db = _channel_sum(dy, o)
        return dx, dw, db, None, None


def _pad_rows(x):
    """Copy a [n, k] matrix into a zero-padded [n, pad4(k)] buffer (tiny tensors only)."""
    n, k = x.shape
    buf = torch.zeros((n, pad4(k)), device=x.device, dtype=torch.float32)
    xi = x.detach().contiguous().view(n, k, 1, 1)
    call('ssg_nchw_to_nhwc_f32', ptr(xi), n, k, 1, 1, ptr(buf), pad4(k), stream_ptr())
    return buf[:, :k]


def linear(x, weight, bias=None, act=ACT_NONE, slope=0.0):
    _lib.require_gpu(x)
    return _Linear.apply(x, weight, bias, int(act), float(slope))


# ----------------------------------------------------------------------------- batch norm (+residual, +activation)
def _synced(group):
    """True when batch-norm sums have to travel: a process group is attached and the job is distributed
    (world > 1, or SSG_DIST_FORCE=1 -- dp.FORCE -- which sends a single rank's collectives through the backend too)."""
    if group is None or not (dist.is_available() and dist.is_initialized()):
        return False
    from . import dp
    return dist.get_world_size(group) > 1 or dp.FORCE


# bumped whenever a training-mode batch norm rewrites running statistics through raw pointers (tensor version counters do
# not see that): part of the stamp of every eval-mode BN-fold cache (archs.BasicBlock._folded)
_STATS_EPOCH = [0]


BN_FUSED_FINALIZE = _os.environ.get('SSG_BN_FUSED_FINALIZE', '1') != '0'


def bn_fin(weight, bias, eps, momentum, var_mode, running_mean, running_var, stats):
    """ctypes `ssg_bn_fin` over the given tensors (stats = [mean, invstd, scale, shift] rows)."""
    f = _lib.BnFin()
    f.weight = weight.data_ptr() if weight is not None else None
    f.bias = bias.data_ptr() if bias is not None else None
    f.eps = eps; f.momentum = momentum; f.var_mode = var_mode
    f.running_mean = running_mean.data_ptr() if running_mean is not None else None
    f.running_var = running_var.data_ptr() if running_var is not None else None
    f.mean = stats[0].data_ptr(); f.invstd = stats[1].data_ptr(); f.scale = stats[2].data_ptr(); f.shift = stats[3].data_ptr()
    return f


def _bn_fwd_impl(x, weight, bias, running_mean, running_var, res, eps, momentum, act, slope, var_mode, group, part=None, apply=True):
    """Timed wrapper (bench.py `hbm_stages`): algorithmic bytes per SURVEY.md 8(d) -- 2 reads + 1 write of the tensor, one read
    less when the statistics rode the producing conv's epilogue, one more for a residual."""
    n, c, h, w = x.shape
    reads = (1 if (part is not None and part.numel() > 0) else 2) + (1 if res is not None else 0)
    if not apply:                 # statistics only (the consumer conv applies scale / shift / act on its input): no pass over x at all
        reads -= 1
    with _hbm('bn_fwd', 4.0 * n * h * w * c * (reads + (1 if apply else 0))):
        return _bn_fwd_body(x, weight, bias, running_mean, running_var, res, eps, momentum, act, slope, var_mode, group, part, apply)


def _bn_apply(x, stats, res, act, slope):
    """y = act(x * scale + shift (+ res)) with the finalized statistics rows [mean, invstd, scale, shift]."""
    n, c, h, w = x.shape
    y = new_nhwc(n, c, h, w, x.device)
    call('ssg_bn_apply_f32', ptr(x), n * h * w, c, _ld(x), ptr(stats[2]), ptr(stats[3]), ptr(res), _ld(res) if res is not None else 0,
         act, slope, ptr(y), _ld(y), stream_ptr())
    return y


def _bn_fwd_body(x, weight, bias, running_mean, running_var, res, eps, momentum, act, slope, var_mode, group, part=None, apply=True):
    """stats -> (all-reduce) -> finalize -> apply.  Returns (y, stats[4,C], count): `count` is None for a local batch norm
    and, when synchronised, the fp64[1] device tensor holding the all-reduced pixel count (ranks may hold unequal batches:
    the count travels with the sums instead of being assumed to be p * world)."""
    n, c, h, w = x.shape
    if c % 4:
        raise ValueError('batch_norm: C %% 4 != 0 unsupported (C=%d)' % c)
    p = n * h * w
    dev = x.device
    synced = _synced(group)
    stats = torch.empty((4, c), dtype=torch.float32, device=dev)      # mean, invstd, scale, shift
    have_part = part is not None and part.numel() > 0
    if have_part and (part.shape[-1] != c or part.numel() % (2 * c)):
        raise ValueError('batch_norm: statistics partials %s do not fit C=%d' % (tuple(part.shape), c))
    if not synced and BN_FUSED_FINALIZE:
        # local batch norm: the second reduce stage finishes the channel (one launch less; same bits as stats + finalize)
        fin = bn_fin(weight, bias, eps, momentum, var_mode, running_mean, running_var, stats)
        if have_part:
            rows = part.numel() // (2 * c)
            ws = _ws(call('ssg_bn_stats_from_partials_workspace_bytes', rows, c), dev)
            call('ssg_bn_stats_from_partials_finalize_f32', ptr(part), rows, c, float(p), C.byref(fin), ptr(ws), stream_ptr())
        else:
            ws = _ws(call('ssg_bn_workspace_bytes', p, c), dev)
            call('ssg_bn_stats_finalize_f32', ptr(x), p, c, _ld(x), C.byref(fin), ptr(ws), stream_ptr())
        sums = None
    else:
        sums = torch.empty(2 * c + 1, dtype=torch.float64, device=dev)
        if have_part:
            # (sum x, sum x^2) came out of the producing conv's epilogue, one row per tile: x is not read for its statistics
            rows = part.numel() // (2 * c)
            ws = _ws(call('ssg_bn_stats_from_partials_workspace_bytes', rows, c), dev)
            call('ssg_bn_stats_from_partials_f32', ptr(part), rows, c, ptr(sums), float(p) if synced else 0.0, ptr(ws), stream_ptr())
        else:
            ws = _ws(call('ssg_bn_workspace_bytes', p, c), dev)
            call('ssg_bn_stats_f32', ptr(x), p, c, _ld(x), ptr(sums), int(synced), ptr(ws), stream_ptr())
        if synced:
            _timed_all_reduce('sync_bn_fwd', sums, group)
        call('ssg_bn_finalize_f32', ptr(sums), 0.0 if synced else float(p), c, ptr(weight), ptr(bias), eps, momentum, var_mode,
             ptr(running_mean), ptr(running_var), ptr(stats[0]), ptr(stats[1]), ptr(stats[2]), ptr(stats[3]), stream_ptr())
    if running_mean is not None or running_var is not None:
        _STATS_EPOCH[0] += 1
    y = _bn_apply(x, stats, res, act, slope) if apply else None
    return y, stats, (sums[2 * c:] if synced else None)


def _bn_bwd_impl(x, y, dy, weight, stats, act, slope, group, count, want_dres, want_dx=True, had_res=True):
    """Timed wrapper (bench.py `hbm_stages`): algorithmic bytes per SURVEY.md 8(d) -- 3 reads + 1 write (x, dy, the mask source;
    dx), plus the dres write where the forward had a residual."""
    n, c, h, w = x.shape
    with _hbm('bn_bwd', 4.0 * n * h * w * c * (3 + (1 if want_dx else 0) + (1 if want_dres else 0))):
        return _bn_bwd_body(x, y, dy, weight, stats, act, slope, group, count, want_dres, want_dx, had_res)


def _bn_bwd_body(x, y, dy, weight, stats, act, slope, group, count, want_dres, want_dx=True, had_res=True):
    """Returns (dx, dres, dweight, dbias).  The activation mask is read from y, or -- when the forward had no
    residual (`had_res=False`) -- recomputed from x with the forward's (scale, shift): y is then not read.
    `count`: what _bn_fwd_impl returned (None = local batch norm)."""
    n, c, h, w = x.shape
    p = n * h * w
    dev = x.device
    if act == ACT_NONE or not had_res:
        y = None
    synced = count is not None
    ws = _ws(call('ssg_bn_workspace_bytes', p, c), dev)
    sums = torch.empty(2 * c + 1, dtype=torch.float64, device=dev)
    call('ssg_bn_bwd_reduce_f32', ptr(x), ptr(y), ptr(dy), p, c, _ld(x), _ld(y) if y is not None else 0, _ld(dy),
         ptr(stats[0]), ptr(stats[1]), ptr(stats[2]), ptr(stats[3]), act, slope, ptr(sums), int(synced), ptr(ws), stream_ptr())
    # local (un-reduced) sums are this rank's weight/bias gradients; data-parallel all-reduces them later
    local = sums.clone() if synced else sums
    if synced:
        _timed_all_reduce('sync_bn_bwd', sums, group)
    dwb = torch.empty((2, c), dtype=torch.float32, device=dev)
    dx = new_nhwc(n, c, h, w, dev) if want_dx else None
    dres = new_nhwc(n, c, h, w, dev) if want_dres else None
    call('ssg_bn_bwd_apply_f32', ptr(x), ptr(y), ptr(dy), p, c, _ld(x), _ld(y) if y is not None else 0, _ld(dy),
         ptr(stats[0]), ptr(stats[1]), ptr(weight), ptr(stats[2]), ptr(stats[3]), ptr(sums), 0.0 if synced else float(p), act, slope,
         ptr(dx), _ld(dx) if dx is not None else 0, ptr(dres), _ld(dres) if dres is not None else 0,
         ptr(dwb[0]), ptr(dwb[1]), stream_ptr())
    if synced:
        dwb = torch.stack([local[c:2 * c], local[:c]]).float()
    return dx, dres, dwb[0], dwb[1]


# SSG_BN_BWD_EPILOGUE=1: bn1's backward sums of a residual block from the epilogue of conv2's input gradient (ssg_conv_desc.bwd_x) instead of
# a reduce pass over (dy1, c1).  Measured (same box, twice on two boxes): bn_bwd 14.4 -> 12.7 ms per step, and the k32 launches that carry
# the epilogue give it back (<8,128> -1.4 %, <16,64> -3 % over all their launches): 191.9 / 191.8 vs 192.1 / 191.9 ms, 189.6 / 190.9 vs
# 189.9 / 190.5.  Off by default.
BN_BWD_EPILOGUE = _os.environ.get('SSG_BN_BWD_EPILOGUE', '0') == '1'


def _bn_bwd_from_partials(x, g, part, weight, stats, group, count):
    """Batch-norm backward whose two sums came out of the producing input-gradient launch (ssg_conv_desc.bwd_x): g is already masked, `part`
    holds per-tile rows (sum g, sum g * (x - mean)).  Fold, scale the second sum by invstd (-> sum g * xhat), all-reduce when synchronised,
    apply.  Returns (dx, dweight, dbias) like _bn_bwd_impl."""
    n, c, h, w = x.shape
    p = n * h * w
    dev = x.device
    synced = count is not None
    rows = part.numel() // (2 * c)
    sums = torch.empty(2 * c + 1, dtype=torch.float64, device=dev)
    ws = _ws(call('ssg_bn_stats_from_partials_workspace_bytes', rows, c), dev)
    call('ssg_bn_stats_from_partials_f32', ptr(part), rows, c, ptr(sums), float(p) if synced else 0.0, ptr(ws), stream_ptr())
    sums[c:2 * c] *= stats[1].double()
    local = sums.clone() if synced else sums
    if synced:
        _timed_all_reduce('sync_bn_bwd', sums, group)
    dwb = torch.empty((2, c), dtype=torch.float32, device=dev)
    dx = new_nhwc(n, c, h, w, dev)
    with _hbm('bn_bwd', 4.0 * p * c * 3):
        call('ssg_bn_bwd_apply_f32', ptr(x), None, ptr(g), p, c, _ld(x), 0, _ld(g),
             ptr(stats[0]), ptr(stats[1]), ptr(weight), ptr(stats[2]), ptr(stats[3]), ptr(sums), 0.0 if synced else float(p), ACT_NONE, 0.0,
             ptr(dx), _ld(dx), None, 0, ptr(dwb[0]), ptr(dwb[1]), stream_ptr())
    if synced:
        dwb = torch.stack([local[c:2 * c], local[:c]]).float()
    return dx, dwb[0], dwb[1]


def _relabel(t, c):
    """The same NHWC-with-stride memory seen with `c` logical channels (c <= pixel stride); a base tensor, not a view."""
    n, _, h, w = t.shape
    return torch.empty(0, device=t.device, dtype=t.dtype).set_(t.untyped_storage(), t.storage_offset(), (n, c, h, w), t.stride())


class _BatchNormAct(torch.autograd.Function):
    """Channel counts that are not multiples of 4 (the 1-channel psi BatchNorm of Attention_block, archs.py:128-132)
    run on the padded lanes too: zero data with zero weight/bias stays zero."""

    @staticmethod
    def forward(ctx, x, weight, bias, running_mean, running_var, res, eps, momentum, act, slope, var_mode, group, part=None):
        x = to_nhwc(x)
        res = to_nhwc(res) if res is not None else None
        c = x.shape[1]
        c4 = pad4(c)
        if c4 != c:
            x = _relabel(x, c4)
            res = _relabel(res, c4) if res is not None else None
            wp = torch.nn.functional.pad(weight.detach(), (0, c4 - c)); bp = torch.nn.functional.pad(bias.detach(), (0, c4 - c))
            rm = torch.nn.functional.pad(running_mean, (0, c4 - c)) if running_mean is not None else None
            rv = torch.nn.functional.pad(running_var, (0, c4 - c), value=1.0) if running_var is not None else None
            y, stats, cnt = _bn_fwd_impl(x, wp, bp, rm, rv, res, eps, momentum, act, slope, var_mode, group)
            if running_mean is not None:
                running_mean.copy_(rm[:c])
            if running_var is not None:
                running_var.copy_(rv[:c])
            weight = wp
        else:
            y, stats, cnt = _bn_fwd_impl(x, weight, bias, running_mean, running_var, res, eps, momentum, act, slope, var_mode, group, part=part)
        ctx.save_for_backward(x, y if (act != ACT_NONE and res is not None) else None, weight, stats)   # no residual: mask is recomputed
        ctx.cfg = (act, slope, group, cnt, res is not None, c)
        return _relabel(y, c) if c4 != c else y

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, dy):
        x, y, weight, stats = ctx.saved_tensors
        act, slope, group, cnt, has_res, c = ctx.cfg
        dy = to_nhwc(dy)
        c4 = x.shape[1]
        if c4 != c:
            dy = _relabel(dy, c4)
        dx, dres, dw, db = _bn_bwd_impl(x, y, dy, weight, stats, act, slope, group, cnt, has_res and ctx.needs_input_grad[5],
                                        had_res=has_res)
        if c4 != c:
            dx = _relabel(dx, c)
            dres = _relabel(dres, c) if dres is not None else None
            dw, db = dw[:c].clone(), db[:c].clone()
        return dx, dw, db, None, None, dres, None, None, None, None, None, None, None


class _AffineAct(torch.autograd.Function):
    """Eval-mode BN: y = x*scale + shift (+res) -> act with constant scale/shift."""

    @staticmethod
    def forward(ctx, x, scale, shift, res, act, slope):
        x = to_nhwc(x)
        res = to_nhwc(res) if res is not None else None
        n, c, h, w = x.shape
        y = new_nhwc(n, c, h, w, x.device)
        call('ssg_bn_apply_f32', ptr(x), n * h * w, c, _ld(x), ptr(scale), ptr(shift), ptr(res), _ld(res) if res is not None else 0,
             act, slope, ptr(y), _ld(y), stream_ptr())
        return y

    @staticmethod
    def backward(ctx, dy):
        raise NotImplementedError('eval-mode batch norm is inference-only in ssunet-gan_amd')


def batch_norm_act(x, bn, res=None, act=ACT_NONE, slope=0.0, group=None, var_mode=None, stats_part=None):
    """nn.BatchNorm2d `bn` (any _BatchNorm holding weight/bias/running stats) + residual + activation.
    `stats_part`: the per-tile statistics the producing conv2d(..., bn_stats=True) returned (train mode only)."""
    _lib.require_gpu(x)
    if bn.training or not bn.track_running_stats:
        if bn.momentum is None:
            raise NotImplementedError('cumulative-average batch norm (momentum=None)')
        if bn.track_running_stats and bn.num_batches_tracked is not None and not _synced(group):
            bn.num_batches_tracked.add_(1)           # the reference's synchronised branch never counts batches (batchnorm.py:57-80)
        if var_mode is None:
            var_mode = getattr(bn, '_ssg_var_mode', 1 if group is not None else 0)
        return _BatchNormAct.apply(x, bn.weight, bn.bias, bn.running_mean if bn.track_running_stats else None,
                                   bn.running_var if bn.track_running_stats else None, res, float(bn.eps), float(bn.momentum),
                                   int(act), float(slope), int(var_mode), group, stats_part)
    with torch.no_grad():
        scale = torch.rsqrt(bn.running_var + bn.eps)
        if bn.weight is not None:
            scale = scale * bn.weight
        shift = -bn.running_mean * scale
        if bn.bias is not None:
            shift = shift + bn.bias
        c = scale.numel()
        if c % 4:                                           # 1-channel psi batch norm: run the padded lanes with scale = shift = 0
            scale = torch.nn.functional.pad(scale, (0, pad4(c) - c)); shift = torch.nn.functional.pad(shift, (0, pad4(c) - c))
            y = _AffineAct.apply(_relabel(to_nhwc(x), pad4(c)), scale.contiguous(), shift.contiguous(),
                                 _relabel(to_nhwc(res), pad4(c)) if res is not None else None, int(act), float(slope))
            return _relabel(y, c)
    return _AffineAct.apply(x, scale.contiguous(), shift.contiguous(), res, int(act), float(slope))


# ----------------------------------------------------------------------------- pool / unpool / upsample
class _MaxPool(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        x = to_nhwc(x)
        n, c, h, w = x.shape
        if c % 4 or h % 2 or w % 2:
            raise ValueError('max_pool2x2: needs C %% 4 == 0 and even H, W (got %s)' % (tuple(x.shape),))
        oh, ow = h // 2, w // 2
        y = new_nhwc(n, c, oh, ow, x.device)
        idx = torch.empty((n, oh, ow, c), dtype=torch.uint8, device=x.device)
        with _hbm('maxpool_fwd', n * h * w * c * (4.0 + 1.0 + 0.25)):        # 1 read + 1/4 write + 1/4 byte index per input element
            call('ssg_maxpool2x2_fwd_f32', ptr(x), n, h, w, c, _ld(x), ptr(y), _ld(y), ptr(idx), stream_ptr())
        ctx.save_for_backward(idx)
        ctx.hw = (h, w)
        ctx.mark_non_differentiable(idx)
        return y, idx

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, dy, _):
        (idx,) = ctx.saved_tensors
        dy = to_nhwc(dy)
        n, c, oh, ow = dy.shape
        h, w = ctx.hw
        dx = new_nhwc(n, c, h, w, dy.device)
        with _hbm('maxpool_bwd', n * h * w * c * (1.0 + 0.25 + 4.0)):
            call('ssg_maxpool2x2_bwd_f32', ptr(dy), _ld(dy), ptr(idx), n, h, w, c, ptr(dx), _ld(dx), stream_ptr())
        return dx


def max_pool2x2(x):
    """nn.MaxPool2d(2, 2, return_indices=True): returns (y, idx) with idx the 1-byte window argmax."""
    _lib.require_gpu(x)
    return _MaxPool.apply(x)


class _MaxPoolSkip(torch.autograd.Function):
    """max_pool2x2 of a tensor that ALSO feeds a skip connection: returns (y, idx, x_skip) with x_skip = x.  One autograd node
    receives both gradients and forms dx = d_skip + pool_backward(dy) in the pool-backward pass, so autograd never launches
    its own 3-tensor add for the encoder outputs (archs.py:628-667: up to 1 GB each at 16 x 512^2)."""

    @staticmethod
    def forward(ctx, x):
        x = to_nhwc(x)
        n, c, h, w = x.shape
        if c % 4 or h % 2 or w % 2:
            raise ValueError('max_pool2x2: needs C %% 4 == 0 and even H, W (got %s)' % (tuple(x.shape),))
        oh, ow = h // 2, w // 2
        y = new_nhwc(n, c, oh, ow, x.device)
        idx = torch.empty((n, oh, ow, c), dtype=torch.uint8, device=x.device)
        with _hbm('maxpool_fwd', n * h * w * c * (4.0 + 1.0 + 0.25)):        # 1 read + 1/4 write + 1/4 byte index per input element
            call('ssg_maxpool2x2_fwd_f32', ptr(x), n, h, w, c, _ld(x), ptr(y), _ld(y), ptr(idx), stream_ptr())
        ctx.save_for_backward(idx)
        ctx.hw = (h, w)
        ctx.mark_non_differentiable(idx)
        return y, idx, x

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, dy, _, dskip):
        (idx,) = ctx.saved_tensors
        h, w = ctx.hw
        if dy is None:
            return dskip
        dy = to_nhwc(dy)
        n, c, oh, ow = dy.shape
        dx = new_nhwc(n, c, h, w, dy.device)
        if dskip is None:
            with _hbm('maxpool_bwd', n * h * w * c * (1.0 + 0.25 + 4.0)):
                call('ssg_maxpool2x2_bwd_f32', ptr(dy), _ld(dy), ptr(idx), n, h, w, c, ptr(dx), _ld(dx), stream_ptr())
        else:
            dskip = to_nhwc(dskip)
            with _hbm('maxpool_bwd', n * h * w * c * (1.0 + 0.25 + 4.0 + 4.0)):      # + the skip gradient it adds
                call('ssg_maxpool2x2_bwd_add_f32', ptr(dy), _ld(dy), ptr(idx), ptr(dskip), _ld(dskip), n, h, w, c, ptr(dx), _ld(dx), stream_ptr())
        return dx


def max_pool2x2_skip(x):
    """(y, idx, x_skip): max_pool2x2(x) plus x handed back for a skip connection; use x_skip (not x) downstream."""
    _lib.require_gpu(x)
    return _MaxPoolSkip.apply(x)


class _MaxUnpool(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, idx):
        x = to_nhwc(x)
        n, c, h, w = x.shape
        if tuple(idx.shape) != (n, h, w, c) or idx.dtype != torch.uint8:
            raise ValueError('max_unpool2x2: indices %s do not match input %s' % (tuple(idx.shape), tuple(x.shape)))
        y = new_nhwc(n, c, 2 * h, 2 * w, x.device)
        with _hbm('unpool_fwd', n * h * w * c * (4.0 + 1.0 + 16.0)):         # 1/4 read + 1/4 byte index + 1 write per output element
            call('ssg_maxunpool2x2_fwd_f32', ptr(x), _ld(x), ptr(idx), n, 2 * h, 2 * w, c, ptr(y), _ld(y), stream_ptr())
        ctx.save_for_backward(idx)
        return y

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, dy):
        (idx,) = ctx.saved_tensors
        dy = to_nhwc(dy)
        n, c, oh, ow = dy.shape
        dx = new_nhwc(n, c, oh // 2, ow // 2, dy.device)
        with _hbm('unpool_bwd', n * (oh // 2) * (ow // 2) * c * (4.0 + 1.0 + 4.0)):
            call('ssg_maxunpool2x2_bwd_f32', ptr(dy), _ld(dy), ptr(idx), n, oh, ow, c, ptr(dx), _ld(dx), stream_ptr())
        return dx, None


def max_unpool2x2(x, idx):
    _lib.require_gpu(x)
    return _MaxUnpool.apply(x, idx)


def _make_upsample(fwd_name, bwd_name, stage):
    class _Up(torch.autograd.Function):
        @staticmethod
        def forward(ctx, x):
            x = to_nhwc(x)
            n, c, h, w = x.shape
            if c % 4:
                raise ValueError('upsample2x: C %% 4 != 0')
            y = new_nhwc(n, c, 2 * h, 2 * w, x.device)
            with _hbm(stage + '_fwd', n * h * w * c * (4.0 + 16.0)):             # 1/4 read + 1 write per output element
                call(fwd_name, ptr(x), n, h, w, c, _ld(x), ptr(y), _ld(y), stream_ptr())
            return y

        @staticmethod
        @torch.autograd.function.once_differentiable
        def backward(ctx, dy):
            dy = to_nhwc(dy)
            n, c, oh, ow = dy.shape
            dx = new_nhwc(n, c, oh // 2, ow // 2, dy.device)
            with _hbm(stage + '_bwd', n * (oh // 2) * (ow // 2) * c * (16.0 + 4.0)):
                call(bwd_name, ptr(dy), _ld(dy), n, oh // 2, ow // 2, c, ptr(dx), _ld(dx), stream_ptr())
            return dx
    return _Up


_BilinearUp = _make_upsample('ssg_upsample2x_bilinear_fwd_f32', 'ssg_upsample2x_bilinear_bwd_f32', 'bilinear')
_NearestUp = _make_upsample('ssg_upsample2x_nearest_fwd_f32', 'ssg_upsample2x_nearest_bwd_f32', 'nearest')


def upsample2x_bilinear(x):
    """nn.Upsample(scale_factor=2, mode='bilinear', align_corners=True)."""
    _lib.require_gpu(x)
    return _BilinearUp.apply(x)


def upsample2x_nearest(x):
    """nn.Upsample(scale_factor=2) (nearest)."""
    _lib.require_gpu(x)
    return _NearestUp.apply(x)


class _AvgPoolFlat(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, o):
        x = to_nhwc(x)
        n, c, h, w = x.shape
        y = torch.empty((n, c * o * o), dtype=torch.float32, device=x.device)
        call('ssg_adaptive_avgpool_flat_fwd_f32', ptr(x), n, h, w, c, _ld(x), o, ptr(y), stream_ptr())
        ctx.cfg = (n, c, h, w, o)
        return y

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, dy):
        n, c, h, w, o = ctx.cfg
        dy = dy.contiguous()
        dx = new_nhwc(n, c, h, w, dy.device, zero=(c % 4 != 0))
        call('ssg_adaptive_avgpool_flat_bwd_f32', ptr(dy), n, h, w, c, o, ptr(dx), _ld(dx), stream_ptr())
        return dx, None


def adaptive_avgpool_flat(x, o=6):
    """nn.AdaptiveAvgPool2d((o, o)) followed by .view(N, -1) in NCHW order."""
    _lib.require_gpu(x)
    return _AvgPoolFlat.apply(x, int(o))


# ----------------------------------------------------------------------------- SPADE modulate
class _SpadeModulate(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gb):
        x = to_nhwc(x); gb = to_nhwc(gb)
        n, c, h, w = x.shape
        if c % 4 or gb.shape[1] != 2 * c:
            raise ValueError('spade_modulate: x has %d channels, gamma|beta has %d' % (c, gb.shape[1]))
        y = new_nhwc(n, c, h, w, x.device)
        with _hbm('spade_modulate_fwd', 16.0 * n * h * w * c):                   # x, gamma, beta in; out
            call('ssg_spade_modulate_fwd_f32', ptr(x), _ld(x), ptr(gb), _ld(gb), n * h * w, c, ptr(y), _ld(y), stream_ptr())
        ctx.save_for_backward(x, gb)
        return y

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, dy):
        x, gb = ctx.saved_tensors
        dy = to_nhwc(dy)
        n, c, h, w = x.shape
        dx = new_nhwc(n, c, h, w, x.device)
        dgb = new_nhwc(n, 2 * c, h, w, x.device)
        call('ssg_spade_modulate_bwd_f32', ptr(x), _ld(x), ptr(gb), _ld(gb), ptr(dy), _ld(dy), n * h * w, c,
             ptr(dx), _ld(dx), ptr(dgb), _ld(dgb), stream_ptr())
        return dx, dgb


def spade_modulate(x, gb):
    """out = x*(1+gamma)+beta with gamma = gb[:, :C], beta = gb[:, C:] (normalization.py:120)."""
    _lib.require_gpu(x)
    return _SpadeModulate.apply(x, gb)


# ----------------------------------------------------------------------------- losses
class _SegLoss(torch.autograd.Function):
    """Fused BCEDiceLoss + MSELoss + IoU/Dice metrics.  Returns a float32[8] tensor `res`
    (layout in include/ssunet_hip.h); gradients flow from res[0] (BCEDice), res[1] (MSE) and res[2] (StableBCE);
    res[3:] are metrics (their incoming gradient is ignored, as the reference computes them in numpy)."""

    @staticmethod
    def forward(ctx, x, t, mc0):
        x = to_nhwc(x); t = to_nhwc(t)
        n, c, h, w = x.shape
        s = h * w
        dev = x.device
        res = torch.empty(8, dtype=torch.float32, device=dev)
        stats = torch.empty(3 * n + 5, dtype=torch.float64, device=dev)
        ws = _ws(call('ssg_seg_loss_workspace_bytes', n, s, c), dev)
        call('ssg_seg_loss_fwd_f32', ptr(x), _ld(x), ptr(t), _ld(t), n, s, c, mc0, ptr(res), ptr(stats), ptr(ws), stream_ptr())
        ctx.save_for_backward(x, t, res, stats)
        ctx.mark_non_differentiable(stats)
        return res, stats

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g, _):
        x, t, res, stats = ctx.saved_tensors
        n, c, h, w = x.shape
        g = g.contiguous()
        dx = new_nhwc(n, c, h, w, x.device)
        call('ssg_seg_loss_bwd_f32', ptr(x), _ld(x), ptr(t), _ld(t), n, h * w, c, ptr(res), ptr(stats),
             C.c_void_p(g.data_ptr()), C.c_void_p(g.data_ptr() + 4), C.c_void_p(g.data_ptr() + 8), ptr(dx), _ld(dx), stream_ptr())
        return dx, None, None


def seg_loss(logits, target, metric_first_channel=1, with_sums=False):
    """Fused loss/metric pass.  Returns res (float32[8]); with_sums=True also returns the fp64[5]
    metric partial sums (for cross-rank reduction of IoU/Dice)."""
    _lib.require_gpu(logits)
    res, stats = _SegLoss.apply(logits, target, int(metric_first_channel))
    if with_sums:
        return res, stats[-5:]
    return res


class _BCEConst(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, label):
        n = x.numel()
        if x.dim() != 2 or x.shape[1] != 1:
            raise ValueError('bce_with_logits_const expects [N, 1] logits')
        loss = torch.empty((), dtype=torch.float32, device=x.device)
        call('ssg_bce_logits_const_fwd_f32', ptr(x), n, x.stride(0), label, ptr(loss), stream_ptr())
        ctx.save_for_backward(x)
        ctx.label = label
        return loss

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        n = x.shape[0]
        buf = torch.empty((n, 4), dtype=torch.float32, device=x.device)
        g = g.contiguous()
        call('ssg_bce_logits_const_bwd_f32', ptr(x), n, x.stride(0), ctx.label, ptr(g), ptr(buf), 4, stream_ptr())
        return buf[:, :1], None


def bce_with_logits_const(x, label):
    """nn.BCEWithLogitsLoss()(x, full_like(x, label)) for x of shape [N, 1]."""
    _lib.require_gpu(x)
    return _BCEConst.apply(x, float(label))


class _NanToZero(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        ld = nhwc_ld(x) if x.dim() == 4 else None
        if ld is not None:                          # operate on the whole padded buffer
            n, c, h, w = x.shape
            dense = x.as_strided((n * h * w * ld,), (1,))
        elif x.is_contiguous():
            dense = x.view(-1)
        else:
            raise ValueError('nan_to_zero_: unsupported layout (stride %s)' % (x.stride(),))
        mask = torch.empty(dense.numel(), dtype=torch.uint8, device=x.device)
        call('ssg_nan_to_zero_f32', ptr(dense), dense.numel(), ptr(mask), stream_ptr())
        ctx.save_for_backward(mask)
        ctx.nhwc = ld is not None
        ctx.mark_dirty(x)
        return x

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g):
        (mask,) = ctx.saved_tensors
        if ctx.nhwc:
            g = to_nhwc(g)
            n, c, h, w = g.shape
            ld = _ld(g)
            out = new_nhwc(n, c, h, w, g.device, ld=ld)
            call('ssg_mask_zero_f32', ptr(g), ptr(mask), n * h * w * ld, ptr(out), stream_ptr())
            return out
        g = g.contiguous()
        out = torch.empty_like(g)
        call('ssg_mask_zero_f32', ptr(g), ptr(mask), g.numel(), ptr(out), stream_ptr())
        return out


def nan_to_zero_(x):
    """x[isnan(x)] = 0 in place, differentiable (train_seg_gan.py:190)."""
    _lib.require_gpu(x)
    return _NanToZero.apply(x)


# ----------------------------------------------------------------------------- depthwise conv (unwired rows A10/A11)
class _DwConv2d(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, stride, pad):
        x = to_nhwc(x)
        n, c, h, w = x.shape
        if c % 4 or weight.shape[0] != c or weight.shape[1] != 1:
            raise ValueError('dwconv2d: depthwise weight [C,1,KH,KW] with C %% 4 == 0 expected, got %s for C=%d' % (tuple(weight.shape), c))
        kh, kw = weight.shape[2:]
        pt, pb, pl, pr = _pad4(pad)
        oh, ow = _out_hw(h, w, kh, kw, stride, pad)
        y = new_nhwc(n, c, oh, ow, x.device)
        wc = weight.contiguous()
        call('ssg_dwconv2d_fwd_f32', ptr(x), n, h, w, c, _ld(x), ptr(wc), ptr(bias), kh, kw, stride, pt, pl, oh, ow, ptr(y), _ld(y), stream_ptr())
        ctx.save_for_backward(x, wc)
        ctx.cfg = (stride, pt, pl, oh, ow, bias is not None)
        return y

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, dy):
        x, wc = ctx.saved_tensors
        stride, pt, pl, oh, ow, has_bias = ctx.cfg
        dy = to_nhwc(dy)
        n, c, h, w = x.shape
        kh, kw = wc.shape[2:]
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = new_nhwc(n, c, h, w, x.device)
            call('ssg_dwconv2d_dgrad_f32', ptr(dy), _ld(dy), n, h, w, c, ptr(wc), kh, kw, stride, pt, pl, oh, ow, ptr(dx), _ld(dx), stream_ptr())
        if ctx.needs_input_grad[1]:
            dw = torch.empty_like(wc)
            ws = _ws(call('ssg_dwconv2d_wgrad_workspace_bytes', n, oh, ow, c, kh, kw), x.device)
            call('ssg_dwconv2d_wgrad_f32', ptr(x), n, h, w, c, _ld(x), ptr(dy), _ld(dy), kh, kw, stride, pt, pl, oh, ow, ptr(dw), ptr(ws), stream_ptr())
        if has_bias and ctx.needs_input_grad[2]:
            db = _channel_sum(dy, c)
        return dx, dw, db, None, None


def dwconv2d(x, weight, bias=None, stride=1, padding=0):
    """Depthwise F.conv2d (groups = C)."""
    _lib.require_gpu(x)
    pad = tuple(int(v) for v in padding) if isinstance(padding, (tuple, list)) else int(padding)
    return _DwConv2d.apply(x, weight, bias, int(stride), pad)


UNARY_SWISH, UNARY_SIGMOID, UNARY_GAUSSIAN = 0, 1, 2


class _Unary(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, op):
        x = to_nhwc(x)
        n, c, h, w = x.shape
        y = new_nhwc(n, c, h, w, x.device)
        call('ssg_unary_fwd_f32', ptr(x), _ld(x), n * h * w, pad4(c), op, ptr(y), _ld(y), stream_ptr())
        ctx.save_for_backward(x)
        ctx.op = op
        return y

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        dy = to_nhwc(dy)
        n, c, h, w = x.shape
        dx = new_nhwc(n, c, h, w, x.device)
        call('ssg_unary_bwd_f32', ptr(x), _ld(x), ptr(dy), _ld(dy), n * h * w, pad4(c), ctx.op, ptr(dx), _ld(dx), stream_ptr())
        return dx, None


def swish(x):
    """x * sigmoid(x) with the reference's hand-written backward (efficientnet_pytorch/utils.py:37-48)."""
    _lib.require_gpu(x)
    return _Unary.apply(x, UNARY_SWISH)


def sigmoid(x):
    _lib.require_gpu(x)
    return _Unary.apply(x, UNARY_SIGMOID)


def gaussian(x):
    """exp(-x*x) (xresidualblock.py:5-7)."""
    _lib.require_gpu(x)
    return _Unary.apply(x, UNARY_GAUSSIAN)


class _Mul(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b):
        a = to_nhwc(a); b = to_nhwc(b)
        n, c, h, w = a.shape
        y = new_nhwc(n, c, h, w, a.device)
        call('ssg_mul_fwd_f32', ptr(a), _ld(a), ptr(b), _ld(b), n * h * w, pad4(c), ptr(y), _ld(y), stream_ptr())
        ctx.save_for_backward(a, b)
        return y

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, dy):
        a, b = ctx.saved_tensors
        dy = to_nhwc(dy)
        n, c, h, w = a.shape
        da = new_nhwc(n, c, h, w, a.device); db = new_nhwc(n, c, h, w, a.device)
        call('ssg_mul_bwd_f32', ptr(a), _ld(a), ptr(b), _ld(b), ptr(dy), _ld(dy), n * h * w, pad4(c),
             ptr(da), _ld(da), ptr(db), _ld(db), stream_ptr())
        return da, db


def mul(a, b):
    _lib.require_gpu(a)
    return _Mul.apply(a, b)


class _PixelGate(torch.autograd.Function):
    """x * sigmoid(g) with a one-channel gate g broadcast over the channels (Attention_block, archs.py:138-144)."""

    @staticmethod
    def forward(ctx, x, g):
        x = to_nhwc(x); g = to_nhwc(g)
        n, c, h, w = x.shape
        if tuple(g.shape) != (n, 1, h, w):
            raise ValueError('pixel_gate: gate must be [N,1,H,W], got %s' % (tuple(g.shape),))
        y = new_nhwc(n, c, h, w, x.device)
        call('ssg_pixel_gate_fwd_f32', ptr(x), _ld(x), ptr(g), _ld(g), n * h * w, pad4(c), ptr(y), _ld(y), stream_ptr())
        ctx.save_for_backward(x, g)
        return y

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, dy):
        x, g = ctx.saved_tensors
        dy = to_nhwc(dy)
        n, c, h, w = x.shape
        dx = new_nhwc(n, c, h, w, x.device); dg = new_nhwc(n, 1, h, w, x.device)
        call('ssg_pixel_gate_bwd_f32', ptr(x), _ld(x), ptr(g), _ld(g), ptr(dy), _ld(dy), n * h * w, pad4(c),
             ptr(dx), _ld(dx), ptr(dg), _ld(dg), stream_ptr())
        return dx, dg


def pixel_gate(x, g):
    _lib.require_gpu(x)
    return _PixelGate.apply(x, g)


class _GlobalAvgPool(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        x = to_nhwc(x)
        n, c, h, w = x.shape
        if c % 4:
            raise ValueError('global_avgpool: C %% 4 != 0')
        y = new_nhwc(n, c, 1, 1, x.device)
        ws = _ws(call('ssg_sample_channel_sum_workspace_bytes', n, h * w, c), x.device)
        call('ssg_sample_channel_sum_f32', ptr(x), _ld(x), None, 0, n, h * w, c, 1.0 / (h * w), ptr(y), ptr(ws), stream_ptr())
        ctx.cfg = (n, c, h, w)
        return y

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, dy):
        n, c, h, w = ctx.cfg
        dy = to_nhwc(dy)
        dx = new_nhwc(n, c, h, w, dy.device)
        call('ssg_broadcast_rows_f32', ptr(dy), n, h * w, c, 1.0 / (h * w), ptr(dx), _ld(dx), stream_ptr())
        return dx


def global_avgpool(x):
    """F.adaptive_avg_pool2d(x, 1) -> [N, C, 1, 1]."""
    _lib.require_gpu(x)
    return _GlobalAvgPool.apply(x)


SE_FUSED = _os.environ.get('SSG_SE_FUSED', '1') != '0'


class _SEGate(torch.autograd.Function):
    """sigmoid(_se_expand(swish(_se_reduce(sq)))) on the pooled [N, C, 1, 1] vector (efficientnet_pytorch/model.py:84-86) as
    2 kernels forward and 3 backward (csrc/se_gate.hip) instead of two 1x1 conv launches with their packs, activations,
    split-K slabs and reducers."""

    @staticmethod
    def forward(ctx, sq, w1, b1, w2, b2):
        sq = to_nhwc(sq)
        n, c = sq.shape[0], sq.shape[1]
        s_ = w1.shape[0]
        dev = sq.device
        w1c = w1.detach().reshape(s_, c).contiguous(); w2c = w2.detach().reshape(c, s_).contiguous()
        h_pre = torch.empty((n, s_), dtype=torch.float32, device=dev)
        gate = new_nhwc(n, c, 1, 1, dev)
        tmp = torch.empty(call('ssg_se_gate_workspace_floats', n, c, s_), dtype=torch.float32, device=dev)
        call('ssg_se_gate_fwd_f32', ptr(sq), _ld(sq), n, c, ptr(w1c), ptr(b1.detach()) if b1 is not None else None, ptr(w2c),
             ptr(b2.detach()) if b2 is not None else None, s_, ptr(h_pre), ptr(gate), _ld(gate), ptr(tmp), stream_ptr())
        ctx.save_for_backward(sq, w1c, w2c, h_pre, gate)
        ctx.cfg = (n, c, s_, tuple(w1.shape), tuple(w2.shape), b1 is not None, b2 is not None)
        return gate

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, dgate):
        sq, w1c, w2c, h_pre, gate = ctx.saved_tensors
        n, c, s_, shp1, shp2, has_b1, has_b2 = ctx.cfg
        dev = sq.device
        dgate = to_nhwc(dgate)
        dsq = new_nhwc(n, c, 1, 1, dev)
        dw1 = torch.empty((s_, c), dtype=torch.float32, device=dev); dw2 = torch.empty((c, s_), dtype=torch.float32, device=dev)
        db1 = torch.empty(s_, dtype=torch.float32, device=dev) if has_b1 else None
        db2 = torch.empty(c, dtype=torch.float32, device=dev) if has_b2 else None
        tmp = torch.empty(call('ssg_se_gate_workspace_floats', n, c, s_), dtype=torch.float32, device=dev)
        call('ssg_se_gate_bwd_f32', ptr(dgate), _ld(dgate), ptr(gate), _ld(gate), ptr(h_pre), ptr(sq), _ld(sq), n, c, ptr(w1c), ptr(w2c), s_,
             ptr(dsq), _ld(dsq), ptr(dw1), ptr(db1), ptr(dw2), ptr(db2), ptr(tmp), stream_ptr())
        return dsq, dw1.reshape(shp1), db1, dw2.reshape(shp2), db2


def se_gate(sq, reduce_conv, expand_conv):
    """The squeeze-excite gate for pooled `sq` [N, C, 1, 1] through the block's `_se_reduce` / `_se_expand` 1x1 conv modules;
    None when the shape is outside the fused kernels' range (the caller then runs the convs)."""
    _lib.require_gpu(sq)
    w1, w2 = reduce_conv.weight, expand_conv.weight
    n, c = sq.shape[0], sq.shape[1]
    if not SE_FUSED or tuple(w1.shape[1:]) != (c, 1, 1) or tuple(w2.shape) != (c, w1.shape[0], 1, 1) or sq.dtype != torch.float32:
        return None
    if not call('ssg_se_gate_ok', n, c, w1.shape[0]):
        return None
    return _SEGate.apply(sq, w1, reduce_conv.bias, w2, expand_conv.bias)


class _ChannelScale(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, s):
        x = to_nhwc(x); s = to_nhwc(s)
        n, c, h, w = x.shape
        if tuple(s.shape) != (n, c, 1, 1) or c % 4:
            raise ValueError('channel_scale: gate %s does not match %s' % (tuple(s.shape), tuple(x.shape)))
        y = new_nhwc(n, c, h, w, x.device)
        call('ssg_channel_scale_fwd_f32', ptr(x), _ld(x), ptr(s), n, h * w, c, ptr(y), _ld(y), stream_ptr())
        ctx.save_for_backward(x, s)
        return y

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, dy):
        x, s = ctx.saved_tensors
        dy = to_nhwc(dy)
        n, c, h, w = x.shape
        dx = new_nhwc(n, c, h, w, x.device)
        call('ssg_channel_scale_fwd_f32', ptr(dy), _ld(dy), ptr(s), n, h * w, c, ptr(dx), _ld(dx), stream_ptr())
        ds = new_nhwc(n, c, 1, 1, x.device)
        ws = _ws(call('ssg_sample_channel_sum_workspace_bytes', n, h * w, c), x.device)
        call('ssg_sample_channel_sum_f32', ptr(dy), _ld(dy), ptr(x), _ld(x), n, h * w, c, 1.0, ptr(ds), ptr(ws), stream_ptr())
        return dx, ds


def channel_scale(x, s):
    """x * s with s of shape [N, C, 1, 1] (squeeze-excite gate, drop-connect mask)."""
    _lib.require_gpu(x)
    return _ChannelScale.apply(x, s)


# ----------------------------------------------------------------------------- spectral norm (unwired row A12)
class _SpectralNormWeight(torch.autograd.Function):
    @staticmethod
    def forward(ctx, weight, u, v, n_iter, eps):
        wm = weight.contiguous()
        rows = wm.shape[0]
        cols = wm.numel() // rows
        out = torch.empty_like(wm)
        sigma = torch.empty((), dtype=torch.float32, device=wm.device)
        ws = _ws(call('ssg_spectral_norm_workspace_bytes', rows, cols), wm.device)
        call('ssg_spectral_norm_fwd_f32', ptr(wm), rows, cols, ptr(u), ptr(v), n_iter, float(eps), ptr(out), ptr(sigma), ptr(ws), stream_ptr())
        # the reference clones u, v after the in-place iteration so backward sees this call's vectors
        ctx.save_for_backward(wm, u.clone(), v.clone(), sigma)
        ctx.mark_non_differentiable(sigma)
        return out, sigma

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, dwsn, _):
        wm, u, v, sigma = ctx.saved_tensors
        rows = wm.shape[0]
        cols = wm.numel() // rows
        dwsn = dwsn.contiguous()
        dw = torch.empty_like(wm)
        ws = _ws(call('ssg_spectral_norm_workspace_bytes', rows, cols), wm.device)
        call('ssg_spectral_norm_bwd_f32', ptr(dwsn), ptr(wm), rows, cols, ptr(u), ptr(v), ptr(sigma), ptr(dw), ptr(ws), stream_ptr())
        return dw, None, None, None, None


class _Conv2dSN(torch.autograd.Function):
    """conv2d(x, weight_orig / sigma) (+bias, +activation) with the spectral norm INSIDE the conv: one in-place power iteration
    on (u, v) -> sigma (spectral_norm.py:73-88), then weight_orig is packed with 1/sigma for the forward and the input
    gradient.  W / sigma is never written out, nothing of the weight is cached across forwards (sigma moves every training
    forward), and the weight gradient is mapped back through dW = dWsn/sigma - (sum(dWsn . W)/sigma^2) u v^T."""

    @staticmethod
    def forward(ctx, x, weight_orig, u, v, bias, stride, pad, act, slope, n_iter, eps):
        x = to_nhwc(x)
        wm = weight_orig.contiguous()
        rows = wm.shape[0]
        cols = wm.numel() // rows
        sigma = torch.empty((), dtype=torch.float32, device=wm.device)
        ws = _ws(call('ssg_spectral_norm_workspace_bytes', rows, cols), wm.device)
        call('ssg_spectral_norm_fwd_f32', ptr(wm), rows, cols, ptr(u), ptr(v), n_iter, float(eps), None, ptr(sigma), ptr(ws), stream_ptr())
        y = _conv_fwd_impl(x, None, wm, bias, stride, pad, act, slope, wscale=sigma)
        ctx.save_for_backward(x, wm, u.clone(), v.clone(), sigma, y if act != ACT_NONE else None)
        ctx.cfg = (stride, pad, act, slope, bias is not None)
        return y

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, dy):
        x, wm, u, v, sigma, y = ctx.saved_tensors
        stride, pad, act, slope, has_bias = ctx.cfg
        dy = to_nhwc(dy)
        if act != ACT_NONE:
            dy = _act_bwd(y, dy, act, slope)
        n, c1, h, w = x.shape
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = _conv_dgrad_impl(dy, wm, stride, pad, h, w, 0, c1, wscale=sigma)
        if ctx.needs_input_grad[1]:
            dwsn = _conv_wgrad_impl(x, None, dy, wm.shape, stride, pad)
            rows = wm.shape[0]
            cols = wm.numel() // rows
            dw = torch.empty_like(wm)
            ws = _ws(call('ssg_spectral_norm_workspace_bytes', rows, cols), wm.device)
            call('ssg_spectral_norm_bwd_f32', ptr(dwsn), ptr(wm), rows, cols, ptr(u), ptr(v), ptr(sigma), ptr(dw), ptr(ws), stream_ptr())
        if has_bias and ctx.needs_input_grad[4]:
            db = _channel_sum(dy, wm.shape[0])
        return dx, dw, None, None, db, None, None, None, None, None, None


def conv2d_sn(x, weight_orig, u, v, bias=None, stride=1, padding=0, act=ACT_NONE, slope=0.0, n_power_iterations=1, eps=1e-12):
    """Spectrally normalised conv2d: see _Conv2dSN.  `n_power_iterations = 0` in eval mode (no update of u, v)."""
    _lib.require_gpu(x)
    pad = tuple(int(p) for p in padding) if isinstance(padding, (tuple, list)) else int(padding)
    return _Conv2dSN.apply(x, weight_orig, u, v, bias, int(stride), pad, int(act), float(slope), int(n_power_iterations), float(eps))


def spectral_norm_weight(weight_orig, u, v, n_power_iterations=1, eps=1e-12):
    """weight_orig / sigma after `n_power_iterations` in-place power-iteration updates of u, v."""
    _lib.require_gpu(weight_orig)
    return _SpectralNormWeight.apply(weight_orig, u, v, int(n_power_iterations), float(eps))


class _Add(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b):
        a = to_nhwc(a); b = to_nhwc(b)
        n, c, h, w = a.shape
        if _ld(a) != _ld(b) or a.shape != b.shape:
            raise ValueError('add: shape/stride mismatch')
        y = new_nhwc(n, c, h, w, a.device, ld=_ld(a))
        call('ssg_add_f32', ptr(a), ptr(b), n * h * w * _ld(a), ptr(y), stream_ptr())
        return y

    @staticmethod
    def backward(ctx, g):
        return g, g


def add(a, b):
    """a + b for two NHWC tensors of the same shape."""
    _lib.require_gpu(a)
    return _Add.apply(a, b)


class _ConcatChannels(torch.autograd.Function):
    @staticmethod
    def forward(ctx, *xs):
        xs = [to_nhwc(x) for x in xs]
        n, _, h, w = xs[0].shape
        cs = [x.shape[1] for x in xs]
        if any(c % 4 for c in cs):
            raise ValueError('concat_channels: every input needs C %% 4 == 0 (got %s)' % cs)
        y = new_nhwc(n, sum(cs), h, w, xs[0].device)
        off = 0
        for x, c in zip(xs, cs):
            dst = y[:, off:off + c]
            call('ssg_copy_channels_f32', ptr(x), _ld(x), n * h * w, c, ptr(dst), _ld(y), stream_ptr())
            off += c
        ctx.cs = cs
        return y

    @staticmethod
    def backward(ctx, g):
        outs, off = [], 0
        for c in ctx.cs:
            outs.append(g[:, off:off + c])
            off += c
        return tuple(outs)


def concat_channels(*xs):
    """torch.cat(xs, 1) materialised (only needed for more than two inputs; convs take two pointers)."""
    _lib.require_gpu(xs[0])
    return xs[0] if len(xs) == 1 else _ConcatChannels.apply(*xs)
