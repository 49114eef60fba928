"""Spectral normalisation, MI355X-native (mirrors the reference's scripts/spectral_norm.py, a vendored
torch.nn.utils.spectral_norm; unwired in the reference -- SURVEY.md 8a row A12).

Same surface: `spectral_norm(module, name='weight', n_power_iterations=1, eps=1e-12, dim=None)`
re-parametrises `module.<name>` as `<name>_orig / sigma` with buffers `<name>_u`, `<name>_v` (same
state_dict keys), one in-place power iteration per training forward (spectral_norm.py:73-88) and none
in eval mode.  The GEMVs, the normalisations and W/sigma run on the HIP kernels (csrc/spectral.hip)."""
import torch
import torch.nn.functional as F
from torch.nn.parameter import Parameter

from . import ops


class SpectralNorm(object):
    # spectral_norm.py:13-18: version 1 = `weight` is recomputed from `weight_orig`, `weight_u`, `weight_v` in every forward
    # and is not part of the state_dict; an unversioned checkpoint also carries `weight` and no `weight_v`.
    _version = 1

    def __init__(self, name='weight', n_power_iterations=1, dim=0, eps=1e-12):
        if n_power_iterations <= 0:
            raise ValueError('Expected n_power_iterations to be positive, but got n_power_iterations={}'.format(n_power_iterations))
        self.name, self.dim, self.n_power_iterations, self.eps = name, dim, n_power_iterations, eps

    def reshape_weight_to_matrix(self, weight):
        if self.dim != 0:
            weight = weight.permute(self.dim, *[d for d in range(weight.dim()) if d != self.dim])
        return weight.reshape(weight.size(0), -1)

    def compute_weight(self, module, do_power_iteration):
        weight = getattr(module, self.name + '_orig')
        u = getattr(module, self.name + '_u')
        v = getattr(module, self.name + '_v')
        if self.dim != 0:
            raise NotImplementedError('spectral_norm with dim != 0 (ConvTranspose) has no HIP path')
        if weight.is_cuda:
            w_sn, _ = ops.spectral_norm_weight(weight, u, v, self.n_power_iterations if do_power_iteration else 0, self.eps)
            return w_sn
        raise RuntimeError('ssunet-gan_amd spectral_norm runs only on a HIP device (no CPU fallback)')

    def __call__(self, module, inputs):
        setattr(module, self.name, self.compute_weight(module, do_power_iteration=module.training))

    def remove(self, module):
        with torch.no_grad():
            weight = self.compute_weight(module, do_power_iteration=False)
        delattr(module, self.name)
        delattr(module, self.name + '_u')
        delattr(module, self.name + '_v')
        delattr(module, self.name + '_orig')
        module.register_parameter(self.name, Parameter(weight.detach()))

    @staticmethod
    def apply(module, name, n_power_iterations, dim, eps):
        for hook in module._forward_pre_hooks.values():
            if isinstance(hook, SpectralNorm) and hook.name == name:
                raise RuntimeError('Cannot register two spectral_norm hooks on the same parameter {}'.format(name))
        fn = SpectralNorm(name, n_power_iterations, dim, eps)
        weight = module._parameters[name]
        with torch.no_grad():
            weight_mat = fn.reshape_weight_to_matrix(weight)
            h, w = weight_mat.size()
            # same RNG consumption as the reference: u then v from normal_(0, 1) (spectral_norm.py:118-121)
            u = F.normalize(weight.new_empty(h).normal_(0, 1), dim=0, eps=fn.eps)
            v = F.normalize(weight.new_empty(w).normal_(0, 1), dim=0, eps=fn.eps)
        delattr(module, fn.name)
        module.register_parameter(fn.name + '_orig', weight)
        setattr(module, fn.name, weight.data)
        module.register_buffer(fn.name + '_u', u)
        module.register_buffer(fn.name + '_v', v)
        module.register_forward_pre_hook(fn)
        module._register_state_dict_hook(_StateDictVersion(fn))
        module._register_load_state_dict_pre_hook(_LoadUnversioned(fn))
        return fn


class _StateDictVersion(object):
    """spectral_norm.py:179-189: stamp `<name>.version` into the module's state_dict metadata."""

    def __init__(self, fn):
        self.fn = fn

    def __call__(self, module, state_dict, prefix, local_metadata):
        meta = local_metadata.setdefault('spectral_norm', {})
        key = self.fn.name + '.version'
        if key in meta:
            raise RuntimeError("Unexpected key in metadata['spectral_norm']: {}".format(key))
        meta[key] = self.fn._version


class _LoadUnversioned(object):
    """spectral_norm.py:147-174: a checkpoint written before version 1 holds (weight_orig, weight, weight_u) and no weight_v.
    Recover v from the invariant u = normalize(W_orig v), sigma = u^T W_orig v with sigma = mean(W_orig / W) (pseudo-inverse
    solve, spectral_norm.py:102-107), and drop the stale `weight` entry.  Host-side checkpoint conversion: plain torch."""

    def __init__(self, fn):
        self.fn = fn

    def __call__(self, state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys, error_msgs):
        fn = self.fn
        version = local_metadata.get('spectral_norm', {}).get(fn.name + '.version', None)
        if version is not None and version >= 1:
            return
        with torch.no_grad():
            w_orig = state_dict[prefix + fn.name + '_orig']
            w = state_dict.pop(prefix + fn.name)
            sigma = (w_orig / w).mean()
            wm = fn.reshape_weight_to_matrix(w_orig)
            u = state_dict[prefix + fn.name + '_u']
            v = torch.linalg.multi_dot([torch.linalg.pinv(wm.t().mm(wm)), wm.t(), u.unsqueeze(1)]).squeeze(1)
            state_dict[prefix + fn.name + '_v'] = v * (sigma / torch.dot(u, torch.mv(wm, v)))


def spectral_norm(module, name='weight', n_power_iterations=1, eps=1e-12, dim=None):
    if dim is None:
        dim = 1 if isinstance(module, (torch.nn.ConvTranspose1d, torch.nn.ConvTranspose2d, torch.nn.ConvTranspose3d)) else 0
    SpectralNorm.apply(module, name, n_power_iterations, dim, eps)
    return module


def remove_spectral_norm(module, name='weight'):
    for k, hook in module._forward_pre_hooks.items():
        if isinstance(hook, SpectralNorm) and hook.name == name:
            hook.remove(module)
            del module._forward_pre_hooks[k]
            return module
    raise ValueError("spectral_norm of '{}' not found in {}".format(name, module))
