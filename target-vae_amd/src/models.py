"""Drop-in `src.models` for TARGET-VAE on MI355X.

Same class names, constructor signatures, attribute / parameter names and default initialisation order
as the reference `src/models.py` (so `state_dict`s and whole-module pickles, reference
train_mnist.py:672-681 / src/utils.py:37-48, are interchangeable), but `forward()` runs the hand-written
gfx950 kernels of libtvae_hip.so through `tvae.ops`.  There is no CPU compute path: calling a hot-path
module on CPU tensors raises.
"""
from __future__ import annotations

import math

import numpy as np
import torch
import torch.nn as nn
from torch.nn.modules.utils import _pair

from tvae import ops as _ops


def _require_gpu(t, who):
    if not t.is_cuda:
        raise RuntimeError(f'{who}: the MI355X build has no CPU compute path (tensor is on {t.device}); '
                           'move the model and data to the GPU')


class ResidLinear(nn.Module):
    """act(linear(x) + x)  (reference src/models.py:22-30).  Inside SpatialGenerator it is executed by the
    fused decoder kernels; standalone it is a thin torch module."""

    def __init__(self, n_in, n_out, activation=nn.LeakyReLU):
        super().__init__()
        self.linear = nn.Linear(n_in, n_out)
        self.act = activation()

    def forward(self, x):
        return self.act(self.linear(x) + x)


class RandomFourierEmbedding2d(nn.Module):
    """cos(x (W/sigma)^T + b) with fixed random W ~ N(0,1), b ~ U(0, 2pi) (reference src/models.py:33-58).
    `sigma` stays a plain tensor attribute (not a buffer) like the reference, for pickle compatibility."""

    def __init__(self, in_dim, embedding_dim, sigma=0.01):
        super().__init__()
        self.in_dim = in_dim
        self.embedding_dim = embedding_dim
        self.sigma = torch.tensor(sigma, dtype=torch.float32)
        self.register_buffer('weight', torch.randn(embedding_dim, in_dim))
        self.register_buffer('bias', torch.rand(embedding_dim) * 2 * np.pi)
        print('# sigma value is {}'.format(self.sigma))

    def forward(self, x):
        if x is None:
            return 0
        _require_gpu(x, 'RandomFourierEmbedding2d')
        x2 = x.reshape(-1, 2).contiguous()
        n = x2.shape[0]
        feat = torch.empty(self.embedding_dim, n, dtype=torch.float32, device=x.device)
        _ops.call('tvae_fourier_fwd', x2, self.weight.contiguous(), self.bias.contiguous(), float(self.sigma), feat, n,
                  self.embedding_dim, n)
        return feat.t().reshape(*x.shape[:-1], self.embedding_dim)


class SpatialGenerator(nn.Module):
    """Per-pixel coordinate MLP decoder (reference src/models.py:65-123)."""

    def __init__(self, latent_dim, hidden_dim, n_out=1, num_layers=1, activation=nn.LeakyReLU, resid=False,
                 fourier_expansion=False, sigma=0.01):
        super().__init__()
        self.fourier_expansion = fourier_expansion
        in_dim = 2
        if fourier_expansion:
            self.embed_latent = RandomFourierEmbedding2d(in_dim, 1024, sigma)
            in_dim = 1024
        self.coord_linear = nn.Linear(in_dim, hidden_dim)
        self.latent_dim = latent_dim
        if latent_dim > 0:
            self.latent_linear = nn.Linear(latent_dim, hidden_dim, bias=False)
        stack = [activation()]
        for _ in range(1, num_layers):
            if resid:
                stack.append(ResidLinear(hidden_dim, hidden_dim, activation=activation))
            else:
                stack.append(nn.Linear(hidden_dim, hidden_dim))
                stack.append(activation())
        stack.append(nn.Linear(hidden_dim, n_out))
        self.layers = nn.Sequential(*stack)

    def _plan(self):
        """(act code, resid flag, [hidden linear modules], output linear) read off `self.layers`."""
        mods = list(self.layers)
        act = _ops.act_code(mods[0])
        hidden, resid = [], False
        for m in mods[1:-1]:
            if isinstance(m, ResidLinear):
                hidden.append(m.linear)
                resid = True
            elif isinstance(m, nn.Linear):
                hidden.append(m)
        return act, resid, hidden, mods[-1]

    def forward(self, x, z):
        _require_gpu(x, 'SpatialGenerator')
        if x.dim() < 3:
            x = x.unsqueeze(0)
        has_l = hasattr(self, 'latent_linear')
        if has_l and z.dim() < 2:
            z = z.unsqueeze(0)
        act, resid, hidden, out = self._plan()
        params = [self.coord_linear.weight, self.coord_linear.bias, self.latent_linear.weight if has_l else None]
        for m in hidden:
            params += [m.weight, m.bias]
        params += [out.weight, out.bias]
        sigma = 0.0
        if self.fourier_expansion:
            params += [self.embed_latent.weight, self.embed_latent.bias]
            sigma = float(self.embed_latent.sigma)
        else:
            params += [None, None]
        # pixel counts that are not a multiple of the GEMM tile (28 x 28, 50 x 50): pad every image's pixel range so that the
        # decoder's fast path applies (tvae/ops.py: decoder_padded_pixels); the padded outputs are sliced away
        Np = x.shape[1]
        Np_p = _ops.decoder_padded_pixels(Np, x.shape[0], self.coord_linear.out_features, len(hidden), out.out_features, resid,
                                          self.fourier_expansion, act)
        if Np_p:
            x = torch.nn.functional.pad(x, (0, 0, 0, Np_p - Np))
        # under torch.no_grad() (eval_model, train_mnist.py:352-387) the inference-mode forward runs: nothing is kept for a backward
        with _ops.inference(not _ops.needs_grad(x, z if has_l else None, *params)):
            y = _ops.DecoderFn.apply(x, z if has_l else None, act, resid, sigma, len(hidden), *params)
        return y[:, :Np] if Np_p else y


class GroupConv(nn.Module):
    """P_n lifting convolution (reference src/models.py:132-225): R rotated copies of every filter, one dense
    correlation with C*R output channels, bias shared over rotations."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, bias=True, input_rot_dim=1,
                 output_rot_dim=4):
        super().__init__()
        self.ksize = kernel_size
        self.in_channels = in_channels
        self.out_channels = out_channels
        self.kernel_size = _pair(kernel_size)
        self.stride = _pair(stride)
        self.padding = _pair(padding)
        self.input_rot_dim = input_rot_dim
        self.output_rot_dim = output_rot_dim
        self.weight = nn.Parameter(torch.Tensor(out_channels, in_channels, input_rot_dim, *self.kernel_size))
        if bias:
            self.bias = nn.Parameter(torch.Tensor(out_channels))
        else:
            self.register_parameter('bias', None)
        self.reset_parameters()

    def reset_parameters(self):
        fan = self.in_channels * self.kernel_size[0] * self.kernel_size[1]
        bound = 1.0 / math.sqrt(fan)
        self.weight.data.uniform_(-bound, bound)
        if self.bias is not None:
            self.bias.data.uniform_(-bound, bound)

    def _check(self):
        if self.stride != (1, 1) or self.input_rot_dim != 1 or self.kernel_size[0] != self.kernel_size[1]:
            raise NotImplementedError('HIP GroupConv: stride 1, square kernel, input_rot_dim 1 (all the reference uses)')

    def trans_filter(self, device):
        """Rotated bank in the reference layout (C, R, Cin, 1, k, k)."""
        self._check()
        _require_gpu(self.weight, 'GroupConv.trans_filter')
        C, Cin, _, k, _ = self.weight.shape
        R = self.output_rot_dim
        return _ops.BankFn.apply(self.weight, R).view(C, R, Cin, 1, k, k)

    def forward(self, input, device):
        self._check()
        _require_gpu(input, 'GroupConv')
        return _ops.GroupConvFn.apply(input, self.weight, self.bias, self.output_rot_dim, self.padding[0])


class InferenceNetwork_UnimodalTranslation_UnimodalRotation(nn.Module):
    """MLP encoder without attention (reference src/models.py:229-260); secondary encoder: the Linear /
    ResidLinear stack runs on the GEMM kernels (`ops.MlpFn`); `self.layers` keeps the reference's parameter names."""

    def __init__(self, n, latent_dim, hidden_dim, num_layers=1, activation=nn.LeakyReLU, resid=False):
        super().__init__()
        self.latent_dim = latent_dim
        self.n = n
        print('n is {}'.format(n))
        stack = [nn.Linear(n, hidden_dim), activation()]
        for _ in range(1, num_layers):
            if resid:
                stack.append(ResidLinear(hidden_dim, hidden_dim, activation=activation))
            else:
                stack.append(nn.Linear(hidden_dim, hidden_dim))
                stack.append(activation())
        stack.append(nn.Linear(hidden_dim, 2 * latent_dim))
        self.layers = nn.Sequential(*stack)

    def forward(self, x):
        _require_gpu(x, 'InferenceNetwork_UnimodalTranslation_UnimodalRotation')
        lins = [(m.linear, True) if isinstance(m, ResidLinear) else (m, False) for m in self.layers
                if isinstance(m, (nn.Linear, ResidLinear))]
        params = [t for lin, _ in lins for t in (lin.weight, lin.bias)]
        out = _ops.MlpFn.apply(x.float(), _ops.act_code(self.layers[1]), tuple(r for _, r in lins), *params)
        return out[:, :self.latent_dim], out[:, self.latent_dim:]


class InferenceNetwork_AttentionTranslation_UnimodalRotation(nn.Module):
    """Translation-attention encoder, rotation pooled by fc_r (reference src/models.py:268-319); secondary
    encoder: the whole encoder on the HIP kernels, the pooled-posterior tail in
    tvae/secondary.py on generic torch."""

    def __init__(self, n, in_channels, latent_dim, kernels_num=128, activation=nn.LeakyReLU, groupconv=0):
        super().__init__()
        self.activation = activation()
        self.latent_dim = latent_dim
        self.input_size = n
        self.kernels_num = kernels_num
        self.groupconv = groupconv
        if groupconv == 0:
            self.conv1 = nn.Conv2d(in_channels, kernels_num, n, padding=n // 2)
            self.conv2 = nn.Conv2d(kernels_num, kernels_num, 1)
        else:
            self.conv1 = GroupConv(in_channels, kernels_num, n, padding=n // 2, input_rot_dim=1,
                                   output_rot_dim=groupconv)
            self.conv2 = nn.Conv2d(kernels_num, kernels_num, 1)
            self.fc_r = nn.Linear(groupconv, 1)
        self.conv_a = nn.Conv2d(kernels_num, 1, 1)
        self.conv_r = nn.Conv2d(kernels_num, 2, 1)
        self.conv_z = nn.Conv2d(kernels_num, 2 * latent_dim, 1)

    def forward(self, x, device, E=None):
        """Reference 4-tuple (attn, a_sampled, theta, z).  `E` optionally injects the Exp(1) draws of the
        Gumbel-softmax (reference: F.gumbel_softmax, models.py:311).  conv1 (lifting or plain convolution), the fc_r
        rotation pooling, conv2 and the three 1x1 heads run on the HIP kernels (`ops.TransAttnEncoderFn`)."""
        _require_gpu(x, 'InferenceNetwork_AttentionTranslation_UnimodalRotation')
        zd = self.latent_dim
        C = self.kernels_num
        Wh = torch.cat([self.conv_a.weight.view(1, C), self.conv_r.weight.view(2, C), self.conv_z.weight.view(2 * zd, C)], 0)
        bh = torch.cat([self.conv_a.bias, self.conv_r.bias, self.conv_z.bias], 0)
        gc = self.groupconv
        fw, fb = (self.fc_r.weight, self.fc_r.bias) if gc > 0 else (None, None)
        heads = _ops.TransAttnEncoderFn.apply(x, self.conv1.weight, self.conv1.bias, fw, fb, self.conv2.weight.view(C, C),
                                              self.conv2.bias, Wh, bh, gc, self.input_size // 2,
                                              _ops.act_code(self.activation))
        b = x.shape[0]
        Ho = int(round((heads.shape[1] // b) ** 0.5))
        hv = heads.view(3 + 2 * zd, b, Ho, Ho).permute(1, 0, 2, 3)
        attn = hv[:, 0:1]
        logits = attn.reshape(b, -1)
        if E is None:
            E = torch.empty_like(logits).exponential_()
        a = torch.softmax(logits - torch.log(E.reshape(logits.shape)), dim=-1)
        return attn, a.view(b, Ho, Ho), hv[:, 1:3], hv[:, 3:]


class InferenceNetwork_AttentionTranslation_AttentionRotation(nn.Module):
    """TARGET-VAE inference network: attention over translation AND rotation (reference src/models.py:326-403)."""

    def __init__(self, n, in_channels, latent_dim, kernels_num=128, kernels_size=65, padding=16,
                 activation=nn.LeakyReLU, groupconv=0, rot_refinement=False, theta_prior=np.pi,
                 normal_prior_over_r=True):
        super().__init__()
        self.activation = activation()
        self.latent_dim = latent_dim
        self.input_size = n
        self.kernels_num = kernels_num
        self.kernels_size = kernels_size
        self.padding = padding
        self.groupconv = groupconv
        self.rot_refinement = rot_refinement
        self.theta_prior = theta_prior
        self.normal_prior_over_r = normal_prior_over_r
        self.conv1 = GroupConv(in_channels, kernels_num, kernels_size, padding=padding, input_rot_dim=1,
                               output_rot_dim=groupconv)
        self.conv2 = nn.Conv3d(kernels_num, kernels_num, 1)
        self.conv_a = nn.Conv3d(kernels_num, 1, 1)
        self.conv_r = nn.Conv3d(kernels_num, 2, 1)
        self.conv_z = nn.Conv3d(kernels_num, 2 * latent_dim, 1)

    # -- helpers shared with the fused training step (tvae/step.py) -------------------------------
    def head_weights(self):
        """conv_a / conv_r / conv_z stacked as one (3+2z, C) matrix + bias (rows: logit, theta_mu,
        theta_logstd, z_mu.., z_logstd..)."""
        C = self.kernels_num
        Wh = torch.cat([self.conv_a.weight.view(1, C), self.conv_r.weight.view(2, C),
                        self.conv_z.weight.view(2 * self.latent_dim, C)], 0)
        bh = torch.cat([self.conv_a.bias, self.conv_r.bias, self.conv_z.bias], 0)
        return Wh, bh

    def output_size(self):
        return self.input_size + 2 * self.padding - self.kernels_size + 1

    def head_tables(self, device, spacing=None):
        """Device constants of the attention head, cached per (device, spacing)."""
        if spacing is None:
            spacing = float(np.float32(2.0 / (self.input_size - 1)))
        key = (str(device), float(spacing))
        cache = self.__dict__.setdefault('_tb_cache', {})
        if key not in cache:
            cache[key] = _ops.HeadTables(self.groupconv, self.output_size(), spacing, self.rot_refinement,
                                         self.theta_prior, self.normal_prior_over_r, device)
        return cache[key]

    def __getstate__(self):
        st = dict(self.__dict__)
        st.pop('_tb_cache', None)       # device tables are not part of a checkpoint
        return st

    def encode_heads(self, x):
        """Feature-major head tensor [3+2z][B*R*Ho*Ho] (conv1 -> act -> conv2 -> act -> heads)."""
        _require_gpu(x, 'InferenceNetwork_AttentionTranslation_AttentionRotation')
        C = self.kernels_num
        Wh, bh = self.head_weights()
        # under torch.no_grad() (eval_model, get_latent) the inference-mode forward runs: no conv2 activation, no sign words
        with _ops.inference(not _ops.needs_grad(x, self.conv1.weight, self.conv1.bias, self.conv2.weight, self.conv2.bias,
                                                Wh, bh)):
            return _ops.EncoderFn.apply(x, self.conv1.weight, self.conv1.bias, self.conv2.weight.view(C, C),
                                        self.conv2.bias, Wh, bh, self.groupconv, self.padding,
                                        _ops.act_code(self.activation))

    def forward(self, x, device, E=None):
        """Reference 7-tuple (attn, q_t_r, p_r, a_sampled, offsets, theta, z).  `E` optionally injects the
        Exp(1) draws of the Gumbel-softmax (reference draws them inside F.gumbel_softmax, models.py:387)."""
        B = x.shape[0]
        R, Ho, zd = self.groupconv, self.output_size(), self.latent_dim
        heads = self.encode_heads(x)
        tb = self.head_tables(x.device)
        if E is None:
            E = torch.empty(B, R * Ho * Ho, dtype=torch.float32, device=x.device).exponential_()
        zeros_z = torch.zeros(B, zd, dtype=torch.float32, device=x.device)
        zeros_t = torch.zeros(B, dtype=torch.float32, device=x.device)
        attn, q, a, _, _, _, _ = _ops.HeadFn.apply(heads, E.reshape(B, -1), zeros_z, zeros_t, tb, B, zd)
        hv = heads.view(3 + 2 * zd, B, R, Ho, Ho)
        theta = hv[1:3].permute(1, 0, 2, 3, 4)
        z = hv[3:].permute(1, 0, 2, 3, 4)
        if self.rot_refinement:
            offsets = tb.off
            theta = torch.stack((theta[:, 0] + offsets.view(1, R, 1, 1), theta[:, 1]), dim=1)
        else:
            offsets = torch.zeros(R, dtype=torch.float32, device=x.device)
        return (attn.view(B, R, Ho, Ho), q.view(B, R, Ho, Ho), tb.p_r.view(R, 1, 1), a.view(B, R, Ho, Ho), offsets,
                theta, z)
