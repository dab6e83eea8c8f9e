"""torch.autograd.Function wrappers over the libtvae_hip.so C ABI (include/tvae_hip.h).

Internal activation layout is feature-major: [feature][image*positions + position].  Each Function
mirrors one stage of the reference hot path and cites the reference lines it replaces.  All compute
happens in the HIP library on the current torch stream; torch only provides device memory.
"""
from __future__ import annotations

import math
import os
from typing import Optional

import numpy as np
import torch

from . import tables
from ._lib import arithmetic, call, get_gemm_mode, parts, query, split_pipe

ACT_NONE, ACT_LRELU, ACT_TANH = 0, 1, 2
LRELU_SLOPE = 0.01


def _expect(cond: bool, what: str) -> None:
    """Operand shapes are validated on the host before any launch: a kernel must never see a buffer smaller than
    its grid assumes (an out-of-bounds access can reset the GPU)."""
    if not cond:
        raise ValueError('tvae.ops: ' + what)


def act_code(activation) -> int:
    """Map an nn activation class/instance (reference ctor argument `activation`) to the kernel code."""
    import torch.nn as nn
    a = activation() if isinstance(activation, type) else activation
    if isinstance(a, nn.LeakyReLU):
        if abs(a.negative_slope - LRELU_SLOPE) > 0:
            raise NotImplementedError('only the default LeakyReLU slope 0.01 is built')
        return ACT_LRELU
    if isinstance(a, nn.Tanh):
        return ACT_TANH
    raise NotImplementedError(f'activation {type(a).__name__} has no HIP kernel (reference CLI: tanh | leakyrelu)')


# ---------------------------------------------------------------------------------------------
# optional in-run kernel timing (bench.py): events are recorded on the stream the kernels are launched on
# ---------------------------------------------------------------------------------------------
KERNEL_EVENTS = None     # set to {} to record (start, end) torch.cuda.Event pairs per entry point
# set to {} to record, per split-pipe entry point, (parts, MFMAs per product block) of EVERY launch as it was actually
# issued (not as the mode would suggest): bench.py derives its executed-FLOP figures from it, the tests assert on it
PARTS_LOG = None


def mfma_per_block(nparts: int, two_valued: bool = False) -> int:
    """Matrix instructions per 32x32x16 product block of a split-pipe launch: 6 (three bf16 parts), 3 (two fp16 parts, h3),
    1 (bf16); against the exact 0 / 1 operand of the two-valued gradient only the other operand's parts count (3 / 2 / 1)."""
    return {3: 3 if two_valued else 6, 2: 2 if two_valued else 3, 1: 1}[nparts]


class _timed:
    def __init__(self, name, nparts=None, two_valued=False):
        self.name = name
        self.nparts = nparts
        self.products = mfma_per_block(nparts, two_valued) if nparts is not None else None

    def __enter__(self):
        if PARTS_LOG is not None and self.nparts is not None:
            PARTS_LOG.setdefault(self.name, []).append((self.nparts, self.products))
        if KERNEL_EVENTS is not None:
            self.s = torch.cuda.Event(enable_timing=True)
            self.e = torch.cuda.Event(enable_timing=True)
            self.s.record()
        return self

    def __exit__(self, *exc):
        if KERNEL_EVENTS is not None:
            self.e.record()
            KERNEL_EVENTS.setdefault(self.name, []).append((self.s, self.e, self.products))
        return False


def _in_forward_arithmetic(backward):
    """Run an autograd backward in the arithmetic its forward ran in (`ctx.arith`): the fused branches a forward takes
    decide what it saves, so the backward must route the same way even when it executes outside the caller's
    `arithmetic(...)` block."""
    def wrapped(ctx, *grads):
        with arithmetic(ctx.arith):
            return backward(ctx, *grads)
    wrapped.__doc__ = backward.__doc__
    return wrapped


PATH_LOG = None          # set to a set() to record which fused branches a step actually took (tests assert on it)

# Inference-mode forward (reference: eval_model runs under torch.no_grad(), train_mnist.py:352-387; get_latent,
# clustering_mnist.py:121-161).  Inside torch.autograd.Function.forward grad mode is ALWAYS off, so the decision is taken by
# the caller (`needs_grad` over the op's differentiable inputs, src/models.py) and handed down through this switch: the
# forward then writes nothing that only a backward would read -- no conv2 activation H (1.14 GB at 64x64 / 256), no sign
# words, no decoder sign bits, no retained spectra -- and saves nothing on ctx.
_INFER = False


def needs_grad(*tensors) -> bool:
    return torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in tensors)


class inference:
    """`with ops.inference(flag):` -- the Functions applied inside run their inference-mode forward when `flag`."""

    def __init__(self, on: bool):
        self.on = bool(on)

    def __enter__(self):
        global _INFER
        self.prev, _INFER = _INFER, self.on
        return self

    def __exit__(self, *exc):
        global _INFER
        _INFER = self.prev
        return False


def _note(name: str) -> None:
    if PATH_LOG is not None:
        PATH_LOG.add(name)


def kernel_event_ms():
    """Mean milliseconds per launch for every recorded entry point (call after torch.cuda.synchronize())."""
    out = {}
    for k, v in (KERNEL_EVENTS or {}).items():
        ts = [s.elapsed_time(e) for s, e, _ in v]
        pr = [p_ for _, _, p_ in v if p_ is not None]
        out[k] = dict(launches=len(ts), mean_ms=sum(ts) / max(len(ts), 1), total_ms=sum(ts),
                      # MFMAs per product block as launched: time-weighted mean (all launches of an entry do the same
                      # algorithmic work in the bench) and the distinct values seen
                      mfma_per_block=(sum(p_ * t for (_, _, p_), t in zip(v, ts) if p_ is not None) /
                                      max(sum(t for (_, _, p_), t in zip(v, ts) if p_ is not None), 1e-30)) if pr else None,
                      mfma_per_block_seen=sorted(set(pr)))
    return out


# ---------------------------------------------------------------------------------------------
# workspaces and device tables (cached per device)
# ---------------------------------------------------------------------------------------------
_WS = {}
_TABLES = {}
# Buffers a captured hipGraph holds raw pointers to (tvae/graph.py calls pin_scratch() after capture).  Growing a scratch
# buffer REPLACES its tensor; without this list the old block would go back to the caching allocator while the graph still
# writes into it on every replay (a later, larger eager batch -- an uneven shard, a test set larger than the captured
# minibatch -- would trigger exactly that: ADVICE r03).  Once pinned, outgrown buffers are kept alive instead of freed: the
# graph keeps its blocks, eager calls get the new, larger ones.
_PINNED = {}             # pin token -> outgrown buffers kept alive for that graph


def pin_scratch() -> int:
    """Called after a capture: returns a token; until `unpin_scratch(token)` every scratch buffer that is outgrown is kept
    alive (the graph holds raw pointers into the blocks that existed at capture time)."""
    tok = (max(_PINNED) + 1) if _PINNED else 1
    _PINNED[tok] = []
    return tok


def unpin_scratch(token: int) -> None:
    _PINNED.pop(token, None)


def _replace_ws(key, new):
    old = _WS.get(key)
    if old is not None:
        for lst in _PINNED.values():
            lst.append(old)
    _WS[key] = new


# debugging aid (TVAE_POISON_WS=1; off by default, costs a fill per call): every workspace view handed out, and every named
# scratch buffer when it is (re)allocated, is filled with NaN first -- a kernel that reads what no kernel of the same call
# wrote then shows up as NaN instead of depending on what an earlier, larger problem left there
POISON_WS = os.environ.get('TVAE_POISON_WS', '0') == '1'


def workspace(device, floats: int) -> torch.Tensor:
    """Split-K scratch (grown on demand, reused across calls on the same stream).  The caller gets a view of EXACTLY the
    size it asked for: several entry points cap their number of reduction slices by the workspace they are handed, so
    passing "whatever the shared buffer has grown to" would make the summation order of a step depend on which other call
    had run before it (the first step of a process differed from every later one by 2e-7 on one tensor: round 3)."""
    key = (device.type, device.index)
    t = _WS.get(key)
    if t is None or t.numel() < floats:
        t = torch.empty(int(floats), dtype=torch.float32, device=device)
        _replace_ws(key, t)
    if POISON_WS:
        t[:int(floats)].fill_(float('nan'))
    return t[:int(floats)]


def _dev_table(key, device, builder):
    k = (key, device.type, device.index)
    t = _TABLES.get(k)
    if t is None:
        t = tuple(torch.from_numpy(np.ascontiguousarray(a)).to(device) for a in builder())
        _TABLES[k] = t
    return t


def tap_tables(k: int, R: int, device):
    return _dev_table(('taps', k, R), device, lambda: tables.rotation_taps(k, R))


def tap_tables_csr(k: int, R: int, device):
    return _dev_table(('csr', k, R), device, lambda: tables.rotation_taps_csr(k, R))


SKINNY_MAX = 8          # head projections with <= 8 output rows use the streaming kernels, wider ones the MFMA GEMM


def _seglen(N: int) -> int:
    # (a multiple of 4: every segment of a 16-byte aligned row then starts 16-byte aligned -- the float4 path of rowdot_seg)
    return max(2048, ((N + 255) // 256 + 3) // 4 * 4)


def _rowsum(X: torch.Tensor, M: int, N: int, out: Optional[torch.Tensor] = None, amax: Optional[torch.Tensor] = None,
            ld: int = 0) -> torch.Tensor:
    """out[m] = sum_n X[m][n] (deterministic two-level reduction).  amax (a zeroed device word): also receives max |X|.
    ld: floats between rows of X (0 = N)."""
    sl = _seglen(N)
    nseg = (N + sl - 1) // sl
    tmp = torch.empty(nseg, M, dtype=torch.float32, device=X.device)
    call('tvae_rowdot_seg', X, ld or N, None, 1, M, N, sl, tmp, amax)
    if out is None:
        out = torch.empty(M, dtype=torch.float32, device=X.device)
    call('tvae_seg_sum', tmp, nseg, M, out, 1.0, 0)
    return out


def _dense_x6_ok(rows: int, N: int) -> bool:
    """Dense layers on the split bf16 pipe: wide layers only (the 512-row tile would waste a 128-row layer)."""
    return split_pipe() and N % 128 == 0 and rows >= 256


def _p3() -> int:
    """parts for an entry point / operand form WITHOUT an h3 instance: the h3 mode runs those in the exact x6 arithmetic."""
    p = parts()
    return 3 if p == 2 else p


def _inf_norm(t: torch.Tensor) -> torch.Tensor:
    """max |t| as a one-element device tensor (one reduction launch, no temporary, no host synchronisation): the h3 scale
    word of an operand that is streamed from memory (include/tvae_hip.h: x_amax)."""
    return torch.linalg.vector_norm(t.reshape(-1), float('inf')).reshape(1)


def _split_weight(W: torch.Tensor, rows: int, K: int, transpose: bool, key: str, scale=None, nparts: int = 3):
    """W (out, in) -> fragment-ready cells for A(row, k) = W[row][k] (forward) or W[k][row] (dgrad): 3 x bf16 parts, or
    (nparts = 2) the 2 x fp16 parts of the h3 arithmetic.
    scale (K floats): A(row, k) *= scale[k] before the split; then also returns the row sums of the scaled operand
    (the two-valued implicit-gradient form of tvae_linear_dgrad_x6, include/tvae_hip.h)."""
    W = W.contiguous()
    w3 = _scratch(W.device, key, query('tvae_dense_x6_bytes', rows, K) // 4)
    csum = torch.empty(rows, dtype=torch.float32, device=W.device) if scale is not None else None
    call('tvae_dense_split2h' if nparts == 2 else 'tvae_dense_split3', W, W.shape[1], w3, w3.numel() * 4, rows, K,
         1 if transpose else 0, scale, csum)
    return w3 if scale is None else (w3, csum)


def _wgrad(dpre, X, M, N, K, virt=None, va=None, act=0, bits=None, rowdot_w=None, a_amax=None, x_amax=None, ld=0):
    """dW = dpre . X^T.  virt = (wo, gy, act): dpre is the saved activation H and the gradient wo[m]*gy[n]*act'(H) is formed
    on the fly; va = (xr, Wc, bc, LB, Np): X is the coordinate layer's output act(..), recomputed (split-pipe path only).
    bits: [H > 0] as stored sign bits (dpre may then be None).  rowdot_w = the layer's own weight [M][K]: also returns
    rowdot[m] = sum_k W[m][k] dW[m][k] / wo[m] (taken before the multiplication), as (dW, rowdot)."""
    dev_ = (dpre if dpre is not None else (X if X is not None else virt[0])).device
    dW = torch.empty(M, K, dtype=torch.float32, device=dev_)
    need = 64 * max(M, 128) * max(K, 128)
    if split_pipe() and M >= 256 and K >= 128 and N % 16 == 0 and N >= 32:
        ws = workspace(dev_, max(query('tvae_linear_wgrad_x6_ws_floats', M, N, K), 1 << 24))
        # h3 instance: the two-valued form from sign bits against the recomputed first-layer operand (two products per block)
        # ... or (round 4) against an operand from memory whose bound the caller supplies (x_amax), or two plain operands
        # from memory with both bounds (a_amax, x_amax)
        lrf_bits = bits is not None and virt and virt[2] == ACT_LRELU
        p = 2 if (parts() == 2 and ((lrf_bits and (va or x_amax is not None)) or
                                    (virt is None and va is None and a_amax is not None and x_amax is not None))) else _p3()
        rowdot = torch.empty(M, dtype=torch.float32, device=dev_) if rowdot_w is not None else None
        with _timed('tvae_linear_wgrad_x6', p, bool(virt) and virt[2] == ACT_LRELU):
            call('tvae_linear_wgrad_x6', dpre, X, dW, ws, ws.numel(), M, N, K, ld or N, ld or N, 0,
                 virt[0] if virt else None, virt[1] if virt else None, virt[2] if virt else act, LRELU_SLOPE,
                 *(va if va else (None, None, None, None, 0)), bits, p,
                 rowdot_w.contiguous() if rowdot_w is not None else None, K, rowdot,
                 a_amax if p == 2 else None, x_amax if p == 2 else None,
                 1 if (p == 2 and x_amax is not None and x_amax.numel() == K and K > 1) else 0)      # one bound per row of X
        return dW if rowdot_w is None else (dW, rowdot)
    _expect(virt is None and va is None and rowdot_w is None, 'implicit operands need the split-pipe weight gradient')
    ws = workspace(dev_, max(need, 1 << 24))
    call('tvae_linear_wgrad', dpre, X, dW, ws, ws.numel(), M, N, K, ld or N, ld or N, 0)
    return dW


# ---------------------------------------------------------------------------------------------
# rotated bank + lifting convolution
# ---------------------------------------------------------------------------------------------
def rotate_bank(weight: torch.Tensor, R: int) -> torch.Tensor:
    """GroupConv.trans_filter (src/models.py:174-197): (C,Cin,1,k,k) -> bank [C*R][Cin*k*k]."""
    C, Cin, D, k, _ = weight.shape
    if D != 1:
        raise NotImplementedError('input_rot_dim != 1 is never used by the reference (models.py:290,346)')
    idx, w = tap_tables(k, R, weight.device)
    bank = torch.empty(C * R, Cin * k * k, dtype=torch.float32, device=weight.device)
    call('tvae_rotate_bank_fwd', weight.contiguous(), idx, w, bank, C, Cin, k, R)
    return bank


def rotate_bank_bwd(dbank: torch.Tensor, C: int, Cin: int, k: int, R: int) -> torch.Tensor:
    ptr, er, ed, ew = tap_tables_csr(k, R, dbank.device)
    dW = torch.empty(C, Cin, 1, k, k, dtype=torch.float32, device=dbank.device)
    call('tvae_rotate_bank_bwd', dbank, ptr, er, ed, ew, dW, C, Cin, k, R, 0)
    return dW


def _use_x6(Cin, n, k, pad) -> bool:
    """Lifting convolution on the bf16 matrix pipe (exact 3 x bf16 operand split, six products; fp32-equivalent)."""
    return split_pipe() and bool(query('tvae_conv1_x6_supported', Cin, n, k, pad))


def _scratch(device, key, floats: int) -> torch.Tensor:
    """Named persistent scratch buffers (split operands of the x6 convolution)."""
    k_ = (key, device.type, device.index)
    t = _WS.get(k_)
    if t is None or t.numel() < floats:
        t = torch.empty(int(floats), dtype=torch.float32, device=device)
        if POISON_WS:
            t.fill_(float('nan'))
        _replace_ws(k_, t)
    return t


# Routing switches.  Two can be set from the environment (each is kept alive by a test): TVAE_CONV_DFT=0 -- the direct split-pipe
# convolution instead of the frequency-domain one (bench.py's conv_direct_form line; tests/test_hip_modules.py runs the direct
# kernels through geometries the DFT plan rejects); TVAE_DEC_PAD=0 -- no padding of 28 x 28 / 50 x 50 pixel ranges to the GEMM
# tile (test_decoder_pads_pixel_ranges_to_the_gemm_tile flips it).  The FUSE_* names below were environment switches while
# their fusions were being A/B'd (rounds 2-5; profiles/README.md has every measurement); all of them won and are plain
# constants now -- the unfused branches they guard remain as the paths of shapes the fused kernels do not cover, and
# FUSE_ENC_TAIL is still flipped by tests/test_hip_modules.py::test_encoder_tail_paths_agree.
CONV_DFT = os.environ.get('TVAE_CONV_DFT', '1') != '0'
FUSE_COLDOT = FUSE_IN_TAIL = FUSE_VIRT_GRAD = FUSE_VIRT_ACT = FUSE_SIGN_BITS = FUSE_ENC_TAIL = True
FUSE_ROW_SUMS = FUSE_ENC_WGRAD = FUSE_NO_H = True
H3_DEEP = True           # round 6: measured bounds -> h3 for every hidden decoder layer (test_deep_decoder_runs_h3_on_measured_bounds)


def _use_dft(B, Cin, n, k, pad, C, R) -> bool:
    """Frequency-domain lifting convolution (DFT along x + batched split-pipe GEMM): 13x fewer matrix FLOPs than the direct
    form; same arithmetic mode as 'x6' (TVAE_CONV_DFT=0 keeps the direct x6 kernels)."""
    return CONV_DFT and split_pipe() and bool(query('tvae_conv1_dft_supported', B, Cin, n, k, pad, C, R))


def conv1_forward(y, weight, bias, C, R, k, pad, act, keep=None, bank=None):
    """Rotated bank + lifting convolution.  `keep` (a dict) receives what the weight gradient can reuse.  `bank`: a ready
    [C*R][Cin*k*k] bank (the plain convolution of --groupconv 0 is the R = 1 case with the weight itself as the bank)."""
    B, Cin, n, _ = y.shape
    Ho = n + 2 * pad - k + 1
    if bank is None:
        bank = rotate_bank(weight, R)
    out = torch.empty(C, B * R * Ho * Ho, dtype=torch.float32, device=y.device)
    if _use_dft(B, Cin, n, k, pad, C, R):
        # (the columns that pad (image, row) to a multiple of 128 are zeroed by the entry point when there are any)
        at = torch.empty(query('tvae_conv1_dft_at_floats', B, Cin, n, k, pad, C, R), dtype=torch.float32,
                         device=y.device)
        ws = _scratch(y.device, 'dft_ws', query('tvae_conv1_dft_ws_floats', B, Cin, n, k, pad, C, R))
        _note('conv1.dft')
        if query('tvae_conv1_dft_ring', B, Cin, n, k, pad, C, R):
            _note('conv1.dft_ring')      # ring (LDS-DMA) transforms along w: abi_conv_dft.hip dft_plan
        with _timed('tvae_conv1_fwd', parts()):
            call('tvae_conv1_fwd_dft', y, bank, bias, out, at, ws, ws.numel(), B, Cin, n, k, pad, C, R, act,
                 LRELU_SLOPE, parts())
        if keep is not None:
            keep['at'] = at
            keep['out_max'] = at[-C:]      # max |out| per channel, left by the output transform (h3 scales of the encoder tail)
        return out
    if _use_x6(Cin, n, k, pad):
        a3 = _scratch(y.device, 'x6_bank', query('tvae_conv1_x6_bank_bytes', C, R, Cin, k) // 4)
        call('tvae_bank_split3', bank, a3, a3.numel() * 4, C, R, Cin, k)
        _note('conv1.x6')
        with _timed('tvae_conv1_fwd', 3):
            call('tvae_conv1_fwd_x6', y, a3, bias, out, B, Cin, n, k, pad, C, R, act, LRELU_SLOPE)
        return out
    _note('conv1.f32')
    with _timed('tvae_conv1_fwd'):
        call('tvae_conv1_fwd', y, bank, bias, out, B, Cin, n, k, pad, C, R, act, LRELU_SLOPE)
    return out


def conv1_wgrad(y, dpre, C, R, k, pad, at=None, dbias=None):
    """Weight gradient of the lifting convolution.  With the frequency-domain path (`at` from the forward call) the bias
    gradient is a by-product: pass `dbias` (C floats) to receive it; returns (dbank, bias_done)."""
    B, Cin, n, _ = y.shape
    dbank = torch.empty(C * R, Cin * k * k, dtype=torch.float32, device=y.device)
    if at is not None:
        wsd = _scratch(y.device, 'dft_ws', query('tvae_conv1_dft_ws_floats', B, Cin, n, k, pad, C, R))
        with _timed('tvae_conv1_wgrad', parts()):
            call('tvae_conv1_wgrad_dft', dpre, at, dbank, dbias, wsd, wsd.numel(), B, Cin, n, k, pad, C, R, parts())
        return dbank
    ws = workspace(y.device, max(1 << 24, 16 * dbank.numel()))
    if _use_x6(Cin, n, k, pad):
        d3 = _scratch(y.device, 'x6_dy', query('tvae_conv1_x6_dy_bytes', B, C, R, n, k, pad) // 4)
        call('tvae_dy_split3', dpre, d3, d3.numel() * 4, B, Cin, n, k, pad, C, R)
        with _timed('tvae_conv1_wgrad', 3):
            call('tvae_conv1_wgrad_x6', y, d3, dbank, ws, ws.numel(), B, Cin, n, k, pad, C, R)
        return dbank
    with _timed('tvae_conv1_wgrad'):
        call('tvae_conv1_wgrad', y, dpre, dbank, ws, ws.numel(), B, Cin, n, k, pad, C, R)
    return dbank


class BankFn(torch.autograd.Function):
    """GroupConv.trans_filter as a differentiable op: (C,Cin,1,k,k) -> [C*R][Cin*k*k]."""

    @staticmethod
    def forward(ctx, weight, R):
        ctx.cfg = (weight.shape[0], weight.shape[1], weight.shape[3], R)
        return rotate_bank(weight, R)

    @staticmethod
    def backward(ctx, g):
        C, Cin, k, R = ctx.cfg
        return rotate_bank_bwd(g.contiguous().view(C * R, Cin * k * k), C, Cin, k, R), None


class GroupConvFn(torch.autograd.Function):
    """GroupConv.forward (src/models.py:202-225) as one op: rotated bank + implicit-GEMM conv + bias.
    Returns the reference layout (B, C, R, Ho, Ho).  No input gradient (the input is data)."""

    @staticmethod
    def forward(ctx, y, weight, bias, R, pad):
        C, Cin, _, k, _ = weight.shape
        y = y.contiguous().view(y.shape[0], Cin, y.shape[-2], y.shape[-1])
        B, n = y.shape[0], y.shape[-1]
        Ho = n + 2 * pad - k + 1
        keep = {}
        out = conv1_forward(y, weight, bias, C, R, k, pad, ACT_NONE, keep)
        ctx.arith = get_gemm_mode()
        ctx.save_for_backward(y)
        ctx.at = keep.get('at')
        ctx.cfg = (C, Cin, k, R, pad, B, Ho, bias is not None)
        return out.view(C, B, R, Ho, Ho).permute(1, 0, 2, 3, 4)

    @staticmethod
    @_in_forward_arithmetic
    def backward(ctx, g):
        (y,) = ctx.saved_tensors
        C, Cin, k, R, pad, B, Ho, has_bias = ctx.cfg
        if ctx.needs_input_grad[0]:
            raise NotImplementedError('GroupConv input gradient is not part of the hot path (input is data)')
        dpre = g.permute(1, 0, 2, 3, 4).contiguous().view(C, B * R * Ho * Ho)
        db = None
        if has_bias and ctx.at is not None:
            db = torch.empty(C, dtype=torch.float32, device=y.device)
        dbank = conv1_wgrad(y, dpre, C, R, k, pad, ctx.at, db)
        dW = rotate_bank_bwd(dbank, C, Cin, k, R)
        if has_bias and db is None:
            db = _rowsum(dpre, C, dpre.shape[1])
        return None, dW, db, None, None


def _enc_tail_perm(device) -> torch.Tensor:
    """Column order of W2^T for tvae_enc_tail_dgrad_x6 (include/tvae_hip.h): slot 16 u + 8 h + j holds row
    16 u + 8 (j >> 2) + 4 h + (j & 3) -- the order in which the first GEMM's accumulator registers feed the second."""
    def build():
        s_ = np.arange(128)
        u, h, j = s_ // 16, (s_ % 16) // 8, s_ % 8
        return (16 * u + 8 * (j // 4) + 4 * h + (j % 4),)
    return _dev_table(('enc_tail_perm',), device, build)[0]


ENC_TAIL_MAX_COLS = 1 << 25      # the fused kernels index 32 rows with 32-bit byte offsets (abi_enc_tail_x6.hip: ET_MAX_LD)


def _enc_tail_fused(C: int, C2: int, nh: int, N: int = 0) -> bool:
    """Fused encoder tail (enc_tail_x6_kernels.hpp): the reference's default 128 channels, <= 7 head rows, split pipe,
    fewer than 2^25 columns (beyond that the unfused path with 64-bit indexing takes over)."""
    return FUSE_ENC_TAIL and split_pipe() and C == 128 and C2 == 128 and nh <= 7 and N < ENC_TAIL_MAX_COLS


def _enc_tail_wide(C: int, C2: int, nh: int, N: int, have_a1max: bool) -> bool:
    """Round 6: the encoder tail with 8 .. 128 head rows (z_dim > 2: the galaxy configuration's z_dim = 50 gives 103) as two
    chained split-pipe GEMMs per direction (enc_tail_wide_kernels.hpp) instead of separate fp32-MFMA GEMMs.  h3 or bf16
    arithmetic (both weights stay in LDS; the exact three-part split does not fit), the reference's 128 channels."""
    return (FUSE_ENC_TAIL and split_pipe() and parts() in (1, 2) and C == 128 and C2 == 128 and SKINNY_MAX <= nh <= 128 and
            N < ENC_TAIL_MAX_COLS and (parts() == 1 or have_a1max))


class EncoderFn(torch.autograd.Function):
    """conv1 -> act -> conv2 (1x1x1) -> act -> {conv_a, conv_r, conv_z} (src/models.py:354-358,390-392).

    Output: heads [3+2*zd][B*R*Ho*Ho] feature-major, rows = (logit, theta_mu, theta_logstd, z_mu.., z_logstd..).
    Backward fuses every activation derivative into the producing dgrad epilogue."""

    @staticmethod
    def forward(ctx, y, w1, b1, W2, b2, Wh, bh, R, pad, act):
        C, Cin, _, k, _ = w1.shape
        _expect(y.numel() == y.shape[0] * Cin * y.shape[-1] * y.shape[-2] and y.shape[-1] == y.shape[-2],
                f'encoder input {tuple(y.shape)} is not (B, {Cin}, n, n)')
        y = y.contiguous().view(y.shape[0], Cin, y.shape[-2], y.shape[-1])
        B, n = y.shape[0], y.shape[-1]
        Ho = n + 2 * pad - k + 1
        _expect(Ho >= 1, f'kernel {k} with padding {pad} does not fit a {n}x{n} image')
        _expect(R in (4, 8, 16), f'groupconv must be 4, 8 or 16 (got {R})')
        N = B * R * Ho * Ho
        C2, nh = W2.shape[0], Wh.shape[0]
        _expect(tuple(W2.shape) == (C2, C) and tuple(Wh.shape) == (nh, C2) and b1.numel() == C and
                b2.numel() == C2 and bh.numel() == nh, 'encoder parameter shapes are inconsistent')
        keep = {}
        infer = _INFER
        A1 = conv1_forward(y, w1, b1, C, R, k, pad, act, keep)
        ctx.at = None if infer else keep.get('at')
        fused = _enc_tail_fused(C, C2, nh, N)
        wide = not fused and _enc_tail_wide(C, C2, nh, N, keep.get('out_max') is not None)
        # inference: the fused tail writes the head rows only -- H is neither allocated nor stored (include/tvae_hip.h, ABI 6)
        H = None if (infer and (fused or wide)) else torch.empty(C2, N, dtype=torch.float32, device=y.device)
        heads = torch.empty(nh, N, dtype=torch.float32, device=y.device)
        bits = None
        a1max = None
        if infer:
            _note('enc.inference')
        if wide:
            # conv2 + the stacked head projection as two chained GEMMs per 32-column chunk (H handed over in registers)
            p_f = parts()
            a1max = keep.get('out_max') if p_f == 2 else None
            w3 = _split_weight(W2, C2, C, False, 'enc_w2', nparts=p_f)
            whp = _split_weight(Wh[:, _enc_tail_perm(y.device)], nh, C2, False, 'enc_whp', nparts=p_f)
            _note('enc.tail_fwd_wide')
            if act == ACT_LRELU and not infer:
                bits = torch.empty(2, N, 4, dtype=torch.int32, device=y.device)
            with _timed('tvae_enc_tail_fwd_wide', p_f):
                call('tvae_enc_tail_fwd_wide', w3, whp, A1, N, b2, bh.contiguous(), nh, H, N, heads, N,
                     bits[0] if bits is not None else None, bits[1] if bits is not None else None, C, N, act, LRELU_SLOPE,
                     p_f, a1max)
        elif fused:
            # conv2 + the stacked head projection in one pass over A1 and one over H, on the split pipe
            # h3 instance: needs max |A1| from A1's producer -- the output transform of the frequency-domain convolution leaves
            # it in the last word behind A^T (keep['at'])
            a1max = keep.get('out_max') if parts() == 2 else None
            p_f = 2 if a1max is not None else _p3()
            w3 = _split_weight(W2, C2, C, False, 'enc_w2', nparts=p_f)
            _note('enc.tail_fwd_x6')
            if act == ACT_LRELU and not infer:           # sign words of H and A1: all the fused data gradient reads of them
                bits = torch.empty(2, N, 4, dtype=torch.int32, device=y.device)
            with _timed('tvae_enc_tail_fwd_x6', p_f):
                call('tvae_enc_tail_fwd_x6', w3, A1, N, b2, Wh.contiguous(), bh.contiguous(), nh, H, N, heads, N,
                     bits[0] if bits is not None else None, bits[1] if bits is not None else None, C, N, act,
                     LRELU_SLOPE, p_f, a1max)
        else:
            call('tvae_linear_fwd', W2.contiguous(), A1, b2, None, 1, None, H, C2, N, C, N, N, act, LRELU_SLOPE)
            if nh <= SKINNY_MAX:
                call('tvae_heads_fwd', Wh.contiguous(), H, N, bh, heads, N, nh, C2, N)
            else:
                call('tvae_linear_fwd', Wh.contiguous(), H, bh, None, 1, None, heads, nh, N, C2, N, N, ACT_NONE,
                     LRELU_SLOPE)
        if infer:
            return heads
        ctx.save_for_backward(y, W2, Wh, A1, H)
        ctx.bits = bits
        ctx.wide = wide
        ctx.a1max = a1max if (fused or wide) else None      # (a view of keep['at']: max |A1| per channel for the h3 weight gradients)
        # wide tail, h3: bound per row of H = act(W2 A1 + b2) for the weight gradient that streams H (|act(x)| <= |x|)
        ctx.hrows = torch.addmv(b2.detach().abs(), W2.detach().abs(), a1max) if (wide and a1max is not None) else None
        ctx.arith = get_gemm_mode()
        ctx.cfg = (C, Cin, k, R, pad, B, Ho, act)
        return heads

    @staticmethod
    @_in_forward_arithmetic
    def backward(ctx, dheads):
        y, W2, Wh, A1, H = ctx.saved_tensors
        C, Cin, k, R, pad, B, Ho, act = ctx.cfg
        N = B * R * Ho * Ho
        C2, nh = W2.shape[0], Wh.shape[0]
        dheads = dheads.contiguous()
        wide_bw = ctx.wide and ctx.bits is not None
        words = torch.zeros(2, dtype=torch.float32, device=y.device) if (wide_bw and parts() == 2) else None
        dmax, dhmax = (words[0:1], words[1:2]) if words is not None else (None, None)
        dbh = _rowsum(dheads, nh, N, amax=dmax)          # (+ max |dheads| by the way: the h3 bound of the wide data gradient)
        dA1 = None
        dH = None
        if wide_bw:
            # dH = act'(H) . Wh^T dheads and dA1 = act'(A1) . W2^T dH in one launch (dH handed over in registers and stored
            # once for the two weight gradients below)
            p_e = parts()
            wht = _split_weight(Wh, C2, nh, True, 'enc_wht', nparts=p_e)
            w3p = _split_weight(W2.t()[:, _enc_tail_perm(y.device)], C, C2, False, 'enc_w2p', nparts=p_e)
            dH = torch.empty(C2, N, dtype=torch.float32, device=y.device)
            dA1 = torch.empty(C, N, dtype=torch.float32, device=y.device)
            _note('enc.tail_dgrad_wide')
            with _timed('tvae_enc_tail_dgrad_wide', p_e):
                call('tvae_enc_tail_dgrad_wide', wht, w3p, dheads, N, nh, ctx.bits[0], ctx.bits[1], dH, N, dA1, N, C, N,
                     LRELU_SLOPE, p_e, dmax)
        elif ctx.bits is not None:
            # dA1 straight from the head gradients and the sign words (dH is formed in registers, never stored)
            p_e = 2 if parts() == 2 else _p3()      # h3 instance: 128 x 128 GEMM on two fp16 parts, scale per 32-column chunk
            w3p = _split_weight(W2.t()[:, _enc_tail_perm(y.device)], C, C2, False, 'enc_w2p', nparts=p_e)
            wh3 = _scratch(y.device, 'enc_wh3', query('tvae_dense_x6_bytes', C2, nh) // 4)
            call('tvae_dense_split3', Wh.contiguous(), C2, wh3, wh3.numel() * 4, C2, nh, 1, None, None)
            dA1 = torch.empty(C, N, dtype=torch.float32, device=y.device)
            _note('enc.tail_dgrad_x6')
            with _timed('tvae_enc_tail_dgrad_x6', p_e):
                call('tvae_enc_tail_dgrad_x6', w3p, wh3, dheads, N, nh, ctx.bits[0], ctx.bits[1], dA1, N, C, N,
                     LRELU_SLOPE, p_e)
        # conv2's weight gradient in one pass from A1, the head gradients and the sign words of H (dH is formed inside the
        # GEMM's operand build and never written); dWh / db2 from a sums-only pass over H
        fuse_w = ctx.bits is not None and FUSE_ENC_WGRAD and nh <= SKINNY_MAX and N % 32 == 0 and not wide_bw
        if not fuse_w and dH is None:
            dH = torch.empty(C2, N, dtype=torch.float32, device=y.device)
        # wide tail in h3: both weight gradients as cooperative reductions over two stored operands (enc_tail_wgrad_plain_kernel)
        wide_w = wide_bw and dmax is not None and ctx.hrows is not None and ctx.a1max is not None and N % 32 == 0
        if wide_w:
            _note('enc.tail_wgrad_wide')
            wsw = _scratch(y.device, 'enc_wgrad_slabs', query('tvae_enc_tail_wgrad_x6_ws_floats', N))
            dWf = torch.empty(2, C2, C, dtype=torch.float32, device=y.device)
            with _timed('tvae_enc_tail_wgrad_wide', 2):
                call('tvae_enc_tail_wgrad_wide', dheads, N, nh, H, N, dWf[0], wsw, wsw.numel(), C2, N, dmax, ctx.hrows)
            dWh = dWf[0, :nh]
            db2 = _rowsum(dH, C2, N, amax=dhmax)         # (+ max |dH|: the bound of the launch below)
            with _timed('tvae_enc_tail_wgrad_wide', 2):
                call('tvae_enc_tail_wgrad_wide', dH, N, C2, A1, N, dWf[1], wsw, wsw.numel(), C, N, dhmax, ctx.a1max)
            dW2 = dWf[1]
        elif wide_bw:
            dWh = _wgrad(dheads, H, nh, N, C2)
            db2 = _rowsum(dH, C2, N)
        elif nh <= SKINNY_MAX:
            # one pass over H: masked dgrad + dWh + the row sums of dH (= db2)
            npan = (N + 511) // 512
            part = workspace(y.device, npan * C2 * (nh + 1) + 4 + 64 * C2 * (nh + 1))      # + room for the two-stage total
            tot = torch.empty(nh + 1, C2, dtype=torch.float32, device=y.device)
            call('tvae_heads_bwd', Wh.contiguous(), dheads, N, H, N, dH, N, nh, C2, N, act, LRELU_SLOPE, part,
                 part.numel(), tot)
            dWh, db2 = tot[:nh], tot[nh]
        else:
            dWh = _wgrad(dheads, H, nh, N, C2)
            call('tvae_linear_dgrad', Wh.contiguous(), dheads, None, H, dH, nh, N, C2, N, N, act, LRELU_SLOPE)
            db2 = _rowsum(dH, C2, N)
        if fuse_w:
            _note('enc.tail_wgrad_x6')
            dW2 = torch.empty(C2, C, dtype=torch.float32, device=y.device)
            wsw = _scratch(y.device, 'enc_wgrad_slabs', query('tvae_enc_tail_wgrad_x6_ws_floats', N))
            p_w = 2 if (parts() == 2 and ctx.a1max is not None) else _p3()
            with _timed('tvae_enc_tail_wgrad_x6', p_w):
                call('tvae_enc_tail_wgrad_x6', A1, N, dheads, N, nh, ctx.bits[0], Wh.contiguous(), dW2, wsw, wsw.numel(), C,
                     N, LRELU_SLOPE, p_w, ctx.a1max if p_w == 2 else None)
        elif not wide_w:
            dW2 = _wgrad(dH, A1, C2, N, C)
        if dA1 is None:
            dA1 = torch.empty(C, N, dtype=torch.float32, device=y.device)
            call('tvae_linear_dgrad', W2.contiguous(), dH, None, A1, dA1, C2, N, C, N, N, act, LRELU_SLOPE)
        del dH
        db1 = torch.empty(C, dtype=torch.float32, device=y.device) if ctx.at is not None else _rowsum(dA1, C, N)
        dbank = conv1_wgrad(y, dA1, C, R, k, pad, ctx.at, db1 if ctx.at is not None else None)
        dw1 = rotate_bank_bwd(dbank, C, Cin, k, R)
        return None, dw1, db1, dW2, db2, dWh, dbh, None, None, None


class TransAttnEncoderFn(torch.autograd.Function):
    """Translation-attention encoder with pooled rotation (reference src/models.py:268-319) on the
    same kernels as the main encoder: conv1 (lifting convolution, or the plain R = 1 convolution of --groupconv 0) ->
    act -> fc_r rotation pooling (`tvae_rot_pool_*`) -> conv2 (1x1) -> act -> {conv_a, conv_r, conv_z} stacked.
    Output: heads [1 + 2 + 2*zd][B*Ho*Ho] feature-major."""

    @staticmethod
    def forward(ctx, y, w1, b1, fw, fb, W2, b2, Wh, bh, R, pad, act):
        C = w1.shape[0]
        plain = w1.dim() == 4                            # nn.Conv2d weight (C, Cin, k, k): groupconv = 0
        Cin, k = w1.shape[1], w1.shape[-1]
        y = y.contiguous().view(y.shape[0], Cin, y.shape[-2], y.shape[-1])
        B, n = y.shape[0], y.shape[-1]
        Ho = n + 2 * pad - k + 1
        _expect(Ho >= 1 and (plain or R in (4, 8, 16)), 'translation-attention encoder geometry')
        Re = 1 if plain else R
        P = Ho * Ho
        keep = {}
        A1 = conv1_forward(y, None if plain else w1, b1, C, Re, k, pad, act, keep,
                           bank=w1.contiguous().view(C, Cin * k * k) if plain else None)
        _note('trans_attn.plain' if plain else 'trans_attn.rot_pool')
        if plain:
            X = A1                                       # [C][B*P]
        else:
            X = torch.empty(C, B * P, dtype=torch.float32, device=y.device)
            call('tvae_rot_pool_fwd', A1, fw.contiguous().view(-1), fb.contiguous(), X, C, B, R, P)
        N = B * P
        C2, nh = W2.shape[0], Wh.shape[0]
        H = torch.empty(C2, N, dtype=torch.float32, device=y.device)
        call('tvae_linear_fwd', W2.contiguous(), X, b2, None, 1, None, H, C2, N, C, N, N, act, LRELU_SLOPE)
        heads = torch.empty(nh, N, dtype=torch.float32, device=y.device)
        if nh <= SKINNY_MAX:
            call('tvae_heads_fwd', Wh.contiguous(), H, N, bh, heads, N, nh, C2, N)
        else:
            call('tvae_linear_fwd', Wh.contiguous(), H, bh, None, 1, None, heads, nh, N, C2, N, N, ACT_NONE, LRELU_SLOPE)
        ctx.save_for_backward(y, w1, fw, W2, Wh, A1, X, H)
        ctx.at = keep.get('at')
        ctx.cfg = (C, Cin, k, R, pad, B, Ho, act, plain)
        ctx.arith = get_gemm_mode()
        return heads

    @staticmethod
    @_in_forward_arithmetic
    def backward(ctx, dheads):
        y, w1, fw, W2, Wh, A1, X, H = ctx.saved_tensors
        C, Cin, k, R, pad, B, Ho, act, plain = ctx.cfg
        P = Ho * Ho
        N = B * P
        C2, nh = W2.shape[0], Wh.shape[0]
        dev = y.device
        dheads = dheads.contiguous()
        dbh = _rowsum(dheads, nh, N)
        dH = torch.empty(C2, N, dtype=torch.float32, device=dev)
        if nh <= SKINNY_MAX:
            npan = (N + 511) // 512
            part = workspace(dev, npan * C2 * (nh + 1) + 4 + 64 * C2 * (nh + 1))      # + room for the two-stage total
            tot = torch.empty(nh + 1, C2, dtype=torch.float32, device=dev)
            call('tvae_heads_bwd', Wh.contiguous(), dheads, N, H, N, dH, N, nh, C2, N, act, LRELU_SLOPE, part,
                 part.numel(), tot)
            dWh, db2 = tot[:nh], tot[nh]
        else:
            dWh = _wgrad(dheads, H, nh, N, C2)
            call('tvae_linear_dgrad', Wh.contiguous(), dheads, None, H, dH, nh, N, C2, N, N, act, LRELU_SLOPE)
            db2 = _rowsum(dH, C2, N)
        dW2 = _wgrad(dH, X, C2, N, C)
        dX = torch.empty(C, N, dtype=torch.float32, device=dev)
        # the pooled tensor is not an activation output: plain data gradient (groupconv 0: X = act(conv1), masked)
        call('tvae_linear_dgrad', W2.contiguous(), dH, None, A1 if plain else None, dX, C2, N, C, N, N,
             act if plain else ACT_NONE, LRELU_SLOPE)
        dfw = dfb = None
        Re = 1 if plain else R
        if plain:
            dA1 = dX
        else:
            dA1 = torch.empty_like(A1)
            nb = min(1024, (C * N + 255) // 256)
            partp = workspace(dev, (R + 1) * nb)
            dtot = torch.empty(R + 1, dtype=torch.float32, device=dev)
            call('tvae_rot_pool_bwd', A1, dX, fw.contiguous().view(-1), dA1, partp, partp.numel(), dtot, C, B, R, P, act,
                 LRELU_SLOPE)
            dfw, dfb = dtot[:R].view(1, R), dtot[R:]
        db1 = torch.empty(C, dtype=torch.float32, device=dev) if ctx.at is not None else _rowsum(dA1, C, dA1.shape[1])
        dbank = conv1_wgrad(y, dA1, C, Re, k, pad, ctx.at, db1 if ctx.at is not None else None)
        dw1 = dbank.view(C, Cin, k, k) if plain else rotate_bank_bwd(dbank, C, Cin, k, R)
        return None, dw1, db1, dfw, dfb, dW2, db2, dWh, dbh, None, None, None


class MlpFn(torch.autograd.Function):
    """Linear / ResidLinear stack of the MLP encoder (reference src/models.py:229-260) on the GEMM kernels, feature-major:
    h0 = act(W0 x + b0); hi = act(Wi h + bi [+ h]); out = Wl h + bl.  x is data (no input gradient)."""

    @staticmethod
    def forward(ctx, x, act, resid, *params):
        nl = len(params) // 2
        _expect(nl >= 2 and len(resid) == nl and x.dim() == 2 and params[0].shape[1] == x.shape[1], 'MLP encoder shapes')
        B = x.shape[0]
        xt = x.t().contiguous()                          # [n][B]
        hs = [xt]
        for i in range(nl):
            W, b = params[2 * i], params[2 * i + 1]
            M, K = W.shape
            _expect(K == hs[-1].shape[0] and (not resid[i] or M == K), 'MLP encoder layer widths')
            h = torch.empty(M, B, dtype=torch.float32, device=x.device)
            call('tvae_linear_fwd', W.contiguous(), hs[-1], b, None, 1, hs[-1] if resid[i] else None, h, M, B, K, B, B,
                 act if i + 1 < nl else ACT_NONE, LRELU_SLOPE)
            hs.append(h)
        _note('mlp_encoder.kernels')
        ctx.save_for_backward(*hs[:-1], *params[0::2])
        ctx.cfg = (act, tuple(resid), nl, B)
        ctx.arith = get_gemm_mode()
        return hs[-1].t()

    @staticmethod
    @_in_forward_arithmetic
    def backward(ctx, dout):
        act, resid, nl, B = ctx.cfg
        hs, Ws = ctx.saved_tensors[:nl], ctx.saved_tensors[nl:]
        d = dout.t().contiguous()                        # gradient of layer i's pre-activation, [M][B]
        grads = [None] * (2 * nl)
        for i in range(nl - 1, -1, -1):
            M, K = Ws[i].shape
            grads[2 * i] = _wgrad(d, hs[i], M, B, K)
            grads[2 * i + 1] = _rowsum(d, M, B)
            if i > 0:
                dprev = torch.empty(K, B, dtype=torch.float32, device=d.device)
                call('tvae_linear_dgrad', Ws[i].contiguous(), d, d if resid[i] else None, hs[i], dprev, M, B, K, B, B, act,
                     LRELU_SLOPE)
                d = dprev
        return (None, None, None, *grads)


# ---------------------------------------------------------------------------------------------
# attention head
# ---------------------------------------------------------------------------------------------
class HeadTables:
    """Device-resident constants of the attention head for one (R, Ho, spacing, prior) configuration."""

    def __init__(self, R, Ho, spacing, rot_refinement, theta_prior, normal_prior_over_r, device):
        self.R, self.Ho, self.P = R, Ho, Ho * Ho
        p_r = tables.rotation_log_prior(R, rot_refinement, theta_prior, normal_prior_over_r)
        off = tables.rotation_offsets(R, rot_refinement)
        grid = tables.translation_grid(Ho, spacing)
        p_tr = tables.joint_log_prior(Ho, spacing, p_r)
        self.p_r = torch.from_numpy(p_r).to(device)
        self.off = torch.from_numpy(off).to(device)
        self.grid = torch.from_numpy(grid.astype(np.float32)).contiguous().to(device)
        self.p_tr = torch.from_numpy(p_tr).to(device)
        self.sigma_p = math.pi / R                       # train_mnist.py:269-272
        self.theta_off_scale = 1.0 if rot_refinement else 0.0


class HeadFn(torch.autograd.Function):
    """Prior add + log-softmax + Gumbel-softmax (src/models.py:382-388) fused with expected-value pooling,
    reparameterised sampling, translation expectation and the KL block (train_mnist.py:192-231,242-282).

    Returns (attn, q_t_r, a_sampled) [B][R*P], z [B][zd], theta [B], dx [B][2], kl [B]."""

    @staticmethod
    def forward(ctx, heads, E, eps_z, eps_t, tb: HeadTables, B, zd):
        RP = tb.R * tb.P
        dev = heads.device
        _expect(tuple(heads.shape) == (3 + 2 * zd, B * RP), f'heads {tuple(heads.shape)} != ({3 + 2 * zd}, {B * RP})')
        _expect(E.numel() == B * RP and eps_z.numel() == B * zd and eps_t.numel() == B, 'noise shapes do not match')
        attn = torch.empty(B, RP, dtype=torch.float32, device=dev)
        q = torch.empty_like(attn)
        a = torch.empty_like(attn)
        z = torch.empty(B, zd, dtype=torch.float32, device=dev)
        th = torch.empty(B, dtype=torch.float32, device=dev)
        dx = torch.empty(B, 2, dtype=torch.float32, device=dev)
        kl = torch.empty(B, dtype=torch.float32, device=dev)
        heads = heads.contiguous()
        E, eps_z, eps_t = E.contiguous(), eps_z.contiguous(), eps_t.contiguous()
        part = _scratch(dev, 'head_part', 1024 * (7 + 3 * (zd + 1)))     # chunk partials (few images, many positions)
        call('tvae_attn_head_fwd', heads, heads.shape[1], E, eps_z, eps_t, tb.p_r, tb.off, tb.p_tr, tb.grid, B, tb.R,
             tb.P, zd, tb.sigma_p, tb.theta_off_scale, attn, q, a, z, th, dx, kl, part, part.numel())
        ctx.save_for_backward(heads, q, a, eps_z, eps_t)
        ctx.tb, ctx.B, ctx.zd = tb, B, zd
        # outputs nothing differentiates (attn, q_t_r, a_sampled in the training step) reach backward as None, not as three
        # zero-filled [B][R P] tensors
        ctx.set_materialize_grads(False)
        return attn, q, a, z, th, dx, kl

    @staticmethod
    def backward(ctx, g_attn, g_q, g_a, g_z, g_th, g_dx, g_kl):
        heads, q, a, eps_z, eps_t = ctx.saved_tensors
        tb, B, zd = ctx.tb, ctx.B, ctx.zd
        dev = heads.device

        def dense(g, shape):
            return torch.zeros(shape, dtype=torch.float32, device=dev) if g is None else g.contiguous()

        g_z, g_th, g_dx, g_kl = dense(g_z, (B, zd)), dense(g_th, (B,)), dense(g_dx, (B, 2)), dense(g_kl, (B,))
        opt = [None if g is None else g.contiguous() for g in (g_attn, g_q, g_a)]
        dheads = torch.empty_like(heads)
        part = _scratch(dev, 'head_part', 1024 * (7 + 3 * (zd + 1)))
        call('tvae_attn_head_bwd', heads, heads.shape[1], q, a, eps_z, eps_t, tb.p_r, tb.off, tb.p_tr, tb.grid, B,
             tb.R, tb.P, zd, tb.sigma_p, tb.theta_off_scale, g_z, g_th, g_dx, g_kl, opt[0], opt[1], opt[2], dheads,
             part, part.numel())
        return dheads, None, None, None, None, None, None


# ---------------------------------------------------------------------------------------------
# coordinate transform, decoder, likelihood
# ---------------------------------------------------------------------------------------------
class CoordFn(torch.autograd.Function):
    """x' = (x - dx) R(theta) (train_mnist.py:222,234-239).  xc [Np][2] -> [B][Np][2]."""

    @staticmethod
    def forward(ctx, xc, dx, theta):
        B, Np = theta.shape[0], xc.shape[0]
        _expect(tuple(xc.shape) == (Np, 2) and tuple(dx.shape) == (B, 2) and theta.dim() == 1, 'coordinate shapes')
        xc, dx, theta = xc.contiguous(), dx.contiguous(), theta.contiguous()
        xr = torch.empty(B, Np, 2, dtype=torch.float32, device=xc.device)
        call('tvae_coord_fwd', xc, dx, theta, xr, B, Np)
        ctx.save_for_backward(xc, dx, theta)
        return xr

    @staticmethod
    def backward(ctx, g):
        xc, dx, theta = ctx.saved_tensors
        B, Np = theta.shape[0], xc.shape[0]
        gdx = torch.empty(B, 2, dtype=torch.float32, device=xc.device)
        gth = torch.empty(B, dtype=torch.float32, device=xc.device)
        call('tvae_coord_bwd', xc, dx, theta, g.contiguous(), gdx, gth, B, Np)
        return None, gdx, gth


DEC_PAD = os.environ.get('TVAE_DEC_PAD', '1') != '0'


def decoder_padded_pixels(Np: int, B: int, F_: int, n_hidden: int, n_out: int, resid: bool, fourier: bool, act: int) -> int:
    """Round 5: images whose pixel count is not a multiple of the 128-column GEMM tile (28 x 28 = 784, 50 x 50 = 2 500) cannot
    take the decoder's fast path -- first layer recomputed inside its consumers, lean epilogues, fused first-layer backward --
    because its tiles must lie inside one image.  Padding every image's pixel range to the next multiple of 128 (coordinates
    (0, 0), outputs sliced away, zero upstream gradient: every reduction of the backward is unaffected) costs <= 15 % more
    columns and removes the stored first layer, its separate backward pass and the generic epilogues.  Returns the padded
    pixel count, or 0 when the decoder should run as it is (src/models.py: SpatialGenerator.forward pads around DecoderFn, so
    autograd differentiates the padding)."""
    Np_p = (Np + 127) // 128 * 128
    if (not DEC_PAD or Np % 128 == 0 or fourier or resid or n_hidden < 1 or n_out != 1 or act != ACT_LRELU or
            not (256 <= F_ <= 512) or Np_p > 1.15 * Np or not (FUSE_VIRT_ACT and FUSE_IN_TAIL) or not _dense_x6_ok(F_, B * Np_p)):
        return 0
    return Np_p


DEC_LD_PAD = int(os.environ.get('TVAE_DEC_LD_PAD', '64'))


def _dec_ld(Nt: int, F_: int, n_hidden: int, fourier: bool) -> int:
    """Row stride (floats) of the decoder's stored [features][B*Np] tensors.  The decoders that STORE activations and gradients
    (Fourier first layer, two or more hidden layers) stream them again as the panels of their weight gradients; with a
    power-of-two column count (galaxy: 8 x 128^2 = 2^17) the 512 rows of a panel sit 2^19 bytes apart, every one in the same
    sets of the L2, and the four workgroups that share a panel each fetch it again (1.88 GB fetched for 0.54 GB,
    profiles/r06_wgrad_probe.txt).  64 floats of padding per row take the rows off that spacing."""
    if DEC_LD_PAD and (fourier or n_hidden >= 2) and Nt % 16384 == 0 and _dense_x6_ok(F_, Nt):      # (rows >= 64 KB x k apart)
        return Nt + DEC_LD_PAD
    return Nt


class DecoderFn(torch.autograd.Function):
    """SpatialGenerator.forward (src/models.py:95-123): [Fourier] -> coord_linear + latent_linear -> act ->
    (Linear | ResidLinear, act) x (L-1) -> Linear(hid, n_out).  x [B][Np][2], z [B][zd] -> (B, Np, n_out).

    params order: Wc, bc, Wl (or None), [W_i, b_i]*(L-1), Wo, bo, then Fourier buffers Wf, bf (or None)."""

    @staticmethod
    def forward(ctx, xr, z, act, resid, sigma, n_hidden, *params):
        Wc, bc, Wl = params[0], params[1], params[2]
        hidden = [(params[3 + 2 * i], params[4 + 2 * i]) for i in range(n_hidden)]
        Wo, bo = params[3 + 2 * n_hidden], params[4 + 2 * n_hidden]
        Wf, bf = params[5 + 2 * n_hidden], params[6 + 2 * n_hidden]
        xr = xr.contiguous()
        _expect(xr.dim() == 3 and xr.shape[2] == 2, f'decoder coordinates {tuple(xr.shape)} are not (B, N, 2)')
        B, Np = xr.shape[0], xr.shape[1]
        Nt = B * Np
        F_, n_out = Wc.shape[0], Wo.shape[0]
        _expect(1 <= n_out <= 4, f'n_out = {n_out} (the skinny output kernels handle 1..4)')
        _expect(Wc.shape[1] == (Wf.shape[0] if Wf is not None else 2) and Wo.shape[1] == F_ and
                all(tuple(W.shape) == (F_, F_) for W, _ in hidden), 'decoder parameter shapes are inconsistent')
        _expect(Wl is None or (z is not None and tuple(z.shape) == (B, Wl.shape[1])), 'latent z does not match latent_linear')
        dev = xr.device
        infer = _INFER
        LB = None
        ldn = _dec_ld(Nt, F_, n_hidden, Wf is not None)
        if ldn != Nt:
            _note('dec.ld_pad')
        # Fourier-feature first layer on the split pipe: the per-image latent term joins the reduction (rows z[img(n)] under
        # the features, weights [Wc | Wl]) instead of being a per-image bias
        four_x6 = Wf is not None and _dense_x6_ok(F_, Nt)
        if Wl is not None:
            z = z.contiguous()
        if Wl is not None and not four_x6:
            zd = Wl.shape[1]
            LB = torch.empty(B, F_, dtype=torch.float32, device=dev)
            call('tvae_latent_bias', Wl.contiguous(), z, LB, B, F_, zd)
        feat = None
        feat_amax = h_bound = h_rows = None       # h3 bounds of operands that are streamed from memory (include/tvae_hip.h: x_amax)
        # without Fourier features the coordinate layer's output is two FMAs and an activation per element: the layers
        # that consume it (first hidden layer: forward, mask of the data gradient, weight gradient) recompute it and the
        # [hid][B*n^2] tensor is never written or read
        virt_act = (FUSE_VIRT_ACT and FUSE_IN_TAIL and Wf is None and n_hidden >= 1 and not resid and 256 <= F_ <= 512
                    and Np % 128 == 0 and Nt >= 32 and _dense_x6_ok(F_, Nt))
        va = (xr.view(Nt, 2), Wc.contiguous(), bc, LB, Np) if virt_act else None
        h = None if virt_act else torch.empty(F_, ldn, dtype=torch.float32, device=dev)
        if virt_act:
            _note('dec.virt_act')
        elif Wf is not None:
            Ff = Wf.shape[0]
            zx = Wl.shape[1] if (four_x6 and Wl is not None) else 0
            feat_all = torch.empty(Ff + zx, ldn, dtype=torch.float32, device=dev)
            feat = feat_all[:Ff]
            call('tvae_fourier_fwd', xr, Wf.contiguous(), bf.contiguous(), sigma, feat, ldn, Ff, Nt)
            if four_x6:
                _note('dec.four_x6')
                if zx:
                    feat_all[Ff:, :Nt].view(zx, B, Np).copy_(z.contiguous().t().unsqueeze(2).expand(zx, B, Np))
                Wfull = torch.cat([Wc, Wl], 1) if zx else Wc
                # h3 (round 4): the streamed operand is cos(.) <= 1 over the latent rows: its bound max(1, max |z|) is known
                # without looking at it; the layer's OUTPUT is then bounded by the largest absolute row sum of the weights
                p_c = parts() if parts() != 2 else 2
                if p_c == 2:
                    feat_amax = torch.clamp(_inf_norm(z), min=1.0) if zx else torch.ones(1, dtype=torch.float32, device=dev)
                    # per-unit bound of this layer's output (the weight gradient of the layer BEHIND it takes one
                    # scale per row of its X operand = per unit here), and its maximum for the forward of that layer
                    h_rows = Wfull.detach().abs().sum(1) * feat_amax + bc.detach().abs()
                    h_bound = h_rows.amax().reshape(1)
                    _note('dec.four_h3')
                w3c = _split_weight(Wfull, F_, Ff + zx, False, 'x6_dense_wc', nparts=p_c)
                with _timed('tvae_linear_fwd_x6', p_c):
                    call('tvae_linear_fwd_x6', w3c, feat_all, bc, None, h, F_, Nt, Ff + zx, ldn, ldn, act, LRELU_SLOPE,
                         None, None, None, None, None, None, None, 0, None, p_c, feat_amax if p_c == 2 else None)
            else:
                call('tvae_linear_fwd', Wc.contiguous(), feat, bc, LB, Np, None, h, F_, Nt, Ff, ldn, ldn, act, LRELU_SLOPE)
        else:
            call('tvae_dec_l0_fwd', xr, Wc.contiguous(), bc, LB, h, ldn, F_, Nt, Np, act, LRELU_SLOPE)
            if parts() == 2 and _dense_x6_ok(F_, Nt):      # the stored coordinate layer: |act(pre)| <= |pre| <= this bound
                h_rows = _inf_norm(xr) * Wc.detach().abs().sum(1) + \
                    ((bc.detach()[None, :] + LB).abs().amax(0) if LB is not None else bc.detach().abs())      # per unit
                h_bound = h_rows.amax().reshape(1)
        hs = [h]
        yh = torch.empty(B, Np, n_out, dtype=torch.float32, device=dev)
        fused_out = False
        sbits = None
        # Round 6: every hidden layer whose input is a STORED activation runs h3 -- the launch that stores hs[l] also leaves
        # max |hs[l]| in a word (tvae_linear_fwd_x6 y_amax, atomic max from its epilogue), the measured bound of the launch that
        # streams it next (forward here, weight gradient in the backward).  An analytic row-sum chain would lose ~5 bits of
        # fp16's exponent headroom per layer; a measured maximum loses none, whatever the depth.
        meas_words = torch.zeros(max(n_hidden, 1), dtype=torch.float32, device=dev) if (
            H3_DEEP and parts() == 2 and n_hidden >= 2 and _dense_x6_ok(F_, Nt)) else None
        h_meas = [None] * (n_hidden + 1)          # h_meas[l]: measured max |hs[l]| (one device word) or None
        rows_l = h_rows                           # per-unit bound of hs[l] by the row-sum chain (None: not available)
        rows_all = [rows_l]
        for li, (W, b) in enumerate(hidden):
            hn = None
            if _dense_x6_ok(F_, Nt):
                # h3 instances: the layer whose streamed operand is the recomputed first-layer activation, or (round 4) the
                # stored output of the first layer under its bound, or (round 6) a deeper layer's stored input under its
                # measured maximum
                h3_mem = parts() == 2 and not (va and li == 0) and (
                    (li == 0 and h_bound is not None and not resid) or (li > 0 and h_meas[li] is not None))
                x_bound = (h_bound if li == 0 else h_meas[li]) if h3_mem else None
                p_l = 2 if (parts() == 2 and ((va and li == 0) or h3_mem)) else _p3()
                if h3_mem:
                    _note('dec.hidden_h3_mem' if li == 0 else 'dec.hidden_h3_meas')
                w3 = _split_weight(W, F_, F_, False, 'x6_dense_w', nparts=p_l)
                # the last hidden layer also applies the single-output Linear that follows it (one pass less over h)
                fuse = FUSE_COLDOT and li == n_hidden - 1 and n_out == 1 and F_ <= 512
                # the backward of the single-output Linear behind the last hidden layer needs only the SIGN of this
                # layer's LeakyReLU output (two-valued implicit gradient): one bit per element, stored by this launch
                bits_ok = (li == n_hidden - 1 and FUSE_SIGN_BITS and FUSE_VIRT_GRAD and n_out == 1 and not resid
                           and act == ACT_LRELU and F_ >= 256 and Nt % 32 == 0)
                if bits_ok and not infer:
                    sbits = torch.empty(F_, Nt // 32, dtype=torch.int32, device=dev)
                    _note('dec.sign_bits')
                # Round 4: with the fused output dot AND the sign bits, nothing downstream needs this layer's activation
                # itself -- the backward takes [H > 0] from the bits and dWo from the identity sum_k W[m][k] G[m][k] +
                # b[m] g0[m] (include/tvae_hip.h: tvae_linear_dgrad_x6 vg_bits) -- so the 2.1 GB tensor is neither written
                # here nor read there.  Same conditions as the backward's row-sum fusion (it forms g0).
                # (inference: same launch without the bits -- only the fused output dot leaves it)
                no_h = (FUSE_NO_H and fuse and (sbits is not None or (infer and bits_ok)) and li == n_hidden - 1 and
                        FUSE_ROW_SUMS and F_ <= 512 and Nt % 128 == 0 and not resid)
                if no_h:
                    _note('dec.no_h_inference' if infer else 'dec.no_h')
                else:
                    hn = torch.empty(F_, ldn, dtype=torch.float32, device=dev)
                # this layer's output feeds another hidden layer: leave its maximum for that launch
                emit = (meas_words[li:li + 1] if (meas_words is not None and li + 1 < n_hidden and hn is not None and not fuse)
                        else None)
                with _timed('tvae_linear_fwd_x6', p_l):
                    call('tvae_linear_fwd_x6', w3, hs[-1], b, hs[-1] if resid else None, hn, F_, Nt, F_, ldn, ldn, act,
                         LRELU_SLOPE, Wo.contiguous() if fuse else None, bo if fuse else None, yh if fuse else None,
                         *(va if va and li == 0 else (None, None, None, None, 0)), sbits if li == n_hidden - 1 else None,
                         p_l, x_bound, emit)
                h_meas[li + 1] = emit
                if rows_l is not None and parts() == 2:      # |act(W h + b [+ h])| <= |W| rows + |b| [+ rows], unit by unit
                    rows_l = torch.addmv(b.detach().abs(), W.detach().abs(), rows_l) + (rows_l if resid else 0.0)
                else:
                    rows_l = None
                rows_all.append(rows_l)
                fused_out = fuse
                _note('dec.fused_out' if fuse else 'dec.hidden_x6')
            else:
                hn = torch.empty(F_, ldn, dtype=torch.float32, device=dev)
                call('tvae_linear_fwd', W.contiguous(), hs[-1], b, None, 1, hs[-1] if resid else None, hn, F_, Nt, F_,
                     ldn, ldn, act, LRELU_SLOPE)
                rows_l = None
                rows_all.append(None)
            hs.append(hn)
        if not fused_out:
            call('tvae_coldot', hs[-1], ldn, F_, Nt, Wo.contiguous(), 1, F_, bo, n_out, yh)
        if infer:
            return yh
        ctx.save_for_backward(xr, z if Wl is not None else None, feat, LB if virt_act else None, *hs,
                              *[p for p in params if p is not None])
        ctx.meta = (act, resid, sigma, n_hidden, Wl is not None, Wf is not None, B, Np)
        ctx.arith = get_gemm_mode()
        ctx.sbits = sbits
        ctx.h_bound, ctx.feat_amax, ctx.h_rows = h_bound, feat_amax, h_rows
        ctx.h_meas, ctx.rows_all = h_meas, rows_all
        return yh

    @staticmethod
    @_in_forward_arithmetic
    def backward(ctx, gy):
        act, resid, sigma, n_hidden, has_l, has_f, B, Np = ctx.meta
        sv = list(ctx.saved_tensors)
        xr, z, feat, LB = sv[0], sv[1], sv[2], sv[3]
        hs = sv[4:4 + n_hidden + 1]
        ps = sv[4 + n_hidden + 1:]
        it = iter(ps)
        Wc, bc = next(it), next(it)
        Wl = next(it) if has_l else None
        hidden = [(next(it), next(it)) for _ in range(n_hidden)]
        Wo, bo = next(it), next(it)
        Wf, bf = (next(it), next(it)) if has_f else (None, None)
        Nt = B * Np
        F_, n_out = Wc.shape[0], Wo.shape[0]
        dev = xr.device
        ldn = _dec_ld(Nt, F_, n_hidden, has_f)           # (row stride of every stored [features][Nt] tensor, as in forward)
        gy = gy.contiguous().view(Nt, n_out)
        gy_max = []                                      # max |gy|, formed once (one reduction over the output gradient) on first use

        def gy_amax():
            if not gy_max:
                gy_max.append(_inf_norm(gy))
            return gy_max[0]
        # last layer: one pass over h gives d (pre-activation gradient), its row sums and dWo
        gyT = gy.t().contiguous()
        dbo = _rowsum(gyT, n_out, Nt)
        # with a single output the gradient of the last hidden activation, Wo[f] gy[n] act'(h[f][n]), is cheap to
        # re-form inside the two GEMMs that consume it: it is then never written (dec_out_bwd only produces row sums)
        virt = (FUSE_VIRT_GRAD and n_out == 1 and n_hidden >= 1 and not resid and _dense_x6_ok(F_, Nt) and F_ >= 256
                and Nt % 16 == 0)
        d = None if virt else torch.empty(F_, ldn, dtype=torch.float32, device=dev)
        tot = torch.empty(1 + n_out, F_, dtype=torch.float32, device=dev)
        # two-valued implicit gradient: the data-gradient launch of the last hidden layer streams H anyway and returns the
        # two row sums this layer's backward needs of it (tot[0] = bias gradient below, tot[1] = dWo): no pass of its own
        fuse_rs = (FUSE_ROW_SUMS and virt and act == ACT_LRELU and F_ <= 512 and Nt % 128 == 0)
        no_h = hs[-1] is None                            # the forward did not store the last hidden activation (dec.no_h)
        _expect(not no_h or (fuse_rs and ctx.sbits is not None), 'decoder backward without the saved activation needs the bits path')
        if not fuse_rs:
            part = workspace(dev, ((Nt + 1023) // 1024) * F_ * (1 + n_out))
            call('tvae_dec_out_bwd', gy, n_out, Wo.contiguous(), hs[-1], ldn, d, ldn, F_, Nt, act, LRELU_SLOPE, part,
                 part.numel(), tot)
        else:
            _note('dec.row_sums_in_dgrad')
        vg = (Wo.contiguous().view(-1), gy.view(-1), act) if virt else None
        if virt:
            _note('dec.virt_grad')
        dWo, drow = tot[1:], tot[0]
        grads_hidden = []
        fused_in = False
        # Round 6 (h3 for every hidden layer): one bound word per stored gradient tensor -- the top one analytically
        # (|d[m][n]| <= max_m sum_o |Wo[o][m]| max |gy|: act' <= 1), every later one measured by the data-gradient launch that
        # stores it (tvae_linear_dgrad_x6 y_amax)
        h3_deep = H3_DEEP and parts() == 2 and _dense_x6_ok(F_, Nt)
        # (words for the measured maxima: only where a stored gradient is streamed again -- a second hidden layer, or the Fourier
        #  first layer behind the first one; the single-hidden-layer configurations need none and launch no fill)
        dmeas = torch.zeros(n_hidden + 1, dtype=torch.float32, device=dev) if (h3_deep and (n_hidden >= 2 or has_f)) else None
        d_bnd = None                                     # bound word of the stored gradient `d` (None: none / not stored)
        if h3_deep and not virt:
            d_bnd = (gy_amax() * Wo.detach().abs().sum(0).amax()).reshape(1)
        for li in range(n_hidden - 1, -1, -1):
            W, b = hidden[li]
            hprev = hs[li]
            use_vg = vg is not None and li == n_hidden - 1
            dsrc = hs[-1] if use_vg else d             # implicit operand: pass the saved activation instead
            va = (xr.view(Nt, 2), Wc.contiguous(), bc, LB, Np) if hprev is None else None   # recomputed first layer
            sbits = ctx.sbits if (use_vg and act == ACT_LRELU) else None      # [h > 0] as stored bits (two-valued form)
            from_bits = use_vg and no_h
            # h3 with the layer's input read from memory (round 4): the stored first-layer output under its forward bound,
            # times max |gy| (the operand of the two-valued form is gy[n] X[k][n])
            xg_amax = None
            if parts() == 2 and use_vg and sbits is not None and va is None and li == 0 and ctx.h_bound is not None:
                # one bound per unit (row of the X operand) where the forward formed them, else the tensor's
                xg_amax = (ctx.h_rows if ctx.h_rows is not None else ctx.h_bound) * gy_amax()      # (vg[1] is gy: n_out == 1)
            elif parts() == 2 and use_vg and sbits is not None and va is None and li > 0 and ctx.h_meas[li] is not None:
                # (round 6) a deeper layer's stored input under its measured maximum, capped per unit by the row-sum chain
                xb_ = ctx.h_meas[li] if ctx.rows_all[li] is None else torch.minimum(ctx.rows_all[li], ctx.h_meas[li])
                xg_amax = xb_ * gy_amax()
            # plain form (both operands stored tensors) in h3: bound of d (one word) and of the layer's input -- the first
            # layer's analytic per-unit bound, or a deeper layer's measured maximum (capped unit by unit by the row-sum chain
            # where the forward formed it: a unit far below the others keeps its own scale)
            plain_h3 = h3_deep and not use_vg and d_bnd is not None and va is None
            xw_amax = None
            if plain_h3:
                if li == 0:
                    xw_amax = ctx.h_rows if ctx.h_rows is not None else ctx.h_bound
                elif ctx.h_meas[li] is not None:
                    xw_amax = ctx.h_meas[li] if ctx.rows_all[li] is None else torch.minimum(ctx.rows_all[li], ctx.h_meas[li])
            if from_bits:                              # + rowdot[m] = sum_k W[m][k] G[m][k] for dWo (the dgrad launch below)
                dW, rowdot = _wgrad(None, hprev, F_, Nt, F_, vg, va, act, sbits, rowdot_w=W, x_amax=xg_amax, ld=ldn)
            elif plain_h3 and xw_amax is not None:
                _note('dec.wgrad_h3_meas')
                dW, rowdot = _wgrad(dsrc, hprev, F_, Nt, F_, None, None, act, None, a_amax=d_bnd, x_amax=xw_amax, ld=ldn), None
            else:
                dW, rowdot = _wgrad(dsrc, hprev, F_, Nt, F_, vg if use_vg else None, va, act, sbits, x_amax=xg_amax, ld=ldn), None
            db = drow if drow is not None else _rowsum(d, F_, Nt, ld=ldn)
            drow = None
            # the data gradient of the FIRST hidden layer can feed the coordinate layer's backward from its epilogue
            # (coordinate gradient + per-panel row sums); its result is then never written
            fuse_in = (FUSE_IN_TAIL and li == 0 and not has_f and not resid and F_ <= 512 and Np % 128 == 0 and
                       _dense_x6_ok(F_, Nt))
            dprev = None if fuse_in else torch.empty(F_, ldn, dtype=torch.float32, device=dev)
            if _dense_x6_ok(F_, Nt):
                # LeakyReLU: the implicit gradient in its two-valued form (3 MFMAs per block instead of 6)
                two_val = use_vg and act == ACT_LRELU
                # h3 instances: the exact 0 / 1 operand (two products), or (round 6) a stored gradient under its bound word
                p_d = 2 if (parts() == 2 and (two_val or (not use_vg and d_bnd is not None))) else _p3()
                if p_d == 2 and not two_val:
                    _note('dec.dgrad_h3_meas')
                emit = dmeas[li:li + 1] if (dmeas is not None and not fuse_in) else None      # max |dprev| for its consumers
                if two_val:
                    w3t, csum = _split_weight(W, F_, F_, True, 'x6_dense_wt', scale=vg[0], nparts=p_d)
                    _note('dec.virt_grad_2val')
                else:
                    w3t, csum = _split_weight(W, F_, F_, True, 'x6_dense_wt', nparts=p_d), None
                if fuse_in:
                    gxr_f = torch.empty(B, Np, 2, dtype=torch.float32, device=dev)
                    part_f = workspace(dev, (Nt // 128) * F_ * 3)
                rs = two_val and fuse_rs
                rs_part = _scratch(dev, 'dec_rs_part', (Nt // 128) * F_ * 2) if rs else None
                with _timed('tvae_linear_dgrad_x6', p_d, two_val):
                    call('tvae_linear_dgrad_x6', w3t, dsrc, d if resid else None, hprev, dprev, F_, Nt, F_, ldn, ldn, act,
                         LRELU_SLOPE, xr.view(Nt, 2) if fuse_in else None, Wc.contiguous() if fuse_in else None,
                         gxr_f if fuse_in else None, part_f if fuse_in else None, part_f.numel() if fuse_in else 0,
                         vg[0] if (use_vg and not two_val) else None, vg[1] if use_vg else None, csum,
                         bc if va else None, LB if va else None, Np if va else 0,
                         rs_part, rs_part.numel() if rs else 0, vg[0] if rs else None, dbo if rs else None,
                         tot[0] if rs else None, tot[1] if rs else None, p_d,
                         sbits if from_bits else None, rowdot, b if from_bits else None,
                         d_bnd if (p_d == 2 and not two_val) else None, emit)
                if two_val and li == 0 and n_hidden == 1 and has_f and parts() == 2:
                    # bound of the gradient this launch leaves in `dprev` (read again by the Fourier first layer's backward):
                    # |dX[k][n]| <= max |gy| * sum_m |wo[m] W[m][k]|
                    ctx.d_bound = gy_amax() * (W.detach().abs() * vg[0].detach().abs()[:, None]).sum(0).amax().reshape(1)
                fused_in = fuse_in
                if fuse_in:
                    _note('dec.fuse_in')
                d_bnd = emit
            else:
                call('tvae_linear_dgrad', W.contiguous(), d, d if resid else None, hprev, dprev, F_, Nt, F_, ldn, ldn,
                     act, LRELU_SLOPE)
                d_bnd = None
            d = dprev
            grads_hidden.append((dW, db))
        grads_hidden.reverse()
        # first layer: d is the pre-activation gradient [F][Nt]
        Simg = torch.empty(B, F_, dtype=torch.float32, device=dev)
        dbc = torch.empty(F_, dtype=torch.float32, device=dev)
        gxr = torch.empty(B, Np, 2, dtype=torch.float32, device=dev) if not fused_in else gxr_f
        if fused_in:
            dWc = torch.empty(F_, 2, dtype=torch.float32, device=dev)
            call('tvae_dec_in_total', part_f, B, Np // 128, F_, Simg, dbc, dWc)
        elif has_f:
            call('tvae_rowdot_seg', d, ldn, None, 1, F_, Nt, Np, Simg)
            call('tvae_seg_sum', Simg, B, F_, dbc, 1.0, 0)
        else:
            # one pass over d: coordinate gradient, per-image sums, bias and coordinate-weight gradients
            dWc = torch.empty(F_, 2, dtype=torch.float32, device=dev)
            part = workspace(dev, B * ((Np + 1023) // 1024) * F_ * 3)
            call('tvae_dec_in_bwd', d, ldn, xr.view(Nt, 2), Wc.contiguous(), F_, B, Np, gxr, Simg, dbc, dWc, part,
                 part.numel())
        dWl = dz = None
        if has_l:
            zd = Wl.shape[1]
            dWl = torch.empty(F_, zd, dtype=torch.float32, device=dev)
            dz = torch.empty(B, zd, dtype=torch.float32, device=dev)
            call('tvae_latent_bwd', Simg, Wl.contiguous(), z, dWl, dz, B, F_, zd)
        if has_f:
            Ff = Wf.shape[0]
            # h3 for the Fourier first layer's backward (round 4): both operands come from memory with known bounds -- the
            # features are cosines (<= 1), d is bounded by the two-valued data gradient that produced it (ctx.d_bound)
            d_bound = getattr(ctx, 'd_bound', None) if parts() == 2 else None
            if d_bound is None and parts() == 2:
                d_bound = d_bnd                          # (round 6) measured by the launch that stored d
            one = torch.ones(1, dtype=torch.float32, device=dev) if d_bound is not None else None
            dWc = _wgrad(d, feat, F_, Nt, Ff, a_amax=d_bound, x_amax=one, ld=ldn)
            dfeat = torch.empty(Ff, ldn, dtype=torch.float32, device=dev)
            if _dense_x6_ok(Ff, Nt):
                p_f = 2 if d_bound is not None else _p3()
                w3t = _split_weight(Wc, Ff, F_, True, 'x6_dense_wct', nparts=p_f)
                with _timed('tvae_linear_dgrad_x6', p_f):
                    call('tvae_linear_dgrad_x6', w3t, d, None, None, dfeat, F_, Nt, Ff, ldn, ldn, ACT_NONE, LRELU_SLOPE,
                         None, None, None, None, 0, None, None, None, None, None, 0, None, 0, None, None, None, None, p_f,
                         None, None, None, d_bound)
            else:
                call('tvae_linear_dgrad', Wc.contiguous(), d, None, None, dfeat, F_, Nt, Ff, ldn, ldn, ACT_NONE, LRELU_SLOPE)
            call('tvae_fourier_bwd', xr, Wf.contiguous(), bf.contiguous(), sigma, dfeat, ldn, Ff, Nt, gxr)
        out = [gxr, dz, None, None, None, None, dWc, dbc, dWl]
        for (dW, db) in grads_hidden:
            out += [dW, db]
        out += [dWo, dbo, None, None]
        return tuple(out)


LIK_KIND = {'bce': 0, 'bce3': 0, 'gauss': 1, 'gauss_var': 2}


class LogLikFn(torch.autograd.Function):
    """Per-image log-likelihood over flat vectors (train_mnist.py:288-291; train_galaxy.py:288-292;
    train_particles.py:284-296,336-338).  yh (B, ...), y (B, ...) -> lp [B]."""

    @staticmethod
    def forward(ctx, yh, y, kind):
        B = y.shape[0]
        ctx.shape = tuple(yh.shape)
        y = y.contiguous().view(B, -1)
        yh = yh.contiguous().view(B, -1)
        L = y.shape[1]
        assert yh.shape[1] == (2 * L if kind == 2 else L), (yh.shape, y.shape, kind)
        lp = torch.empty(B, dtype=torch.float32, device=yh.device)
        call('tvae_loglik_fwd', yh, y, lp, B, L, kind)
        ctx.save_for_backward(yh, y)
        ctx.kind = kind
        return lp

    @staticmethod
    def backward(ctx, g):
        yh, y = ctx.saved_tensors
        B, L = y.shape
        gyh = torch.empty_like(yh)
        call('tvae_loglik_bwd', yh, y, g.contiguous(), gyh, B, L, ctx.kind)
        return gyh.view(ctx.shape), None, None


class ElboFn(torch.autograd.Function):
    """(elbo f64, log_p f32, kl f64) of a minibatch from the per-image terms lp [B], kl [B] (train_mnist.py:282,291-292:
    log_p = lp.mean(), kl_div = kl.double().mean(), elbo = log_p - kl_div) -- one launch forward, one backward, instead of a
    dozen ATen launches for two means, a cast, a subtraction and their chain rules."""

    @staticmethod
    def forward(ctx, lp, kl):
        B = lp.shape[0]
        _expect(lp.dim() == 1 and tuple(kl.shape) == (B,) and B >= 1, 'per-image ELBO terms must be [B] vectors')
        dev_ = lp.device
        elbo = torch.empty((), dtype=torch.float64, device=dev_)
        logp = torch.empty((), dtype=torch.float32, device=dev_)
        kld = torch.empty((), dtype=torch.float64, device=dev_)
        call('tvae_elbo_reduce', lp.contiguous(), kl.contiguous(), B, elbo, logp, kld)
        ctx.B = B
        ctx.set_materialize_grads(False)
        return elbo, logp, kld

    @staticmethod
    def backward(ctx, g_elbo, g_logp, g_kld):
        B = ctx.B
        ref = next(g for g in (g_elbo, g_logp, g_kld) if g is not None)
        g_lp = torch.empty(B, dtype=torch.float32, device=ref.device)
        g_kl = torch.empty(B, dtype=torch.float32, device=ref.device)

        def as_(g, dt):
            return None if g is None else g.to(dt).contiguous()

        call('tvae_elbo_reduce_bwd', as_(g_elbo, torch.float64), as_(g_logp, torch.float32), as_(g_kld, torch.float64), B, g_lp,
             g_kl)
        return g_lp, g_kl


class CtfFn(torch.autograd.Function):
    """Per-image CTF filter of the reconstruction (train_particles.py:298-302): depthwise cross-correlation with an
    odd kernel, zero padding kc//2.  y_mu (B, n*n), ctf (B, 1, kc, kc) -> (B, n*n).  The filters are constants."""

    @staticmethod
    def forward(ctx, y_mu, ctf, n):
        B, kc = ctf.shape[0], ctf.shape[-1]
        _expect(kc % 2 == 1 and ctf.numel() == B * kc * kc and y_mu.numel() == B * n * n,
                f'CTF filters {tuple(ctf.shape)} / reconstruction {tuple(y_mu.shape)} do not match n = {n}')
        y_mu = y_mu.contiguous()
        ctf = ctf.contiguous()
        out = torch.empty_like(y_mu)
        call('tvae_ctf_corr', y_mu, ctf, out, B, n, kc, 0)
        ctx.save_for_backward(ctf)
        ctx.n = n
        return out

    @staticmethod
    def backward(ctx, g):
        (ctf,) = ctx.saved_tensors
        B, kc = ctf.shape[0], ctf.shape[-1]
        gi = torch.empty_like(g)
        call('tvae_ctf_corr', g.contiguous(), ctf, gi, B, ctx.n, kc, 1)
        return gi, None, None


class MaskedLogLikFn(torch.autograd.Function):
    """Gaussian log-likelihood inside a circle of `radius` pixels centred at the inferred translation
    (train_particles.py:309-333,338).  The mask is built on the device from dx (the reference builds it on the host
    with numpy every step); it carries no gradient, like the reference's dx.detach()."""

    @staticmethod
    def forward(ctx, yh, y, dx, spacing, radius, n):
        B = y.shape[0]
        ctx.shape = tuple(yh.shape)
        _expect(yh.numel() == B * n * n and y.numel() == B * n * n and tuple(dx.shape) == (B, 2), 'masked likelihood shapes')
        yh = yh.contiguous().view(B, -1)
        y = y.contiguous().view(B, -1)
        dx = dx.detach().contiguous()
        lp = torch.empty(B, dtype=torch.float32, device=yh.device)
        call('tvae_loglik_masked_fwd', yh, y, dx, 1.0 / spacing, float(radius), B, n, lp)
        ctx.save_for_backward(yh, y, dx)
        ctx.cfg = (spacing, radius, n)
        return lp

    @staticmethod
    def backward(ctx, g):
        yh, y, dx = ctx.saved_tensors
        spacing, radius, n = ctx.cfg
        gyh = torch.empty_like(yh)
        call('tvae_loglik_masked_bwd', yh, y, dx, 1.0 / spacing, float(radius), y.shape[0], n, g.contiguous(), gyh)
        return gyh.view(ctx.shape), None, None, None, None, None


def adam_flat(p, g, m, v, step, lr, b1=0.9, b2=0.999, eps=1e-8, grad_scale=1.0):
    """Fused Adam on flat buffers (torch.optim.Adam defaults; reference train_mnist.py:579,323)."""
    bc1 = 1.0 - b1 ** step
    bc2 = 1.0 - b2 ** step
    call('tvae_adam_flat', p, g, m, v, p.numel(), lr, b1, b2, eps, bc1, math.sqrt(bc2), grad_scale)
