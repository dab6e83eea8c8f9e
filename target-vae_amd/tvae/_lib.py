"""ctypes binding of libtvae_hip.so (C ABI declared in include/tvae_hip.h).

The product path has NO CPU fallback: if the HIP library is missing or a tensor is not a contiguous
fp32/int32 CUDA(HIP) tensor, the call raises.  PyTorch is used only for device memory and streams.
"""
from __future__ import annotations

import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# TVAE_LIB: another build of the library (same-box A/B of two builds: profiles/tools/ab_kernels.sh TVAE_LIB "a.so b.so")
LIB_PATH = os.environ.get('TVAE_LIB') or os.path.join(os.path.dirname(_HERE), 'csrc', 'build', 'libtvae_hip.so')

# signature codes: p = device pointer (tensor or None), i = int, l = long, f = float; the trailing
# stream argument is appended automatically (torch.cuda.current_stream()).
SIGNATURES = {
    'tvae_rotate_bank_fwd': 'ppppiiii',
    'tvae_rotate_bank_bwd': 'ppppppiiiii',
    'tvae_conv1_fwd': 'ppppiiiiiiiif',
    'tvae_conv1_wgrad': 'ppppl iiiiiii'.replace(' ', ''),
    'tvae_bank_split3': 'ppliiii',
    'tvae_conv1_fwd_x6': 'ppppiiiiiiiif',
    'tvae_dy_split3': 'ppliiiiiii',
    'tvae_conv1_wgrad_x6': 'ppppliiiiiii',
    'tvae_conv1_fwd_dft': 'ppppppliiiiiiiifi',
    'tvae_conv1_wgrad_dft': 'pppppliiiiiiii',
    'tvae_dense_split3': 'plpliiipp',
    'tvae_dense_split2h': 'plpliiipp',
    'tvae_linear_fwd_x6': 'pppppiiillifpppppppipipp',
    'tvae_linear_dgrad_x6': 'pppppiiillifpppplpppppiplppppippppp',
    'tvae_dec_in_total': 'piiippp',
    'tvae_dgrad_rowsum_total': 'piippfpppp',
    'tvae_linear_wgrad_x6': 'ppppliiillippifppppipiplpppi',
    'tvae_linear_fwd': 'ppppippiiillif',
    'tvae_linear_dgrad': 'pppppiiillif',
    'tvae_linear_wgrad': 'ppppliiilli',
    'tvae_rowdot_seg': 'plpiiiipp',
    'tvae_seg_sum': 'pilpfi',
    'tvae_coldot': 'pliipiipip',
    'tvae_outer_mask': 'pipiiplpliiif',
    'tvae_act_bwd': 'ppplif',
    'tvae_dec_out_bwd': 'pipplplilifplp',
    'tvae_dec_in_bwd': 'plppiiipppppl',
    'tvae_heads_fwd': 'pplppliil',
    'tvae_heads_bwd': 'pplplpliilifplp',
    'tvae_attn_head_fwd': 'plpppppppiiiiffppppppppl',
    'tvae_attn_head_bwd': 'plppppppppiiiiffpppppppppl',
    'tvae_get_latent': 'plpppiiiifppp',
    'tvae_enc_tail_fwd_x6': 'pplpppiplplppilifip',
    'tvae_enc_tail_dgrad_x6': 'ppplippplilfi',
    'tvae_enc_tail_wgrad_x6': 'plplipppplilfip',
    'tvae_enc_tail_fwd_wide': 'ppplppiplplppilifip',
    'tvae_enc_tail_dgrad_wide': 'ppplippplplilfip',
    'tvae_enc_tail_wgrad_wide': 'pliplpplilpp',
    'tvae_rot_pool_fwd': 'ppppiiii',
    'tvae_rot_pool_bwd': 'ppppplpiiiiif',
    'tvae_coord_fwd': 'ppppii',
    'tvae_coord_bwd': 'ppppppii',
    'tvae_dec_l0_fwd': 'pppppliliif',   # xr, Wc, bc, LB, h, ldh(l), F(i), Ntot(l), Np(i), act(i), slope(f)
    'tvae_latent_bias': 'pppiii',
    'tvae_latent_bwd': 'pppppiii',
    'tvae_fourier_fwd': 'pppfplil',
    'tvae_fourier_bwd': 'pppfplilp',
    'tvae_loglik_fwd': 'pppiii',
    'tvae_loglik_bwd': 'ppppiii',
    'tvae_elbo_reduce': 'ppippp',
    'tvae_elbo_reduce_bwd': 'pppipp',
    'tvae_ctf_corr': 'pppiiii',
    'tvae_loglik_masked_fwd': 'pppffiip',
    'tvae_loglik_masked_bwd': 'pppffiipp',
    'tvae_adam_flat': 'pppplfffffff',
}

# pure host queries (no stream argument): name -> (argument codes, return code)
QUERIES = {
    'tvae_conv1_x6_supported': ('iiii', 'i'),
    'tvae_conv1_x6_bank_bytes': ('iiii', 'l'),
    'tvae_conv1_x6_dy_bytes': ('iiiiii', 'l'),
    'tvae_dense_x6_bytes': ('ii', 'l'),
    'tvae_conv1_dft_supported': ('iiiiiii', 'i'),
    'tvae_conv1_dft_at_floats': ('iiiiiii', 'l'),
    'tvae_conv1_dft_ws_floats': ('iiiiiii', 'l'),
    'tvae_conv1_dft_frame': ('iiiiiii', 'i'),
    'tvae_conv1_dft_ring': ('iiiiiii', 'i'),
    'tvae_enc_tail_wgrad_x6_ws_floats': ('l', 'l'),
    'tvae_enc_tail_wide_max_rows': ('', 'i'),
    'tvae_linear_wgrad_x6_ws_floats': ('iii', 'l'),
}

# trailing arguments a caller may leave out (beyond all-pointer tails, which are always optional)
OPTIONAL_TAIL = {'tvae_linear_wgrad_x6': 6}      # rd_w, rd_ldw, rd_rowdot, a_amax, x_amax (ABI 5), x_amax_rows (ABI 6)

_CT = {'p': ctypes.c_void_p, 'i': ctypes.c_int, 'l': ctypes.c_long, 'f': ctypes.c_float}
_lib = None


class TvaeHipError(RuntimeError):
    pass


def lib():
    """Load the shared library once; fail loudly when it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise TvaeHipError(
                f'{LIB_PATH} not found: build it with `make -C target-vae_amd/csrc` '
                '(or __graft_entry__.build()).  There is no CPU fallback.')
        L = ctypes.CDLL(LIB_PATH)
        for name, sig in SIGNATURES.items():
            fn = getattr(L, name)
            fn.restype = ctypes.c_int
            fn.argtypes = [_CT[c] for c in sig] + [ctypes.c_void_p]
        for name, (sig, ret) in QUERIES.items():
            fn = getattr(L, name)
            fn.restype = _CT[ret]
            fn.argtypes = [_CT[c] for c in sig]
        L.tvae_abi_version.restype = ctypes.c_int
        if L.tvae_abi_version() != ABI_VERSION:
            raise TvaeHipError(f'{LIB_PATH} has ABI version {L.tvae_abi_version()}, this package needs {ABI_VERSION}: rebuild')
        _lib = L
    return _lib


ABI_VERSION = 7
# Arithmetic of the matrix products.  The C ABI is stateless: the mode is host-side ROUTING only -- it decides which
# entry points tvae.ops calls and with which `parts`:
#   'h3'   (default) *_x6 / *_dft entry points with parts = 2: operands as TWO fp16 parts under a power-of-two tensor scale,
#          three products per block (two against an exact 0 / 1 operand), fp32 accumulate -- fp32-equivalent results (at
#          least as accurate against fp64 as the fp32 matrix pipe: profiles/experiments/f16_split_probe.hip, the parity
#          tests run in it at the same tolerances); launches without an h3 instance run 'x6'
#   'x6'   the same entry points with parts = 3: operands split EXACTLY into three bf16 numbers, six products
#   'f32'  the exact fp32-MFMA entry points
#   'bf16' parts = 1: operands rounded to one bf16 number -- the throughput mode of BASELINE.json configs 2 / 5, not
#          fp32-equivalent and never the default.
# Default from TVAE_GEMM; `with arithmetic('f32'): ...` scopes a different mode to a block, so two models with different
# arithmetic coexist in one process.
GEMM_MODES = ('f32', 'x6', 'h3', 'bf16')
_mode = os.environ.get('TVAE_GEMM', 'h3')
if _mode not in GEMM_MODES:
    raise TvaeHipError(f'TVAE_GEMM={_mode!r}: choose from {GEMM_MODES}')


def set_gemm_mode(mode: str) -> None:
    global _mode
    if mode not in GEMM_MODES:
        raise TvaeHipError(f'unknown GEMM mode {mode!r}; choose from {GEMM_MODES}')
    _mode = mode


def get_gemm_mode() -> str:
    return _mode


def split_pipe() -> bool:
    """True when matrix products go to the bf16 matrix pipe (*_x6 / *_dft entry points)."""
    return _mode in ('x6', 'h3', 'bf16')


def parts() -> int:
    """Parts per operand for the *_x6 / *_dft entry points in the current mode: 3 = exact bf16 split (six products), 2 = two
    fp16 parts (three products, "h3": entry points without an h3 instance run x6), 1 = plain bf16."""
    return {'bf16': 1, 'h3': 2}.get(_mode, 3)


class arithmetic:
    """Context manager: run the enclosed tvae.ops calls (forward AND the backward they record -- autograd Functions
    capture the mode at forward time) in the given arithmetic."""

    def __init__(self, mode: str):
        if mode not in GEMM_MODES:
            raise TvaeHipError(f'unknown GEMM mode {mode!r}; choose from {GEMM_MODES}')
        self.mode = mode

    def __enter__(self):
        global _mode
        self.old, _mode = _mode, self.mode
        return self

    def __exit__(self, *exc):
        global _mode
        _mode = self.old
        return False


# gradient all-reduce over RCCL behind the C ABI (include/tvae_hip.h; host functions, only the collective takes a stream)
RCCL_SYMBOLS = ('tvae_rccl_available', 'tvae_rccl_unique_id', 'tvae_rccl_comm_init', 'tvae_allreduce_flat',
                'tvae_rccl_comm_destroy')


def exported_symbols():
    return ['tvae_abi_version'] + sorted(SIGNATURES) + sorted(QUERIES) + list(RCCL_SYMBOLS)


class RcclComm:
    """An RCCL communicator owned through the C ABI (tvae_rccl_comm_init): `unique_id()` on one rank, the 128 bytes to every
    rank by any means, then `RcclComm(nranks, id, rank)` on every rank (a collective, on the current device).
    EXPERIMENTAL (TVAE_DP_ABI=1): it has run with one rank only; the default data-parallel path calls RCCL through
    torch.distributed.  `close()` (or garbage collection) destroys the communicator -- before destroy_process_group()."""

    @staticmethod
    def _check_instance():
        """tvae_rccl_available(): 1 = the RCCL instance already loaded in this process (torch's), 2 = a copy the library had to
        load itself.  Inside a process whose torch.distributed can use its own RCCL a second instance means two sets of
        communicators and proxy threads -- refused unless TVAE_RCCL_OWN_COPY=1 says the caller knows."""
        import sys
        src = lib().tvae_rccl_available()
        if not src:
            raise TvaeHipError('no RCCL library could be resolved in this process')
        if src == 2 and 'torch.distributed' in sys.modules and os.environ.get('TVAE_RCCL_OWN_COPY', '0') != '1':
            import torch.distributed as dist
            if dist.is_available() and dist.is_nccl_available():
                raise TvaeHipError('libtvae_hip.so would load a SECOND copy of RCCL beside the one torch.distributed uses '
                                   '(RTLD_NOLOAD found none): set TVAE_RCCL_OWN_COPY=1 to allow it')
        return src

    def __init__(self, nranks: int, id128: bytes, rank: int):
        L = lib()
        self._comm = ctypes.c_void_p()
        self.instance = self._check_instance()           # 1 = the process's own RCCL, 2 = a private copy
        buf = ctypes.create_string_buffer(bytes(id128), 128)
        rc = L.tvae_rccl_comm_init(ctypes.byref(self._comm), int(nranks), buf, int(rank))
        if rc:
            raise TvaeHipError(f'tvae_rccl_comm_init failed ({_rccl_rc(rc)})')

    @staticmethod
    def unique_id() -> bytes:
        RcclComm._check_instance()
        buf = ctypes.create_string_buffer(128)
        rc = lib().tvae_rccl_unique_id(buf)
        if rc:
            raise TvaeHipError(f'tvae_rccl_unique_id failed ({_rccl_rc(rc)})')
        return buf.raw

    def all_reduce_(self, t) -> None:
        """In-place sum all-reduce of a contiguous fp32 CUDA tensor on the current torch stream (asynchronous)."""
        if not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()):
            raise TvaeHipError('tvae_allreduce_flat: contiguous fp32 CUDA tensor expected')
        if not self._comm:
            raise TvaeHipError('tvae_allreduce_flat on a closed communicator')
        L = lib()
        L.tvae_allreduce_flat.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_long, ctypes.c_void_p]
        rc = L.tvae_allreduce_flat(self._comm, t.data_ptr(), t.numel(), torch.cuda.current_stream().cuda_stream)
        if rc:
            raise TvaeHipError(f'tvae_allreduce_flat failed ({_rccl_rc(rc)})')

    def close(self) -> None:
        if getattr(self, '_comm', None):
            lib().tvae_rccl_comm_destroy(self._comm)
            self._comm = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


_NCCL_RESULT = {1: 'ncclUnhandledCudaError', 2: 'ncclSystemError', 3: 'ncclInternalError', 4: 'ncclInvalidArgument',
                5: 'ncclInvalidUsage', 6: 'ncclRemoteError', 7: 'ncclInProgress'}


def _rccl_rc(rc: int) -> str:
    """Return codes of the tvae_rccl_* entry points: 100001 = no RCCL, 100100 + r = ncclResult_t r, else a hipError_t."""
    if rc == 100001:
        return 'no RCCL library resolved'
    if rc > 100100:
        return f'ncclResult_t {rc - 100100} {_NCCL_RESULT.get(rc - 100100, "?")}'
    return f'hipError_t {rc}'


def query(name, *args) -> int:
    """Pure host query of the library (geometry / workspace sizes); no GPU work."""
    sig, _ = QUERIES[name]
    if len(args) != len(sig):
        raise TvaeHipError(f'{name}: expected {len(sig)} arguments, got {len(args)}')
    return int(getattr(lib(), name)(*[int(a) for a in args]))


# float64 arguments (the reference's ELBO / KL dtype), by (entry point, argument position): elbo / kld of tvae_elbo_reduce,
# g_elbo / g_kld of tvae_elbo_reduce_bwd.  Every other pointer of those entry points is float* (a float64 `kl`
# passed where float* is expected would be reinterpreted silently).
_F64_OK = {('tvae_elbo_reduce', 3), ('tvae_elbo_reduce', 5), ('tvae_elbo_reduce_bwd', 0), ('tvae_elbo_reduce_bwd', 2)}


def _ptr(t, name, pos):
    if t is None:
        return None
    if not torch.is_tensor(t):
        raise TvaeHipError(f'{name} arg {pos}: expected a tensor or None, got {type(t)}')
    if not t.is_cuda:
        raise TvaeHipError(f'{name} arg {pos}: tensor must live on the GPU (no CPU fallback)')
    if (name, pos) in _F64_OK:
        if t.dtype != torch.float64:
            raise TvaeHipError(f'{name} arg {pos}: dtype {t.dtype}, the C ABI takes double* here')
    elif t.dtype not in (torch.float32, torch.int32):
        raise TvaeHipError(f'{name} arg {pos}: dtype {t.dtype} not supported (fp32 / int32 only)')
    if not t.is_contiguous():
        raise TvaeHipError(f'{name} arg {pos}: tensor must be contiguous')
    return t.data_ptr()


def call(name, *args):
    """Invoke a C-ABI entry point on the current torch stream."""
    L = lib()
    sig = SIGNATURES[name]
    if len(args) < len(sig) and (set(sig[len(args):]) == {'p'} or len(sig) - len(args) <= OPTIONAL_TAIL.get(name, 0)):
        # optional trailing arguments (added by later ABI versions): NULL pointers / zero strides
        args = args + tuple(None if c == 'p' else 0 for c in sig[len(args):])
    if len(args) != len(sig):
        raise TvaeHipError(f'{name}: expected {len(sig)} arguments, got {len(args)}')
    conv = []
    for pos, (c, a) in enumerate(zip(sig, args)):
        if c == 'p':
            conv.append(_ptr(a, name, pos))
        elif c == 'f':
            conv.append(float(a))
        else:
            conv.append(int(a))
    stream = torch.cuda.current_stream().cuda_stream
    rc = getattr(L, name)(*conv, stream)
    if rc != 0:
        raise TvaeHipError(f'{name} failed with hipError_t {rc}')
