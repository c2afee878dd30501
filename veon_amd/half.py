"""The 16-bit operand type of the MFMA path: bf16 (default) or IEEE fp16.

Every native kernel that takes half-precision operands (GEMM, attention,
LayerNorm-to-half, the conv kernels, the padded volumes / images, the hand-over
kernels) is compiled twice from the same sources (``veon_amd/build.py``):
``libveon_hip.so`` with bf16 and ``libveon_hip_f16.so`` with fp16 operands, both
accumulating in fp32 on ``v_mfma_f32_16x16x32_{bf16,f16}``.  The flavour is a
process-wide choice, made BEFORE models are built (they cache packed half weights):

    VEON_HALF=fp16 python bench.py --workload VEONL        # or
    veon_amd.half.set_half_dtype(torch.float16)

BASELINE configs[2] names bf16, configs[4] fp16; the reference itself runs fp32
(``# fp16 = dict(loss_scale='dynamic')`` is commented out in configs/veon/*.py).
fp16 keeps 3 more mantissa bits than bf16 (tighter parity, tests/test_half_mode_gpu.py)
at the price of a 65504 range: operands of the kernels are LayerNorm outputs, GELU
hiddens, BN/ReLU activations and softmax probabilities, the residual streams and
every accumulation stay fp32.
"""
import os

import torch

_BY_NAME = {'bf16': torch.bfloat16, 'bfloat16': torch.bfloat16,
            'fp16': torch.float16, 'float16': torch.float16, 'half': torch.float16}
_NAME = {torch.bfloat16: 'bf16', torch.float16: 'fp16'}


def _parse(dt):
    if isinstance(dt, str):
        if dt.lower() not in _BY_NAME:
            raise ValueError('half dtype %r: expected bf16 or fp16' % (dt,))
        return _BY_NAME[dt.lower()]
    if dt not in _NAME:
        raise ValueError('half dtype %r: expected torch.bfloat16 or torch.float16' % (dt,))
    return dt


_dtype = _parse(os.environ.get('VEON_HALF', 'bf16'))


def dtype():
    """torch dtype of the half operands of this process (torch.bfloat16 / float16)."""
    return _dtype


def name():
    """'bf16' or 'fp16'."""
    return _NAME[_dtype]


def is_half(dt):
    """``dt`` is THE half dtype of this process (a module's ``*_dtype`` switch)."""
    return dt is not None and dt == _dtype


def set_half_dtype(dt):
    """Select the flavour for everything built from now on; returns the previous
    dtype.  Modules built under the other flavour hold packed weights of the other
    type and raise on their next native call."""
    global _dtype
    prev, _dtype = _dtype, _parse(dt)
    return prev


class use:
    """``with half.use(torch.float16): ...`` (tests)."""

    def __init__(self, dt):
        self.dt = _parse(dt)

    def __enter__(self):
        self.prev = set_half_dtype(self.dt)
        return self

    def __exit__(self, *exc):
        set_half_dtype(self.prev)
        return False
