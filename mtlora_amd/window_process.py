"""Drop-in for kernels/window_process/window_process.py (reference :11-63): same class names, same
``apply`` signatures and sign convention, backed by the gfx950 gather kernels in csrc/window_process.hip.

    x_windows = WindowProcess.apply(x, B, H, W, C, -shift_size, window_size)              # roll(-s) + partition
    x         = WindowProcessReverse.apply(windows, B, H, W, C, shift_size, window_size)  # merge + roll(+s)

The module-level functions mirror the pybind module ``swin_window_process``
(swin_window_process.cpp:127-132).  Unlike the reference, bf16 is supported, H != W is exact, the
kernels run on the current stream, and the backward is the true adjoint (tested against autograd;
the reference's backward tests compare forward outputs only, unit_test.py:148-195).
"""
import torch

from .functional import (roll_and_window_partition_backward, roll_and_window_partition_forward,
                         window_merge_and_roll_backward, window_merge_and_roll_forward)

__all__ = ["WindowProcess", "WindowProcessReverse", "roll_and_window_partition_forward",
           "roll_and_window_partition_backward", "window_merge_and_roll_forward", "window_merge_and_roll_backward"]


class WindowProcess(torch.autograd.Function):
    @staticmethod
    def forward(ctx, input, B, H, W, C, shift_size, window_size):
        ctx.args = (B, H, W, C, shift_size, window_size)
        return roll_and_window_partition_forward(input, B, H, W, C, shift_size, window_size)

    @staticmethod
    def backward(ctx, grad_in):
        grad_out = roll_and_window_partition_backward(grad_in.contiguous(), *ctx.args)
        return grad_out, None, None, None, None, None, None


class WindowProcessReverse(torch.autograd.Function):
    @staticmethod
    def forward(ctx, input, B, H, W, C, shift_size, window_size):
        ctx.args = (B, H, W, C, shift_size, window_size)
        return window_merge_and_roll_forward(input, B, H, W, C, shift_size, window_size)

    @staticmethod
    def backward(ctx, grad_in):
        grad_out = window_merge_and_roll_backward(grad_in.contiguous(), *ctx.args)
        return grad_out, None, None, None, None, None, None
