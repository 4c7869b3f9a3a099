"""Worst case of the 16-bit modes over seeded fixture draws and kernel variants (the table behind FP16_DRAW_BARS in tests/test_gpu_parity.py).

    python tests/diagnostics/fp16_margin.py [fp16|bf16]        (MI355X)
"""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
import torch

from rift_amd import _ffi
from tests import test_gpu_parity as T

if __name__ == "__main__":
    mode = sys.argv[1] if len(sys.argv) > 1 else "fp16"
    torch.cuda.set_device(0)
    _ffi.load_library()
    table = T.fp16_margin_table(_ffi, mode)
    for shape, worst in table.items():
        print(f"{mode} worst over 8 draws x 3 variants, {shape}: " + "  ".join(f"{k} {v:.3e}" for k, v in worst.items()))
