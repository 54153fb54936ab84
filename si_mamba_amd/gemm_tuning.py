"""Library-GEMM solution selection for the projections around the HIP kernels.

The four projections of every mixer, the encoder's 1x1 convolutions and the heads are plain library GEMMs
(hipBLASLt / rocBLAS through PyTorch-ROCm).  PyTorch's TunableOp can time every solution both libraries offer for
a GEMM shape and remember the winner; ``si_mamba_amd/tuned/gemm_gfx950.csv`` holds that choice for the shapes of
the benchmark configurations (made once on an MI355X by tools/tune_gemm.py; the file carries the library
versions it is valid for and PyTorch ignores it when they differ).  ``enable_tuned_gemms()`` switches TunableOp on
in read-only mode: listed shapes take the recorded solution, every other shape the library default; nothing is
timed or written at run time.  Measured on the bench step: 67.9 -> 62.5 ms (fp32, B=64).  bf16 (autocast): only the
plain GEMMs are listed; the library's candidate sweep for bf16 STRIDED-BATCHED GEMMs faults the GPU on dense operands
(tools/tune_gemm_offline.py, profiles/r02c_bf16_tune_fault*.log), so the mixer's bf16 projections keep the default.
"""
from __future__ import annotations

import os

DEFAULT_FILE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tuned", "gemm_gfx950.csv")


def enable_tuned_gemms(path: str | None = None) -> bool:
    """-> True when the solution file was found and handed to TunableOp."""
    import torch
    import torch.cuda.tunable as tunable
    path = DEFAULT_FILE if path is None else path
    if not torch.cuda.is_available() or not os.path.exists(path):
        return False
    tunable.enable(True)
    tunable.tuning_enable(False)          # look-up only
    tunable.record_untuned_enable(False)
    # Look-up only.  TunableOp rewrites its table to `filename` when the process ends: switch that off where the
    # PyTorch build has the switch (torch.cuda.tunable.write_file_on_exit); otherwise aim it at ONE fixed scratch
    # name, so the shipped file is never touched and runs do not leave a file per process behind.
    if hasattr(tunable, "write_file_on_exit"):
        tunable.write_file_on_exit(False)
    else:
        tunable.set_filename(os.path.join(os.environ.get("TMPDIR", "/tmp"), "simamba_tunableop_scratch.csv"))
    return bool(tunable.read_file(path))
