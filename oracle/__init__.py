"""CPU oracle for the SI-Mamba hot path.  TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import anything from this package -- and there only as the
checker, never as the thing measured or shipped.  The product package
(``si_mamba_amd``) never imports it and fails loudly when the HIP library is
missing.

Parity status (see DESIGN.md "Oracle"):
  * selective scan / causal conv1d / Mamba mixer: the arithmetic lives in the
    un-vendored third-party wheels ``mamba-ssm`` / ``causal-conv1d``
    (reference README.md:55-56; effective mamba-ssm version unpinned because
    models/point_mamba.py:26 imports ``mamba_ssm.modules.mamba2``).  The
    reference repo holds no tests or golden vectors for them, so this
    restatement of their published algorithm is **parity unpinned** by the
    reference; it is cross-checked against an independent float64 closed-form
    summation (tests/test_oracle_scan.py).
  * spectral ordering, token assembly (SAST / HLT), Block / MixerModel / create_block / _init_weights:
    **pinned to the reference's own code.**  ``oracle/pin_from_reference.py`` reads
    /root/reference/models/point_mamba.py (:115-272, :620-841, :889-898, :982-989, :1059-1112) and models/block.py
    (:17-76) at run time, compiles those definitions unmodified (``device='cuda'`` mapped to the CPU) and stores
    what they return in tests/golden/ref_*.npz.  tests/test_oracle_pinned.py holds this package to them (adjacency,
    orders, index maps, block data flow: bit for bit; eigenpairs: bit for bit on the LAPACK configuration that
    generated them, 3e-6 otherwise); tests/test_gpu_pinned.py and tests/test_gpu_spectral.py hold the HIP path to
    the same files.  The mixer INSIDE those blocks is still this package's restatement (previous item).
  * state-dict contract: pinned by the parameter table in the reference's
    ``logs/finetuned_hardest.log:100-426`` (tests/golden/param_table_*.json).
"""
