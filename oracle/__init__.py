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
  * spectral ordering: restated from reference models/point_mamba.py:620-841
    and executed with the same stock ``torch.linalg.eigh`` / ``topk`` /
    ``sort`` calls the reference makes (CPU, LAPACK).  The reference has no
    fixtures for it either: **parity unpinned**, but the library calls are
    the reference's own.
  * state-dict contract: pinned by the parameter table in the reference's
    ``logs/finetuned_hardest.log:100-426`` (tests/golden/param_table_*.json).
"""
