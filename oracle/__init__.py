"""CPU oracle for the DDPM spectrogram-synthesis hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is part of the shipped
product: only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg may import it, and there only as the checker.  The product
path (``music-spectrogram-diffusion_amd``) never imports this package and fails
loudly when its HIP library is missing.

Parity status (see DESIGN.md "Oracle"):
  * ``layers`` ops are PINNED against the reference's own known-answer tests
    (``layers_test.py``), restated in ``tests/test_oracle_kat.py``.
  * ``diffusion_utils`` / ``network`` / ``models`` have no reference tests and
    JAX cannot be imported in this environment: for those the oracle is a
    line-by-line restatement ("parity unpinned" by the reference; pinned only by
    the mathematical identities in ``tests/test_oracle_identities.py``).
"""
