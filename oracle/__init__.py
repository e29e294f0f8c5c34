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
    JAX cannot be imported in this environment.  They are PINNED against outputs of
    the reference's OWN code: ``tests/golden/ref_shim.py`` installs a NumPy stand-in
    for the slice of jax / flax.linen those files use and
    ``tests/golden/make_ref_golden.py`` executes the reference's
    ``predict_batch_with_aux`` (models.py -> network.py / layers.py /
    diffusion_utils.py, imported from /root/reference) in float64 on seeded inputs;
    ``tests/test_ref_golden.py`` holds the oracle to those fixtures at 1e-9 (whole
    sampled segments, encodings, single decoder passes; every sampler / schedule /
    model-output / cross-attention branch) and the parameter tree to the one the
    reference's ``module.init`` creates.  What that does NOT pin: XLA's float32
    rounding (the stand-in's arithmetic is NumPy) and jax.random (noise is an
    input).
"""
