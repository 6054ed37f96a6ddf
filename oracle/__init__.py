"""CPU oracle for the OmniServe quantized-inference hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is part of the product:
only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import it, and only as the checker.  The product path
(``omniserve_amd`` -> ``libomniserve_hip.so``) never routes through it and
fails loudly when the HIP library is missing.

Every function cites the reference file:line (relative to the upstream
mit-han-lab/omniserve checkout) whose arithmetic it restates.

Parity pin status
-----------------
* Weight-packing layout (``qweight``, ``s1_szeros``, ``s2_scales``,
  ``s2_zeros``): PINNED.  ``tests/golden/make_golden.py`` imports the
  reference's own ``W4A8OF16LinearDynamicInputScale.from_linear`` packer and
  the committed ``tests/golden/w4a8_pack_*.npz`` fixtures hold its outputs;
  ``tests/test_oracle_golden.py`` checks the oracle's closed-form packer
  against them bit-for-bit.
* Kernel arithmetic (GEMM epilogues, act-quant, norm, KV4, attention):
  PARITY UNPINNED.  The reference has no CPU path, no tests and no golden
  vectors for its CUDA kernels and they
  cannot be compiled here (nvcc + inline PTX).  These functions are a
  line-by-line restatement of the CUDA sources; integer stages are exact by
  construction, floating-point stages follow the reference's rounding points
  with IEEE division / sqrt in place of ``--use_fast_math`` approximations.
"""
