"""CPU oracle for the RIFT train_cbv policy-update hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``rift_amd/`` (the product) may import
this package; only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` do, and only as the checker / baseline.

Every function restates a piece of the reference algorithm in plain
PyTorch-CPU fp32 / NumPy and cites the reference file:line it follows
(paths relative to the upstream checkout root).

Pinning status
--------------
The reference holds no golden vectors for this path.
The oracle is pinned against outputs of the reference itself, imported on CPU
in the build container by ``tests/golden/gen_golden.py`` (fixtures committed
under ``tests/golden/``).  One boundary stays *parity unpinned*: the
third-party ``natten==0.14.6`` ``NeighborhoodAttention1D`` CUDA op is not in
the reference tree, so its published algorithm is restated in
``oracle/pluto_ref.py::neighborhood_attention_1d`` and the same restatement is
what the imported reference runs on.
"""
