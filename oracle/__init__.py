"""CPU oracle for the embedded-SCF hot path of UCL-CCS/Nbed.

TEST INFRASTRUCTURE ONLY.  Nothing under ``nbed_amd/`` (the product) may import
this package; only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` do, and there only as the checker.

Each function is a plain numpy (or C, under ``oracle/c``) restatement of the
reference's algorithm and cites the reference file:line it follows
(paths relative to the reference repository root).

Pinning status (see DESIGN.md "Oracle"):

* Every numpy/scipy line of the path that lives in the reference's own files
  (huzinaga_scf loop, Huzinaga operator, energy_elec, environment projector,
  environment deletion, SPADE, concentric localisation, spin-orbital scatter,
  post-embed energy arithmetic) is pinned: ``tests/golden/make_golden.py``
  imported the reference in the build container (with inert, name-only
  stand-ins for the absent ``pyscf``/``openfermion`` packages), ran the
  reference functions on deterministic toy inputs and stored input/output
  pairs under ``tests/golden/``; ``tests/test_oracle_golden.py`` checks this
  package against them.
* The arithmetic that lives inside the third-party dependency PySCF 2.9.0
  (J/K contraction, ``lib.diis.DIIS``, ``scf.hf.kernel``/CDIIS, ``ao2mo``) is
  NOT in /root/reference and PySCF is not installed: those pieces restate the
  published algorithms and are pinned by mathematical identities and by the
  real-molecule literals the reference's tests hold (global UHF water/STO-3G,
  tests/test_driver.py:52-61) through ``oracle.gto`` -- see DESIGN.md.
"""
