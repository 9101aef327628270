"""
CPU oracle for the pyGPSO hot path (GP fit + per-leaf UCB predict + ternary geometry + loop).

TEST INFRASTRUCTURE ONLY.  Nothing in ``pygpso_amd/`` imports this package.  Allowed importers:
``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` -- and there
only as the checker / the timed CPU baseline, never as the shipped compute path.

Parity status: PINNED.  The arithmetic lives in un-vendored third-party dependencies of the
reference (``gpflow>=2.0.0`` on TensorFlow, ``scipy`` L-BFGS-B; ``requirements.txt:5,12`` -- no
upper pins, no lock file; the notebooks were produced 2020-05-13, i.e. GPflow 2.0.x).  Neither is
installable here, so this package restates GPflow-2 ``GPR`` semantics in
float64 numpy/scipy and is pinned against every known-answer value the reference itself holds for
this path (``tests/golden/reference_goldens.json``: G1-G8 transcribed from
``tests/test_gp_surrogate.py``, ``tests/test_optimisation.py`` and the example notebooks' logged
outputs).  ``tests/test_oracle_goldens.py`` is the pin.
"""
