"""CPU oracle for the embedding / interaction hot path.

TEST INFRASTRUCTURE ONLY.  This package is a plain-PyTorch / numpy restatement
(CPU, fp32, autograd) of what the reference's ``recsys/models``, the train step
in ``recsys/utils.py`` and the batching in ``recsys/dataset.py`` compute.  It is
pinned against golden vectors captured from the imported reference
(``tests/golden/*.npz``, generator ``tests/golden/make_golden.py``).

Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import it, and only as the checker / the timed CPU baseline.
Nothing under ``recsys_amd/`` imports it: the product path is the HIP library
and fails loudly when that library is missing.

Models with no counterpart in the reference (26-field FM / DeepFM over
Criteo-shaped inputs, DCN-v2) live in ``oracle/fields.py``; their parity is
pinned only as far as they reduce to the reference's 4-field formulas.
"""
