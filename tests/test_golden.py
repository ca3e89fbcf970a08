"""The oracle against its committed regression vectors (tests/golden/oracle_regression.npz, made by
tests/golden/make_golden.py).  Self-generated: pins the oracle across rounds, not against the reference."""
import importlib.util
import os

import numpy as np


def test_oracle_matches_committed_vectors(repo_root):
    path = os.path.join(repo_root, "tests", "golden")
    spec = importlib.util.spec_from_file_location("make_golden", os.path.join(path, "make_golden.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    now = mod.build()
    with np.load(os.path.join(path, "oracle_regression.npz")) as z:
        assert set(z.files) == set(now)
        for k in z.files:
            np.testing.assert_allclose(now[k], z[k], rtol=1e-10, atol=1e-12, err_msg=k)
