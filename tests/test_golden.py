"""The committed golden vectors are reproduced bit-for-bit by the oracle (CPU)."""
import os

import numpy as np
import pytest

from conftest import ROOT
from util import CASES, TERMS, build, g

GOLD = np.load(os.path.join(ROOT, "tests", "golden", "golden_small.npz"))
SEED = 20220101


@pytest.mark.parametrize("name", list(CASES))
def test_oracle_reproduces_golden(oracle, name):
    s, dt, kT, flags = build(oracle, name)
    assert np.array_equal(s.positions(), GOLD[f"{name}/x0"])          # the synthetic inputs are deterministic
    for t, m in TERMS.items():
        assert np.allclose(s.forces(m), GOLD[f"{name}/F_{t}"], rtol=1e-12, atol=1e-12)
        assert np.allclose(s.energy(m), GOLD[f"{name}/E_{t}"], rtol=1e-12, atol=1e-12)
    s.begin_phase()
    s.run(10, dt, kT, seed=SEED, flags=flags)
    assert np.allclose(s.positions(), GOLD[f"{name}/x_philox10"], rtol=0, atol=1e-12)
