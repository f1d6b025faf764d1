"""prepare/refine restatements against fixtures recorded from the reference's own Python modules."""
import importlib
import json
import os

import numpy as np

from conftest import ROOT

prep = importlib.import_module("2022a-genome-dynamics_amd.prepare")
FX = np.load(os.path.join(ROOT, "tests", "golden", "prepare_fixtures.npz"))
META = json.load(open(os.path.join(ROOT, "tests", "golden", "prepare_fixtures.json")))


def test_system_definition_matches_reference_fixture():
    s = prep.make_system([tuple(r) for r in META["genome"]], META["config"])
    assert np.array_equal(s["particle_types"], FX["types"])
    assert np.allclose(s["ab_factors"], FX["ab"], rtol=0, atol=1e-7)
    assert np.array_equal(s["chromosome_ranges"], FX["chains"][:, :2])
    assert np.array_equal(s["centromere_ranges"], FX["chains"][:, 2:])
    assert np.array_equal(s["nucleolus_ranges"], FX["nucleolus_spans"])
    assert np.array_equal(s["nucleolus_bonds"], FX["nucleolus_bonds"])
    assert s["chromosome_names"] == META["chain_names"] and s["nucleolus_names"] == META["nucleolus_names"]


def test_spline_refinement_matches_reference_fixture():
    for k in range(3):
        fine = prep.refine_path_spline(FX[f"path{k}"], len(FX[f"fine{k}"]))
        assert np.abs(fine - FX[f"fine{k}"]).max() < 1e-9 * max(1.0, np.abs(FX[f"fine{k}"]).max())


def test_refine_positions_layout():
    coarse = np.cumsum(np.random.default_rng(1).normal(size=(9, 3)), axis=0)
    fine = prep.refine_positions(coarse, [(0, 5), (5, 9)], [(0, 48), (48, 85)], 10, [(3, 85), (3, 86)], 87)
    assert np.allclose(fine[:48], prep.refine_path_spline(coarse[:5], 50)[:48])
    assert np.allclose(fine[48:85], prep.refine_path_spline(coarse[5:], 40)[:37])
    assert np.array_equal(fine[85], fine[3]) and np.array_equal(fine[86], fine[3])


def test_seed_derivation():
    a, b = prep.derive_seeds(20220101)
    rs = np.random.RandomState(20220101)
    assert (a, b) == (rs.randint(1000000), rs.randint(1000000)) and a != b
