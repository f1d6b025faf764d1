"""Host-side kinetics of the 1 kb model: the product's restatement
(2022a-genome-dynamics_amd/host/gd_1kb_kinetics.hpp) against the reference's own loop simulator and reservoir
sampler -- through golden fixtures generated from the reference sources compiled in place
(tests/golden/make_ref_1kb_fixtures.py), and live against oracle/_ref/libref1kb.so where it exists.
Every loop record after every step and the generator's next draw must be identical."""
import os

import numpy as np
import pytest

import kinetics_util as ku

GOLD = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_1kb_kinetics.npz"))


@pytest.fixture(scope="module")
def product(tmp_path_factory):
    return ku.Probe(ku.build_product_probe(tmp_path_factory.mktemp("probe")))


@pytest.mark.parametrize("name", sorted(ku.LOOP_SCENARIOS))
def test_loop_extrusion_matches_reference_fixture(product, name):
    sc = ku.LOOP_SCENARIOS[name]
    loops, nxt = product.loops(sc)
    assert np.array_equal(loops, GOLD[f"loops_{name}"])
    assert nxt == int(GOLD[f"loops_{name}_next"])
    # the scenario exercises something: loops were loaded, and (unless static) moved
    assert (loops[..., 2] > 0).any()
    free = loops[..., 2] == 0
    assert (loops[free][:, :2] == sc["length"]).all()            # free slots park at chain_length
    assert (loops[~free][:, 0] <= loops[~free][:, 1]).all()
    if name != "static":
        assert not np.array_equal(loops[0], loops[-1])


def test_loop_scenarios_cover_the_branches():
    g = GOLD
    assert (g["loops_clear"][21:, :, 2] > 0).sum() < (g["loops_clear"][:20, :, 2] > 0).sum()    # cleared, then reloaded
    assert (g["loops_handcuffs_preload"][0, :, 2] > 0).sum() > 4                                 # handcuffs + preloaded
    b = g["loops_boundaries"]
    loaded = b[..., 2] > 0
    for site in (50, 120, 121, 250):                                                            # feet never sit on a boundary
        assert not ((b[..., 0] == site) & loaded).any() and not ((b[..., 1] == site) & loaded).any()
    assert (g["loops_dense_poisson"][-1, :, 2] > 0).all()                                        # slots saturate


@pytest.mark.parametrize("case", range(len(ku.RESERVOIR_CASES)))
def test_reservoir_matches_reference_fixture(product, case):
    cap, n, seed = ku.RESERVOIR_CASES[case]
    items, nxt = product.reservoir(cap, n, seed)
    assert np.array_equal(items, GOLD[f"reservoir_{case}"])
    assert nxt == int(GOLD[f"reservoir_{case}_next"])
    assert len(items) == min(cap, n) and len(set(items.tolist())) == len(items) and (items < max(n, 1)).all()


def test_reservoir_is_uniform(product):
    hits = np.zeros(40)
    for seed in range(600):
        items, _ = product.reservoir(4, 40, 1000 + seed)
        hits[items.astype(int)] += 1
    expect = 600 * 4 / 40
    assert abs(hits - expect).max() < 5 * np.sqrt(expect)


@pytest.mark.skipif(not os.path.exists(ku.REF_LIB), reason="oracle/_ref/libref1kb.so not built (needs /root/reference)")
def test_against_live_reference_build(product):
    ref = ku.Probe(ku.REF_LIB)
    rng = np.random.default_rng(5)
    for k in range(40):
        length = int(rng.integers(20, 400))
        sc = dict(length=length, max_loops=int(rng.integers(1, 20)), loading=float(rng.uniform(0, 30)), unloading=float(rng.uniform(0.05, 2)),
                  forward=float(rng.uniform(0, 80)), backward=float(rng.uniform(0, 20)), seed=int(rng.integers(1, 2**31)), steps=25,
                  dt=float(rng.uniform(0.005, 0.05)), preload=bool(rng.integers(0, 2)),
                  boundaries=rng.choice(length, size=int(rng.integers(0, 6)), replace=False).tolist(),
                  attach=[(int(p), float(rng.uniform(0, 1))) for p in rng.choice(length, size=3, replace=False)],
                  detach=[(int(p), float(rng.uniform(0, 1))) for p in rng.choice(length, size=3, replace=False)],
                  handcuffs=rng.integers(0, length, size=int(rng.integers(0, 4))).tolist())
        if k % 3 == 0:
            sc["crossing"] = float(rng.uniform(0, 20))
        elif k % 3 == 1:
            sc["crossing"] = float("inf")
        a, na = product.loops(sc)
        b, nb = ref.loops(sc)
        assert np.array_equal(a, b) and na == nb, sc
    for k in range(30):
        cap, n, seed = int(rng.integers(1, 50)), int(rng.integers(0, 3000)), int(rng.integers(1, 2**31))
        a, na = product.reservoir(cap, n, seed)
        b, nb = ref.reservoir(cap, n, seed)
        assert np.array_equal(a, b) and na == nb
