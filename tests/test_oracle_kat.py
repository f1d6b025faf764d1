"""Known-answer and consistency tests that pin the CPU oracle (no GPU needed).

The reference has no tests or golden vectors for this path, so the
oracle is pinned by published KAT vectors (Philox: Random123; mt19937_64: ISO C++ [rand.predef]),
analytic two-body values, finite differences and brute-force cross-checks."""
import ctypes as C

import numpy as np
import pytest

from util import CASES, TERMS, build, g


def test_philox_random123_kat(oracle):
    f = oracle.dll.oracle_philox4x32_10
    vecs = [
        ((0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
        ((0xffffffff,) * 4, (0xffffffff,) * 2, (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
        ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0),
         (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1)),
    ]
    for ctr, key, exp in vecs:
        c = (C.c_uint32 * 4)(*ctr); k = (C.c_uint32 * 2)(*key); o = (C.c_uint32 * 4)()
        f(c, k, o)
        assert tuple(o) == exp


def test_mt19937_64_kat(oracle):
    f = oracle.dll.oracle_mt64_nth
    f.restype = C.c_uint64
    f.argtypes = [C.c_uint64, C.c_int]
    assert f(5489, 10000) == 9981545732273789042      # [rand.predef] 10000th value of default mt19937_64


def test_philox_normals_statistics(oracle):
    f = oracle.dll.oracle_philox_normal3
    f.argtypes = [C.c_uint64, C.c_uint32, C.c_int64, C.c_uint32, C.POINTER(C.c_double)]
    z = np.empty((20000, 3))
    buf = (C.c_double * 3)()
    for i in range(len(z)):
        f(12345, i, 7, 0, buf)
        z[i] = buf[:]
    assert abs(z.mean()) < 0.02 and abs(z.var() - 1) < 0.03
    assert abs(np.mean(z ** 4) - 3) < 0.15                      # Gaussian kurtosis
    assert abs(np.corrcoef(z[:, 0], z[:, 1])[0, 1]) < 0.02


def _two_bead(lib, r, setup, box=None):
    s = g.System(lib, 2, 1, box=box)
    setup(s)
    s.set_positions(np.array([[0.0, 0, 0], [r, 0, 0]]))
    return s


@pytest.mark.parametrize("P,Q", [(2, 3), (8, 3), (4, 2), (12, 1), (6, 4)])
def test_softcore_two_bead_analytic(oracle, P, Q):
    eps, sig = 2.5, 0.3
    for u in (0.25, 0.5, 0.999, 1.0, 1.5):
        r = u * sig
        s = _two_bead(oracle, r, lambda s: s.set_pair_softcore(eps, sig, 0.0, 0.0, P, Q, 8, 3, mix=False))
        e_ref = eps * (1 - u ** P) ** Q if u < 1 else 0.0
        f_ref = eps * P * Q / sig * (1 - u ** P) ** (Q - 1) * u ** (P - 1) if u < 1 else 0.0   # -dU/dr
        assert s.energy()[0] == pytest.approx(e_ref, rel=1e-12, abs=1e-15)
        F = s.forces()[0]
        assert F[1, 0] == pytest.approx(f_ref, rel=1e-12, abs=1e-14)     # bead 1 pushed towards +x
        assert F[0, 0] == pytest.approx(-f_ref, rel=1e-12, abs=1e-14)


def test_ab_mixture_weights(oracle):
    # a = (a_i+a_j)/2, b = (b_i+b_j)/2 (simulation_driver_forcefield.cc:32-33,44)
    def setup(s):
        s.set_bead_params(a=np.array([1.0, 0.5]), b=np.array([0.0, 5.0]))
        s.set_pair_softcore(2.0, 0.30, 3.0, 0.24, 2, 3, 8, 3, mix=True)
    r = 0.2
    s = _two_bead(oracle, r, setup)
    ua, ub = r / 0.30, r / 0.24
    e = 0.75 * 2.0 * (1 - ua ** 2) ** 3 + 2.5 * 3.0 * (1 - ub ** 8) ** 3
    assert s.energy()[0] == pytest.approx(e, rel=1e-12)


@pytest.mark.parametrize("kind,K,b,r,e,f", [
    (g.POT_HARMONIC, 70.0, 0.0, 0.3, 0.5 * 70 * 0.09, -70 * 0.3),
    (g.POT_SPRING, 100.0, 1.0, 0.8, 0.5 * 100 * 0.04, +100 * 0.2),
    (g.POT_SPRING, 100.0, 1.0, 1.3, 0.5 * 100 * 0.09, -100 * 0.3),
    (g.POT_SEMISPRING, 500.0, 0.2, 0.15, 0.0, 0.0),          # one-sided: free below the rest length
    (g.POT_SEMISPRING, 500.0, 0.2, 0.25, 0.5 * 500 * 0.0025, -500 * 0.05),
])
def test_bond_potentials_analytic(oracle, kind, K, b, r, e, f):
    s = _two_bead(oracle, r, lambda s: s.add_bond_range(g.System.bond_params(kind, k_a=K, l_a=b), 0, 2, 1))
    assert s.energy()[0] == pytest.approx(e, rel=1e-12, abs=1e-15)
    assert s.forces()[0][1, 0] == pytest.approx(f, rel=1e-12, abs=1e-13)   # radial force on bead 1


def test_bond_scale_and_mixing(oracle):
    # K = (a Ka + b Kb)/s^2, l = (a la + b lb) s   (simulation_driver_forcefield.cc:60-77)
    s = g.System(oracle, 2, 1)
    s.set_bead_params(a=np.array([1.0, 0.0]), b=np.array([0.0, 1.0]))
    s.add_bond_range(g.System.bond_params(g.POT_SEMISPRING, k_a=70, l_a=0.2, k_b=30, l_b=0.1, mix=True, scale_by_bond_scale=True), 0, 2, 1)
    s.set_scaling(1.0, 1.0, 0.5, 1.0)
    s.set_positions(np.array([[0.0, 0, 0], [0.2, 0, 0]]))
    K, l = (0.5 * 70 + 0.5 * 30) / 0.25, (0.5 * 0.2 + 0.5 * 0.1) * 0.5
    assert s.energy()[0] == pytest.approx(0.5 * K * (0.2 - l) ** 2, rel=1e-12)


def test_bending_analytic(oracle):
    # U = e (1 - cos theta): straight 0, right angle e, folded back 2e
    for third, e_ref in (([2.0, 0, 0], 0.0), ([1.0, 1.0, 0], 1.5), ([0.0, 0, 0], 3.0)):
        s = g.System(oracle, 3, 1)
        s.add_bending_range(0, 3, 1.5)
        s.set_positions(np.array([[0.0, 0, 0], [1.0, 0, 0], third]))
        assert s.energy()[0] == pytest.approx(e_ref, abs=1e-12)
    s = g.System(oracle, 3, 1)                                   # per-bead energy of the MIDDLE bead
    s.set_bead_params(bending_energy=np.array([9.0, 2.0, 9.0]))
    s.add_bending_range(0, 3, 0.0, per_bead=True)
    s.set_positions(np.array([[0.0, 0, 0], [1.0, 0, 0], [1.0, 1.0, 0]]))
    assert s.energy()[0] == pytest.approx(2.0, abs=1e-12)


def test_sphere_wall_exact(oracle):
    # a=b=c: the second-order surface construction is exact; inward soft wall with half diameters,
    # outward harmonic, axial reaction sums to the normal force.
    R, eps, sig = 2.0, 2.0, 0.3
    for dist_in in (0.05, 0.1, 0.2):
        s = g.System(oracle, 1, 1)
        s.set_bead_params(a=np.array([1.0]), b=np.array([0.0]))
        s.set_ellipsoid_wall(eps, sig, 0.0, 0.24, 1.0, 0.0, 5000.0, (1e4,) * 3, 1e-4, (R,) * 3, scale_by_bead_scale=False)
        n = np.array([1.0, 2.0, 2.0]) / 3.0
        s.set_positions((n * (R - dist_in))[None])
        u = dist_in / (sig / 2)
        e_ref = eps * (1 - u * u) ** 3 if u < 1 else 0.0      # wall a-weight = (1+1)/2
        f_ref = eps * 6 / (sig / 2) * (1 - u * u) ** 2 * u if u < 1 else 0.0
        assert s.energy()[0] == pytest.approx(e_ref, rel=1e-9, abs=1e-13)
        F = s.forces()[0, 0]
        assert np.allclose(F, -f_ref * n, rtol=1e-9, atol=1e-12)    # pushed inwards
        react = np.array(s.context().axial_reaction)
        assert react.sum() == pytest.approx(f_ref, rel=1e-9, abs=1e-12)
    s = g.System(oracle, 1, 1)
    s.set_ellipsoid_wall(eps, sig, 0.0, 0.24, 1.0, 0.0, 5000.0, (1e4,) * 3, 1e-4, (R,) * 3)
    s.set_positions(np.array([[0.0, 0.0, R + 0.1]]))
    assert s.energy()[0] == pytest.approx(0.5 * 5000 * 0.01, rel=1e-9)
    assert s.forces()[0, 0, 2] == pytest.approx(-5000 * 0.1, rel=1e-9)


def test_softwell_droplet_exact_and_gradient(oracle):
    # nucleolar droplet attraction (simulation_driver_forcefield.cc:153-178): U = -eps / (1 + (r/decay)^6) for r < cutoff,
    # between target beads only (documented choice of the form); F = -grad U; non-targets feel nothing
    eps, dec, cut = 1.5, 0.2, 0.4
    n = np.array([1.0, -2.0, 2.0]) / 3.0
    for r in (0.05, 0.19, 0.2, 0.3, 0.399, 0.401, 0.8):
        s = g.System(oracle, 3, 1)
        s.set_pair_softwell(eps, dec, cut, [0, 1])
        s.set_positions(np.array([[0.0, 0.0, 0.0], r * n, 0.1 * n])[None])       # bead 2 sits between them, not a target
        u6 = (r / dec) ** 6
        e_ref = -eps / (1 + u6) if r < cut else 0.0
        f_ref = eps * 6 * (r / dec) ** 5 / dec / (1 + u6) ** 2 if r < cut else 0.0      # magnitude of the attraction
        assert s.energy()[0] == pytest.approx(e_ref, rel=1e-12, abs=1e-15)
        F = s.forces()[0]
        assert np.allclose(F[0], f_ref * n, rtol=1e-10, atol=1e-13) and np.allclose(F[1], -f_ref * n, rtol=1e-10, atol=1e-13)
        assert np.all(F[2] == 0)
        assert s.energy(g.TERM_ALL & ~g.TERM_PAIR)[0] == 0.0                     # part of the pair term
    rng = np.random.default_rng(4)
    x = rng.random((30, 3)) * 0.8
    tg = rng.choice(30, size=12, replace=False)
    for box in (None, (0.9, 0.9, 0.9)):
        s = g.System(oracle, 30, 1, box=box)
        s.set_pair_softwell(eps, dec, cut, tg)
        s.set_positions(x[None])
        F = s.forces()[0]
        assert np.allclose(F.sum(axis=0), 0, atol=1e-12)
        assert np.all(F[np.setdiff1d(np.arange(30), tg)] == 0)
        h = 1e-6
        for i in tg[:4]:
            for k in range(3):
                xp, xm = x.copy(), x.copy()
                xp[i, k] += h; xm[i, k] -= h
                s.set_positions(xp[None]); ep = s.energy()[0]
                s.set_positions(xm[None]); em = s.energy()[0]
                assert F[i, k] == pytest.approx(-(ep - em) / (2 * h), rel=1e-5, abs=1e-6)
    s.set_pair_softwell(0.0, dec, cut, [])                                        # removable
    assert s.energy()[0] == 0.0
    with pytest.raises(g.GdynError):
        s.set_pair_softwell(eps, dec, cut, [31])


def test_inner_sphere_wall_exact_and_gradient(oracle):
    # excluded core (4-sim-ab/sphere/src/simulation_driver.cc:184-228): outside, the wall-type soft repulsion on the
    # gap to the surface (half diameters, factors (0,1)); inside, a harmonic push back out; F = -grad U throughout
    Rin, eps, sa, sb, K = 0.8, 3.0, 0.30, 0.24, 40.0
    n = np.array([2.0, -1.0, 2.0]) / 3.0
    for a, b in ((1.0, 0.0), (0.0, 1.0), (0.5, 0.5)):
        for gap in (0.02, 0.08, 0.119, 0.2, -0.1, -0.4):
            s = g.System(oracle, 1, 1)
            s.set_bead_params(a=np.array([a]), b=np.array([b]))
            s.set_inner_sphere_wall(Rin, eps, sa, eps, sb, 0.0, 1.0, K)
            s.set_positions((n * (Rin + gap))[None])
            if gap > 0:
                ua, ub = gap / (sa / 2), gap / (sb / 2)
                wa, wb = 0.5 * a, 0.5 * (b + 1.0)
                e_ref = wa * (eps * (1 - ua ** 2) ** 3 if ua < 1 else 0.0) + wb * (eps * (1 - ub ** 8) ** 3 if ub < 1 else 0.0)
                f_ref = wa * (eps * 6 / (sa / 2) * (1 - ua ** 2) ** 2 * ua if ua < 1 else 0.0) \
                    + wb * (eps * 24 / (sb / 2) * (1 - ub ** 8) ** 2 * ub ** 7 if ub < 1 else 0.0)
            else:
                e_ref, f_ref = 0.5 * K * gap * gap, -K * gap
            assert s.energy()[0] == pytest.approx(e_ref, rel=1e-9, abs=1e-13)
            assert np.allclose(s.forces()[0, 0], f_ref * n, rtol=1e-9, atol=1e-12)      # always pushed outwards
    rng = np.random.default_rng(3)
    x = rng.normal(size=(40, 3)) * 0.6
    s = g.System(oracle, 40, 1)
    s.set_bead_params(a=rng.random(40), b=rng.random(40))
    s.set_inner_sphere_wall(Rin, eps, sa, eps, sb, 0.0, 1.0, K)
    s.set_positions(x[None])
    F = s.forces(g.TERM_WALL)[0]
    h = 1e-6
    for i in (0, 7, 23):
        for k in range(3):
            xp, xm = x.copy(), x.copy()
            xp[i, k] += h; xm[i, k] -= h
            s.set_positions(xp[None]); ep = s.energy(g.TERM_WALL)[0]
            s.set_positions(xm[None]); em = s.energy(g.TERM_WALL)[0]
            assert F[i, k] == pytest.approx(-(ep - em) / (2 * h), rel=1e-5, abs=1e-6)
    with pytest.raises(g.GdynError):
        s.set_inner_sphere_wall(0.0, eps, sa, eps, sb, 0.0, 1.0, K)


@pytest.mark.parametrize("name", list(CASES))
def test_force_is_minus_gradient(oracle, name):
    s, *_ = build(oracle, name)
    x0 = s.positions()
    rng = np.random.default_rng(3)
    F = s.forces()
    h = 1e-6
    for i in rng.choice(s.N, 12, replace=False):
        for k in range(3):
            xp = x0.copy(); xp[0, i, k] += h; s.set_positions(xp); ep = s.energy()[0]
            xm = x0.copy(); xm[0, i, k] -= h; s.set_positions(xm); em = s.energy()[0]
            assert F[0, i, k] == pytest.approx(-(ep - em) / (2 * h), rel=2e-5, abs=2e-4)


@pytest.mark.parametrize("name", list(CASES))
def test_internal_forces_sum_to_zero(oracle, name):
    s, *_ = build(oracle, name)
    internal = g.TERM_PAIR | g.TERM_BOND | g.TERM_BEND | g.TERM_DYNAMIC
    F = s.forces(internal)
    assert np.abs(F.sum(axis=1)).max() <= 1e-9 * np.abs(F).sum()


@pytest.mark.parametrize("box", [None, (2.9,) * 3, (0.9, 1.3, 2.9)])
def test_neighbor_search_equals_bruteforce(oracle, box):
    rng = np.random.default_rng(5)
    n = 1500
    x = rng.random((n, 3)) * (np.array(box) if box else 3.0) * (1.3 if box else 1.0)   # periodic: also outside the cell
    x[:50] = np.round(x[:50] / 0.3) * 0.3          # beads exactly on cell boundaries
    s = g.System(oracle, n, 1, box=box)
    s.set_positions(x)
    oracle.dll.oracle_set_bruteforce(s._h, 0)
    cells = {tuple(p) for p in s.search_pairs(0.3)}
    oracle.dll.oracle_set_bruteforce(s._h, 1)
    brute = {tuple(p) for p in s.search_pairs(0.3)}
    assert cells == brute and len(brute) > 100


def test_contact_map_counts_equal_dense_accumulation(oracle):
    """contact_map::update / accumulate / clear (contact_map.cc:26-91): after several updates on different structures the rows
    are the non-zeros of the summed 0/1 matrices of the pairs within the distance, in row-major order; clear() empties one map."""
    rng = np.random.default_rng(8)
    n, R = 400, 2
    s = g.System(oracle, n, R)
    dense = np.zeros((R, n, n), dtype=np.int64)
    assert s.contacts(0).shape == (0, 3)
    for k, dist in enumerate((0.3, 0.3, 0.45)):
        x = rng.random((R, n, 3)) * 2.0
        s.set_positions(x)
        s.contacts_update(dist)
        for r in range(R):
            d = np.linalg.norm(x[r][:, None] - x[r][None], axis=2)
            dense[r] += np.triu(d < dist, 1)
    for r in range(R):
        rows = s.contacts(r)
        i, j = np.nonzero(dense[r])                    # row-major order
        assert np.array_equal(rows[:, 0], i) and np.array_equal(rows[:, 1], j) and np.array_equal(rows[:, 2], dense[r][i, j])
        assert rows[:, 2].max() >= 2
    s.contacts_clear(1)
    assert len(s.contacts(1)) == 0 and len(s.contacts(0)) == np.count_nonzero(dense[0])
    s.contacts_clear()
    assert len(s.contacts(0)) == 0
    with pytest.raises(RuntimeError):
        s.contacts_update(0.0)


def test_verlet_list_trajectory_equals_bruteforce(oracle):
    out = []
    for brute in (1, 0):
        s, dt, kT, flags = build(oracle, "genome")
        oracle.dll.oracle_set_bruteforce(s._h, brute)
        s.begin_phase()
        s.run(60, dt, kT, seed=11, flags=flags)
        out.append(s.positions())
    assert np.abs(out[0] - out[1]).max() < 1e-11


def test_free_diffusion_msd(oracle):
    # <|dx|^2> = 6 mu kT t for free beads (Euler-Maruyama noise scale sqrt(2 mu kT dt))
    n, steps, dt, kT = 4000, 50, 1e-3, 0.7
    s = g.System(oracle, n, 1)
    mu = np.full(n, 1.0); mu[n // 2:] = 2.0
    s.set_bead_params(mobility=mu)
    s.set_positions(np.zeros((n, 3)))
    s.run(steps, dt, kT, seed=5)
    d2 = (s.positions()[0] ** 2).sum(axis=1)
    for sel, m in ((slice(0, n // 2), 1.0), (slice(n // 2, n), 2.0)):
        assert d2[sel].mean() == pytest.approx(6 * m * kT * steps * dt, rel=0.06)


def test_callback_state_sequence(oracle):
    # bead/bond scale follow 1-(1-s0)exp(-t/tau) with t = step*dt; the wall follows its ODE
    s, dt, kT, flags = build(oracle, "genome")
    s.begin_phase()
    R0 = np.array(s.context().semiaxes)
    s.run(1, dt, 0.0, noise=g.NOISE_ZERO, flags=flags)
    c = s.context()
    assert c.step == 1 and c.time == pytest.approx(dt)
    assert c.bead_scale == pytest.approx(1 - 0.2 * np.exp(-dt / 1.0), rel=1e-14)
    react = np.array(c.axial_reaction)
    assert np.allclose(np.array(c.semiaxes), R0 + dt * 1e-4 * (react - 1e4 * R0), rtol=1e-14)


def test_quantisation_matches_store(oracle):
    s = g.System(oracle, 3, 1)
    x = np.array([[0.1234567, -3.7654321, 5.00000763], [1e-6, -1e-6, 0.5], [2.0000076, 7.99999, -7.99999]])
    s.set_positions(x)
    q = s.positions_f32(quantize=True)[0]
    ref = np.rint(x.astype(np.float32) * np.float32(65536)) / np.float32(65536)      # simulation_store.cc:403-407
    assert np.array_equal(q, ref.astype(np.float32))


def test_error_behaviour(oracle):
    s, dt, kT, flags = build(oracle, "ab_box")
    with pytest.raises(g.GdynError) as e:
        s.run(1, dt, kT, spacestep=0.1)
    assert e.value.code == 6                       # GD_EUNSUPPORTED: adaptive step
    with pytest.raises(g.GdynError):
        s.run(1, dt, kT, flags=g.RUN_WALL_DYNAMICS)  # no wall configured
    with pytest.raises(g.GdynError):
        s.add_bond_range(g.System.bond_params(g.POT_HARMONIC, 1.0), 0, s.N + 1)
    with pytest.raises(g.GdynError):
        s.set_pair_softcore(1.0, 1.0, p_a=3)
    with pytest.raises(g.GdynError):
        s.set_positions(np.full((1, s.N, 3), np.nan))
    with pytest.raises(g.GdynError):
        g.System(oracle, 0, 1)


def test_replica_seeds_reproduce_solo_runs(oracle):
    """gd_run_desc.replica_seeds: replica r of a batched handle draws the stream a ONE-replica run with seed
    replica_seeds[r] draws (the reference's ensemble is one process per seed, 5-sim-genome/scripts/run_simulation:8-25),
    so a batched trajectory equals the solo one; without it replica r draws from (seed, r)."""
    from util import build
    seeds = np.array([11, 2 ** 40 + 5, 12345], dtype=np.uint64)
    sb, dt, kT, flags = build(oracle, "genome", n_replicas=3)
    x0 = sb.positions()
    x0[1] += 0.01
    sb.set_positions(x0)
    sb.begin_phase()
    sb.run(6, dt, kT, seed=999, flags=flags, replica_seeds=seeds)
    xb = sb.positions()
    for r in range(3):
        s1, *_ = build(oracle, "genome")
        s1.set_positions(x0[r][None])
        s1.begin_phase()
        s1.run(6, dt, kT, seed=int(seeds[r]), flags=flags)
        assert np.array_equal(s1.positions()[0], xb[r]), r
        assert s1.context().semiaxes[0] == sb.context(r).semiaxes[0]
    sc, *_ = build(oracle, "genome", n_replicas=3)
    sc.set_positions(x0)
    sc.begin_phase()
    sc.run(6, dt, kT, seed=int(seeds[0]), flags=flags)
    assert np.array_equal(sc.positions()[0], xb[0]) and not np.array_equal(sc.positions()[2], xb[2])


def test_deferred_callback_is_the_reference_observation_point(oracle):
    """GD_RUN_DEFER_CALLBACK: after the run the positions are those of step k and the context is what callback(k-1) left --
    what the reference's callback(k) sees when it computes mean_energy and saves the context, before update_bead_scale() /
    update_wall_semiaxes() (simulation_driver_interphase.cc:20-22 vs :42-43).  Applying the pending callback -- explicitly,
    by the next run, or by a force evaluation -- gives the state of an undeferred run bit for bit."""
    from util import build
    sa, dt, kT, flags = build(oracle, "genome")
    sb, *_ = build(oracle, "genome")
    for s in (sa, sb):
        s.begin_phase()
    sa.run(7, dt, kT, seed=3, flags=flags)
    sb.run(6, dt, kT, seed=3, flags=flags)
    c6 = sb.context()
    sb.run(1, dt, kT, seed=3, flags=flags | g.RUN_DEFER_CALLBACK)
    cp = sb.context()
    assert cp.callback_pending == 1 and cp.step == 6
    assert cp.bead_scale == c6.bead_scale and tuple(cp.semiaxes) == tuple(c6.semiaxes)       # the context callback(6) left
    assert np.array_equal(sb.positions(), sa.positions())                                      # ... with the positions of step 7
    e_deferred = sb.energy()[0]
    assert e_deferred != sa.energy()[0]                                                        # (the scales moved in callback(7))
    # the same number from first principles: positions of step 7 under the context of step 6
    sc, *_ = build(oracle, "genome")
    sc.set_positions(sa.positions())
    sc.set_context(0, 6, c6.bead_scale, c6.bond_scale, tuple(c6.semiaxes))
    assert sc.energy()[0] == e_deferred
    sb.apply_callback()
    ca, cb = sa.context(), sb.context()
    assert cb.callback_pending == 0 and cb.step == ca.step == 7 and cb.time == ca.time
    assert cb.bead_scale == ca.bead_scale and cb.bond_scale == ca.bond_scale and tuple(cb.semiaxes) == tuple(ca.semiaxes)
    assert sb.energy()[0] == sa.energy()[0]
    # the next run applies a pending callback itself; so does a force evaluation
    for how in ("run", "forces"):
        s1, *_ = build(oracle, "genome")
        s2, *_ = build(oracle, "genome")
        for s in (s1, s2):
            s.begin_phase()
        s1.run(5, dt, kT, seed=4, flags=flags)
        s2.run(5, dt, kT, seed=4, flags=flags | g.RUN_DEFER_CALLBACK)
        if how == "forces":
            assert np.array_equal(s1.forces(), s2.forces()) and s2.context().callback_pending == 0
        s1.run(5, dt, kT, seed=4, flags=flags)
        s2.run(5, dt, kT, seed=4, flags=flags)
        assert np.array_equal(s1.positions(), s2.positions())
        assert tuple(s1.context().semiaxes) == tuple(s2.context().semiaxes) and s2.context().step == 10


def test_wall_distance_matches_the_reference_geometry_module(oracle):
    """Row a9, the nearest-surface construction of the ellipsoid wall: the oracle restates the author's second-order distance
    (5-sim-genome/src/analyze_lamina/geometry.py:13-28).  Fixtures recorded by IMPORTING that module here
    (tests/golden/make_wall_fixtures.py): distances of 360 points from three ellipsoids.  The oracle's displacement from the
    surface is read back through the ABI -- outside the wall the force is -k x displacement (harmonic, k = 1), inside it the
    soft-core energy eps (1 - r^2/s^2)^3 inverts to r.  The module regularises its quadratic with a + 1e-6 (EPSILON, :4,24);
    the comparison allows exactly that."""
    import os
    from conftest import ROOT
    fx = np.load(os.path.join(ROOT, "tests", "golden", "wall_distance_fixtures.npz"))
    eps, sigma = 2.0, 0.6                               # the wall acts at half the diameter: s = 0.3
    for k in range(3):
        semi, pts, ref = fx[f"semi{k}"], fx[f"points{k}"], fx[f"dist{k}"]
        n = len(pts)
        s = g.System(oracle, n, 1)
        s.set_bead_params(a=np.ones(n), b=np.zeros(n))
        s.set_ellipsoid_wall(eps, sigma, 0.0, sigma, 1.0, 0.0, 1.0, (1.0, 1.0, 1.0), 1e-4, tuple(semi), scale_by_bead_scale=False)
        s.set_positions(pts)
        F = s.forces(g.TERM_WALL)[0]
        inv2 = semi ** -2.0
        a = ((inv2 ** 3)[None, :] * pts * pts).sum(axis=1)          # the `a` of the module's quadratic (its EPSILON acts on it)
        inside = (pts * pts * inv2[None, :]).sum(axis=1) < 1.0
        d_out = np.linalg.norm(F, axis=1)                            # |F| = k |displacement|, k = 1
        assert np.allclose(d_out[~inside] * (a / (a + 1e-6))[~inside], ref[~inside], rtol=1e-9, atol=1e-12)
        # inside: one bead at a time through the energy (U = eps (1 - r^2 / s^2)^3 for r < s)
        hs = 0.5 * sigma
        checked = 0
        for i in np.nonzero(inside & (ref < 0.9 * hs))[0][:25]:
            s1 = g.System(oracle, 1, 1)
            s1.set_bead_params(a=np.ones(1), b=np.zeros(1))
            s1.set_ellipsoid_wall(eps, sigma, 0.0, sigma, 1.0, 0.0, 1.0, (1.0, 1.0, 1.0), 1e-4, tuple(semi), scale_by_bead_scale=False)
            s1.set_positions(pts[i][None, None])
            U = s1.energy(g.TERM_WALL)[0]
            r = hs * np.sqrt(1.0 - (U / eps) ** (1.0 / 3.0))
            assert r * a[i] / (a[i] + 1e-6) == pytest.approx(ref[i], rel=1e-7, abs=1e-10), (k, i)
            checked += 1
        assert checked >= 5 and (~inside).sum() >= 50


def test_compensated_update_arithmetic_emulated():
    """The arithmetic of k_step's compensated position update (gdyn_kernels.hip, integrate section), emulated in numpy float32:
    t = lo + e; x' = x + t; b = x' - x; lo' = (x - (x' - b)) + (t - b) -- Knuth's two-sum, exact in round-to-nearest -- against the
    fp64 sum of the same increments, in the regime of simulation_fine_sampling (|x| of 3 ... 8, increments of 0.1 ... 4 ulp).  The
    plain fp32 update `x += e` on the same increments loses per cent of the displacement and leaves coordinates where they were."""
    rng = np.random.default_rng(1)
    n, steps = 20000, 300
    x0 = (rng.uniform(3.0, 8.0, n) * rng.choice([-1.0, 1.0], n)).astype(np.float32)
    inc = (rng.normal(0.0, 9.0, n) * 1e-7).astype(np.float32)            # mu F dt per step, F ~ 9: constant over the run
    x, lo, xp = x0.copy(), np.zeros(n, np.float32), x0.copy()
    exact = x0.astype(np.float64)
    for _ in range(steps):
        e = inc
        t = lo + e
        xn = x + t
        b = xn - x
        lo = (x - (xn - b)) + (t - b)
        x = xn
        xp = xp + e
        exact = exact + e.astype(np.float64)
    d_exact = exact - x0
    d_comp = (x.astype(np.float64) + lo.astype(np.float64)) - x0
    d_plain = xp.astype(np.float64) - x0
    med = np.median(np.abs(d_exact))
    assert np.abs(d_comp - d_exact).max() <= 1e-6 * med                       # exact to the rounding of lo + e
    assert np.all(np.abs(lo) <= np.spacing(np.abs(x)) * 0.5000001)            # the pair stays normalised
    assert np.median(np.abs(d_plain - d_exact)) > 5e-3 * med                  # the plain update: per cent of the displacement ...
    moving = np.abs(d_exact) > 1e-6
    assert (d_plain[moving] == 0).mean() > 0.01                               # ... and coordinates that never move
