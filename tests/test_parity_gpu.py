"""Parity of the HIP path (through the C-ABI of libgdyn) against the CPU oracle, the committed golden
vectors, and size-independent properties at the BASELINE.json sizes.  Tolerances are the fp32
tolerances stated in tests/util.py / DESIGN.md."""
import os

import numpy as np
import pytest

from conftest import ROOT
from util import (CASES, ENERGY_RTOL, FORCE_RTOL, POS_ATOL_1STEP, POS_ATOL_20STEP, TERMS, build, g, wl)

pytestmark = pytest.mark.gpu
GOLD = np.load(os.path.join(ROOT, "tests", "golden", "golden_small.npz"))
SEED = 20220101
PATHS = {"generic": 1, "tiled": 2}


def _paths(name):
    # both kernel paths for every configuration (periodic boxes are tiled by whole rows of cells)
    return ["generic", "tiled"]


def _assert_path(s, path):
    """The list in use was built for the requested kernel path (a forced tiled path silently falls back to the
    generic one when a tile does not fit: the test must know which one it exercised)."""
    assert s.context().list_path == PATHS[path], (s.context().list_path, path)


def _cases_paths():
    return [(n, p) for n in CASES for p in _paths(n)]


@pytest.mark.parametrize("name,path", _cases_paths())
def test_forces_and_energies_vs_golden(hip, name, path):
    s, *_ = build(hip, name)
    s.set_tuning(kernel_path=PATHS[path])
    scale = np.abs(GOLD[f"{name}/F_all"]).max()
    escale = sum(abs(GOLD[f"{name}/E_{t}"][0]) for t in TERMS if t != "all")
    for t, m in TERMS.items():
        F = s.forces(m)
        assert np.abs(F - GOLD[f"{name}/F_{t}"]).max() <= FORCE_RTOL * scale, t
        assert abs(s.energy(m)[0] - GOLD[f"{name}/E_{t}"][0]) <= ENERGY_RTOL * escale, t
    _assert_path(s, path)


@pytest.mark.parametrize("name,path", _cases_paths())
def test_trajectories_vs_golden(hip, name, path):
    _, _, dt, kT, flags = CASES[name]
    for tag, steps, noise, temp, tol in (("philox1", 1, g.NOISE_PHILOX, kT, POS_ATOL_1STEP),
                                         ("philox10", 10, g.NOISE_PHILOX, kT, POS_ATOL_20STEP),
                                         ("zero20", 20, g.NOISE_ZERO, 0.0, POS_ATOL_20STEP)):
        s, *_ = build(hip, name)
        s.set_tuning(kernel_path=PATHS[path])
        s.begin_phase()
        s.run(steps, dt, temp, seed=SEED, noise=noise, flags=flags)
        _assert_path(s, path)
        scale = max(1.0, np.abs(GOLD[f"{name}/x0"]).max() / 8)          # fp32 ulp grows with |x| (1 kb box ~ 40 units)
        assert np.abs(s.positions() - GOLD[f"{name}/x_{tag}"]).max() <= tol * scale, tag
        c = s.context()
        ref = GOLD[f"{name}/ctx_{tag}"]
        assert c.step == int(ref[0]) and c.time == pytest.approx(ref[1], rel=1e-12)
        assert c.bead_scale == pytest.approx(ref[2], rel=1e-12) and c.bond_scale == pytest.approx(ref[3], rel=1e-12)
        if name == "genome":
            assert np.allclose(np.array(c.semiaxes), ref[4:7], rtol=0, atol=1e-8)
            assert np.allclose(np.array(c.axial_reaction), ref[7:10], rtol=2e-4, atol=1e-2)


@pytest.mark.parametrize("name", list(CASES))
def test_injected_noise_trajectory(hip, name):
    _, _, dt, kT, flags = CASES[name]
    s, *_ = build(hip, name)
    z = np.random.default_rng(SEED).normal(size=(5, 1, s.N, 3))
    s.begin_phase()
    s.run(5, dt, kT, noise=g.NOISE_HOST, host_noise=z, flags=flags)
    scale = max(1.0, np.abs(GOLD[f"{name}/x0"]).max() / 8)
    assert np.abs(s.positions() - GOLD[f"{name}/x_host5"]).max() <= POS_ATOL_20STEP * scale


@pytest.mark.parametrize("name,nrep", [("genome", 3), ("chromatin_1kb", 3), ("genome", 8)])
def test_replica_batch_matches_oracle(hip, oracle, name, nrep):
    """R replicas in one launch: each replica is its own trajectory with its own Philox stream and context.
    (3 replicas: every XCD runs a slice of each replica; 8: whole replicas per XCD -- the block maps of block_map().)"""
    _, _, dt, kT, flags = CASES[name]
    sh, *_ = build(hip, name, n_replicas=nrep)
    so, *_ = build(oracle, name, n_replicas=nrep)
    x0 = so.positions()
    x0[1] += 0.01 * np.random.default_rng(1).normal(size=x0[1].shape)     # make the replicas differ
    for s in (sh, so):
        s.set_positions(x0)
        s.begin_phase()
    scale = np.abs(so.forces()).max()
    assert np.abs(sh.forces() - so.forces()).max() <= FORCE_RTOL * scale
    for s in (sh, so):
        s.run(6, dt, kT, seed=SEED, flags=flags)
    xh, xo = sh.positions(), so.positions()
    assert np.abs(xh - xo).max() <= POS_ATOL_20STEP * max(1.0, np.abs(xo).max() / 8)
    assert np.abs(xo[0] - xo[2]).max() > 1e-4            # same start, different replica index => different noise
    for r in range(nrep):
        assert np.allclose(np.array(sh.context(r).semiaxes), np.array(so.context(r).semiaxes), atol=1e-8)


@pytest.mark.parametrize("path", ["generic", "tiled"])
def test_replica_seeds_match_oracle(hip, oracle, path):
    """gd_run_desc.replica_seeds on the device: replica r draws the stream (replica_seeds[r], 0), as the oracle does (which the
    CPU suite checks against one-replica runs bit for bit)."""
    _, _, dt, kT, flags = CASES["genome"]
    seeds = np.array([11, 2 ** 40 + 5, 12345], dtype=np.uint64)
    out = []
    for lib in (hip, oracle):
        s, *_ = build(lib, "genome", n_replicas=3)
        if lib is hip:
            s.set_tuning(kernel_path=PATHS[path])
        s.begin_phase()
        s.run(6, dt, kT, seed=999, flags=flags, replica_seeds=seeds)
        out.append(s.positions())
    assert np.abs(out[0] - out[1]).max() <= POS_ATOL_20STEP
    assert np.abs(out[1][0] - out[1][1]).max() > 1e-4


@pytest.mark.parametrize("box", [None, (2.9,) * 3])
def test_contact_map_equals_oracle(hip, oracle, box):
    """gd_contacts_*: the device tables (insert, growth by rehash, dump through the radix sort, clear) against the oracle's sorted
    merge over several updates of several replicas; rows identical except for pairs within fp32 rounding of the distance."""
    rng = np.random.default_rng(12)
    n, R = 1500, 3
    sh, so = g.System(hip, n, R, box=box), g.System(oracle, n, R, box=box)
    near = {}                                            # pairs that may legitimately differ, per replica
    for k, dist in enumerate((0.3, 0.3, 0.22, 0.36, 0.3)):
        x = (rng.random((R, n, 3)) * (np.array(box) if box else 3.0) * (0.6 if k == 2 else 1.0)).astype(np.float32).astype(np.float64)
        if k == 1:
            x[1] = xprev[1]                              # replica 1 stands still once: its counts reach 2 everywhere
        xprev = x
        for s in (sh, so):
            s.set_positions(x)
            s.contacts_update(dist)
        for r in range(R):
            ph = {tuple(p) for p in sh.search_pairs(dist, replica=r)}
            so1 = g.System(oracle, n, 1, box=box); so1.set_positions(x[r][None])
            near.setdefault(r, set()).update(ph ^ {tuple(p) for p in so1.search_pairs(dist)})
    total = 0
    for r in range(R):
        rh, ro = sh.contacts(r), so.contacts(r)
        kh = {(int(a), int(b)): int(c) for a, b, c in rh}
        ko = {(int(a), int(b)): int(c) for a, b, c in ro}
        assert len(kh) == len(rh) and np.all(rh[:, 0] < rh[:, 1])
        key = rh[:, 0].astype(np.uint64) << np.uint64(32) | rh[:, 1].astype(np.uint64)
        assert np.all(np.diff(key.astype(np.int64)) > 0)                   # row-major order, no duplicates
        for pr in set(kh) | set(ko):
            if kh.get(pr, 0) != ko.get(pr, 0):
                assert pr in near[r] and abs(kh.get(pr, 0) - ko.get(pr, 0)) <= 1
        total += len(rh)
    assert total > 20000 and max(c for c in kh.values()) >= 1
    assert max(int(c) for c in sh.contacts(1)[:, 2]) >= 2
    sh.contacts_clear(0); so.contacts_clear(0)
    assert len(sh.contacts(0)) == 0 and len(so.contacts(0)) == 0 and abs(len(sh.contacts(2)) - len(so.contacts(2))) <= len(near[2])
    sh.contacts_clear()
    assert all(len(sh.contacts(r)) == 0 for r in range(R))
    x = (rng.random((R, n, 3)) * (np.array(box) if box else 3.0)).astype(np.float32).astype(np.float64)
    sh.set_positions(x); sh.contacts_update(0.3)
    assert len(sh.contacts(0)) == len(sh.search_pairs(0.3, replica=0))     # a cleared map counts again


@pytest.mark.parametrize("box", [None, (2.9,) * 3, (0.9, 1.3, 2.9)])
def test_neighbor_search_pair_set(hip, oracle, box):
    """md::neighbor_searcher::search: the pair SET is exact (beads on cell/box boundaries, aliasing small grids)."""
    rng = np.random.default_rng(5)
    n = 1500
    x = rng.random((n, 3)) * (np.array(box) if box else 3.0) * (1.3 if box else 1.0)
    x[:50] = np.round(x[:50] / 0.3) * 0.3
    x = x.astype(np.float32).astype(np.float64)            # identical fp32-representable inputs on both sides
    sh, so = g.System(hip, n, 1, box=box), g.System(oracle, n, 1, box=box)
    for s in (sh, so):
        s.set_positions(x)
    ph = {tuple(p) for p in sh.search_pairs(0.3)}
    po = {tuple(p) for p in so.search_pairs(0.3)}
    diff = ph ^ po
    # a pair may differ only if its distance is within fp32 rounding of the cutoff
    for i, j in diff:
        d = x[i] - x[j]
        if box:
            d -= np.array(box) * np.rint(d / np.array(box))
        assert abs(np.linalg.norm(d) - 0.3) < 1e-6
    assert len(ph) > 100


def test_lists_on_the_box_of_the_build_before(hip, oracle):
    """Warm list builds lay their cell grid on the bounding box the build BEFORE recorded (gdyn_kernels.hip: k_scatter / k_tiles /
    grid_warm): a free gas that doubles its extent during a run leaves that box at every build -- beads beyond it are clamped
    into the boundary cells.  After the run the resident lists (built on such boxes, tiled path) give the oracle's forces and the
    oracle's pair set on the same positions, for each of three replicas that expand at different rates; so does the first
    build after gd_set_positions has moved the beads somewhere else entirely (a build with its own bounding-box pass)."""
    R, n = 3, 6000
    rng = np.random.default_rng(17)
    x0 = ((rng.random((R, n, 3)) - 0.5) * np.array([2.0, 2.4, 2.8])).astype(np.float32).astype(np.float64)
    def make(lib, r_count):
        s = g.System(lib, n, r_count)
        s.set_bead_params(a=(np.arange(n) % 2).astype(float), b=((np.arange(n) + 1) % 2).astype(float),
                          mobility=np.ones(n))
        s.set_pair_softcore(2.0, 0.3, 2.0, 0.24)
        return s
    sh = make(hip, R)
    sh.set_tuning(kernel_path=2, rebuild_interval=3, adapt_interval=0)
    sh.set_positions(x0)
    sh.begin_phase()
    b0 = sh.context().rebuilds
    sh.run(240, 5e-5, 2.0, seed=SEED, replica_seeds=[11, 12, 13])       # sigma = 0.014 per step and axis: the cloud grows by ~0.5 per side
    assert sh.context().rebuilds - b0 >= 60 and sh.context().list_path == 2
    so = make(oracle, 1)
    def check(xs):
        Fh = sh.forces()
        for r in range(R):
            so.set_positions(xs[r])
            Fo = so.forces()
            assert np.abs(Fh[r] - Fo).max() <= FORCE_RTOL * max(np.abs(Fo).max(), 1.0)
            ph = {tuple(p) for p in sh.search_pairs(0.3, replica=r)}
            po = {tuple(p) for p in so.search_pairs(0.3)}
            for i, j in ph ^ po:
                assert abs(np.linalg.norm(xs[r][i] - xs[r][j]) - 0.3) < 1e-6
            assert len(po) > 1000
    x1 = sh.positions()
    assert np.abs(x1).max() > np.abs(x0).max() + 0.1          # (it did leave the box it started in)
    rb = sh.context().rebuilds
    check(x1)
    assert sh.context().rebuilds <= rb + 1
    x2 = (x0[::-1] * 0.7 + np.array([5.0, -3.0, 2.0])).astype(np.float32).astype(np.float64)      # elsewhere, other extents: nothing of the old box applies
    sh.set_positions(x2)
    check(x2)


def test_contact_pairs_from_the_resident_list(hip, oracle):
    """Contact-map / glue search (simulation_interphase/contact_map.cc:31-91, glues/glue_simulator.cpp:41,67-77) served
    from the Verlet list that is already on the device: S-genome-30k x 4 replicas, a few steps after a list build, search at
    contactmap_distance 0.4 (> force cutoff 0.3, < list radius 0.525).  The pair SET equals the oracle's search over the
    same positions, for every replica; no list build is spent while the beads have moved little, at most one when they
    have moved far -- and the force list stays valid (the next run does not rebuild at its first step)."""
    R = 4
    s, info = wl.genome_interphase(hip, n_beads=30000, n_replicas=R)
    so = g.System(oracle, 30000, 1)
    dt, kT = info["timestep"], info["temperature"]
    flags = g.RUN_UPDATE_SCALES | g.RUN_WALL_DYNAMICS
    s.set_tuning(rebuild_interval=12, adapt_interval=0)
    s.begin_phase()
    for steps, may_rebuild in ((2, 0), (9, 1)):
        s.run(steps, dt, kT, seed=SEED, flags=flags)
        rb0 = s.context().rebuilds
        x = s.positions()
        for r in range(R):
            ph = {tuple(p) for p in s.search_pairs(0.4, replica=r)}
            so.set_positions(x[r][None])
            po = {tuple(p) for p in so.search_pairs(0.4)}
            assert len(po) > 150000
            for i, j in ph ^ po:       # only pairs within fp32 rounding of the search distance may differ
                assert abs(np.linalg.norm(x[r][i] - x[r][j]) - 0.4) < 2e-6
        c = s.context()
        assert c.rebuilds - rb0 <= may_rebuild, (steps, rb0, c.rebuilds)
        assert c.list_path == 2
    # the list left behind is a valid force list: forces agree with the oracle, and running on does not rebuild at once
    rb1 = s.context().rebuilds
    s.run(1, dt, kT, seed=SEED, flags=flags)
    assert s.context().rebuilds == rb1
    so2, _ = wl.genome_interphase(oracle, n_beads=30000, n_replicas=1)
    so2.set_positions(s.positions()[0][None])
    c0 = s.context(0)
    so2.set_context(0, c0.step, c0.bead_scale, c0.bond_scale, list(c0.semiaxes))
    Fo = so2.forces()
    assert np.abs(s.forces()[0] - Fo[0]).max() <= FORCE_RTOL * np.abs(Fo).max()


def test_quantised_snapshot(hip):
    s = g.System(hip, 3, 1)
    x = np.array([[0.1234567, -3.7654321, 5.00000763], [1e-6, -1e-6, 0.5], [2.0000076, 7.99999, -7.99999]])
    s.set_positions(x)
    ref = np.rint(x.astype(np.float32) * np.float32(65536)) / np.float32(65536)
    assert np.array_equal(s.positions_f32(quantize=True)[0], ref.astype(np.float32))
    assert np.array_equal(s.positions_f32()[0], x.astype(np.float32))


def test_edge_cases(hip, oracle):
    # a single bead, beads on top of each other, an empty bond range, a bead at the wall centre
    for lib in (hip, oracle):
        s = g.System(lib, 1, 1)
        s.set_pair_softcore(2.0, 0.3, 2.0, 0.24)
        s.set_ellipsoid_wall(2.0, 0.3, 2.0, 0.24, 5.0, 5.0, 5000.0, (1e4,) * 3, 1e-4, (2.0,) * 3)
        s.add_bond_range(g.System.bond_params(g.POT_HARMONIC, 1.0), 0, 1)     # no bond in a 1-bead range
        s.set_positions(np.zeros((1, 1, 3)))
        assert np.all(s.forces() == 0) and s.energy()[0] == 0
        s.run(3, 1e-5, 1.0, seed=1)
        assert np.isfinite(s.positions()).all()
    out = []
    for lib in (hip, oracle):
        s = g.System(lib, 4, 1)
        s.set_pair_softcore(2.0, 0.3, 2.0, 0.24, mix=False)
        s.add_bond_range(g.System.bond_params(g.POT_SPRING, 10.0, 0.2), 0, 4)
        s.set_positions(np.array([[0.0, 0, 0], [0.0, 0, 0], [0.1, 0, 0], [5.0, 5, 5]]))   # coincident beads: r = 0
        out.append((s.forces(), s.energy()))
    assert np.allclose(out[0][0], out[1][0], atol=1e-4) and np.allclose(out[0][1], out[1][1], rtol=1e-6)


def test_errors_match_oracle(hip):
    s, dt, kT, flags = build(hip, "ab_box")
    with pytest.raises(g.GdynError) as e:
        s.run(1, dt, kT, spacestep=0.1)
    assert e.value.code == 6
    with pytest.raises(g.GdynError):
        s.run(1, dt, kT, flags=g.RUN_WALL_DYNAMICS)
    with pytest.raises(g.GdynError):
        s.run(1, dt, kT, noise=g.NOISE_MT19937)          # the reference's RNG class exists only in the oracle
    with pytest.raises(g.GdynError):
        s.add_bond_range(g.System.bond_params(g.POT_HARMONIC, 1.0), 0, s.N + 1)
    with pytest.raises(g.GdynError):
        s.set_positions(np.full((1, s.N, 3), np.inf))


def test_rollback_is_transparent(hip):
    """A deliberately too-long rebuild interval violates the Verlet skin; the chunk is rolled back and re-run,
    and the trajectory still matches the safe one to rounding."""
    out = []
    for interval in (1, 60):
        s, dt, kT, flags = build(hip, "genome")
        s.set_tuning(rebuild_interval=interval, adapt_interval=0, list_width=128)   # wide list: only skin violations can roll back
        s.begin_phase()
        s.run(60, dt, kT, seed=SEED, flags=flags)
        out.append((s.positions(), s.context().rollbacks))
    assert out[1][1] >= 1 and out[0][1] == 0
    assert np.abs(out[0][0] - out[1][0]).max() <= 1e-4


def test_short_runs_do_not_inflate_the_rebuild_interval(hip):
    """The callers' cadence (simulation_driver_interphase.cc:12-44: chunks that end at the next logging / sampling
    step; bench.py --warmup 5 --steps 20): runs shorter than one rebuild interval measure the displacement of a
    PARTIAL interval and must not be used to lengthen the interval -- the next long run would violate the skin
    and be rolled back."""
    s, info = wl.genome_interphase(hip, n_beads=30000, n_replicas=8)
    s.set_tuning(skin=0.75)                              # (a fixed width: the selection by tile class would move the interval with it)
    dt, kT = info["timestep"], info["temperature"]
    flags = g.RUN_UPDATE_SCALES | g.RUN_WALL_DYNAMICS
    s.begin_phase()
    s.run(400, dt, kT, seed=SEED, flags=flags)           # lets the interval adapt on complete intervals
    c0 = s.context()
    k_verified = c0.rebuild_interval
    assert k_verified >= 4
    s.begin_phase()                                      # invalidates the list, like the bench's begin_phase
    s.run(5, dt, kT, seed=SEED, flags=flags)
    assert s.context().rebuild_interval == k_verified    # 5 < K steps: no complete interval, no adaptation
    s.run(20, dt, kT, seed=SEED, flags=flags)
    for _ in range(10):
        s.run(1, dt, kT, seed=SEED, flags=flags)
    c1 = s.context()
    assert c1.rollbacks == c0.rollbacks, (c0.rollbacks, c1.rollbacks)
    assert c1.rebuild_interval <= int(1.3 * k_verified) + 1, (k_verified, c1.rebuild_interval)


def test_device_philox_normals_kat(hip, oracle):
    """The device's Brownian noise against the oracle's Philox4x32-10 + Box-Muller, variate by variate: free beads on
    a lattice wider than the cutoff (no force acts), mu = 1, dt = 1, kT = 1/2, so one step moves bead i of replica r
    by exactly the three normals of counter (i, step, r).  8 steps x 32 replicas x 4096 beads = 1.05 M triples, steps
    on both sides of 2^32 (the counter's high word).  The tiled kernel is the one under test (the pair term is on)."""
    import ctypes as C
    n_side, R = 16, 32
    N = n_side ** 3
    grid = (np.arange(n_side) - (n_side - 1) / 2) * 0.4
    x0 = np.stack(np.meshgrid(grid, grid, grid, indexing="ij"), axis=-1).reshape(N, 3)
    f = oracle.dll.oracle_philox_normal3
    f.argtypes = [C.c_uint64, C.c_uint32, C.c_int64, C.c_uint32, C.POINTER(C.c_double)]
    s = g.System(hip, N, R)
    s.set_bead_params(a=np.ones(N), b=np.zeros(N))
    s.set_pair_softcore(2.0, 0.3, 2.0, 0.24)
    s.set_tuning(kernel_path=2, rebuild_interval=1, adapt_interval=0)
    worst, zs = 0.0, []
    buf = (C.c_double * 3)()
    rng = np.random.default_rng(3)
    for k, step0 in enumerate([0, 1, 2, 1000, 2 ** 32 - 2, 2 ** 32 - 1, 2 ** 32, 2 ** 40 + 12345]):
        s.set_positions(x0)
        for r in range(R):
            s.set_context(r, step0, 1.0, 1.0)
        s.run(1, 1.0, 0.5, seed=SEED + k, noise=g.NOISE_PHILOX)
        assert s.context().list_path == 2 and s.context(R - 1).step == step0 + 1
        z = s.positions() - x0[None]
        zs.append(z)
        # every variate of 64 random (replica, bead) pairs + the extreme ones, against the oracle's scalar function
        idx = [(int(r), int(i)) for r, i in zip(rng.integers(0, R, 64), rng.integers(0, N, 64))]
        flat = np.abs(z).reshape(R, N, 3).max(axis=2)
        idx += [tuple(int(v) for v in np.unravel_index(np.argmax(flat), flat.shape)), tuple(int(v) for v in np.unravel_index(np.argmin(flat), flat.shape))]
        for r, i in idx:
            f(SEED + k, i, step0 + 1, r, buf)
            worst = max(worst, float(np.abs(z[r, i] - np.array(buf[:])).max()))
    # fp32 position arithmetic (ulp(4) = 4.8e-7 at the lattice edge) + the hardware log / sin / cos of the Box-Muller
    assert worst <= 3e-6, worst
    z = np.concatenate([v.reshape(-1, 3) for v in zs])
    assert len(z) >= 1_000_000
    assert abs(z.mean()) < 3e-3 and abs(z.var() - 1) < 3e-3 and abs(np.mean(z ** 4) - 3) < 0.02
    assert abs(np.corrcoef(z[:, 0], z[:, 1])[0, 1]) < 3e-3 and abs(np.corrcoef(z[:-1, 2], z[1:, 0])[0, 1]) < 3e-3
    # the whole array against the oracle's own stepper for one of the steps (all 131 072 triples of that step)
    so = g.System(oracle, N, R)
    so.set_positions(x0)
    for r in range(R):
        so.set_context(r, 2 ** 32 - 1, 1.0, 1.0)
    so.run(1, 1.0, 0.5, seed=SEED + 5, noise=g.NOISE_PHILOX)
    assert np.abs((so.positions() - x0[None]) - zs[5]).max() <= 3e-6


@pytest.mark.parametrize("n_beads,n_replicas", [(30000, 2), (62178, 1)])
def test_full_size_genome_vs_oracle(hip, oracle, n_beads, n_replicas):
    """The production sizes against the oracle directly (one fp64 force / energy evaluation each, plus a 5-step
    noisy trajectory at 30 000 beads): S-genome-30k and the 62 178-bead production model, tiled path."""
    sh, info = wl.genome_interphase(hip, n_beads=n_beads, n_replicas=n_replicas, bead_scale_init=0.8)
    so, _ = wl.genome_interphase(oracle, n_beads=n_beads, n_replicas=n_replicas, bead_scale_init=0.8)
    Fo = so.forces()
    scale = np.abs(Fo).max()
    for t in ("all", "pair", "bond", "wall"):
        Fh, Fr = sh.forces(TERMS[t]), (Fo if t == "all" else so.forces(TERMS[t]))
        assert np.abs(Fh - Fr).max() <= FORCE_RTOL * scale, t
    assert sh.context().list_path == 2
    eo = so.energy()
    assert np.all(np.abs(sh.energy() - eo) <= 3 * ENERGY_RTOL * np.abs(eo))
    if n_beads == 30000:
        flags = g.RUN_UPDATE_SCALES | g.RUN_WALL_DYNAMICS
        for s in (sh, so):
            s.begin_phase()
            s.run(5, info["timestep"], info["temperature"], seed=SEED, flags=flags)
        assert np.abs(sh.positions() - so.positions()).max() <= POS_ATOL_20STEP
        for r in range(n_replicas):
            assert np.allclose(np.array(sh.context(r).semiaxes), np.array(so.context(r).semiaxes), rtol=0, atol=1e-8)
            assert np.allclose(np.array(sh.context(r).axial_reaction), np.array(so.context(r).axial_reaction), rtol=2e-4, atol=1e-2)


def test_full_size_1kb_vs_oracle(hip, oracle):
    """S-1kb-250k (periodic box, repulsion + attraction, springs, bending, loops, glues) against one oracle evaluation."""
    sh, info = wl.chromatin_1kb(hip, n_beads=250000)
    so, _ = wl.chromatin_1kb(oracle, n_beads=250000)
    Fo = so.forces()
    scale = np.abs(Fo).max()
    assert np.abs(sh.forces() - Fo).max() <= FORCE_RTOL * scale
    for t in ("pair", "bond", "bend", "dynamic"):
        assert np.abs(sh.forces(TERMS[t]) - so.forces(TERMS[t])).max() <= FORCE_RTOL * scale, t
    eo = so.energy()
    esum = sum(abs(so.energy(TERMS[t])[0]) for t in ("pair", "bond", "bend", "dynamic"))
    assert abs(sh.energy()[0] - eo[0]) <= 3 * ENERGY_RTOL * esum
    # the LDS-tiled path of periodic boxes (whole rows of cells) at the list radius it fits at: 53 x 53 = 2 809 rows of cells
    sh.set_tuning(skin=0.35, kernel_path=2)
    assert np.abs(sh.forces() - Fo).max() <= FORCE_RTOL * scale
    assert sh.context().list_path == 2
    assert abs(sh.energy()[0] - eo[0]) <= 3 * ENERGY_RTOL * esum
    ph, po = {tuple(p) for p in sh.search_pairs(1.2)}, {tuple(p) for p in so.search_pairs(1.2)}
    x, L = so.positions()[0], float(info["box"])
    assert len(po) > 200000 and len(ph ^ po) < 100
    for i, j in ph ^ po:       # the device holds the positions in fp32 (ulp 7.6e-6 at |x| ~ 100): only pairs that close to the radius
        d = x[i] - x[j]
        d -= L * np.rint(d / L)
        assert abs(np.linalg.norm(d) - 1.2) < 3e-5


def test_wall_context_on_a_grid_larger_than_the_chip(hip, oracle):
    """30 000 beads x 16 replicas = 944 blocks, more than are resident at once (256 CUs x 3): the per-step callback
    state (semiaxes, axial reaction: reduced from per-block partials by every block's prologue) must not depend on
    which blocks of the previous step have already been replaced -- the partials are double-buffered."""
    R = 16
    sh, info = wl.genome_interphase(hip, n_beads=30000, n_replicas=R, bead_scale_init=0.9)
    so, _ = wl.genome_interphase(oracle, n_beads=30000, n_replicas=R, bead_scale_init=0.9)
    flags = g.RUN_UPDATE_SCALES | g.RUN_WALL_DYNAMICS
    for s in (sh, so):
        s.begin_phase()
        s.run(12, info["timestep"], info["temperature"], seed=SEED, flags=flags)
    for r in range(R):
        ch, co = sh.context(r), so.context(r)
        assert ch.step == co.step == 12
        assert np.allclose(np.array(ch.semiaxes), np.array(co.semiaxes), rtol=0, atol=2e-9), r
        assert np.allclose(np.array(ch.axial_reaction), np.array(co.axial_reaction), rtol=3e-4, atol=2e-2), r
    assert np.abs(sh.positions() - so.positions()).max() <= POS_ATOL_20STEP


def test_many_tiles_trajectory_matches_oracle(hip, oracle):
    """16 x 30 000 beads = 944 tiles (more blocks than the chip holds at once): 12 steps with rebuilds, wall dynamics and scale
    updates on the tiled path against the oracle."""
    R = 16
    sh, info = wl.genome_interphase(hip, n_beads=30000, n_replicas=R, bead_scale_init=0.9)
    so, _ = wl.genome_interphase(oracle, n_beads=30000, n_replicas=R, bead_scale_init=0.9)
    sh.set_tuning(kernel_path=2)
    flags = g.RUN_UPDATE_SCALES | g.RUN_WALL_DYNAMICS
    for s in (sh, so):
        s.begin_phase()
        s.run(12, info["timestep"], info["temperature"], seed=SEED, flags=flags)
    assert sh.context().list_path == 2
    assert np.abs(sh.positions() - so.positions()).max() <= POS_ATOL_20STEP
    for r in range(R):
        assert np.allclose(np.array(sh.context(r).semiaxes), np.array(so.context(r).semiaxes), rtol=0, atol=2e-9), r


def test_callback_scales_host_and_device_paths(hip, oracle):
    """The scales a callback sets are pure functions of the step index: with every replica at the same step the host
    evaluates them (same libm exp as the oracle: identical bits); replicas at different steps make the device evaluate
    them (fp64 exp of the device library: 1e-12).  Both against the oracle, with the trajectories."""
    _, kw, dt, kT, flags = CASES["genome"]
    for different in (False, True):
        R = 3
        sh, _ = wl.genome_interphase(hip, n_replicas=R, **kw)
        so, _ = wl.genome_interphase(oracle, n_replicas=R, **kw)
        for s in (sh, so):
            s.begin_phase()
            if different:
                for r in range(R):
                    c = s.context(r)
                    s.set_context(r, 1000 * r, c.bead_scale, c.bond_scale, tuple(c.semiaxes))
            s.run(12, dt, kT, seed=SEED, flags=flags)
        for r in range(R):
            ch, co = sh.context(r), so.context(r)
            assert ch.step == co.step == (1000 * r if different else 0) + 12
            if different:
                assert abs(ch.bead_scale - co.bead_scale) <= 1e-12 and abs(ch.bond_scale - co.bond_scale) <= 1e-12
            else:
                assert ch.bead_scale == co.bead_scale and ch.bond_scale == co.bond_scale
        assert np.abs(sh.positions() - so.positions()).max() <= POS_ATOL_20STEP


@pytest.mark.parametrize("path", ["generic", "tiled"])
def test_deferred_callback_matches_oracle(hip, oracle, path):
    """GD_RUN_DEFER_CALLBACK on the device (the callback stays pending in the device context; k_ctx or the next launch applies
    it): context, energy at the reference's observation point and the continued trajectory against the oracle, across a chunk
    boundary and a list rebuild."""
    _, kw, dt, kT, flags = CASES["genome"]
    out = []
    for lib in (hip, oracle):
        s, _ = wl.genome_interphase(lib, n_replicas=2, **kw)
        if lib is hip:
            s.set_tuning(kernel_path=PATHS[path])
        s.begin_phase()
        rec = []
        for n in (1, 9, 40):
            s.run(n, dt, kT, seed=SEED, flags=flags | g.RUN_DEFER_CALLBACK)
            c = s.context(1)
            rec.append((c.step, c.callback_pending, c.bead_scale, tuple(c.semiaxes), s.energy().copy()))
            if n == 9:
                s.apply_callback()
                assert s.context(1).callback_pending == 0 and s.context(1).step == c.step + 1
        s.run(3, dt, kT, seed=SEED, flags=flags)
        out.append((rec, s.positions(), s.context(0)))
    (rh, xh, ch), (ro, xo, co) = out
    for (sh_, ph, bh, semh, eh), (so_, po, bo, semo, eo) in zip(rh, ro):
        assert sh_ == so_ and ph == po == 1 and bh == bo
        assert np.allclose(semh, semo, rtol=0, atol=1e-8)
        assert np.all(np.abs(eh - eo) <= 3 * ENERGY_RTOL * np.abs(eo).sum() + 1e-3)
    assert ch.step == co.step == 53 and ch.callback_pending == 0
    assert np.abs(xh - xo).max() <= 1e-4           # 53 noisy steps of a dense system (fp32 vs fp64 divergence)


# ---------------------------------------------------------------- BASELINE sizes

def test_full_size_genome_properties(hip):
    """S-genome-30k x 2 replicas: generic and tiled paths agree, internal forces sum to zero, the pair set is
    the KD-tree pair set, and T=0 dynamics is a descent (energy never increases)."""
    from scipy.spatial import cKDTree
    s, info = wl.genome_interphase(hip, n_beads=30000, n_replicas=2)
    x0 = s.positions()
    internal = g.TERM_PAIR | g.TERM_BOND
    res = {}
    for path in ("generic", "tiled"):
        s.set_tuning(kernel_path=PATHS[path])
        res[path] = (s.forces(), s.forces(internal), s.energy())
    scale = np.abs(res["generic"][0]).max()
    assert np.abs(res["generic"][0] - res["tiled"][0]).max() <= 1e-5 * scale
    assert np.allclose(res["generic"][2], res["tiled"][2], rtol=1e-6)
    Fi = res["tiled"][1]
    assert np.abs(Fi.sum(axis=1)).max() <= 2e-6 * np.abs(Fi).sum(axis=(1, 2)).max()
    pairs = {tuple(p) for p in s.search_pairs(0.3, replica=1)}
    ref = {tuple(sorted(p)) for p in cKDTree(x0[1].astype(np.float32).astype(np.float64)).query_pairs(0.3)}
    assert len(pairs ^ ref) <= 3 and len(ref) > 100000           # only pairs within fp32 rounding of the cutoff may differ
    s.set_tuning(kernel_path=0)
    s.begin_phase()
    e = [s.energy().sum()]
    for _ in range(4):
        s.run(25, 1e-5, 0.0, noise=g.NOISE_ZERO)
        e.append(s.energy().sum())
    assert all(b < a for a, b in zip(e, e[1:]))


def test_full_size_1kb_properties(hip):
    """S-1kb-250k (periodic): internal forces sum to zero, noise has the right variance, descent at T=0."""
    s, info = wl.chromatin_1kb(hip, n_beads=250000)
    res = {}
    for path in ("generic", "tiled"):       # periodic LDS tiles (whole rows of cells, ~4 900 entries here: index-encoded lists)
        s.set_tuning(kernel_path=PATHS[path])
        res[path] = (s.forces(), s.energy())
    assert np.abs(res["generic"][0] - res["tiled"][0]).max() <= 1e-5 * np.abs(res["generic"][0]).max()
    assert np.allclose(res["generic"][1], res["tiled"][1], rtol=1e-6)
    s.set_tuning(kernel_path=0)
    F = s.forces(g.TERM_PAIR | g.TERM_BOND | g.TERM_BEND | g.TERM_DYNAMIC)
    assert np.abs(F.sum(axis=1)).max() <= 2e-6 * np.abs(F).sum()
    x0 = s.positions()
    e0 = s.energy()[0]
    s.run(10, 1e-4, 0.0, noise=g.NOISE_ZERO)
    assert s.energy()[0] < e0
    s.set_positions(x0)
    sf = g.System(hip, 250000, 1)                     # free beads: MSD = 6 mu kT t
    sf.set_positions(np.zeros((250000, 3)))
    sf.run(20, 1e-3, 0.5, seed=3)
    assert (sf.positions()[0] ** 2).sum(axis=1).mean() == pytest.approx(6 * 0.5 * 20 * 1e-3, rel=0.01)


def test_philox_stream_is_counter_based(hip):
    """Noise depends only on (seed, bead, step, replica): running 10 steps equals 4 + 6 steps."""
    a, dt, kT, flags = build(hip, "spindle")
    b, *_ = build(hip, "spindle")
    a.begin_phase(); b.begin_phase()
    a.run(10, dt, kT, seed=9)
    b.run(4, dt, kT, seed=9); b.run(6, dt, kT, seed=9)
    assert np.abs(a.positions() - b.positions()).max() <= 2e-5


def test_dense_cluster_falls_back_to_generic_path(hip, oracle):
    """Every bead inside one neighbourhood: the LDS tile of a block would be the whole system (> LDS budget).
    The device flags it, the handle re-plans on the global-gather path, and results still match the oracle."""
    n = 12000
    rng = np.random.default_rng(2)
    x = rng.normal(size=(n, 3)) * 0.25                    # ~12k beads within a radius of ~0.6: hundreds of neighbours each
    out = []
    for lib in (hip, oracle):
        s = g.System(lib, n, 1)
        s.set_bead_params(a=np.ones(n), b=np.zeros(n))
        s.set_pair_softcore(2.0, 0.30, 2.0, 0.24, mix=True)
        s.set_positions(x)
        out.append(s.forces())
        if lib is hip:
            s.run(3, 1e-6, 0.0, noise=g.NOISE_ZERO)
            assert np.isfinite(s.positions()).all()
    scale = np.abs(out[1]).max()
    assert np.abs(out[0] - out[1]).max() <= 2e-4 * scale        # ~2000 neighbours per bead: fp32 summation error grows


def test_rollback_with_replicas(hip, oracle):
    """Skin violations roll every replica of the chunk back; all replicas still match the oracle afterwards."""
    sh, dt, kT, flags = build(hip, "genome", n_replicas=2)
    so, *_ = build(oracle, "genome", n_replicas=2)
    sh.set_tuning(rebuild_interval=40, adapt_interval=0, list_width=128)
    for s in (sh, so):
        s.begin_phase()
        s.run(40, dt, kT, seed=SEED, flags=flags)
    assert sh.context().rollbacks >= 1
    assert np.abs(sh.positions() - so.positions()).max() <= 1e-4
    for r in range(2):
        assert np.allclose(np.array(sh.context(r).semiaxes), np.array(so.context(r).semiaxes), atol=1e-8)


@pytest.mark.parametrize("powers", [(4, 2, 12, 1), (6, 4, 2, 3), (2, 1, 8, 4)])
def test_runtime_softcore_powers_and_unpackable_factors(hip, oracle, powers):
    """The generic code paths: runtime (P,Q) soft cores, a/b factors that are not fp16-exact (so they are gathered
    from the float2 array and the tiled path is off), non-uniform mobilities, negative (attractive) energy."""
    n = 1200
    rng = np.random.default_rng(11)
    x = rng.random((n, 3)) * 2.2
    a, b = rng.random(n) * 0.9 + 0.05, rng.random(n) * 1.3
    mob = 0.5 + rng.random(n)
    pa, qa, pb, qb = powers
    out = []
    for lib in (hip, oracle):
        s = g.System(lib, n, 2)
        s.set_bead_params(a=a, b=b, mobility=mob)
        s.set_pair_softcore(2.0, 0.30, -0.7, 0.36, pa, qa, pb, qb, mix=True)
        s.add_bond_range(g.System.bond_params(g.POT_SPRING, 40.0, 0.25, k_b=10.0, l_b=0.2, mix=True), 0, n)
        xs = np.stack([x, x + 0.01 * rng.normal(size=x.shape)]) if lib is hip else xs
        s.set_positions(xs)
        f, e = s.forces(), s.energy()
        s.run(5, 2e-5, 0.3, seed=SEED)
        out.append((f, e, s.positions()))
    scale = np.abs(out[1][0]).max()
    assert np.abs(out[0][0] - out[1][0]).max() <= FORCE_RTOL * scale
    assert np.allclose(out[0][1], out[1][1], rtol=1e-5)
    assert np.abs(out[0][2] - out[1][2]).max() <= POS_ATOL_20STEP


def test_inner_sphere_wall_matches_oracle(hip, oracle):
    """The excluded core of the 4-sim-ab sphere model (gd_set_inner_sphere_wall): forces, energy and a short trajectory."""
    rng = np.random.default_rng(11)
    n = 3000
    x = rng.normal(size=(n, 3))
    x *= (rng.random(n) ** (1 / 3) * 1.6 / np.linalg.norm(x, axis=1))[:, None]      # uniform in a ball: some inside the core
    a = (rng.random(n) < 0.5).astype(float)
    systems = []
    for lib in (hip, oracle):
        s = g.System(lib, n, 1)
        s.set_bead_params(a=a, b=1.0 - a, mobility=np.ones(n))
        s.set_pair_softcore(2.0, 0.30, 2.0, 0.24, 2, 3, 8, 3, mix=True)
        s.set_ellipsoid_wall(4.0, 0.30, 4.0, 0.24, 0.0, 1.0, 50.0, (0.0,) * 3, 0.0, (1.7,) * 3, scale_by_bead_scale=False)
        s.set_inner_sphere_wall(0.6, 3.0, 0.30, 3.0, 0.24, 0.0, 1.0, 40.0)
        s.set_positions(x[None])
        s.begin_phase()
        systems.append(s)
    sh, so = systems
    fo, fw = so.forces(), so.forces(g.TERM_WALL)
    assert np.abs(fw).max() > 1.0
    assert np.abs(sh.forces() - fo).max() <= FORCE_RTOL * np.abs(fo).max()
    assert np.abs(sh.forces(g.TERM_WALL) - fw).max() <= FORCE_RTOL * np.abs(fw).max()
    assert sh.energy()[0] == pytest.approx(so.energy()[0], rel=ENERGY_RTOL)
    for s in systems:
        s.run(10, 1e-5, 1.0, seed=SEED)
    assert np.abs(sh.positions() - so.positions()).max() <= POS_ATOL_20STEP


@pytest.mark.parametrize("uniform_mobility", [True, False])
def test_softwell_droplet_matches_oracle(hip, oracle, uniform_mobility):
    """Nucleolar droplet attraction (gd_set_pair_softwell) on top of the interphase force field: forces, energy, trajectory."""
    sh, info = wl.genome_interphase(hip, n_beads=4000)
    so, _ = wl.genome_interphase(oracle, n_beads=4000)
    rng = np.random.default_rng(2)
    tg = np.sort(rng.choice(4000, size=300, replace=False))
    mob = np.ones(4000) if uniform_mobility else rng.uniform(0.5, 1.5, size=4000)
    x = so.positions()
    x[0, tg] = x[0, tg[0]] + 0.25 * rng.normal(size=(300, 3))           # a clump, so that many target pairs are in range
    for s in (sh, so):
        s.set_bead_params(mobility=mob)
        s.set_pair_softwell(0.8, 0.2, 0.4, tg)
        s.set_positions(x)
        s.begin_phase()
    fo = so.forces()
    fsw = fo - _without_softwell_forces(so, tg)
    assert np.abs(fsw).max() > 0.5
    assert np.abs(sh.forces() - fo).max() <= FORCE_RTOL * np.abs(fo).max()
    assert sh.energy()[0] == pytest.approx(so.energy()[0], rel=ENERGY_RTOL * 5)
    for s in (sh, so):
        s.run(10, 1e-5, 1.0, seed=SEED, flags=g.RUN_WALL_DYNAMICS)
    assert np.abs(sh.positions() - so.positions()).max() <= POS_ATOL_20STEP


def _without_softwell_forces(s, tg):
    s.set_pair_softwell(0.0, 0.2, 0.4, [])
    f = s.forces()
    s.set_pair_softwell(0.8, 0.2, 0.4, tg)
    return f


def test_equilibrium_statistics_match_oracle(hip, oracle):
    """Trajectories diverge (chaos), statistics must not: 6000 steps of the 2000-bead interphase model at T = 1 on the device
    (fp32, list cadence, rollbacks) and in the oracle (fp64); time averages over the second half agree within their noise."""
    out = {}
    for name, lib in (("hip", hip), ("oracle", oracle)):
        s, info = wl.genome_interphase(lib, n_beads=2000)
        s.begin_phase()
        s.run(3000, 1e-5, 1.0, seed=SEED + 5, flags=g.RUN_WALL_DYNAMICS)
        e, bond, rg = [], [], []
        for k in range(30):
            s.run(100, 1e-5, 1.0, seed=SEED + 5, flags=g.RUN_WALL_DYNAMICS)
            x = s.positions()[0]
            e.append(float(s.energy()[0]) / 2000)
            d = np.linalg.norm(np.diff(x, axis=0), axis=1)
            bond.append(np.median(d))                       # chain bonds dominate the median (few inter-chain jumps)
            rg.append(np.sqrt(((x - x.mean(axis=0)) ** 2).sum(axis=1).mean()))
        out[name] = (np.mean(e), np.std(e), np.mean(bond), np.mean(rg), np.array(s.context().semiaxes))
    eh, sh_, bh, rh, ah = out["hip"]
    eo, so_, bo, ro, ao = out["oracle"]
    assert abs(eh - eo) <= 5 * max(sh_, so_) / np.sqrt(30) + 0.01 * abs(eo)
    assert bh == pytest.approx(bo, rel=0.02)
    assert rh == pytest.approx(ro, rel=0.01)
    assert np.allclose(ah, ao, rtol=2e-3)                   # the wall ODE integrates the same mean reaction


def test_contact_maps_at_scale_accumulate_like_the_oracle(hip, oracle):
    """gd_contacts_update inside a run at the benchmark's size (8 x 30 000 beads, tiled lists, the reference's contact distance
    0.4 x bead_scale > the force cutoff): three updates 37 steps apart -- served from the resident list or after one build at the
    larger radius -- then the dumped rows of two replicas against the oracle's accumulation over the same three structures."""
    R = 8
    s, info = wl.genome_interphase(hip, n_beads=30000, n_replicas=R)
    flags = g.RUN_UPDATE_SCALES | g.RUN_WALL_DYNAMICS
    s.begin_phase()
    s.run(300, info["timestep"], info["temperature"], seed=SEED, flags=flags)
    so = g.System(oracle, 30000, 2)
    near = [set(), set()]
    for k in range(3):
        s.run(37, info["timestep"], info["temperature"], seed=SEED + 1 + k, flags=flags | g.RUN_DEFER_CALLBACK)
        dist = 0.4 * s.context(0).bead_scale
        s.contacts_update(dist)
        x = s.positions()
        s.apply_callback()
        so.set_positions(x[[1, R - 1]])
        so.contacts_update(dist)
    assert s.context(0).list_path == 2
    for q, r in enumerate((1, R - 1)):
        rh, ro = s.contacts(r), so.contacts(q)
        assert len(ro) > 200000 and abs(len(rh) - len(ro)) < 40
        kh = dict(zip(map(tuple, rh[:, :2].tolist()), rh[:, 2].tolist()))
        ko = dict(zip(map(tuple, ro[:, :2].tolist()), ro[:, 2].tolist()))
        diff = [pr for pr in set(kh) | set(ko) if kh.get(pr, 0) != ko.get(pr, 0)]
        assert len(diff) < 40 and all(abs(kh.get(pr, 0) - ko.get(pr, 0)) == 1 for pr in diff)      # (borderline pairs only)
        key = rh[:, 0].astype(np.int64) << 32 | rh[:, 1].astype(np.int64)
        assert np.all(np.diff(key) > 0) and rh[:, 2].max() == 3


def test_energy_after_a_deferred_run_uses_the_verified_resident_list(hip, oracle):
    """The drivers observe after a GD_RUN_DEFER_CALLBACK run: an accepted chunk has verified the resident list for exactly the
    positions and cutoff gd_compute_energy then sees, so it must not cost a list build -- and the energy must be the one a fresh
    list gives (and the oracle's).  Without the deferral (scales moved behind the last step) the evaluation builds, as before."""
    R = 4
    s, info = wl.genome_interphase(hip, n_beads=30000, n_replicas=R)
    flags = g.RUN_UPDATE_SCALES | g.RUN_WALL_DYNAMICS
    s.begin_phase()
    s.run(300, info["timestep"], info["temperature"], seed=SEED, flags=flags)
    s.run(33, info["timestep"], info["temperature"], seed=SEED + 1, flags=flags | g.RUN_DEFER_CALLBACK)
    b0 = s.context().rebuilds
    e_resident = s.energy()
    assert s.context().rebuilds == b0                     # no build for the observation
    x = s.positions()
    s2, _ = wl.genome_interphase(hip, n_beads=30000, n_replicas=R)      # the same state on a fresh handle: its evaluation builds a list
    s2.set_positions(x)
    s2.begin_phase()
    for r in range(R):
        c = s.context(r)
        s2.set_context(r, c.step, c.bead_scale, c.bond_scale, c.semiaxes)
    e_fresh = s2.energy()
    assert s2.context().rebuilds >= 1
    assert np.abs(e_resident - e_fresh).max() <= 2e-6 * np.abs(e_fresh).max()
    s.apply_callback()
    s.run(20, info["timestep"], info["temperature"], seed=SEED + 2, flags=flags)      # no deferral: the scales moved behind the last step
    b1 = s.context().rebuilds
    s.energy()
    assert s.context().rebuilds == b1 + 1


@pytest.mark.parametrize("n_core,width,path_expected", [(360, 400, 2), (1100, 1200, 2)])
def test_dense_cluster_within_and_beyond_the_tiled_record(hip, oracle, n_core, width, path_expected):
    """A ball of radius 0.1 in which every pair is inside the near radius + a dilute background, tiled path requested.
    360 beads: 359 near entries per bead, more than the 248 a list class held before the record's count fields were widened
    (11 bits of near entries in fours, 6 bits of far chunks) -- stays on the tiled path and matches the oracle.
    1 100 beads: beyond the 1 016 near entries rounds 2-3 capped the class at -- the field holds 8 184: tiled as well.  (A near
    class beyond THAT needs a tile of more than the 8 192 entries the LDS holds: such a state leaves the tiled path through the
    tile overflow, test_dense_cluster_falls_back_to_generic_path; the far class beyond its 504 entries: the next test.)"""
    rng = np.random.default_rng(11)
    n = n_core + 840
    v = rng.normal(size=(n_core, 3))
    core = 0.1 * v / np.linalg.norm(v, axis=1)[:, None] * rng.random((n_core, 1)) ** (1 / 3)
    x = np.concatenate([core, (rng.random((n - n_core, 3)) - 0.5) * 6.0]).astype(np.float32).astype(np.float64)
    out = []
    for lib in (hip, oracle):
        s = g.System(lib, n, 1)
        s.set_bead_params(a=(np.arange(n) % 2).astype(float), b=((np.arange(n) + 1) % 2).astype(float))
        s.set_pair_softcore(2.0, 0.3, 2.0, 0.24)
        if lib is hip:
            s.set_tuning(kernel_path=2, list_width=width)
        s.set_positions(x)
        out.append((s.forces(), s.energy(), {tuple(p) for p in s.search_pairs(0.3)}, s.context().list_path))
    (Fh, Eh, Ph, path), (Fo, Eo, Po, _) = out
    assert np.abs(Fh - Fo).max() <= FORCE_RTOL * np.abs(Fo).max()
    assert abs(Eh[0] - Eo[0]) <= 1e-5 * abs(Eo[0])
    for i, j in Ph ^ Po:
        assert abs(np.linalg.norm(x[i] - x[j]) - 0.3) < 1e-6
    assert len(Po) > n_core * (n_core - 1) // 2 - 10
    assert path == path_expected


def test_far_class_beyond_the_tiled_record_builds_single_class_lists(hip, oracle):
    """Two balls of 600 beads (radius 0.01) 0.49 apart: every bead holds the other ball in its FAR class (near radius 0.446, list
    radius 0.525 at the pinned width) -- 600 entries where the record's far field counts 504.  The build flags it (bit 1 of the
    overflow flag) and the handle builds single-class lists: tiled path kept, forces / energy / pair set as the oracle's; after
    the balls have been pulled apart the next builds return to two classes and still agree."""
    rng = np.random.default_rng(5)
    nb, n = 600, 2 * 600 + 800
    def ball(c):
        v = rng.normal(size=(nb, 3))
        return c + 0.01 * v / np.linalg.norm(v, axis=1)[:, None] * rng.random((nb, 1)) ** (1 / 3)
    x = np.concatenate([ball(np.array([0.0, 0.0, 0.0])), ball(np.array([0.49, 0.0, 0.0])),
                        (rng.random((n - 2 * nb, 3)) - 0.5) * 6.0 + np.array([0.0, 0.0, 5.0])]).astype(np.float32).astype(np.float64)
    def make(lib):
        s = g.System(lib, n, 1)
        s.set_bead_params(a=(np.arange(n) % 2).astype(float), b=((np.arange(n) + 1) % 2).astype(float))
        s.set_pair_softcore(2.0, 0.3, 2.0, 0.24)
        if lib is hip:
            s.set_tuning(kernel_path=2, skin=0.75, near_fraction=0.65)
        return s
    sh, so = make(hip), make(oracle)
    for xs in (x, np.concatenate([x[:nb], x[nb:2 * nb] + np.array([2.0, 0.0, 0.0]), x[2 * nb:]])):
        res = []
        for s in (sh, so):
            s.set_positions(xs)
            res.append((s.forces(), s.energy(), {tuple(p) for p in s.search_pairs(0.3)}))
        (Fh, Eh, Ph), (Fo, Eo, Po) = res
        assert np.abs(Fh - Fo).max() <= FORCE_RTOL * np.abs(Fo).max()
        assert abs(Eh[0] - Eo[0]) <= 1e-5 * abs(Eo[0])
        for i, j in Ph ^ Po:
            assert abs(np.linalg.norm(xs[i] - xs[j]) - 0.3) < 1e-6
        assert len(Po) >= 2 * (nb * (nb - 1) // 2)
        assert sh.context().list_path == 2


# ---------------------------------------------------------------- trajectories at scale, at the benchmark's cadence

POS_ATOL_42STEP = 1e-4      # |dx| after 42 noisy steps, fp32 device vs fp64 oracle (dense soft spheres: errors grow ~ e^(0.1 step))


def _adapted_state(hip, R, steps=350):
    """S-genome-30k x R on the device after `steps` steps: rebuild interval adapted from complete intervals (about 14)."""
    s, info = wl.genome_interphase(hip, n_beads=30000, n_replicas=R)
    flags = g.RUN_UPDATE_SCALES | g.RUN_WALL_DYNAMICS
    s.begin_phase()
    s.run(steps, info["timestep"], info["temperature"], seed=SEED + 1, flags=flags)
    return s, info, flags


def test_adapted_interval_trajectory_matches_oracle(hip, oracle):
    """16 x 30 000 beads, tiled path, at the cadence the benchmark runs at: the interval adapted to ~14 steps, 42 noisy steps
    = three complete intervals, each ending with the steps in which the far list class joins (gdyn_kernels.hip, pair loop),
    against the fp64 oracle started from the same positions and context."""
    R = 16
    sh, info, flags = _adapted_state(hip, R)
    c0 = sh.context()
    K = c0.rebuild_interval
    assert 8 <= K <= 30 and c0.list_path == 2, (K, c0.list_path)
    so, _ = wl.genome_interphase(oracle, n_beads=30000, n_replicas=R)
    so.set_positions(sh.positions())
    for r in range(R):
        c = sh.context(r)
        so.set_context(r, c.step, c.bead_scale, c.bond_scale, tuple(c.semiaxes))
    for s in (sh, so):
        s.run(42, info["timestep"], info["temperature"], seed=SEED, flags=flags)
    c1 = sh.context()
    assert c1.rollbacks == c0.rollbacks and c1.rebuilds - c0.rebuilds >= 42 // (K + 2), (c0.rebuilds, c1.rebuilds, K)
    assert c1.list_path == 2
    assert np.abs(sh.positions() - so.positions()).max() <= POS_ATOL_42STEP
    for r in range(R):
        assert sh.context(r).step == so.context(r).step
        assert np.allclose(np.array(sh.context(r).semiaxes), np.array(so.context(r).semiaxes), rtol=0, atol=1e-8), r


def test_far_class_rule_against_both_classes_every_step(hip):
    """The far list class is skipped while it cannot matter (d0 >= cutoff + D_i + D): the same state stepped over three
    intervals of 11 steps (a) with the product rule and (b) with a near radius ~ cutoff, i.e. every step walks both classes (what a plain
    Verlet list does), and (c) on the generic path (one class).  Skipped entries contribute exactly zero, so the trajectories
    agree to the order of summation -- deterministic (T = 0) and noisy."""
    R = 16
    s0, info, flags = _adapted_state(hip, R)
    x0 = s0.positions()
    ctx = [s0.context(r) for r in range(R)]
    s0.close()
    for noise, kT in ((g.NOISE_ZERO, 0.0), (g.NOISE_PHILOX, info["temperature"])):
        out = []
        for tune in (dict(kernel_path=2), dict(kernel_path=2, near_fraction=1e-3), dict(kernel_path=1)):
            s, _ = wl.genome_interphase(hip, n_beads=30000, n_replicas=R)
            s.set_tuning(rebuild_interval=11, adapt_interval=0, **tune)
            s.set_positions(x0)
            for r in range(R):
                s.set_context(r, ctx[r].step, ctx[r].bead_scale, ctx[r].bond_scale, tuple(ctx[r].semiaxes))
            s.run(33, info["timestep"], kT, seed=SEED, noise=noise, flags=flags)
            c = s.context()
            # (a fresh handle may size its tile / list width with one rolled-back chunk; a skin violation would cut the interval)
            assert c.rollbacks <= 1 and c.list_path == tune["kernel_path"] and c.rebuild_interval == 11, (c.rollbacks, c.list_path, c.rebuild_interval)
            out.append(s.positions())
            s.close()
        assert np.abs(out[0] - out[1]).max() <= POS_ATOL_20STEP, noise          # product rule vs both classes every step
        assert np.abs(out[0] - out[2]).max() <= POS_ATOL_20STEP, noise          # tiled vs generic


@pytest.mark.parametrize("n_beads", [62178, 250000])
def test_full_size_noisy_trajectories(hip, oracle, n_beads):
    """Five noisy steps at the production bead count of the whole-genome model and at the size of the 1 kb chromosome
    (periodic box, loops and glues), device vs oracle."""
    if n_beads == 62178:
        sh, info = wl.genome_interphase(hip, n_beads=n_beads, bead_scale_init=0.8)
        so, _ = wl.genome_interphase(oracle, n_beads=n_beads, bead_scale_init=0.8)
        flags = g.RUN_UPDATE_SCALES | g.RUN_WALL_DYNAMICS
    else:
        sh, info = wl.chromatin_1kb(hip, n_beads=n_beads)
        so, _ = wl.chromatin_1kb(oracle, n_beads=n_beads)
        flags = 0
    for s in (sh, so):
        s.begin_phase()
        s.run(5, info["timestep"], info["temperature"], seed=SEED, flags=flags)
    xh, xo = sh.positions(), so.positions()
    d = xh - xo
    if n_beads == 250000:
        L = float(info["box"])
        d -= L * np.rint(d / L)              # a bead within rounding of the box face may be wrapped on one side only
    assert np.abs(d).max() <= POS_ATOL_20STEP * max(1.0, np.abs(xo).max() / 8)
    assert sh.context().step == so.context().step == 5


def test_step_split_by_tile_class_matches_oracle(hip, oracle):
    """When the largest tile of a build lies just above the three-block LDS class (3 312 < capacity < 4 096, the S-genome-62k x 64
    case) a step is two launches -- the blocks whose tile fits 3 312, then the rest.  Every bead must still be moved exactly once:
    a list width that lands in that class, five noisy steps of two replicas against the oracle."""
    chosen = None
    for skin in (1.1, 1.05, 1.15, 1.0, 1.2):
        sh, info = wl.genome_interphase(hip, n_beads=62178, n_replicas=2, bead_scale_init=0.8)
        sh.set_tuning(skin=skin, adapt_interval=0, rebuild_interval=4)
        sh.begin_phase()
        sh.energy()                                   # builds the list: the tile class of this width
        if 3312 < sh.context().tile_capacity < 4096:
            chosen = skin
            break
        sh.close()
    assert chosen is not None, "no list width in the split class"
    so, _ = wl.genome_interphase(oracle, n_beads=62178, n_replicas=2, bead_scale_init=0.8)
    flags = g.RUN_UPDATE_SCALES | g.RUN_WALL_DYNAMICS
    so.begin_phase()
    for s in (sh, so):
        s.run(5, info["timestep"], info["temperature"], seed=SEED, flags=flags)
    c = sh.context()
    assert 3312 < c.tile_capacity < 4096 and c.list_path == 2 and c.rollbacks == 0
    assert np.abs(sh.positions() - so.positions()).max() <= POS_ATOL_20STEP


def test_step_split_in_a_dense_state_matches_oracle(hip, oracle):
    """The same split one class up: S-genome-30k compressed to half its size (about 200 list entries per bead) builds tiles beyond
    5 072 entries -- one block per CU -- next to tiles that fit two; five noisy steps of two replicas against the oracle."""
    sh, info = wl.genome_interphase(hip, n_beads=30000, n_replicas=2)
    so, _ = wl.genome_interphase(oracle, n_beads=30000, n_replicas=2)
    x = sh.positions() * 0.5
    for s in (sh, so):
        s.set_positions(x)
        s.begin_phase()
    sh.set_tuning(adapt_interval=0, rebuild_interval=3)
    for s in (sh, so):
        s.run(5, info["timestep"], info["temperature"], seed=SEED, flags=0)
    c = sh.context()
    assert c.tile_capacity > 5072 and c.list_path == 2 and c.list_entries / 30000 > 150
    assert np.abs(sh.positions() - so.positions()).max() <= 3 * POS_ATOL_20STEP      # (forces ten times the benchmark state's)


def test_auto_skin_sweep_keeps_results_and_settles(hip, oracle):
    """gd_tuning.auto_skin: the list width is selected from measured chunk times while the run goes on (candidate widths, each
    for a few verified chunks).  Whatever it selects, the lists are verified: after the sweep the forces on the current positions
    equal the oracle's, the list radius is one of the candidates', and the sweep itself rolled back at most a chunk per candidate."""
    R = 8
    s, info = wl.genome_interphase(hip, n_beads=6000, n_replicas=R)
    s.set_tuning(auto_skin=1)
    flags = g.RUN_UPDATE_SCALES | g.RUN_WALL_DYNAMICS
    s.begin_phase()
    s.run(6000, info["timestep"], info["temperature"], seed=SEED, flags=flags)
    c = s.context()
    cut = 0.30
    assert cut * 1.1 < c.list_radius < cut * 2.0, c.list_radius          # a width between 0.1 and 1.0 x the cutoff
    assert c.rollbacks <= 8 and c.step == 6000
    so, _ = wl.genome_interphase(oracle, n_beads=6000, n_replicas=1)
    so.set_positions(s.positions()[3][None])
    c3 = s.context(3)
    so.set_context(0, c3.step, c3.bead_scale, c3.bond_scale, tuple(c3.semiaxes))
    Fo = so.forces()
    assert np.abs(s.forces()[3] - Fo[0]).max() <= FORCE_RTOL * np.abs(Fo).max()


@pytest.mark.parametrize("path", ["generic", "tiled"])
def test_mixed_bond_sets_beyond_the_premixed_table(hip, oracle, path):
    """AB-mixed bond sets (K = a Ka + b Kb, l = a la + b lb, simulation_driver_forcefield.cc:58-88) are resolved per bond on the
    host while the distinct (set, a, b) combinations fit the 16-entry table; with many distinct A/B factors they do not, and the
    kernels mix at run time.  Both against the oracle: 9 distinct factor values (fp16-exact) -> dozens of combinations."""
    rng = np.random.default_rng(21)
    n = 1200
    vals = np.arange(9) / 8.0
    a, b = rng.choice(vals, n), rng.choice(vals, n)
    x = np.cumsum(rng.normal(scale=0.12, size=(n, 3)), axis=0)
    x -= x.mean(axis=0)
    out = []
    for lib in (hip, oracle):
        s = g.System(lib, n, 1)
        s.set_bead_params(a=a, b=b)
        s.set_pair_softcore(2.0, 0.3, 2.0, 0.24)
        s.add_bond_range(g.System.bond_params(g.POT_SEMISPRING, k_a=70.0, k_b=40.0, l_a=0.2, l_b=0.25, mix=True), 0, n, 1)
        s.add_bond_range(g.System.bond_params(g.POT_HARMONIC, k_a=5.0, k_b=3.0, mix=True), 0, n, 2)
        if lib is hip:
            s.set_tuning(kernel_path=PATHS[path])
        s.set_positions(x)
        out.append((s.forces(), s.forces(g.TERM_BOND), s.energy(g.TERM_BOND)[0], s.context().list_path))
        s.run(5, 1e-5, 1.0, seed=SEED)
        out[-1] += (s.positions(),)
    (Fh, Fbh, Ebh, ph, xh), (Fo, Fbo, Ebo, _, xo) = out
    scale = np.abs(Fo).max()
    assert ph == PATHS[path]
    assert np.abs(Fh - Fo).max() <= FORCE_RTOL * scale and np.abs(Fbh - Fbo).max() <= FORCE_RTOL * scale
    assert abs(Ebh - Ebo) <= 3 * ENERGY_RTOL * abs(Ebo)
    assert np.abs(xh - xo).max() <= POS_ATOL_20STEP


# ---------------------------------------------------------------- the benchmark's launch shape

def test_benchmark_launch_shape_matches_oracle(hip, oracle):
    """128 x 30 000 beads in one handle -- the shape bench.py times: 7 552 blocks, whole replicas per XCD (block_map), tiled lists,
    the interval adapted to ~14, 1.4 GB of state.  Forces of replicas 0, 63 and 127 and a 14-step noisy run (one rebuild interval,
    the far list class joining at its end) against the fp64 oracle, each replica on the stream a solo run with its seed draws."""
    R, picks = 128, (0, 63, 127)
    sh, info, flags = _adapted_state(hip, R, steps=600)
    c0 = sh.context()
    assert c0.list_path == 2 and 8 <= c0.rebuild_interval <= 30, (c0.list_path, c0.rebuild_interval)
    x0 = sh.positions()
    ctx = {r: sh.context(r) for r in picks}
    Fh = sh.forces()
    ors = {}
    for r in picks:
        so, _ = wl.genome_interphase(oracle, n_beads=30000, n_replicas=1)
        so.set_positions(x0[r][None])
        so.set_context(0, ctx[r].step, ctx[r].bead_scale, ctx[r].bond_scale, tuple(ctx[r].semiaxes))
        Fo = so.forces()[0]
        assert np.abs(Fh[r] - Fo).max() <= FORCE_RTOL * np.abs(Fo).max(), r
        ors[r] = so
    del Fh
    seeds = SEED + 1000 + np.arange(R, dtype=np.uint64)
    sh.run(14, info["timestep"], info["temperature"], seed=0, flags=flags, replica_seeds=seeds)
    c1 = sh.context()
    assert c1.list_path == 2 and c1.rollbacks == c0.rollbacks
    xh = sh.positions()
    for r in picks:
        so = ors[r]
        so.run(14, info["timestep"], info["temperature"], seed=int(seeds[r]), flags=flags)
        assert np.abs(xh[r] - so.positions()[0]).max() <= POS_ATOL_20STEP, r
        assert sh.context(r).step == so.context(0).step
        assert np.allclose(np.array(sh.context(r).semiaxes), np.array(so.context(0).semiaxes), rtol=0, atol=1e-8), r
        so.close()


# ---------------------------------------------------------------- small-dt runs: compensated positions

@pytest.mark.parametrize("n_beads", [30000, 62178])
def test_fine_timestep_displacement_field_matches_oracle(hip, oracle, n_beads):
    """The reference's deterministic configuration (simulation_fine_sampling/simulation_driver.cc:30-34: T = 0, dt = 1e-5 / 100,
    fp64) moves a bead by mu F dt ~ 1e-7 ... 2e-6 per step at |x| up to 6.3 / 8.0 -- 0.1 ... 4 ulp of an fp32 coordinate -- and what it
    outputs is exactly those small displacements.  300 steps from a relaxed state, device vs oracle, on the DISPLACEMENT x(300) - x(0):
      * max error <= 1e-3 of the median displacement (measured 6e-5), median error <= 1e-5 of it (measured 1e-7);
      * no coordinate that moves in the oracle stands still on the device, none moves the other way;
      * gd_run selects the compensated update by itself at T = 0 (gd_context.compensated);
      * the plain fp32 update (GD_RUN_UNCOMPENSATED) on the same state misses the first bound by more than 20 x (measured 250 x: 3 % median
        error, 5 % of the coordinates stuck) -- i.e. the bounds can tell the two apart."""
    s, info = wl.genome_interphase(hip, n_beads=n_beads)
    s.begin_phase()
    s.run(3000, info["timestep"], info["temperature"], seed=SEED + 3, flags=0)       # relax the random-walk start: |F| ~ 9 per component
    assert s.context().compensated == 0                   # sigma = 4.5e-3 per step: plain update
    x0 = s.positions()
    s.close()
    so, _ = wl.genome_interphase(oracle, n_beads=n_beads)
    so.set_positions(x0)
    so.begin_phase()
    so.run(300, 1e-7, 0.0, seed=1, flags=g.RUN_WALL_DYNAMICS)
    do = so.positions()[0] - x0[0]
    med = np.median(np.abs(do))
    assert 1e-4 < med < 1e-3, med
    moving = np.abs(do) > 1e-6
    errs = {}
    for tag, fl in (("auto", 0), ("plain", g.RUN_UNCOMPENSATED)):
        sh, _ = wl.genome_interphase(hip, n_beads=n_beads)
        sh.set_positions(x0)
        sh.begin_phase()
        sh.run(300, 1e-7, 0.0, seed=1, flags=g.RUN_WALL_DYNAMICS | fl)
        assert sh.context().compensated == (1 if tag == "auto" else 0) and sh.context().list_path == 2
        dh = sh.positions()[0] - x0[0]
        errs[tag] = np.abs(dh - do)
        if tag == "auto":
            assert errs[tag].max() <= 1e-3 * med, (errs[tag].max(), med)
            assert np.median(errs[tag]) <= 1e-5 * med
            assert np.all(dh[moving] != 0) and np.all(np.sign(dh[moving]) == np.sign(do[moving]))
            assert abs(sh.context().semiaxes[0] - so.context().semiaxes[0]) <= 1e-9        # (a semiaxis of 6 ... 8 moved by ~1e-6 in these steps)
        sh.close()
    assert errs["plain"].max() > 20 * 1e-3 * med and np.median(errs["plain"]) > 1e-3 * med


def test_fp64_positions_survive_the_boundary_and_compensated_runs_on_both_paths(hip, oracle):
    """gd_set_positions keeps what fp32 drops of the fp64 input as the residual of the compensated update: gd_get_positions returns
    the input to ~1e-14, a noisy (uncompensated) run discards the residuals (positions are fp32 values again), and a T = 0 run
    at dt = 1e-7 matches the oracle on both kernel paths from an fp64 start the oracle sees identically."""
    _, _, _, _, flags = CASES["genome"]
    rng = np.random.default_rng(5)
    for path in ("generic", "tiled"):
        sh, dt, kT, _ = build(hip, "genome")
        so, *_ = build(oracle, "genome")
        sh.set_tuning(kernel_path=PATHS[path])
        x = so.positions() + 1e-9 * rng.standard_normal(so.positions().shape)      # not representable in fp32
        for s in (sh, so):
            s.set_positions(x)
            s.begin_phase()
        assert np.abs(sh.positions() - x).max() <= 1e-13
        for s in (sh, so):
            s.run(200, 1e-7, 0.0, seed=3, flags=flags)
        _assert_path(sh, path)
        assert sh.context().compensated == 1
        dh, do = sh.positions() - x, so.positions() - x
        assert np.abs(dh - do).max() <= 2e-3 * np.median(np.abs(do)), (path, np.abs(dh - do).max(), np.median(np.abs(do)))
        sh.run(3, dt, kT, seed=4, flags=flags)
        assert sh.context().compensated == 0
        xs = sh.positions()
        assert np.array_equal(xs, xs.astype(np.float32).astype(np.float64))


def test_list_width_follows_the_tile_class(hip, oracle):
    """The default list width is a rule on the state (gdyn_capi.hip, class_skin): 0.9 x cutoff where the largest tile of the wider
    list fits the three-block LDS class (S-genome-30k), 0.75 where it does not (S-genome-62k); a caller-chosen skin stays.  The
    width changes cost only: forces on the state that the wide list produced match the oracle."""
    cut = 0.30
    for n_beads, R, wide in ((30000, 16, True), (62178, 8, False)):
        s, info = wl.genome_interphase(hip, n_beads=n_beads, n_replicas=R)
        s.begin_phase()
        s.run(3000, info["timestep"], info["temperature"], seed=SEED + 9, flags=0)
        for _ in range(4):      # (the random-walk start is denser than the relaxed state: its tiles shrink into the class as it relaxes)
            if not wide or s.context().list_radius > cut * 1.8:
                break
            s.run(1500, info["timestep"], info["temperature"], seed=SEED + 9, flags=0)
        c = s.context()
        assert c.list_path == 2 and c.tile_capacity == 3312 or not wide, (c.list_path, c.tile_capacity)
        assert abs(c.list_radius - cut * (1.0 + (0.9 if wide else 0.75))) < 1e-3, (n_beads, c.list_radius, c.largest_tile, c.tile_capacity)
        if wide:
            assert c.rebuild_interval >= 17, c.rebuild_interval
            x = s.positions()
            so, _ = wl.genome_interphase(oracle, n_beads=n_beads, n_replicas=1)
            so.set_positions(x[3][None])
            Fo = so.forces()[0]
            assert np.abs(s.forces()[3] - Fo).max() <= FORCE_RTOL * np.abs(Fo).max()
        s.close()
    s, info = wl.genome_interphase(hip, n_beads=30000, n_replicas=16)
    s.set_tuning(skin=0.75)
    s.begin_phase()
    s.run(1500, info["timestep"], info["temperature"], seed=SEED + 9, flags=0)
    assert abs(s.context().list_radius - cut * 1.75) < 1e-3


# ---------------------------------------------------------------------------------------------------------------------------
# round 5: one seed -> one trajectory; independent handles on independent host threads

def _run_once_for_determinism(hip, n_beads, n_rep, relax, steps):
    s, info = wl.genome_interphase(hip, n_beads=n_beads, n_replicas=n_rep)
    s.set_tuning(kernel_path=PATHS["tiled"])
    dt, kT = info["timestep"], info["temperature"]
    s.begin_phase()
    s.run(relax, dt, kT, seed=SEED + 5, flags=0)
    s.begin_phase()
    flags = g.RUN_UPDATE_SCALES | g.RUN_WALL_DYNAMICS
    c0 = s.context()
    s.run(steps, dt, kT, seed=SEED, flags=flags)
    _assert_path(s, "tiled")
    s.contacts_update(0.4 * s.context().bead_scale)
    out = dict(x=s.positions(), x32=s.positions_f32(), rows=[s.contacts(r) for r in (0, n_rep - 1)], e=s.energy(),
               rebuilds=s.context().rebuilds - c0.rebuilds, K=s.context().rebuild_interval, semi=np.array(s.context().semiaxes),
               react=np.array(s.context().axial_reaction))
    s.close()
    return out


def test_same_seed_gives_the_same_trajectory_bit_for_bit(hip):
    """The reference is one thread in fp64: one seed, one trajectory (5-sim-genome/scripts/run_simulation:8-25;
    simulation_fine_sampling/simulation_driver.cc:44-54 restarts from a stored frame and relies on it).  On the device the order
    of the beads inside a neighbour-search cell used to be the arrival order of atomics, and with it every list order and fp32
    summation order: since round 5 the slot order is (cell, bead id) and the thread order inside a block is stable, so two runs of
    the same handle set-up and seed are identical BIT FOR BIT -- positions (fp64 boundary and fp32), energies, wall state and the
    contact-map rows -- over several rebuild intervals of the tiled path at 30 000 beads x 8 replicas."""
    a = _run_once_for_determinism(hip, 30000, 8, 600, 90)
    b = _run_once_for_determinism(hip, 30000, 8, 600, 90)
    assert a["rebuilds"] >= 3 and a["rebuilds"] == b["rebuilds"] and a["K"] == b["K"], (a["rebuilds"], b["rebuilds"], a["K"], b["K"])
    assert np.array_equal(a["x"], b["x"]) and np.array_equal(a["x32"], b["x32"])
    assert np.array_equal(a["e"], b["e"]) and np.array_equal(a["semi"], b["semi"]) and np.array_equal(a["react"], b["react"])
    for ra, rb in zip(a["rows"], b["rows"]):
        assert len(ra) > 1000 and np.array_equal(ra, rb)


def test_same_seed_same_trajectory_on_the_generic_and_periodic_paths(hip):
    """The same property on the global-gather path and on a periodic box (whole-row tiles): 1 kb model, 3 replicas."""
    outs = []
    for path in ("generic", "tiled"):
        pair = []
        for _ in range(2):
            s, dt, kT, flags = build(hip, "chromatin_1kb", n_replicas=3)
            s.set_tuning(kernel_path=PATHS[path])
            s.begin_phase()
            s.run(120, dt, kT, seed=SEED, flags=flags)
            _assert_path(s, path)
            pair.append((s.positions(), s.context().rebuilds))
            s.close()
        assert pair[0][1] == pair[1][1] and pair[0][1] >= 2
        assert np.array_equal(pair[0][0], pair[1][0]), path
        outs.append(pair[0][0])


def test_two_handles_on_two_host_threads_race_their_first_launches(hip, oracle):
    """include/gdyn.h: independent handles may live on different threads (and devices).  Two tiled handles are created and stepped
    from two host threads at the same time -- their first launches race through the per-device kernel set-up (the > 64 KB LDS
    opt-in, once per device under std::call_once since round 5; it used to sit behind an unsynchronised `static bool once` in
    every launcher) -- and each must match the oracle as a solo handle does."""
    import threading
    n_beads, steps = 4000, 30
    results, errors = {}, []
    gate = threading.Barrier(2)

    def worker(k):
        try:
            gate.wait()
            s, info = wl.genome_interphase(hip, n_beads=n_beads, n_replicas=2 + k, bead_scale_init=0.8)
            s.set_tuning(kernel_path=PATHS["tiled"])
            F = s.forces()
            s.begin_phase()
            s.run(steps, info["timestep"], info["temperature"], seed=SEED + k, flags=g.RUN_UPDATE_SCALES | g.RUN_WALL_DYNAMICS)
            results[k] = (F, s.positions(), s.context().list_path)
            s.close()
        except Exception as e:      # noqa: BLE001 -- reported by the main thread
            errors.append((k, repr(e)))

    th = [threading.Thread(target=worker, args=(k,)) for k in range(2)]
    [t.start() for t in th]
    [t.join() for t in th]
    assert not errors, errors
    for k in range(2):
        so, info = wl.genome_interphase(oracle, n_beads=n_beads, n_replicas=2 + k, bead_scale_init=0.8)
        Fo = so.forces()
        so.begin_phase()
        so.run(steps, info["timestep"], info["temperature"], seed=SEED + k, flags=g.RUN_UPDATE_SCALES | g.RUN_WALL_DYNAMICS)
        F, x, path = results[k]
        assert path == PATHS["tiled"]
        assert np.abs(F - Fo).max() <= FORCE_RTOL * np.abs(Fo).max()
        assert np.abs(x - so.positions()).max() <= 2 * POS_ATOL_20STEP
        so.close()


def test_ragged_rows_globule_and_gas_in_one_handle(hip, oracle):
    """Ragged rows (round 5): a dense globule (lists of several hundred entries) and a dilute gas (a handful) in ONE handle -- the
    rows of a k_step wave are as wide as that wave's longest list, not the handle's.  Forces and the pair set of the resident
    lists equal the oracle's on the same positions (two replicas, different globules); the lists take a fraction of what uniform
    rows of the longest list would; a short run (the globule swells, its beads' lists change by whole chunks between builds:
    predicted widths + on-device repair, no rollback for a row) still matches the oracle's trajectory."""
    rng = np.random.default_rng(23)
    n_glob, n_gas, R = 3000, 9000, 2
    n = n_glob + n_gas
    def cloud(seed):
        g0 = np.random.default_rng(seed)
        v = g0.normal(size=(n_glob, 3))
        glob = 1.0 * v / np.linalg.norm(v, axis=1)[:, None] * g0.random((n_glob, 1)) ** (1 / 3)       # ~700 beads per unit volume
        gas = (g0.random((n_gas, 3)) - 0.5) * 9.0 + np.array([7.0, 0.0, 0.0])                          # ~12 per unit volume, elsewhere
        return np.concatenate([glob, gas])
    x = np.stack([cloud(101), cloud(202)]).astype(np.float32).astype(np.float64)
    def make(lib, r_count):
        s = g.System(lib, n, r_count)
        s.set_bead_params(a=(np.arange(n) % 2).astype(float), b=((np.arange(n) + 1) % 2).astype(float), mobility=np.ones(n))
        s.set_pair_softcore(2.0, 0.3, 2.0, 0.24)
        return s
    sh = make(hip, R)
    sh.set_tuning(kernel_path=2)
    sh.set_positions(x)
    Fh = sh.forces()
    c = sh.context()
    assert c.list_path == 2
    so = make(oracle, 1)
    for r in range(R):
        so.set_positions(x[r])
        Fo = so.forces()
        assert np.abs(Fh[r] - Fo).max() <= FORCE_RTOL * np.abs(Fo).max()
        ph = {tuple(p) for p in sh.search_pairs(0.3, replica=r)}
        po = {tuple(p) for p in so.search_pairs(0.3)}
        for i, j in ph ^ po:
            assert abs(np.linalg.norm(x[r][i] - x[r][j]) - 0.3) < 1e-6
        assert len(po) > 50000
    # the globule's beads hold hundreds of entries, the gas a few: mean far below the longest, and the rows follow the mean
    lens = np.zeros(n)
    for i, j in po:
        lens[i] += 1; lens[j] += 1
    assert lens[:n_glob].mean() > 60 and lens[n_glob:].mean() < 3
    # (contact pairs are within 0.3, list entries within the list radius 0.525: ~5 x as many; uniform rows hold the longest list for every bead)
    uniform_bytes = 2.0 * R * ((n + 511) // 512 * 512) * 5.0 * lens.max()
    assert 0 < c.list_bytes < 0.5 * uniform_bytes, (c.list_bytes, uniform_bytes, lens.max())
    # a short run from here: the globule swells (lists shrink and shift between the classes), builds every 2 steps
    sh.set_tuning(kernel_path=2, rebuild_interval=2, adapt_interval=0)
    sh.begin_phase()
    rb0 = sh.context().rollbacks
    sh.run(12, 2e-6, 1.0, seed=SEED, replica_seeds=[5, 6])
    c1 = sh.context()
    assert c1.list_path == 2 and c1.rollbacks == rb0, (c1.rollbacks, rb0)
    x1 = sh.positions()
    for r in range(R):
        so.set_positions(x[r]); so.begin_phase()
        so.run(12, 2e-6, 1.0, seed=[5, 6][r])
        assert np.abs(x1[r] - so.positions()[0]).max() <= POS_ATOL_20STEP


def test_fine_timestep_with_the_droplet_term_keeps_its_share_of_the_displacement(hip, oracle):
    """The compensated update covers the droplet kernel too (round 5): gd_fine_sampling's force field includes the nucleolar droplet
    attraction when nucleolus_droplet_energy != 0 (simulation_driver_forcefield.cc:153-178), and at T = 0, dt = 1e-7 the droplet's share of
    mu F dt on a nucleolar bead is around or below an ulp of its coordinate -- added in plain fp32 behind k_step it was rounded away.
    300 steps, device vs oracle, on the displacement of the TARGET beads: within 1e-3 of their median displacement; with the droplet
    force switched off on the device only, the same bound fails (so the test sees the term)."""
    n, nt = 6000, 300
    rng = np.random.default_rng(4)
    tg = np.sort(rng.choice(n, size=nt, replace=False))
    s0, info = wl.genome_interphase(hip, n_beads=n)
    s0.begin_phase()
    s0.run(2000, info["timestep"], info["temperature"], seed=SEED + 3, flags=0)
    x0 = s0.positions()
    s0.close()
    x0[0, tg] = x0[0, tg[0]] + 0.3 * rng.normal(size=(nt, 3))           # a clump of targets: many droplet pairs in range
    x0 = x0.astype(np.float32).astype(np.float64)
    def run(lib, energy):
        s, _ = wl.genome_interphase(lib, n_beads=n)
        s.set_pair_softwell(energy, 0.2, 0.4, tg)
        s.set_positions(x0)
        s.begin_phase()
        s.run(300, 1e-7, 0.0, seed=1, flags=g.RUN_WALL_DYNAMICS)
        comp = s.context().compensated
        d = s.positions()[0] - x0[0]
        s.close()
        return d, comp
    do, _ = run(oracle, 0.05)
    do0, _ = run(oracle, 0.0)
    share = np.abs(do[tg] - do0[tg])                       # what the droplet term contributes to the targets' displacement
    med = np.median(np.abs(do[tg]))
    assert np.median(share) > 1e-8 and np.median(share) < 0.2 * med, (np.median(share), med)      # a small share: the part plain fp32 drops
    dh, comp = run(hip, 0.05)
    assert comp == 1
    err = np.abs(dh[tg] - do[tg])
    assert err.max() <= 1e-3 * med, (err.max(), med)
    assert np.abs(dh[tg] - do0[tg]).max() > 3 * err.max()          # (without the term the targets would be off by its share)


def test_rows_recover_from_a_first_guess_that_is_far_too_small(hip, oracle):
    """Ragged rows without history start from the caller's guess (gd_tuning.list_width).  A guess of 8 entries where the lists hold ~30
    makes EVERY k_step wave outgrow its rows at the first build: more waves than the repair launch has blocks -- flagged, the build is
    repeated with a repair block per wave and a pool sized from what the cursor counted -- and the handle ends up with exact rows:
    forces and pair set as the oracle's, then a run without a rollback for a row."""
    n, R = 8000, 2
    sh, info = wl.genome_interphase(hip, n_beads=n, n_replicas=R)
    so, _ = wl.genome_interphase(oracle, n_beads=n, n_replicas=R)
    sh.set_tuning(kernel_path=2, list_width=8)
    x = sh.positions()
    Fh, Fo = sh.forces(), so.forces()
    assert sh.context().list_path == 2
    assert np.abs(Fh - Fo).max() <= FORCE_RTOL * np.abs(Fo).max()
    for r in range(R):
        ph = {tuple(p) for p in sh.search_pairs(0.3, replica=r)}
        po = {tuple(p) for p in so.search_pairs(0.3, replica=r)}
        for i, j in ph ^ po:
            assert abs(np.linalg.norm(x[r][i] - x[r][j]) - 0.3) < 1e-6
    c0 = sh.context()
    assert c0.list_bytes > 0 and c0.list_entries / n > 15
    for s in (sh, so):
        s.begin_phase()
        s.run(20, info["timestep"], info["temperature"], seed=SEED, flags=g.RUN_UPDATE_SCALES | g.RUN_WALL_DYNAMICS)
    assert np.abs(sh.positions() - so.positions()).max() <= POS_ATOL_20STEP
