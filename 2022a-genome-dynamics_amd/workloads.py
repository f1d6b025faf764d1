"""Synthetic inputs of SURVEY.md section 8(d) (the reference ships no config or data files).

Every number is either a reference default (cited) or a builder-chosen constant that is
returned in the ``info`` dict so benchmark lines can print it.  Builders take any
:class:`Lib` (product or oracle) and return a configured :class:`System`.
"""
from __future__ import annotations

import numpy as np

from . import (POT_HARMONIC, POT_SEMISPRING, POT_SPRING, POT_SOFTCORE, System)

MASTER_SEED = 20220101

# hg19 chromosome lengths in Mb (chr1..22, X, Y); diploid male = 22 pairs + X + Y = 46 chains
_HG19_MB = [249, 243, 198, 191, 181, 171, 159, 146, 141, 136, 135, 134, 115, 107, 103, 90, 81, 78, 59, 63, 48, 51, 155, 59]


# every key of the stage-5 configuration with the reference's default (5-sim-genome/src/config_entries.inc:1-90);
# the C++ side treats all of them as mandatory, like the reference (simulation_common/simulation_config.cc:29-33)
DEFAULT_CONFIG = {
    "a_core_diameter": 0.2, "b_core_diameter": 0.2, "a_core_repulsion": 2.0, "b_core_repulsion": 2.0,
    "chromatin_bond_spring": 0.1, "chromatin_bond_length": 0.2, "chromatin_mobility": 1.0,
    "a_core_bond_spring": 0.0, "a_core_bond_length": 0.0, "b_core_bond_spring": 0.0, "b_core_bond_length": 0.0,
    "a_core_2nd_bond_spring": 0.0, "b_core_2nd_bond_spring": 0.0,
    "nucleolus_sidebeads": 2, "nucleolus_a_factor": 5, "nucleolus_b_factor": 5, "nucleolus_bond_spring": 5.0,
    "nucleolus_bond_length": 0.0, "nucleolus_droplet_energy": 0.0, "nucleolus_droplet_decay": 0.2,
    "nucleolus_droplet_cutoff": 0.4, "nucleolus_mobility": 1.0,
    "wall_init_semiaxes": [1.0, 1.0, 1.0], "wall_semiaxes_spring": [1.0e4, 1.0e4, 1.0e4], "wall_packing_spring": 5000,
    "wall_a_factor": 5, "wall_b_factor": 5, "wall_mobility": 1.0e-4,
    "bead_scale_init": 1.0, "bead_scale_tau": 1.0, "bond_scale_init": 1.0, "bond_scale_tau": 1.0,
    "init_coarse_graining": 100, "init_bead_diameter": 0.2, "init_bead_repulsion": 5.0, "init_bond_length": 0.2,
    "init_bond_spring": 500.0, "init_bend_energy": 0.0, "init_spindle_spring": 1.0, "init_spindle_point": [0, 0, 0],
    "init_packing_radius": 1.0, "init_packing_spring": 0.0, "init_start_point": [5, 0, 0], "init_start_stddev": 1.0,
    "init_mobility": 1.0, "init_temperature": 0.1, "init_timestep": 1e-4, "init_spacestep": 0,
    "init_spindle_steps": 10000, "init_packing_steps": 10000, "init_sampling_interval": 1000,
    "init_logging_interval": 1000, "init_refinement_method": "spline",
    "relaxation_temperature": 1.0, "relaxation_timestep": 1.0e-5, "relaxation_spacestep": 0, "relaxation_steps": 10000,
    "relaxation_sampling_interval": 100, "relaxation_logging_interval": 100,
    "interphase_temperature": 1.0, "interphase_timestep": 1e-5, "interphase_spacestep": 0, "interphase_steps": 10000,
    "interphase_sampling_interval": 1000, "interphase_logging_interval": 100,
    "contactmap_distance": 0.4, "contactmap_update_interval": 100, "contactmap_thinning_rate": 100,
    "spindle_seed": 0, "interphase_seed": 0,
}


def chain_lengths(n_beads, min_len=5):
    sizes = np.array(_HG19_MB[:22] * 2 + _HG19_MB[22:], dtype=float)  # 46 chains
    raw = sizes / sizes.sum() * n_beads
    lens = np.maximum(np.floor(raw).astype(int), min_len)
    # distribute the remainder to the largest chains, deterministically
    k = 0
    order = np.argsort(-sizes, kind="stable")
    while lens.sum() < n_beads:
        lens[order[k % len(order)]] += 1
        k += 1
    while lens.sum() > n_beads:
        i = order[k % len(order)]
        if lens[i] > min_len:
            lens[i] -= 1
        k += 1
    return lens


def ab_types(n_beads, rng):
    """40% A (1,0) / 40% B (0,1) / 20% u (.5,.5) in runs of 10-50 beads
    (type values: 2-signal/src/model_genome/model_genome.py:61-65)."""
    a = np.empty(n_beads)
    b = np.empty(n_beads)
    i = 0
    while i < n_beads:
        run = int(rng.integers(10, 51))
        t = rng.choice(3, p=[0.4, 0.4, 0.2])
        av, bv = [(1.0, 0.0), (0.0, 1.0), (0.5, 0.5)][t]
        a[i:i + run] = av
        b[i:i + run] = bv
        i += run
    return a, b


def confined_random_walks(lens, radius, step, rng):
    """Space-filling random-walk chains inside a sphere (builder-chosen initial condition)."""
    n = int(np.sum(lens))
    pos = np.empty((n, 3))
    starts = np.concatenate([[0], np.cumsum(lens)[:-1]])
    nch = len(lens)
    # start points uniform in the ball of radius 0.9 R
    d = rng.normal(size=(nch, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    cur = d * (0.9 * radius * rng.random(nch)[:, None] ** (1 / 3))
    for k in range(int(np.max(lens))):
        alive = lens > k
        pos[starts[alive] + k] = cur[alive]
        s = rng.normal(size=(nch, 3))
        s *= step / np.linalg.norm(s, axis=1, keepdims=True)
        nxt = cur + s
        for _ in range(8):   # steps that would leave the sphere are redrawn (no back-tracking onto old beads)
            out = np.linalg.norm(nxt, axis=1) > 0.97 * radius
            if not out.any():
                break
            s2 = rng.normal(size=(int(out.sum()), 3))
            s2 *= step / np.linalg.norm(s2, axis=1, keepdims=True)
            nxt[out] = cur[out] + s2
        out = np.linalg.norm(nxt, axis=1) > 0.97 * radius
        nxt[out] = cur[out] * (1 - step / radius)   # last resort: step towards the centre
        cur = nxt
    return pos


def genome_interphase(lib, n_beads=30000, n_replicas=1, device=0, second_bond_spring=5.0, bead_scale_init=1.0,
                      wall_spring=1.0e4, seed=MASTER_SEED, phi=0.30):
    """cfg3 / S-genome: 100 kb interphase whole-genome model with the force fields of
    5-sim-genome/src/simulation_interphase/simulation_driver_forcefield.cc:19-235."""
    rng = np.random.default_rng(seed)
    lens = chain_lengths(n_beads)
    a, b = ab_types(n_beads, rng)
    sig_a, sig_b, eps = 0.30, 0.24, 2.0          # 4-sim-ab/box/src/simulation/simulation_config.hpp:12-15
    sig_mean = 0.27
    radius = sig_mean * (n_beads / (8 * phi)) ** (1 / 3)   # volume fraction phi of beads of diameter sig_mean
    s = System(lib, n_beads, n_replicas, device=device)
    s.set_bead_params(a=a, b=b, mobility=np.ones(n_beads))   # chromatin_mobility 1.0, config_entries.inc:10
    s.set_pair_softcore(eps, sig_a, eps, sig_b, 2, 3, 8, 3, mix=True, scale_by_bead_scale=True)
    chain = System.bond_params(POT_SEMISPRING, k_a=70.0, l_a=0.2, k_b=70.0, l_b=0.2, mix=True, scale_by_bond_scale=True)
    loop = System.bond_params(POT_HARMONIC, k_a=second_bond_spring, k_b=second_bond_spring, mix=True, scale_by_bond_scale=True)
    st = 0
    ranges = []
    for n in lens:
        s.add_bond_range(chain, st, st + n, 1)
        if second_bond_spring != 0:
            s.add_bond_range(loop, st, st + n, 2)
        ranges.append((st, st + int(n)))
        st += int(n)
    s.set_ellipsoid_wall(eps, sig_a, eps, sig_b, wall_a_factor=5.0, wall_b_factor=5.0, packing_spring=5000.0,
                         semiaxes_spring=(wall_spring,) * 3, mobility=1.0e-4, init_semiaxes=(radius,) * 3)
    s.set_scaling(bead_scale_init, 1.0, bead_scale_init, 1.0)   # tau defaults, config_entries.inc:40-43
    x0 = np.stack([confined_random_walks(lens, radius, 0.2, np.random.default_rng(seed + 1 + r)) for r in range(n_replicas)])
    s.set_positions(x0)
    info = dict(workload="S-genome-%dk" % round(n_beads / 1000), n_beads=n_beads, chains=len(lens), sigma_a=sig_a,
                sigma_b=sig_b, eps=eps, bond_spring=70.0, bond_length=0.2, second_bond_spring=second_bond_spring,
                wall_radius=radius, wall_spring=wall_spring, bead_scale_init=bead_scale_init, phi=phi,
                temperature=1.0, timestep=1.0e-5, ranges=ranges)
    return s, info


def spindle(lib, n_beads=300, n_replicas=1, device=0, seed=MASTER_SEED, bend_energy=1.0):
    """cfg2 / S-spindle: coarse ana/telophase model, forces of
    5-sim-genome/src/simulation_spindle/simulation_driver.cc:90-172 with the init_* defaults
    (config_entries.inc:46-66) and init_bend_energy 1.0 so the angle kernel is exercised."""
    rng = np.random.default_rng(seed)
    lens = chain_lengths(n_beads)
    s = System(lib, n_beads, n_replicas, device=device)
    s.set_bead_params(mobility=np.ones(n_beads))
    s.set_pair_softcore(5.0, 0.2, 0.0, 0.0, 2, 3, 8, 3, mix=False)
    bond = System.bond_params(POT_SEMISPRING, k_a=500.0, l_a=0.2)
    st = 0
    cen = []
    ranges = []
    for n in lens:
        n = int(n)
        s.add_bond_range(bond, st, st + n, 1)
        s.add_bending_range(st, st + n, bend_energy, per_bead=False)
        c = st + n // 2
        cen += [c - 1, c, c + 1]
        ranges.append((st, st + n))
        st += n
    s.add_point_source(POT_HARMONIC, 1.0, 0.0, (0.0, 0.0, 0.0), targets=np.array(cen, dtype=np.uint32))
    s.add_point_source(POT_SEMISPRING, 0.5, 1.0, (0.0, 0.0, 0.0))   # packing well (init_packing_spring 0 by default; 0.5 to exercise it)
    # randomly-directed rods around init_start_point (5,0,0), simulation_driver.cc:183-204
    x0 = np.empty((n_replicas, n_beads, 3))
    for r in range(n_replicas):
        rr = np.random.default_rng(seed + 1 + r)
        for (b0, b1) in ranges:
            c = np.array([5.0, 0, 0]) + rr.normal(size=3)
            d = rr.normal(size=3)
            d *= 0.2 / np.linalg.norm(d)
            k = np.arange(b1 - b0)[:, None]
            x0[r, b0:b1] = c - d * (b1 - b0) / 2 + k * d
    s.set_positions(x0)
    info = dict(workload="S-spindle-%d" % n_beads, n_beads=n_beads, temperature=0.1, timestep=1.0e-4, ranges=ranges)
    return s, info


def ab_box(lib, n_chains=100, chain_len=20, box=4.0, n_replicas=1, device=0, seed=MASTER_SEED):
    """cfg1 / S-AB-box: periodic A/B blend, forces of 4-sim-ab/box/src/simulation/simulation_driver.cc:93-141."""
    n = n_chains * chain_len
    a = np.zeros(n)
    b = np.zeros(n)
    for c in range(n_chains):   # alternating pure-A / pure-B chains (4-sim-ab/sphere/scripts/make_chain_definition:9-20)
        (a if c % 2 == 0 else b)[c * chain_len:(c + 1) * chain_len] = 1.0
    s = System(lib, n, n_replicas, box=(box,) * 3, device=device)
    s.set_bead_params(a=a, b=b, mobility=np.ones(n))
    s.set_pair_softcore(2.0, 0.30, 2.0, 0.24, 2, 3, 8, 3, mix=True)
    bond = System.bond_params(POT_HARMONIC, k_a=70.0)
    for c in range(n_chains):
        s.add_bond_range(bond, c * chain_len, (c + 1) * chain_len, 1)
    x0 = np.empty((n_replicas, n, 3))
    for r in range(n_replicas):
        rr = np.random.default_rng(seed + 1 + r)
        for c in range(n_chains):   # straight rods, simulation_driver.cc:151-180
            ctr = rr.random(3) * box
            d = rr.normal(size=3)
            d /= np.linalg.norm(d)
            k = np.arange(chain_len)[:, None] - (chain_len - 1) / 2
            x0[r, c * chain_len:(c + 1) * chain_len] = ctr + 0.1 * k * d
    s.set_positions(x0)
    info = dict(workload="S-AB-box-%d" % n, n_beads=n, box=box, temperature=1.0, timestep=1.0e-5)
    return s, info


def chromatin_1kb(lib, n_beads=250000, n_replicas=1, device=0, seed=MASTER_SEED, phi=0.1, n_loops=2500, n_glues=5000):
    """cfg4 / S-1kb: single chain, periodic box, forces of 3-sim-1kb/src/simulation/simulation.cpp:99-184."""
    rng = np.random.default_rng(seed)
    sig, eps, sig_att, eps_att = 1.0, 2.0, 1.5, 0.2
    box = (n_beads * np.pi / 6 * sig ** 3 / phi) ** (1 / 3)
    s = System(lib, n_beads, n_replicas, box=(box,) * 3, device=device)
    s.set_bead_params(mobility=np.ones(n_beads), bending_energy=np.full(n_beads, 1.0))
    s.set_pair_softcore(eps, sig, -eps_att, sig_att, 2, 3, 8, 3, mix=False)
    s.add_bond_range(System.bond_params(POT_SPRING, k_a=100.0, l_a=1.0), 0, n_beads, 1)
    s.add_bending_range(0, n_beads, 0.0, per_bead=True)
    # random walk with bond length 1 (3-sim-1kb/src/simulation/inits/utils.hpp:9-47), unwrapped coordinates
    steps = rng.normal(size=(n_beads, 3))
    steps /= np.linalg.norm(steps, axis=1, keepdims=True)
    x = np.cumsum(steps, axis=0)
    x += box / 2 - x.mean(axis=0)
    s.set_positions(np.broadcast_to(x, (n_replicas, n_beads, 3)))
    # static loop and glue pair lists (kinetics off for timing): pick pairs that are close in space
    i = rng.integers(0, n_beads - 200, size=n_loops)
    loops = np.stack([i, i + rng.integers(20, 200, size=n_loops)], axis=1).astype(np.uint32)
    s.set_dynamic_pairs(0, System.bond_params(POT_SPRING, k_a=10.0, l_a=sig), loops)
    j = rng.integers(0, n_beads - 4, size=n_glues)
    glues = np.stack([j, j + 3], axis=1).astype(np.uint32)
    s.set_dynamic_pairs(1, System.bond_params(POT_SOFTCORE, k_a=-1.0, l_a=1.5, p=8, q=3, minimum_image=True), glues)
    info = dict(workload="S-1kb-%dk" % round(n_beads / 1000), n_beads=n_beads, box=box, temperature=1.0, timestep=1.0e-4)
    return s, info
