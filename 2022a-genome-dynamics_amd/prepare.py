"""Input side of the whole-genome pipeline: what the reference's `prepare` and `refine`
steps compute before the stepping path runs, restated on numpy.

* :func:`derive_seeds` -- spindle/interphase seeds from the master seed
  (5-sim-genome/src/prepare/run.py:49-55: two draws of ``RandomState(seed).randint(1000000)``).
* :func:`make_system` -- genome bead table -> particles, chains, nucleolar side beads and bonds
  (5-sim-genome/src/prepare/system_definition.py:56-140; type precedence :16-24).
* :func:`refine_path_spline`, :func:`refine_positions` -- cubic-spline upsampling of the coarse packed
  conformation with bin-midpoint parameterisation (5-sim-genome/src/refine/refinement.py:9-19, refine/run.py:9-46).

Pinned by fixtures generated here by importing the reference modules (tests/golden/make_prepare_fixtures.py).
"""
from __future__ import annotations

import numpy as np
from scipy.interpolate import CubicSpline

SEED_MAX = 999999
TYPE_A, TYPE_B, TYPE_U, TYPE_CEN, TYPE_ANOR, TYPE_BNOR, TYPE_NUC = 1, 2, 3, 4, 5, 6, 7
# a bead carrying several tags takes the first matching type in this order
_TAG_PRECEDENCE = (("anor", TYPE_ANOR), ("bnor", TYPE_BNOR), ("cen", TYPE_CEN), ("A", TYPE_A), ("B", TYPE_B), ("u", TYPE_U))


def derive_seeds(seed):
    rs = np.random.RandomState(seed)
    return int(rs.randint(SEED_MAX + 1)), int(rs.randint(SEED_MAX + 1))


def _type_of(tags):
    have = set(tags.split(","))
    for tag, code in _TAG_PRECEDENCE:
        if tag in have:
            return code
    raise ValueError(f"bead with no known tag: {tags!r}")


def make_system(genome, config):
    """genome: iterable of rows (chain, start, end, A, B, tags) in file order.  Returns a dict of arrays in the
    dtypes of the trajectory file's /metadata group."""
    types, ab, chains = [], [], []          # chains: [name, start, end, cen_start, cen_end]
    for chain, _s, _e, a, b, tags in genome:
        if not chains or chains[-1][0] != chain:
            if any(c[0] == chain for c in chains):
                raise ValueError(f"chain {chain!r} is not contiguous in the genome table")
            chains.append([chain, len(types), len(types), None, None])
        t = _type_of(tags)
        if t == TYPE_CEN:
            if chains[-1][3] is None:
                chains[-1][3] = len(types)
            chains[-1][4] = len(types)
        types.append(t)
        ab.append((float(a), float(b)))
        chains[-1][2] = len(types)
    spans, bonds = [], []
    for name, start, end, _c0, _c1 in chains:         # nucleolar side beads hang off ACTIVE NORs only
        first = len(types)
        for nor in range(start, end):
            if types[nor] != TYPE_ANOR:
                continue
            for _ in range(int(config["nucleolus_sidebeads"])):
                bonds.append((nor, len(types)))
                types.append(TYPE_NUC)
                ab.append((float(config["nucleolus_a_factor"]), float(config["nucleolus_b_factor"])))
        if len(types) != first:
            spans.append((name, first, len(types)))
    return {
        "particle_types": np.array(types, dtype=np.int8),
        "ab_factors": np.array(ab, dtype=np.float32).reshape(-1, 2),
        "chromosome_names": [c[0] for c in chains],
        "chromosome_ranges": np.array([[c[1], c[2]] for c in chains], dtype=np.int32).reshape(-1, 2),
        "centromere_ranges": np.array([[c[3] or 0, c[4] or 0] for c in chains], dtype=np.int32).reshape(-1, 2),
        "nucleolus_names": [s[0] for s in spans],
        "nucleolus_ranges": np.array([[s[1], s[2]] for s in spans], dtype=np.int32).reshape(-1, 2),
        "nucleolus_bonds": np.array(bonds, dtype=np.int32).reshape(-1, 2),
    }


def refine_path_spline(path, n):
    """Interpolating cubic spline through `path` (knot k at u=(k+1/2)/len, not-a-knot ends, which is what an
    s=0 FITPACK spline is), sampled at the n bin midpoints (k+1/2)/n."""
    path = np.asarray(path, dtype=float)
    u = (np.arange(len(path)) + 0.5) / len(path)
    fine_u = (np.arange(n) + 0.5) / n
    return CubicSpline(u, path, axis=0, bc_type="not-a-knot", extrapolate=True)(fine_u)


def refine_positions(coarse_positions, coarse_ranges, fine_ranges, coarse_graining, nucleolus_bonds, n_particles):
    """Initial 100 kb conformation from the packed coarse one: every chain is upsampled x coarse_graining and
    truncated to its fine length; nucleolar beads start on top of their NOR."""
    fine = np.empty((n_particles, 3))
    for (cb, ce), (fb, fe) in zip(coarse_ranges, fine_ranges):
        chain = refine_path_spline(coarse_positions[cb:ce], (ce - cb) * coarse_graining)
        fine[fb:fe] = chain[:fe - fb]
    for nor, nuc in nucleolus_bonds:
        fine[nuc] = fine[nor]
    return fine
