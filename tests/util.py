"""Shared helpers for the test-suite: small instances of the four reference configurations."""
import importlib

import numpy as np

PKG = "2022a-genome-dynamics_amd"
g = importlib.import_module(PKG)
wl = importlib.import_module(PKG + ".workloads")

# name -> (builder, kwargs, timestep, temperature, run flags)
CASES = {
    "genome": (wl.genome_interphase, dict(n_beads=1500, bead_scale_init=0.8), 1e-5, 1.0,
               g.RUN_UPDATE_SCALES | g.RUN_WALL_DYNAMICS),
    "spindle": (wl.spindle, dict(n_beads=300), 1e-4, 0.1, 0),
    "ab_box": (wl.ab_box, dict(n_chains=40, chain_len=20, box=2.9), 1e-5, 1.0, 0),
    "chromatin_1kb": (wl.chromatin_1kb, dict(n_beads=3000, n_loops=30, n_glues=60), 1e-4, 1.0, 0),
}
TERMS = {"pair": g.TERM_PAIR, "bond": g.TERM_BOND, "bend": g.TERM_BEND, "point": g.TERM_POINT,
         "wall": g.TERM_WALL, "dynamic": g.TERM_DYNAMIC, "all": g.TERM_ALL}

# fp32 tolerances of the device path against the fp64 oracle (stated in DESIGN.md)
FORCE_RTOL = 5e-5      # |dF| <= FORCE_RTOL * max|F_all|   (per-term forces are compared on the all-terms scale)
ENERGY_RTOL = 2e-6     # |dE| <= ENERGY_RTOL * sum of |term energies|
POS_ATOL_1STEP = 2e-6  # |dx| after one step
POS_ATOL_20STEP = 3e-5 # |dx| after 20 steps (trajectories of dense soft-sphere systems diverge exponentially)


def build(lib, name, **over):
    builder, kw, dt, kT, flags = CASES[name]
    kw = dict(kw)
    kw.update(over)
    s, info = builder(lib, **kw)
    return s, dt, kT, flags


def force_scale(sys_):
    return float(np.abs(sys_.forces(g.TERM_ALL)).max())


# ---- std::mt19937_64 and libstdc++ distributions, for the drivers' initial conditions
import ctypes as C


def mt64(oracle, seed, n):
    f = oracle.dll.oracle_mt64_nth
    f.restype = C.c_uint64
    f.argtypes = [C.c_uint64, C.c_int]
    return f(seed, n)



class Mt64:
    """std::mt19937_64 draws through the oracle's generator, plus libstdc++'s std::normal_distribution<double>
    (Marsaglia polar method on generate_canonical<double, 53>: one 64-bit draw per uniform) -- what
    simulation_spindle/simulation_driver.cc:189-199 consumes from `_random`."""

    def __init__(self, oracle, seed):
        self.oracle, self.seed, self.n = oracle, seed, 0

    def draw(self):
        self.n += 1
        return mt64(self.oracle, self.seed, self.n)

    def canonical(self):
        r = float(self.draw()) / 18446744073709551616.0
        return r if r < 1.0 else float(np.nextafter(1.0, 0.0))

    def uniform(self, a, b):
        """One draw of a std::uniform_real_distribution<double>{a, b}."""
        return self.canonical() * (b - a) + a

    def normals(self, count):
        """`count` values from a freshly constructed distribution object."""
        import math
        out, saved = [], None
        while len(out) < count:
            if saved is not None:
                out.append(saved)
                saved = None
                continue
            while True:
                x = 2.0 * self.canonical() - 1.0
                y = 2.0 * self.canonical() - 1.0
                r2 = x * x + y * y
                if not (r2 > 1.0 or r2 == 0.0):
                    break
            mult = math.sqrt(-2 * math.log(r2) / r2)
            saved = x * mult
            out.append(y * mult)
        return out


