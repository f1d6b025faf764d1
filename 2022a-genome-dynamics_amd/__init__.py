"""gdyn -- MI355X-native Brownian-dynamics stepper for bead-spring chromatin polymers.

Python host side of the C-ABI in ``include/gdyn.h``.  The product library is
``csrc/libgdyn.so`` (hand-written HIP for gfx950); :func:`load` fails loudly when it
is missing -- there is no CPU fallback.  :class:`Lib` is a plain ctypes binding of the
ABI and works for any shared object exporting it (tests bind the CPU oracle with it).

The class :class:`System` mirrors the part of micromd's ``md::system`` /
``md::simulate_brownian_dynamics`` interface that the reference drivers use
(e.g. 5-sim-genome/src/simulation_interphase/simulation_driver_forcefield.cc:19-235,
simulation_driver_interphase.cc:48-55).
"""
from __future__ import annotations

import ctypes as C
import os
from dataclasses import dataclass

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIBGDYN_PATH = os.path.join(_HERE, "csrc", "libgdyn.so")      # the product library; the package reads no environment variable

GD_BOX_OPEN, GD_BOX_PERIODIC = 0, 1
POT_HARMONIC, POT_SPRING, POT_SEMISPRING, POT_SOFTCORE = 0, 1, 2, 3
NOISE_PHILOX, NOISE_ZERO, NOISE_HOST, NOISE_MT19937 = 0, 1, 2, 3
RUN_UPDATE_SCALES, RUN_WALL_DYNAMICS, RUN_DEFER_CALLBACK, RUN_COMPENSATED, RUN_UNCOMPENSATED = 1, 2, 4, 8, 16
ALL_REPLICAS = 0xffffffff
ABI_VERSION = 5            # GD_ABI_VERSION of the include/gdyn.h these ctypes structures mirror
TERM_PAIR, TERM_BOND, TERM_BEND, TERM_POINT, TERM_WALL, TERM_DYNAMIC, TERM_ALL = 1, 2, 4, 8, 16, 32, 63

_STATUS = {1: "GD_EINVAL", 2: "GD_ENODEVICE", 3: "GD_EHIP", 4: "GD_ENOMEM", 5: "GD_ESTATE", 6: "GD_EUNSUPPORTED"}


class GdynError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"{_STATUS.get(code, code)}: {msg}")
        self.code = code


class _Desc(C.Structure):
    _fields_ = [("n_beads", C.c_uint32), ("n_replicas", C.c_uint32), ("device", C.c_int32),
                ("box_kind", C.c_int32), ("box", C.c_double * 3)]


class PairSoftcore(C.Structure):
    _fields_ = [("eps_a", C.c_double), ("sigma_a", C.c_double), ("eps_b", C.c_double), ("sigma_b", C.c_double),
                ("p_a", C.c_int32), ("q_a", C.c_int32), ("p_b", C.c_int32), ("q_b", C.c_int32),
                ("mix", C.c_int32), ("scale_by_bead_scale", C.c_int32)]


class BondParams(C.Structure):
    _fields_ = [("kind", C.c_int32), ("mix", C.c_int32), ("k_a", C.c_double), ("k_b", C.c_double),
                ("l_a", C.c_double), ("l_b", C.c_double), ("scale_by_bond_scale", C.c_int32),
                ("p", C.c_int32), ("q", C.c_int32), ("minimum_image", C.c_int32)]


class Wall(C.Structure):
    _fields_ = [("eps_a", C.c_double), ("sigma_a", C.c_double), ("eps_b", C.c_double), ("sigma_b", C.c_double),
                ("p_a", C.c_int32), ("q_a", C.c_int32), ("p_b", C.c_int32), ("q_b", C.c_int32),
                ("wall_a_factor", C.c_double), ("wall_b_factor", C.c_double),
                ("scale_by_bead_scale", C.c_int32), ("packing_spring", C.c_double),
                ("semiaxes_spring", C.c_double * 3), ("mobility", C.c_double), ("init_semiaxes", C.c_double * 3)]


class Context(C.Structure):
    _fields_ = [("step", C.c_int64), ("time", C.c_double), ("bead_scale", C.c_double), ("bond_scale", C.c_double),
                ("semiaxes", C.c_double * 3), ("axial_reaction", C.c_double * 3),
                ("list_entries", C.c_uint64), ("rebuilds", C.c_uint64), ("rollbacks", C.c_uint64),
                ("rebuild_interval", C.c_uint32), ("list_radius", C.c_double), ("list_path", C.c_uint32),
                ("callback_pending", C.c_uint32), ("tile_capacity", C.c_uint32), ("compensated", C.c_uint32), ("largest_tile", C.c_uint32),
                ("row_repairs", C.c_uint32), ("near_entries", C.c_uint64), ("list_bytes", C.c_uint64)]


class _RunDesc(C.Structure):
    _fields_ = [("temperature", C.c_double), ("timestep", C.c_double), ("spacestep", C.c_double),
                ("steps", C.c_int64), ("seed", C.c_uint64), ("noise_mode", C.c_int32), ("flags", C.c_int32),
                ("host_noise", C.POINTER(C.c_double)), ("replica_seeds", C.POINTER(C.c_uint64))]


class Tuning(C.Structure):
    _fields_ = [("skin", C.c_double), ("rebuild_interval", C.c_uint32), ("adapt_interval", C.c_uint32),
                ("list_width", C.c_uint32), ("kernel_path", C.c_uint32), ("near_fraction", C.c_double),
                ("auto_skin", C.c_uint32)]


class Timing(C.Structure):
    _fields_ = [("step_kernel_ms", C.c_double), ("rebuild_ms", C.c_double), ("total_ms", C.c_double),
                ("step_launches", C.c_uint64), ("rebuild_launches", C.c_uint64),
                ("list_entries_visited", C.c_uint64)]


# every symbol include/gdyn.h declares
class InnerSphere(C.Structure):
    _fields_ = [("radius", C.c_double), ("eps_a", C.c_double), ("sigma_a", C.c_double), ("eps_b", C.c_double), ("sigma_b", C.c_double),
                ("p_a", C.c_int32), ("q_a", C.c_int32), ("p_b", C.c_int32), ("q_b", C.c_int32),
                ("wall_a_factor", C.c_double), ("wall_b_factor", C.c_double), ("spring", C.c_double)]


ABI_SYMBOLS = [
    "gd_last_error", "gd_backend_name", "gd_abi_version", "gd_create_abi", "gd_destroy", "gd_set_positions", "gd_get_positions",
    "gd_get_positions_f32", "gd_set_bead_params", "gd_set_pair_softcore", "gd_add_bond_range",
    "gd_add_bond_pairs", "gd_set_dynamic_pairs", "gd_add_bending_range", "gd_add_point_source",
    "gd_set_ellipsoid_wall", "gd_set_inner_sphere_wall", "gd_set_pair_softwell", "gd_set_scaling", "gd_get_context", "gd_begin_phase", "gd_set_context",
    "gd_run", "gd_apply_callback", "gd_compute_energy", "gd_compute_forces", "gd_search_pairs", "gd_set_tuning",
    "gd_get_timing", "gd_get_stream", "gd_contacts_update", "gd_contacts_fetch", "gd_contacts_clear",
]


def _dptr(a):
    return None if a is None else a.ctypes.data_as(C.POINTER(C.c_double))


def _uptr(a):
    return None if a is None else a.ctypes.data_as(C.POINTER(C.c_uint32))


class Lib:
    """ctypes binding of one shared object exporting the gdyn.h ABI."""

    def __init__(self, path):
        if not os.path.exists(path):
            raise FileNotFoundError(f"gdyn shared library not found: {path}")
        self.path = path
        self.dll = C.CDLL(path, mode=getattr(os, "RTLD_LOCAL", 0) | getattr(os, "RTLD_NOW", 2))
        for name in ABI_SYMBOLS:
            if not hasattr(self.dll, name):
                raise OSError(f"{path}: missing ABI symbol {name}")
        d = self.dll
        d.gd_last_error.restype = C.c_char_p
        d.gd_backend_name.restype = C.c_char_p
        d.gd_abi_version.restype = C.c_int
        if d.gd_abi_version() != ABI_VERSION:      # the ctypes structures below mirror ONE version of include/gdyn.h
            raise OSError(f"{path}: ABI version {d.gd_abi_version()}, this binding mirrors version {ABI_VERSION} of include/gdyn.h")
        d.gd_create_abi.argtypes = [C.c_int, C.POINTER(_Desc), C.POINTER(C.c_void_p)]
        d.gd_destroy.argtypes = [C.c_void_p]
        d.gd_set_positions.argtypes = [C.c_void_p, C.POINTER(C.c_double)]
        d.gd_get_positions.argtypes = [C.c_void_p, C.POINTER(C.c_double)]
        d.gd_get_positions_f32.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.c_int]
        d.gd_set_bead_params.argtypes = [C.c_void_p] + [C.POINTER(C.c_double)] * 4
        d.gd_set_pair_softcore.argtypes = [C.c_void_p, C.POINTER(PairSoftcore)]
        d.gd_add_bond_range.argtypes = [C.c_void_p, C.POINTER(BondParams), C.c_uint32, C.c_uint32, C.c_uint32]
        d.gd_add_bond_pairs.argtypes = [C.c_void_p, C.POINTER(BondParams), C.POINTER(C.c_uint32), C.c_uint32]
        d.gd_set_dynamic_pairs.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(BondParams), C.POINTER(C.c_uint32), C.c_uint32]
        d.gd_add_bending_range.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_double, C.c_int]
        d.gd_add_point_source.argtypes = [C.c_void_p, C.c_int, C.c_double, C.c_double, C.POINTER(C.c_double),
                                          C.POINTER(C.c_uint32), C.c_uint32]
        d.gd_set_ellipsoid_wall.argtypes = [C.c_void_p, C.POINTER(Wall)]
        d.gd_set_inner_sphere_wall.argtypes = [C.c_void_p, C.POINTER(InnerSphere)]
        d.gd_set_pair_softwell.argtypes = [C.c_void_p, C.c_double, C.c_double, C.c_double, C.c_void_p, C.c_uint32]
        d.gd_set_scaling.argtypes = [C.c_void_p] + [C.c_double] * 4
        d.gd_get_context.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(Context)]
        d.gd_begin_phase.argtypes = [C.c_void_p, C.POINTER(C.c_double)]
        d.gd_set_context.argtypes = [C.c_void_p, C.c_uint32, C.c_int64, C.c_double, C.c_double, C.POINTER(C.c_double)]
        d.gd_run.argtypes = [C.c_void_p, C.POINTER(_RunDesc)]
        d.gd_apply_callback.argtypes = [C.c_void_p]
        d.gd_compute_energy.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(C.c_double)]
        d.gd_compute_forces.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(C.c_double)]
        d.gd_search_pairs.argtypes = [C.c_void_p, C.c_uint32, C.c_double, C.POINTER(C.c_uint32), C.c_uint64,
                                      C.POINTER(C.c_uint64)]
        d.gd_contacts_update.argtypes = [C.c_void_p, C.c_double]
        d.gd_contacts_fetch.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(C.c_uint32), C.c_uint64, C.POINTER(C.c_uint64)]
        d.gd_contacts_clear.argtypes = [C.c_void_p, C.c_uint32]
        d.gd_set_tuning.argtypes = [C.c_void_p, C.POINTER(Tuning)]
        d.gd_get_timing.argtypes = [C.c_void_p, C.POINTER(Timing)]
        d.gd_get_stream.argtypes = [C.c_void_p, C.POINTER(C.c_void_p)]

    @property
    def backend(self):
        return self.dll.gd_backend_name().decode()

    def check(self, rc):
        if rc != 0:
            raise GdynError(rc, self.dll.gd_last_error().decode(errors="replace"))


_product = None


def load(path=None):
    """Load the HIP product library (or, for developer builds, the HIP library at `path`: a file name is looked up in
    csrc/).  Raises if the extension has not been built."""
    global _product
    if path:
        path = path if os.path.sep in path else os.path.join(_HERE, "csrc", path)
        lib = Lib(path)
        if lib.backend != "hip":
            raise ImportError(f"{path} reports backend {lib.backend!r}, expected 'hip'")
        return lib
    if _product is None:
        if not os.path.exists(LIBGDYN_PATH):
            raise ImportError(
                f"{LIBGDYN_PATH} is missing: build the HIP extension first "
                "(python -c 'import __graft_entry__ as g; g.build()'). There is no CPU fallback.")
        lib = Lib(LIBGDYN_PATH)
        if lib.backend != "hip":
            raise ImportError(f"{LIBGDYN_PATH} reports backend {lib.backend!r}, expected 'hip'")
        _product = lib
    return _product


@dataclass
class RunResult:
    timing: Timing


class System:
    """Host-side mirror of md::system + md::simulate_brownian_dynamics over the C-ABI."""

    def __init__(self, lib: Lib, n_beads, n_replicas=1, box=None, device=0):
        self.lib, self.N, self.R = lib, int(n_beads), int(n_replicas)
        d = _Desc(self.N, self.R, device, GD_BOX_PERIODIC if box is not None else GD_BOX_OPEN,
                  (C.c_double * 3)(*(box if box is not None else (0, 0, 0))))
        self._h = C.c_void_p()
        lib.check(lib.dll.gd_create_abi(ABI_VERSION, C.byref(d), C.byref(self._h)))

    def close(self):
        if getattr(self, "_h", None):
            self.lib.dll.gd_destroy(self._h)
            self._h = None

    __del__ = close

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    # -- md::system -----------------------------------------------------------
    def _f64(self, a, shape):
        a = np.ascontiguousarray(a, dtype=np.float64)
        if a.shape != shape:
            raise ValueError(f"expected shape {shape}, got {a.shape}")
        return a

    def set_positions(self, xyz):
        xyz = np.asarray(xyz, dtype=np.float64)
        if xyz.shape == (self.N, 3):
            xyz = np.broadcast_to(xyz, (self.R, self.N, 3))
        a = self._f64(xyz, (self.R, self.N, 3))
        self.lib.check(self.lib.dll.gd_set_positions(self._h, _dptr(a)))

    def positions(self):
        out = np.empty((self.R, self.N, 3))
        self.lib.check(self.lib.dll.gd_get_positions(self._h, _dptr(out)))
        return out

    def positions_f32(self, quantize=False):
        out = np.empty((self.R, self.N, 3), dtype=np.float32)
        self.lib.check(self.lib.dll.gd_get_positions_f32(self._h, out.ctypes.data_as(C.POINTER(C.c_float)), int(quantize)))
        return out

    def set_bead_params(self, a=None, b=None, mobility=None, bending_energy=None):
        arrs = [None if v is None else self._f64(v, (self.N,)) for v in (a, b, mobility, bending_energy)]
        self.lib.check(self.lib.dll.gd_set_bead_params(self._h, *[_dptr(v) for v in arrs]))

    # -- force fields ---------------------------------------------------------
    def set_pair_softcore(self, eps_a, sigma_a, eps_b=0.0, sigma_b=0.0, p_a=2, q_a=3, p_b=8, q_b=3, mix=True,
                          scale_by_bead_scale=False):
        p = PairSoftcore(eps_a, sigma_a, eps_b, sigma_b, p_a, q_a, p_b, q_b, int(mix), int(scale_by_bead_scale))
        self.lib.check(self.lib.dll.gd_set_pair_softcore(self._h, C.byref(p)))

    @staticmethod
    def bond_params(kind, k_a, l_a=0.0, k_b=0.0, l_b=0.0, mix=False, scale_by_bond_scale=False, p=8, q=3,
                    minimum_image=False):
        return BondParams(kind, int(mix), k_a, k_b, l_a, l_b, int(scale_by_bond_scale), p, q, int(minimum_image))

    def add_bond_range(self, params, start, end, stride=1):
        self.lib.check(self.lib.dll.gd_add_bond_range(self._h, C.byref(params), start, end, stride))

    def add_bond_pairs(self, params, pairs):
        pairs = np.ascontiguousarray(pairs, dtype=np.uint32).reshape(-1, 2)
        self.lib.check(self.lib.dll.gd_add_bond_pairs(self._h, C.byref(params), _uptr(pairs), len(pairs)))

    def set_dynamic_pairs(self, slot, params, pairs):
        pairs = np.ascontiguousarray(pairs, dtype=np.uint32).reshape(-1, 2)
        self.lib.check(self.lib.dll.gd_set_dynamic_pairs(self._h, slot, C.byref(params), _uptr(pairs), len(pairs)))

    def add_bending_range(self, start, end, energy=0.0, per_bead=False):
        self.lib.check(self.lib.dll.gd_add_bending_range(self._h, start, end, energy, int(per_bead)))

    def add_point_source(self, kind, k, b, point, targets=None):
        pt = (C.c_double * 3)(*point)
        t = None if targets is None else np.ascontiguousarray(targets, dtype=np.uint32)
        self.lib.check(self.lib.dll.gd_add_point_source(self._h, kind, k, b, pt, _uptr(t), 0 if t is None else len(t)))

    def set_ellipsoid_wall(self, eps_a, sigma_a, eps_b, sigma_b, wall_a_factor, wall_b_factor, packing_spring,
                           semiaxes_spring, mobility, init_semiaxes, p_a=2, q_a=3, p_b=8, q_b=3,
                           scale_by_bead_scale=True):
        w = Wall(eps_a, sigma_a, eps_b, sigma_b, p_a, q_a, p_b, q_b, wall_a_factor, wall_b_factor,
                 int(scale_by_bead_scale), packing_spring, (C.c_double * 3)(*semiaxes_spring), mobility,
                 (C.c_double * 3)(*init_semiaxes))
        self.lib.check(self.lib.dll.gd_set_ellipsoid_wall(self._h, C.byref(w)))

    def set_pair_softwell(self, energy, decay, cutoff, targets):
        t = np.ascontiguousarray(targets, dtype=np.uint32)
        self.lib.check(self.lib.dll.gd_set_pair_softwell(self._h, energy, decay, cutoff, t.ctypes.data if len(t) else None, len(t)))

    def set_inner_sphere_wall(self, radius, eps_a, sigma_a, eps_b, sigma_b, wall_a_factor, wall_b_factor, spring,
                              p_a=2, q_a=3, p_b=8, q_b=3):
        w = InnerSphere(radius, eps_a, sigma_a, eps_b, sigma_b, p_a, q_a, p_b, q_b, wall_a_factor, wall_b_factor, spring)
        self.lib.check(self.lib.dll.gd_set_inner_sphere_wall(self._h, C.byref(w)))

    def set_scaling(self, bead_scale_init, bead_scale_tau, bond_scale_init, bond_scale_tau):
        self.lib.check(self.lib.dll.gd_set_scaling(self._h, bead_scale_init, bead_scale_tau, bond_scale_init, bond_scale_tau))

    # -- context --------------------------------------------------------------
    def context(self, replica=0):
        c = Context()
        self.lib.check(self.lib.dll.gd_get_context(self._h, replica, C.byref(c)))
        return c

    def begin_phase(self, semiaxes=None):
        a = None
        if semiaxes is not None:
            a = np.asarray(semiaxes, dtype=np.float64)
            if a.shape == (3,):
                a = np.broadcast_to(a, (self.R, 3))
            a = self._f64(a, (self.R, 3))
        self.lib.check(self.lib.dll.gd_begin_phase(self._h, _dptr(a)))

    def set_context(self, replica, step, bead_scale, bond_scale, semiaxes=None):
        s = None if semiaxes is None else (C.c_double * 3)(*semiaxes)
        self.lib.check(self.lib.dll.gd_set_context(self._h, replica, step, bead_scale, bond_scale, s))

    # -- md::simulate_brownian_dynamics -----------------------------------------
    def run(self, steps, timestep, temperature=1.0, seed=0, noise=NOISE_PHILOX, flags=0, host_noise=None, spacestep=0.0,
            replica_seeds=None):
        hn = None
        if host_noise is not None:
            hn = self._f64(host_noise, (steps, self.R, self.N, 3))
        rs = None
        if replica_seeds is not None:
            rs = np.ascontiguousarray(replica_seeds, dtype=np.uint64)
            if rs.shape != (self.R,):
                raise ValueError(f"replica_seeds: expected shape ({self.R},)")
        rd = _RunDesc(temperature, timestep, spacestep, steps, seed, noise, flags, _dptr(hn),
                      None if rs is None else rs.ctypes.data_as(C.POINTER(C.c_uint64)))
        self.lib.check(self.lib.dll.gd_run(self._h, C.byref(rd)))
        return self.timing()

    def apply_callback(self):
        """Apply the state updates a run with RUN_DEFER_CALLBACK left pending (no-op otherwise)."""
        self.lib.check(self.lib.dll.gd_apply_callback(self._h))

    # -- observation ----------------------------------------------------------
    def energy(self, terms=TERM_ALL):
        out = np.empty(self.R)
        self.lib.check(self.lib.dll.gd_compute_energy(self._h, terms, _dptr(out)))
        return out

    def forces(self, terms=TERM_ALL):
        out = np.empty((self.R, self.N, 3))
        self.lib.check(self.lib.dll.gd_compute_forces(self._h, terms, _dptr(out)))
        return out

    def search_pairs(self, dcut, replica=0):
        n = C.c_uint64(0)
        self.lib.check(self.lib.dll.gd_search_pairs(self._h, replica, dcut, None, 0, C.byref(n)))
        out = np.empty((n.value, 2), dtype=np.uint32)
        if n.value:
            self.lib.check(self.lib.dll.gd_search_pairs(self._h, replica, dcut, _uptr(out), n.value, C.byref(n)))
        return out

    def contacts_update(self, distance):
        """contact_map::update of every replica (contact_map.cc:31-74)."""
        self.lib.check(self.lib.dll.gd_contacts_update(self._h, distance))

    def contacts(self, replica=0):
        """contact_map::accumulate (contact_map.cc:77-91): (n, 3) uint32 rows (i, j, count), row-major order."""
        n = C.c_uint64(0)
        self.lib.check(self.lib.dll.gd_contacts_fetch(self._h, replica, None, 0, C.byref(n)))
        out = np.empty((n.value, 3), dtype=np.uint32)
        if n.value:
            self.lib.check(self.lib.dll.gd_contacts_fetch(self._h, replica, _uptr(out), n.value, C.byref(n)))
        return out

    def contacts_clear(self, replica=ALL_REPLICAS):
        self.lib.check(self.lib.dll.gd_contacts_clear(self._h, replica))

    def set_tuning(self, skin=0.0, rebuild_interval=0, adapt_interval=1, list_width=0, kernel_path=0, near_fraction=0.0, auto_skin=0):
        t = Tuning(skin, rebuild_interval, adapt_interval, list_width, kernel_path, near_fraction, auto_skin)
        self.lib.check(self.lib.dll.gd_set_tuning(self._h, C.byref(t)))

    def timing(self):
        t = Timing()
        self.lib.check(self.lib.dll.gd_get_timing(self._h, C.byref(t)))
        return t

    def debug_bench(self, what, n=20):
        """Developer builds only (csrc/gdyn_dev.h, libgdyn_dev.so): kernel micro-benchmark on the current state."""
        f = getattr(self.lib.dll, "gd_debug_bench", None)
        if f is None:
            raise GdynError(6, "gd_debug_bench: not in this library (developer builds only: make -C csrc dev; load('libgdyn_dev.so'))")
        f.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_double)]
        ms = C.c_double(0)
        self.lib.check(f(self._h, what, n, C.byref(ms)))
        return ms.value

    def stream(self):
        p = C.c_void_p()
        self.lib.check(self.lib.dll.gd_get_stream(self._h, C.byref(p)))
        return p.value
