"""ctypes binding of oracle/kinetics_probe.cpp (either build) and the scenarios shared by the fixture generator
and the tests."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_LIB = os.path.join(ROOT, "oracle", "_ref", "libref1kb.so")
ST = C.c_size_t


def build_product_probe(outdir):
    """The probe over the product header 2022a-genome-dynamics_amd/host/gd_1kb_kinetics.hpp."""
    out = os.path.join(str(outdir), "libgd1kb_probe.so")
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-fPIC", "-shared", "-Wall", "-Wextra", "-Werror",
                           "-I" + os.path.join(ROOT, "2022a-genome-dynamics_amd", "host"), "-o", out,
                           os.path.join(ROOT, "oracle", "kinetics_probe.cpp")])
    return out


class Probe:
    def __init__(self, path):
        self.dll = C.CDLL(path)
        self.dll.probe_loops.restype = C.c_int
        self.dll.probe_loops.argtypes = [ST, ST, C.c_double, C.c_double, C.c_double, C.c_double, C.c_double, C.c_void_p, ST, C.c_void_p,
                                         C.c_void_p, ST, C.c_void_p, C.c_void_p, ST, C.c_void_p, ST, C.c_int, C.c_uint64, C.c_int,
                                         C.c_double, C.c_int, C.c_void_p, C.POINTER(C.c_uint64)]
        self.dll.probe_reservoir.restype = C.c_int
        self.dll.probe_reservoir.argtypes = [ST, ST, C.c_uint64, C.c_void_p, C.POINTER(ST), C.POINTER(C.c_uint64)]

    def loops(self, sc):
        def sz(v):
            return np.ascontiguousarray(v, dtype=np.uint64)

        def fl(v):
            return np.ascontiguousarray(v, dtype=np.float64)
        b, ap, av = sz(sc.get("boundaries", [])), sz([p for p, _ in sc.get("attach", [])]), fl([v for _, v in sc.get("attach", [])])
        dp, dv, hc = sz([p for p, _ in sc.get("detach", [])]), fl([v for _, v in sc.get("detach", [])]), sz(sc.get("handcuffs", []))
        out = np.zeros((sc["steps"] + 1, sc["max_loops"], 3), dtype=np.int64)
        nxt = C.c_uint64()
        rc = self.dll.probe_loops(sc["length"], sc["max_loops"], sc.get("loading", 0.0), sc.get("unloading", 0.0), sc.get("forward", 0.0),
                                  sc.get("backward", 0.0), sc.get("crossing", float("nan")), b.ctypes.data, len(b), ap.ctypes.data,
                                  av.ctypes.data, len(ap), dp.ctypes.data, dv.ctypes.data, len(dp), hc.ctypes.data, len(hc),
                                  int(sc.get("preload", False)), sc["seed"], sc["steps"], sc["dt"], sc.get("clear_at", -1),
                                  out.ctypes.data, C.byref(nxt))
        assert rc == 0
        return out, nxt.value

    def reservoir(self, capacity, n_items, seed):
        out = np.zeros(max(capacity, 1), dtype=np.uint64)
        n, nxt = ST(), C.c_uint64()
        assert self.dll.probe_reservoir(capacity, n_items, seed, out.ctypes.data, C.byref(n), C.byref(nxt)) == 0
        return out[:n.value].copy(), nxt.value


# scenarios cover: plain extrusion with loading/unloading, boundaries with convergent detachability, roadblocks,
# finite and infinite crossing rates, handcuff loops, preloading, clearing, a large Poisson mean (normal-approximation
# branch of libstdc++'s sampler) and saturation of the loop slots
LOOP_SCENARIOS = {
    "plain": dict(length=400, max_loops=12, loading=3.0, unloading=0.5, forward=40.0, backward=2.0, seed=1, steps=60, dt=0.02),
    "boundaries": dict(length=300, max_loops=10, loading=4.0, unloading=0.3, forward=60.0, backward=5.0, seed=2, steps=80, dt=0.02,
                       boundaries=[50, 120, 121, 250], detach=[(51, 0.1), (119, 0.1), (122, 0.05), (249, 0.2)]),
    "roadblocks_crossing": dict(length=200, max_loops=16, loading=10.0, unloading=0.2, forward=50.0, backward=10.0, crossing=8.0, seed=3,
                                steps=80, dt=0.02, attach=[(60, 0.2), (61, 0.2), (140, 0.0)], boundaries=[100]),
    "free_crossing": dict(length=150, max_loops=8, loading=6.0, unloading=0.4, forward=30.0, backward=30.0, crossing=float("inf"), seed=4,
                          steps=50, dt=0.03),
    "handcuffs_preload": dict(length=500, max_loops=40, loading=20.0, unloading=1.0, forward=80.0, backward=1.0, crossing=2.0, seed=5,
                              steps=40, dt=0.01, handcuffs=[10, 10, 250, 499, 0], preload=True, boundaries=[300], detach=[(301, 0.1)]),
    "clear": dict(length=120, max_loops=6, loading=15.0, unloading=0.1, forward=20.0, backward=0.0, seed=6, steps=30, dt=0.05, clear_at=20),
    "dense_poisson": dict(length=5000, max_loops=64, loading=2000.0, unloading=3.0, forward=10.0, backward=1.0, crossing=30.0, seed=7,
                          steps=15, dt=0.05),
    "static": dict(length=50, max_loops=3, forward=0.0, backward=0.0, seed=8, steps=5, dt=0.1, handcuffs=[5, 20]),
}
RESERVOIR_CASES = [(5, 3, 11), (5, 5, 12), (5, 6, 13), (8, 1000, 14), (1, 50, 15), (64, 5000, 16), (3, 0, 17)]
