"""The stage-4 drivers above the C-ABI: `gd_ab_box` (periodic A/B blend, cfg1 of SURVEY.md 8d) and
`gd_ab_sphere` (confined blend).  As in test_host_driver.py the driver source is backend-agnostic: the CPU tests
link it against the oracle library, the GPU tests against libgdyn, and both are compared with the same sequence
of ABI calls issued from Python (including the reference's rod initialisation, drawn from std::mt19937_64
through libstdc++'s distributions)."""
import json
import math
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT
from test_host_driver import HOST, _env, _make, _make_oracle, _tool
from util import Mt64, g

pytestmark = pytest.mark.skipif(not os.path.exists("/opt/conda/include/hdf5.h"), reason="HDF5 C library not in this image")

CHAINS, LEN, STEPS = 20, 12, 60
N = CHAINS * LEN


def _inputs(tmp, geo):
    rows = ["chain\tA\tB"]
    for c in range(CHAINS):            # alternating pure-A / pure-B chains, one mixed chain
        ab = (1, 0) if c % 2 == 0 else (0, 1)
        if c == 7:
            ab = (0.5, 0.5)
        rows += [f"chain{c}\t{ab[0]}\t{ab[1]}"] * LEN
    (tmp / "beads.tsv").write_text("\n".join(rows) + "\n")
    cfg = dict(beads_filename=str(tmp / "beads.tsv"), init_bond_length=0.1, steps=STEPS, logging_interval=10, sampling_interval=20,
               seed=5, bond_spring=70.0)
    if geo == "box":
        cfg.update(box_size=2.0)
    elif geo == "sphere":
        cfg.update(outer_wall_radius=1.2, outer_wall_multiplier=2.0, outer_wall_spring=50.0)
    else:       # "shell": sphere with the excluded core of sphere/src/simulation_driver.cc:184-228
        cfg.update(outer_wall_radius=1.3, outer_wall_multiplier=2.0, outer_wall_spring=50.0, inner_wall_radius=0.5,
                   inner_wall_multiplier=1.5, inner_wall_spring=40.0)
    (tmp / "config.json").write_text(json.dumps(cfg))
    return cfg


def _python_ab(lib, oracle, cfg, geo, seed):
    d = dict(a_core_diameter=0.30, b_core_diameter=0.24, a_core_repulsion=2.0, b_core_repulsion=2.0, mobility=1.0, temperature=1.0,
             timestep=1e-5)
    d.update(cfg)
    a = np.zeros(N)
    b = np.zeros(N)
    for c in range(CHAINS):
        va, vb = ((1, 0) if c % 2 == 0 else (0, 1)) if c != 7 else (0.5, 0.5)
        a[c * LEN:(c + 1) * LEN], b[c * LEN:(c + 1) * LEN] = va, vb
    rnd = Mt64(oracle, seed)
    x = np.zeros((N, 3))
    for c in range(CHAINS):            # 4-sim-ab/box/src/simulation/simulation_driver.cc:151-180
        if geo == "box":
            center = np.array([rnd.uniform(0, d["box_size"]) for _ in range(3)])
        else:
            while True:     # rejection of centres inside the core (sphere/src/simulation_driver.cc:246-251)
                center = np.array([rnd.uniform(-d["outer_wall_radius"], d["outer_wall_radius"]) for _ in range(3)])
                if not math.sqrt(center[0] * center[0] + center[1] * center[1] + center[2] * center[2]) < d.get("inner_wall_radius", 0.0):
                    break
        z = rnd.normals(3)
        inv = 1 / math.sqrt(z[0] * z[0] + z[1] * z[1] + z[2] * z[2])
        direction = np.array([z[0] * inv, z[1] * inv, z[2] * inv])
        delta, pos = np.zeros(3), np.zeros(3)
        for i in range(c * LEN, (c + 1) * LEN):
            x[i] = pos
            delta = delta + (pos - center)
            pos = pos + d["init_bond_length"] * direction
        delta = delta / float(LEN)
        x[c * LEN:(c + 1) * LEN] -= delta
    s = g.System(lib, N, 1, box=(d["box_size"],) * 3 if geo == "box" else None)
    s.set_bead_params(a=a, b=b, mobility=np.full(N, d["mobility"]))
    s.set_pair_softcore(d["a_core_repulsion"], d["a_core_diameter"], d["b_core_repulsion"], d["b_core_diameter"], 2, 3, 8, 3, mix=True)
    bond = g.System.bond_params(g.POT_HARMONIC, k_a=d["bond_spring"])
    for c in range(CHAINS):
        s.add_bond_range(bond, c * LEN, (c + 1) * LEN, 1)
    if geo != "box":
        m = d["outer_wall_multiplier"]
        s.set_ellipsoid_wall(m * d["a_core_repulsion"], d["a_core_diameter"], m * d["b_core_repulsion"], d["b_core_diameter"], 0.0, 1.0,
                             d["outer_wall_spring"], (0.0, 0.0, 0.0), 0.0, (d["outer_wall_radius"],) * 3, scale_by_bead_scale=False)
        if d.get("inner_wall_radius", 0.0) >= 1e-6:
            mi = d["inner_wall_multiplier"]
            s.set_inner_sphere_wall(d["inner_wall_radius"], mi * d["a_core_repulsion"], d["a_core_diameter"], mi * d["b_core_repulsion"],
                                    d["b_core_diameter"], 0.0, 1.0, d["inner_wall_spring"])
    s.set_positions(x)
    s.begin_phase()
    pos, energy = {0: s.positions()[0].copy()}, {0: float(s.energy()[0]) / N}
    step = 0
    while step < STEPS:
        s.run(10, d["timestep"], d["temperature"], seed=0)
        step += 10
        energy[step] = float(s.energy()[0]) / N
        if step % 20 == 0:
            pos[step] = s.positions()[0].copy()
    s.close()
    return x, pos, energy


def _dataset(tmp, path):
    shape = [int(v) for v in _tool("dataset", tmp / "out.h5", path, tmp / "ds.f64").split()]
    return np.fromfile(tmp / "ds.f64", dtype="<f8").reshape(shape)


def _check(tmp, lib, oracle, driver, geo, atol, env=None, seed_arg=None):
    cfg = _inputs(tmp, geo)
    args = [str(driver)] + (["-s", str(seed_arg)] if seed_arg is not None else []) + [str(tmp / "config.json"), str(tmp / "out.h5")]
    r = subprocess.run(args, capture_output=True, text=True, env=env)
    assert r.returncode == 0, r.stderr
    lines = r.stderr.splitlines()
    assert lines[0] == "[sim] sampling..."
    logs = [ln.split("\t") for ln in lines[1:] if ln.startswith("[sim] ")]
    assert [int(f[1]) for f in logs] == list(range(0, STEPS + 1, 10))
    seed = cfg["seed"] if seed_arg is None else seed_arg
    x0, pos, energy = _python_ab(lib, oracle, cfg, geo, seed)
    for f in logs:
        assert f[2].startswith("E: ")
        assert float(f[2][3:]) == pytest.approx(energy[int(f[1])], rel=2e-5 if atol == 0 else 2e-2)
    # file layout of 4-sim-ab/box/src/simulation/simulation_store.cc:20-76
    saved = json.loads(_tool("strings", tmp / "out.h5", "/metadata/config"))
    assert saved["seed"] == seed and saved["mobility"] == 1.0 and saved["beads_filename"] == cfg["beads_filename"]
    assert ("box_size" in saved) == (geo == "box") and ("outer_wall_radius" in saved) == (geo != "box")
    ab = _dataset(tmp, "/metadata/ab_factors")
    assert ab.shape == (N, 2) and ab[0].tolist() == [1, 0] and ab[LEN].tolist() == [0, 1] and ab[7 * LEN].tolist() == [0.5, 0.5]
    assert _dataset(tmp, "/metadata/chain_ranges").tolist() == [[c * LEN, (c + 1) * LEN] for c in range(CHAINS)]
    assert _tool("strings", tmp / "out.h5", "/snapshots/.steps").split() == ["0", "20", "40", "60"]
    for step, want in pos.items():
        got = _dataset(tmp, f"/snapshots/{step}/positions")
        assert got.shape == (N, 3)
        assert np.abs(got - want).max() <= 1.1e-3 + atol, step       # the file keeps 3 decimal digits (scale-offset filter)
    # rods: every chain is straight with the configured bond length
    d = np.diff(x0.reshape(CHAINS, LEN, 3), axis=1)
    assert np.allclose(np.linalg.norm(d, axis=2), cfg["init_bond_length"], atol=1e-12)
    if os.path.exists("/opt/conda/bin/h5dump"):
        hdr = subprocess.check_output(["/opt/conda/bin/h5dump", "-H", "-p", str(tmp / "out.h5")], text=True)
        p = hdr[hdr.index('DATASET "positions"'):][:900]
        assert "H5T_IEEE_F32LE" in p and "SCALEOFFSET" in p and "DEFLATE { LEVEL 1 }" in p


@pytest.mark.parametrize("geo", ["box", "sphere", "shell"])
def test_ab_driver_on_oracle(tmp_path, oracle, geo):
    drv = _make_oracle("gd_ab_box" if geo == "box" else "gd_ab_sphere", tmp_path)
    _check(tmp_path, oracle, oracle, drv, geo, atol=0, env=_env(os.path.join(ROOT, "oracle")), seed_arg=11 if geo == "box" else None)


def test_ab_driver_errors(tmp_path, oracle):
    drv = _make_oracle("gd_ab_box", tmp_path)
    env = _env(os.path.join(ROOT, "oracle"))
    cfg = _inputs(tmp_path, "box")
    (tmp_path / "beads.tsv").write_text("chain A B\nc 1 0\n")                 # simulation_data.cc:14-20
    r = subprocess.run([str(drv), str(tmp_path / "config.json"), str(tmp_path / "out.h5")], capture_output=True, text=True, env=env)
    assert r.returncode == 1 and "unexpected beads data header" in r.stderr
    r = subprocess.run([str(drv), str(tmp_path / "missing.json"), str(tmp_path / "out.h5")], capture_output=True, text=True, env=env)
    assert r.returncode == 1 and "cannot open config file" in r.stderr       # main.cc:50-54
    assert subprocess.run([str(drv)], capture_output=True, text=True, env=env).returncode == 1


@pytest.mark.gpu
@pytest.mark.parametrize("geo", ["box", "sphere", "shell"])
def test_ab_driver_on_gpu(tmp_path, hip, oracle, geo):
    _check(tmp_path, oracle, oracle, _make("gd_ab_box" if geo == "box" else "gd_ab_sphere", ".", "../csrc", "gdyn"), geo, atol=2e-4)
