"""The stage-3 driver `gd_1kb` above the C-ABI: configuration parsing, output datasets, log
lines, and a call-by-call replay -- the driver records (--trace <dir>) its initial positions, the integrator seed and
every loop / glue list it uploads; the same ABI calls issued from Python must reproduce every saved frame.  The
kinetics that produce those lists are pinned separately against the reference's own code (test_1kb_kinetics.py)."""
import json
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT
from test_host_driver import _env, _make, _make_oracle, _tool
from util import g

pytestmark = pytest.mark.skipif(not os.path.exists("/opt/conda/include/hdf5.h"), reason="HDF5 C library not in this image")

LENGTHS = [180, 120]
N = sum(LENGTHS)
STEPS = 60


def _config(tmp):
    return {
        "sampling": {"temperature": 1.0, "timestep": 1e-4, "steps": STEPS, "loop_update_interval": 5, "glue_update_interval": 10,
                     "logging_interval": 10, "sampling_interval": 20, "random_seed": 77, "loop_preloading": True,
                     "output_filename": str(tmp / "out.h5")},
        "chain": {"box_size": 9.0, "initial_bond_length": 1.0, "repulsive_diameter": 1.0, "repulsive_energy": 2.0, "attractive_diameter": 1.5,
                  "attractive_energy": 0.2, "bond_length": 1.0, "bond_spring": 100.0, "bending_energy": 1.0},
        "loop": {"bond_spring": 20.0, "forward_speed": 4000.0, "backward_speed": 400.0, "loading_rate_density": 8.0, "unloading_rate": 30.0,
                 "convergent_detachability": 0.1, "crossing_rate": 50.0, "max_loops": 24},
        "glue": {"max_glues": 30, "glue_energy": 3.0, "glue_distance": 1.6, "glue_binding_rate": 400.0, "glue_unbinding_rate": 300.0},
        "chains": [
            {"length": LENGTHS[0], "forward_boundaries": [40], "backward_boundaries": [140], "roadblocks": [90], "loaded_loops": [60, 100],
             "blocks": [{"start": 20, "end": 50, "bending_energy": 4.0}, {"start": 100, "end": 110}]},
            {"length": LENGTHS[1], "loaded_loops": [30]},
        ],
    }


def _dataset(tmp, path):
    shape = [int(v) for v in _tool("dataset", tmp / "out.h5", path, tmp / "ds.f64").split()]
    return np.fromfile(tmp / "ds.f64", dtype="<f8").reshape(shape)


def _replay(lib, cfg, x0, trace):
    ch, sm = cfg["chain"], cfg["sampling"]
    s = g.System(lib, N, 1, box=(ch["box_size"],) * 3)
    bend = np.full(N, ch["bending_energy"])
    bend[20:50] = 4.0
    s.set_bead_params(mobility=np.ones(N), bending_energy=bend)
    s.set_pair_softcore(ch["repulsive_energy"], ch["repulsive_diameter"], -ch["attractive_energy"], ch["attractive_diameter"], 2, 3, 8, 3, mix=False)
    st = 0
    for n in LENGTHS:
        s.add_bond_range(g.System.bond_params(g.POT_SPRING, k_a=ch["bond_spring"], l_a=ch["bond_length"]), st, st + n, 1)
        s.add_bending_range(st, st + n, 0.0, per_bead=True)
        st += n
    loop_bond = g.System.bond_params(g.POT_SPRING, k_a=cfg["loop"]["bond_spring"], l_a=ch["repulsive_diameter"])
    glue_bond = g.System.bond_params(g.POT_SOFTCORE, k_a=-cfg["glue"]["glue_energy"], l_a=cfg["glue"]["glue_distance"], p=8, q=3, minimum_image=True)
    s.set_positions(x0)
    s.begin_phase()
    events = {}
    seed = None
    for ln in trace:
        f = ln.split()
        if f[0] == "seed":
            seed = int(f[1])
        else:
            pairs = np.array(f[3:], dtype=np.uint32).reshape(-1, 2)
            assert len(pairs) == int(f[2])
            events.setdefault(int(f[1]), []).append((f[0], pairs))

    def apply(step):
        for what, pairs in events.get(step, []):
            s.set_dynamic_pairs(0 if what == "loops" else 1, loop_bond if what == "loops" else glue_bond, pairs)
    apply(-1)
    frames, energies = [], {}
    step = 0
    while True:
        if step % 10 == 0:
            energies[step] = float(s.energy()[0]) / N
        if step % 20 == 0:
            frames.append(s.positions()[0].astype(np.float32))
        apply(step)
        if step >= STEPS:
            break
        s.run(5, sm["timestep"], sm["temperature"], seed=seed)
        step += 5
    s.close()
    return frames, energies, events


def _check(tmp, lib, driver, atol, env=None):
    cfg = _config(tmp)
    (tmp / "config.json").write_text(json.dumps(cfg, indent=1))
    (tmp / "trace").mkdir()
    r = subprocess.run([str(driver), "--trace", str(tmp / "trace"), str(tmp / "config.json")], capture_output=True, text=True, env=env)
    assert r.returncode == 0, r.stderr
    logs = [ln.split("\t") for ln in r.stderr.splitlines()]
    assert [int(f[0]) for f in logs] == list(range(0, STEPS + 1, 10))
    assert all(f[1].startswith("E: ") and f[2].startswith("L: ") and f[3].startswith("G: ") for f in logs)
    x0 = np.fromfile(tmp / "trace" / "init.f64", dtype="<f8").reshape(N, 3)
    trace = (tmp / "trace" / "trace.txt").read_text().splitlines()
    # initial condition: one continuous walk with unit bonds inside each chain, centroids inside the box
    st = 0
    for n in LENGTHS:
        seg = x0[st:st + n]
        assert np.allclose(np.linalg.norm(np.diff(seg, axis=0), axis=1), 1.0, atol=1e-12)
        assert ((seg.mean(axis=0) >= 0) & (seg.mean(axis=0) <= cfg["chain"]["box_size"])).all()
        st += n
    # output datasets (3-sim-1kb/src/simulation/store.cpp:17-58)
    assert _tool("strings", tmp / "out.h5", "/config_source").strip() == json.dumps(cfg, indent=1).strip()
    eff = json.loads(_tool("strings", tmp / "out.h5", "/config"))
    assert eff["chain"]["monomer_mobility"] == 1.0 and eff["loop"]["max_loops"] == 24 and eff["sampling"]["random_seed"] == 77
    assert _dataset(tmp, "/chain_ranges").tolist() == [[0, 180], [180, 300]]
    pos = _dataset(tmp, "/positions_history")
    loops = _dataset(tmp, "/loops_history").astype(np.int64)
    assert pos.shape == (STEPS // 20 + 1, N, 3) and loops.shape == (STEPS // 20 + 1, 24, 3)
    # replay through the ABI
    frames, energies, events = _replay(lib, cfg, x0, trace)
    for k, want in enumerate(frames):
        assert np.abs(pos[k] - want.astype(np.float64)).max() <= atol, k
    # the saved loop records agree with what was uploaded (zero-length loops are not uploaded: no force)
    kb2 = 0.5 * cfg["loop"]["bond_spring"] * cfg["chain"]["repulsive_diameter"] ** 2
    for k in range(loops.shape[0]):
        step = 20 * k
        active = loops[k][loops[k][:, 2] > 0][:, :2]
        uploaded = [p for w, p in events.get(step - 5 if step else -1, []) if w == "loops"][-1]
        assert np.array_equal(active[active[:, 0] != active[:, 1]], uploaded.astype(np.int64))
        assert (loops[k][loops[k][:, 2] == 0][:, :2] == N).all()
        f = [f for f in logs if int(f[0]) == step][0]
        assert float(f[2][3:]) == pytest.approx(len(active) / N, rel=1e-5)
        zero = int((active[:, 0] == active[:, 1]).sum())
        assert float(f[1][3:]) == pytest.approx(energies[step] + zero * kb2 / N, rel=2e-5 if atol == 0 else 2e-2, abs=1e-6 if atol == 0 else 1e-3)
    glue_counts = [len(p) for st_ in sorted(events) for w, p in events[st_] if w == "glues"]
    assert max(glue_counts) > 0 and max(glue_counts) <= 30                       # glues formed, capacity respected
    assert len({tuple(map(tuple, p)) for st_ in events for w, p in events[st_] if w == "glues"}) > 2    # and turned over
    loop_sets = [p for st_ in sorted(events) for w, p in events[st_] if w == "loops"]
    assert any(not np.array_equal(a, b) for a, b in zip(loop_sets, loop_sets[1:]))                     # loops extruded
    for p in loop_sets:                                                           # no foot on a boundary site
        assert not np.isin(p, [40, 140]).any()
    return r


def test_1kb_driver_on_oracle(tmp_path, oracle):
    drv = _make_oracle("gd_1kb", tmp_path)
    env = _env(os.path.join(ROOT, "oracle"))
    r1 = _check(tmp_path, oracle, drv, atol=0, env=env)
    # deterministic for a fixed seed; -s changes the trajectory; -o / -C override the configuration
    first = _dataset(tmp_path, "/positions_history")
    r2 = subprocess.run([str(drv), str(tmp_path / "config.json")], capture_output=True, text=True, env=env)
    strip = lambda t: [ln for ln in t.splitlines()]
    assert strip(r1.stderr) == strip(r2.stderr) and np.array_equal(first, _dataset(tmp_path, "/positions_history"))
    (tmp_path / "chains.json").write_text(json.dumps([{"length": 50}]))
    plain = _config(tmp_path)
    del plain["loop"], plain["glue"]                                               # both objects are optional (config.cpp:111-123)
    plain["sampling"]["loop_preloading"] = False                                   # (preloading divides loading by unloading rate)
    (tmp_path / "plain.json").write_text(json.dumps(plain))
    r3 = subprocess.run([str(drv), "-s", "5", "-o", str(tmp_path / "other.h5"), "-C", str(tmp_path / "chains.json"), str(tmp_path / "plain.json")],
                        capture_output=True, text=True, env=env)
    assert r3.returncode == 0, r3.stderr
    shape = _tool("dataset", tmp_path / "other.h5", "/positions_history", tmp_path / "o.f64").split()
    assert shape == [str(STEPS // 20 + 1), "50", "3"]
    hdr = subprocess.check_output(["/opt/conda/bin/h5dump", "-H", str(tmp_path / "other.h5")], text=True)
    assert "loops_history" not in hdr                                            # no loop slots -> no dataset (store.cpp:43-48)


def test_1kb_driver_errors(tmp_path, oracle):
    drv = _make_oracle("gd_1kb", tmp_path)
    env = _env(os.path.join(ROOT, "oracle"))
    run = lambda *a: subprocess.run([str(drv), *map(str, a)], capture_output=True, text=True, env=env)
    r = run()
    assert r.returncode == 1 and "config file is not specified" in r.stderr           # main.cpp:125-127
    r = run(tmp_path / "missing.json")
    assert r.returncode == 1 and "failed to load config file" in r.stderr             # main.cpp:180-184
    cfg = _config(tmp_path)
    del cfg["sampling"]["timestep"]
    (tmp_path / "bad.json").write_text(json.dumps(cfg))
    r = run(tmp_path / "bad.json")
    assert r.returncode == 1 and "failed to parse config file - " in r.stderr and "timestep" in r.stderr
    cfg = _config(tmp_path)
    del cfg["glue"]["glue_energy"]                                                   # all glue members are required (config.cpp:66-75)
    (tmp_path / "bad.json").write_text(json.dumps(cfg))
    assert "glue_energy" in run(tmp_path / "bad.json").stderr
    assert run("-x", tmp_path / "bad.json").returncode == 1
    r = run("-h")
    assert r.returncode == 0 and "Loop formation simulator" in r.stderr


@pytest.mark.gpu
def test_1kb_driver_on_gpu(tmp_path, hip, oracle):
    _check(tmp_path, oracle, _make("gd_1kb", ".", "../csrc", "gdyn"), atol=5e-4)
