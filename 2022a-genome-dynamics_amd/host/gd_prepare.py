#!/usr/bin/env python3
"""gd_prepare -- creates the trajectory file a whole-genome run starts from.

The reference's `scripts/prepare [--seed S] <config> <genome> <output>` (5-sim-genome/src/prepare/__main__.py:13-32,
prepare/run.py:21-123): simulation config = defaults + the user's JSON + the seeds derived from the master seed
(run.py:36-57); genome bead table -> particles, chains, nucleolar side beads and bonds (system_definition.py:56-140);
written as /metadata/{config, ab_factors f32, particle_types i8 enum, chromosome_ranges / centromere_ranges /
nucleolus_ranges i32 + "keys" attributes, nucleolus_bonds i32} plus the empty /snapshots/<phase> groups.

The tables come from the package's prepare module (pinned by fixtures recorded from the reference's own modules);
h5py is not in this image, so the file itself is written by gd_h5tool on the HDF5 C API.
"""
import argparse
import csv
import importlib
import json
import os
import subprocess
import sys
import tempfile
from collections import OrderedDict

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


def _module(name):
    """A module of the package this program ships in (the package directory name is not an identifier: by path)."""
    pkg_dir = os.path.dirname(HERE)
    root = os.path.dirname(pkg_dir)
    if root not in sys.path:
        sys.path.insert(0, root)
    return importlib.import_module(os.path.basename(pkg_dir) + "." + name)


def h5tool(*args, capture=False):
    env = dict(os.environ, LD_LIBRARY_PATH=os.path.join(HERE, "h5lib") + ":" + os.environ.get("LD_LIBRARY_PATH", ""))
    cmd = [os.path.join(HERE, "gd_h5tool"), *map(str, args)]
    if capture:
        return subprocess.check_output(cmd, env=env, text=True)
    subprocess.check_call(cmd, env=env)
    return None


def load_config(filename, seed, defaults, derive_seeds):
    config = OrderedDict(defaults)
    with open(filename) as fh:
        config.update(json.load(fh, object_pairs_hook=OrderedDict))
    if seed is None:                      # run.py:46-47: a seed in the config file, else a fresh one
        seed = config.setdefault("seed", int(np.random.randint(1000000)))
    config["seed"] = seed
    config["spindle_seed"], config["interphase_seed"] = derive_seeds(seed)
    return config


def read_genome(filename):
    with open(filename, newline="") as fh:
        rows = list(csv.DictReader(fh, delimiter="\t"))
    return [(r["chain"], int(r["start"]), int(r["end"]), float(r["A"]), float(r["B"]), r["tags"]) for r in rows]


def main(argv=None):
    ap = argparse.ArgumentParser(prog="gd_prepare", description="Prepare a trajectory file from simulation config and data.")
    ap.add_argument("--seed", type=int, default=None, help="random seed")
    ap.add_argument("configfile")
    ap.add_argument("genomefile")
    ap.add_argument("outputfile")
    a = ap.parse_args(argv)
    prep, wl = _module("prepare"), _module("workloads")
    config = load_config(a.configfile, a.seed, wl.DEFAULT_CONFIG, prep.derive_seeds)
    system = prep.make_system(read_genome(a.genomefile), config)
    with tempfile.TemporaryDirectory() as tmp:
        with open(os.path.join(tmp, "config.json"), "w") as fh:
            fh.write(json.dumps(config))
        system["ab_factors"].astype("<f4").tofile(os.path.join(tmp, "ab.f32"))
        system["particle_types"].astype("i1").tofile(os.path.join(tmp, "types.i8"))
        with open(os.path.join(tmp, "chromosomes.tsv"), "w") as fh:
            for name, (b, e), (c0, c1) in zip(system["chromosome_names"], system["chromosome_ranges"], system["centromere_ranges"]):
                fh.write(f"{name} {b} {e} {c0} {c1}\n")
        with open(os.path.join(tmp, "nucleoli.tsv"), "w") as fh:
            for name, (b, e) in zip(system["nucleolus_names"], system["nucleolus_ranges"]):
                fh.write(f"{name} {b} {e}\n")
        system["nucleolus_bonds"].astype("<i4").tofile(os.path.join(tmp, "nucleolus_bonds.i32"))
        h5tool("make-metadata", a.outputfile, tmp)
    return 0


if __name__ == "__main__":
    sys.exit(main())
