#!/usr/bin/env python3
"""bench.py -- bead-steps/s of the Brownian-dynamics hot path on the 100 kb whole-genome model.

A "step" is one Brownian-dynamics step of every replica resident on a GPU (S-genome-30k,
SURVEY.md section 8d: 46 chains, 30 000 beads, AB soft-core pairs + semispring chain bonds +
(i,i+2) harmonic bonds + ellipsoid wall with on-device wall dynamics, T=1, dt=1e-5), R replicas
batched per GPU, inputs resident in HBM.  N>1: one process per GPU (torch.distributed / RCCL),
independent replicas per rank (weak scaling), RCCL only for the broadcast of the model inputs and
the gather of summary statistics -- no data-path collective.

Prints ONE JSON line on rank 0 (see the contract in the task description).
"""
import argparse
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
PKG = "2022a-genome-dynamics_amd"
HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md


_FARM_WORKER = r"""
import importlib, sys, time, numpy as np
root, lib, npy, n_beads, steps, seed = sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4]), int(sys.argv[5]), int(sys.argv[6])
sys.path.insert(0, root)
g = importlib.import_module("2022a-genome-dynamics_amd"); wl = importlib.import_module("2022a-genome-dynamics_amd.workloads")
s, _ = wl.genome_interphase(g.Lib(lib), n_beads=n_beads, n_replicas=1)
s.set_positions(np.load(npy)[None]); s.begin_phase()
flags = g.RUN_UPDATE_SCALES | g.RUN_WALL_DYNAMICS
s.run(5, 1e-5, 1.0, seed=seed, noise=g.NOISE_MT19937, flags=flags)
print("ready", flush=True); sys.stdin.readline()
t0 = time.perf_counter(); s.run(steps, 1e-5, 1.0, seed=seed, noise=g.NOISE_MT19937, flags=flags); print(time.perf_counter() - t0, flush=True)
"""


def _oracle_rate(g, wl, lib_path, x0, n_beads, budget_s):
    orc = g.Lib(lib_path)
    s, _ = wl.genome_interphase(orc, n_beads=n_beads, n_replicas=1)
    s.set_positions(x0[None])
    s.begin_phase()
    flags = g.RUN_UPDATE_SCALES | g.RUN_WALL_DYNAMICS
    t0 = time.perf_counter()
    s.run(10, 1e-5, 1.0, seed=1, noise=g.NOISE_MT19937, flags=flags)
    per = (time.perf_counter() - t0) / 10
    steps = int(max(10, min(5000, budget_s / per)))
    t0 = time.perf_counter()
    s.run(steps, 1e-5, 1.0, seed=2, noise=g.NOISE_MT19937, flags=flags)
    el = time.perf_counter() - t0
    s.close()
    return steps, el


def cpu_baseline(g, wl, x0, n_beads, budget_s=12.0):
    """The oracle (CPU restatement, single thread, fp64, Verlet list, mt19937_64 normals like the reference's
    RNG class) timed on this host on the same workload, starting from the GPU-equilibrated positions:
    (i) 1 core with the reference's flags (-O2 -msse4 -mno-avx, 5-sim-genome/Makefile:7-12) -- faithful to the
    single-threaded reference; (ii) SURVEY 8d's farm comparison: one independent replica per host thread with the
    -O3 build.  Returns (cpu_baseline, farm) -- `farm` is extra information, not the contract's baseline."""
    import subprocess
    import tempfile
    steps, el = _oracle_rate(g, wl, os.path.join(ROOT, "oracle", "liboracle.so"), x0, n_beads, budget_s)
    base = {"value": n_beads * steps / el, "unit": "bead-steps/s", "cores": 1, "kind": "port",
            "sample": f"{steps} steps of 1 replica x {n_beads} beads (oracle/liboracle.so, fp64, Verlet list, "
                      f"mt19937_64 normals, flags -O2 -march=x86-64 -msse4 -mno-avx), {el:.1f} s on 1 of {os.cpu_count()} host threads"}
    farm = None
    fast = os.path.join(ROOT, "oracle", "liboracle_fast.so")
    try:
        cores = min(len(os.sched_getaffinity(0)), 64)
        try:        # a cgroup CPU quota below the affinity mask (the GPU box grants 16 CPUs per GPU): more processes only time-share
            quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
            if quota != "max":
                cores = max(1, min(cores, int(int(quota) / int(period))))
        except (OSError, ValueError):
            pass
        if os.path.exists(fast) and cores > 1:
            fsteps, fel = _oracle_rate(g, wl, fast, x0, n_beads, 2.0)
            fsteps = int(max(10, budget_s * 0.6 * fsteps / fel))      # the loaded machine runs slower than the 1-core probe
            with tempfile.TemporaryDirectory() as tmp:
                npy = os.path.join(tmp, "x0.npy")
                np.save(npy, x0)
                procs = [subprocess.Popen([sys.executable, "-c", _FARM_WORKER, ROOT, fast, npy, str(n_beads), str(fsteps), str(100 + k)],
                                          stdin=subprocess.PIPE, stdout=subprocess.PIPE, text=True) for k in range(cores)]
                for p in procs:
                    assert p.stdout.readline().strip() == "ready"
                t0 = time.perf_counter()
                for p in procs:
                    p.stdin.write("go\n"); p.stdin.flush()
                times = [float(p.stdout.readline()) for p in procs]
                wall = time.perf_counter() - t0
                for p in procs:
                    p.wait()
            farm = {"value": cores * n_beads * fsteps / wall, "unit": "bead-steps/s", "cores": cores, "kind": "port",
                    "one_core_value": n_beads * fsteps / min(times),
                    "sample": f"{cores} processes x {fsteps} steps of 1 replica x {n_beads} beads (oracle/liboracle_fast.so, "
                              f"-O3 -march=x86-64-v3), {wall:.1f} s wall"}
    except (OSError, ValueError, AssertionError) as e:
        farm = {"error": repr(e)}
    return base, farm


FORCE_RTOL = 5e-5      # tests/util.py: |dF| <= FORCE_RTOL * max|F| (fp32 device vs fp64 oracle)


def check_against_oracle(g, wl, sys_, n_beads, replica):
    """The state the timed steps left behind, checked: forces of one replica of the benchmark handle (same launch shape as the timed
    steps: all R replicas, tiled lists, the adapted interval) against the fp64 oracle on the same positions and context.
    Part of the cpu_baseline leg (the oracle is the checker, never the thing measured)."""
    x = sys_.positions()[replica]
    c = sys_.context(replica)
    Fh = sys_.forces()[replica]
    orc = g.Lib(os.path.join(ROOT, "oracle", "liboracle.so"))
    so, _ = wl.genome_interphase(orc, n_beads=n_beads, n_replicas=1)
    so.set_positions(x[None])
    so.begin_phase()
    so.set_context(0, c.step, c.bead_scale, c.bond_scale, tuple(c.semiaxes))
    Fo = so.forces()[0]
    so.close()
    scale = float(np.abs(Fo).max())
    err = float(np.abs(Fh - Fo).max() / scale)
    return {"replica": replica, "max_rel_force_err": err, "tolerance": FORCE_RTOL, "max_abs_force": scale,
            "what": "forces of one replica of the timed handle after the timed steps vs oracle/liboracle.so (fp64), |dF|max / |F|max",
            "list_path": {0: "none", 1: "generic", 2: "tiled"}[sys_.context(0).list_path], "ok": err <= FORCE_RTOL}


def _kernel_source_sha():
    import hashlib
    h = hashlib.sha256()
    for f in ("gdyn_kernels.hip", "gdyn_types.h"):
        h.update(open(os.path.join(ROOT, PKG, "csrc", f), "rb").read())
    return h.hexdigest()[:16]


def _cached_traffic(n_beads, replicas, list_entries_per_bead, why=None):
    """HBM bytes per k_step launch from the committed PMC passes (profiles/*_traffic.json: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE of
    this very workload, profiles/README.md).  Counters cannot be read from inside the process, so this is a cached profile value --
    used only while it describes the kernel being timed: same workload, same kernel source (hash), list length within 5 %.
    `why` (a list) receives the reason when a pass of this workload exists but does not apply ("source hash" / "list length")."""
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_traffic*.json")), reverse=True):
        try:
            tj = json.load(open(path))
            if tj["workload"] != {"n_beads": n_beads, "replicas_per_gpu": replicas}:
                continue
            if tj.get("kernel_source_sha") != _kernel_source_sha():
                if why is not None:
                    why.append("source hash")
                continue
            L0 = tj.get("list_entries_per_bead")
            if L0 and abs(list_entries_per_bead - L0) > 0.05 * L0:
                if why is not None:
                    why.append("list length")
                continue
            return {"bytes": tj["k_step"]["corrected_bytes_per_launch"], "source": "profiles/" + os.path.basename(path),
                    "build_bytes": (tj.get("build") or {}).get("corrected_bytes_per_build")}
        except (OSError, KeyError, ValueError):
            pass
    return None


def _cached_replay(n_beads, replicas):
    """The replay measurement of k_step (tools/replay.py -> profiles/r*_replay.json: its memory pattern alone, its arithmetic alone,
    on this workload's relaxed state), while it describes the kernel being timed (same kernel source hash)."""
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_replay.json")), reverse=True):
        try:
            rj = json.load(open(path))
            if rj.get("workload") == {"n_beads": n_beads, "replicas_per_gpu": replicas} and rj.get("kernel_source_sha") == _kernel_source_sha():
                return dict(rj, source="profiles/" + os.path.basename(path))
        except (OSError, KeyError, ValueError):
            pass
    return None


def other_workloads(g, wl, hip, dev_index, budget_steps=600):
    """The measurements BASELINE.md / SURVEY 8d list beside the headline, each after its own (short) relaxation, same
    clock as the headline (wall time of gd_run): ms per step, bead-steps/s, rollbacks.  Reported under config.other_workloads."""
    out = []

    def run_one(tag, make, flags, relax, steps, tune=None):
        s, info = make()
        if tune:
            s.set_tuning(**tune)
        dt, kT = info["timestep"], info["temperature"]
        N, R = info["n_beads"], s.R
        s.begin_phase()
        if relax:
            s.run(relax, dt, kT, seed=5, flags=0)
            if tune and tune.get("auto_skin"):
                # the list-width selection sweeps candidate widths for a few thousand steps: the timed steps start once the width
                # has stood still (and nothing rolled back) for 1 000 steps, within a bound
                for _ in range(10):
                    c0 = s.context()
                    s.run(1000, dt, kT, seed=5, flags=0)
                    c1 = s.context()
                    if c1.list_radius == c0.list_radius and c1.rollbacks == c0.rollbacks:
                        break
                    relax += 1000
            s.begin_phase()
        # (the relaxation ran with static scales and wall: the timed phase's flags switch the wall dynamics on, and the interval has to
        # re-adapt on complete intervals of THAT regime -- two chunks of twelve intervals, as the headline does before its window)
        s.run(max(steps // 4, 40, 24 * int(s.context().rebuild_interval)), dt, kT, seed=6, flags=flags)
        rb0 = s.context().rollbacks
        t0 = time.perf_counter()
        tm = s.run(steps, dt, kT, seed=7, flags=flags)
        el = time.perf_counter() - t0
        c = s.context()
        out.append({"workload": tag, "n_beads": N, "replicas": R, "steps": steps, "relax_steps": relax,
                    "bead_steps_per_s": N * R * steps / el, "ms_per_step": el / steps * 1e3,
                    "k_step_ms": tm.step_kernel_ms / max(tm.step_launches, 1), "rebuild_ms_per_step": tm.rebuild_ms / max(tm.step_launches, 1),
                    "list_entries_per_bead": c.list_entries / N, "rebuild_interval": int(c.rebuild_interval), "list_radius": c.list_radius,
                    "rollbacks_in_timed_steps": int(c.rollbacks - rb0), "kernel_path": {0: "none", 1: "generic", 2: "tiled"}[c.list_path],
                    "list_GB": c.list_bytes / 1e9, "row_repairs_last_chunk": int(c.row_repairs)})
        tr = _cached_traffic(N, R, c.list_entries / N)
        if tr:      # a committed PMC pass of this workload exists: HBM fraction of its step kernel, as for the headline
            gbs = tr["bytes"] / (out[-1]["k_step_ms"] * 1e-3) / 1e9
            out[-1]["roofline"] = {"bound": "hbm", "kernel": "k_step", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
                                   "traffic": tr["bytes"], "traffic_source": tr["source"]}
        s.close()

    f3 = g.RUN_UPDATE_SCALES | g.RUN_WALL_DYNAMICS
    run_one("S-genome-30k x 1 replica (what one reference-shaped driver process runs)",
            lambda: wl.genome_interphase(hip, n_beads=30000, n_replicas=1, device=dev_index), f3, 4000, 2 * budget_steps)
    # (relaxed as long as the headline's state, 20 000 steps: after the 4 000 of the other entries the random-walk start has not swollen
    # yet -- 33 list entries per bead where the relaxed state has 26 -- and rounds 2-4 read that as a deficit of the bead count)
    run_one("S-genome-62k x 32 replicas (production bead count)",
            lambda: wl.genome_interphase(hip, n_beads=62178, n_replicas=32, device=dev_index), f3, 20000, budget_steps)
    run_one("S-genome-62k x 64 replicas (production bead count, as many beads per launch as the headline)",
            lambda: wl.genome_interphase(hip, n_beads=62178, n_replicas=64, device=dev_index), f3, 20000, budget_steps)
    run_one("S-genome-30k x 128, bead_scale_init 0.5 (time-varying cutoff, simulation_driver_forcefield.cc:47-49)",
            lambda: wl.genome_interphase(hip, n_beads=30000, n_replicas=128, bead_scale_init=0.5, device=dev_index), f3, 4000, budget_steps)
    run_one("S-genome-30k x 128, 2nd-bond spring 0 (variant of SURVEY 8d)",
            lambda: wl.genome_interphase(hip, n_beads=30000, n_replicas=128, second_bond_spring=0.0, device=dev_index), f3, 4000, budget_steps)
    # the two states the headline's relaxed start is kinder than (profiles/r04_soak_128x60000.txt, r04_pipeline_scale_128.json):
    # the wall after it has contracted (60 000 steps take it from 6.15 to 4.47: the same model prepared at that volume, phi 0.78),
    # and the dense globule a freshly refined genome starts from (~290 list entries per bead at the default width: phi 3.3, timed
    # from 200 steps after the start, while it decondenses).  Proxies built from the synthetic generator, not saved states.
    run_one("S-genome-30k x 128 at the contracted wall (phi 0.78 = R_wall 4.47, the end of a 60 000-step run)",
            lambda: wl.genome_interphase(hip, n_beads=30000, n_replicas=128, phi=0.78, device=dev_index), f3, 4000, budget_steps)
    run_one("S-genome-30k x 32, dense start (phi 3.3: the list length of a freshly refined globule; decondensing)",
            lambda: wl.genome_interphase(hip, n_beads=30000, n_replicas=32, phi=3.3, device=dev_index), f3, 200, budget_steps)
    # (gd_tuning.auto_skin: the list width is selected for the workload from measured chunk times; the relaxation is long enough
    # for that sweep and for the one that follows when the tiles outgrow their class.  The genome workloads run at the library default, 0.75, which their sweeps confirm.)
    run_one("S-1kb-250k x 4 replicas (periodic, loops + glues static; auto_skin)",
            lambda: wl.chromatin_1kb(hip, n_beads=250000, n_replicas=4, device=dev_index), 0, 9000, budget_steps, tune=dict(auto_skin=1))
    run_one("S-1kb-250k x 16 replicas (as many beads per launch as the headline; auto_skin)",
            lambda: wl.chromatin_1kb(hip, n_beads=250000, n_replicas=16, device=dev_index), 0, 9000, budget_steps, tune=dict(auto_skin=1))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=1000, help="untimed steps with the timed phase's flags (the wall dynamics switch on here)")
    ap.add_argument("--beads", type=int, default=30000)
    ap.add_argument("--replicas", type=int, default=128, help="replicas batched per GPU (64: -5 %%, 256: -7 %% on one MI355X)")
    ap.add_argument("--equil", type=int, default=20000,
                    help="untimed relaxation steps before warmup (SURVEY 8d cfg3: 20 000 from the random-walk start)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the other workloads of the measurement row (config.other_workloads)")
    ap.add_argument("--save-state", default="", help="write the relaxed positions (after --equil) to this .npy and exit")
    ap.add_argument("--load-state", default="", help="start from relaxed positions saved by --save-state (no relaxation "
                    "launches: used for the committed rocprof summaries, so that kernel averages cover the timed state only)")
    ap.add_argument("--skin", type=float, default=0.0)
    ap.add_argument("--interval", type=int, default=0)
    ap.add_argument("--dist-backend", default="nccl", help="nccl (= RCCL over xGMI) on the GPU node; gloo for rehearsals")
    ap.add_argument("--lib", default="", help="developer builds: another HIP library of csrc/ (e.g. libgdyn_dev.so) instead of the product")
    ap.add_argument("--single-device", action="store_true", help="rehearsal: every rank uses GPU 0 (gloo; RCCL may refuse two ranks on one device)")
    ap.add_argument("--allow-stale-traffic", action="store_true",
                    help="development: do not fail when no committed PMC pass matches the kernel source hash (roofline.traffic null)")
    ap.add_argument("--launch-dry-run", action="store_true", help="with --gpus N > 1 and no launcher: print the launch command and exit")
    a = ap.parse_args()

    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # Started without a launcher (`python bench.py --gpus N`): this process becomes the parent of N fresh rank processes
        # (one per GPU, torch.distributed.run) and never touches the GPU itself -- nothing before this point imports torch
        # or loads the HIP library.  Rank 0's JSON line reaches stdout through the inherited descriptor; the parent returns
        # the launcher's status.  (The reference's ensemble shape: one process per seed, 5-sim-genome/scripts/run_simulation:8-25.)
        import socket
        import subprocess
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + [x for x in sys.argv[1:] if x != "--launch-dry-run"]
        if a.launch_dry_run:
            print(json.dumps({"launch": cmd, "torch_imported": "torch" in sys.modules}))
            return 0
        return subprocess.run(cmd).returncode

    status = 0
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != a.gpus:
        raise SystemExit(f"bench.py: --gpus {a.gpus} but the launcher started {world} rank(s) (WORLD_SIZE)")
    import torch
    import torch.distributed as dist
    dev_index = 0 if a.single_device else local_rank
    torch.cuda.set_device(dev_index)
    tdev = "cuda" if a.dist_backend == "nccl" else "cpu"
    grouped = world > 1 or all(k in os.environ for k in ("WORLD_SIZE", "RANK", "MASTER_ADDR", "MASTER_PORT"))     # under a launcher the group is formed even for one rank: the same RCCL
    if grouped:                                           # calls as the eight-rank farm then run on a one-GPU box
        if a.dist_backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev_index))
        else:
            dist.init_process_group(a.dist_backend)
        assert dist.get_world_size() == a.gpus, (dist.get_world_size(), a.gpus)

    g = importlib.import_module(PKG)
    wl = importlib.import_module(PKG + ".workloads")
    farm = importlib.import_module(PKG + ".farm")
    # one host thread + one stream per device: each rank stays on its GPU's share of the host CPUs (before the HIP library starts threads)
    cpu_share = farm.bind_to_cpu_share(local_rank, int(os.environ.get("LOCAL_WORLD_SIZE", world)))
    hip = g.load(a.lib or None)   # fails loudly if the HIP extension is missing

    R, N = a.replicas, a.beads
    # rank 0 generates the model inputs; broadcast to the farm (configs[4]); every rank then owns R
    # independent replicas -- no collective inside the timed region
    x0 = None
    if rank == 0:
        sys_, info = wl.genome_interphase(hip, n_beads=N, n_replicas=R, device=dev_index)
        x0 = sys_.positions()
    if grouped:
        x0 = farm.broadcast_array(x0, (R, N, 3), np.float64, device=tdev)
        if rank != 0:
            sys_, info = wl.genome_interphase(hip, n_beads=N, n_replicas=R, device=dev_index)
            sys_.set_positions(x0)
    if a.skin > 0 or a.interval > 0:
        sys_.set_tuning(skin=a.skin, rebuild_interval=a.interval, adapt_interval=0 if a.interval else 1)
    flags = g.RUN_UPDATE_SCALES | g.RUN_WALL_DYNAMICS
    seed = farm.replica_seed(wl.MASTER_SEED, rank)    # independent trajectories per rank (and per replica index)
    dt, kT = info["timestep"], info["temperature"]

    sys_.begin_phase()
    if a.load_state:
        sys_.set_positions(np.load(a.load_state))
    elif a.equil > 0:
        sys_.run(a.equil, dt, kT, seed=seed + 17, flags=0)      # relaxation: static scales / wall
        sys_.begin_phase()
    if a.save_state:
        np.save(a.save_state, sys_.positions())
        return
    if a.warmup > 0:
        sys_.run(a.warmup, dt, kT, seed=seed, flags=flags)

    rb0 = sys_.context(0).rollbacks
    farm.barrier(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    tm = sys_.run(a.steps, dt, kT, seed=seed, flags=flags)      # synchronous: returns after the stream drained
    torch.cuda.synchronize()
    el_own = time.perf_counter() - t0      # this rank alone (reported per rank; `value` uses the max over ranks behind the barrier)
    farm.barrier()
    el = farm.max_over_ranks(time.perf_counter() - t0, device=tdev)
    rollbacks_timed = sys_.context(0).rollbacks - rb0

    # steady-state rate: a window of at least 40 rebuild intervals (the timed region of a short run holds one build or two by
    # chance), time-boxed; reported beside `value`, never instead of it
    K_now = max(int(sys_.context(0).rebuild_interval), 1)
    n_ss = min(max(40 * K_now, 400), 4000)
    # (ten intervals untimed first: with --warmup 5 the wall dynamics were switched on 25 steps ago, and the interval adaptation has
    # not seen a complete interval of this phase yet -- the window below is the STEADY state)
    sys_.run(10 * K_now, dt, kT, seed=seed, flags=flags)
    farm.barrier(); torch.cuda.synchronize()
    t_ss = time.perf_counter()
    sys_.run(n_ss, dt, kT, seed=seed, flags=flags)
    torch.cuda.synchronize(); farm.barrier()
    el_ss = farm.max_over_ranks(time.perf_counter() - t_ss, device=tdev)

    # second figure: the same stepping with the reference's observation cadence -- mean energy every
    # interphase_logging_interval = 100 steps, a quantised snapshot every interphase_sampling_interval = 1000 steps
    # (config_entries.inc:81-82); 1000 extra steps, not part of `value`
    farm.barrier()
    sys_.positions_f32(quantize=True)      # (a driver takes thousands of snapshots: not the first, cold one)
    t1 = time.perf_counter()
    for k in range(10):
        # as the drivers do it: stop where the reference's callback observes (before its state updates), observe, apply them
        sys_.run(100, dt, kT, seed=seed, flags=flags | g.RUN_DEFER_CALLBACK)
        e_obs = sys_.energy()
        if k == 9:
            snap = sys_.positions_f32(quantize=True)
        sys_.apply_callback()
    obs_rate = N * R * 1000 / (time.perf_counter() - t1)
    del snap, e_obs

    ctx = sys_.context(0)
    e_mean = float(sys_.energy().mean() / N)
    gathered = farm.gather_stats([e_mean, ctx.semiaxes[0], float(ctx.rebuild_interval), float(ctx.rollbacks)], device=tdev)
    # farm hygiene, checked on rank 0: every rank's CPU share (0/1 per CPU of the machine) and its own rate
    ncpu = os.cpu_count() or 1
    shares = farm.gather_stats([1.0 if c in cpu_share else 0.0 for c in range(ncpu)], device=tdev)
    rates = farm.gather_stats([N * R * a.steps / el_own], device=tdev)

    if rank == 0:
        # HBM bytes per k_step launch: the committed PMC passes while they describe this kernel and list (else the algorithmic
        # figure alone is reported)
        L_bead = tm.list_entries_visited / max(int(tm.step_launches), 1) / (N * R)
        why_not = []
        tr = _cached_traffic(N, R, L_bead, why_not)
        traffic, traffic_src = (tr["bytes"], tr["source"]) if tr else (None, None)
        launches = max(int(tm.step_launches), 1)
        L_launch = tm.list_entries_visited / launches                 # directed entries, all replicas
        kms = tm.step_kernel_ms / launches                            # HIP events on the handle's stream around the step launches
        # SURVEY 8d's per-unit figure: 204 B per bead-step (44 N + 28 L at the reference list L = 5.7 N) -- a list-length-
        # independent price: the bytes a gather-from-HBM design with a 1.2 x cutoff list would move.  The tiled kernel
        # reads neighbours from LDS and keeps a longer list (2 B per entry), so its own traffic is the PMC figure.
        alg_bytes = 204.0 * N * R
        alg_gbs = alg_bytes / (kms * 1e-3) / 1e9
        meas_gbs = traffic / (kms * 1e-3) / 1e9 if traffic else None
        # list entries a bead HAS to walk per step: its near class (counted by the build in the fours k_step walks, gd_context.near_entries
        # of replica 0); the far class joins only in the last steps of an interval -- a lower bound, as a minimum should be
        walked = ctx.near_entries / N if ctx.near_entries else L_launch / (N * R)
        K_live = max(int(ctx.rebuild_interval), 1)
        whole_bytes = (traffic + tr["build_bytes"] / K_live) if (tr and tr.get("build_bytes")) else None
        whole_frac = whole_bytes / (tm.total_ms / launches * 1e-3) / 1e9 / HBM_PEAK_GBS if whole_bytes else None
        out = {
            "metric": "bead-steps/sec on 100kb whole-genome model, 1 GPU and 8-GPU replica farm",
            "value": N * R * world * a.steps / el, "unit": "bead-steps/s",
            "n_gpus": world, "rccl_ranks": world if (grouped and a.dist_backend == "nccl") else 0, "dist_backend": a.dist_backend if grouped else None,
            "cpus_per_rank": len(cpu_share), "cpu_shares_disjoint": bool((shares.sum(axis=0) <= 1.0).all()) if world > 1 else None,
            "bead_steps_per_s_per_rank": [float(v[0]) for v in rates],
            "steps": a.steps, "warmup": a.warmup, "ms_per_step": el / a.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"S-genome-{round(N / 1000)}k (5-sim-genome interphase force field, wall dynamics + scale updates on)",
                       "n_beads": N, "replicas_per_gpu": R, "global_replicas": R * world, "parallelism": f"replica-farm x{world}",
                       "timestep": dt, "temperature": kT, "list_entries_per_bead": L_launch / (N * R),
                       "rebuild_interval": int(ctx.rebuild_interval), "list_radius": ctx.list_radius, "list_GB": ctx.list_bytes / 1e9,
                       "row_repairs_last_chunk": int(ctx.row_repairs),
                       "rollbacks": int(ctx.rollbacks), "rollbacks_in_timed_steps": int(rollbacks_timed), "equil_steps": a.equil,
                       "steady_state_bead_steps_per_s": N * R * world * n_ss / el_ss, "steady_state_steps": n_ss,
                       "bead_steps_per_s_with_reference_cadence_rank0": obs_rate,
                       "mean_energy_per_bead": [float(v[0]) for v in gathered], "wall_semiaxis": [float(v[1]) for v in gathered]},
            # frac = measured HBM bytes of the dominant kernel / its launch time / 8 TB/s (<= 1 by construction).
            # whole_step_counter_frac: the same counters over the WHOLE step -- k_step's bytes + the list build's bytes (every kernel of
            # the build chain, same PMC passes) / the live rebuild interval, over the device time per step (steps + builds).
            # design_min_bytes_per_launch: what this design has to move per k_step launch -- per bead-step 16 B position read + 24 B of
            # per-thread records + 16 B adjacency chunk + 2 B per NEAR list entry (near_entries_per_bead; the far class joins in the
            # last steps of an interval only) + 16 B position store: counter traffic well above it would be wasted re-reads.
            "roofline": {"bound": "hbm", "kernel": "k_step", "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "achieved": meas_gbs if meas_gbs is not None else alg_gbs,
                         "frac": (meas_gbs if meas_gbs is not None else alg_gbs) / HBM_PEAK_GBS,
                         "frac_kind": "pmc_traffic_over_live_kernel_time" if meas_gbs is not None else "survey_204B_per_bead_step_over_live_kernel_time",
                         "traffic": traffic, "traffic_source": traffic_src,
                         "design_min_bytes_per_launch": (72.0 + 2.0 * walked) * N * R, "near_entries_per_bead": walked,
                         "whole_step_counter_frac": whole_frac, "whole_step_counter_bytes_per_step": whole_bytes,
                         "avg_launch_ms": kms,
                         "rebuild_ms_per_step": tm.rebuild_ms / launches, "device_total_ms_per_step": tm.total_ms / launches},
        }
        # SURVEY 8d's list-independent price (204 B per bead-step: a gather-from-HBM design with a 1.2 x cutoff list), kept for
        # reference only -- NOT a roofline claim: the tiled kernel does not move those bytes
        out["config"]["survey_pricing_non_credit"] = {"bytes_per_launch": alg_bytes, "k_step_frac": alg_gbs / HBM_PEAK_GBS,
                                                      "whole_step_frac": 204.0 * N * R * a.steps / el / 1e9 / HBM_PEAK_GBS}
        rp = _cached_replay(N, R)
        if rp:
            # what actually bounds k_step: VALU issue.  Its arithmetic alone (operands resident, same grid and LDS class) takes t_alu,
            # its memory pattern alone t_mem; the product kernel overlaps the two almost completely (profiles/README.md)
            out["roofline"]["issue_bound"] = {"t_alu_us": rp["t_alu_us"], "t_mem_us": rp["t_mem_us"], "t_kstep_fresh_list_us": rp["t_kstep_us"],
                                              "overlap": rp["overlap"], "kstep_over_alu_bound": rp["t_alu_us"] / rp["t_kstep_us"], "source": rp["source"]}
        if not a.no_cpu_baseline and world == 1:
            out["checked"] = check_against_oracle(g, wl, sys_, N, R // 2)
            out["cpu_baseline"], farm_cpu = cpu_baseline(g, wl, sys_.positions()[0], N)
            out["config"]["gpu_over_cpu_1core"] = out["value"] / out["cpu_baseline"]["value"]
            if farm_cpu is not None:
                out["config"]["cpu_farm_one_replica_per_core"] = farm_cpu      # `cores` = the cores this process may run on (affinity)
        if not a.no_extra and world == 1:
            sys_.close()
            out["config"]["other_workloads"] = other_workloads(g, wl, hip, dev_index)
        print(json.dumps(out), flush=True)
        if world > 1:
            # a farm whose ranks are not what the line says is not a measurement: on the real backend every rank is an RCCL rank,
            # and the ranks' CPU shares do not overlap (unless binding was switched off or the launcher pinned the ranks itself)
            if a.dist_backend == "nccl" and out["rccl_ranks"] != world:
                sys.stderr.write(f"bench.py: {out['rccl_ranks']} RCCL rank(s) for a world of {world}\n"); status = 5
            if not out["cpu_shares_disjoint"] and not os.environ.get("GDYN_NO_BIND") and len(cpu_share) < ncpu:
                sys.stderr.write("bench.py: the ranks' CPU shares overlap\n"); status = 5
        if traffic is None and "source hash" in why_not and "list length" not in why_not and not a.lib and not a.allow_stale_traffic:
            # the headline's roofline needs the counter traffic: a committed PMC pass of THIS kernel source (profiles/r*_traffic.json is
            # keyed by the hash of gdyn_kernels.hip + gdyn_types.h and by the list length).  After a kernel edit: tools/profile_round.sh
            sys.stderr.write("bench.py: the committed PMC traffic of this workload was taken from another kernel source (roofline.traffic is null): "
                             "run tools/profile_round.sh and commit profiles/<tag>_traffic.json, or pass --allow-stale-traffic\n")
            status = 4
        if "checked" in out and not out["checked"]["ok"]:
            sys.stderr.write(f"bench.py: the timed state fails the oracle check: {out['checked']}\n")
            status = 3
    sys_.close()
    if grouped:
        dist.destroy_process_group()
    return status


if __name__ == "__main__":
    sys.exit(main())
