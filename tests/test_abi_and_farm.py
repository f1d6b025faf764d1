"""Host-side checks that need no GPU: the C-ABI library loads and exports every symbol gdyn.h
declares, fails loudly without a device, and the replica farm plumbing works at world_size 2 (gloo)."""
import os
import re
import subprocess
import sys
import textwrap

import numpy as np
import pytest

from conftest import ROOT
from util import g

PKG_DIR = "2022a-genome-dynamics_amd"


def _built():
    if not os.path.exists(g.LIBGDYN_PATH):
        subprocess.check_call(["make", "-C", os.path.dirname(g.LIBGDYN_PATH)])
    return g.Lib(g.LIBGDYN_PATH)


def test_header_symbols_match_binding():
    hdr = open(os.path.join(ROOT, "include", "gdyn.h")).read()
    declared = set(re.findall(r"^(?:int|const char \*)\s*(gd_\w+)\(", hdr, flags=re.M))
    assert declared == set(g.ABI_SYMBOLS), declared ^ set(g.ABI_SYMBOLS)


def test_abi_version_is_checked_at_handle_creation(oracle):
    """A caller built against another version of include/gdyn.h (struct layouts) is refused at gd_create: the macro passes the
    header's GD_ABI_VERSION, the ctypes binding its own constant."""
    hdr = open(os.path.join(ROOT, "include", "gdyn.h")).read()
    assert int(re.search(r"#define GD_ABI_VERSION (\d+)", hdr).group(1)) == g.ABI_VERSION
    import ctypes as C
    for lib in (oracle, _built()):
        assert lib.dll.gd_abi_version() == g.ABI_VERSION
        h = C.c_void_p()
        d = g._Desc(10, 1, 0, 0, (C.c_double * 3)(0, 0, 0))
        rc = lib.dll.gd_create_abi(g.ABI_VERSION - 1, C.byref(d), C.byref(h))
        assert rc == 1 and b"ABI version" in lib.dll.gd_last_error()        # GD_EINVAL


def test_libgdyn_loads_and_exports_abi():
    lib = _built()                                   # Lib() raises on any missing symbol
    assert lib.backend == "hip"
    for name in g.ABI_SYMBOLS:
        assert hasattr(lib.dll, name)


def test_oracle_exports_same_abi(oracle):
    for name in g.ABI_SYMBOLS:
        assert hasattr(oracle.dll, name)


def test_no_cpu_fallback_without_device():
    import torch
    if torch.cuda.device_count() > 0:
        pytest.skip("a GPU is visible; the no-device error path cannot be exercised")
    lib = _built()
    with pytest.raises(g.GdynError) as e:
        g.System(lib, 10, 1)
    assert e.value.code == 2 and "no HIP device" in str(e.value)        # GD_ENODEVICE


def test_product_never_references_oracle():
    pkg = os.path.join(ROOT, "2022a-genome-dynamics_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp", "Makefile")):
                txt = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "liboracle" not in txt and "gdyn_oracle" not in txt, f


WORKER = textwrap.dedent('''
    import importlib, os, sys
    import numpy as np
    import torch.distributed as dist
    sys.path.insert(0, {root!r})
    g = importlib.import_module("2022a-genome-dynamics_amd")
    wl = importlib.import_module("2022a-genome-dynamics_amd.workloads")
    farm = importlib.import_module("2022a-genome-dynamics_amd.farm")
    dist.init_process_group("gloo")
    rank, world = farm.world()
    lib = g.Lib(os.path.join({root!r}, "oracle", "liboracle.so"))     # CPU stand-in for the device in this plumbing test
    R, N = 2, 400
    x0 = None
    if rank == 0:
        s0, _ = wl.genome_interphase(lib, n_beads=N, n_replicas=R)
        x0 = s0.positions()
    x0 = farm.broadcast_array(x0, (R, N, 3), np.float64)
    s, info = wl.genome_interphase(lib, n_beads=N, n_replicas=R)
    s.set_positions(x0)
    s.begin_phase()
    farm.barrier()
    s.run(5, info["timestep"], 1.0, seed=farm.replica_seed(wl.MASTER_SEED, rank), flags=3)
    x = s.positions()
    stats = farm.gather_stats([float(np.abs(x0).sum()), float(x[0, 0, 0]), float(s.energy().mean() / N)])
    t = farm.max_over_ranks(1.0 + rank)
    assert t == world
    if rank == 0:
        assert stats.shape == (world, 3)
        assert stats[0, 0] == stats[1, 0]            # identical broadcast inputs
        assert stats[0, 1] != stats[1, 1]            # independent noise streams
        print("FARM_OK", stats[:, 2])
    dist.destroy_process_group()
''')


def test_replica_farm_gloo_world2(tmp_path, oracle):
    script = tmp_path / "worker.py"
    script.write_text(WORKER.format(root=ROOT))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                          "--master-addr", "127.0.0.1", "--master-port", "29533", str(script)],
                         capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    assert "FARM_OK" in out.stdout


@pytest.mark.gpu
def test_bench_farm_rehearsal_two_ranks_one_gpu(hip):
    """The N > 1 path of bench.py exactly as the driver launches it (one process per rank through torch.distributed.run),
    rehearsed on one MI355X: two ranks share GPU 0, gloo stands in for RCCL (the farm only broadcasts the inputs and
    gathers summary statistics; there is no collective in the timed region).  Small workload, same code path."""
    import json
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                          "--master-addr", "127.0.0.1", "--master-port", "29547", os.path.join(ROOT, "bench.py"),
                          "--gpus", "2", "--dist-backend", "gloo", "--single-device", "--beads", "3000", "--replicas", "8",
                          "--equil", "200", "--steps", "40", "--warmup", "10", "--no-extra", "--no-cpu-baseline"],
                         capture_output=True, text=True, timeout=600, env=dict(os.environ, MASTER_ADDR="127.0.0.1"))
    assert out.returncode == 0, out.stderr[-3000:]
    line = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["scaling"] == "weak" and line["config"]["global_replicas"] == 16
    assert line["value"] > 0 and line["steps"] == 40 and line["warmup"] == 10
    assert len(line["config"]["mean_energy_per_bead"]) == 2          # one summary row gathered from each rank
    assert line["config"]["mean_energy_per_bead"][0] != line["config"]["mean_energy_per_bead"][1]      # independent trajectories
    assert line["cpu_shares_disjoint"] is True and len(line["bead_steps_per_s_per_rank"]) == 2 and min(line["bead_steps_per_s_per_rank"]) > 0


@pytest.mark.gpu
def test_bench_one_rank_under_the_launcher_runs_the_rccl_collectives(hip):
    """The farm's collectives on the real backend: `torch.distributed.run --nproc-per-node 1 bench.py --gpus 1` forms a one-rank
    RCCL group on the MI355X and runs the same calls as the eight-rank farm (communicator set-up, the broadcast of the (R, N, 3)
    float64 inputs from device memory, barriers, the max-over-ranks all-reduce, the gather of the summary rows)."""
    import json
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1",
                          "--master-addr", "127.0.0.1", "--master-port", "29561", os.path.join(ROOT, "bench.py"),
                          "--gpus", "1", "--beads", "3000", "--replicas", "8", "--equil", "200", "--steps", "40", "--warmup", "10",
                          "--no-extra", "--no-cpu-baseline"],
                         capture_output=True, text=True, timeout=600,
                         env=dict(os.environ, MASTER_ADDR="127.0.0.1", NCCL_DEBUG="VERSION"))
    assert out.returncode == 0, out.stderr[-3000:]
    line = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 1 and line["rccl_ranks"] == 1 and line["dist_backend"] == "nccl"
    assert line["value"] > 0 and len(line["config"]["mean_energy_per_bead"]) == 1


@pytest.mark.gpu
def test_farm_collectives_on_a_one_rank_rccl_group(hip):
    """farm.py's four collectives called directly on device tensors of a one-rank RCCL group, values checked."""
    code = (
        "import os, sys, importlib, numpy as np, torch, torch.distributed as dist\n"
        f"sys.path.insert(0, {ROOT!r})\n"
        "farm = importlib.import_module('2022a-genome-dynamics_amd.farm')\n"
        "torch.cuda.set_device(0)\n"
        "dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))\n"
        "x = np.random.default_rng(3).normal(size=(8, 3000, 3))\n"
        "y = farm.broadcast_array(x, x.shape, np.float64, device='cuda')\n"
        "assert y.dtype == np.float64 and np.array_equal(x, y)\n"
        "g = farm.gather_stats([1.5, -2.0, 3.25], device='cuda')\n"
        "assert g.shape == (1, 3) and g.tolist() == [[1.5, -2.0, 3.25]]\n"
        "assert farm.max_over_ranks(0.125, device='cuda') == 0.125\n"
        "farm.barrier(); torch.cuda.synchronize()\n"
        "print('RCCL_OK', dist.get_backend(), torch.cuda.nccl.version())\n"
        "dist.destroy_process_group()\n")
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600,
                         env=dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29563"))
    assert out.returncode == 0, out.stderr[-3000:]
    assert "RCCL_OK nccl" in out.stdout


def test_bench_self_launch_builds_the_rank_command_without_touching_torch():
    """`python bench.py --gpus N` without a launcher: the parent only starts N rank processes (torch.distributed.run) and
    must not have imported torch (nothing may initialise the GPU before the ranks exist)."""
    import json
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--steps", "7", "--launch-dry-run"],
                         capture_output=True, text=True, timeout=120,
                         env={k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")})
    assert out.returncode == 0, out.stderr[-2000:]
    d = json.loads(out.stdout.strip().splitlines()[-1])
    assert d["torch_imported"] is False
    cmd = d["launch"]
    assert cmd[1:3] == ["-m", "torch.distributed.run"] and "--nproc-per-node=4" in cmd and "127.0.0.1" in cmd
    assert cmd[cmd.index(os.path.join(ROOT, "bench.py")) + 1:] == ["--gpus", "4", "--steps", "7"]


def test_bench_rejects_a_world_size_other_than_gpus():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1"], capture_output=True, text=True,
                         timeout=120, env=dict(os.environ, WORLD_SIZE="3", RANK="0", LOCAL_RANK="0"))
    assert out.returncode != 0 and "--gpus 2" in out.stderr


@pytest.mark.gpu
def test_bench_gpus2_without_a_launcher_starts_its_own_ranks(hip):
    """The way the driver runs the 1-GPU line, with --gpus 2: bench.py itself starts the two ranks (fresh child processes;
    both on GPU 0 here, gloo standing in for RCCL) and relays rank 0's line."""
    import json
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dist-backend", "gloo", "--single-device",
                          "--beads", "3000", "--replicas", "8", "--equil", "200", "--steps", "40", "--warmup", "10", "--no-extra",
                          "--no-cpu-baseline"], capture_output=True, text=True, timeout=600,
                         env={k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")})
    assert out.returncode == 0, out.stderr[-3000:]
    line = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["dist_backend"] == "gloo" and line["rccl_ranks"] == 0
    assert line["config"]["global_replicas"] == 16 and line["value"] > 0


def test_per_device_setup_guard_runs_once_per_device_and_blocks_until_done(tmp_path):
    """csrc/gdyn_once.hpp (the guard gd_create runs the kernels' LDS opt-in under): one set-up per device ordinal, no caller
    past it before it has returned, the status kept -- 32 threads over 4 ordinals with a slow set-up (tests/native/test_once.cpp)."""
    exe = str(tmp_path / "test_once")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-pthread", "-I", os.path.join(ROOT, PKG_DIR, "csrc"), "-o", exe,
                           os.path.join(ROOT, "tests", "native", "test_once.cpp")])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=60)
    assert out.returncode == 0 and "once: ok" in out.stdout, out.stdout + out.stderr
    # the same under ThreadSanitizer (sanitizers run on the CPU builds only): no report
    exe_t = str(tmp_path / "test_once_tsan")
    if subprocess.call(["g++", "-O1", "-std=c++17", "-pthread", "-fsanitize=thread", "-I", os.path.join(ROOT, PKG_DIR, "csrc"), "-o", exe_t,
                        os.path.join(ROOT, "tests", "native", "test_once.cpp")], stderr=subprocess.DEVNULL) == 0:
        out = subprocess.run([exe_t], capture_output=True, text=True, timeout=120)
        assert out.returncode == 0 and "once: ok" in out.stdout and "ThreadSanitizer" not in out.stderr, out.stdout + out.stderr


def test_no_launcher_keeps_an_unsynchronised_once_flag():
    """The kernels' per-device attributes are set in ONE place (gd_kernels_init_device, under the guard); no launcher may carry a
    `static bool once` of its own again."""
    src = open(os.path.join(ROOT, PKG_DIR, "csrc", "gdyn_kernels.hip")).read()
    code = "\n".join(ln.split("//")[0] for ln in src.splitlines())
    assert "static bool" not in code
    assert code.count("hipFuncAttributeMaxDynamicSharedMemorySize") == 1      # (the one lambda of gd_kernels_init_device)
    capi = open(os.path.join(ROOT, PKG_DIR, "csrc", "gdyn_capi.hip")).read()
    assert "gd_kernels_init_device()" in capi and "DeviceOnce" in capi


@pytest.mark.gpu
def test_bench_two_rccl_ranks_on_one_device_or_the_refusal_recorded(hip):
    """8-GPU readiness as far as one GPU goes: `bench.py --gpus 2` on the REAL backend (nccl = RCCL) with both ranks on GPU 0.  RCCL
    may refuse two ranks on one device (it did on the pool's image: the refusal text becomes the skip reason, so the record shows
    what was attempted); where it is allowed, the line must carry rccl_ranks == 2, disjoint CPU shares and one rate per rank."""
    import json
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                          "--master-addr", "127.0.0.1", "--master-port", "29571", os.path.join(ROOT, "bench.py"),
                          "--gpus", "2", "--dist-backend", "nccl", "--single-device", "--beads", "3000", "--replicas", "8",
                          "--equil", "200", "--steps", "40", "--warmup", "10", "--no-extra", "--no-cpu-baseline"],
                         capture_output=True, text=True, timeout=600, env=dict(os.environ, MASTER_ADDR="127.0.0.1", NCCL_DEBUG="WARN"))
    if out.returncode != 0:
        text = out.stderr + out.stdout
        refusal = [ln.strip() for ln in text.splitlines() if "Duplicate GPU" in ln or "invalid usage" in ln.lower() or "ncclInvalidUsage" in ln]
        assert refusal, text[-3000:]      # any other failure is a failure
        pytest.skip("RCCL refuses two ranks on one device: " + refusal[0][:300])
    line = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["rccl_ranks"] == 2 and line["dist_backend"] == "nccl"
    assert line["cpu_shares_disjoint"] is True and len(line["bead_steps_per_s_per_rank"]) == 2
    assert line["config"]["global_replicas"] == 16 and line["value"] > 0


def test_every_entry_point_makes_its_device_current_before_its_first_hip_call():
    """include/gdyn.h: independent handles may live on different threads AND devices.  Audit of csrc/gdyn_capi.hip: every extern "C"
    entry point that reaches the HIP runtime sets the handle's device first -- hipSetDevice itself or one of the helpers that start
    with it (prepare, fetch_xyz, apply_pending, finalize_topology)."""
    import re
    src = open(os.path.join(ROOT, PKG_DIR, "csrc", "gdyn_capi.hip")).read()
    src = re.sub(r"//[^\n]*", "", src)
    starts = [m.start() for m in re.finditer(r'^extern "C" ', src, re.M)] + [len(src)]
    helpers = r"hipSetDevice|prepare\(s\)|fetch_xyz\(s|apply_pending\(s\)|finalize_topology\(s\)"
    for h in ("prepare", "fetch_xyz", "apply_pending", "finalize_topology"):      # the helpers do start with it (apply_pending: once something is pending)
        body = src[src.index("static int " + h + "("):]
        body = body[:body.index("\n}\n")]
        first_hip = re.search(r"\bhip[A-Z]\w+\s*\(|gd_launch_", body)
        assert first_hip and (first_hip.group(0).startswith("hipSetDevice") or re.search(helpers, body[:first_hip.start()])), h
    checked = 0
    for a, b in zip(starts, starts[1:]):
        fn = src[a:b]
        first_line = fn.split("\n", 1)[0]
        fn = first_line if first_line.count("{") == first_line.count("}") and "{" in first_line else (fn[:fn.index("\n}\n")] if "\n}\n" in fn else fn)
        name = re.search(r"(gd_\w+)\s*\(", fn).group(1)
        uses = re.search(r"\bhip(?!GetDeviceCount|GetErrorString|Success|Error_t)[A-Z]\w+\s*\(|gd_launch_|build_now\(|ensure_fresh_list\(|search_device\(|upload_ctx\(|download_ctx\(|\.resize\(", fn)
        if not uses:
            continue
        sets = re.search(helpers, fn)
        assert sets and sets.start() <= uses.start(), name
        checked += 1
    assert checked >= 12


def test_committed_pmc_traffic_describes_the_kernel_source_in_the_tree():
    """bench.py exits with status 4 when the committed PMC traffic of the headline workload was taken from another kernel source
    (hash of gdyn_kernels.hip + gdyn_types.h): a kernel edit without a new profile round must fail HERE, on the CPU, not in the
    driver's bench run."""
    import glob
    import importlib.util
    import json
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    sha = bench._kernel_source_sha()
    matches = []
    for path in glob.glob(os.path.join(ROOT, "profiles", "r*_traffic*.json")):
        tj = json.load(open(path))
        if tj.get("workload") == {"n_beads": 30000, "replicas_per_gpu": 128} and tj.get("kernel_source_sha") == sha:
            matches.append(path)
    assert matches, f"no profiles/r*_traffic.json for kernel source {sha}: run tools/profile_round.sh and tools/traffic_json.py"
    tr = bench._cached_traffic(30000, 128, json.load(open(matches[0]))["list_entries_per_bead"])
    assert tr and tr["bytes"] > 1e8 and tr["build_bytes"] > 1e8
