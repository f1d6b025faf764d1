// gd_farm -- replica farm launcher: one driver process per GPU, started before anything touches a GPU.
//
//   gd_farm [--gpus G] [--dry-run] <driver> <trajectory>...
//
// The reference's ensemble is one run per seed, each on its own prepared trajectory file
// (5-sim-genome/scripts/run_simulation:8-25; the analyses read them back as output-*.h5,
// 5-sim-genome/src/contact_map/contact_map.py:14-39).  The farm splits the files into G contiguous groups and starts
//   <driver> --device g <files of group g>
// for g = 0..G-1; each driver batches its files as replicas of one libgdyn handle (gd_interphase).  Trajectories are
// independent: no data moves between the processes, the "gather" is the set of output files.  The launcher itself makes
// no HIP call (a process that has initialised a GPU must not exec another program on this platform); G defaults to the
// number of GPU nodes the kernel driver lists.  Exit status: the largest child status.
// Each driver process is bound to its GPU's share of the CPUs this launcher may run on (share g of G, contiguous): the stepping
// thread, its packing pool and the writer thread of one GPU do not migrate over, or contend with, those of another
// (--no-bind turns that off).  A batched driver prints its own device's rate when it ends ("[rate] device g: ... bead-steps/s");
// when every process has ended the launcher prints one summary line per device:
//   [farm] gpu g: <n> file(s), status <s>, <t> s, cpus <first>-<last>
#include <sched.h>
#include <spawn.h>
#include <sys/wait.h>
#include <unistd.h>

#include <chrono>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <string>
#include <vector>

extern char **environ;

namespace {

int count_gpus()
{
    // KFD topology: GPU nodes have simd_count > 0 (CPU nodes list 0)
    int n = 0;
    for (int node = 0; node < 64; node++) {
        std::ifstream in("/sys/class/kfd/kfd/topology/nodes/" + std::to_string(node) + "/properties");
        if (!in) break;
        std::string key; long value;
        while (in >> key >> value) if (key == "simd_count" && value > 0) { n++; break; }
    }
    return std::max(n, 1);
}

}  // namespace

int main(int argc, char **argv)
{
    int gpus = 0;
    bool dry = false, bind = true;
    std::vector<std::string> rest;
    for (int i = 1; i < argc; i++) {
        std::string const a = argv[i];
        if (a == "--gpus" && i + 1 < argc) gpus = std::atoi(argv[++i]);
        else if (a == "--dry-run") dry = true;
        else if (a == "--no-bind") bind = false;
        else rest.push_back(a);
    }
    if (rest.size() < 2) {
        std::cerr << "usage: gd_farm [--gpus G] [--dry-run] [--no-bind] <driver> <trajectory>...\n";
        return 1;
    }
    if (gpus <= 0) gpus = count_gpus();
    std::string const driver = rest[0];
    std::vector<std::string> const files(rest.begin() + 1, rest.end());
    int const groups = (int)std::min<std::size_t>((std::size_t)gpus, files.size());
    struct child { pid_t pid = 0; int gpu = 0; std::size_t files = 0; int status = -1; double seconds = 0; int cpu_first = -1, cpu_last = -1; };
    std::vector<child> kids;
    // the CPUs this process may run on, in order: child g inherits share g of `groups` (set on the launcher around the spawn)
    cpu_set_t mine;
    CPU_ZERO(&mine);
    std::vector<int> cpus;
    if (sched_getaffinity(0, sizeof mine, &mine) == 0)
        for (int c = 0; c < CPU_SETSIZE; c++) if (CPU_ISSET(c, &mine)) cpus.push_back(c);
    bind = bind && (int)cpus.size() >= groups && groups > 1;
    auto const t0 = std::chrono::steady_clock::now();
    std::size_t at = 0;
    for (int g = 0; g < groups; g++) {
        std::size_t const n = files.size() / groups + ((std::size_t)g < files.size() % groups ? 1 : 0);
        std::vector<std::string> args = {driver, "--device", std::to_string(g)};
        for (std::size_t k = 0; k < n; k++) args.push_back(files[at + k]);
        at += n;
        std::cerr << "[farm] gpu " << g << ":";
        for (auto const &a : args) std::cerr << ' ' << a;
        std::cerr << '\n';
        if (dry) continue;
        std::vector<char *> av;
        for (auto &a : args) av.push_back(a.data());
        av.push_back(nullptr);
        child k;
        k.gpu = g; k.files = n;
        if (bind) {
            std::size_t const per = cpus.size() / (std::size_t)groups, lo = (std::size_t)g * per, hi = g == groups - 1 ? cpus.size() : lo + per;
            cpu_set_t share;
            CPU_ZERO(&share);
            for (std::size_t c = lo; c < hi; c++) CPU_SET(cpus[c], &share);
            if (sched_setaffinity(0, sizeof share, &share) == 0) { k.cpu_first = cpus[lo]; k.cpu_last = cpus[hi - 1]; }
        }
        int const rc = posix_spawnp(&k.pid, driver.c_str(), nullptr, nullptr, av.data(), environ);
        if (bind) (void)sched_setaffinity(0, sizeof mine, &mine);
        if (rc != 0) { std::cerr << "[farm] cannot start " << driver << ": " << std::strerror(rc) << '\n'; return 127; }
        kids.push_back(k);
    }
    int worst = 0;
    for (std::size_t left = kids.size(); left > 0; left--) {      // in the order they end: each gets its own wall time
        int status = 0;
        pid_t const pid = waitpid(-1, &status, 0);
        if (pid < 0) { worst = std::max(worst, 126); break; }
        int const code = WIFEXITED(status) ? WEXITSTATUS(status) : 128 + (WIFSIGNALED(status) ? WTERMSIG(status) : 0);
        for (auto &k : kids) if (k.pid == pid) { k.status = code; k.seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); }
        worst = std::max(worst, code);
    }
    for (auto const &k : kids) {
        std::cerr << "[farm] gpu " << k.gpu << ": " << k.files << " file(s), status " << k.status << ", " << k.seconds << " s";
        if (k.cpu_first >= 0) std::cerr << ", cpus " << k.cpu_first << '-' << k.cpu_last;
        std::cerr << '\n';
    }
    return worst;
}
