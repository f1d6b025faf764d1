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
#include <spawn.h>
#include <sys/wait.h>
#include <unistd.h>

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
    bool dry = false;
    std::vector<std::string> rest;
    for (int i = 1; i < argc; i++) {
        std::string const a = argv[i];
        if (a == "--gpus" && i + 1 < argc) gpus = std::atoi(argv[++i]);
        else if (a == "--dry-run") dry = true;
        else rest.push_back(a);
    }
    if (rest.size() < 2) {
        std::cerr << "usage: gd_farm [--gpus G] [--dry-run] <driver> <trajectory>...\n";
        return 1;
    }
    if (gpus <= 0) gpus = count_gpus();
    std::string const driver = rest[0];
    std::vector<std::string> const files(rest.begin() + 1, rest.end());
    int const groups = (int)std::min<std::size_t>((std::size_t)gpus, files.size());
    std::vector<pid_t> pids;
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
        pid_t pid = 0;
        int const rc = posix_spawnp(&pid, driver.c_str(), nullptr, nullptr, av.data(), environ);
        if (rc != 0) { std::cerr << "[farm] cannot start " << driver << ": " << std::strerror(rc) << '\n'; return 127; }
        pids.push_back(pid);
    }
    int worst = 0;
    for (pid_t pid : pids) {
        int status = 0;
        if (waitpid(pid, &status, 0) < 0) { worst = std::max(worst, 126); continue; }
        int const code = WIFEXITED(status) ? WEXITSTATUS(status) : 128 + (WIFSIGNALED(status) ? WTERMSIG(status) : 0);
        if (code != 0) std::cerr << "[farm] process " << pid << " ended with status " << code << '\n';
        worst = std::max(worst, code);
    }
    return worst;
}
