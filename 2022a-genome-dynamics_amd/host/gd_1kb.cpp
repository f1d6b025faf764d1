// gd_1kb -- the 1 kb-resolution chromatin loop-formation driver on libgdyn.
//
// Mirrors the reference program `main [-hCos] <config>` of 3-sim-1kb/src/simulation (main.cpp:40-186,
// simulation.cpp:37-305, topology.cpp:6-21, loops.cpp:12-92, glues.cpp:4-17, inits.cpp:10-22,
// inits/box_initializer.cpp:18-38, inits/utils.hpp:9-47, store.cpp:17-58): same JSON configuration, same
// command-line options, same log lines, same output datasets.  The Brownian dynamics runs on the device in chunks
// that end at the next logging / sampling / loop-update / glue-update step; the loop and glue lists are the
// re-uploadable pair slots of the C-ABI (the reference's custom md::forcefield subclasses, forces/loop_forcefield.cpp,
// forces/glue_forcefield.cpp); loop extrusion and glue kinetics stay on the host on std::mt19937_64
// (gd_1kb_kinetics.hpp), the glue candidate search uses the device cell list (gd_search_pairs).
//
// Options beyond the reference's: -d <device>; --trace <dir> (test support): writes <dir>/init.f64 (initial positions) and
// <dir>/trace.txt with the integrator seed and every loop / glue list uploaded, so a run can be replayed call by call;
// --auto-skin: let the library select the list width from measured chunk times (gd_tuning.auto_skin).  Off by default: the
// selection changes cost only, but it reads a clock -- the cell decomposition, and with it the fp32 summation order of a
// trajectory, would differ from run to run, where the reference gives one trajectory per seed.  (--fixed-skin, the former
// spelling of the default, is still accepted.)
// The program reads no environment variable.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <memory>
#include <optional>
#include <random>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/gdyn.h"
#include "gd_1kb_config.hpp"
#include "gd_1kb_kinetics.hpp"
#include "gd_1kb_store.hpp"

namespace {

using namespace gd1kb;

void chk(int rc) { if (rc != GD_OK) throw std::runtime_error(gd_last_error()); }

struct chain_assignment { std::size_t start = 0, end = 0; chain_config const *config = nullptr; };

std::vector<chain_assignment> make_chain_assignments(simulation_config const &config)      // topology.cpp:6-21
{
    std::vector<chain_assignment> out;
    std::size_t offset = 0;
    for (auto const &chain : config.chains) {
        out.push_back({offset, offset + chain.length, &chain});
        offset += chain.length;
    }
    return out;
}

std::mt19937_64 make_random(std::uint64_t seed)      // simulation.cpp:29-34
{
    std::seed_seq seq{seed};
    return std::mt19937_64{seq};
}

// one virtual lattice over the concatenated chains; boundaries stop factors from hopping across (loops.cpp:12-92)
gd::loop_extruder make_loop_extruder(simulation_config const &config, std::vector<chain_assignment> const &chains)
{
    std::size_t length = 0, max_loops = 0;
    for (auto const &c : chains) length += c.config->length;
    if (config.loop.max_loops) max_loops = *config.loop.max_loops;
    else for (auto const &c : chains) max_loops += c.config->loaded_loops.size();
    gd::loop_extruder loops{length, max_loops};
    loops.set_forward_speed(config.loop.forward_speed);
    loops.set_backward_speed(config.loop.backward_speed);
    loops.set_loading_rate(config.loop.loading_rate_density * double(length));
    loops.set_unloading_rate(config.loop.unloading_rate);
    if (config.loop.crossing_rate) loops.set_crossing_rate(*config.loop.crossing_rate);
    for (auto const &c : chains) {
        for (auto pos : c.config->forward_boundaries) {        // factors stall on the right neighbour
            loops.add_boundary(c.start + pos);
            if (c.start + pos + 1 < c.end) loops.set_site_detachability(c.start + pos + 1, config.loop.convergent_detachability);
        }
        for (auto pos : c.config->backward_boundaries) {       // ... and on the left neighbour
            loops.add_boundary(c.start + pos);
            if (c.start + pos >= c.start + 1) loops.set_site_detachability(c.start + pos - 1, config.loop.convergent_detachability);
        }
        for (auto pos : c.config->roadblocks) loops.set_site_attachability(c.start + pos, config.loop.roadblock_attachability);
    }
    for (auto const &c : chains)
        for (auto pos : c.config->loaded_loops) loops.load_loop(c.start + pos);
    return loops;
}

class simulation {
public:
    simulation(simulation_config const &config, int device, std::string const &trace_dir = "", bool auto_skin = false)
        : _config(config), _random(make_random(config.sampling.random_seed)), _store(config.sampling.output_filename),
          _chains(make_chain_assignments(_config)), _loops(make_loop_extruder(_config, _chains)),
          _glues(_config.glue.max_glues, _config.glue.glue_distance, _config.glue.glue_binding_rate, _config.glue.glue_unbinding_rate,
                 _config.chain.box_size)
    {
        for (auto const &c : _chains) _n += c.config->length;
        if (_n == 0) throw std::runtime_error("no monomers: the configuration defines no chains");
        _auto_skin = auto_skin;
        if (!trace_dir.empty()) { _trace_dir = trace_dir; _trace.open(_trace_dir + "/trace.txt"); }
        setup_system(device);
        std::vector<int> ranges;
        for (auto const &c : _chains) { ranges.push_back(int(c.start)); ranges.push_back(int(c.end)); }
        _store.save_metadata(format_simulation_config(_config), _config.config_text, ranges);
    }
    ~simulation() { gd_destroy(_sys); }

    void run()
    {
        initialize_particles();
        if (_config.sampling.loop_preloading) _loops.preload(_random);
        run_simulation();
    }

private:
    // particles + force fields (simulation.cpp:57-186)
    void setup_system(int device)
    {
        auto const &ch = _config.chain;
        gd_desc desc{};
        desc.n_beads = (uint32_t)_n; desc.n_replicas = 1; desc.device = device; desc.box_kind = GD_BOX_PERIODIC;
        desc.box[0] = desc.box[1] = desc.box[2] = ch.box_size;
        chk(gd_create(&desc, &_sys));
        {   // the list width: the library's state-based default, or (--auto-skin) selected from measured chunk times
            gd_tuning tune{};
            tune.adapt_interval = 1; tune.auto_skin = _auto_skin ? 1 : 0;
            chk(gd_set_tuning(_sys, &tune));
        }
        std::vector<double> mobility(_n, ch.monomer_mobility), bending(_n, ch.bending_energy);
        for (auto const &c : _chains)
            for (auto const &block : c.config->blocks)
                for (std::size_t i = block.start; i < block.end; i++)
                    if (block.bending_energy) bending.at(c.start + i) = *block.bending_energy;
        chk(gd_set_bead_params(_sys, nullptr, nullptr, mobility.data(), bending.data()));
        gd_pair_softcore pair{};      // softcore<2,3>{rep} + softcore<8,3>{-attr}, unmixed
        pair.eps_a = ch.repulsive_energy; pair.sigma_a = ch.repulsive_diameter; pair.p_a = 2; pair.q_a = 3;
        pair.eps_b = ch.attractive_energy * -1; pair.sigma_b = ch.attractive_diameter; pair.p_b = 8; pair.q_b = 3;
        chk(gd_set_pair_softcore(_sys, &pair));
        gd_bond_params bond{};
        bond.kind = GD_POT_SPRING; bond.k_a = ch.bond_spring; bond.l_a = ch.bond_length;
        for (auto const &c : _chains) {
            chk(gd_add_bond_range(_sys, &bond, (uint32_t)c.start, (uint32_t)c.end, 1));
            chk(gd_add_bending_range(_sys, (uint32_t)c.start, (uint32_t)c.end, 0.0, /*per_bead=*/1));
        }
        _loop_bond.kind = GD_POT_SPRING; _loop_bond.k_a = _config.loop.bond_spring; _loop_bond.l_a = ch.repulsive_diameter;
        _glue_bond.kind = GD_POT_SOFTCORE; _glue_bond.k_a = -_config.glue.glue_energy; _glue_bond.l_a = _config.glue.glue_distance;
        _glue_bond.p = 8; _glue_bond.q = 3; _glue_bond.minimum_image = 1;
    }

    // chain centroids uniform in the box, one continuous random walk over all monomers, then every chain is shifted
    // onto its centroid (inits/box_initializer.cpp:18-38, inits/utils.hpp:9-47)
    void initialize_particles()
    {
        auto const &ch = _config.chain;
        std::vector<double> centroids;
        for (std::size_t c = 0; c < _chains.size(); c++) {
            std::uniform_real_distribution<double> coord{0, ch.box_size};
            for (int k = 0; k < 3; k++) centroids.push_back(coord(_random));
        }
        _xyz.assign(3 * _n, 0.0);
        double walk[3] = {0, 0, 0};
        for (std::size_t i = 0; i < _n; i++) {
            for (int k = 0; k < 3; k++) _xyz[3 * i + k] = walk[k];
            std::normal_distribution<double> normal;
            double d[3] = {normal(_random), normal(_random), normal(_random)};
            double const inv = 1 / std::sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
            for (int k = 0; k < 3; k++) walk[k] += ch.initial_bond_length * (d[k] * inv);
        }
        for (std::size_t c = 0; c < _chains.size(); c++) {
            auto const &chain = _chains[c];
            double offset[3] = {0, 0, 0};
            for (std::size_t i = chain.start; i < chain.end; i++)
                for (int k = 0; k < 3; k++) offset[k] += _xyz[3 * i + k] - centroids[3 * c + k];
            for (int k = 0; k < 3; k++) offset[k] /= double(chain.end - chain.start);
            for (std::size_t i = chain.start; i < chain.end; i++)
                for (int k = 0; k < 3; k++) _xyz[3 * i + k] -= offset[k];
        }
        chk(gd_set_positions(_sys, _xyz.data()));
        if (!_trace_dir.empty()) {
            std::ofstream out(_trace_dir + "/init.f64", std::ios::binary);
            out.write(reinterpret_cast<char const *>(_xyz.data()), (std::streamsize)(_xyz.size() * sizeof(double)));
        }
    }

    void upload_loops(long step)
    {
        // a freshly loaded factor holds a zero-length loop (start == end): no force, the constant energy K b^2 / 2 of
        // the spring at r = 0; the pair slot takes distinct beads only, so that constant is added in show_progress
        std::vector<uint32_t> pairs;
        _zero_length_loops = 0;
        for (auto const &l : _loops.loops()) {
            if (!l.id) continue;
            if (l.start == l.end) { _zero_length_loops++; continue; }
            pairs.push_back((uint32_t)l.start); pairs.push_back((uint32_t)l.end);
        }
        chk(gd_set_dynamic_pairs(_sys, 0, &_loop_bond, pairs.data(), (uint32_t)(pairs.size() / 2)));
        trace("loops", step, pairs);
    }

    void upload_glues(long step)
    {
        std::vector<std::pair<uint32_t, uint32_t>> sorted;
        for (auto const &g : _glues.pairs()) sorted.push_back({g.i, g.j});
        std::sort(sorted.begin(), sorted.end());
        std::vector<uint32_t> pairs;
        for (auto const &g : sorted) { pairs.push_back(g.first); pairs.push_back(g.second); }
        chk(gd_set_dynamic_pairs(_sys, 1, &_glue_bond, pairs.data(), (uint32_t)(pairs.size() / 2)));
        trace("glues", step, pairs);
    }

    void trace(char const *what, long step, std::vector<uint32_t> const &pairs)
    {
        if (!_trace.is_open()) return;
        _trace << what << ' ' << step << ' ' << pairs.size() / 2;
        for (auto v : pairs) _trace << ' ' << v;
        _trace << '\n' << std::flush;
    }

    // simulation.cpp:245-260
    void step_loops(long step)
    {
        auto const &s = _config.sampling;
        double const leap = s.timestep * double(s.loop_update_interval);
        if (!s.clear_loops_at || step < s.clear_loops_at) _loops.step(leap, _random);
        if (step + 1 == s.clear_loops_at) _loops.clear();
        upload_loops(step);
    }

    // simulation.cpp:263-270; candidates come from the device cell list
    void step_glues(long step)
    {
        if (!_glues.enabled()) return;
        auto const &s = _config.sampling;
        double const leap = s.timestep * double(s.glue_update_interval);
        chk(gd_get_positions(_sys, _xyz.data()));
        uint64_t n = 0;
        chk(gd_search_pairs(_sys, 0, _glues.reach(), nullptr, 0, &n));
        std::vector<uint32_t> candidates(2 * n);
        if (n) chk(gd_search_pairs(_sys, 0, _glues.reach(), candidates.data(), n, &n));
        _glues.update(leap, _xyz.data(), candidates, _random);
        upload_glues(step);
    }

    // "step \t E: .. \t L: .. \t G: .." (simulation.cpp:273-296)
    void show_progress(long step)
    {
        double e = 0;
        chk(gd_compute_energy(_sys, GD_TERM_ALL, &e));
        e += double(_zero_length_loops) * 0.5 * _loop_bond.k_a * _loop_bond.l_a * _loop_bond.l_a;
        double const n = double(_n);
        std::clog << step << '\t' << "E: " << e / n << '\t' << "L: " << double(_loops.loaded()) / n << '\t'
                  << "G: " << double(_glues.size()) / n << '\n';
    }

    void save_sample()
    {
        chk(gd_get_positions(_sys, _xyz.data()));
        std::vector<long long> loops;
        for (auto const &l : _loops.loops()) { loops.push_back((long long)l.start); loops.push_back((long long)l.end); loops.push_back((long long)l.id); }
        _store.save_snapshot(_xyz.data(), _n, loops);
    }

    static long next_multiple(long step, long interval) { return (step / interval + 1) * interval; }

    // simulation.cpp:212-242
    void run_simulation()
    {
        auto const &s = _config.sampling;
        for (long interval : {s.logging_interval, s.sampling_interval, s.loop_update_interval, s.glue_update_interval})
            if (interval <= 0) throw std::runtime_error("intervals must be positive");
        auto callback = [&](long step) {
            if (step % s.logging_interval == 0) show_progress(step);
            if (step % s.sampling_interval == 0) save_sample();
            if (step % s.loop_update_interval == 0) step_loops(step);
            if (step % s.glue_update_interval == 0) step_glues(step);
        };
        chk(gd_begin_phase(_sys, nullptr));
        upload_loops(-1);                 // the initially loaded / preloaded loops act from the first energy evaluation on
        callback(0);
        gd_run_desc run{};
        run.temperature = s.temperature; run.timestep = s.timestep; run.seed = _random(); run.noise_mode = GD_NOISE_PHILOX;
        if (_trace.is_open()) _trace << "seed " << run.seed << '\n' << std::flush;
        long step = 0;
        while (step < s.steps) {
            long const next = std::min<long>(s.steps, std::min({next_multiple(step, s.logging_interval), next_multiple(step, s.sampling_interval),
                                                               next_multiple(step, s.loop_update_interval), next_multiple(step, s.glue_update_interval)}));
            run.steps = next - step; chk(gd_run(_sys, &run)); step = next;
            callback(step);
        }
    }

    simulation_config _config;
    std::mt19937_64 _random;         // shared by the initialiser, loop preloading, the integrator seed and all kinetics
    history_store _store;
    std::vector<chain_assignment> _chains;
    gd::loop_extruder _loops;
    gd::glue_binder _glues;
    gd_system *_sys = nullptr;
    std::size_t _n = 0, _zero_length_loops = 0;
    gd_bond_params _loop_bond{}, _glue_bond{};
    std::vector<double> _xyz;
    std::string _trace_dir;
    std::ofstream _trace;
    bool _auto_skin = false;
};

void show_usage()
{
    std::cerr << "Loop formation simulator\n"
                 "usage: main [-hCosd] <config>\n\n"
                 "  <config>     JSON file specifying simulation parameters\n\n"
                 "options:\n"
                 "  -C <config>  override chain definitions (config 'chains' key) by additional JSON file\n"
                 "  -o <output>  override output HDF5 filename (config 'output_filename' key)\n"
                 "  -s <seed>    override random seed (config 'random_seed' key)\n"
                 "  -d <device>  GPU index (default 0)\n"
                 "  --auto-skin  select the neighbour-list width from measured step times (faster on some models; the\n"
                 "               trajectory of a seed then depends on timing -- off by default)\n"
                 "  --trace <dir> write the initial positions and every uploaded loop / glue list to <dir> (replay support)\n"
                 "  -h           print this usage message and exit\n\n";
}

std::string load_text(std::string const &filename)
{
    std::ifstream file{filename};
    std::string text;
    if (!std::getline(file, text, '\0')) throw std::runtime_error{"failed to load config file"};
    return text;
}

}  // namespace

int main(int argc, char **argv)
{
    try {
        std::optional<std::string> chains_filename, output_filename;
        std::optional<std::uint64_t> seed;
        std::vector<std::string> positional;
        int device = 0;
        std::string trace_dir;
        bool auto_skin = false;
        for (int i = 1; i < argc; i++) {
            std::string const arg = argv[i];
            auto value = [&]() -> std::string { if (i + 1 >= argc) throw std::runtime_error{"bad option"}; return argv[++i]; };
            if (arg == "-h") { show_usage(); return 0; }
            else if (arg == "-C") chains_filename = value();
            else if (arg == "-o") output_filename = value();
            else if (arg == "-s") seed = std::stoull(value());
            else if (arg == "-d") device = std::stoi(value());
            else if (arg == "--trace") trace_dir = value();
            else if (arg == "--auto-skin") auto_skin = true;
            else if (arg == "--fixed-skin") auto_skin = false;
            else if (arg.size() > 1 && arg[0] == '-') throw std::runtime_error{"bad option"};
            else positional.push_back(arg);
        }
        if (positional.size() != 1) throw std::runtime_error{"config file is not specified"};
        simulation_config config;
        {
            auto const text = load_text(positional[0]);
            try {
                config = parse_simulation_config(text);
            } catch (std::exception const &err) {
                throw std::runtime_error{"failed to parse config file - " + std::string{err.what()}};
            }
        }
        if (chains_filename) {
            auto const text = load_text(*chains_filename);
            try {
                config.chains = parse_chains_config(text);
            } catch (std::exception const &err) {
                throw std::runtime_error{"failed to parse chains config file - " + std::string{err.what()}};
            }
        }
        if (output_filename) config.sampling.output_filename = *output_filename;
        if (seed) config.sampling.random_seed = *seed;
        simulation{config, device, trace_dir, auto_skin}.run();
        return 0;
    } catch (std::exception const &err) {
        std::cerr << "error: " << err.what() << '\n';
        return 1;
    }
}
