// gd_ab_driver.hpp -- the stage-4 A/B copolymer drivers (periodic box and confining sphere) on libgdyn.
//
// Mirrors `simulation [-s <seed>] <config> <out>` of 4-sim-ab/box/src/simulation (main.cc:24-65,
// simulation_config.hpp:11-26, simulation_config.cc:11-37, simulation_data.cc:10-44, simulation_driver.cc:28-217)
// and of 4-sim-ab/sphere/src (simulation_config.hpp:11-31, simulation_driver.cc:141-230,238-275): same JSON
// keys and defaults, same bead TSV input, same chain detection, same rod initialisation, same log lines and
// output file layout.  The micromd calls are replaced by the C-ABI of include/gdyn.h.
//
// Decisions where the reference leans on micromd internals that are not in the tree:
//  * `md::random_engine` is taken to be std::mt19937_64 (as the stage-3/5 drivers spell it out);
//  * the integrator seed is left at "micromd's default" by the reference (simulation_driver.cc:211-216): 0 here;
//  * softcore_potential<P> (one-argument form) is softcore<P,3>;
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <ctime>
#include <fstream>
#include <iomanip>
#include <iostream>
#include <random>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

#include <json.hpp>   // nlohmann/json single header

#include "../../include/gdyn.h"
#include "gd_ab_store.hpp"

namespace gd_ab {

enum class geometry { box, sphere };

inline void chk(int rc) { if (rc != GD_OK) throw std::runtime_error(gd_last_error()); }

// (de)serialisable parameters: name, type, default -- box: simulation_config.hpp:11-26, sphere: :11-31
#define GD_AB_COMMON_HEAD(X)                 \
    X(double,        a_core_diameter,   0.30) \
    X(double,        b_core_diameter,   0.24) \
    X(double,        a_core_repulsion,  2.0 ) \
    X(double,        b_core_repulsion,  2.0 ) \
    X(double,        bond_spring,       70  ) \
    X(double,        mobility,          1.0 )
#define GD_AB_BOX(X)                         \
    X(double,        box_size,          1.0 )
#define GD_AB_SPHERE(X)                          \
    X(double,        outer_wall_radius,     1.0) \
    X(double,        outer_wall_multiplier, 1.0) \
    X(double,        outer_wall_spring,     1.0) \
    X(double,        inner_wall_radius,     0.0) \
    X(double,        inner_wall_multiplier, 1.0) \
    X(double,        inner_wall_spring,     1.0)
#define GD_AB_COMMON_TAIL(X)                  \
    X(double,        init_bond_length,  0.0 ) \
    X(double,        temperature,       1.0 ) \
    X(double,        timestep,          1e-5) \
    X(long,          steps,             1000) \
    X(long,          logging_interval,  1000) \
    X(long,          sampling_interval, 1000) \
    X(std::string,   beads_filename,    ""  ) \
    X(std::uint64_t, seed,              0   )

struct simulation_config {
    std::string output;
    int device = 0;
#define X(T, var, init) T var = init;
    GD_AB_COMMON_HEAD(X) GD_AB_BOX(X) GD_AB_SPHERE(X) GD_AB_COMMON_TAIL(X)
#undef X
};

template <typename Op>
void foreach_parameter(geometry geo, simulation_config &c, Op op)
{
#define X(T, var, init) op(#var, c.var);
    GD_AB_COMMON_HEAD(X)
    if (geo == geometry::box) { GD_AB_BOX(X) } else { GD_AB_SPHERE(X) }
    GD_AB_COMMON_TAIL(X)
#undef X
}

// entries not listed in the JSON keep their defaults (simulation_config.cc:11-24)
inline void load_simulation_config(geometry geo, std::istream &in, simulation_config &config)
{
    auto const json = nlohmann::json::parse(in);
    foreach_parameter(geo, config, [&](std::string const &name, auto &var) {
        auto node = json.find(name);
        if (node != json.end()) var = node->get<std::decay_t<decltype(var)>>();
    });
}

inline std::string dump_simulation_config(geometry geo, simulation_config const &config)
{
    nlohmann::json json;
    foreach_parameter(geo, const_cast<simulation_config &>(config), [&](std::string const &name, auto &var) { json[name] = var; });
    return json.dump(/*pretty=*/true);      // `true` converts to an indent of 1, as in simulation_config.cc:36
}

struct bead_data { std::string chain; double a_factor = 0, b_factor = 0; };
struct chain_data { std::size_t start = 0, end = 0; };

// simulation_data.cc:10-44
inline std::vector<bead_data> load_beads_data(std::string const &filename)
{
    std::ifstream file{filename};
    std::string header;
    std::getline(file, header);
    if (header != "chain\tA\tB") throw std::runtime_error("unexpected beads data header");
    std::vector<bead_data> beads;
    for (std::string line; std::getline(file, line);) {
        if (line.find_first_not_of(" \t\r") == std::string::npos) continue;
        std::istringstream record{line};
        bead_data bead;
        record >> bead.chain >> bead.a_factor >> bead.b_factor;
        beads.push_back(bead);
    }
    return beads;
}

class simulation_driver {
public:
    simulation_driver(geometry geo, simulation_config const &config)
        : _geo(geo), _config(config), _store(config.output), _random(config.seed)
    {
        _store.save_config(dump_simulation_config(_geo, _config));
        setup_particles();
        setup_forcefield();
    }
    ~simulation_driver() { gd_destroy(_sys); }

    void run() { run_initialization(); run_sampling(); }

private:
    // beads -> particles and chains of consecutive equal chain names (simulation_driver.cc:39-84)
    void setup_particles()
    {
        auto const beads = load_beads_data(_config.beads_filename);
        if (beads.empty()) throw std::runtime_error("no beads in " + _config.beads_filename);
        _n = beads.size();
        std::string cur_chain;
        std::size_t cur_start = 0;
        for (std::size_t i = 0; i < _n; i++) {
            if (beads[i].chain != cur_chain) {
                if (i != cur_start) _chains.push_back({cur_start, i});
                cur_chain = beads[i].chain;
                cur_start = i;
            }
        }
        _chains.push_back({cur_start, _n});

        gd_desc desc{};
        desc.n_beads = (uint32_t)_n; desc.n_replicas = 1; desc.device = _config.device;
        if (_geo == geometry::box) {
            desc.box_kind = GD_BOX_PERIODIC;
            desc.box[0] = desc.box[1] = desc.box[2] = _config.box_size;
        } else {
            desc.box_kind = GD_BOX_OPEN;
        }
        chk(gd_create(&desc, &_sys));
        std::vector<double> a(_n), b(_n), mobility(_n, _config.mobility);
        std::vector<float> ab;
        for (std::size_t i = 0; i < _n; i++) {
            a[i] = beads[i].a_factor; b[i] = beads[i].b_factor;
            ab.push_back(float(beads[i].a_factor)); ab.push_back(float(beads[i].b_factor));
        }
        chk(gd_set_bead_params(_sys, a.data(), b.data(), mobility.data(), nullptr));
        _store.save_beads(ab);
        std::vector<int> ranges;
        for (auto const &c : _chains) { ranges.push_back(int(c.start)); ranges.push_back(int(c.end)); }
        _store.save_chains(ranges);
    }

    void setup_forcefield()
    {
        // short-range mixed repulsion (:93-124); neighbour distance max(sigma_a, sigma_b) is implied by the cutoffs
        gd_pair_softcore pair{};
        pair.eps_a = _config.a_core_repulsion; pair.sigma_a = _config.a_core_diameter; pair.p_a = 2; pair.q_a = 3;
        pair.eps_b = _config.b_core_repulsion; pair.sigma_b = _config.b_core_diameter; pair.p_b = 8; pair.q_b = 3;
        pair.mix = 1;
        chk(gd_set_pair_softcore(_sys, &pair));
        // harmonic chain bonds (:128-141)
        gd_bond_params bond{};
        bond.kind = GD_POT_HARMONIC; bond.k_a = _config.bond_spring;
        for (auto const &c : _chains) chk(gd_add_bond_range(_sys, &bond, (uint32_t)c.start, (uint32_t)c.end, 1));
        if (_geo == geometry::sphere) { setup_forcefield_outer_wall(); setup_forcefield_inner_wall(); }
    }

    // sphere/src/simulation_driver.cc:141-181: wall factors (0,1), half diameters, energies times the multiplier,
    // harmonic restoring force outside; a static sphere is the ellipsoid with three equal, frozen semiaxes
    void setup_forcefield_outer_wall()
    {
        gd_wall wall{};
        wall.eps_a = _config.outer_wall_multiplier * _config.a_core_repulsion; wall.sigma_a = _config.a_core_diameter; wall.p_a = 2; wall.q_a = 3;
        wall.eps_b = _config.outer_wall_multiplier * _config.b_core_repulsion; wall.sigma_b = _config.b_core_diameter; wall.p_b = 8; wall.q_b = 3;
        wall.wall_a_factor = 0; wall.wall_b_factor = 1;
        wall.packing_spring = _config.outer_wall_spring;
        for (int k = 0; k < 3; k++) wall.init_semiaxes[k] = _config.outer_wall_radius;
        chk(gd_set_ellipsoid_wall(_sys, &wall));
    }

    // sphere/src/simulation_driver.cc:184-228: an excluded core (absent below a radius of 1e-6): harmonic push-out inside,
    // the wall-type soft repulsion (factors (0,1), half diameters, energies times the multiplier) outside
    void setup_forcefield_inner_wall()
    {
        if (_config.inner_wall_radius < 1e-6) return;
        gd_inner_sphere wall{};
        wall.radius = _config.inner_wall_radius;
        wall.eps_a = _config.inner_wall_multiplier * _config.a_core_repulsion; wall.sigma_a = _config.a_core_diameter; wall.p_a = 2; wall.q_a = 3;
        wall.eps_b = _config.inner_wall_multiplier * _config.b_core_repulsion; wall.sigma_b = _config.b_core_diameter; wall.p_b = 8; wall.q_b = 3;
        wall.wall_a_factor = 0; wall.wall_b_factor = 1;
        wall.spring = _config.inner_wall_spring;
        chk(gd_set_inner_sphere_wall(_sys, &wall));
    }

    // straight rods with the centroid at a random point (box :147-180; sphere :238-275)
    void run_initialization()
    {
        std::vector<double> xyz(3 * _n);
        for (auto const &chain : _chains) {
            double center[3];
            if (_geo == geometry::box) {
                std::uniform_real_distribution<double> start_coord{0, _config.box_size};
                for (int k = 0; k < 3; k++) center[k] = start_coord(_random);
            } else {
                std::uniform_real_distribution<double> center_coord{-_config.outer_wall_radius, _config.outer_wall_radius};
                do {
                    for (int k = 0; k < 3; k++) center[k] = center_coord(_random);
                } while (std::sqrt(center[0] * center[0] + center[1] * center[1] + center[2] * center[2]) < _config.inner_wall_radius);
            }
            std::normal_distribution<double> normal;
            double dir[3];
            for (int k = 0; k < 3; k++) dir[k] = normal(_random);
            double const inv = 1 / std::sqrt(dir[0] * dir[0] + dir[1] * dir[1] + dir[2] * dir[2]);
            for (int k = 0; k < 3; k++) dir[k] *= inv;
            double delta[3] = {0, 0, 0}, pos[3] = {0, 0, 0};
            for (std::size_t i = chain.start; i < chain.end; i++)
                for (int k = 0; k < 3; k++) {
                    xyz[3 * i + k] = pos[k];
                    delta[k] += pos[k] - center[k];
                    pos[k] += _config.init_bond_length * dir[k];
                }
            for (int k = 0; k < 3; k++) delta[k] /= double(chain.end - chain.start);
            for (std::size_t i = chain.start; i < chain.end; i++)
                for (int k = 0; k < 3; k++) xyz[3 * i + k] -= delta[k];
        }
        chk(gd_set_positions(_sys, xyz.data()));
    }

    static long next_multiple(long step, long interval) { return (step / interval + 1) * interval; }

    // :183-217
    void run_sampling()
    {
        std::clog << "[sim] sampling...\n";
        std::vector<double> xyz(3 * _n);
        auto callback = [&](long step) {
            if (step % _config.logging_interval == 0) {
                double e = 0;
                chk(gd_compute_energy(_sys, GD_TERM_ALL, &e));
                std::time_t const now = std::time(nullptr);
                std::clog << "[sim] " << std::put_time(std::localtime(&now), "%F %T") << '\t' << step << '\t' << "E: " << e / double(_n) << '\n';
            }
            if (step % _config.sampling_interval == 0) {
                chk(gd_get_positions(_sys, xyz.data()));
                _store.save_snapshot(step, xyz.data(), _n);
            }
        };
        chk(gd_begin_phase(_sys, nullptr));
        callback(0);
        gd_run_desc run{};
        run.temperature = _config.temperature; run.timestep = _config.timestep; run.seed = 0; run.noise_mode = GD_NOISE_PHILOX;
        long step = 0;
        while (step < _config.steps) {
            long const next = std::min<long>(_config.steps, std::min(next_multiple(step, _config.logging_interval),
                                                                     next_multiple(step, _config.sampling_interval)));
            run.steps = next - step; chk(gd_run(_sys, &run)); step = next;
            callback(step);
        }
    }

    geometry _geo;
    simulation_config _config;
    gd::ab_store _store;
    std::mt19937_64 _random;
    gd_system *_sys = nullptr;
    std::size_t _n = 0;
    std::vector<chain_data> _chains;
};

// usage: simulation [-s <seed>] [-d <device>] <config> <out>   (main.cc:24-34; -d is an addition)
inline int main_ab(geometry geo, int argc, char **argv)
{
    try {
        simulation_config config;
        bool have_seed = false;
        std::uint64_t seed = 0;
        std::vector<std::string> positional;
        for (int i = 1; i < argc; i++) {
            std::string const arg = argv[i];
            if (arg == "-s" && i + 1 < argc) { seed = std::stoull(argv[++i]); have_seed = true; }
            else if (arg == "-d" && i + 1 < argc) config.device = std::stoi(argv[++i]);
            else if (arg == "-h" || arg == "--help") positional.clear(), i = argc;
            else positional.push_back(arg);
        }
        if (positional.size() != 2) {
            std::cerr << "usage:\n  simulation [-s <seed>] [-d <device>] <config> <out>\n\n"
                         "  <config>    input JSON configuration file\n  <out>       output HDF5 trajectory file\n";
            return 1;
        }
        std::ifstream config_file{positional[0]};
        if (!config_file) throw std::runtime_error("cannot open config file");
        load_simulation_config(geo, config_file, config);
        if (have_seed) config.seed = seed;
        config.output = positional[1];
        simulation_driver sim{geo, config};
        sim.run();
    } catch (std::exception const &e) {
        std::cerr << "error: " << e.what() << '\n';
        return 1;
    }
    return 0;
}

}  // namespace gd_ab
