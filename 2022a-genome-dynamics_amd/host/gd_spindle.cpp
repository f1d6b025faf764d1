// gd_spindle -- the coarse-grained anatelophase (spindle + packing) driver on libgdyn.
//
// Mirrors the reference program `simulation_spindle <trajectory.h5>`
// (5-sim-genome/src/simulation_spindle/: main.cc, simulation_driver.cc:28-310, simulation_driver.hpp): chains of
// ceil(len / init_coarse_graining) beads, uniform softcore<2,3> repulsion, semispring chain bonds, cosine
// bending, a harmonic spindle point source on the three centromere beads of every chain ("spindle" phase)
// plus a semispring packing well on all beads ("packing" phase).  Same input/output file, config keys,
// phases, log lines and snapshot cadence; micromd calls are replaced by the C-ABI of include/gdyn.h.
//
// The rod initialisation draws from std::mt19937_64{spindle_seed} through std::normal_distribution<double>
// exactly as simulation_driver.cc:189-207 does (same generator, same draw order, a fresh distribution object
// per chain), so step-0 positions agree with a reference binary built against the same libstdc++.  The
// Brownian noise is libgdyn's Philox stream (micromd's generator is not reproducible).
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <ctime>
#include <iomanip>
#include <iostream>
#include <random>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/gdyn.h"
#include "gd_config.hpp"
#include "gd_genome_model.hpp"
#include "gd_store.hpp"

namespace {

using gd::chk;

struct chain_range { gd::chromosome_range chromosome; std::size_t start = 0, end = 0, centromere = 0; };

class simulation_driver {
public:
    simulation_driver(gd::trajectory_store &store, int device)
        : _store(store), _config(gd::parse_simulation_config(store.load_config_text())), _random(_config.spindle_seed)
    {
        setup_chains();
        setup_system(device);
    }
    ~simulation_driver() { gd_destroy(_sys); }

    void run()
    {
        run_initialization();
        run_phase("spindle", _config.init_spindle_steps);
        run_phase("packing", _config.init_packing_steps);
    }

private:
    // simulation_driver.cc:46-68
    void setup_chains()
    {
        std::size_t const coarse = _config.init_coarse_graining;
        if (coarse == 0) throw std::runtime_error("init_coarse_graining must be positive");
        std::size_t start = 0;
        for (auto const &chrom : _store.load_chromosomes()) {
            std::size_t const size = chrom.end - chrom.start;
            std::size_t const cen = (chrom.centromere_start + chrom.centromere_end) / 2;
            std::size_t const coarse_size = (size + coarse - 1) / coarse, coarse_cen = (cen - chrom.start) / coarse;
            _chains.push_back({chrom, start, start + coarse_size, start + coarse_cen});
            start += coarse_size;
        }
        _n = start;
    }

    // particles (:71-80), repulsion (:92-105), bonds + bending (:108-135); the two point sources are added when
    // their phase starts (:218, :250)
    void setup_system(int device)
    {
        if (_n == 0) throw std::runtime_error("no chromosomes in the trajectory file");
        gd_desc desc{};
        desc.n_beads = (uint32_t)_n; desc.n_replicas = 1; desc.device = device; desc.box_kind = GD_BOX_OPEN;
        chk(gd_create(&desc, &_sys));
        std::vector<double> mobility(_n, _config.init_mobility);
        chk(gd_set_bead_params(_sys, nullptr, nullptr, mobility.data(), nullptr));
        gd_pair_softcore pair{};
        pair.eps_a = _config.init_bead_repulsion; pair.sigma_a = _config.init_bead_diameter; pair.p_a = 2; pair.q_a = 3;
        pair.p_b = 2; pair.q_b = 3;          // unused second term (eps_b = 0)
        chk(gd_set_pair_softcore(_sys, &pair));
        gd_bond_params bond{};
        bond.kind = GD_POT_SEMISPRING; bond.k_a = _config.init_bond_spring; bond.l_a = _config.init_bond_length;
        for (auto const &chain : _chains) {
            chk(gd_add_bond_range(_sys, &bond, (uint32_t)chain.start, (uint32_t)chain.end, 1));
            chk(gd_add_bending_range(_sys, (uint32_t)chain.start, (uint32_t)chain.end, _config.init_bend_energy, 0));
        }
        _buffer.resize(3 * _n);
    }

    void add_spindle_forcefield()
    {
        std::vector<uint32_t> centromeres;
        for (auto const &chain : _chains) {
            if (!(chain.centromere > chain.start && chain.centromere + 1 < chain.end))     // the reference asserts this
                throw std::runtime_error("centromere of " + chain.chromosome.name + " is at a chain end");
            for (int d = -1; d <= 1; d++) centromeres.push_back((uint32_t)(chain.centromere + d));
        }
        double const point[3] = {_config.init_spindle_point.x, _config.init_spindle_point.y, _config.init_spindle_point.z};
        chk(gd_add_point_source(_sys, GD_POT_HARMONIC, _config.init_spindle_spring, 0, point, centromeres.data(),
                                (uint32_t)centromeres.size()));
    }

    void add_packing_forcefield()
    {
        double const point[3] = {_config.init_spindle_point.x, _config.init_spindle_point.y, _config.init_spindle_point.z};
        chk(gd_add_point_source(_sys, GD_POT_SEMISPRING, _config.init_packing_spring, _config.init_packing_radius, point, nullptr, 0));
    }

    // randomly directed rods (:184-208)
    void run_initialization()
    {
        std::vector<double> xyz(3 * _n);
        double const c0[3] = {_config.init_start_point.x, _config.init_start_point.y, _config.init_start_point.z};
        for (auto const &chain : _chains) {
            std::normal_distribution<double> normal;
            double centroid[3], step[3];
            for (int k = 0; k < 3; k++) centroid[k] = c0[k] + _config.init_start_stddev * normal(_random);
            for (int k = 0; k < 3; k++) step[k] = normal(_random);
            double const norm = std::sqrt(step[0] * step[0] + step[1] * step[1] + step[2] * step[2]);
            for (int k = 0; k < 3; k++) step[k] = _config.init_bond_length * (step[k] * (1 / norm));
            double pos[3];
            for (int k = 0; k < 3; k++) pos[k] = centroid[k] - step[k] * (double)(chain.end - chain.start) / 2;
            for (std::size_t i = chain.start; i < chain.end; i++)
                for (int k = 0; k < 3; k++) { xyz[3 * i + k] = pos[k]; pos[k] += step[k]; }
        }
        chk(gd_set_positions(_sys, xyz.data()));
    }

    static long next_multiple(long step, long interval) { return (step / interval + 1) * interval; }

    // run_spindle_phase / run_packing_phase (:211-272)
    void run_phase(std::string const &phase, long steps)
    {
        _store.set_phase(phase);
        save_chains();
        if (phase == "spindle") add_spindle_forcefield(); else add_packing_forcefield();
        chk(gd_begin_phase(_sys, nullptr));
        auto callback = [&](long step) {
            if (step % _config.init_sampling_interval == 0) {
                chk(gd_get_positions_f32(_sys, _buffer.data(), /*quantize=*/1));
                _store.save_positions(step, _buffer.data(), _n);
            }
            if (step % _config.init_logging_interval == 0) print_progress(phase, step);
        };
        callback(0);
        gd_run_desc run{};
        run.temperature = _config.init_temperature; run.timestep = _config.init_timestep; run.spacestep = _config.init_spacestep;
        run.seed = _random(); run.noise_mode = GD_NOISE_PHILOX; run.flags = 0;
        long step = 0;
        while (step < steps) {
            long const next = std::min<long>(steps, std::min(next_multiple(step, _config.init_sampling_interval),
                                                             next_multiple(step, _config.init_logging_interval)));
            run.steps = next - step; chk(gd_run(_sys, &run)); step = next;
            callback(step);
        }
    }

    void print_progress(std::string const &phase, long step)
    {
        std::time_t const now = std::time(nullptr);
        double e = 0;
        chk(gd_compute_energy(_sys, GD_TERM_ALL, &e));
        std::clog << "[" + phase + "] " << std::put_time(std::localtime(&now), "%F %T") << '\t' << step << '\t'
                  << "E: " << e / (double)_n << '\n';
    }

    // coarse chain table of the phase (:295-309)
    void save_chains()
    {
        std::vector<gd::chromosome_range> chroms;
        for (auto const &chain : _chains) {
            gd::chromosome_range c;
            c.name = chain.chromosome.name; c.start = chain.start; c.end = chain.end;
            chroms.push_back(c);
        }
        _store.save_chromosomes(chroms);
    }

    gd::trajectory_store &_store;
    gd::simulation_config _config;
    std::mt19937_64 _random;      // rod normals, then the spindle-phase seed, then the packing-phase seed
    std::vector<chain_range> _chains;
    gd_system *_sys = nullptr;
    std::size_t _n = 0;
    std::vector<float> _buffer;
};

}  // namespace

int main(int argc, char **argv)
{
    if (argc < 2 || argc > 3) {
        std::cerr << "usage: gd_spindle <trajectory> [device]\n";
        return 1;
    }
    try {
        gd::trajectory_store store{argv[1]};
        simulation_driver driver{store, argc == 3 ? std::stoi(argv[2]) : 0};
        driver.run();
    } catch (std::exception const &e) {
        std::cerr << "error: " << e.what() << '\n';
        return 1;
    }
    return 0;
}
