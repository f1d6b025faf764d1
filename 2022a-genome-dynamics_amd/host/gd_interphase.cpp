// gd_interphase -- the 100 kb whole-genome relaxation + interphase driver on libgdyn.
//
// Mirrors the reference program `simulation_interphase <trajectory.h5>`
// (5-sim-genome/src/simulation_interphase/: main.cc:15-27, simulation_driver.cc:15-58,
// simulation_driver_particles.cc:8-36, simulation_driver_forcefield.cc:8-235,
// simultion_driver_relaxation.cc:8-46, simulation_driver_interphase.cc:8-80): same input/output file, same
// config keys, same phases, same log lines, same snapshot cadence.  The micromd calls are replaced by the
// C-ABI of include/gdyn.h; the per-step callback state (time, scales, wall ODE) advances on the device and the
// host only intervenes at logging / sampling / contact-map steps.
//
// Differences from the reference, by construction: fp32 device arithmetic and a Philox noise stream (micromd's
// generator is not reproducible); `spacestep` must be 0; the softwell droplet force
// (nucleolus_droplet_energy != 0) uses a documented choice of micromd's potential form (include/gdyn.h).
#include <algorithm>
#include <array>
#include <cmath>
#include <cstdint>
#include <ctime>
#include <iomanip>
#include <iostream>
#include <map>
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

// Time-integrated contact map (simulation_interphase/contact_map.cc:31-91): unique pairs (i<j) within the
// contact distance are counted at every update; accumulate() lists (i, j, count) in row-major order.
class contact_map {
public:
    void set_contact_distance(double d) { _distance = d; }
    void clear() { _counts.clear(); }
    void update(gd_system *sys)
    {
        if (!(_distance > 0)) return;     // the reference's default distance is 0 until the first callback set it
        uint64_t n = 0;
        chk(gd_search_pairs(sys, 0, _distance, nullptr, 0, &n));
        _buffer.resize(2 * n);
        if (n) chk(gd_search_pairs(sys, 0, _distance, _buffer.data(), n, &n));
        for (uint64_t k = 0; k < n; k++) _counts[{_buffer[2 * k], _buffer[2 * k + 1]}] += 1;
    }
    std::vector<std::array<std::uint32_t, 3>> accumulate() const
    {
        std::vector<std::array<std::uint32_t, 3>> out;
        out.reserve(_counts.size());
        for (auto const &kv : _counts) out.push_back({kv.first.first, kv.first.second, kv.second});
        return out;
    }
private:
    double _distance = 0;
    std::map<std::pair<std::uint32_t, std::uint32_t>, std::uint32_t> _counts;
    std::vector<std::uint32_t> _buffer;
};

class simulation_driver {
public:
    simulation_driver(gd::trajectory_store &store, int device)
        : _store(store), _config(gd::parse_simulation_config(store.load_config_text())), _random(_config.interphase_seed)
    {
        // compatibility defaults of older runs (simulation_driver.cc:20-29)
        auto set_default = [](double &var, double def) { if (var == 0) var = def; };
        set_default(_config.a_core_bond_spring, _config.chromatin_bond_spring);
        set_default(_config.a_core_bond_length, _config.chromatin_bond_length);
        set_default(_config.b_core_bond_spring, _config.chromatin_bond_spring);
        set_default(_config.b_core_bond_length, _config.chromatin_bond_length);
        setup(device);
    }
    ~simulation_driver() { gd_destroy(_sys); }

    void run() { run_relaxation(); run_simulation(); }

private:
    void setup(int device)
    {
        _chromosomes = _store.load_chromosomes();
        _sys = gd::build_genome_system(_store, _config, device, /*loop_bonds=*/true, /*mixed_chain_bonds=*/true, _n);
        // setup_context (simulation_driver.cc:43-51)
        _context = gd::context{};
        _context.wall_semiaxes[0] = _config.wall_init_semiaxes.x; _context.wall_semiaxes[1] = _config.wall_init_semiaxes.y;
        _context.wall_semiaxes[2] = _config.wall_init_semiaxes.z;
        _context.bead_scale = _config.bead_scale_init; _context.bond_scale = _config.bond_scale_init;
        _buffer.resize(3 * _n);
    }

    void print_progress(char const *phase, long step)
    {
        std::time_t const now = std::time(nullptr);
        double const radius = std::cbrt(_context.wall_semiaxes[0] * _context.wall_semiaxes[1] * _context.wall_semiaxes[2]);
        std::clog << "[" << phase << "] " << std::put_time(std::localtime(&now), "%F %T") << '\t' << step << '\t'
                  << "t: " << _context.time << '\t' << "R: " << radius << '\t' << "E: " << _context.mean_energy << '\n';
    }

    void mean_energy()
    {
        double e = 0;
        chk(gd_compute_energy(_sys, GD_TERM_ALL, &e));
        _context.mean_energy = e / (double)_n;
    }

    void save_snapshot(long step)
    {
        chk(gd_get_positions_f32(_sys, _buffer.data(), /*quantize=*/1));      // 16 fractional bits, rounded on the device
        _store.save_positions(step, _buffer.data(), _n);
        _store.save_context(step, _context);
    }

    // advance to `target`, stopping one step early to capture the context the reference's callback(target) sees:
    // its log/snapshot use the scales and semiaxes left by callback(target-1) (they are updated at the END of a callback)
    void advance(gd_run_desc &run, long &step, long target)
    {
        if (target - step > 1) { run.steps = target - step - 1; chk(gd_run(_sys, &run)); }
        gd_context ctx;
        chk(gd_get_context(_sys, 0, &ctx));
        _context.bead_scale = ctx.bead_scale; _context.bond_scale = ctx.bond_scale;
        std::copy(ctx.semiaxes, ctx.semiaxes + 3, _context.wall_semiaxes);
        if (target > step) { run.steps = 1; chk(gd_run(_sys, &run)); }
        step = target;
    }

    void run_relaxation()
    {
        _store.set_phase("relaxation");
        auto const init = _store.load_positions(0);
        if (init.size() != _n) throw std::runtime_error("relaxation/0/positions has the wrong number of beads");
        std::vector<double> xyz(3 * _n);
        for (std::size_t i = 0; i < _n; i++) for (int k = 0; k < 3; k++) xyz[3 * i + k] = init[i][k];
        chk(gd_set_positions(_sys, xyz.data()));
        chk(gd_begin_phase(_sys, _context.wall_semiaxes));
        auto callback = [&](long step) {
            bool const logging = step % _config.relaxation_logging_interval == 0, sampling = step % _config.relaxation_sampling_interval == 0;
            if (logging || sampling) mean_energy();
            if (logging) print_progress("relax", step);
            if (sampling) save_snapshot(step);
        };
        callback(0);
        gd_run_desc run{};
        run.temperature = _config.relaxation_temperature; run.timestep = _config.relaxation_timestep;
        run.spacestep = _config.relaxation_spacestep; run.seed = _random(); run.noise_mode = GD_NOISE_PHILOX; run.flags = 0;
        long step = 0;
        while (step < _config.relaxation_steps) {
            long const next = std::min<long>(_config.relaxation_steps, std::min(next_multiple(step, _config.relaxation_logging_interval),
                                                                                next_multiple(step, _config.relaxation_sampling_interval)));
            run.steps = next - step; chk(gd_run(_sys, &run)); step = next;
            callback(step);
        }
    }

    static long next_multiple(long step, long interval) { return (step / interval + 1) * interval; }

    void run_simulation()
    {
        _store.set_phase("interphase");
        double const dt = _config.interphase_timestep;
        chk(gd_begin_phase(_sys, _context.wall_semiaxes));       // step = 0, time = 0
        gd_context last;
        chk(gd_get_context(_sys, 0, &last));
        double reaction[3] = {last.axial_reaction[0], last.axial_reaction[1], last.axial_reaction[2]};

        // host part of callback(step): everything except the state updates that run on the device
        auto observe = [&](long step) {
            _context.time = (double)step * dt;
            bool const logging = step % _config.interphase_logging_interval == 0, sampling = step % _config.interphase_sampling_interval == 0;
            long const frame = step / _config.interphase_sampling_interval;
            if (logging || sampling) mean_energy();
            if (logging) print_progress("inter", step);
            if (sampling) save_snapshot(step);
            if (step % _config.contactmap_update_interval == 0) _contacts.update(_sys);
            if (sampling && frame % _config.contactmap_thinning_rate == 0) { _store.save_contacts(step, _contacts.accumulate()); _contacts.clear(); }
        };

        // callback(0): observation, then update_bead_scale() and update_wall_semiaxes() on the host
        // (simulation_driver_interphase.cc:42-43,59-80); the packing reaction is that of the last force evaluation
        observe(0);
        _context.bead_scale = 1 - (1 - _config.bead_scale_init) * std::exp(-0.0 / _config.bead_scale_tau);
        _context.bond_scale = 1 - (1 - _config.bond_scale_init) * std::exp(-0.0 / _config.bond_scale_tau);
        _contacts.set_contact_distance(_config.contactmap_distance * _context.bead_scale);
        double const spring[3] = {_config.wall_semiaxes_spring.x, _config.wall_semiaxes_spring.y, _config.wall_semiaxes_spring.z};
        for (int k = 0; k < 3; k++)
            _context.wall_semiaxes[k] += dt * _config.wall_mobility * (reaction[k] - spring[k] * _context.wall_semiaxes[k]);
        chk(gd_set_context(_sys, 0, 0, _context.bead_scale, _context.bond_scale, _context.wall_semiaxes));

        gd_run_desc run{};
        run.temperature = _config.interphase_temperature; run.timestep = dt; run.spacestep = _config.interphase_spacestep;
        run.seed = _random(); run.noise_mode = GD_NOISE_PHILOX; run.flags = GD_RUN_UPDATE_SCALES | GD_RUN_WALL_DYNAMICS;
        long step = 0;
        while (step < _config.interphase_steps) {
            long const next = std::min<long>(_config.interphase_steps,
                                             std::min({next_multiple(step, _config.interphase_logging_interval),
                                                       next_multiple(step, _config.interphase_sampling_interval),
                                                       next_multiple(step, _config.contactmap_update_interval)}));
            advance(run, step, next);
            observe(step);
            // the contact distance for later updates is the one set at the end of this callback
            gd_context ctx;
            chk(gd_get_context(_sys, 0, &ctx));
            _contacts.set_contact_distance(_config.contactmap_distance * ctx.bead_scale);
        }
    }

    gd::trajectory_store &_store;
    gd::simulation_config _config;
    gd::context _context;
    contact_map _contacts;
    std::mt19937_64 _random;     // 1st draw: relaxation seed, 2nd: interphase seed
    gd_system *_sys = nullptr;
    std::vector<gd::chromosome_range> _chromosomes;
    std::size_t _n = 0;
    std::vector<float> _buffer;
};

}  // namespace

int main(int argc, char **argv)
{
    if (argc < 2 || argc > 3) {
        std::cerr << "usage: gd_interphase <trajectory> [device]\n";
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
