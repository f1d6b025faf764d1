// gd_fine_sampling -- deterministic fine-time-step continuation of an interphase trajectory on libgdyn.
//
// Mirrors the reference program `simulation_fine_sampling <trajectory.h5>`
// (5-sim-genome/src/simulation_fine_sampling/: simulation_driver.cc:15-58, simulation_driver_forcefield.cc:8-184,
// simulation_driver_interphase.cc:8-62): restart from the interphase snapshot of step 700000 (positions + context),
// bead/bond scales forced to 1, temperature 0, time step 1e-7, 100 000 steps sampled every 100 steps into the
// phase "fine_sampling"; the force field is the interphase one without the (i,i+2) loop bonds and with unmixed
// chromatin bonds; only the wall ODE is advanced by the callback.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <ctime>
#include <future>
#include <iomanip>
#include <iostream>
#include <memory>
#include <random>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/gdyn.h"
#include "gd_config.hpp"
#include "gd_genome_model.hpp"
#include "gd_async_io.hpp"
#include "gd_store.hpp"

namespace {

using gd::chk;

class simulation_driver {
public:
    // restart_step is 700000 in the reference (simulation_driver.cc:45); tests restart from shorter trajectories
    simulation_driver(gd::trajectory_store &store, int device, long restart_step, long steps_override = 0)
        : _store(store), _config(gd::parse_simulation_config(store.load_config_text())),
          _random(_config.interphase_seed ^ std::uint64_t(700000))
    {
        _sys = gd::build_genome_system(_store, _config, device, /*loop_bonds=*/false, /*mixed_chain_bonds=*/false, _n);
        // the hard-wired overrides of simulation_driver.cc:30-34
        _config.interphase_temperature = 0;
        _config.interphase_sampling_interval = 100;
        _config.interphase_steps = 1000 * 100;
        _config.interphase_timestep = 1e-5 / 100;
        if (steps_override > 0) _config.interphase_steps = steps_override;      // --steps: shorter runs (tests, timing)
        setup_context(restart_step);
    }
    ~simulation_driver() { gd_destroy(_sys); }

    void run() { run_simulation(); _writer.drain(); }

private:
    // simulation_driver.cc:38-57
    void setup_context(long step)
    {
        _store.set_phase("interphase");
        auto const init = _store.load_positions(step);
        if (init.size() != _n) throw std::runtime_error("interphase snapshot has the wrong number of beads");
        std::vector<double> xyz(3 * _n);
        for (std::size_t i = 0; i < _n; i++) for (int k = 0; k < 3; k++) xyz[3 * i + k] = init[i][k];
        chk(gd_set_positions(_sys, xyz.data()));
        _context = _store.load_context(step);
        _context.bead_scale = 1;
        _context.bond_scale = 1;
        _buffer.resize(3 * _n);
    }

    void print_progress(char const *phase, long step)
    {
        std::time_t const now = std::time(nullptr);
        double const radius = std::cbrt(_context.wall_semiaxes[0] * _context.wall_semiaxes[1] * _context.wall_semiaxes[2]);
        std::clog << "[" << phase << "] " << std::put_time(std::localtime(&now), "%F %T") << '\t' << step << '\t'
                  << "t: " << _context.time << '\t' << "R: " << radius << '\t' << "E: " << _context.mean_energy << '\n';
    }

    static long next_multiple(long step, long interval) { return (step / interval + 1) * interval; }

    // simulation_driver_interphase.cc:8-62
    void run_simulation()
    {
        _store.set_phase("fine_sampling");
        double const dt = _config.interphase_timestep;
        chk(gd_begin_phase(_sys, _context.wall_semiaxes));
        gd_context last;
        chk(gd_get_context(_sys, 0, &last));       // packing reaction of the last force evaluation (none yet: zero)

        auto observe = [&](long step) {
            _context.time = (double)step * dt;
            bool const logging = step % _config.interphase_logging_interval == 0, sampling = step % _config.interphase_sampling_interval == 0;
            if (logging || sampling) {
                double e = 0;
                chk(gd_compute_energy(_sys, GD_TERM_ALL, &e));
                _context.mean_energy = e / (double)_n;
            }
            if (logging) print_progress("fine", step);
            if (sampling) {
                // a snapshot every 100 steps of ~20 us: deflating it (~9 ms) on the stepping thread would be five times the stepping.
                // Each snapshot is packed on a thread of its own as soon as it is downloaded; the writer thread takes them in order
                // (gd_async_io.hpp), at most `pending` of them in flight.
                chk(gd_get_positions_f32(_sys, _buffer.data(), /*quantize=*/1));
                auto xyz = std::make_shared<std::vector<float>>(_buffer);
                std::size_t const n = _n;
                auto packed = std::make_shared<std::future<gd::h5::packed_array>>(std::async(std::launch::async, [xyz, n] {
                    auto p = gd::h5::plan_packed(n, 3, sizeof(float));
                    for (std::size_t c = 0; c < p.chunk_count(); c++) gd::h5::pack_chunk(p, c, xyz->data());
                    return p;
                }));
                gd::context const ctx = _context;
                _writer.submit([this, step, packed, ctx] {
                    _store.save_positions_packed(step, packed->get());
                    _store.save_context(step, ctx);
                });
            }
        };

        // callback(0): observation at the restart scales, then update_wall_semiaxes() on the host
        chk(gd_set_context(_sys, 0, 0, _context.bead_scale, _context.bond_scale, _context.wall_semiaxes));
        observe(0);
        double const spring[3] = {_config.wall_semiaxes_spring.x, _config.wall_semiaxes_spring.y, _config.wall_semiaxes_spring.z};
        for (int k = 0; k < 3; k++)
            _context.wall_semiaxes[k] += dt * _config.wall_mobility * (last.axial_reaction[k] - spring[k] * _context.wall_semiaxes[k]);
        chk(gd_set_context(_sys, 0, 0, _context.bead_scale, _context.bond_scale, _context.wall_semiaxes));

        gd_run_desc run{};
        run.temperature = _config.interphase_temperature; run.timestep = dt; run.spacestep = _config.interphase_spacestep;
        run.seed = _random(); run.noise_mode = GD_NOISE_PHILOX; run.flags = GD_RUN_WALL_DYNAMICS;   // scales stay at 1
        long step = 0;
        while (step < _config.interphase_steps) {
            long const next = std::min<long>(_config.interphase_steps, std::min(next_multiple(step, _config.interphase_logging_interval),
                                                                                next_multiple(step, _config.interphase_sampling_interval)));
            // callback(next) computes the energy, logs and saves on the semiaxes callback(next - 1) left, THEN moves the wall
            // (simulation_driver_interphase.cc:19-33): the last callback of the chunk stays pending over the observation
            run.steps = next - step; run.flags = GD_RUN_WALL_DYNAMICS | GD_RUN_DEFER_CALLBACK; chk(gd_run(_sys, &run));
            gd_context ctx;
            chk(gd_get_context(_sys, 0, &ctx));
            std::copy(ctx.semiaxes, ctx.semiaxes + 3, _context.wall_semiaxes);
            step = next;
            observe(step);
            chk(gd_apply_callback(_sys));
        }
    }

    gd::trajectory_store &_store;
    gd::simulation_config _config;
    gd::context _context;
    std::mt19937_64 _random;
    gd_system *_sys = nullptr;
    std::size_t _n = 0;
    std::vector<float> _buffer;
    gd::async_writer _writer{std::max<std::size_t>(1, gd::usable_threads(12))};      // snapshots in flight (packing threads + the queue); destroyed first
};

}  // namespace

int main(int argc, char **argv)
{
    // (the reference program takes the trajectory only; --steps <n> shortens the hard-wired 100 000 steps.  No environment variable is read.)
    long steps_override = 0;
    std::vector<std::string> pos;
    for (int i = 1; i < argc; i++) {
        std::string const arg = argv[i];
        if (arg == "--steps" && i + 1 < argc) steps_override = std::stol(argv[++i]);
        else pos.push_back(arg);
    }
    if (pos.size() < 1 || pos.size() > 3) {
        std::cerr << "usage: gd_fine_sampling [--steps <n>] <trajectory> [device [restart_step]]\n";
        return 1;
    }
    try {
        gd::trajectory_store store{pos[0]};
        simulation_driver driver{store, pos.size() >= 2 ? std::stoi(pos[1]) : 0, pos.size() == 3 ? std::stol(pos[2]) : 700000, steps_override};
        driver.run();
    } catch (std::exception const &e) {
        std::cerr << "error: " << e.what() << '\n';
        return 1;
    }
    return 0;
}
