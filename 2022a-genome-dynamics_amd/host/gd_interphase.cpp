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
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <ctime>
#include <iomanip>
#include <iostream>
#include <map>
#include <memory>
#include <mutex>
#include <random>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "../../include/gdyn.h"
#include "gd_async_io.hpp"
#include "gd_config.hpp"
#include "gd_genome_model.hpp"
#include "gd_store.hpp"

namespace {

using gd::chk;

// --timing: where the wall time of a run goes (stepping thread and writer thread), one line on stderr at the end
class timing_table {
public:
    struct scope {
        timing_table &t; char const *name; std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
        ~scope() { t.add(name, std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count()); }
    };
    void add(char const *name, double seconds) { std::lock_guard<std::mutex> lk(_m); _t[name] += seconds; }
    void note(std::string const &line) { std::lock_guard<std::mutex> lk(_m); _notes.push_back(line); }
    void print() const
    {
        for (auto const &n : _notes) std::clog << "[timing] " << n << '\n';
        std::clog << "[timing]";
        for (auto const &kv : _t) std::clog << ' ' << kv.first << ' ' << std::fixed << std::setprecision(3) << kv.second;
        std::clog << '\n';
    }
private:
    std::mutex _m;
    std::map<std::string, double> _t;
    std::vector<std::string> _notes;
};
timing_table g_timing;
#define GD_CONCAT2(a, b) a##b
#define GD_CONCAT(a, b) GD_CONCAT2(a, b)      // (two levels: __LINE__ expands before the paste)
#define TIMED(name) timing_table::scope GD_CONCAT(timed_scope_, __LINE__){g_timing, name}

// Time-integrated contact maps (simulation_interphase/contact_map.cc:26-91) live on the device, one per replica (gd_contacts_*):
// an update is one pair search over all replicas plus one insert launch, and only a dump moves rows to the host.
std::vector<std::array<std::uint32_t, 3>> fetch_contacts(gd_system *sys, uint32_t replica)
{
    uint64_t n = 0;
    chk(gd_contacts_fetch(sys, replica, nullptr, 0, &n));
    std::vector<std::array<std::uint32_t, 3>> rows(n);
    if (n) chk(gd_contacts_fetch(sys, replica, rows.data()->data(), n, &n));
    return rows;
}

// One driver = one libgdyn handle = R replicas = R trajectory files.  R = 1 is the reference program; R > 1 batches R runs
// of the reference's ensemble (one process per seed, each with its own prepared file: 5-sim-genome/scripts/run_simulation:8-25,
// read back as output-*.h5 by contact_map/contact_map.py:14-39) into one launch: replica r takes its initial structure, its
// seeds and its outputs from file r; every replica draws the noise stream its own one-replica run would draw
// (gd_run_desc.replica_seeds), so a batched trajectory equals the solo one up to fp32 summation order.
class simulation_driver {
public:
    simulation_driver(std::vector<std::unique_ptr<gd::trajectory_store>> &stores, int device, bool auto_skin = false)
        : _stores(stores), _R(stores.size()), _config(gd::parse_simulation_config(stores[0]->load_config_text())), _auto_skin(auto_skin)
    {
        // compatibility defaults of older runs (simulation_driver.cc:20-29)
        auto set_default = [](double &var, double def) { if (var == 0) var = def; };
        set_default(_config.a_core_bond_spring, _config.chromatin_bond_spring);
        set_default(_config.a_core_bond_length, _config.chromatin_bond_length);
        set_default(_config.b_core_bond_spring, _config.chromatin_bond_spring);
        set_default(_config.b_core_bond_length, _config.chromatin_bond_length);
        for (std::size_t r = 0; r < _R; r++) {
            auto const cfg = gd::parse_simulation_config(_stores[r]->load_config_text());
            // one handle runs one force field and one schedule: every config entry except the seeds must agree
            if (r > 0 && !gd::same_model_config(_stores[r]->load_config_text(), _stores[0]->load_config_text()))
                throw std::runtime_error("batched trajectories must share one simulation config (seeds aside)");
            _random.emplace_back(cfg.interphase_seed);          // 1st draw: relaxation seed, 2nd: interphase seed
        }
        TIMED("setup");
        setup(device);
    }
    ~simulation_driver() { gd_destroy(_sys); }

    void run()
    {
        g_timing.note("packing pool of " + std::to_string(gd::usable_threads()) + " threads on " + std::to_string(gd::usable_cpus()) + " usable CPUs");
        run_relaxation(); report("relaxation"); run_simulation(); report("interphase");
        TIMED("writer_wait");
        _writer.drain();
    }
    // bead-steps of the whole run (all files of this process): what a farm's per-device rate is made of
    double bead_steps() const { return (double)_n * (double)_R * (double)(_config.relaxation_steps + _config.interphase_steps); }
    void report(char const *phase)      // (--timing) the list statistics of the handle at the end of a phase
    {
        g_timing.note(std::string(phase) + ": " + std::to_string(_energy_calls) + " energy evaluations so far, " + std::to_string(_energy_builds) +
                      " list builds inside them, slowest " + std::to_string(_energy_max) + " s");
        gd_context c;
        chk(gd_get_context(_sys, 0, &c));
        char line[256];
        std::snprintf(line, sizeof line, "%s: list path %u, %.1f entries per bead, radius %.4f, interval %u, %llu builds, %llu rollbacks, lists %.2f GB",
                      phase, c.list_path, (double)c.list_entries / (double)_n, c.list_radius, c.rebuild_interval,
                      (unsigned long long)c.rebuilds, (unsigned long long)c.rollbacks, (double)c.list_bytes / 1e9);
        g_timing.note(line);
    }

private:
    void setup(int device)
    {
        _sys = gd::build_genome_system(*_stores[0], _config, device, /*loop_bonds=*/true, /*mixed_chain_bonds=*/true, _n, (uint32_t)_R);
        for (std::size_t r = 1; r < _R; r++) {
            auto const p0 = _stores[0]->load_particle_data(), pr = _stores[r]->load_particle_data();
            if (pr.size() != p0.size()) throw std::runtime_error("batched trajectories must hold the same model (bead count differs)");
            for (std::size_t i = 0; i < p0.size(); i++)
                if (pr[i].a != p0[i].a || pr[i].b != p0[i].b) throw std::runtime_error("batched trajectories must hold the same model (A/B factors differ)");
            // the topology the handle is built from is file 0's: chains, nucleolar ranges and bonds must agree as well
            auto const c0 = _stores[0]->load_chromosomes(), cr = _stores[r]->load_chromosomes();
            bool same = c0.size() == cr.size();
            for (std::size_t i = 0; same && i < c0.size(); i++) same = c0[i].start == cr[i].start && c0[i].end == cr[i].end;
            auto const n0 = _stores[0]->load_nucleolus_ranges(), nr = _stores[r]->load_nucleolus_ranges();
            same = same && n0.size() == nr.size();
            for (std::size_t i = 0; same && i < n0.size(); i++) same = n0[i].begin == nr[i].begin && n0[i].end == nr[i].end;
            auto const b0 = _stores[0]->load_nucleolus_bonds(), br = _stores[r]->load_nucleolus_bonds();
            same = same && b0.size() == br.size();
            for (std::size_t i = 0; same && i < b0.size(); i++) same = b0[i].nor_index == br[i].nor_index && b0[i].nuc_index == br[i].nuc_index;
            if (!same) throw std::runtime_error("batched trajectories must hold the same model (chromosome / nucleolus tables differ)");
        }
        // setup_context (simulation_driver.cc:43-51)
        gd::context c{};
        c.wall_semiaxes[0] = _config.wall_init_semiaxes.x; c.wall_semiaxes[1] = _config.wall_init_semiaxes.y;
        c.wall_semiaxes[2] = _config.wall_init_semiaxes.z;
        c.bead_scale = _config.bead_scale_init; c.bond_scale = _config.bond_scale_init;
        _context.assign(_R, c);
        {   // the list width follows the structure (a freshly refined genome is a dense globule that decondenses over the run): by
            // the library's rules on the state -- tile class, rows sized per wave -- so that a seed gives one trajectory, as in
            // the reference (scripts/run_simulation:8-25); --auto-skin selects it from measured chunk times instead
            gd_tuning tune{};
            tune.adapt_interval = 1; tune.auto_skin = _auto_skin ? 1 : 0;
            chk(gd_set_tuning(_sys, &tune));
        }
        _buffer.resize(3 * _n * _R);
        _energy.resize(_R);
    }

    std::vector<double> semiaxes() const
    {
        std::vector<double> v(3 * _R);
        for (std::size_t r = 0; r < _R; r++) std::copy(_context[r].wall_semiaxes, _context[r].wall_semiaxes + 3, v.begin() + 3 * r);
        return v;
    }

    void print_progress(char const *phase, long step)
    {
        std::time_t const now = std::time(nullptr);
        for (std::size_t r = 0; r < _R; r++) {
            auto const &c = _context[r];
            double const radius = std::cbrt(c.wall_semiaxes[0] * c.wall_semiaxes[1] * c.wall_semiaxes[2]);
            std::clog << "[" << phase;
            if (_R > 1) std::clog << ":" << r;
            std::clog << "] " << std::put_time(std::localtime(&now), "%F %T") << '\t' << step << '\t'
                      << "t: " << c.time << '\t' << "R: " << radius << '\t' << "E: " << c.mean_energy << '\n';
        }
    }

    void mean_energy()
    {
        TIMED("energy");
        gd_context before, after;
        chk(gd_get_context(_sys, 0, &before));
        auto const t0 = std::chrono::steady_clock::now();
        chk(gd_compute_energy(_sys, GD_TERM_ALL, _energy.data()));
        double const dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        chk(gd_get_context(_sys, 0, &after));
        _energy_calls++; _energy_builds += after.rebuilds - before.rebuilds; _energy_max = std::max(_energy_max, dt);
        for (std::size_t r = 0; r < _R; r++) _context[r].mean_energy = _energy[r] / (double)_n;
    }

    // Output leaves the stepping thread as a job (gd_async_io.hpp): the chunks of all R files are packed on the pool's threads, the
    // HDF5 calls follow on the writer thread, the device goes on stepping meanwhile.
    void save_snapshot(long step)
    {
        std::shared_ptr<std::vector<float>> xyz;
        {
            TIMED("snapshot_download");
            chk(gd_get_positions_f32(_sys, _buffer.data(), /*quantize=*/1));      // 16 fractional bits, rounded on the device
            xyz = std::make_shared<std::vector<float>>(_buffer);
        }
        auto ctx = std::make_shared<std::vector<gd::context>>(_context);
        TIMED("writer_wait");
        _writer.submit([this, step, xyz, ctx] {
            std::vector<gd::h5::packed_array> packed(_R, gd::h5::plan_packed(_n, 3, sizeof(float)));
            std::size_t const per = packed[0].chunk_count();
            {
                TIMED("w:pack");
                _pool.parallel_for(_R * per, [&](std::size_t t) { gd::h5::pack_chunk(packed[t / per], t % per, xyz->data() + 3 * _n * (t / per)); });
            }
            TIMED("w:hdf5");
            for (std::size_t r = 0; r < _R; r++) {
                _stores[r]->save_positions_packed(step, packed[r]);
                _stores[r]->save_context(step, (*ctx)[r]);
            }
        });
    }

    void save_contacts(long step)
    {
        auto rows = std::make_shared<std::vector<std::vector<std::array<std::uint32_t, 3>>>>(_R);
        {
            TIMED("contacts_fetch");
            for (std::size_t r = 0; r < _R; r++) (*rows)[r] = fetch_contacts(_sys, (uint32_t)r);
            chk(gd_contacts_clear(_sys, GD_ALL_REPLICAS));
        }
        TIMED("writer_wait");
        _writer.submit([this, step, rows] {
            std::vector<gd::h5::packed_array> packed(_R);
            std::vector<std::pair<std::size_t, std::size_t>> tasks;
            for (std::size_t r = 0; r < _R; r++) {
                packed[r] = gd::h5::plan_packed((*rows)[r].size(), 3, sizeof(std::uint32_t));
                for (std::size_t c = 0; c < packed[r].chunk_count(); c++) tasks.push_back({r, c});
            }
            {
                TIMED("w:pack");
                _pool.parallel_for(tasks.size(), [&](std::size_t t) { gd::h5::pack_chunk(packed[tasks[t].first], tasks[t].second, (*rows)[tasks[t].first].data()); });
            }
            TIMED("w:hdf5");
            for (std::size_t r = 0; r < _R; r++) _stores[r]->save_contacts_packed(step, packed[r]);
        });
    }

    // advance to `target` and leave the state updates of callback(target) pending (GD_RUN_DEFER_CALLBACK): what the host
    // part of the reference's callback(target) sees -- mean_energy, the log line, the saved context and the contact search all
    // run BEFORE update_bead_scale() / update_wall_semiaxes() (simulation_driver_interphase.cc:20-43), i.e. on the scales, the
    // semiaxes and the contact distance that callback(target - 1) left
    void advance(gd_run_desc &run, long &step, long target)
    {
        run.steps = target - step; run.flags |= GD_RUN_DEFER_CALLBACK;
        { TIMED("gd_run"); chk(gd_run(_sys, &run)); }
        for (std::size_t r = 0; r < _R; r++) {
            gd_context ctx;
            chk(gd_get_context(_sys, (uint32_t)r, &ctx));
            _context[r].bead_scale = ctx.bead_scale; _context[r].bond_scale = ctx.bond_scale;
            std::copy(ctx.semiaxes, ctx.semiaxes + 3, _context[r].wall_semiaxes);
        }
        _contact_distance = _config.contactmap_distance * _context[0].bead_scale;     // set by update_bead_scale() of callback(target - 1), :66
                                                                                      // (replicas of one handle are at the same step: one scale)
        step = target;
    }

    void run_relaxation()
    {
        std::vector<double> xyz(3 * _n * _R);
        for (std::size_t r = 0; r < _R; r++) {
            _stores[r]->set_phase("relaxation");
            auto const init = _stores[r]->load_positions(0);
            if (init.size() != _n) throw std::runtime_error("relaxation/0/positions has the wrong number of beads");
            for (std::size_t i = 0; i < _n; i++) for (int k = 0; k < 3; k++) xyz[3 * (_n * r + i) + k] = init[i][k];
        }
        chk(gd_set_positions(_sys, xyz.data()));
        chk(gd_begin_phase(_sys, semiaxes().data()));
        auto callback = [&](long step) {
            bool const logging = step % _config.relaxation_logging_interval == 0, sampling = step % _config.relaxation_sampling_interval == 0;
            if (logging || sampling) mean_energy();
            if (logging) print_progress("relax", step);
            if (sampling) save_snapshot(step);
        };
        callback(0);
        std::vector<uint64_t> seeds(_R);
        for (std::size_t r = 0; r < _R; r++) seeds[r] = _random[r]();
        gd_run_desc run{};
        run.temperature = _config.relaxation_temperature; run.timestep = _config.relaxation_timestep;
        run.spacestep = _config.relaxation_spacestep; run.seed = seeds[0]; run.noise_mode = GD_NOISE_PHILOX; run.flags = 0;
        run.replica_seeds = _R > 1 ? seeds.data() : nullptr;
        long step = 0;
        while (step < _config.relaxation_steps) {
            long const next = std::min<long>(_config.relaxation_steps, std::min(next_multiple(step, _config.relaxation_logging_interval),
                                                                                next_multiple(step, _config.relaxation_sampling_interval)));
            run.steps = next - step; { TIMED("gd_run"); chk(gd_run(_sys, &run)); } step = next;
            callback(step);
        }
    }

    static long next_multiple(long step, long interval) { return (step / interval + 1) * interval; }

    void run_simulation()
    {
        { TIMED("writer_wait"); _writer.drain(); }          // (the relaxation's last snapshot goes to the relaxation phase)
        for (auto &st : _stores) st->set_phase("interphase");
        double const dt = _config.interphase_timestep;
        chk(gd_begin_phase(_sys, semiaxes().data()));       // step = 0, time = 0
        std::vector<std::array<double, 3>> reaction(_R);
        for (std::size_t r = 0; r < _R; r++) {
            gd_context last;
            chk(gd_get_context(_sys, (uint32_t)r, &last));
            reaction[r] = {last.axial_reaction[0], last.axial_reaction[1], last.axial_reaction[2]};
        }

        // host part of callback(step): everything except the state updates that run on the device
        auto observe = [&](long step) {
            for (auto &c : _context) c.time = (double)step * dt;
            bool const logging = step % _config.interphase_logging_interval == 0, sampling = step % _config.interphase_sampling_interval == 0;
            long const frame = step / _config.interphase_sampling_interval;
            if (logging || sampling) mean_energy();
            if (logging) print_progress("inter", step);
            if (sampling) save_snapshot(step);
            if (step % _config.contactmap_update_interval == 0 && _contact_distance > 0) {   // (the reference's distance is 0 until callback(0) has set it)
                TIMED("contacts_update");
                chk(gd_contacts_update(_sys, _contact_distance));
            }
            if (sampling && frame % _config.contactmap_thinning_rate == 0) save_contacts(step);
        };

        // callback(0): observation, then update_bead_scale() and update_wall_semiaxes() on the host
        // (simulation_driver_interphase.cc:42-43,59-80); the packing reaction is that of the last force evaluation
        observe(0);
        double const spring[3] = {_config.wall_semiaxes_spring.x, _config.wall_semiaxes_spring.y, _config.wall_semiaxes_spring.z};
        for (std::size_t r = 0; r < _R; r++) {
            auto &c = _context[r];
            c.bead_scale = 1 - (1 - _config.bead_scale_init) * std::exp(-0.0 / _config.bead_scale_tau);
            c.bond_scale = 1 - (1 - _config.bond_scale_init) * std::exp(-0.0 / _config.bond_scale_tau);
            _contact_distance = _config.contactmap_distance * c.bead_scale;
            for (int k = 0; k < 3; k++)
                c.wall_semiaxes[k] += dt * _config.wall_mobility * (reaction[r][k] - spring[k] * c.wall_semiaxes[k]);
            chk(gd_set_context(_sys, (uint32_t)r, 0, c.bead_scale, c.bond_scale, c.wall_semiaxes));
        }

        std::vector<uint64_t> seeds(_R);
        for (std::size_t r = 0; r < _R; r++) seeds[r] = _random[r]();
        gd_run_desc run{};
        run.temperature = _config.interphase_temperature; run.timestep = dt; run.spacestep = _config.interphase_spacestep;
        run.seed = seeds[0]; run.noise_mode = GD_NOISE_PHILOX; run.flags = GD_RUN_UPDATE_SCALES | GD_RUN_WALL_DYNAMICS;
        run.replica_seeds = _R > 1 ? seeds.data() : nullptr;
        long step = 0;
        while (step < _config.interphase_steps) {
            long const next = std::min<long>(_config.interphase_steps,
                                             std::min({next_multiple(step, _config.interphase_logging_interval),
                                                       next_multiple(step, _config.interphase_sampling_interval),
                                                       next_multiple(step, _config.contactmap_update_interval)}));
            advance(run, step, next);
            observe(step);
            chk(gd_apply_callback(_sys));      // update_bead_scale() + update_wall_semiaxes() of callback(step), on the device
        }
    }

    std::vector<std::unique_ptr<gd::trajectory_store>> &_stores;
    std::size_t _R;
    gd::simulation_config _config;
    bool _auto_skin = false;
    std::vector<gd::context> _context;
    double _contact_distance = 0;
    std::vector<std::mt19937_64> _random;
    gd_system *_sys = nullptr;
    std::size_t _n = 0;
    std::vector<float> _buffer;
    gd::thread_pool _pool{gd::usable_threads()};
    gd::async_writer _writer;      // (after the pool and the stores it uses: destroyed first)
    std::vector<double> _energy;
    unsigned long _energy_calls = 0, _energy_builds = 0;
    double _energy_max = 0;
};

}  // namespace

int main(int argc, char **argv)
{
    // gd_interphase <trajectory> [device]                      the reference's command line
    // gd_interphase [--device d] <trajectory> <trajectory>...  R prepared files as R replicas of one handle
    // options: --timing (wall-time split on stderr at the end), --auto-skin (list width selected from measured chunk times: the
    // trajectory of a seed then depends on timing; off by default.  --fixed-skin, the former spelling of the default, is accepted)
    std::vector<std::string> files;
    int device = 0;
    bool timing = false, auto_skin = false;
    auto const t_start = std::chrono::steady_clock::now();
    for (int i = 1; i < argc; i++) {
        std::string const arg = argv[i];
        if (arg == "--timing") timing = true;
        else if (arg == "--auto-skin") auto_skin = true;
        else if (arg == "--fixed-skin") auto_skin = false;
        else if (arg == "--device" && i + 1 < argc) device = std::stoi(argv[++i]);
        else files.push_back(arg);
    }
    if (files.size() == 2 && !files[1].empty() && files[1].find_first_not_of("0123456789") == std::string::npos) {
        device = std::stoi(files[1]); files.pop_back();
    }
    if (files.empty()) {
        std::cerr << "usage: gd_interphase <trajectory> [device]\n       gd_interphase [--device d] <trajectory> <trajectory>...\n";
        return 1;
    }
    try {
        std::vector<std::unique_ptr<gd::trajectory_store>> stores;
        {
            TIMED("open_files");
            for (auto const &f : files) stores.push_back(std::make_unique<gd::trajectory_store>(f));
        }
        double bead_steps = 0;
        {
            simulation_driver driver{stores, device, auto_skin};
            driver.run();
            bead_steps = driver.bead_steps();
        }
        { TIMED("close_files"); stores.clear(); }
        double const total_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count();
        g_timing.add("total", total_s);
        // a batched run (the farm's unit: one process per GPU) reports its device's rate, files opened to files closed -- the first
        // 8-GPU run of gd_farm yields the scaling table from these lines alone (a solo run keeps the reference's output)
        if (files.size() > 1 || timing)
            std::clog << "[rate] device " << device << ": " << files.size() << " file(s), " << bead_steps << " bead-steps in " << total_s << " s = "
                      << bead_steps / total_s << " bead-steps/s\n";
        if (timing) g_timing.print();
    } catch (std::exception const &e) {
        std::cerr << "error: " << e.what() << '\n';
        return 1;
    }
    return 0;
}
