// gd_store.hpp -- trajectory store of the 5-sim-genome drivers on the plain HDF5 C API.
//
// Same on-disk layout, dataset names, types and filters as the reference's simulation_store
// (5-sim-genome/src/simulation_common/simulation_store.cc:44-313,318-407; layout documented in
// 5-sim-genome/README.md:12-49), so files stay readable by the reference's Python readers
// (5-sim-genome/src/script_common/store.py):
//   /metadata/{config (JSON string), ab_factors (N,2), chromosome_ranges (C,2)+attr keys, centromere_ranges,
//              nucleolus_ranges, nucleolus_bonds}
//   /snapshots/<phase>/.steps            numerically ordered variable-length strings
//   /snapshots/<phase>/<step>/positions  float32 (N,3) rounded to 2^-16, <=1 MiB chunks, shuffle + deflate 6
//   /snapshots/<phase>/<step>/context    JSON {time, bead_scale, bond_scale, wall_semiaxes, mean_energy, wall_energy}
//   /snapshots/<phase>/<step>/contact_map  uint32 (M,3) rows (i,j,count)
//   /snapshots/<phase>/metadata/chromosome_ranges (+ attr keys)
// HighFive is not available in this image; nothing here is taken from it.
#pragma once
#include <hdf5.h>

#include "gd_h5util.hpp"

#include <array>
#include <cstdint>
#include <map>
#include <set>
#include <stdexcept>
#include <string>
#include <vector>

namespace gd {

struct chromosome_range { std::string name; std::size_t start = 0, end = 0, centromere_start = 0, centromere_end = 0; };
struct index_range { std::size_t begin = 0, end = 0; };
struct nucleolus_bond { std::size_t nor_index = 0, nuc_index = 0; };
struct ab_factor { double a = 0, b = 0; };

struct context {   // simulation_common/simulation_context.hpp
    double time = 0;
    double wall_semiaxes[3] = {0, 0, 0};
    double bead_scale = 1, bond_scale = 1;
    double mean_energy = 0, wall_energy = 0;
};


class trajectory_store {
public:
    // opens read-write (the reference's constructor); create=true makes a new file (input generators)
    explicit trajectory_store(std::string const &filename, bool create = false);
    ~trajectory_store();
    trajectory_store(trajectory_store const &) = delete;
    trajectory_store &operator=(trajectory_store const &) = delete;

    // metadata
    std::string load_config_text();
    std::vector<chromosome_range> load_chromosomes();
    std::vector<ab_factor> load_particle_data();
    std::vector<index_range> load_nucleolus_ranges();
    std::vector<nucleolus_bond> load_nucleolus_bonds();
    // types: /metadata/particle_types (i8 enum, prepare/run.py:81-88) when given; nucleolus_names: the keys of nucleolus_ranges
    void save_metadata(std::string const &config_json, std::vector<ab_factor> const &ab, std::vector<chromosome_range> const &chroms,
                       std::vector<index_range> const &nucleolus_ranges, std::vector<nucleolus_bond> const &bonds,
                       std::vector<std::int8_t> const *types = nullptr, std::vector<std::string> const *nucleolus_names = nullptr);
    void create_phase_groups();         // /snapshots/{spindle,packing,relaxation,interphase} (prepare/run.py:60-68)
    std::vector<std::int8_t> load_particle_types(std::vector<std::pair<std::string, int>> *enum_members = nullptr);
    std::string load_keys(std::string const &dataset);      // the JSON text of /metadata/<dataset>'s "keys" attribute
    // /snapshots/<phase>/<step>/positions as plain float64 (N,3), replacing the snapshot group (refine/run.py:41-46)
    void replace_positions_f64(long step, double const *xyz, std::size_t n);

    // snapshots
    void set_phase(std::string const &phase) { _phase = phase; }
    void save_chromosomes(std::vector<chromosome_range> const &chroms);
    void save_positions(long step, float const *xyz_quantized, std::size_t n);      // already rounded to 2^-16 (device)
    void save_positions(long step, double const *xyz, std::size_t n);               // quantises like simulation_store.cc:257-268
    void save_context(long step, context const &c);
    void save_contacts(long step, std::vector<std::array<std::uint32_t, 3>> const &contacts);
    // the same datasets from chunks packed beforehand, possibly on other threads (gd_h5util.hpp: plan_packed / pack_chunk)
    void save_positions_packed(long step, h5::packed_array const &p);
    void save_contacts_packed(long step, h5::packed_array const &p);
    std::vector<std::array<double, 3>> load_positions(long step);
    context load_context(long step);
    std::vector<long> load_steps();
    void flush();

private:
    hid_t require_group(hid_t parent, std::string const &name);
    hid_t snapshot_group(long step);
    void update_ordered_steps(hid_t phase_group, long step);
    hid_t _file = -1;
    std::string _phase = "unknown";
};

float quantize16(double v);   // nearbyint(float(v) * 2^16) / 2^16, simulation_store.cc:403-407

}  // namespace gd
