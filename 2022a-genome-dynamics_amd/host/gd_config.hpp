// gd_config.hpp -- the JSON configuration stored in /metadata/config of a trajectory file.
// Key names, types and "every key is mandatory" behaviour follow the reference
// (5-sim-genome/src/config_entries.inc:1-90; simulation_common/simulation_config.cc:22-36, which throws
// "<name> is not configured" for a missing key).
#pragma once
#include <cstdint>
#include <stdexcept>
#include <string>

#include <json.hpp>

namespace gd {

struct vec3 { double x = 0, y = 0, z = 0; };

struct simulation_config {
    // A/B cores
    double a_core_diameter, b_core_diameter, a_core_repulsion, b_core_repulsion;
    // chromatin
    double chromatin_bond_spring, chromatin_bond_length, chromatin_mobility;
    double a_core_bond_spring, a_core_bond_length, b_core_bond_spring, b_core_bond_length;
    double a_core_2nd_bond_spring, b_core_2nd_bond_spring;
    // nucleolus
    std::uint64_t nucleolus_sidebeads;
    double nucleolus_a_factor, nucleolus_b_factor, nucleolus_bond_spring, nucleolus_bond_length;
    double nucleolus_droplet_energy, nucleolus_droplet_decay, nucleolus_droplet_cutoff, nucleolus_mobility;
    // membrane
    vec3 wall_init_semiaxes, wall_semiaxes_spring;
    double wall_packing_spring, wall_a_factor, wall_b_factor, wall_mobility;
    // scaling
    double bead_scale_init, bead_scale_tau, bond_scale_init, bond_scale_tau;
    // initialisation stage (read for completeness; used by the spindle driver)
    std::uint64_t init_coarse_graining;
    double init_bead_diameter, init_bead_repulsion, init_bond_length, init_bond_spring, init_bend_energy, init_spindle_spring;
    vec3 init_spindle_point;
    double init_packing_radius, init_packing_spring;
    vec3 init_start_point;
    double init_start_stddev, init_mobility, init_temperature, init_timestep, init_spacestep;
    long init_spindle_steps, init_packing_steps, init_sampling_interval, init_logging_interval;
    std::string init_refinement_method;
    // relaxation / interphase
    double relaxation_temperature, relaxation_timestep, relaxation_spacestep;
    long relaxation_steps, relaxation_sampling_interval, relaxation_logging_interval;
    double interphase_temperature, interphase_timestep, interphase_spacestep;
    long interphase_steps, interphase_sampling_interval, interphase_logging_interval;
    double contactmap_distance;
    long contactmap_update_interval, contactmap_thinning_rate;
    std::uint64_t spindle_seed, interphase_seed;
};

namespace detail {
inline nlohmann::json const &need(nlohmann::json const &j, char const *name)
{
    auto it = j.find(name);
    if (it == j.end()) throw std::runtime_error(std::string(name) + " is not configured");
    return *it;
}
template <typename T> void get(nlohmann::json const &j, char const *name, T &v) { v = need(j, name).get<T>(); }
inline void get(nlohmann::json const &j, char const *name, vec3 &v)
{
    auto const &n = need(j, name);
    v = {n.at(0).get<double>(), n.at(1).get<double>(), n.at(2).get<double>()};
}
}  // namespace detail

inline simulation_config parse_simulation_config(std::string const &text)
{
    auto const j = nlohmann::json::parse(text);
    simulation_config c;
#define GD_KEY(name) detail::get(j, #name, c.name);
    GD_KEY(a_core_diameter) GD_KEY(b_core_diameter) GD_KEY(a_core_repulsion) GD_KEY(b_core_repulsion)
    GD_KEY(chromatin_bond_spring) GD_KEY(chromatin_bond_length) GD_KEY(chromatin_mobility)
    GD_KEY(a_core_bond_spring) GD_KEY(a_core_bond_length) GD_KEY(b_core_bond_spring) GD_KEY(b_core_bond_length)
    GD_KEY(a_core_2nd_bond_spring) GD_KEY(b_core_2nd_bond_spring)
    GD_KEY(nucleolus_sidebeads) GD_KEY(nucleolus_a_factor) GD_KEY(nucleolus_b_factor) GD_KEY(nucleolus_bond_spring)
    GD_KEY(nucleolus_bond_length) GD_KEY(nucleolus_droplet_energy) GD_KEY(nucleolus_droplet_decay)
    GD_KEY(nucleolus_droplet_cutoff) GD_KEY(nucleolus_mobility)
    GD_KEY(wall_init_semiaxes) GD_KEY(wall_semiaxes_spring) GD_KEY(wall_packing_spring) GD_KEY(wall_a_factor)
    GD_KEY(wall_b_factor) GD_KEY(wall_mobility)
    GD_KEY(bead_scale_init) GD_KEY(bead_scale_tau) GD_KEY(bond_scale_init) GD_KEY(bond_scale_tau)
    GD_KEY(init_coarse_graining) GD_KEY(init_bead_diameter) GD_KEY(init_bead_repulsion) GD_KEY(init_bond_length)
    GD_KEY(init_bond_spring) GD_KEY(init_bend_energy) GD_KEY(init_spindle_spring) GD_KEY(init_spindle_point)
    GD_KEY(init_packing_radius) GD_KEY(init_packing_spring) GD_KEY(init_start_point) GD_KEY(init_start_stddev)
    GD_KEY(init_mobility) GD_KEY(init_temperature) GD_KEY(init_timestep) GD_KEY(init_spacestep)
    GD_KEY(init_spindle_steps) GD_KEY(init_packing_steps) GD_KEY(init_sampling_interval) GD_KEY(init_logging_interval)
    GD_KEY(init_refinement_method)
    GD_KEY(relaxation_temperature) GD_KEY(relaxation_timestep) GD_KEY(relaxation_spacestep) GD_KEY(relaxation_steps)
    GD_KEY(relaxation_sampling_interval) GD_KEY(relaxation_logging_interval)
    GD_KEY(interphase_temperature) GD_KEY(interphase_timestep) GD_KEY(interphase_spacestep) GD_KEY(interphase_steps)
    GD_KEY(interphase_sampling_interval) GD_KEY(interphase_logging_interval)
    GD_KEY(contactmap_distance) GD_KEY(contactmap_update_interval) GD_KEY(contactmap_thinning_rate)
    GD_KEY(spindle_seed) GD_KEY(interphase_seed)
#undef GD_KEY
    return c;
}

// Two stored configurations describe the same model and schedule: every entry except the seeds agrees -- the master seed a
// file was prepared with and the two seeds derived from it (prepare/run.py:46-55); the batched driver runs R prepared files
// as R replicas of ONE handle, i.e. one force field, one cadence.
inline bool same_model_config(std::string const &text_a, std::string const &text_b)
{
    auto a = nlohmann::json::parse(text_a), b = nlohmann::json::parse(text_b);
    for (char const *seed : {"seed", "spindle_seed", "interphase_seed"}) { a.erase(seed); b.erase(seed); }
    return a == b;
}

}  // namespace gd
