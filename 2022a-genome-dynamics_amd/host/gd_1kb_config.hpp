// gd_1kb_config.hpp -- configuration of the 1 kb chromatin simulation (stage 3), parsed from JSON.
//
// Same structure, key names, defaults and required/optional split as the reference's config.hpp:11-105 and the
// member-trait tables of config.cpp:9-123 (there via jsoncons, which is not in this image; here via nlohmann/json):
//   sampling{temperature*, timestep*, steps*, clear_loops_at, loop_preloading, loop_update_interval,
//            glue_update_interval, logging_interval, sampling_interval, random_seed, output_filename}
//   chain{box_size, initial_bond_length, repulsive_diameter, repulsive_energy, attractive_diameter,
//         attractive_energy, bond_length, bond_spring, bending_energy, monomer_mobility}
//   loop{bond_spring*, forward_speed*, backward_speed*, loading_rate_density, unloading_rate,
//        convergent_detachability, roadblock_attachability, crossing_rate?, max_loops?}
//   glue{max_glues*, glue_energy*, glue_distance*, glue_binding_rate*, glue_unbinding_rate*}
//   chains[{length*, forward_boundaries, backward_boundaries, roadblocks, loaded_loops, blocks[{start*, end*, bending_energy?}]}]
// (* = required when the enclosing object is present; `sampling` itself is required).
#pragma once
#include <cstdint>
#include <optional>
#include <stdexcept>
#include <string>
#include <vector>

#include <json.hpp>

namespace gd1kb {

struct sampling_config {
    double temperature = 1, timestep = 0;
    long steps = 0, clear_loops_at = 0;
    bool loop_preloading = false;
    long loop_update_interval = 1, glue_update_interval = 1, logging_interval = 1, sampling_interval = 1;
    std::uint64_t random_seed = 0;
    std::string output_filename;
};
struct chain_type_config {
    double box_size = 1, initial_bond_length = 0, repulsive_diameter = 0, repulsive_energy = 0, attractive_diameter = 0,
           attractive_energy = 0, bond_length = 0, bond_spring = 0, bending_energy = 0, monomer_mobility = 1;
};
struct loop_type_config {
    double bond_spring = 0, forward_speed = 0, backward_speed = 0, loading_rate_density = 0, unloading_rate = 0,
           convergent_detachability = 1, roadblock_attachability = 1;
    std::optional<double> crossing_rate;
    std::optional<std::size_t> max_loops;
};
struct glue_type_config {
    std::size_t max_glues = 0;
    double glue_energy = 0, glue_distance = 0, glue_binding_rate = 0, glue_unbinding_rate = 0;
};
struct block_config { std::size_t start = 0, end = 0; std::optional<double> bending_energy; };
struct chain_config {
    std::size_t length = 0;
    std::vector<std::size_t> forward_boundaries, backward_boundaries, roadblocks, loaded_loops;
    std::vector<block_config> blocks;
};
struct simulation_config {
    sampling_config sampling;
    chain_type_config chain;
    loop_type_config loop;
    glue_type_config glue;
    std::vector<chain_config> chains;
    std::string config_text;
};

namespace detail {
using json = nlohmann::json;

template <typename T>
void required(json const &j, char const *owner, char const *key, T &out)
{
    auto it = j.find(key);
    if (it == j.end()) throw std::runtime_error(std::string(owner) + ": missing required member '" + key + "'");
    out = it->get<T>();
}
template <typename T>
void optional(json const &j, char const *key, T &out)
{
    auto it = j.find(key);
    if (it != j.end() && !it->is_null()) out = it->get<T>();
}
template <typename T>
void optional(json const &j, char const *key, std::optional<T> &out)
{
    auto it = j.find(key);
    if (it != j.end() && !it->is_null()) out = it->get<T>();
}
// "Infinity" is how JSON writers spell an infinite crossing rate (basic_loop_simulator.cpp:71-83 special-cases it)
inline void optional_rate(json const &j, char const *key, std::optional<double> &out)
{
    auto it = j.find(key);
    if (it == j.end() || it->is_null()) return;
    if (it->is_string()) {
        std::string const s = it->get<std::string>();
        if (s == "Infinity" || s == "inf" || s == "Inf") { out = HUGE_VAL; return; }
        throw std::runtime_error(std::string(key) + ": not a number");
    }
    out = it->get<double>();
}

inline chain_config parse_chain(json const &j)
{
    chain_config c;
    required(j, "chain_config", "length", c.length);
    optional(j, "forward_boundaries", c.forward_boundaries);
    optional(j, "backward_boundaries", c.backward_boundaries);
    optional(j, "roadblocks", c.roadblocks);
    optional(j, "loaded_loops", c.loaded_loops);
    auto it = j.find("blocks");
    if (it != j.end())
        for (auto const &b : *it) {
            block_config block;
            required(b, "block_config", "start", block.start);
            required(b, "block_config", "end", block.end);
            optional(b, "bending_energy", block.bending_energy);
            c.blocks.push_back(block);
        }
    return c;
}
}  // namespace detail

inline std::vector<chain_config> parse_chains_config(std::string const &text)
{
    std::vector<chain_config> chains;
    for (auto const &j : nlohmann::json::parse(text)) chains.push_back(detail::parse_chain(j));
    return chains;
}

inline simulation_config parse_simulation_config(std::string const &text)
{
    using namespace detail;
    auto const root = json::parse(text);
    simulation_config c;
    auto it = root.find("sampling");
    if (it == root.end()) throw std::runtime_error("simulation_config: missing required member 'sampling'");
    {
        auto const &j = *it;
        auto &s = c.sampling;
        required(j, "sampling_config", "temperature", s.temperature);
        required(j, "sampling_config", "timestep", s.timestep);
        required(j, "sampling_config", "steps", s.steps);
        optional(j, "clear_loops_at", s.clear_loops_at);
        optional(j, "loop_preloading", s.loop_preloading);
        optional(j, "loop_update_interval", s.loop_update_interval);
        optional(j, "glue_update_interval", s.glue_update_interval);
        optional(j, "logging_interval", s.logging_interval);
        optional(j, "sampling_interval", s.sampling_interval);
        optional(j, "random_seed", s.random_seed);
        optional(j, "output_filename", s.output_filename);
    }
    if ((it = root.find("chain")) != root.end()) {
        auto const &j = *it;
        auto &s = c.chain;
        optional(j, "box_size", s.box_size);
        optional(j, "initial_bond_length", s.initial_bond_length);
        optional(j, "repulsive_diameter", s.repulsive_diameter);
        optional(j, "repulsive_energy", s.repulsive_energy);
        optional(j, "attractive_diameter", s.attractive_diameter);
        optional(j, "attractive_energy", s.attractive_energy);
        optional(j, "bond_length", s.bond_length);
        optional(j, "bond_spring", s.bond_spring);
        optional(j, "bending_energy", s.bending_energy);
        optional(j, "monomer_mobility", s.monomer_mobility);
    }
    if ((it = root.find("loop")) != root.end()) {
        auto const &j = *it;
        auto &s = c.loop;
        required(j, "loop_type_config", "bond_spring", s.bond_spring);
        required(j, "loop_type_config", "forward_speed", s.forward_speed);
        required(j, "loop_type_config", "backward_speed", s.backward_speed);
        optional(j, "loading_rate_density", s.loading_rate_density);
        optional(j, "unloading_rate", s.unloading_rate);
        optional(j, "convergent_detachability", s.convergent_detachability);
        optional(j, "roadblock_attachability", s.roadblock_attachability);
        optional_rate(j, "crossing_rate", s.crossing_rate);
        optional(j, "max_loops", s.max_loops);
    }
    if ((it = root.find("glue")) != root.end()) {
        auto const &j = *it;
        auto &s = c.glue;
        required(j, "glue_type_config", "max_glues", s.max_glues);
        required(j, "glue_type_config", "glue_energy", s.glue_energy);
        required(j, "glue_type_config", "glue_distance", s.glue_distance);
        required(j, "glue_type_config", "glue_binding_rate", s.glue_binding_rate);
        required(j, "glue_type_config", "glue_unbinding_rate", s.glue_unbinding_rate);
    }
    if ((it = root.find("chains")) != root.end())
        for (auto const &j : *it) c.chains.push_back(parse_chain(j));
    c.config_text = text;
    return c;
}

// the effective configuration, re-encoded (stored as /config next to the verbatim /config_source)
inline std::string format_simulation_config(simulation_config const &c)
{
    using nlohmann::json;
    json j;
    auto const &s = c.sampling;
    j["sampling"] = {{"temperature", s.temperature}, {"timestep", s.timestep}, {"steps", s.steps}, {"clear_loops_at", s.clear_loops_at},
                     {"loop_preloading", s.loop_preloading}, {"loop_update_interval", s.loop_update_interval},
                     {"glue_update_interval", s.glue_update_interval}, {"logging_interval", s.logging_interval},
                     {"sampling_interval", s.sampling_interval}, {"random_seed", s.random_seed}, {"output_filename", s.output_filename}};
    auto const &ch = c.chain;
    j["chain"] = {{"box_size", ch.box_size}, {"initial_bond_length", ch.initial_bond_length}, {"repulsive_diameter", ch.repulsive_diameter},
                  {"repulsive_energy", ch.repulsive_energy}, {"attractive_diameter", ch.attractive_diameter},
                  {"attractive_energy", ch.attractive_energy}, {"bond_length", ch.bond_length}, {"bond_spring", ch.bond_spring},
                  {"bending_energy", ch.bending_energy}, {"monomer_mobility", ch.monomer_mobility}};
    auto const &l = c.loop;
    j["loop"] = {{"bond_spring", l.bond_spring}, {"forward_speed", l.forward_speed}, {"backward_speed", l.backward_speed},
                 {"loading_rate_density", l.loading_rate_density}, {"unloading_rate", l.unloading_rate},
                 {"convergent_detachability", l.convergent_detachability}, {"roadblock_attachability", l.roadblock_attachability}};
    if (l.crossing_rate) { if (std::isinf(*l.crossing_rate)) j["loop"]["crossing_rate"] = "Infinity"; else j["loop"]["crossing_rate"] = *l.crossing_rate; }
    if (l.max_loops) j["loop"]["max_loops"] = *l.max_loops;
    auto const &g = c.glue;
    j["glue"] = {{"max_glues", g.max_glues}, {"glue_energy", g.glue_energy}, {"glue_distance", g.glue_distance},
                 {"glue_binding_rate", g.glue_binding_rate}, {"glue_unbinding_rate", g.glue_unbinding_rate}};
    j["chains"] = json::array();
    for (auto const &chain : c.chains) {
        json jc = {{"length", chain.length}, {"forward_boundaries", chain.forward_boundaries}, {"backward_boundaries", chain.backward_boundaries},
                   {"roadblocks", chain.roadblocks}, {"loaded_loops", chain.loaded_loops}, {"blocks", json::array()}};
        for (auto const &b : chain.blocks) {
            json jb = {{"start", b.start}, {"end", b.end}};
            if (b.bending_energy) jb["bending_energy"] = *b.bending_energy;
            jc["blocks"].push_back(jb);
        }
        j["chains"].push_back(jc);
    }
    return j.dump();
}

}  // namespace gd1kb
