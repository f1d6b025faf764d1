// gd_genome_model.hpp -- builds the 100 kb whole-genome system of the stage-5 drivers on libgdyn
// (particles: simulation_interphase/simulation_driver_particles.cc:8-36; force fields:
// simulation_interphase/simulation_driver_forcefield.cc:19-235).  The fine-sampling driver uses the same
// model without the (i,i+2) loop bonds and with unmixed chromatin bonds
// (simulation_fine_sampling/simulation_driver_forcefield.cc:8-184).
#pragma once
#include <algorithm>
#include <stdexcept>
#include <vector>

#include "../../include/gdyn.h"
#include "gd_config.hpp"
#include "gd_store.hpp"

namespace gd {

inline void chk(int rc) { if (rc != GD_OK) throw std::runtime_error(gd_last_error()); }

inline gd_system *build_genome_system(trajectory_store &store, simulation_config const &config, int device, bool loop_bonds,
                                      bool mixed_chain_bonds, std::size_t &n_out, std::uint32_t n_replicas = 1)
{
    auto const particles = store.load_particle_data();
    auto const chromosomes = store.load_chromosomes();
    auto const nucleoli = store.load_nucleolus_ranges();
    std::size_t const n = particles.size();
    n_out = n;
    gd_system *sys = nullptr;
    gd_desc desc{};
    desc.n_beads = (uint32_t)n; desc.n_replicas = n_replicas; desc.device = device; desc.box_kind = GD_BOX_OPEN;
    chk(gd_create(&desc, &sys));
    try {
        // particles: a/b factors from the metadata, mobility per range
        std::vector<double> a(n), b(n), mobility(n, 1.0);
        for (std::size_t i = 0; i < n; i++) { a[i] = particles[i].a; b[i] = particles[i].b; }
        for (auto const &c : chromosomes) for (std::size_t i = c.start; i < c.end; i++) mobility[i] = config.chromatin_mobility;
        for (auto const &r : nucleoli) for (std::size_t i = r.begin; i < r.end; i++) mobility[i] = config.nucleolus_mobility;
        chk(gd_set_bead_params(sys, a.data(), b.data(), mobility.data(), nullptr));
        // general A/B repulsion
        gd_pair_softcore pair{};
        pair.eps_a = config.a_core_repulsion; pair.sigma_a = config.a_core_diameter; pair.p_a = 2; pair.q_a = 3;
        pair.eps_b = config.b_core_repulsion; pair.sigma_b = config.b_core_diameter; pair.p_b = 8; pair.q_b = 3;
        pair.mix = 1; pair.scale_by_bead_scale = 1;
        chk(gd_set_pair_softcore(sys, &pair));
        // chromosome connectivity and mean-field (i,i+2) loops
        gd_bond_params chain{};
        chain.kind = GD_POT_SEMISPRING; chain.scale_by_bond_scale = 1;
        if (mixed_chain_bonds) {
            chain.mix = 1;
            chain.k_a = config.a_core_bond_spring; chain.k_b = config.b_core_bond_spring;
            chain.l_a = config.a_core_bond_length; chain.l_b = config.b_core_bond_length;
        } else {
            chain.k_a = config.chromatin_bond_spring; chain.l_a = config.chromatin_bond_length;
        }
        gd_bond_params loop{};
        loop.kind = GD_POT_HARMONIC; loop.mix = 1; loop.scale_by_bond_scale = 1;
        loop.k_a = config.a_core_2nd_bond_spring; loop.k_b = config.b_core_2nd_bond_spring;
        for (auto const &c : chromosomes) {
            chk(gd_add_bond_range(sys, &chain, (uint32_t)c.start, (uint32_t)c.end, 1));
            if (loop_bonds) chk(gd_add_bond_range(sys, &loop, (uint32_t)c.start, (uint32_t)c.end, 2));
        }
        // nucleolar side chains
        gd_bond_params nuc{};
        nuc.kind = GD_POT_SEMISPRING; nuc.scale_by_bond_scale = 1;
        nuc.k_a = config.nucleolus_bond_spring; nuc.l_a = config.nucleolus_bond_length;
        std::vector<uint32_t> pairs;
        for (auto const &bond : store.load_nucleolus_bonds()) { pairs.push_back((uint32_t)bond.nor_index); pairs.push_back((uint32_t)bond.nuc_index); }
        if (!pairs.empty()) chk(gd_add_bond_pairs(sys, &nuc, pairs.data(), (uint32_t)(pairs.size() / 2)));
        // nucleolar droplet attraction, only when its energy is set (simulation_driver_forcefield.cc:153-178)
        if (config.nucleolus_droplet_energy != 0) {
            std::vector<uint32_t> nucleolar;
            for (auto const &r : nucleoli) for (std::size_t i = r.begin; i < r.end; i++) nucleolar.push_back((uint32_t)i);
            chk(gd_set_pair_softwell(sys, config.nucleolus_droplet_energy, config.nucleolus_droplet_decay, config.nucleolus_droplet_cutoff,
                                     nucleolar.data(), (uint32_t)nucleolar.size()));
        }
        // nuclear membrane
        gd_wall wall{};
        wall.eps_a = config.a_core_repulsion; wall.sigma_a = config.a_core_diameter; wall.p_a = 2; wall.q_a = 3;
        wall.eps_b = config.b_core_repulsion; wall.sigma_b = config.b_core_diameter; wall.p_b = 8; wall.q_b = 3;
        wall.wall_a_factor = config.wall_a_factor; wall.wall_b_factor = config.wall_b_factor; wall.scale_by_bead_scale = 1;
        wall.packing_spring = config.wall_packing_spring; wall.mobility = config.wall_mobility;
        wall.semiaxes_spring[0] = config.wall_semiaxes_spring.x; wall.semiaxes_spring[1] = config.wall_semiaxes_spring.y;
        wall.semiaxes_spring[2] = config.wall_semiaxes_spring.z;
        wall.init_semiaxes[0] = config.wall_init_semiaxes.x; wall.init_semiaxes[1] = config.wall_init_semiaxes.y;
        wall.init_semiaxes[2] = config.wall_init_semiaxes.z;
        chk(gd_set_ellipsoid_wall(sys, &wall));
        chk(gd_set_scaling(sys, config.bead_scale_init, config.bead_scale_tau, config.bond_scale_init, config.bond_scale_tau));
    } catch (...) {
        gd_destroy(sys);
        throw;
    }
    return sys;
}

}  // namespace gd
