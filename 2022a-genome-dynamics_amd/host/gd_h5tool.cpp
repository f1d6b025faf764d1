// gd_h5tool -- small command-line companion of the trajectory store (used by the tests and for demos; the
// reference creates its input files with Python/h5py, 5-sim-genome/src/prepare, which is not available here).
//   gd_h5tool make-input <out.h5> <config.json> <chroms.tsv> <ab.f64> <positions.f64> [<nucleolus_bonds.u32> [<nucleolus_ranges.u32>]]
//        chroms.tsv rows: name start end centromere_start centromere_end; raw little-endian arrays (N,2)/(N,3)
//   gd_h5tool steps <file> <phase>                     numerically ordered step list
//   gd_h5tool positions <file> <phase> <step> <out.f64>
//   gd_h5tool context <file> <phase> <step>            prints the JSON context fields
//   gd_h5tool contacts <file> <phase> <step>           prints "i j count" rows
//   gd_h5tool dataset <file> <path> <out.f64>          any numeric dataset as raw doubles; prints its shape
//   gd_h5tool strings <file> <path>                    a string dataset (scalar or 1-d), one item per line
#include <cstdio>
#include <fstream>
#include <iostream>
#include <sstream>
#include <string>
#include <vector>

#include <hdf5.h>

#include "gd_store.hpp"

static std::vector<char> slurp(std::string const &path)
{
    std::ifstream in(path, std::ios::binary);
    if (!in) throw std::runtime_error("cannot read " + path);
    return std::vector<char>((std::istreambuf_iterator<char>(in)), std::istreambuf_iterator<char>());
}

int main(int argc, char **argv)
{
    try {
        std::string const cmd = argc > 1 ? argv[1] : "";
        if (cmd == "make-input" && argc >= 7 && argc <= 9) {
            auto const cfg = slurp(argv[3]);
            std::vector<gd::chromosome_range> chroms;
            std::ifstream tsv(argv[4]);
            for (std::string line; std::getline(tsv, line);) {
                std::istringstream ss(line);
                gd::chromosome_range c;
                if (ss >> c.name >> c.start >> c.end >> c.centromere_start >> c.centromere_end) chroms.push_back(c);
            }
            auto const abraw = slurp(argv[5]), posraw = slurp(argv[6]);
            std::size_t const n = abraw.size() / (2 * sizeof(double));
            if (posraw.size() != n * 3 * sizeof(double)) throw std::runtime_error("ab and position arrays disagree on N");
            auto const *ab = reinterpret_cast<double const *>(abraw.data());
            std::vector<gd::ab_factor> abv(n);
            for (std::size_t i = 0; i < n; i++) abv[i] = {ab[2 * i], ab[2 * i + 1]};
            std::vector<gd::nucleolus_bond> bonds;
            std::vector<gd::index_range> nranges;
            if (argc >= 8) {
                auto const raw = slurp(argv[7]);
                auto const *p = reinterpret_cast<std::uint32_t const *>(raw.data());
                for (std::size_t k = 0; k + 1 < raw.size() / sizeof(std::uint32_t); k += 2) bonds.push_back({p[k], p[k + 1]});
            }
            if (argc == 9) {
                auto const raw = slurp(argv[8]);
                auto const *p = reinterpret_cast<std::uint32_t const *>(raw.data());
                for (std::size_t k = 0; k + 1 < raw.size() / sizeof(std::uint32_t); k += 2) nranges.push_back({p[k], p[k + 1]});
            }
            gd::trajectory_store store(argv[2], /*create=*/true);
            store.save_metadata(std::string(cfg.begin(), cfg.end()), abv, chroms, nranges, bonds);
            store.set_phase("relaxation");
            store.save_positions(0, reinterpret_cast<double const *>(posraw.data()), n);
            return 0;
        }
        if (cmd == "steps" && argc == 4) {
            gd::trajectory_store store(argv[2]);
            store.set_phase(argv[3]);
            for (long s : store.load_steps()) std::cout << s << '\n';
            return 0;
        }
        if (cmd == "positions" && argc == 6) {
            gd::trajectory_store store(argv[2]);
            store.set_phase(argv[3]);
            auto const pos = store.load_positions(std::stol(argv[4]));
            std::ofstream out(argv[5], std::ios::binary);
            out.write(reinterpret_cast<char const *>(pos.data()), (std::streamsize)(pos.size() * 3 * sizeof(double)));
            return 0;
        }
        if (cmd == "context" && argc == 5) {
            gd::trajectory_store store(argv[2]);
            store.set_phase(argv[3]);
            auto const c = store.load_context(std::stol(argv[4]));
            std::printf("{\"time\": %.17g, \"bead_scale\": %.17g, \"bond_scale\": %.17g, \"wall_semiaxes\": [%.17g, %.17g, %.17g], \"mean_energy\": %.17g, \"wall_energy\": %.17g}\n",
                        c.time, c.bead_scale, c.bond_scale, c.wall_semiaxes[0], c.wall_semiaxes[1], c.wall_semiaxes[2], c.mean_energy, c.wall_energy);
            return 0;
        }
        if (cmd == "contacts" && argc == 5) {
            hid_t f = H5Fopen(argv[2], H5F_ACC_RDONLY, H5P_DEFAULT);
            std::string const path = std::string("/snapshots/") + argv[3] + "/" + argv[4] + "/contact_map";
            hid_t ds = H5Dopen2(f, path.c_str(), H5P_DEFAULT);
            if (ds < 0) return 0;
            hid_t sp = H5Dget_space(ds);
            hsize_t dims[2];
            H5Sget_simple_extent_dims(sp, dims, nullptr);
            std::vector<std::uint32_t> v(dims[0] * 3);
            H5Dread(ds, H5T_NATIVE_UINT32, H5S_ALL, H5S_ALL, H5P_DEFAULT, v.data());
            for (hsize_t k = 0; k < dims[0]; k++) std::cout << v[3 * k] << ' ' << v[3 * k + 1] << ' ' << v[3 * k + 2] << '\n';
            return 0;
        }
        if (cmd == "dataset" && argc == 5) {
            hid_t f = H5Fopen(argv[2], H5F_ACC_RDONLY, H5P_DEFAULT);
            hid_t ds = H5Dopen2(f, argv[3], H5P_DEFAULT);
            if (f < 0 || ds < 0) throw std::runtime_error(std::string("no dataset ") + argv[3]);
            hid_t sp = H5Dget_space(ds);
            hsize_t dims[8] = {0};
            int const nd = H5Sget_simple_extent_dims(sp, dims, nullptr);
            hsize_t count = 1;
            for (int k = 0; k < nd; k++) { count *= dims[k]; std::cout << dims[k] << (k + 1 < nd ? " " : "\n"); }
            std::vector<double> v(count);
            if (count && H5Dread(ds, H5T_NATIVE_DOUBLE, H5S_ALL, H5S_ALL, H5P_DEFAULT, v.data()) < 0) throw std::runtime_error("cannot read dataset");
            std::ofstream out(argv[4], std::ios::binary);
            out.write(reinterpret_cast<char const *>(v.data()), (std::streamsize)(count * sizeof(double)));
            return 0;
        }
        if (cmd == "strings" && argc == 4) {
            hid_t f = H5Fopen(argv[2], H5F_ACC_RDONLY, H5P_DEFAULT);
            std::string const path = argv[3];
            auto const cut = path.rfind('/');
            hid_t loc = cut == std::string::npos || cut == 0 ? H5Gopen2(f, "/", H5P_DEFAULT) : H5Gopen2(f, path.substr(0, cut).c_str(), H5P_DEFAULT);
            if (f < 0 || loc < 0) throw std::runtime_error("no such group");
            std::string const name = cut == std::string::npos ? path : path.substr(cut + 1);
            hid_t ds = H5Dopen2(loc, name.c_str(), H5P_DEFAULT);
            if (ds < 0) throw std::runtime_error("no dataset " + path);
            hid_t sp = H5Dget_space(ds);
            bool const scalar = H5Sget_simple_extent_type(sp) == H5S_SCALAR;
            H5Sclose(sp); H5Dclose(ds);
            if (scalar) std::cout << gd::h5::read_string(loc, name) << '\n';
            else for (auto const &s : gd::h5::read_string_list(loc, name)) std::cout << s << '\n';
            return 0;
        }
        std::cerr << "usage: gd_h5tool make-input|steps|positions|context|contacts|dataset|strings ...\n";
        return 1;
    } catch (std::exception const &e) {
        std::cerr << "error: " << e.what() << '\n';
        return 1;
    }
}
