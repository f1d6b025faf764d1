// gd_h5tool -- small command-line companion of the trajectory store (used by the tests and for demos; the
// reference creates its input files with Python/h5py, 5-sim-genome/src/prepare, which is not available here).
//   gd_h5tool make-input <out.h5> <config.json> <chroms.tsv> <ab.f64> <positions.f64> [<nucleolus_bonds.u32> [<nucleolus_ranges.u32>]]
//        chroms.tsv rows: name start end centromere_start centromere_end; raw little-endian arrays (N,2)/(N,3)
//   gd_h5tool make-metadata <out.h5> <dir>             the file `prepare` creates (5-sim-genome/src/prepare/run.py:21-123) from raw tables in <dir>:
//        config.json, ab.f32 (N,2), types.i8 (N), chromosomes.tsv (name start end cen_start cen_end), nucleoli.tsv (name start end),
//        nucleolus_bonds.i32 (M,2)
//   gd_h5tool dump-metadata <file> <dir>               the same tables back out of a file (+ enum.tsv, keys_*.json)
//   gd_h5tool put-positions-f64 <file> <phase> <step> <in.f64>   replaces the snapshot with a float64 positions dataset (refine/run.py:41-46)
//   gd_h5tool steps <file> <phase>                     numerically ordered step list
//   gd_h5tool positions <file> <phase> <step> <out.f64>
//   gd_h5tool context <file> <phase> <step>            prints the JSON context fields
//   gd_h5tool contacts <file> <phase> <step>           prints "i j count" rows
//   gd_h5tool dataset <file> <path> <out.f64>          any numeric dataset as raw doubles; prints its shape
//   gd_h5tool strings <file> <path>                    a string dataset (scalar or 1-d), one item per line
//   gd_h5tool io-selftest                              gd_async_io.hpp: jobs run in submission order, drain() fences, a job's exception reaches the
//        submitting thread, the pool runs every index once and rethrows
//   gd_h5tool packed-check <file> <rows>               the same (rows,3) uint32 and float arrays written by the library's filter pipeline and
//        as hand-packed chunks on a thread pool (gd_h5util.hpp, gd_async_io.hpp); reads both back, compares values, chunking and filters
#include <atomic>
#include <chrono>
#include <cstdio>
#include <fstream>
#include <iostream>
#include <sstream>
#include <string>
#include <thread>
#include <vector>

#include <hdf5.h>

#include "gd_async_io.hpp"
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
        if (cmd == "make-metadata" && argc == 4) {
            std::string const dir = std::string(argv[3]) + "/";
            auto const cfg = slurp(dir + "config.json"), abraw = slurp(dir + "ab.f32"), tyraw = slurp(dir + "types.i8"), nbraw = slurp(dir + "nucleolus_bonds.i32");
            std::size_t const n = tyraw.size();
            if (abraw.size() != n * 2 * sizeof(float)) throw std::runtime_error("ab.f32 and types.i8 disagree on N");
            auto const *ab = reinterpret_cast<float const *>(abraw.data());
            std::vector<gd::ab_factor> abv(n);
            for (std::size_t i = 0; i < n; i++) abv[i] = {ab[2 * i], ab[2 * i + 1]};
            std::vector<std::int8_t> types(tyraw.begin(), tyraw.end());
            std::vector<gd::chromosome_range> chroms;
            std::ifstream tsv(dir + "chromosomes.tsv");
            for (std::string line; std::getline(tsv, line);) {
                std::istringstream ss(line);
                gd::chromosome_range c;
                if (ss >> c.name >> c.start >> c.end >> c.centromere_start >> c.centromere_end) chroms.push_back(c);
            }
            std::vector<gd::index_range> nranges;
            std::vector<std::string> nnames;
            std::ifstream ntsv(dir + "nucleoli.tsv");
            for (std::string line; std::getline(ntsv, line);) {
                std::istringstream ss(line);
                std::string name; gd::index_range r;
                if (ss >> name >> r.begin >> r.end) { nnames.push_back(name); nranges.push_back(r); }
            }
            std::vector<gd::nucleolus_bond> bonds;
            auto const *nb = reinterpret_cast<std::int32_t const *>(nbraw.data());
            for (std::size_t k = 0; k + 1 < nbraw.size() / sizeof(std::int32_t); k += 2) bonds.push_back({(std::size_t)nb[k], (std::size_t)nb[k + 1]});
            gd::trajectory_store store(argv[2], /*create=*/true);
            store.save_metadata(std::string(cfg.begin(), cfg.end()), abv, chroms, nranges, bonds, &types, &nnames);
            store.create_phase_groups();
            return 0;
        }
        if (cmd == "dump-metadata" && argc == 4) {
            std::string const dir = std::string(argv[3]) + "/";
            gd::trajectory_store store(argv[2]);
            auto put = [&](std::string const &name, void const *data, std::size_t bytes) {
                std::ofstream out(dir + name, std::ios::binary);
                out.write(static_cast<char const *>(data), (std::streamsize)bytes);
            };
            auto const cfg = store.load_config_text();
            put("config.json", cfg.data(), cfg.size());
            std::vector<float> ab;
            for (auto const &f : store.load_particle_data()) { ab.push_back((float)f.a); ab.push_back((float)f.b); }
            put("ab.f32", ab.data(), ab.size() * sizeof(float));
            std::vector<std::pair<std::string, int>> members;
            auto const types = store.load_particle_types(&members);
            put("types.i8", types.data(), types.size());
            std::ofstream en(dir + "enum.tsv");
            for (auto const &m : members) en << m.first << ' ' << m.second << '\n';
            std::ofstream ch(dir + "chromosomes.tsv");
            for (auto const &c : store.load_chromosomes()) ch << c.name << ' ' << c.start << ' ' << c.end << ' ' << c.centromere_start << ' ' << c.centromere_end << '\n';
            std::vector<std::int32_t> nr, nb;
            for (auto const &r : store.load_nucleolus_ranges()) { nr.push_back((std::int32_t)r.begin); nr.push_back((std::int32_t)r.end); }
            for (auto const &b : store.load_nucleolus_bonds()) { nb.push_back((std::int32_t)b.nor_index); nb.push_back((std::int32_t)b.nuc_index); }
            put("nucleolus_ranges.i32", nr.data(), nr.size() * sizeof(std::int32_t));
            put("nucleolus_bonds.i32", nb.data(), nb.size() * sizeof(std::int32_t));
            for (char const *ds : {"chromosome_ranges", "centromere_ranges", "nucleolus_ranges"}) {
                auto const keys = store.load_keys(ds);
                put(std::string("keys_") + ds + ".json", keys.data(), keys.size());
            }
            return 0;
        }
        if (cmd == "put-positions-f64" && argc == 6) {
            auto const raw = slurp(argv[5]);
            if (raw.size() % (3 * sizeof(double))) throw std::runtime_error("positions file is not (N,3) float64");
            gd::trajectory_store store(argv[2]);
            store.set_phase(argv[3]);
            store.replace_positions_f64(std::stol(argv[4]), reinterpret_cast<double const *>(raw.data()), raw.size() / (3 * sizeof(double)));
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
        if (cmd == "io-selftest") {
            bool ok = true;
            {   // order and fence
                std::vector<int> seen;
                gd::async_writer w(2);
                for (int k = 0; k < 50; k++) w.submit([&seen, k] { if (k % 7 == 0) std::this_thread::sleep_for(std::chrono::milliseconds(2)); seen.push_back(k); });
                w.drain();
                for (int k = 0; k < 50; k++) ok = ok && seen.size() == 50 && seen[(std::size_t)k] == k;
            }
            {   // a failing job: the error surfaces at the next submit or drain, later jobs are dropped, the writer stays usable
                gd::async_writer w(4);
                int ran = 0;
                w.submit([] { throw std::runtime_error("job failed"); });
                bool caught = false;
                try { for (int k = 0; k < 20; k++) { w.submit([&ran] { ran++; }); std::this_thread::sleep_for(std::chrono::milliseconds(1)); } w.drain(); }
                catch (std::runtime_error const &e) { caught = std::string(e.what()) == "job failed"; }
                int after = 0;
                w.submit([&after] { after = 1; });
                w.drain();
                ok = ok && caught && after == 1;
            }
            {   // pool: every index exactly once, from several threads; an exception comes back to the caller
                gd::thread_pool pool(4);
                std::vector<std::atomic<int>> hits(1000);
                for (int round = 0; round < 20; round++) pool.parallel_for(hits.size(), [&](std::size_t i) { hits[i]++; });
                for (auto &h : hits) ok = ok && h == 20;
                bool caught = false;
                try { pool.parallel_for(100, [](std::size_t i) { if (i == 37) throw std::runtime_error("task failed"); }); }
                catch (std::runtime_error const &) { caught = true; }
                pool.parallel_for(10, [&](std::size_t i) { hits[i]++; });
                ok = ok && caught && hits[0] == 21;
            }
            std::cout << (ok ? "io-selftest ok" : "io-selftest FAILED") << ", " << gd::usable_cpus() << " usable CPUs\n";
            return ok ? 0 : 1;
        }
        if (cmd == "packed-check" && argc == 4) {
            using namespace gd::h5;
            std::size_t const rows = std::stoul(argv[3]);
            std::vector<std::uint32_t> u(3 * rows);
            std::vector<float> x(3 * rows);
            for (std::size_t i = 0; i < 3 * rows; i++) { u[i] = (std::uint32_t)((i * 2654435761u) >> (i % 13)); x[i] = (float)((double)((i * 40503u) % 1000003u) / 65536.0); }
            hid_t f = H5Fcreate(argv[2], H5F_ACC_TRUNC, H5P_DEFAULT, H5P_DEFAULT);
            if (f < 0) throw std::runtime_error("cannot create the file");
            gd::thread_pool pool(gd::usable_threads(8));
            auto pu = plan_packed(rows, 3, sizeof(std::uint32_t)), px = plan_packed(rows, 3, sizeof(float));
            pool.parallel_for(pu.chunk_count(), [&](std::size_t c) { pack_chunk(pu, c, u.data()); });
            pool.parallel_for(px.chunk_count(), [&](std::size_t c) { pack_chunk(px, c, x.data()); });
            { hid a(write_array<std::uint32_t>(f, "u_lib", u.data(), rows, 3, H5T_NATIVE_UINT32, H5T_STD_U32LE)), b(write_packed_array(f, "u_packed", pu, H5T_STD_U32LE)); }
            { hid a(write_array<float>(f, "x_lib", x.data(), rows, 3, H5T_NATIVE_FLOAT, H5T_IEEE_F32LE)), b(write_packed_array(f, "x_packed", px, H5T_IEEE_F32LE)); }
            H5Fclose(f);
            f = H5Fopen(argv[2], H5F_ACC_RDONLY, H5P_DEFAULT);
            auto same_layout = [&](char const *a, char const *b) {
                hid da(H5Dopen2(f, a, H5P_DEFAULT)), db(H5Dopen2(f, b, H5P_DEFAULT));
                hid pa(H5Dget_create_plist(da)), pb(H5Dget_create_plist(db));
                hsize_t ca[2] = {0, 0}, cb[2] = {0, 0};
                if (rows && (H5Pget_chunk(pa, 2, ca) != 2 || H5Pget_chunk(pb, 2, cb) != 2 || ca[0] != cb[0] || ca[1] != cb[1])) return false;
                if (H5Pget_nfilters(pa) != H5Pget_nfilters(pb)) return false;
                for (int k = 0; k < H5Pget_nfilters(pa); k++) {
                    unsigned fa, fb, va[4] = {0}, vb[4] = {0}; size_t na = 4, nb = 4;
                    if (H5Pget_filter2(pa, (unsigned)k, &fa, &na, va, 0, nullptr, nullptr) != H5Pget_filter2(pb, (unsigned)k, &fb, &nb, vb, 0, nullptr, nullptr)) return false;
                    if (na != nb || (na && va[0] != vb[0])) return false;
                }
                return true;
            };
            std::size_t n = 0;
            bool ok = read_array<std::uint32_t>(f, "u_lib", 3, H5T_NATIVE_UINT32, &n) == u && read_array<std::uint32_t>(f, "u_packed", 3, H5T_NATIVE_UINT32, &n) == u &&
                      read_array<float>(f, "x_lib", 3, H5T_NATIVE_FLOAT, &n) == x && read_array<float>(f, "x_packed", 3, H5T_NATIVE_FLOAT, &n) == x &&
                      same_layout("u_lib", "u_packed") && same_layout("x_lib", "x_packed");
            std::cout << (ok ? "packed-check ok " : "packed-check FAILED ") << rows << " rows, " << pu.chunk_count() << " chunks\n";
            return ok ? 0 : 1;
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
        std::cerr << "usage: gd_h5tool make-input|make-metadata|dump-metadata|put-positions-f64|steps|positions|context|contacts|dataset|strings ...\n";
        return 1;
    } catch (std::exception const &e) {
        std::cerr << "error: " << e.what() << '\n';
        return 1;
    }
}
