// gd_ab_store.hpp -- trajectory file of the stage-4 A/B copolymer drivers on the plain HDF5 C API.
//
// Same dataset names, shapes and types as the reference's simulation_store
// (4-sim-ab/box/src/simulation/simulation_store.cc:13-76; identical in 4-sim-ab/sphere/src):
//   /metadata/config        JSON string (pretty-printed, keys sorted -- nlohmann dump)
//   /metadata/ab_factors    float32 (N,2)
//   /metadata/chain_ranges  int32 (C,2)
//   /snapshots/<step>/positions  float32 (N,3), deflate level 1 after a decimal scale-offset of 3 digits
//                                ({.compression = 1, .scaleoffset = 3}, simulation_store.cc:63-67) -- lossy, 1e-3
//   /snapshots/.steps       1-d variable-length strings in the order the snapshots were written
// The reference writes through snsinfu/h5 (not in this image); nothing here is taken from it.
#pragma once
#include <string>
#include <vector>

#include "gd_h5util.hpp"

namespace gd {

class ab_store {
public:
    explicit ab_store(std::string const &filename)
    {
        H5Eset_auto2(H5E_DEFAULT, nullptr, nullptr);
        _file = H5Fcreate(filename.c_str(), H5F_ACC_TRUNC, H5P_DEFAULT, H5P_DEFAULT);      // mode "w"
        if (_file < 0) throw h5_error("cannot create " + filename);
    }
    ~ab_store() { if (_file >= 0) H5Fclose(_file); }
    ab_store(ab_store const &) = delete;
    ab_store &operator=(ab_store const &) = delete;

    void save_config(std::string const &json)
    {
        h5::hid meta(group(_file, "metadata"));
        h5::write_string(meta, "config", json);
    }

    void save_beads(std::vector<float> const &ab)      // (N,2) row-major
    {
        h5::hid meta(group(_file, "metadata"));
        plain(meta, "ab_factors", ab.data(), ab.size() / 2, 2, H5T_NATIVE_FLOAT, H5T_IEEE_F32LE);
    }

    void save_chains(std::vector<int> const &ranges)   // (C,2) row-major
    {
        h5::hid meta(group(_file, "metadata"));
        plain(meta, "chain_ranges", ranges.data(), ranges.size() / 2, 2, H5T_NATIVE_INT, H5T_STD_I32LE);
    }

    void save_snapshot(long step, double const *xyz, std::size_t n)
    {
        std::string const key = std::to_string(step);
        h5::hid snaps(group(_file, "snapshots")), snap(group(snaps, key));
        h5::unlink_if_present(snap, "positions");
        hsize_t dims[2] = {n, 3};
        h5::hid space(H5Screate_simple(2, dims, nullptr)), props(H5Pcreate(H5P_DATASET_CREATE));
        if (n > 0) {
            H5Pset_chunk(props, 2, dims);
            H5Pset_scaleoffset(props, H5Z_SO_FLOAT_DSCALE, 3);
            H5Pset_deflate(props, 1);
        }
        h5::hid ds(H5Dcreate2(snap, "positions", H5T_IEEE_F32LE, space, H5P_DEFAULT, props, H5P_DEFAULT));
        h5::check(ds >= 0, "cannot create positions");
        if (n > 0) h5::check(H5Dwrite(ds, H5T_NATIVE_DOUBLE, H5S_ALL, H5S_ALL, H5P_DEFAULT, xyz) >= 0, "cannot write positions");
        // append the key (simulation_store.cc:69-75)
        auto keys = h5::read_string_list(snaps, ".steps");
        keys.push_back(key);
        h5::write_string_list(snaps, ".steps", keys);
        H5Fflush(_file, H5F_SCOPE_GLOBAL);
    }

private:
    static hid_t group(hid_t parent, std::string const &name)
    {
        hid_t g = h5::exists(parent, name) ? H5Gopen2(parent, name.c_str(), H5P_DEFAULT)
                                           : H5Gcreate2(parent, name.c_str(), H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT);
        h5::check(g >= 0, "cannot open group " + name);
        return g;
    }
    template <typename T>
    static void plain(hid_t loc, std::string const &name, T const *data, std::size_t rows, std::size_t cols, hid_t mem, hid_t file)
    {
        h5::unlink_if_present(loc, name);
        hsize_t dims[2] = {rows, cols};
        h5::hid space(H5Screate_simple(2, dims, nullptr));
        h5::hid ds(H5Dcreate2(loc, name.c_str(), file, space, H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT));
        h5::check(ds >= 0, "cannot create " + name);
        if (rows > 0) h5::check(H5Dwrite(ds, mem, H5S_ALL, H5S_ALL, H5P_DEFAULT, data) >= 0, "cannot write " + name);
    }
    hid_t _file = -1;
};

}  // namespace gd
