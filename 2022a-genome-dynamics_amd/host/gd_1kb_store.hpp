// gd_1kb_store.hpp -- output file of the stage-3 (1 kb) driver on the plain HDF5 C API.
//
// Dataset names, shapes and types of 3-sim-1kb/src/simulation/store.cpp:17-58:
//   /config            effective configuration as a JSON string
//   /config_source     the configuration file verbatim
//   /chain_ranges      int32 (C,2)
//   /positions_history float32 (T,N,3), one frame appended per sample, deflate level 1
//   /loops_history     int32 (T,max_loops,3) rows (start, end, id), only when the model has loop slots
// The reference writes through snsinfu/h5 (not in this image); nothing here is taken from it.
#pragma once
#include <string>
#include <vector>

#include "gd_h5util.hpp"

namespace gd1kb {

class history_store {
public:
    explicit history_store(std::string const &filename)
    {
        H5Eset_auto2(H5E_DEFAULT, nullptr, nullptr);
        _file = H5Fcreate(filename.c_str(), H5F_ACC_TRUNC, H5P_DEFAULT, H5P_DEFAULT);
        if (_file < 0) throw gd::h5_error("cannot create " + filename);
    }
    ~history_store()
    {
        if (_positions >= 0) H5Dclose(_positions);
        if (_loops >= 0) H5Dclose(_loops);
        if (_file >= 0) H5Fclose(_file);
    }
    history_store(history_store const &) = delete;
    history_store &operator=(history_store const &) = delete;

    void save_metadata(std::string const &config, std::string const &source, std::vector<int> const &chain_ranges)
    {
        gd::h5::write_string(_file, "config", config);
        gd::h5::write_string(_file, "config_source", source);
        hsize_t dims[2] = {chain_ranges.size() / 2, 2};
        gd::h5::hid space(H5Screate_simple(2, dims, nullptr));
        gd::h5::hid ds(H5Dcreate2(_file, "chain_ranges", H5T_STD_I32LE, space, H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT));
        gd::h5::check(ds >= 0, "cannot create chain_ranges");
        if (dims[0]) gd::h5::check(H5Dwrite(ds, H5T_NATIVE_INT, H5S_ALL, H5S_ALL, H5P_DEFAULT, chain_ranges.data()) >= 0, "cannot write chain_ranges");
    }

    // positions: (n,3) doubles (stored as float32); loops: (max_loops,3) or empty when the model has no loop slots
    void save_snapshot(double const *positions, std::size_t n, std::vector<long long> const &loops)
    {
        if (_positions < 0) _positions = create_history("positions_history", n, H5T_IEEE_F32LE);
        if (_loops < 0 && !loops.empty()) _loops = create_history("loops_history", loops.size() / 3, H5T_STD_I32LE);
        append(_positions, _frames, n, H5T_NATIVE_DOUBLE, positions);
        if (_loops >= 0) append(_loops, _frames, loops.size() / 3, H5T_NATIVE_LLONG, loops.data());
        _frames++;
        H5Fflush(_file, H5F_SCOPE_GLOBAL);
    }

private:
    hid_t create_history(char const *name, std::size_t rows, hid_t file_type)
    {
        hsize_t dims[3] = {0, rows, 3}, maxdims[3] = {H5S_UNLIMITED, rows, 3}, chunk[3] = {1, rows ? rows : 1, 3};
        gd::h5::hid space(H5Screate_simple(3, dims, maxdims)), props(H5Pcreate(H5P_DATASET_CREATE));
        H5Pset_chunk(props, 3, chunk);
        H5Pset_deflate(props, 1);
        hid_t ds = H5Dcreate2(_file, name, file_type, space, H5P_DEFAULT, props, H5P_DEFAULT);
        gd::h5::check(ds >= 0, std::string("cannot create ") + name);
        return ds;
    }
    static void append(hid_t ds, hsize_t frame, std::size_t rows, hid_t mem_type, void const *data)
    {
        hsize_t dims[3] = {frame + 1, rows, 3};
        gd::h5::check(H5Dset_extent(ds, dims) >= 0, "cannot extend history");
        gd::h5::hid fspace(H5Dget_space(ds));
        hsize_t start[3] = {frame, 0, 0}, count[3] = {1, rows, 3};
        H5Sselect_hyperslab(fspace, H5S_SELECT_SET, start, nullptr, count, nullptr);
        gd::h5::hid mspace(H5Screate_simple(3, count, nullptr));
        if (rows) gd::h5::check(H5Dwrite(ds, mem_type, mspace, fspace, H5P_DEFAULT, data) >= 0, "cannot append to history");
    }
    hid_t _file = -1, _positions = -1, _loops = -1;
    hsize_t _frames = 0;
};

}  // namespace gd1kb
