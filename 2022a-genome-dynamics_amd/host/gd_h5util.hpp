// gd_h5util.hpp -- thin helpers over the HDF5 C API shared by the trajectory stores of the drivers
// (stage 5: gd_store, stage 4: gd_ab_store, stage 3: gd_1kb_store).
#pragma once
#include <hdf5.h>

#include <zlib.h>

#include <algorithm>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

namespace gd {

struct h5_error : std::runtime_error { using std::runtime_error::runtime_error; };

namespace h5 {

struct hid {   // closes whatever kind of handle it owns
    hid_t id = -1;
    explicit hid(hid_t i = -1) : id(i) {}
    hid(hid const &) = delete;
    hid &operator=(hid const &) = delete;
    ~hid()
    {
        if (id < 0) return;
        switch (H5Iget_type(id)) {
        case H5I_GROUP: H5Gclose(id); break;
        case H5I_DATASET: H5Dclose(id); break;
        case H5I_DATASPACE: H5Sclose(id); break;
        case H5I_DATATYPE: H5Tclose(id); break;
        case H5I_ATTR: H5Aclose(id); break;
        case H5I_GENPROP_LST: H5Pclose(id); break;
        default: break;
        }
    }
    operator hid_t() const { return id; }
};

inline void check(bool ok, std::string const &what) { if (!ok) throw h5_error("hdf5: " + what); }
inline bool exists(hid_t loc, std::string const &name) { return H5Lexists(loc, name.c_str(), H5P_DEFAULT) > 0; }
inline void unlink_if_present(hid_t loc, std::string const &name) { if (exists(loc, name)) H5Ldelete(loc, name.c_str(), H5P_DEFAULT); }

inline hid_t vlen_string_type()
{
    hid_t t = H5Tcopy(H5T_C_S1);
    H5Tset_size(t, H5T_VARIABLE);
    H5Tset_cset(t, H5T_CSET_UTF8);
    return t;
}

inline void write_string(hid_t loc, std::string const &name, std::string const &value)
{
    unlink_if_present(loc, name);
    hid type(vlen_string_type()), space(H5Screate(H5S_SCALAR));
    hid ds(H5Dcreate2(loc, name.c_str(), type, space, H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT));
    check(ds >= 0, "cannot create " + name);
    char const *p = value.c_str();
    check(H5Dwrite(ds, type, H5S_ALL, H5S_ALL, H5P_DEFAULT, &p) >= 0, "cannot write " + name);
}

inline std::string read_string_from(hid_t obj, bool attribute)
{
    hid ftype(attribute ? H5Aget_type(obj) : H5Dget_type(obj));
    std::string out;
    if (H5Tis_variable_str(ftype) > 0) {
        hid mtype(vlen_string_type());
        char *p = nullptr;
        herr_t rc = attribute ? H5Aread(obj, mtype, &p) : H5Dread(obj, mtype, H5S_ALL, H5S_ALL, H5P_DEFAULT, &p);
        check(rc >= 0, "cannot read string");
        if (p) { out = p; H5free_memory(p); }
    } else {   // fixed-length string
        std::size_t n = H5Tget_size(ftype);
        std::vector<char> buf(n + 1, 0);
        herr_t rc = attribute ? H5Aread(obj, ftype, buf.data()) : H5Dread(obj, ftype, H5S_ALL, H5S_ALL, H5P_DEFAULT, buf.data());
        check(rc >= 0, "cannot read string");
        out = buf.data();
    }
    return out;
}

inline std::string read_string(hid_t loc, std::string const &name)
{
    hid ds(H5Dopen2(loc, name.c_str(), H5P_DEFAULT));
    check(ds >= 0, "missing dataset " + name);
    return read_string_from(ds, false);
}

// (rows, cols) array, <= 1 MiB chunks, shuffle + deflate 6 (simulation_store.cc:318-345)
template <typename T>
hid_t write_array(hid_t loc, std::string const &name, T const *data, std::size_t rows, std::size_t cols, hid_t mem_type, hid_t file_type)
{
    unlink_if_present(loc, name);
    hsize_t dims[2] = {rows, cols};
    hid space(H5Screate_simple(2, dims, nullptr)), props(H5Pcreate(H5P_DATASET_CREATE));
    if (rows > 0) {
        hsize_t chunk[2] = {std::min<hsize_t>((1024 * 1024) / (sizeof(T) * cols), rows), cols};
        H5Pset_chunk(props, 2, chunk);
        H5Pset_shuffle(props);
        H5Pset_deflate(props, 6);
    }
    hid_t ds = H5Dcreate2(loc, name.c_str(), file_type, space, H5P_DEFAULT, props, H5P_DEFAULT);
    check(ds >= 0, "cannot create " + name);
    if (rows > 0) check(H5Dwrite(ds, mem_type, H5S_ALL, H5S_ALL, H5P_DEFAULT, data) >= 0, "cannot write " + name);
    return ds;
}

// The same dataset written in two steps, so that the expensive one needs no HDF5 call and can run on any thread:
//   pack_chunk         one chunk through the dataset's filter pipeline by hand -- the byte shuffle of H5Z_FILTER_SHUFFLE (byte b of
//                      every element, plane after plane), then deflate level 6 as H5Z_FILTER_DEFLATE stores it (a zlib stream);
//                      the last chunk of a dataset is padded to the full chunk size with zeros, as the library does;
//   write_packed_array the dataset with write_array's creation properties, its chunks handed to the file as they are
//                      (H5Dwrite_chunk, filter mask 0 = every filter applied): readers decode them like any other chunk.
// A batched driver writes R trajectory files at every snapshot; deflate is ~25 ms per MiB on one core and the HDF5 library is
// not re-entrant, so the chunks are packed by a pool of threads and only the cheap part stays serial (gd_async_io.hpp).
struct packed_array {
    std::size_t rows = 0, cols = 0, chunk_rows = 0, elem = 0;
    std::vector<std::vector<unsigned char>> chunks;
    std::size_t chunk_count() const { return rows ? (rows + chunk_rows - 1) / chunk_rows : 0; }
};

inline packed_array plan_packed(std::size_t rows, std::size_t cols, std::size_t elem)
{
    packed_array p;
    p.rows = rows; p.cols = cols; p.elem = elem;
    p.chunk_rows = rows ? std::min<std::size_t>((1024 * 1024) / (elem * cols), rows) : 0;
    p.chunks.resize(p.chunk_count());
    return p;
}

inline void pack_chunk(packed_array &p, std::size_t c, void const *data)      // data: the whole (rows, cols) array
{
    std::size_t const row_bytes = p.cols * p.elem, chunk_bytes = p.chunk_rows * row_bytes;
    std::size_t const first = c * p.chunk_rows, valid = std::min(p.chunk_rows, p.rows - first) * row_bytes;
    std::vector<unsigned char> shuffled(chunk_bytes, 0);
    unsigned char const *src = static_cast<unsigned char const *>(data) + first * row_bytes;
    std::size_t const n = chunk_bytes / p.elem, nvalid = valid / p.elem;
    for (std::size_t b = 0; b < p.elem; b++) {
        unsigned char *plane = shuffled.data() + b * n;
        for (std::size_t i = 0; i < nvalid; i++) plane[i] = src[i * p.elem + b];
    }
    uLongf size = compressBound((uLong)chunk_bytes);
    p.chunks[c].resize(size);
    if (compress2(p.chunks[c].data(), &size, shuffled.data(), (uLong)chunk_bytes, 6) != Z_OK) throw h5_error("hdf5: deflate failed");
    p.chunks[c].resize(size);
}

inline hid_t write_packed_array(hid_t loc, std::string const &name, packed_array const &p, hid_t file_type)
{
    unlink_if_present(loc, name);
    hsize_t dims[2] = {p.rows, p.cols};
    hid space(H5Screate_simple(2, dims, nullptr)), props(H5Pcreate(H5P_DATASET_CREATE));
    if (p.rows > 0) {
        hsize_t chunk[2] = {p.chunk_rows, p.cols};
        H5Pset_chunk(props, 2, chunk);
        H5Pset_shuffle(props);
        H5Pset_deflate(props, 6);
    }
    hid_t ds = H5Dcreate2(loc, name.c_str(), file_type, space, H5P_DEFAULT, props, H5P_DEFAULT);
    check(ds >= 0, "cannot create " + name);
    for (std::size_t c = 0; c < p.chunk_count(); c++) {
        hsize_t offset[2] = {c * p.chunk_rows, 0};
        check(H5Dwrite_chunk(ds, H5P_DEFAULT, 0, offset, p.chunks[c].size(), p.chunks[c].data()) >= 0, "cannot write a chunk of " + name);
    }
    return ds;
}

template <typename T>
std::vector<T> read_array(hid_t loc, std::string const &name, std::size_t cols, hid_t mem_type, std::size_t *rows_out = nullptr)
{
    hid ds(H5Dopen2(loc, name.c_str(), H5P_DEFAULT));
    check(ds >= 0, "missing dataset " + name);
    hid space(H5Dget_space(ds));
    hsize_t dims[2] = {0, 0};
    int nd = H5Sget_simple_extent_ndims(space);
    check(nd == 2 || nd == 1, name + ": expected a 2-d dataset");
    H5Sget_simple_extent_dims(space, dims, nullptr);
    if (nd == 1) dims[1] = dims[0] ? cols : 0;
    std::vector<T> out;
    if (dims[0] > 0 && dims[1] > 0) {   // empty datasets have dataspace {0,0} (simulation_store.cc:130-137)
        check(dims[1] == cols, name + ": wrong number of columns");
        out.resize(dims[0] * cols);
        check(H5Dread(ds, mem_type, H5S_ALL, H5S_ALL, H5P_DEFAULT, out.data()) >= 0, "cannot read " + name);
    }
    if (rows_out) *rows_out = out.size() / cols;
    return out;
}

inline void write_string_list(hid_t loc, std::string const &name, std::vector<std::string> const &items)
{
    unlink_if_present(loc, name);
    hsize_t n = items.size();
    hid type(vlen_string_type()), space(H5Screate_simple(1, &n, nullptr));
    hid ds(H5Dcreate2(loc, name.c_str(), type, space, H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT));
    check(ds >= 0, "cannot create " + name);
    std::vector<char const *> p;
    for (auto const &s : items) p.push_back(s.c_str());
    if (n) check(H5Dwrite(ds, type, H5S_ALL, H5S_ALL, H5P_DEFAULT, p.data()) >= 0, "cannot write " + name);
}

inline std::vector<std::string> read_string_list(hid_t loc, std::string const &name)
{
    std::vector<std::string> out;
    if (!exists(loc, name)) return out;
    hid ds(H5Dopen2(loc, name.c_str(), H5P_DEFAULT)), space(H5Dget_space(ds)), type(vlen_string_type());
    hssize_t n = H5Sget_simple_extent_npoints(space);
    if (n <= 0) return out;
    std::vector<char *> p((std::size_t)n, nullptr);
    check(H5Dread(ds, type, H5S_ALL, H5S_ALL, H5P_DEFAULT, p.data()) >= 0, "cannot read " + name);
    for (auto q : p) out.emplace_back(q ? q : "");
    H5Dvlen_reclaim(type, space, H5P_DEFAULT, p.data());
    return out;
}


}  // namespace h5
}  // namespace gd
