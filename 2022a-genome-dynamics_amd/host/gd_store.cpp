#include "gd_store.hpp"

#include <algorithm>
#include <cmath>
#include <cstring>

#include <json.hpp>   // nlohmann/json 3.x single header

namespace gd {
namespace {

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

void check(bool ok, std::string const &what) { if (!ok) throw h5_error("hdf5: " + what); }
bool exists(hid_t loc, std::string const &name) { return H5Lexists(loc, name.c_str(), H5P_DEFAULT) > 0; }
void unlink_if_present(hid_t loc, std::string const &name) { if (exists(loc, name)) H5Ldelete(loc, name.c_str(), H5P_DEFAULT); }

hid_t vlen_string_type()
{
    hid_t t = H5Tcopy(H5T_C_S1);
    H5Tset_size(t, H5T_VARIABLE);
    H5Tset_cset(t, H5T_CSET_UTF8);
    return t;
}

void write_string(hid_t loc, std::string const &name, std::string const &value)
{
    unlink_if_present(loc, name);
    hid type(vlen_string_type()), space(H5Screate(H5S_SCALAR));
    hid ds(H5Dcreate2(loc, name.c_str(), type, space, H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT));
    check(ds >= 0, "cannot create " + name);
    char const *p = value.c_str();
    check(H5Dwrite(ds, type, H5S_ALL, H5S_ALL, H5P_DEFAULT, &p) >= 0, "cannot write " + name);
}

std::string read_string_from(hid_t obj, bool attribute)
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

std::string read_string(hid_t loc, std::string const &name)
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

void write_string_list(hid_t loc, std::string const &name, std::vector<std::string> const &items)
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

std::vector<std::string> read_string_list(hid_t loc, std::string const &name)
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

}  // namespace

float quantize16(double v)
{
    float const scale = 65536.0f;
    return std::nearbyint(static_cast<float>(v) * scale) / scale;
}

trajectory_store::trajectory_store(std::string const &filename, bool create)
{
    H5Eset_auto2(H5E_DEFAULT, nullptr, nullptr);   // errors are reported through exceptions
    _file = create ? H5Fcreate(filename.c_str(), H5F_ACC_TRUNC, H5P_DEFAULT, H5P_DEFAULT) : H5Fopen(filename.c_str(), H5F_ACC_RDWR, H5P_DEFAULT);
    if (_file < 0) throw h5_error("cannot open " + filename);
}

trajectory_store::~trajectory_store() { if (_file >= 0) H5Fclose(_file); }
void trajectory_store::flush() { H5Fflush(_file, H5F_SCOPE_GLOBAL); }

hid_t trajectory_store::require_group(hid_t parent, std::string const &name)
{
    hid_t g = exists(parent, name) ? H5Gopen2(parent, name.c_str(), H5P_DEFAULT)
                                   : H5Gcreate2(parent, name.c_str(), H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT);
    check(g >= 0, "cannot open group " + name);
    return g;
}

std::string trajectory_store::load_config_text()
{
    hid meta(H5Gopen2(_file, "metadata", H5P_DEFAULT));
    check(meta >= 0, "missing /metadata");
    return read_string(meta, "config");
}

std::vector<chromosome_range> trajectory_store::load_chromosomes()
{
    hid meta(H5Gopen2(_file, "metadata", H5P_DEFAULT));
    check(meta >= 0, "missing /metadata");
    std::size_t nc = 0, ncen = 0;
    auto ranges = read_array<int>(meta, "chromosome_ranges", 2, H5T_NATIVE_INT, &nc);
    auto cens = read_array<int>(meta, "centromere_ranges", 2, H5T_NATIVE_INT, &ncen);
    check(nc == ncen, "chromosome_ranges and centromere_ranges differ in length");
    hid ds(H5Dopen2(meta, "chromosome_ranges", H5P_DEFAULT)), attr(H5Aopen(ds, "keys", H5P_DEFAULT));
    check(attr >= 0, "chromosome_ranges has no 'keys' attribute");
    auto const keys = nlohmann::json::parse(read_string_from(attr, true));   // name -> row index
    std::vector<chromosome_range> out(nc);
    for (std::size_t i = 0; i < nc; i++) {
        for (auto it = keys.begin(); it != keys.end(); ++it) if (it.value() == i) { out[i].name = it.key(); break; }
        out[i].start = (std::size_t)ranges[2 * i]; out[i].end = (std::size_t)ranges[2 * i + 1];
        out[i].centromere_start = (std::size_t)cens[2 * i]; out[i].centromere_end = (std::size_t)cens[2 * i + 1];
    }
    return out;
}

std::vector<ab_factor> trajectory_store::load_particle_data()
{
    hid meta(H5Gopen2(_file, "metadata", H5P_DEFAULT));
    std::size_t n = 0;
    auto v = read_array<double>(meta, "ab_factors", 2, H5T_NATIVE_DOUBLE, &n);
    std::vector<ab_factor> out(n);
    for (std::size_t i = 0; i < n; i++) out[i] = {v[2 * i], v[2 * i + 1]};
    return out;
}

std::vector<index_range> trajectory_store::load_nucleolus_ranges()
{
    hid meta(H5Gopen2(_file, "metadata", H5P_DEFAULT));
    std::size_t n = 0;
    auto v = read_array<int>(meta, "nucleolus_ranges", 2, H5T_NATIVE_INT, &n);
    std::vector<index_range> out(n);
    for (std::size_t i = 0; i < n; i++) out[i] = {(std::size_t)v[2 * i], (std::size_t)v[2 * i + 1]};
    return out;
}

std::vector<nucleolus_bond> trajectory_store::load_nucleolus_bonds()
{
    hid meta(H5Gopen2(_file, "metadata", H5P_DEFAULT));
    std::size_t n = 0;
    auto v = read_array<int>(meta, "nucleolus_bonds", 2, H5T_NATIVE_INT, &n);
    std::vector<nucleolus_bond> out(n);
    for (std::size_t i = 0; i < n; i++) out[i] = {(std::size_t)v[2 * i], (std::size_t)v[2 * i + 1]};
    return out;
}

static void write_ranges(hid_t group, std::string const &name, std::vector<chromosome_range> const &chroms)
{
    std::vector<int> ranges;
    nlohmann::json keys;
    for (auto const &c : chroms) {
        keys[c.name] = ranges.size() / 2;
        ranges.push_back((int)c.start); ranges.push_back((int)c.end);
    }
    hid ds(write_array<int>(group, name, ranges.data(), chroms.size(), 2, H5T_NATIVE_INT, H5T_STD_I32LE));
    hid type(vlen_string_type()), space(H5Screate(H5S_SCALAR));
    hid attr(H5Acreate2(ds, "keys", type, space, H5P_DEFAULT, H5P_DEFAULT));
    std::string const text = keys.dump();
    char const *p = text.c_str();
    check(H5Awrite(attr, type, &p) >= 0, "cannot write keys attribute");
}

void trajectory_store::save_metadata(std::string const &config_json, std::vector<ab_factor> const &ab, std::vector<chromosome_range> const &chroms,
                                     std::vector<index_range> const &nranges, std::vector<nucleolus_bond> const &bonds)
{
    hid meta(require_group(_file, "metadata"));
    write_string(meta, "config", config_json);
    std::vector<float> abv;
    for (auto const &f : ab) { abv.push_back((float)f.a); abv.push_back((float)f.b); }
    hid(write_array<float>(meta, "ab_factors", abv.data(), ab.size(), 2, H5T_NATIVE_FLOAT, H5T_IEEE_F32LE));
    write_ranges(meta, "chromosome_ranges", chroms);
    std::vector<int> cen, nr, nb;
    for (auto const &c : chroms) { cen.push_back((int)c.centromere_start); cen.push_back((int)c.centromere_end); }
    for (auto const &r : nranges) { nr.push_back((int)r.begin); nr.push_back((int)r.end); }
    for (auto const &b : bonds) { nb.push_back((int)b.nor_index); nb.push_back((int)b.nuc_index); }
    hid(write_array<int>(meta, "centromere_ranges", cen.data(), chroms.size(), 2, H5T_NATIVE_INT, H5T_STD_I32LE));
    hid(write_array<int>(meta, "nucleolus_ranges", nr.data(), nranges.size(), 2, H5T_NATIVE_INT, H5T_STD_I32LE));
    hid(write_array<int>(meta, "nucleolus_bonds", nb.data(), bonds.size(), 2, H5T_NATIVE_INT, H5T_STD_I32LE));
    flush();
}

void trajectory_store::update_ordered_steps(hid_t phase_group, long step)
{
    std::set<long> sorted;
    for (auto const &s : read_string_list(phase_group, ".steps")) sorted.insert(std::stol(s));
    sorted.insert(step);
    std::vector<std::string> items;
    for (long s : sorted) items.push_back(std::to_string(s));
    write_string_list(phase_group, ".steps", items);
}

hid_t trajectory_store::snapshot_group(long step)
{
    hid snaps(require_group(_file, "snapshots")), phase(require_group(snaps, _phase));
    hid_t g = require_group(phase, std::to_string(step));
    update_ordered_steps(phase, step);
    return g;
}

std::vector<long> trajectory_store::load_steps()
{
    hid snaps(require_group(_file, "snapshots")), phase(require_group(snaps, _phase));
    std::vector<long> out;
    for (auto const &s : read_string_list(phase, ".steps")) out.push_back(std::stol(s));
    return out;
}

void trajectory_store::save_chromosomes(std::vector<chromosome_range> const &chroms)
{
    hid snaps(require_group(_file, "snapshots")), phase(require_group(snaps, _phase)), meta(require_group(phase, "metadata"));
    write_ranges(meta, "chromosome_ranges", chroms);
    flush();
}

void trajectory_store::save_positions(long step, float const *xyz, std::size_t n)
{
    hid snap(snapshot_group(step));
    hid(write_array<float>(snap, "positions", xyz, n, 3, H5T_NATIVE_FLOAT, H5T_IEEE_F32LE));
    flush();
}

void trajectory_store::save_positions(long step, double const *xyz, std::size_t n)
{
    std::vector<float> q(3 * n);
    for (std::size_t i = 0; i < 3 * n; i++) q[i] = quantize16(xyz[i]);
    save_positions(step, q.data(), n);
}

void trajectory_store::save_context(long step, context const &c)
{
    nlohmann::json j;
    j["time"] = c.time; j["bead_scale"] = c.bead_scale; j["bond_scale"] = c.bond_scale;
    j["wall_semiaxes"] = std::vector<double>{c.wall_semiaxes[0], c.wall_semiaxes[1], c.wall_semiaxes[2]};
    j["mean_energy"] = c.mean_energy; j["wall_energy"] = c.wall_energy;
    hid snap(snapshot_group(step));
    write_string(snap, "context", j.dump());
    flush();
}

context trajectory_store::load_context(long step)
{
    hid snap(snapshot_group(step));
    auto const j = nlohmann::json::parse(read_string(snap, "context"));
    context c;
    c.time = j["time"]; c.bead_scale = j["bead_scale"]; c.bond_scale = j["bond_scale"];
    c.mean_energy = j["mean_energy"]; c.wall_energy = j["wall_energy"];
    std::vector<double> v = j["wall_semiaxes"];
    for (int k = 0; k < 3; k++) c.wall_semiaxes[k] = v.at(k);
    return c;
}

void trajectory_store::save_contacts(long step, std::vector<std::array<std::uint32_t, 3>> const &contacts)
{
    if (contacts.empty()) return;
    hid snap(snapshot_group(step));
    hid(write_array<std::uint32_t>(snap, "contact_map", contacts[0].data(), contacts.size(), 3, H5T_NATIVE_UINT32, H5T_STD_U32LE));
    flush();
}

std::vector<std::array<double, 3>> trajectory_store::load_positions(long step)
{
    hid snap(snapshot_group(step));
    std::size_t n = 0;
    auto v = read_array<double>(snap, "positions", 3, H5T_NATIVE_DOUBLE, &n);
    std::vector<std::array<double, 3>> out(n);
    for (std::size_t i = 0; i < n; i++) out[i] = {v[3 * i], v[3 * i + 1], v[3 * i + 2]};
    return out;
}

}  // namespace gd
