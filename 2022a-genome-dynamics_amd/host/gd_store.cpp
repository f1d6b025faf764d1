#include "gd_store.hpp"
#include "gd_h5util.hpp"

#include <algorithm>
#include <cmath>
#include <cstring>

#include <json.hpp>   // nlohmann/json 3.x single header

namespace gd {
using namespace h5;


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

static void write_keys(hid_t ds, nlohmann::json const &keys)
{
    hid type(vlen_string_type()), space(H5Screate(H5S_SCALAR));
    hid attr(H5Acreate2(ds, "keys", type, space, H5P_DEFAULT, H5P_DEFAULT));
    std::string const text = keys.dump();
    char const *p = text.c_str();
    check(H5Awrite(attr, type, &p) >= 0, "cannot write keys attribute");
}

static void write_ranges(hid_t group, std::string const &name, std::vector<chromosome_range> const &chroms)
{
    std::vector<int> ranges;
    nlohmann::json keys = nlohmann::json::object();
    for (auto const &c : chroms) {
        keys[c.name] = ranges.size() / 2;
        ranges.push_back((int)c.start); ranges.push_back((int)c.end);
    }
    hid ds(write_array<int>(group, name, ranges.data(), chroms.size(), 2, H5T_NATIVE_INT, H5T_STD_I32LE));
    write_keys(ds, keys);
}

// particle type codes and their descriptive names (prepare/system_definition.py:5-24; the enum of /metadata/particle_types)
static std::pair<char const *, int> const particle_type_names[] = {
    {"active_NOR", 5}, {"silent_NOR", 6}, {"centromere", 4}, {"A", 1}, {"B", 2}, {"u", 3}, {"nucleolus", 7}};

void trajectory_store::save_metadata(std::string const &config_json, std::vector<ab_factor> const &ab, std::vector<chromosome_range> const &chroms,
                                     std::vector<index_range> const &nranges, std::vector<nucleolus_bond> const &bonds,
                                     std::vector<std::int8_t> const *types, std::vector<std::string> const *nucleolus_names)
{
    hid meta(require_group(_file, "metadata"));
    write_string(meta, "config", config_json);
    std::vector<float> abv;
    for (auto const &f : ab) { abv.push_back((float)f.a); abv.push_back((float)f.b); }
    hid(write_array<float>(meta, "ab_factors", abv.data(), ab.size(), 2, H5T_NATIVE_FLOAT, H5T_IEEE_F32LE));
    if (types) {
        check(types->size() == ab.size(), "particle_types and ab_factors differ in length");
        unlink_if_present(meta, "particle_types");
        hid etype(H5Tenum_create(H5T_STD_I8LE));
        for (auto const &m : particle_type_names) { std::int8_t v = (std::int8_t)m.second; H5Tenum_insert(etype, m.first, &v); }
        hsize_t n = types->size();
        hid space(H5Screate_simple(1, &n, nullptr));
        hid ds(H5Dcreate2(meta, "particle_types", etype, space, H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT));
        check(ds >= 0, "cannot create particle_types");
        if (n) check(H5Dwrite(ds, etype, H5S_ALL, H5S_ALL, H5P_DEFAULT, types->data()) >= 0, "cannot write particle_types");
    }
    write_ranges(meta, "chromosome_ranges", chroms);
    std::vector<int> cen, nr, nb;
    nlohmann::json chrom_keys = nlohmann::json::object(), nuc_keys = nlohmann::json::object();
    for (auto const &c : chroms) { chrom_keys[c.name] = cen.size() / 2; cen.push_back((int)c.centromere_start); cen.push_back((int)c.centromere_end); }
    for (std::size_t k = 0; k < nranges.size(); k++) {
        if (nucleolus_names) nuc_keys[nucleolus_names->at(k)] = k;
        nr.push_back((int)nranges[k].begin); nr.push_back((int)nranges[k].end);
    }
    for (auto const &b : bonds) { nb.push_back((int)b.nor_index); nb.push_back((int)b.nuc_index); }
    {
        hid ds(write_array<int>(meta, "centromere_ranges", cen.data(), chroms.size(), 2, H5T_NATIVE_INT, H5T_STD_I32LE));
        write_keys(ds, chrom_keys);          // same keys as chromosome_ranges (prepare/run.py:101-108)
    }
    {
        hid ds(write_array<int>(meta, "nucleolus_ranges", nr.data(), nranges.size(), 2, H5T_NATIVE_INT, H5T_STD_I32LE));
        write_keys(ds, nuc_keys);            // chain name of each span (prepare/run.py:111-118)
    }
    hid(write_array<int>(meta, "nucleolus_bonds", nb.data(), bonds.size(), 2, H5T_NATIVE_INT, H5T_STD_I32LE));
    flush();
}

void trajectory_store::create_phase_groups()
{
    hid snaps(require_group(_file, "snapshots"));
    for (char const *phase : {"spindle", "packing", "relaxation", "interphase"}) hid(require_group(snaps, phase));
    flush();
}

std::vector<std::int8_t> trajectory_store::load_particle_types(std::vector<std::pair<std::string, int>> *enum_members)
{
    hid meta(H5Gopen2(_file, "metadata", H5P_DEFAULT));
    hid ds(H5Dopen2(meta, "particle_types", H5P_DEFAULT));
    check(ds >= 0, "missing dataset particle_types");
    hid space(H5Dget_space(ds)), ftype(H5Dget_type(ds));
    std::vector<std::int8_t> out((std::size_t)std::max<hssize_t>(H5Sget_simple_extent_npoints(space), 0));
    if (!out.empty()) check(H5Dread(ds, H5T_NATIVE_INT8, H5S_ALL, H5S_ALL, H5P_DEFAULT, out.data()) >= 0, "cannot read particle_types");
    if (enum_members && H5Tget_class(ftype) == H5T_ENUM) {
        int const nm = H5Tget_nmembers(ftype);
        for (int k = 0; k < nm; k++) {
            char *name = H5Tget_member_name(ftype, (unsigned)k);
            std::int8_t v = 0;
            H5Tget_member_value(ftype, (unsigned)k, &v);
            enum_members->push_back({name ? name : "", (int)v});
            if (name) H5free_memory(name);
        }
    }
    return out;
}

std::string trajectory_store::load_keys(std::string const &dataset)
{
    hid meta(H5Gopen2(_file, "metadata", H5P_DEFAULT));
    hid ds(H5Dopen2(meta, dataset.c_str(), H5P_DEFAULT));
    check(ds >= 0, "missing dataset " + dataset);
    hid attr(H5Aopen(ds, "keys", H5P_DEFAULT));
    check(attr >= 0, dataset + " has no 'keys' attribute");
    return read_string_from(attr, true);
}

void trajectory_store::replace_positions_f64(long step, double const *xyz, std::size_t n)
{
    hid snaps(require_group(_file, "snapshots")), phase(require_group(snaps, _phase));
    unlink_if_present(phase, std::to_string(step));             // `del phase["0"]`
    hid snap(H5Gcreate2(phase, std::to_string(step).c_str(), H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT));
    check(snap >= 0, "cannot create snapshot group");
    hsize_t dims[2] = {n, 3};
    hid space(H5Screate_simple(2, dims, nullptr));
    hid ds(H5Dcreate2(snap, "positions", H5T_IEEE_F64LE, space, H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT));      // h5py's default: contiguous float64
    check(ds >= 0, "cannot create positions");
    if (n) check(H5Dwrite(ds, H5T_NATIVE_DOUBLE, H5S_ALL, H5S_ALL, H5P_DEFAULT, xyz) >= 0, "cannot write positions");
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

void trajectory_store::save_positions_packed(long step, h5::packed_array const &p)
{
    hid snap(snapshot_group(step));
    hid(write_packed_array(snap, "positions", p, H5T_IEEE_F32LE));
    flush();
}

void trajectory_store::save_contacts_packed(long step, h5::packed_array const &p)
{
    if (p.rows == 0) return;
    hid snap(snapshot_group(step));
    hid(write_packed_array(snap, "contact_map", p, H5T_STD_U32LE));
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
