// gd_1kb_kinetics.hpp -- host-side kinetics of the 1 kb chromatin model: cohesin loop extrusion on a 1-d lattice,
// and binding/unbinding of pairwise "glues" between spatially close monomers.
//
// Restates the stochastic rules of 3-sim-1kb/src/simulation/loops/basic_loop_simulator.cpp:22-340,
// glues/glue_simulator.cpp:34-80 and glues/reservoir_sampler.hpp:56-93.  Every random decision is drawn from the
// caller's std::mt19937_64 through the same standard distributions in the same order as the reference, so the
// loop and reservoir trajectories are identical draw for draw (pinned in tests/test_1kb_kinetics.py against the
// reference sources compiled into oracle/_ref and against golden fixtures generated from them).  The glue
// candidate order is micromd's neighbour-search order in the reference (not reproducible); here candidates are
// visited in ascending (i, j) order.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <random>
#include <unordered_set>
#include <utility>
#include <vector>

namespace gd {

// (start, end, id) with id >= 1 for a loaded loop, id == 0 and start == end == chain_length for a free slot --
// the record layout of loops_history (loop_simulator.hpp:11-16, store.cpp:43-48)
struct loop_record { std::size_t start = 0, end = 0, id = 0; };

class loop_extruder {
public:
    using rng = std::mt19937_64;

    loop_extruder(std::size_t chain_length, std::size_t max_loops)
        : _sites(chain_length), _slots(max_loops, loop_record{chain_length, chain_length, 0}) {}

    // --- model parameters
    void set_loading_rate(double r) { _load = r; }
    void set_unloading_rate(double r) { _unload = r; }
    void set_forward_speed(double v) { _fwd = v; }
    void set_backward_speed(double v) { _bwd = v; }
    // an infinite crossing rate means "factors pass freely": no collision checks at all
    void set_crossing_rate(double r) { _collide = !std::isinf(r); _cross = _collide ? r : 0; }
    void set_site_attachability(std::size_t pos, double m) { _sites[pos].attach = m; }
    void set_site_detachability(std::size_t pos, double m) { _sites[pos].detach = m; }
    void add_boundary(std::size_t pos) { _sites[pos].barrier = true; }

    std::size_t chain_length() const { return _sites.size(); }
    std::vector<loop_record> const &loops() const { return _slots; }
    std::size_t loaded() const
    {
        return (std::size_t)std::count_if(_slots.begin(), _slots.end(), [](loop_record const &l) { return l.id != 0; });
    }

    // puts a zero-length loop at `pos` into the first free slot; silently ignored on a barrier or when full
    void load_loop(std::size_t pos)
    {
        if (_sites[pos].barrier) return;
        for (auto &slot : _slots)
            if (slot.id == 0) {
                slot = loop_record{pos, pos, _next_id++};
                _sites[pos].feet += 2;
                return;
            }
    }

    void clear()
    {
        for (auto &slot : _slots) slot = free_slot();
        for (auto &site : _sites) site.feet = 0;
    }

    // one kinetic update over a time interval dt: unloading, loading, then motion of the loop feet
    void step(double dt, rng &random)
    {
        unload_phase(dt, random);
        load_phase(dt, random);
        move_phase(dt, random);
    }

    // loads the steady-state expectation loading/unloading of loops at uniformly random sites
    void preload(rng &random)
    {
        double const ratio = _load / _unload;
        if (!std::isfinite(ratio) || ratio < 0) return;      // unloading rate 0: undefined in the reference (size_t(NaN or inf))
        auto const expected = std::size_t(ratio);
        std::uniform_int_distribution<std::size_t> pick{0, chain_length() - 1};
        for (std::size_t k = 0; k < expected; k++) {
            auto const pos = pick(random);
            if (_sites[pos].barrier) continue;
            if (!std::bernoulli_distribution{_sites[pos].attach}(random)) continue;
            load_loop(pos);
        }
    }

private:
    struct site { double attach = 1, detach = 1; bool barrier = false; int feet = 0; };

    loop_record free_slot() const { return loop_record{chain_length(), chain_length(), 0}; }
    static double probability(double rate, double dt) { return -std::expm1(-rate * dt); }

    void unload_phase(double dt, rng &random)
    {
        if (_unload == 0) return;
        for (auto &slot : _slots) {
            if (slot.id == 0) continue;
            // a factor held at either foot (low detachability) leaves more slowly
            double const rate = _unload * std::min(_sites[slot.start].detach, _sites[slot.end].detach);
            if (std::bernoulli_distribution{probability(rate, dt)}(random)) {
                _sites[slot.start].feet -= 1;
                _sites[slot.end].feet -= 1;
                slot = free_slot();
            }
        }
    }

    void load_phase(double dt, rng &random)
    {
        if (_load == 0) return;
        int const arrivals = std::poisson_distribution<int>{_load * dt}(random);
        std::uniform_int_distribution<std::size_t> pick{0, chain_length() - 1};
        std::poisson_distribution<int> crossings{_cross * dt};      // one object for all arrivals of this update
        for (int k = 0; k < arrivals; k++) {
            auto const pos = pick(random);
            if (_sites[pos].barrier) continue;
            if (_collide && crossings(random) < _sites[pos].feet) continue;      // landing on occupied sites = crossing
            if (!std::bernoulli_distribution{_sites[pos].attach}(random)) continue;
            load_loop(pos);
        }
    }

    // one attempted hop of a loop foot from `pos` to the neighbouring site `dest`
    void hop(std::size_t &pos, std::size_t dest, double speed, double dt, rng &random)
    {
        if (_sites[dest].barrier) return;
        double const eff_dt = dt * _sites[pos].detach * _sites[dest].attach;
        std::poisson_distribution<int> crossings{_cross * eff_dt};
        std::bernoulli_distribution slip{probability(speed, eff_dt)};
        if (_collide && _sites[dest].feet > 0) {
            if (crossings(random) < _sites[dest].feet) return;      // needs one crossing per foot already there
        } else if (!slip(random)) {
            return;
        }
        _sites[pos].feet -= 1;
        _sites[dest].feet += 1;
        pos = dest;
    }

    void move_phase(double dt, rng &random)
    {
        std::size_t const last = chain_length() - 1;
        for (auto &slot : _slots) {
            if (slot.id == 0) continue;
            // upstream foot: extends leftwards, shrinks rightwards; downstream foot: the mirror image
            if (slot.start > 0) hop(slot.start, slot.start - 1, _fwd, dt, random);
            if (slot.start < last) hop(slot.start, slot.start + 1, _bwd, dt, random);
            if (slot.end < last) hop(slot.end, slot.end + 1, _fwd, dt, random);
            if (slot.end > 0) hop(slot.end, slot.end - 1, _bwd, dt, random);
            if (slot.start > slot.end) std::swap(slot.start, slot.end);
        }
    }

    std::vector<site> _sites;
    std::vector<loop_record> _slots;
    std::size_t _next_id = 1;
    double _load = 0, _unload = 0, _fwd = 0, _bwd = 0, _cross = 0;
    bool _collide = false;
};

// Uniform sample of at most `capacity` items from a stream of unknown length (Li's "Algorithm L"), with the
// reference's draw order: uniform_real for the acceptance weight, geometric for the gap, uniform_int for the slot.
template <typename T>
class reservoir {
public:
    explicit reservoir(std::size_t capacity) : _capacity(capacity) { _kept.reserve(capacity); }
    std::vector<T> const &items() const { return _kept; }
    std::size_t seen() const { return _seen; }

    template <typename RNG>
    void feed(T const &value, RNG &random)
    {
        _seen++;
        if (_seen <= _capacity) { _kept.push_back(value); return; }
        if (_seen == _capacity + 1) next_gap(random);
        if (_gap != 0) { _gap--; return; }
        _kept[std::uniform_int_distribution<std::size_t>{0, _capacity - 1}(random)] = value;
        next_gap(random);
    }

private:
    template <typename RNG>
    void next_gap(RNG &random)
    {
        _weight *= std::pow(std::uniform_real_distribution<double>{}(random), 1 / double(_capacity));
        _gap = std::geometric_distribution<std::size_t>{_weight}(random);
    }
    std::size_t _capacity, _seen = 0, _gap = 0;
    std::vector<T> _kept;
    double _weight = 1;
};

struct glue_pair {
    std::uint32_t i = 0, j = 0;
    bool operator==(glue_pair const &o) const { return i == o.i && j == o.j; }
};
struct glue_pair_hash {     // the reference's bucket function (glue_simulator.hpp:29-40): fixes the set's iteration order
    std::size_t operator()(glue_pair const &p) const noexcept { return (std::size_t(p.i & p.j) << 32) | std::size_t(p.i ^ p.j); }
};

class glue_binder {
public:
    using rng = std::mt19937_64;
    using pair_set = std::unordered_set<glue_pair, glue_pair_hash>;

    glue_binder(std::size_t max_glues, double max_distance, double binding_rate, double unbinding_rate, double box)
        : _max(max_glues), _reach(max_distance), _on(binding_rate), _off(unbinding_rate), _box(box) {}

    std::size_t size() const { return _bound.size(); }
    pair_set const &pairs() const { return _bound; }
    bool enabled() const { return _max != 0; }
    double reach() const { return _reach; }

    // positions: (N,3) doubles; candidates: unique pairs (i<j) within reach(), any order (sorted here).
    void update(double dt, double const *positions, std::vector<std::uint32_t> &candidates, rng &random)
    {
        if (_max == 0) return;
        // unbinding: stretched beyond reach, or a rate event (drawn only if still within reach)
        for (auto it = _bound.cbegin(); it != _bound.cend();) {
            auto const cur = it++;
            double d2 = 0;
            for (int k = 0; k < 3; k++) {
                double d = positions[3 * cur->i + k] - positions[3 * cur->j + k];
                d -= _box * std::nearbyint(d / _box);           // periodic_box::shortest_displacement
                d2 += d * d;
            }
            std::bernoulli_distribution release{-std::expm1(-_off * dt)};
            if (std::sqrt(d2) > _reach || release(random)) _bound.erase(cur);
        }
        // binding: every unbound candidate fires with the rate probability; the free capacity is filled uniformly
        std::size_t const npairs = candidates.size() / 2;
        std::vector<std::pair<std::uint32_t, std::uint32_t>> sorted(npairs);
        for (std::size_t k = 0; k < npairs; k++) sorted[k] = {candidates[2 * k], candidates[2 * k + 1]};
        std::sort(sorted.begin(), sorted.end());
        reservoir<glue_pair> picked{_max - _bound.size()};
        for (auto const &c : sorted) {
            glue_pair const pair{c.first, c.second};
            std::bernoulli_distribution capture{-std::expm1(-_on * dt)};
            if (_bound.count(pair) == 0 && capture(random)) picked.feed(pair, random);
        }
        for (auto const &pair : picked.items()) _bound.insert(_bound.end(), pair);
    }

private:
    std::size_t _max;
    double _reach, _on, _off, _box;
    pair_set _bound;
};

}  // namespace gd
