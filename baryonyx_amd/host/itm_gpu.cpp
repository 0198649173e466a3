// baryonyx_amd/host/itm_gpu.cpp -- the two drivers of the in-the-middle
// heuristic, host side, over the HIP device layer (include/bxh.h):
//
//   itm::solve    == solver_functor / solve_problem
//                    (lib/src/itm-solver-common.hpp:63-319), one device
//                    instance;
//   itm::optimize == optimize_problem / optimize_functor /
//                    best_solution_recorder / storage
//                    (lib/src/itm-optimizer-common.hpp:93-908): the std::thread
//                    pool becomes ONE batch of `thread` solver instances on
//                    the GPU advanced in lock-step (one launch == one
//                    compute_order::run of every running instance); the
//                    shared population stays on the host.
//
// Every sweep (compute_order::run / push_and_run, lib/src/itm-common.hpp:
// 682-880) is a bxh_sweep call; nothing on this path falls back to the CPU.
#include "private.hpp"
#include "../../include/bxh.h"

#include <algorithm>
#include <functional>
#include <climits>
#include <cmath>
#include <cstring>
#include <numeric>
#include <sstream>
#include <thread>

namespace baryonyx {
namespace itm {
namespace {

// ---------------------------------------------------------------- bit_array
// lib/src/bit-array.hpp:42-337: 64-bit blocks, size / 64 + 1 of them.
struct bit_array
{
    int m_size = 0;
    std::vector<std::uint64_t> w;
    bit_array() = default;
    explicit bit_array(int n)
      : m_size(n)
      , w(static_cast<size_t>(n / 64 + 1), 0)
    {}
    bool empty() const noexcept { return w.empty(); }
    int size() const noexcept { return m_size; }
    int block_size() const noexcept { return static_cast<int>(w.size()); }
    void set(int i) noexcept { w[i >> 6] |= 1ull << (i & 63); }
    void unset(int i) noexcept { w[i >> 6] &= ~(1ull << (i & 63)); }
    void invert(int i) noexcept { w[i >> 6] ^= 1ull << (i & 63); }
    void set(int i, bool v) noexcept { v ? set(i) : unset(i); }
    // bit-array.hpp:144-149 as the reference wrote it (`ref ^= (size_t(val) ^ ref) & mask`):
    // the bit takes `v` at offset 0 of a block and is cleared at every other offset.
    // Only the mutation loop calls it (itm-optimizer-common.hpp:523).
    void set_as_reference(int i, bool v) noexcept
    {
        const std::uint64_t mask = 1ull << (i & 63);
        w[i >> 6] ^= (static_cast<std::uint64_t>(v) ^ w[i >> 6]) & mask;
    }
    int operator[](int i) const noexcept { return static_cast<int>((w[i >> 6] >> (i & 63)) & 1ull); }
    bool operator==(const bit_array& o) const noexcept { return m_size == o.m_size && w == o.w; }
    void ones() noexcept { std::fill(w.begin(), w.end(), ~std::uint64_t{ 0 }); }
    void zeros() noexcept { std::fill(w.begin(), w.end(), std::uint64_t{ 0 }); }
    // bit-array.hpp:183-236 (`assign`; nothing on the path calls it, kept for the
    // reference's own check_bit_array expectations, lib/test/lib.cpp:677-783): the
    // block of `from` keeps its top from % 64 bits, the block of `to` its low
    // 64 - to % 64 bits, the rest of both and every whole block between come from `o`.
    void assign(const bit_array& o, int from, int to)
    {
        size_t first = static_cast<size_t>(from) >> 6, last = static_cast<size_t>(to) >> 6;
        const int fo = from & 63, to_ = to & 63;
        if (fo) {
            const std::uint64_t top = ~std::uint64_t{ 0 } << (64 - fo);
            w[first] = (w[first] & top) | (o.w[first] & ~top);
            ++first;
        }
        if (to_) {
            const std::uint64_t top = ~std::uint64_t{ 0 } << (64 - to_);
            w[last] = (w[last] & ~top) | (o.w[last] & top);
        } else
            ++last; // (`last` is exclusive below; with an offset the reference steps it back by one)
        for (size_t b = first; b + 1 < last; ++b)
            w[b] = o.w[b];
    }
};

std::size_t
hash_of(const bit_array& a) noexcept // bit_array_hash, bit-array.hpp:410-423
{
    std::size_t ret = 0;
    for (auto b : a.w)
        ret ^= std::hash<std::uint64_t>()(b) + 0x9e3779b9 + (ret << 6) + (ret >> 2);
    return ret;
}

// ---------------------------------------------------------------- costs
// default_cost_type / quadratic_cost_type (lib/src/itm-common.hpp:1000-1440),
// host side: normalisation, delta, objective evaluation.  Only the linear
// part is uploaded; a quadratic objective is refused before that.
template<typename Float>
struct cost_type
{
    const objective_function& obj;
    int n;
    std::vector<Float> linear;
    // quadratic_cost_type (itm-common.hpp:1150-1209): for every variable the
    // partners it shares a quadratic term with, in term order
    std::vector<int> q_ptr, q_id;
    std::vector<Float> q_factor;

    cost_type(const objective_function& o, int n_)
      : obj(o)
      , n(n_)
      , linear(static_cast<size_t>(n_), Float{ 0 })
    {
        for (const auto& e : o.elements)
            linear[e.variable_index] += static_cast<Float>(e.factor);
        if (!o.qelements.empty()) {
            q_ptr.assign(static_cast<size_t>(n) + 1, 0);
            for (int v = 0; v != n; ++v) {
                q_ptr[v + 1] = q_ptr[v];
                for (const auto& q : o.qelements)
                    if (q.variable_index_a == v || q.variable_index_b == v) {
                        q_id.push_back(q.variable_index_a == v ? q.variable_index_b
                                                               : q.variable_index_a);
                        q_factor.push_back(static_cast<Float>(q.factor));
                        ++q_ptr[v + 1];
                    }
            }
        }
    }

    bool quadratic() const noexcept { return !q_ptr.empty(); }

    // itm-common.hpp:1129-1132 / :1400-1411
    Float operator[](int i) const noexcept
    {
        Float ret = linear[i];
        if (quadratic())
            for (int f = q_ptr[i], l = q_ptr[i + 1]; f != l; ++f)
                ret += q_factor[f];
        return ret;
    }

    double results(const bit_array& x, double constant) const noexcept
    {
        for (const auto& e : obj.elements)
            if (x[e.variable_index])
                constant += e.factor;
        for (const auto& e : obj.qelements)
            if (x[e.variable_index_a] && x[e.variable_index_b])
                constant += e.factor;
        return constant;
    }

    Float min_abs() const noexcept // :1091-1101 / :1328-1341
    {
        Float mn = std::numeric_limits<Float>::max();
        for (auto v : linear)
            if (v)
                mn = std::min(mn, std::abs(v));
        for (auto v : q_factor)
            if (v)
                mn = std::min(mn, std::abs(v));
        return mn;
    }

    void normalize(solver_parameters::cost_norm_type norm, random_engine& rng)
    {
        using nt = solver_parameters::cost_norm_type;
        Float div{ 0 };
        switch (norm) {
        case nt::none:
            return;
        case nt::l1: // :1343-1360
            for (auto v : linear)
                div += std::abs(v);
            for (auto v : q_factor)
                div += std::abs(v);
            break;
        case nt::l2: // sum of squares, no square root (itm-common.hpp:1107-1117, :1362-1380)
            for (auto v : linear)
                div += v * v;
            for (auto v : q_factor)
                div += v * v;
            break;
        case nt::random: // only the linear part is spread (:1221-1326), then loo
            random_norm(rng);
            div = *std::max_element(linear.begin(), linear.end());
            for (auto v : q_factor)
                div = std::max(v, div);
            break;
        default: // infinity norm: the largest element (itm-common.hpp:1119-1127, :1382-1398)
            div = *std::max_element(linear.begin(), linear.end());
            for (auto v : q_factor)
                div = std::max(v, div);
            break;
        }
        if (std::isnormal(div)) {
            for (auto& v : linear)
                v /= div;
            for (auto& v : q_factor)
                v /= div;
        }
    }

private:
    // make_random_norm, itm-common.hpp:1025-1075: equal costs are spread with
    // uniform draws so that no two are equal.
    void random_norm(random_engine& rng)
    {
        std::vector<std::pair<Float, int>> r(static_cast<size_t>(n));
        for (int i = 0; i != n; ++i)
            r[i] = { linear[i], i };
        std::sort(r.begin(), r.end(), [](const auto& a, const auto& b) { return a.first < b.first; });
        auto spread = [&rng](auto b, auto e, Float lo, Float hi) {
            std::uniform_real_distribution<Float> d(lo, hi);
            for (; b != e; ++b)
                b->first = d(rng);
        };
        auto begin = r.begin();
        for (auto next = r.begin(); next != r.end(); ++next)
            if (next->first != begin->first) {
                if (std::distance(begin, next) > 1)
                    spread(begin, next, begin->first, next->first);
                begin = next;
            }
        if (std::distance(begin, r.end()) > 1) {
            if (begin == r.begin()) {
                spread(begin, r.end(), begin->first, begin->first + 1);
            } else {
                const double x1 = static_cast<double>(std::distance(r.begin(), begin));
                const double ml = static_cast<double>(r.front().first - begin->first) / (0.0 - x1);
                const double pl = static_cast<double>(begin->first) - ml * x1;
                const Float top = static_cast<Float>(std::ceil(ml * static_cast<double>(r.size()) + pl));
                spread(begin, r.end(), begin->first, top);
            }
        }
        std::sort(r.begin(), r.end(), [](const auto& a, const auto& b) { return a.second < b.second; });
        for (int i = 0; i != n; ++i)
            linear[i] = r[i].first;
    }
};

bool
is_better(bool maximize, double lhs, double rhs) noexcept
{
    return maximize ? lhs > rhs : lhs < rhs;
}

double
bad_value(bool maximize) noexcept
{
    return maximize ? -std::numeric_limits<double>::infinity()
                    : std::numeric_limits<double>::infinity();
}

template<typename Float>
bool
init_x(bool maximize, Float cost, int value_if_cost_0) noexcept
{
    if (maximize ? cost > 0 : cost < 0)
        return true;
    if (cost == 0)
        return value_if_cost_0;
    return false;
}

// x initialisers, lib/src/itm-common.hpp:255-374
template<typename Float>
void
init_with_bastert(bit_array& x, const cost_type<Float>& c, bool maximize, int value_if_0)
{
    for (int i = 0; i != x.size(); ++i)
        x.set(i, init_x(maximize, c[i], value_if_0));
}

void
init_with_random(bit_array& x, random_engine& rng, double p)
{
    std::bernoulli_distribution dist(p);
    for (int i = 0; i != x.size(); ++i)
        x.set(i, dist(rng));
}

template<typename Float>
void
init_with_pre_solve(bit_array& x_pessimistic,
                    bit_array& x_optimistic,
                    random_engine& rng,
                    const cost_type<Float>& c,
                    bool maximize,
                    const std::vector<merged_constraint>& constraints,
                    double init_random)
{
    struct rc
    {
        float value;
        int factor;
        int id;
    };
    std::vector<rc> R;
    std::bernoulli_distribution dist(init_random);
    for (const auto& cst : constraints) {
        if (!dist(rng))
            continue;
        const int r_size = static_cast<int>(cst.elements.size());
        R.resize(cst.elements.size());
        for (int i = 0; i != r_size; ++i)
            R[i] = { static_cast<float>(c[cst.elements[i].variable_index]),
                     cst.elements[i].factor,
                     cst.elements[i].variable_index };
        std::shuffle(R.begin(), R.end(), rng);
        std::sort(R.begin(), R.end(), [maximize](const auto& l, const auto& r) {
            return is_better(maximize, l.value, r.value);
        });
        auto stop = [maximize](float v) { return maximize ? v < 0 : v > 0; };
        if (!x_pessimistic.empty()) {
            int sum = 0, best = -2;
            for (int i = -1; i < r_size; ++i) {
                if (cst.min <= sum && sum <= cst.max) {
                    best = i;
                    break;
                }
                if (i + 1 < r_size)
                    sum += R[i + 1].factor;
            }
            int i = 0;
            for (; i <= best; ++i)
                x_pessimistic.set(R[i].id);
            for (; i != r_size; ++i)
                x_pessimistic.unset(R[i].id);
        }
        if (!x_optimistic.empty()) {
            int sum = 0, best = -2;
            for (int i = -1; i < r_size; ++i) {
                if (cst.min <= sum && sum <= cst.max)
                    best = i;
                if (best != -2 && i + 1 < r_size && stop(R[i + 1].value))
                    break;
                if (i + 1 < r_size)
                    sum += R[i + 1].factor;
            }
            int i = 0;
            for (; i <= best; ++i)
                x_optimistic.set(R[i].id);
            for (; i != r_size; ++i)
                x_optimistic.unset(R[i].id);
        }
    }
}

unsigned
engine_state(const random_engine& e)
{
    std::ostringstream os;
    os << e;
    return static_cast<unsigned>(std::stoul(os.str()));
}

void
set_engine_state(random_engine& e, unsigned s)
{
    std::istringstream is(std::to_string(s));
    is >> e;
}

random_engine::result_type
seed_of(const context& ctx) // init_random_generator_seed, itm-common.hpp:1442-1452
{
    if (ctx.parameters.seed <= 0)
        return static_cast<random_engine::result_type>(
          std::chrono::system_clock::now().time_since_epoch().count());
    return static_cast<random_engine::result_type>(ctx.parameters.seed);
}

// ------------------------------------------------------------- device batch
// RAII over a bxh handle: the "Solver" objects of the reference.
struct device_batch
{
    bxh_handle h = nullptr;
    int batch = 0, m = 0, n = 0, nwords = 0;
    std::vector<double> kappa;
    std::vector<std::uint8_t> active;
    std::vector<bxh_status> status;

    template<typename Float>
    device_batch(const context& ctx,
                 const std::vector<merged_constraint>& csts,
                 int variables,
                 const cost_type<Float>& norm_costs,
                 int kind,
                 bool maximize,
                 int batch_,
                 int device = -1)
      : batch(batch_)
      , m(static_cast<int>(csts.size()))
      , n(variables)
    {
        std::vector<int> row_ptr(static_cast<size_t>(m) + 1, 0), col, factor, cmin(m), cmax(m);
        for (int k = 0; k != m; ++k) {
            row_ptr[k + 1] = row_ptr[k] + static_cast<int>(csts[k].elements.size());
            cmin[k] = csts[k].min;
            cmax[k] = csts[k].max;
            for (const auto& e : csts[k].elements) {
                col.push_back(e.variable_index);
                factor.push_back(e.factor);
            }
        }
        std::vector<double> cost(static_cast<size_t>(n));
        for (int i = 0; i != n; ++i)
            cost[i] = static_cast<double>(norm_costs.linear[i]);
        std::vector<double> qf(norm_costs.q_factor.begin(), norm_costs.q_factor.end());
        bxh_problem pb{};
        pb.m = m;
        pb.n = n;
        pb.nnz = static_cast<std::int64_t>(col.size());
        pb.row_ptr = row_ptr.data();
        pb.col = col.data();
        pb.factor = factor.data();
        pb.cmin = cmin.data();
        pb.cmax = cmax.data();
        pb.cost = cost.data();
        if (norm_costs.quadratic()) {
            pb.n_quad = static_cast<std::int32_t>(norm_costs.q_id.size());
            pb.q_ptr = norm_costs.q_ptr.data();
            pb.q_id = norm_costs.q_id.data();
            pb.q_factor = qf.data();
        }
        const int ft = sizeof(Float) == 4 ? BXH_FLOAT : BXH_DOUBLE;
        const int rc = bxh_create(&pb, kind, ft, maximize ? 1 : 0, batch,
                                  device < 0 ? ctx.device : device, &h);
        if (rc == BXH_ERR_OOM)
            throw solver_failure(solver_error_tag::not_enough_memory);
        if (rc != BXH_OK)
            throw solver_failure(solver_error_tag::no_solver_available);
        nwords = bxh_nwords(h);
        kappa.assign(static_cast<size_t>(batch), 0.0);
        active.assign(static_cast<size_t>(batch), 1);
        status.resize(static_cast<size_t>(batch));
    }
    device_batch(const device_batch&) = delete;
    device_batch& operator=(const device_batch&) = delete;
    ~device_batch() { bxh_destroy(h); }

    void check(int rc) const
    {
        if (rc == BXH_ERR_OOM)
            throw solver_failure(solver_error_tag::not_enough_memory);
        if (rc != BXH_OK)
            throw std::runtime_error(std::string("bxh: ") + bxh_last_error());
    }
    void set_x(int i, const bit_array& x) { check(bxh_set_x(h, i, x.w.data())); }
    void get_x(int i, bit_array& x) { check(bxh_get_x(h, i, x.w.data())); }
    void set_rng(int i, const random_engine& e) { check(bxh_set_rng(h, i, engine_state(e))); }
    // compute_order's constructor, once per solve / optimize run
    void order_create(int order) { check(bxh_order_create(h, order)); }
    // compute_order::init of every instance, or of those `mask` selects
    void order_init(int order, const std::vector<std::uint8_t>* mask = nullptr)
    {
        check(bxh_order_init_masked(h, order, mask ? mask->data() : nullptr, nullptr));
    }
    // one compute_order::run / push_and_run of the active instances
    void sweep(double delta, double theta, bool push, double obj_amp, int order)
    {
        bxh_sweep_args a{};
        a.kappa = kappa.data();
        a.delta = delta;
        a.theta = theta;
        a.obj_amp = obj_amp;
        a.is_push = push ? 1 : 0;
        a.order = order;
        a.active = active.data();
        check(bxh_sweep(h, &a, status.data()));
    }
    // the same, without waiting for the outcome (sync() fetches it)
    void sweep_async(double delta, double theta, bool push, double obj_amp, int order)
    {
        bxh_sweep_args a{};
        a.kappa = kappa.data();
        a.delta = delta;
        a.theta = theta;
        a.obj_amp = obj_amp;
        a.is_push = push ? 1 : 0;
        a.order = order;
        a.active = active.data();
        check(bxh_sweep_async(h, &a));
    }
    void sync() { check(bxh_status_sync(h, status.data())); }
    // several sweeps of solver_functor's loop in one launch (bxh_run)
    bool run_supported() const { return bxh_run_supported(h) != 0; }
    bxh_run_status run(double delta, double theta, int order, int max_sweeps, long first_i, long w_limit,
                       double kappa_max, const std::vector<double>& kappa_inc)
    {
        bxh_run_args a{};
        a.kappa = kappa.data();
        a.delta = delta;
        a.theta = theta;
        a.order = order;
        a.max_sweeps = max_sweeps;
        a.first_i = static_cast<int>(std::min<long>(first_i, 0x7fffffff - max_sweeps));
        a.w_limit = static_cast<int>(std::min<long>(w_limit, 0x7fffffff));
        a.kappa_max = kappa_max;
        a.kappa_inc = kappa_inc.data();
        a.active = active.data();
        std::vector<bxh_run_status> rs(static_cast<size_t>(batch));
        check(bxh_run(h, &a, status.data(), rs.data()));
        return rs[0];
    }
    void get_best_x(int i, bit_array& x) { check(bxh_get_best_x(h, i, x.w.data())); }
    // x / engine state of every instance in one transfer
    void get_x_all(std::uint64_t* words) { check(bxh_get_x_all(h, words)); }
    void set_x_all(const std::uint64_t* words) { check(bxh_set_x_all(h, words)); }
    void set_rng_all(const std::uint32_t* states) { check(bxh_set_rng_all(h, states)); }
    // the population of optimize() on the device (bxh.h: bxh_pop_*, bxh_reinit)
    void pop_create(int slots) { check(bxh_pop_create(h, slots)); }
    void pop_set(int slot, const bit_array& x) { check(bxh_pop_set(h, slot, x.w.data())); }
    void pop_store(int slot, int i) { check(bxh_pop_store(h, slot, i)); }
    void reinit(const std::vector<bxh_reinit_job>& jobs)
    {
        check(bxh_reinit(h, static_cast<int>(jobs.size()), jobs.data()));
    }
    void pop_get(int slot, bit_array& x) { check(bxh_pop_get(h, slot, x.w.data())); }
};

// The worker pool of optimize() over several devices: worker b lives on device
// b % D as instance b / D of that device's batch (one process drives all of
// them; the population stays on the host, shared by every worker as the
// reference shares its storage between threads, so there is no collective on
// the path: the incumbent is whatever the shared recorder holds).
struct device_pool
{
    std::vector<std::unique_ptr<device_batch>> dev;
    int workers = 0, m = 0;
    struct ref
    {
        device_batch* d;
        int i;
    };
    std::vector<ref> at;
    struct proxy_kappa
    {
        device_pool& p;
        double& operator[](int b) { return p.at[b].d->kappa[p.at[b].i]; }
    } kappa{ *this };
    struct proxy_active
    {
        device_pool& p;
        std::uint8_t& operator[](int b) { return p.at[b].d->active[p.at[b].i]; }
    } active{ *this };
    struct proxy_status
    {
        device_pool& p;
        bxh_status& operator[](int b) { return p.at[b].d->status[p.at[b].i]; }
    } status{ *this };

    // `norm_costs`: one cost vector for every worker, or one PER WORKER (the random
    // cost norm: optimize_functor normalises with the worker's own engine,
    // itm-optimizer-common.hpp:633-638).  A handle holds one cost vector, so with
    // per-worker costs every worker gets a handle of its own (batch 1, dealt round
    // robin over the devices; the launches of a round still overlap, one stream each).
    std::vector<double> delta_of; // per handle, when the workers' costs differ
    template<typename Float>
    device_pool(const context& ctx,
                const std::vector<merged_constraint>& csts,
                int variables,
                const std::vector<cost_type<Float>>& norm_costs,
                int kind,
                bool maximize,
                int workers_)
      : workers(workers_)
    {
        std::vector<int> ids = ctx.devices.empty() ? std::vector<int>{ ctx.device } : ctx.devices;
        if (static_cast<int>(ids.size()) > workers)
            ids.resize(static_cast<size_t>(workers));
        const bool own_costs = norm_costs.size() > 1;
        const int D = own_costs ? workers : static_cast<int>(ids.size());
        if (own_costs) {
            // One handle per worker = one copy of the matrix per worker (CSR, padded CSC,
            // descriptors: ~64 bytes per element) beside its P (padded, ~1.2 elements).  Said
            // up front, before the first of `workers` bxh_create calls, instead of an
            // allocation failure somewhere in the middle of them: the devices are MI355X,
            // 288 GB each.
            double nnz = 0.0;
            for (const auto& c : csts)
                nnz += static_cast<double>(c.elements.size());
            const double per_handle = nnz * (64.0 + 1.2 * sizeof(Float)) +
                                      static_cast<double>(variables) * (8.0 + sizeof(Float)) +
                                      static_cast<double>(csts.size()) * (32.0 + 2.0 * sizeof(Float));
            const double per_device = per_handle * static_cast<double>((workers + ids.size() - 1) / ids.size());
            if (per_device > 0.85 * 288.0e9)
                throw solver_failure(solver_error_tag::not_enough_memory);
        }
        for (int d = 0; d != D; ++d) {
            const int share = (workers - d + D - 1) / D; // workers d, d + D, ...
            dev.emplace_back(std::make_unique<device_batch>(
              ctx, csts, variables, norm_costs[own_costs ? static_cast<size_t>(d) : 0], kind, maximize, share,
              ids[static_cast<size_t>(d) % ids.size()]));
        }
        for (int b = 0; b != workers; ++b)
            at.push_back(ref{ dev[static_cast<size_t>(b % D)].get(), b / D });
        m = dev[0]->m;
        init_mirrors();
    }
    // Host mirrors of every instance's x and engine state, one block per device:
    // a round of optimize() moves them with ONE transfer per device and direction
    // instead of one per worker (64 synchronising copies per round otherwise).
    std::vector<std::vector<std::uint64_t>> xmirror; // [device][batch * nwords]
    std::vector<std::vector<std::uint32_t>> rmirror; // [device][batch]
    std::vector<char> x_dirty, rng_dirty, launched;  // per device
    int dev_of(int b) const { return b % static_cast<int>(dev.size()); }
    void init_mirrors()
    {
        const size_t D = dev.size();
        xmirror.resize(D);
        rmirror.resize(D);
        x_dirty.assign(D, 0);
        rng_dirty.assign(D, 0);
        launched.assign(D, 0);
        for (size_t d = 0; d != D; ++d) {
            xmirror[d].assign(static_cast<size_t>(dev[d]->batch) * dev[d]->nwords, 0);
            rmirror[d].assign(static_cast<size_t>(dev[d]->batch), 1u);
        }
    }
    void set_x(int b, const bit_array& x)
    {
        const int d = dev_of(b);
        std::copy(x.w.begin(), x.w.end(),
                  xmirror[d].begin() + static_cast<size_t>(at[b].i) * dev[d]->nwords);
        x_dirty[d] = 1;
    }
    void get_x(int b, bit_array& x) const
    {
        const int d = dev_of(b);
        const auto* src = xmirror[d].data() + static_cast<size_t>(at[b].i) * dev[d]->nwords;
        std::copy(src, src + x.w.size(), x.w.begin());
    }
    void set_rng(int b, const random_engine& e)
    {
        const int d = dev_of(b);
        rmirror[d][static_cast<size_t>(at[b].i)] = engine_state(e);
        rng_dirty[d] = 1;
    }
    // ---- the population on the devices (every device holds a copy) ----------
    std::vector<std::vector<bxh_reinit_job>> pending; // restarts not yet launched, per device
    std::vector<char> stale_x;                        // the mirror of x is behind the device
    void pop_create(int slots)
    {
        pending.assign(dev.size(), {});
        stale_x.assign(dev.size(), 0);
        for (auto& d : dev)
            d->pop_create(slots);
    }
    void pop_set_all(int slot, const bit_array& x)
    {
        for (auto& d : dev)
            d->pop_set(slot, x);
    }
    // slot <- x of worker b: a device copy where the worker lives, an upload elsewhere
    void pop_from_worker(int slot, int b, const bit_array& x)
    {
        const int db = dev_of(b);
        for (int d = 0; d != static_cast<int>(dev.size()); ++d) {
            if (d == db)
                dev[d]->pop_store(slot, at[b].i);
            else
                dev[d]->pop_set(slot, x);
        }
    }
    void add_job(int b, bxh_reinit_job jb)
    {
        jb.inst = at[b].i;
        pending[static_cast<size_t>(dev_of(b))].push_back(jb);
    }
    void launch_jobs()
    {
        for (size_t d = 0; d != dev.size(); ++d)
            if (!pending.empty() && !pending[d].empty()) {
                dev[d]->reinit(pending[d]);
                pending[d].clear();
                stale_x[d] = 1;
            }
    }
    // what the host changed since the last sweep goes to the devices
    void flush()
    {
        for (size_t d = 0; d != dev.size(); ++d) {
            if (x_dirty[d])
                dev[d]->set_x_all(xmirror[d].data());
            if (rng_dirty[d])
                dev[d]->set_rng_all(rmirror[d].data());
            x_dirty[d] = rng_dirty[d] = 0;
        }
    }
    void order_create(int order)
    {
        for (auto& d : dev)
            d->order_create(order);
    }
    void order_init(int order, const std::vector<std::uint8_t>* mask)
    {
        flush(); // compute_order::init reads the x the restarts just wrote
        const int D = static_cast<int>(dev.size());
        for (int d = 0; d != D; ++d) {
            std::vector<std::uint8_t> local(static_cast<size_t>(dev[d]->batch), 0);
            bool any = false;
            for (int i = 0; i != dev[d]->batch; ++i) {
                local[i] = mask ? (*mask)[static_cast<size_t>(i * D + d)] : 1;
                any = any || local[i];
            }
            if (any)
                dev[d]->order_init(order, &local);
        }
    }
    // one sweep of the active workers on every device (launched on all, then awaited)
    void sweep_async(double delta, double theta, bool push, double obj_amp, int order)
    {
        flush();
        for (size_t d = 0; d != dev.size(); ++d) {
            bool any = false;
            for (auto a : dev[d]->active)
                any = any || a;
            if (any) {
                dev[d]->sweep_async(delta_of.empty() ? delta : delta_of[d], theta, push, obj_amp, order);
                launched[d] = 1;
            }
        }
    }
    // outcome of the sweeps: the status blocks, and x / the engines into the mirrors
    void sync()
    {
        for (size_t d = 0; d != dev.size(); ++d) {
            dev[d]->sync();
            if (launched[d]) {
                bool moved = false; // (a sweep over empty lists is not even launched)
                for (int i = 0; i != dev[d]->batch; ++i)
                    moved = moved || dev[d]->status[i].rows_in_sweep != 0;
                if (moved || (!stale_x.empty() && stale_x[d])) {
                    dev[d]->get_x_all(xmirror[d].data());
                    if (!stale_x.empty())
                        stale_x[d] = 0;
                }
                launched[d] = 0;
            }
            for (int i = 0; i != dev[d]->batch; ++i)
                rmirror[d][static_cast<size_t>(i)] = dev[d]->status[i].rng;
        }
    }
};

int
device_kind(problem_solver_type t)
{
    switch (t) {
    case problem_solver_type::equalities_01:
        return BXH_EQ_01;
    case problem_solver_type::inequalities_01:
        return BXH_INEQ_01;
    case problem_solver_type::equalities_101:
        return BXH_EQ_101;
    case problem_solver_type::inequalities_101:
        return BXH_INEQ_101;
    default: // equalities_Z also goes to the inequalities-Z solver, itm.hpp:108-109
        return BXH_INEQ_Z;
    }
}

void
fill_result(result& r, const problem& pb, int variables, int constraints)
{
    r.strings = pb.strings;
    r.variable_name = pb.vars.names;
    r.affected_vars = pb.affected_vars;
    r.variables = variables;
    r.constraints = constraints;
}

void
convert(const bit_array& x, double value, solution& sol, int variables)
{
    sol.value = value;
    sol.variables.resize(static_cast<size_t>(variables));
    for (int i = 0; i != variables; ++i)
        sol.variables[i] = static_cast<var_value>(x[i]);
}

// ------------------------------------------------------------------ solve
template<typename Float>
result
solve_problem(const context& ctx, const problem& pb)
{
    if (ctx.start)
        ctx.start(ctx.parameters);

    result ret;
    const int variables = static_cast<int>(pb.vars.values.size());
    const auto constraints = make_merged_constraints(ctx, pb);
    const bool maximize = pb.type == objective_function_type::maximize;
    const auto& p = ctx.parameters;

    if (constraints.empty() || pb.vars.values.empty()) {
        ret.status = result_status::success;
        ret.solutions.resize(1);
        ret.solutions.back().value = pb.objective.value;
    } else {
        random_engine rng(seed_of(ctx));
        const cost_type<Float> original_costs(pb.objective, variables);
        const double cost_constant = pb.objective.value;

        // solver_functor::operator(), itm-solver-common.hpp:63-225
        bit_array x(variables);
        int best_remaining = INT_MAX;
        cost_type<Float> norm_costs(original_costs);
        norm_costs.normalize(p.cost_norm, rng);

        const auto kappa_step = static_cast<Float>(p.kappa_step);
        const auto kappa_max = static_cast<Float>(p.kappa_max);
        const auto alpha = static_cast<Float>(p.alpha);
        const auto theta = static_cast<Float>(p.theta);
        const auto delta = p.delta < 0 ? norm_costs.min_abs() - theta * norm_costs.min_abs()
                                       : static_cast<Float>(p.delta);
        const auto pushing_k_factor = static_cast<Float>(p.pushing_k_factor);
        const auto pushing_objective_amplifier = static_cast<Float>(p.pushing_objective_amplifier);
        const long w_limit = static_cast<long>(p.w);
        const int order = static_cast<int>(p.order);

        device_batch slv(ctx, constraints, variables, norm_costs,
                         device_kind(pb.problem_type), maximize, 1);
        {
            std::bernoulli_distribution choose_mutation(p.init_policy_random);
            bit_array empty_x;
            switch (p.init_policy) {
            case solver_parameters::init_policy_type::pessimistic_solve:
                init_with_pre_solve(x, empty_x, rng, original_costs, maximize, constraints, 1.0);
                break;
            case solver_parameters::init_policy_type::optimistic_solve:
                init_with_pre_solve(empty_x, x, rng, original_costs, maximize, constraints, 1.0);
                break;
            default:
                init_with_bastert(x, original_costs, maximize, 0);
                break;
            }
            for (int i = 0, e = x.size(); i != e; ++i)
                if (choose_mutation(rng))
                    x.invert(i);
        }

        // raw_result<Mode> m_best
        bit_array best_x;
        double best_value = bad_value(maximize);
        long best_loop = 0;
        int best_remaining_constraints = INT_MAX;
        auto store_remaining = [&](int remaining, long i) {
            if (best_remaining_constraints > remaining) {
                slv.get_x(0, x);
                best_x = x;
                best_loop = i;
                best_remaining_constraints = remaining;
            }
        };
        auto store_value = [&](long i) {
            slv.get_x(0, x);
            const double current = original_costs.results(x, cost_constant);
            if (is_better(maximize, current, best_value)) {
                best_x = x;
                best_loop = i;
                best_remaining_constraints = 0;
                best_value = current;
            }
        };

        bool start_push = false;
        auto kappa = static_cast<Float>(p.kappa_min);
        const auto begin = std::chrono::steady_clock::now();
        auto timeout = [&]() {
            if (p.time_limit <= 0)
                return false;
            return std::chrono::duration<double>(std::chrono::steady_clock::now() - begin).count() >
                   p.time_limit;
        };

        slv.set_x(0, x);
        slv.set_rng(0, rng);
        slv.order_create(order);
        slv.order_init(order);

        // Short rows: the loop below runs on the device, up to 256 sweeps per launch
        // (bxh_run: kappa moves there, by the increments worked out here in Float);
        // BXC_SINGLE_SWEEPS=1 keeps one launch per sweep.
        const bool many = slv.run_supported() && order <= 2 && !std::getenv("BXC_SINGLE_SWEEPS");
        std::vector<double> kappa_inc;
        if (many) {
            kappa_inc.resize(static_cast<size_t>(slv.m) + 1);
            for (int r = 0; r <= slv.m; ++r)
                kappa_inc[static_cast<size_t>(r)] = static_cast<double>(static_cast<Float>(
                  kappa_step * std::pow(static_cast<Float>(r) / static_cast<Float>(slv.m), alpha)));
        }
        bool main_done = false;
        if (many) {
            main_done = true;
            long i = 0;
            while (i != p.limit) {
                const long left = p.limit < 0 ? 256 : p.limit - i;
                const int chunk = static_cast<int>(std::min<long>(left, 256));
                slv.kappa[0] = static_cast<double>(kappa);
                const bxh_run_status rs = slv.run(static_cast<double>(delta), static_cast<double>(theta), order,
                                                  chunk, i, w_limit, static_cast<double>(kappa_max), kappa_inc);
                if (rs.best_remaining != INT_MAX && rs.best_remaining < best_remaining) {
                    if (best_remaining_constraints > rs.best_remaining) { // store_remaining
                        slv.get_best_x(0, x);
                        best_x = x;
                        best_loop = rs.best_i;
                        best_remaining_constraints = rs.best_remaining;
                    }
                    best_remaining = rs.best_remaining;
                }
                kappa = static_cast<Float>(rs.kappa);
                if (rs.stop == 1) {
                    store_value(i + rs.sweeps - 1);
                    best_remaining = 0;
                    start_push = true;
                    break;
                }
                if (rs.stop == 2) {
                    ret.status = result_status::kappa_max_reached;
                    break;
                }
                i += rs.sweeps;
                if (timeout()) {
                    ret.status = result_status::time_limit_reached;
                    break;
                }
            }
        }
        for (long i = 0; !main_done && i != p.limit; ++i) {
            slv.kappa[0] = static_cast<double>(kappa);
            slv.sweep(static_cast<double>(delta), static_cast<double>(theta), false, 0.0, order);
            const int remaining = slv.status[0].remaining;

            if (remaining == 0) {
                store_value(i);
                best_remaining = remaining;
                start_push = true;
                break;
            }
            if (remaining < best_remaining) {
                store_remaining(remaining, i);
                best_remaining = remaining;
            }
            if (i > w_limit)
                kappa += kappa_step * std::pow(static_cast<Float>(remaining) /
                                                 static_cast<Float>(slv.m),
                                               alpha);
            if (kappa > kappa_max) {
                ret.status = result_status::kappa_max_reached;
                break;
            }
            if (timeout()) {
                ret.status = result_status::time_limit_reached;
                break;
            }
        }

        if (!start_push) {
            ret.status = result_status::limit_reached;
        } else {
            for (int push = 0; push < p.pushes_limit; ++push) {
                slv.kappa[0] = static_cast<double>(pushing_k_factor * kappa);
                slv.sweep(static_cast<double>(delta), static_cast<double>(theta), true,
                          static_cast<double>(pushing_objective_amplifier), order);
                int remaining = slv.status[0].remaining;
                if (remaining == 0)
                    store_value(-push * p.pushing_iteration_limit - 1);
                if (timeout())
                    break;
                if (many && p.pushing_iteration_limit > 0) {
                    // the iterations behind a push sweep: the same loop, on the device
                    // (the first sweeps one launch each: most of these loops end after one or two
                    // sweeps, and a multi-sweep launch costs three transfers more than a sweep)
                    bool out_of_time = false;
                    int iter = 0;
                    bool ended = false;
                    for (; iter < std::min(p.pushing_iteration_limit, 3); ++iter) {
                        slv.kappa[0] = static_cast<double>(kappa);
                        slv.sweep(static_cast<double>(delta), static_cast<double>(theta), false, 0.0, order);
                        remaining = slv.status[0].remaining;
                        if (remaining == 0) {
                            store_value(-push * p.pushing_iteration_limit - iter - 1);
                            ended = true;
                            break;
                        }
                        if (iter > p.w)
                            kappa += kappa_step * std::pow(static_cast<Float>(remaining) /
                                                             static_cast<Float>(slv.m),
                                                           alpha);
                        if (kappa > kappa_max || timeout()) {
                            ended = true;
                            break;
                        }
                    }
                    while (!ended && iter < p.pushing_iteration_limit) {
                        const int chunk = std::min(p.pushing_iteration_limit - iter, 256);
                        slv.kappa[0] = static_cast<double>(kappa);
                        const bxh_run_status rs = slv.run(static_cast<double>(delta), static_cast<double>(theta),
                                                          order, chunk, iter, static_cast<long>(p.w),
                                                          static_cast<double>(kappa_max), kappa_inc);
                        kappa = static_cast<Float>(rs.kappa);
                        if (rs.stop == 1) {
                            store_value(-push * p.pushing_iteration_limit - (iter + rs.sweeps - 1) - 1);
                            break;
                        }
                        if (rs.stop == 2)
                            break;
                        iter += rs.sweeps;
                        if (timeout()) {
                            out_of_time = true;
                            break;
                        }
                    }
                    (void)out_of_time;
                    continue;
                }
                for (int iter = 0; iter < p.pushing_iteration_limit; ++iter) {
                    slv.kappa[0] = static_cast<double>(kappa);
                    slv.sweep(static_cast<double>(delta), static_cast<double>(theta), false, 0.0, order);
                    remaining = slv.status[0].remaining;
                    if (remaining == 0) {
                        store_value(-push * p.pushing_iteration_limit - iter - 1);
                        break;
                    }
                    if (iter > p.w)
                        kappa += kappa_step * std::pow(static_cast<Float>(remaining) /
                                                         static_cast<Float>(slv.m),
                                                       alpha);
                    if (kappa > kappa_max)
                        break;
                    if (timeout())
                        break;
                }
            }
        }

        if (best_remaining_constraints == 0)
            ret.status = result_status::success;
        if (!best_x.empty()) {
            ret.solutions.resize(1);
            convert(best_x, best_value, ret.solutions[0], variables);
        }
    }

    fill_result(ret, pb, variables, static_cast<int>(constraints.size()));
    if (ctx.finish)
        ctx.finish(ret);
    return ret;
}

// --------------------------------------------------------------- optimize
struct raw_result
{
    bit_array x;
    double value;
    double duration = 0.0;
    std::size_t hash = 0;
    long loop = 0;
    int remaining_constraints = INT_MAX;
    bool is_solution() const noexcept { return remaining_constraints == 0; }
};

// One worker's distributions and engine (local_context,
// itm-optimizer-common.hpp:54-91).
struct local_context
{
    random_engine rng;
    std::normal_distribution<double> choose_sol_dist;
    std::normal_distribution<double> variable_p_dist;
    std::normal_distribution<double> value_p_dist;
    std::uniform_int_distribution<int> bad_solution_choose;
    std::uniform_int_distribution<std::size_t> crossover_dist;
    std::bernoulli_distribution crossover_bastert_insertion;
    local_context(const solver_parameters& p, random_engine::result_type seed)
      : rng(seed)
      , choose_sol_dist(p.init_crossover_solution_selection_mean,
                        p.init_crossover_solution_selection_stddev)
      , variable_p_dist(p.init_mutation_variable_mean, p.init_mutation_variable_stddev)
      , value_p_dist(p.init_mutation_value_mean, p.init_mutation_value_stddev)
      , bad_solution_choose(p.init_population_size / 5, p.init_population_size - 1)
      , crossover_dist(0)
      , crossover_bastert_insertion(p.init_crossover_bastert_insertion)
    {}
};

// storage + best_solution_recorder (itm-optimizer-common.hpp:93-601); the
// batch advances in lock-step, so no mutex is needed.
template<typename Float>
struct recorder
{
    std::vector<int> indices;
    std::vector<raw_result> data;
    bit_array bastert, random;
    // population on the devices (f3): told of every bit vector that enters a slot --
    // (slot, worker whose x it is or -1, the vector); slots `size` and `size + 1`
    // are the Bastert vector and the scratch vector of crossover()
    std::function<void(int, int, const bit_array&)> on_store;
    const cost_type<Float>& costs;
    double cost_constant;
    int size;
    bool maximize;
    std::vector<double> kappa_append;
    std::chrono::steady_clock::time_point start;

    recorder(random_engine& rng,
             int workers,
             const cost_type<Float>& costs_,
             double cost_constant_,
             int variables,
             const std::vector<merged_constraint>& csts,
             int population,
             bool maximize_)
      : indices(static_cast<size_t>(population))
      , data(static_cast<size_t>(population))
      , bastert(variables)
      , random(variables)
      , costs(costs_)
      , cost_constant(cost_constant_)
      , size(population)
      , maximize(maximize_)
      , kappa_append(static_cast<size_t>(workers), 0.0)
      , start(std::chrono::steady_clock::now())
    {
        for (auto& e : data) {
            e.x = bit_array(variables);
            e.value = bad_value(maximize);
        }
        init_with_bastert(bastert, costs, maximize, 0);
        for (int i = 0, e = size / 2; i != e; ++i) {
            data[i].x = bastert;
            std::bernoulli_distribution dist(std::clamp(static_cast<double>(i) / (5 * e), 0.0, 1.0));
            for (int v = 0; v != variables; ++v)
                if (dist(rng))
                    data[i].x.invert(v);
        }
        for (int i = size / 2, e = size; i + 1 < e; i += 2) {
            init_with_random(data[i].x, rng, 0.2);
            init_with_random(data[i + 1].x, rng, 0.8);
            init_with_pre_solve(data[i].x, data[i + 1].x, rng, costs, maximize, csts,
                                std::clamp(static_cast<double>(i) / (5 * e), 0.0, 1.0));
        }
        // hash, objective and violated rows of every start point: independent of
        // each other (no draw involved), so on the host's cores
        auto evaluate = [&](int i) {
            data[i].hash = hash_of(data[i].x);
            data[i].value = costs.results(data[i].x, cost_constant);
            data[i].remaining_constraints = 0;
            for (const auto& cst : csts) {
                int sum = 0;
                for (const auto& el : cst.elements)
                    if (data[i].x[el.variable_index])
                        sum += el.factor;
                if (!(cst.min <= sum && sum <= cst.max))
                    ++data[i].remaining_constraints;
            }
        };
        {
            const unsigned hw = std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
            const int nt = std::max(1, std::min<int>(static_cast<int>(hw), size));
            std::vector<std::thread> pool;
            for (int t = 1; t < nt; ++t)
                pool.emplace_back([&, t]() {
                    for (int i = t; i < size; i += nt)
                        evaluate(i);
                });
            for (int i = 0; i < size; i += nt)
                evaluate(i);
            for (auto& th : pool)
                th.join();
        }
        std::iota(indices.begin(), indices.end(), 0);
        sort();
    }

    void sort()
    {
        std::sort(indices.begin(), indices.end(), [this](int i1, int i2) {
            const int c1 = data[i1].remaining_constraints, c2 = data[i2].remaining_constraints;
            if (c1 < c2)
                return true;
            if (c1 == c2)
                return is_better(maximize, data[i1].value, data[i2].value);
            return false;
        });
    }

    int choose_a_solution(local_context& ctx)
    {
        double value;
        do
            value = ctx.choose_sol_dist(ctx.rng);
        while (value < 0 || value > 1);
        return static_cast<int>(size * value);
    }

    void cross(local_context& ctx, bit_array& x, const bit_array& first, const bit_array& second)
    {
        for (int i = 0, e = x.block_size(); i != e; ++i) {
            const auto x1 = first.w[i], x2 = second.w[i];
            x.w[i] = ((x1 ^ x2) & ctx.crossover_dist(ctx.rng)) ^ x1;
        }
    }

    void crossover(local_context& ctx, bit_array& x) // :373-421
    {
        if (ctx.crossover_bastert_insertion(ctx.rng)) {
            const int first = indices[choose_a_solution(ctx)];
            std::bernoulli_distribution b(0.5);
            if (b(ctx.rng)) {
                if (b(ctx.rng))
                    cross(ctx, x, data[first].x, bastert);
                else
                    data[first].x = bastert;
            } else {
                init_with_random(random, ctx.rng, 0.5);
                if (b(ctx.rng))
                    cross(ctx, x, data[first].x, random);
                else
                    data[first].x = random;
            }
        } else {
            const int first = indices[choose_a_solution(ctx)];
            int second = indices[choose_a_solution(ctx)];
            while (first == second)
                second = indices[choose_a_solution(ctx)];
            cross(ctx, x, data[first].x, data[second].x);
        }
    }

    // ---- the same restart as a job for the device (bxh_reinit): every decision
    // and every draw of reinit() that is not a bit operation on a whole vector is
    // taken here, in the same order; the masks of cross() and the mutation loop
    // run on the device from the engine states noted in the job ----------------
    void cross_job(local_context& ctx, bxh_reinit_job& jb, int first_slot, int second_slot, int blocks)
    {
        jb.cross = 1;
        jb.first = first_slot;
        jb.second = second_slot;
        jb.rng_cross = engine_state(ctx.rng);
        for (int i = 0; i != blocks; ++i)
            (void)ctx.crossover_dist(ctx.rng); // (the device forms the masks; here they only advance the engine)
    }

    void crossover_job(local_context& ctx, bxh_reinit_job& jb, int blocks) // :373-421
    {
        if (ctx.crossover_bastert_insertion(ctx.rng)) {
            const int first = indices[choose_a_solution(ctx)];
            std::bernoulli_distribution b(0.5);
            if (b(ctx.rng)) {
                if (b(ctx.rng))
                    cross_job(ctx, jb, first, size, blocks);
                else {
                    data[first].x = bastert;
                    if (on_store)
                        on_store(first, -1, bastert);
                }
            } else {
                init_with_random(random, ctx.rng, 0.5);
                if (b(ctx.rng)) {
                    if (on_store)
                        on_store(size + 1, -1, random);
                    cross_job(ctx, jb, first, size + 1, blocks);
                } else {
                    data[first].x = random;
                    if (on_store)
                        on_store(first, -1, random);
                }
            }
        } else {
            const int first = indices[choose_a_solution(ctx)];
            int second = indices[choose_a_solution(ctx)];
            while (first == second)
                second = indices[choose_a_solution(ctx)];
            cross_job(ctx, jb, first, second, blocks);
        }
    }

    void mutation_job(local_context& ctx, bxh_reinit_job& jb) // :494-526
    {
        jb.mutate = 0;
        if (!(ctx.value_p_dist.mean() == 0.0 && ctx.value_p_dist.stddev() == 0.0)) {
            double val_p, var_p;
            do
                var_p = ctx.variable_p_dist(ctx.rng);
            while (var_p <= 0.0 || var_p >= 1.0);
            do
                val_p = ctx.value_p_dist(ctx.rng);
            while (val_p < 0.0 || val_p > 1.0);
            jb.mutate = 1;
            jb.var_p = var_p;
            jb.val_p = val_p;
        }
        jb.rng_mut = engine_state(ctx.rng);
    }

    double reinit_job(local_context& ctx, int worker, const solver_parameters& p, bxh_reinit_job& jb,
                      int blocks) // :528-554
    {
        jb = bxh_reinit_job{};
        double kappa = p.kappa_min;
        if (kappa_append[worker] < p.init_kappa_improve_stop) {
            kappa_append[worker] += p.init_kappa_improve_increase;
            kappa = p.kappa_min + (p.kappa_max - p.kappa_min) * kappa_append[worker];
        } else {
            kappa_append[worker] = p.init_kappa_improve_start;
            crossover_job(ctx, jb, blocks);
        }
        mutation_job(ctx, jb);
        return kappa;
    }

    void mutation(local_context& ctx, bit_array& x) // :494-526
    {
        if (ctx.value_p_dist.mean() == 0.0 && ctx.value_p_dist.stddev() == 0.0)
            return;
        double val_p, var_p;
        do
            var_p = ctx.variable_p_dist(ctx.rng);
        while (var_p <= 0.0 || var_p >= 1.0);
        do
            val_p = ctx.value_p_dist(ctx.rng);
        while (val_p < 0.0 || val_p > 1.0);
        std::bernoulli_distribution dist_var_p(var_p), dist_value_p(val_p);
        for (int i = 0, e = x.size(); i != e; ++i)
            if (dist_var_p(ctx.rng))
                x.set_as_reference(i, dist_value_p(ctx.rng));
    }

    double reinit(local_context& ctx, int worker, const solver_parameters& p, bit_array& x) // :528-554
    {
        double kappa = p.kappa_min;
        if (kappa_append[worker] < p.init_kappa_improve_stop) {
            kappa_append[worker] += p.init_kappa_improve_increase;
            kappa = p.kappa_min + (p.kappa_max - p.kappa_min) * kappa_append[worker];
        } else {
            kappa_append[worker] = p.init_kappa_improve_start;
            crossover(ctx, x);
        }
        mutation(ctx, x);
        return kappa;
    }

    int store_worker = -1; // the worker whose device x the vector given to replace() is (or -1)
    void replace(local_context& ctx, const bit_array& x, std::size_t hash, double value,
                 long loop, int remaining)
    {
        const int id = indices[ctx.bad_solution_choose(ctx.rng)];
        data[id].x = x;
        if (on_store)
            on_store(id, store_worker, x);
        data[id].value = value;
        data[id].duration =
          std::chrono::duration<double>(std::chrono::steady_clock::now() - start).count();
        data[id].hash = hash;
        data[id].loop = loop;
        data[id].remaining_constraints = remaining;
        sort();
    }

    void try_advance(local_context& ctx, const bit_array& x, int remaining, long loop) // :556-570
    {
        const auto hash = hash_of(x);
        for (int i = 0; i != size; ++i)
            if (data[i].remaining_constraints == remaining && data[i].hash == hash)
                return;
        replace(ctx, x, hash, costs.results(x, cost_constant), loop, remaining);
    }

    void try_update(local_context& ctx, const bit_array& x, double value, long loop) // :572-585
    {
        const auto hash = hash_of(x);
        for (int i = 0; i != size; ++i)
            if (data[i].remaining_constraints == 0 && data[i].value == value && data[i].hash == hash)
                return;
        replace(ctx, x, hash, value, loop, 0);
    }

    const raw_result& best(int i) const { return data[indices[i]]; }
    const raw_result& worst() const // get_worst, :328-338
    {
        int last = 1;
        const int end = static_cast<int>(indices.size());
        while (last != end && data[indices[last]].is_solution())
            ++last;
        return last == end ? data[indices[last - 1]] : data[indices[last]];
    }
};

// generate_seed, lib/src/itm-common.hpp:1457-1484
std::vector<random_engine::result_type>
generate_seed(random_engine& rng, unsigned count)
{
    using type = random_engine::result_type;
    std::vector<type> ret(count);
    std::uniform_int_distribution<type> dst(std::numeric_limits<type>::min(),
                                            std::numeric_limits<type>::max());
    ret[0] = dst(rng);
    for (unsigned i = 1; i != count; ++i) {
        ret[i] = dst(rng);
        unsigned j = i - 1;
        while (j > 0) {
            if (ret[j] == ret[i]) {
                ret[j] = dst(rng);
                j = i;
            }
            --j;
        }
    }
    return ret;
}

enum phase_t
{
    PH_REINIT,
    PH_MAIN,
    PH_PUSH,
    PH_PUSH_ITER
};

template<typename Float>
result
optimize_problem(const context& ctx, const problem& pb)
{
    result r;
    if (ctx.start)
        ctx.start(ctx.parameters);

    const bool timing = std::getenv("BXC_TIMING") != nullptr;
    const auto t_begin = std::chrono::steady_clock::now();
    auto since_begin = [&]() {
        return std::chrono::duration<double>(std::chrono::steady_clock::now() - t_begin).count();
    };
    auto lap = [&](const char* what) {
        if (timing)
            std::fprintf(stderr, "optimize: %-28s %8.3f ms\n", what, since_begin() * 1e3);
    };
    optimize_timing& tm = last_optimize_timing();
    tm = optimize_timing{};
    const auto constraints = make_merged_constraints(ctx, pb);
    const int variables = static_cast<int>(pb.vars.values.size());
    if (constraints.empty() || pb.vars.values.empty()) {
        r.status = result_status::success;
        r.solutions.resize(1);
        r.solutions.back().value = pb.objective.value;
        fill_result(r, pb, variables, 0);
        return r;
    }
    lap("merged constraints");
    const auto& p = ctx.parameters;
    const bool maximize = pb.type == objective_function_type::maximize;
    random_engine rng(seed_of(ctx));
    const cost_type<Float> cost(pb.objective, variables);
    const double cost_constant = pb.objective.value;

    // `thread` workers -> one batch of solver instances (64 when unset)
    const int workers = p.thread <= 0 ? 64 : p.thread;
    recorder<Float> best_recorder(rng, workers, cost, cost_constant, variables, constraints,
                                  p.init_population_size, maximize);
    const auto seeds = generate_seed(rng, static_cast<unsigned>(workers));

    std::vector<local_context> lctx;
    lctx.reserve(static_cast<size_t>(workers));
    for (int i = 0; i != workers; ++i)
        lctx.emplace_back(p, seeds[i]);

    // optimize_functor::operator() prologue (:627-656), identical for every
    // worker except for the engine the random cost norm draws from
    // (itm-optimizer-common.hpp:633-638, make_random_norm itm-common.hpp:1025-1075):
    // with that norm every worker normalises with its own engine and so holds its
    // own cost vector -- and its own automatic delta.
    const bool own_costs = p.cost_norm == solver_parameters::cost_norm_type::random && workers > 1;
    std::vector<cost_type<Float>> norm_costs_w(own_costs ? static_cast<size_t>(workers) : 1, cost);
    for (size_t b = 0; b != norm_costs_w.size(); ++b)
        norm_costs_w[b].normalize(p.cost_norm, lctx[b].rng);
    const cost_type<Float>& norm_costs = norm_costs_w[0];
    const auto kappa_step = static_cast<Float>(p.kappa_step);
    const auto kappa_max = static_cast<Float>(p.kappa_max);
    const auto alpha = static_cast<Float>(p.alpha);
    const auto theta = static_cast<Float>(p.theta);
    auto delta_for = [&](const cost_type<Float>& c) {
        return p.delta < 0 ? c.min_abs() - theta * c.min_abs() : static_cast<Float>(p.delta);
    };
    const auto delta = delta_for(norm_costs);
    const auto pushing_k_factor = static_cast<Float>(p.pushing_k_factor);
    const auto pushing_objective_amplifier = static_cast<Float>(p.pushing_objective_amplifier);
    const long w_limit = static_cast<long>(p.w);
    const int order = static_cast<int>(p.order);

    lap("merged constraints, recorder");
    tm.host_setup = since_begin();
    device_pool slv(ctx, constraints, variables, norm_costs_w, device_kind(pb.problem_type),
                    maximize, workers);
    if (own_costs)
        for (const auto& c : norm_costs_w)
            slv.delta_of.push_back(static_cast<double>(delta_for(c)));

    struct worker
    {
        bit_array x;
        Float kappa{}, kappa_start{};
        long i = 0;
        int push = 0, iter = 0;
        int best_remaining = INT_MAX;
        long calls = 0;
        phase_t phase = PH_REINIT;
        double value = 0.0;    // objective of x, valid while value_ok
        bool value_ok = false;
    };
    std::vector<worker> wk(static_cast<size_t>(workers));
    for (auto& w : wk)
        w.x = bit_array(variables);

    // f3: the population also lives on the devices, a restart builds the worker's x
    // there (bxh_reinit) from the parents' device copies: no bit vector goes to a
    // device at a restart.  BXC_HOST_REINIT=1 keeps the round-1 way (x built on the
    // host, the batch of x uploaded), for A/B runs and tests.
    const bool device_population = std::getenv("BXC_HOST_REINIT") == nullptr;
    if (device_population) {
        slv.pop_create(best_recorder.size);
        for (int id = 0; id != best_recorder.size; ++id)
            slv.pop_set_all(id, best_recorder.data[static_cast<size_t>(id)].x);
        slv.pop_set_all(best_recorder.size, best_recorder.bastert);
        best_recorder.on_store = [&](int slot, int worker, const bit_array& x) {
            slv.launch_jobs(); // (restarts noted so far read the slot as it was)
            if (worker >= 0)
                slv.pop_from_worker(slot, worker, x);
            else
                slv.pop_set_all(slot, x);
        };
    }
    lap("device pool + population");
    tm.device_setup = since_begin() - tm.host_setup;
    const int blocks = wk[0].x.block_size();
    if (device_population && blocks != slv.dev[0]->nwords)
        throw std::runtime_error("bit_array blocks != device words");

    const auto start = std::chrono::steady_clock::now();
    auto elapsed = [&]() {
        return std::chrono::duration<double>(std::chrono::steady_clock::now() - start).count();
    };
    double next_update = 1.0;
    // optimize() never returns without a positive time limit in the
    // reference (utils.hpp:201-202); keep that contract but allow a sweep
    // budget through context_set_parameters("sweep-budget", n) for tests.
    const double time_limit = p.time_limit;
    const long sweep_budget = ctx.sweep_budget > 0 ? ctx.sweep_budget : LONG_MAX;

    auto kappa_inc = [&](worker& w, int remaining) {
        w.kappa += kappa_step *
                   std::pow(static_cast<Float>(remaining) / static_cast<Float>(slv.m), alpha);
    };
    auto next_push = [&](worker& w) {
        ++w.push;
        w.phase = w.push < p.pushes_limit ? PH_PUSH : PH_REINIT;
    };

    // one compute_order per worker, constructed once (itm-optimizer-common.hpp:655)
    slv.order_create(order);
    std::vector<std::uint8_t> init_mask(static_cast<size_t>(workers), 0);
    std::vector<std::uint8_t> swept(static_cast<size_t>(workers), 0);

    long sweeps = 0;
    for (;;) {
        if (time_limit > 0 && elapsed() > time_limit)
            break;
        if (sweeps >= sweep_budget)
            break;
        const double t_round = elapsed();

        // -- restarts (reinit + compute.init), per worker in id order --------
        if (device_population)
            slv.flush(); // (engine states the bookkeeping changed: in front of the jobs' own)
        bool any_init = false;
        std::fill(init_mask.begin(), init_mask.end(), std::uint8_t{ 0 });
        for (int b = 0; b != workers; ++b) {
            worker& w = wk[b];
            if (w.phase != PH_REINIT)
                continue;
            init_mask[static_cast<size_t>(b)] = 1;
            ++w.calls;
            w.value_ok = false;
            if (device_population) {
                bxh_reinit_job jb;
                w.kappa_start = static_cast<Float>(best_recorder.reinit_job(lctx[b], b, p, jb, blocks));
                slv.add_job(b, jb); // x and the engine of the worker: left on the device by the job
            } else {
                w.kappa_start = static_cast<Float>(best_recorder.reinit(lctx[b], b, p, w.x));
                slv.set_x(b, w.x);
                slv.set_rng(b, lctx[b].rng);
            }
            w.kappa = w.kappa_start;
            w.best_remaining = INT_MAX;
            w.i = 0;
            w.push = 0;
            w.phase = PH_MAIN;
            any_init = true;
        }
        if (device_population)
            slv.launch_jobs();
        if (any_init)
            slv.order_init(order, &init_mask); // compute.init of the restarted workers only

        // -- one sweep of everybody: run() for MAIN / PUSH_ITER, push_and_run
        //    for PUSH (two launches when both kinds are present), then the
        //    bookkeeping of every worker in id order -------------------------
        std::fill(swept.begin(), swept.end(), std::uint8_t{ 0 });
        for (int pass = 0; pass != 2; ++pass) {
            const bool push = pass == 1;
            bool any = false;
            for (int b = 0; b != workers; ++b) {
                const bool mine = (wk[b].phase == PH_PUSH) == push;
                slv.active[b] = mine ? 1 : 0;
                if (mine)
                    swept[static_cast<size_t>(b)] = 1;
                any = any || mine;
                slv.kappa[b] = static_cast<double>(push ? pushing_k_factor : wk[b].kappa);
            }
            if (!any)
                continue;
            slv.sweep_async(static_cast<double>(delta), static_cast<double>(theta), push,
                            push ? static_cast<double>(pushing_objective_amplifier) : 0.0, order);
        }
        const double t_wait = elapsed();
        slv.sync();
        const double t_back = elapsed();
        for (int b = 0; b != workers; ++b) {
            if (!swept[static_cast<size_t>(b)])
                continue;
            worker& w = wk[b];
            const int remaining = slv.status[b].remaining;
            set_engine_state(lctx[b].rng, slv.status[b].rng);
            // (a sweep that updated no row left x as it was: its objective value,
            // n additions, is still the one computed for the sweep before)
            if (slv.status[b].rows_in_sweep != 0)
                w.value_ok = false;
            auto update = [&](long loop) {
                if (!w.value_ok) {
                    slv.get_x(b, w.x);
                    w.value = cost.results(w.x, cost_constant);
                    w.value_ok = true;
                }
                best_recorder.store_worker = b; // (w.x is the worker's x on its device)
                best_recorder.try_update(lctx[b], w.x, w.value, loop);
                best_recorder.store_worker = -1;
            };
            switch (w.phase) {
            case PH_MAIN: // :668-700
                if (remaining == 0) {
                    update(w.i);
                    w.best_remaining = 0;
                    w.push = 0;
                    w.phase = p.pushes_limit > 0 ? PH_PUSH : PH_REINIT;
                    break;
                }
                w.best_remaining = std::min(remaining, w.best_remaining);
                if (w.i > w_limit)
                    kappa_inc(w, remaining);
                ++w.i;
                if (w.kappa > kappa_max || w.i == p.limit) {
                    slv.get_x(b, w.x);
                    best_recorder.store_worker = b;
                    best_recorder.try_advance(lctx[b], w.x, w.best_remaining, p.limit);
                    best_recorder.store_worker = -1;
                    w.phase = PH_REINIT;
                }
                break;
            case PH_PUSH: // :705-723
                if (remaining == 0)
                    update(-w.push * p.pushing_iteration_limit - 1);
                w.kappa = w.kappa_start;
                w.iter = 0;
                if (p.pushing_iteration_limit > 0)
                    w.phase = PH_PUSH_ITER;
                else
                    next_push(w);
                break;
            default: // PH_PUSH_ITER, :724-749
                if (remaining == 0) {
                    update(-w.push * p.pushing_iteration_limit - w.iter - 1);
                    next_push(w);
                    break;
                }
                if (w.iter > p.w)
                    kappa_inc(w, remaining);
                ++w.iter;
                if (w.kappa > kappa_max || w.iter >= p.pushing_iteration_limit)
                    next_push(w);
                break;
            }
            // the host engine may have advanced (try_update): the device
            // copy is refreshed at the next reinit or right here
            if (engine_state(lctx[b].rng) != slv.status[b].rng)
                slv.set_rng(b, lctx[b].rng);
        }
        ++sweeps;
        tm.rounds_host += (t_wait - t_round) + (elapsed() - t_back);
        ++tm.round_count;

        if (ctx.update && elapsed() >= next_update) {
            next_update += 1.0;
            long calls = 0;
            for (const auto& w : wk)
                calls += w.calls;
            const auto& b0 = best_recorder.best(0);
            ctx.update(b0.remaining_constraints, b0.value, b0.loop, b0.duration, calls);
        }
    }

    lap("rounds");
    tm.rounds = since_begin() - tm.host_setup - tm.device_setup;
    // diagnostic (tests): every device's copy of the population == the host's
    if (device_population && std::getenv("BXC_CHECK_POP")) {
        bit_array got(variables);
        for (auto& d : slv.dev)
            for (int id = 0; id != best_recorder.size; ++id) {
                d->pop_get(id, got);
                if (got.w != best_recorder.data[static_cast<size_t>(id)].x.w)
                    throw std::runtime_error("device population differs from the host's in slot " +
                                             std::to_string(id));
            }
    }
    if (ctx.trace) {
        optimize_trace& t = *ctx.trace;
        t.rounds = sweeps;
        for (int i = 0; i != best_recorder.size; ++i) {
            const auto& e = best_recorder.best(i);
            t.pop_remaining.push_back(e.remaining_constraints);
            t.pop_value.push_back(e.value);
            t.pop_hash.push_back(static_cast<std::uint64_t>(e.hash));
            t.pop_x.emplace_back(e.x.w.begin(), e.x.w.end());
        }
        for (int b = 0; b != workers; ++b) {
            t.worker_rng.push_back(engine_state(lctx[b].rng));
            t.worker_calls.push_back(wk[b].calls);
        }
    }
    fill_result(r, pb, variables, static_cast<int>(constraints.size()));
    const auto& first = best_recorder.best(0);
    r.status = first.is_solution() ? result_status::success : result_status::time_limit_reached;
    r.duration = first.duration;
    r.loop = first.loop;
    r.remaining_constraints = first.remaining_constraints;
    switch (p.storage) {
    case solver_parameters::storage_type::one:
        r.solutions.resize(1);
        convert(first.x, first.value, r.solutions[0], variables);
        break;
    case solver_parameters::storage_type::bound:
        r.solutions.resize(2);
        convert(first.x, first.value, r.solutions[0], variables);
        convert(best_recorder.worst().x, best_recorder.worst().value, r.solutions[1], variables);
        break;
    case solver_parameters::storage_type::five:
        r.solutions.resize(5);
        for (int i = 0; i != 5; ++i)
            convert(best_recorder.best(i).x, best_recorder.best(i).value, r.solutions[i], variables);
        break;
    }
    if (ctx.finish)
        ctx.finish(r);
    return r;
}

} // namespace

// tests: the host's bit_array taken through a list of operations (op, p, q) --
// 0 zeros, 1 ones, 2 set, 3 unset, 4 invert, 5 get -> `gets`, 6 / 7 assign from an
// all-zeros / all-ones array over [p, q), 8 set(p, bool q) as the reference writes
// it; returns the block count (lib/test/lib.cpp:677-783, tests/test_oracle_bitarray.py)
int
bit_array_script(int size, const int* ops, int nops, std::uint64_t* words, int* gets)
{
    bit_array a(size), other(size);
    int ng = 0;
    for (int i = 0; i != nops; ++i) {
        const int op = ops[3 * i], p = ops[3 * i + 1], q = ops[3 * i + 2];
        switch (op) {
        case 0: a.zeros(); break;
        case 1: a.ones(); break;
        case 2: a.set(p); break;
        case 3: a.unset(p); break;
        case 4: a.invert(p); break;
        case 5: gets[ng++] = a[p]; break;
        case 6: other.zeros(); a.assign(other, p, q); break;
        case 7: other.ones(); a.assign(other, p, q); break;
        case 8: a.set_as_reference(p, q != 0); break;
        default: return -1;
        }
    }
    std::copy(a.w.begin(), a.w.end(), words);
    return a.block_size();
}

result
solve(const context& ctx, const problem& pb)
{
    switch (ctx.parameters.float_type) {
    case solver_parameters::floating_point_type::float_type:
        return solve_problem<float>(ctx, pb);
    case solver_parameters::floating_point_type::double_type:
        return solve_problem<double>(ctx, pb);
    default: // long double has no device form
        throw solver_failure(solver_error_tag::no_solver_available);
    }
}

result
optimize(const context& ctx, const problem& pb)
{
    switch (ctx.parameters.float_type) {
    case solver_parameters::floating_point_type::float_type:
        return optimize_problem<float>(ctx, pb);
    case solver_parameters::floating_point_type::double_type:
        return optimize_problem<double>(ctx, pb);
    default:
        throw solver_failure(solver_error_tag::no_solver_available);
    }
}

} // namespace itm
} // namespace baryonyx
