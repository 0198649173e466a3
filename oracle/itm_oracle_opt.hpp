// oracle/itm_oracle_opt.hpp -- TEST INFRASTRUCTURE (see itm_oracle.hpp): a CPU
// restatement of baryonyx's optimize() over the oracle Solver:
//
//   local_context            lib/src/itm-optimizer-common.hpp:54-91
//   storage                  :93-458   (population, crossover, insert, sort)
//   best_solution_recorder   :460-601  (mutation, reinit, try_advance, try_update)
//   optimize_functor         :603-753  (the restart loop of one worker)
//   optimize_problem         :776-908  (recorder, generate_seed, the worker pool)
//   generate_seed            lib/src/itm-common.hpp:1457-1484
//
// The reference runs the workers as free-running std::threads that meet in the
// storage under a shared_mutex and stops them on a wall clock, so its outcome
// depends on thread timing.  Parity needs a schedule: here the workers are
// still one thread each running the reference's loop as written, but a
// conductor lets them advance in ROUNDS --
//   1. the workers that are about to restart run reinit + compute.init, one
//      after the other in worker order,
//   2. every worker runs ONE sweep (compute.run or push_and_run), all in
//      parallel (a sweep touches nothing shared),
//   3. every worker runs the code between that sweep and its next one
//      (try_update / try_advance, the kappa schedule, loop control), one after
//      the other in worker order,
// and the run ends after a fixed number of rounds instead of a time limit.
// The GPU path (baryonyx_amd/host/itm_gpu.cpp: optimize_problem) runs the same
// schedule with one device batch as the pool.
#ifndef ORACLE_ITM_ORACLE_OPT_HPP
#define ORACLE_ITM_ORACLE_OPT_HPP

#include "itm_oracle.hpp"

#include <ctime>
#include <condition_variable>
#include <mutex>
#include <sstream>
#include <random>
#include <thread>

namespace orc {

// local_context, itm-optimizer-common.hpp:54-91
struct LocalContext
{
    random_engine rng;
    std::normal_distribution<double> choose_sol_dist;
    std::normal_distribution<double> variable_p_dist;
    std::normal_distribution<double> value_p_dist;
    std::uniform_int_distribution<int> bad_solution_choose;
    std::uniform_int_distribution<std::size_t> crossover_dist;
    std::bernoulli_distribution crossover_bastert_insertion;
    const double init_kappa_improve_start;
    const double init_kappa_improve_increase;
    const double init_kappa_improve_stop;
    const unsigned thread_id;

    LocalContext(const Params& p, unsigned thread_id_, random_engine::result_type seed)
      : rng(seed)
      , choose_sol_dist(p.init_crossover_solution_selection_mean,
                        p.init_crossover_solution_selection_stddev)
      , variable_p_dist(p.init_mutation_variable_mean, p.init_mutation_variable_stddev)
      , value_p_dist(p.init_mutation_value_mean, p.init_mutation_value_stddev)
      , bad_solution_choose(p.init_population_size / 5, p.init_population_size - 1)
      , crossover_dist(0)
      , crossover_bastert_insertion(p.init_crossover_bastert_insertion)
      , init_kappa_improve_start(p.init_kappa_improve_start)
      , init_kappa_improve_increase(p.init_kappa_improve_increase)
      , init_kappa_improve_stop(p.init_kappa_improve_stop)
      , thread_id(thread_id_)
    {}
};

// storage, itm-optimizer-common.hpp:93-458 (the mutex is the conductor's order)
template<typename Float>
class Storage
{
    std::vector<int> m_indices;
    std::vector<RawResult> m_data;
    BitArray m_bastert;
    BitArray m_random;
    const Cost<Float>& costs;
    double cost_constant;
    const int m_size;
    const bool maximize;

    void replace_result(int id, const BitArray& x, double value, std::size_t hash, long loop,
                        int remaining_constraints) noexcept // :128-144
    {
        m_data[id].x = x;
        m_data[id].value = value;
        m_data[id].hash = hash;
        m_data[id].loop = loop;
        m_data[id].remaining_constraints = remaining_constraints;
    }
    int choose_a_bad_solution(LocalContext& ctx) { return ctx.bad_solution_choose(ctx.rng); }
    int choose_a_solution(LocalContext& ctx) // :151-159
    {
        double value;
        do
            value = ctx.choose_sol_dist(ctx.rng);
        while (value < 0 || value > 1);
        return static_cast<int>(m_size * value);
    }
    void sort() noexcept // :424-442
    {
        std::sort(m_indices.begin(), m_indices.end(), [this](int i1, int i2) {
            const int cst_1 = m_data[i1].remaining_constraints;
            const int cst_2 = m_data[i2].remaining_constraints;
            if (cst_1 < cst_2)
                return true;
            if (cst_1 == cst_2)
                return is_better_solution(maximize, m_data[i1].value, m_data[i2].value);
            return false;
        });
    }

public:
    Storage(random_engine& rng,
            const Cost<Float>& costs_,
            double cost_constant_,
            int population_size,
            int variables,
            const Constraints& cs,
            bool maximize_) // :162-226
      : m_indices(static_cast<size_t>(population_size))
      , m_data(static_cast<size_t>(population_size), RawResult(maximize_))
      , m_bastert(variables)
      , m_random(variables)
      , costs(costs_)
      , cost_constant(cost_constant_)
      , m_size(population_size)
      , maximize(maximize_)
    {
        for (auto& elem : m_data)
            elem.x = BitArray(variables);
        init_with_bastert(m_bastert, costs_, maximize, variables, 0);
        for (int i = 0, e = m_size / 2; i != e; ++i) {
            m_data[i].x = m_bastert;
            std::bernoulli_distribution dist(std::clamp(static_cast<double>(i) / (5 * e), 0.0, 1.0));
            for (int v = 0; v != variables; ++v)
                if (dist(rng))
                    m_data[i].x.invert(v);
        }
        for (int i = m_size / 2, e = m_size; i + 1 < e; i += 2) {
            init_with_random(m_data[i].x, rng, variables, 0.2);
            init_with_random(m_data[i + 1].x, rng, variables, 0.8);
            init_with_pre_solve(m_data[i].x, m_data[i + 1].x, rng, costs_, maximize, cs,
                                std::clamp(static_cast<double>(i) / (5 * e), 0.0, 1.0));
        }
        for (int i = 0, e = m_size; i != e; ++i) {
            m_data[i].hash = bit_array_hash(m_data[i].x);
            m_data[i].value = costs.results(m_data[i].x, cost_constant_);
            m_data[i].remaining_constraints = 0;
            for (int k = 0; k != cs.m; ++k) {
                int sum = 0;
                for (int el = cs.row_ptr[k]; el != cs.row_ptr[k + 1]; ++el)
                    if (m_data[i].x[cs.col[el]])
                        sum += cs.factor[el];
                if (!(cs.cmin[k] <= sum && sum <= cs.cmax[k]))
                    ++m_data[i].remaining_constraints;
            }
        }
        std::iota(m_indices.begin(), m_indices.end(), 0);
        sort();
    }

    void insert_advance(LocalContext& ctx, const BitArray& x, std::size_t hash,
                        int remaining_constraints, long loop) noexcept // :240-272
    {
        const int id_to_delete = m_indices[choose_a_bad_solution(ctx)];
        replace_result(id_to_delete, x, costs.results(x, cost_constant), hash, loop,
                       remaining_constraints);
        sort();
    }
    void insert_solution(LocalContext& ctx, const BitArray& x, std::size_t hash, double value,
                         long loop) noexcept // :274-300
    {
        const int id_to_delete = m_indices[choose_a_bad_solution(ctx)];
        replace_result(id_to_delete, x, value, hash, loop, 0);
        sort();
    }
    bool can_be_inserted(std::size_t hash, int constraints) const noexcept // :302-313
    {
        for (int i = 0; i != m_size; ++i)
            if (m_data[i].remaining_constraints == constraints && m_data[i].hash == hash)
                return false;
        return true;
    }
    bool can_be_inserted(std::size_t hash, double value) const noexcept // :315-326
    {
        for (int i = 0; i != m_size; ++i)
            if (m_data[i].remaining_constraints == 0 && m_data[i].value == value &&
                m_data[i].hash == hash)
                return false;
        return true;
    }
    const RawResult& get_worst() const noexcept // :328-338
    {
        int last = 1;
        const int end = static_cast<int>(m_indices.size());
        while (last != end && m_data[m_indices[last]].is_solution())
            ++last;
        return last == end ? m_data[m_indices[last - 1]] : m_data[m_indices[last]];
    }
    const RawResult& get_best(int i) const noexcept { return m_data[m_indices[i]]; }
    int size() const noexcept { return m_size; }

    void crossover(LocalContext& ctx, BitArray& x, const BitArray& first,
                   const BitArray& second) // :359-371
    {
        for (int i = 0, e = x.block_size(); i != e; ++i) {
            const auto x1 = first.w[i];
            const auto x2 = second.w[i];
            x.w[i] = ((x1 ^ x2) & ctx.crossover_dist(ctx.rng)) ^ x1;
        }
    }
    void crossover(LocalContext& ctx, BitArray& x) // :373-421
    {
        if (ctx.crossover_bastert_insertion(ctx.rng)) {
            const int first = m_indices[choose_a_solution(ctx)];
            std::bernoulli_distribution b(0.5);
            if (b(ctx.rng)) {
                if (b(ctx.rng))
                    crossover(ctx, x, m_data[first].x, m_bastert);
                else
                    m_data[first].x = m_bastert;
            } else {
                init_with_random(m_random, ctx.rng, x.size(), 0.5);
                if (b(ctx.rng))
                    crossover(ctx, x, m_data[first].x, m_random);
                else
                    m_data[first].x = m_random;
            }
        } else {
            const int first = m_indices[choose_a_solution(ctx)];
            int second = m_indices[choose_a_solution(ctx)];
            while (first == second)
                second = m_indices[choose_a_solution(ctx)];
            crossover(ctx, x, m_data[first].x, m_data[second].x);
        }
    }
};

// best_solution_recorder, itm-optimizer-common.hpp:460-601
template<typename Float>
struct Recorder
{
    Storage<Float> m_storage;
    std::vector<double> m_kappa_append;

    Recorder(random_engine& rng, unsigned thread_number, const Cost<Float>& costs,
             double cost_constant, int variables, const Constraints& cs, int population_size,
             bool maximize)
      : m_storage(rng, costs, cost_constant, population_size, variables, cs, maximize)
      , m_kappa_append(std::size_t{ thread_number }, 0.0)
    {}

    void mutation(LocalContext& ctx, BitArray& x) // :494-526
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
        std::bernoulli_distribution dist_var_p(var_p);
        std::bernoulli_distribution dist_value_p(val_p);
        for (int i = 0, e = x.size(); i != e; ++i)
            if (dist_var_p(ctx.rng))
                x.set_as_reference(i, dist_value_p(ctx.rng)); // :523, see BitArray
    }

    double reinit(LocalContext& ctx, double kappa_min, double kappa_max, BitArray& x) // :528-554
    {
        double kappa = kappa_min;
        if (m_kappa_append[ctx.thread_id] < ctx.init_kappa_improve_stop) {
            m_kappa_append[ctx.thread_id] += ctx.init_kappa_improve_increase;
            kappa = kappa_min + (kappa_max - kappa_min) * m_kappa_append[ctx.thread_id];
        } else {
            m_kappa_append[ctx.thread_id] = ctx.init_kappa_improve_start;
            m_storage.crossover(ctx, x);
        }
        mutation(ctx, x);
        return kappa;
    }

    void try_advance(LocalContext& ctx, const BitArray& solution, int remaining_constraints,
                     long loop) // :556-570
    {
        const auto hash = bit_array_hash(solution);
        if (m_storage.can_be_inserted(hash, remaining_constraints))
            m_storage.insert_advance(ctx, solution, hash, remaining_constraints, loop);
    }
    void try_update(LocalContext& ctx, const BitArray& solution, double value, long loop) // :572-585
    {
        const auto hash = bit_array_hash(solution);
        if (m_storage.can_be_inserted(hash, value))
            m_storage.insert_solution(ctx, solution, hash, value, loop);
    }
};

// generate_seed, itm-common.hpp:1457-1484
inline std::vector<random_engine::result_type>
generate_seed(random_engine& rng, unsigned thread_number)
{
    using type = random_engine::result_type;
    std::vector<type> ret(thread_number);
    std::uniform_int_distribution<type> dst(std::numeric_limits<type>::min(),
                                            std::numeric_limits<type>::max());
    ret[0] = dst(rng);
    for (unsigned i = 1; i != thread_number; ++i) {
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

// The schedule (see the header of this file).
class Conductor
{
public:
    enum Gate : int
    {
        RUNNING = 0,
        AT_RESTART,
        AT_SWEEP,
        AT_POST,
        EXITED
    };

    explicit Conductor(int workers)
      : state(static_cast<size_t>(workers), RUNNING)
      , go(static_cast<size_t>(workers), 0)
      , wcv(static_cast<size_t>(workers))
    {}

    // worker side: park at a gate until released; false = the run is over
    bool arrive(int b, Gate g)
    {
        std::unique_lock<std::mutex> lk(mu);
        state[b] = g;
        cv.notify_one(); // (the conductor alone waits on cv)
        wcv[b].wait(lk, [&] { return go[b] != 0 || stop; });
        if (!go[b])
            return false; // stopped while parked
        go[b] = 0;
        state[b] = RUNNING;
        return true;
    }
    void exited(int b)
    {
        std::lock_guard<std::mutex> lk(mu);
        state[b] = EXITED;
        cv.notify_one();
    }

    // conductor side
    void wait_parked(int b)
    {
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [&] { return state[b] != RUNNING && go[b] == 0; });
    }
    Gate where(int b)
    {
        std::lock_guard<std::mutex> lk(mu);
        return static_cast<Gate>(state[b]);
    }
    void release(int b)
    {
        std::lock_guard<std::mutex> lk(mu);
        go[b] = 1;
        wcv[b].notify_one();
    }
    void finish()
    {
        std::lock_guard<std::mutex> lk(mu);
        stop = true;
        for (auto& c : wcv)
            c.notify_one();
    }

    long run(long rounds)
    {
        const int W = static_cast<int>(state.size());
        for (int b = 0; b != W; ++b)
            wait_parked(b);
        long done = 0;
        for (; done != rounds; ++done) {
            for (int b = 0; b != W; ++b) // 1. restarts, in worker order
                if (where(b) == AT_RESTART) {
                    release(b);
                    wait_parked(b);
                }
            for (int b = 0; b != W; ++b) // 2. one sweep each, together
                if (where(b) == AT_SWEEP)
                    release(b);
            for (int b = 0; b != W; ++b)
                wait_parked(b);
            for (int b = 0; b != W; ++b) // 3. bookkeeping, in worker order
                if (where(b) == AT_POST) {
                    release(b);
                    wait_parked(b);
                }
        }
        finish();
        return done;
    }

private:
    std::mutex mu;
    std::condition_variable cv;
    std::vector<int> state;
    std::vector<int> go;
    std::vector<std::condition_variable> wcv; // one per worker: no thundering herd
    bool stop = false;
};

struct OptimizeResult
{
    int status = ST_UNINITIALIZED;
    long loop = 0;
    int remaining_constraints = INT_MAX;
    std::vector<RawResult> population; // best first
    RawResult worst;
    std::vector<unsigned> worker_rng;  // engine state of every worker at the end
    std::vector<long> worker_calls;    // optimize_functor::m_call_number
    std::uint64_t row_updates = 0;
    std::uint64_t contract_violations = 0; // summed over the workers' solvers
    long rounds = 0;
    double sweep_seconds = 0.0; // CPU time inside compute.run / push_and_run, summed over the workers
};

inline double
thread_cpu_seconds() // CPU time of the calling thread (not the time it waited for a core)
{
    timespec ts;
    clock_gettime(CLOCK_THREAD_CPUTIME_ID, &ts);
    return static_cast<double>(ts.tv_sec) + 1e-9 * static_cast<double>(ts.tv_nsec);
}

inline unsigned
engine_state_of(const random_engine& e)
{
    std::ostringstream os;
    os << e;
    return static_cast<unsigned>(std::stoul(os.str()));
}

// optimize_functor::operator(), itm-optimizer-common.hpp:620-752, with the
// conductor's gates where the reference reads `stop_task`.
template<typename Float>
inline void
optimize_worker(Conductor& cd,
                int b,
                LocalContext& lc,
                long& call_number,
                Recorder<Float>& best_recorder,
                const Constraints& cs,
                const Cost<Float>& original_costs,
                double cost_constant,
                int kind,
                bool maximize,
                const Params& p,
                std::uint64_t& row_updates,
                double& sweep_seconds)
{
    const int variables = cs.n;
    BitArray x(variables);
    const auto norm_costs = normalize_costs(original_costs, p.cost_norm, lc.rng);
    const auto kappa_step = static_cast<Float>(p.kappa_step);
    const auto kappa_max = static_cast<Float>(p.kappa_max);
    const auto alpha = static_cast<Float>(p.alpha);
    const auto theta = static_cast<Float>(p.theta);
    const auto delta =
      p.delta < 0 ? compute_delta<Float>(norm_costs, theta) : static_cast<Float>(p.delta);
    const auto pushing_k_factor = static_cast<Float>(p.pushing_k_factor);
    const auto pushing_objective_amplifier = static_cast<Float>(p.pushing_objective_amplifier);
    const long w_limit = static_cast<long>(p.w);

    Solver<Float> slv(lc.rng, cs, norm_costs, kind, maximize, p.sort_mode);
    ComputeOrder compute(p.order, variables, p.sort_mode);
    bool is_a_solution = false;
    (void)is_a_solution;

    auto sweep = [&](Float kappa, bool push) -> int { // one compute.run / push_and_run
        if (!cd.arrive(b, Conductor::AT_SWEEP))
            return -1;
        const double t0 = thread_cpu_seconds();
        const int remaining =
          push ? compute.run(slv, x, lc.rng, kappa, delta, theta, true, pushing_objective_amplifier)
               : compute.run(slv, x, lc.rng, kappa, delta, theta);
        sweep_seconds += thread_cpu_seconds() - t0;
        row_updates = slv.row_updates;
        if (!cd.arrive(b, Conductor::AT_POST))
            return -1;
        return remaining;
    };

    for (;;) { // while (!stop_task.load())
        if (!cd.arrive(b, Conductor::AT_RESTART))
            return;
        ++call_number;
        const auto kappa_start = static_cast<Float>(best_recorder.reinit(lc, p.kappa_min, p.kappa_max, x));
        auto kappa = kappa_start;
        compute.init(slv, x);

        auto best_remaining = INT_MAX;
        is_a_solution = false;

        for (long i = 0; i != p.limit; ++i) {
            const int remaining = sweep(kappa, false);
            if (remaining < 0)
                return;
            if (remaining == 0) {
                best_recorder.try_update(lc, x, original_costs.results(x, cost_constant), i);
                best_remaining = 0;
                is_a_solution = true;
                break;
            } else {
                best_remaining = std::min(remaining, best_remaining);
            }
            if (i > w_limit)
                kappa += kappa_step *
                         std::pow(static_cast<Float>(remaining) / static_cast<Float>(slv.m), alpha);
            if (kappa > kappa_max)
                break;
        }

        if (best_remaining > 0) {
            best_recorder.try_advance(lc, x, best_remaining, p.limit);
            continue;
        }

        for (int push = 0; push < p.pushes_limit; ++push) {
            int remaining = sweep(pushing_k_factor, true); // (kappa = pushing_k_factor alone, :709)
            if (remaining < 0)
                return;
            if (remaining == 0) {
                best_recorder.try_update(lc, x, original_costs.results(x, cost_constant),
                                         -push * p.pushing_iteration_limit - 1);
                is_a_solution = true;
            }
            kappa = kappa_start;
            for (int iter = 0; iter < p.pushing_iteration_limit; ++iter) {
                remaining = sweep(kappa, false);
                if (remaining < 0)
                    return;
                if (remaining == 0) {
                    best_recorder.try_update(lc, x, original_costs.results(x, cost_constant),
                                             -push * p.pushing_iteration_limit - iter - 1);
                    is_a_solution = true;
                    break;
                }
                if (iter > p.w)
                    kappa += kappa_step * std::pow(static_cast<Float>(remaining) /
                                                     static_cast<Float>(slv.m),
                                                   alpha);
                if (kappa > kappa_max)
                    break;
            }
        }
    }
}

// optimize_problem, itm-optimizer-common.hpp:776-908, for `rounds` rounds.
template<typename Float>
inline OptimizeResult
optimize(const Constraints& cs,
         const Objective& obj,
         int kind,
         bool maximize,
         const Params& p,
         int thread,
         long rounds)
{
    OptimizeResult r;
    const std::uint64_t contract_before = g_contract_violations.load();
    random_engine rng(p.seed);
    const int variables = cs.n;
    const Cost<Float> cost(obj, variables);
    const double cost_constant = obj.constant;

    Recorder<Float> best_recorder(rng, static_cast<unsigned>(thread), cost, cost_constant, variables,
                                  cs, p.init_population_size, maximize);
    const auto seeds = generate_seed(rng, static_cast<unsigned>(thread));

    std::vector<std::unique_ptr<LocalContext>> lctx;
    for (int i = 0; i != thread; ++i)
        lctx.emplace_back(std::make_unique<LocalContext>(p, static_cast<unsigned>(i), seeds[i]));
    std::vector<long> calls(static_cast<size_t>(thread), 0);
    std::vector<std::uint64_t> updates(static_cast<size_t>(thread), 0);
    std::vector<double> seconds(static_cast<size_t>(thread), 0.0);

    Conductor cd(thread);
    std::vector<std::thread> pool;
    for (int i = 0; i != thread; ++i)
        pool.emplace_back([&, i] {
            optimize_worker<Float>(cd, i, *lctx[i], calls[i], best_recorder, cs, cost, cost_constant,
                                   kind, maximize, p, updates[i], seconds[i]);
            cd.exited(i);
        });
    r.rounds = cd.run(rounds);
    for (auto& t : pool)
        t.join();

    const auto& first = best_recorder.m_storage.get_best(0);
    r.status = first.is_solution() ? ST_SUCCESS : ST_TIME_LIMIT_REACHED;
    r.loop = first.loop;
    r.remaining_constraints = first.remaining_constraints;
    for (int i = 0; i != best_recorder.m_storage.size(); ++i)
        r.population.push_back(best_recorder.m_storage.get_best(i));
    r.worst = best_recorder.m_storage.get_worst();
    for (int i = 0; i != thread; ++i) {
        r.worker_rng.push_back(engine_state_of(lctx[i]->rng));
        r.worker_calls.push_back(calls[i]);
        r.row_updates += updates[i];
        r.sweep_seconds += seconds[i];
    }
    r.contract_violations = g_contract_violations.load() - contract_before; // (one optimize at a time)
    return r;
}

} // namespace orc

#endif
