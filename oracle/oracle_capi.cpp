// oracle/oracle_capi.cpp -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
// extern "C" surface over oracle/itm_oracle.hpp (see oracle_capi.h).
#include "oracle_capi.h"
#include "itm_oracle.hpp"
#include "itm_oracle_opt.hpp"

#include <sstream>
#include <string>

namespace {

thread_local std::string g_error;

void
fill(orc::Constraints& cs, orc::Objective& obj, const orc_problem* pb)
{
    cs.m = pb->m;
    cs.n = pb->n;
    cs.row_ptr.assign(pb->row_ptr, pb->row_ptr + pb->m + 1);
    cs.col.assign(pb->col, pb->col + pb->nnz);
    if (pb->factor)
        cs.factor.assign(pb->factor, pb->factor + pb->nnz);
    else
        cs.factor.assign(static_cast<size_t>(pb->nnz), 1);
    cs.cmin.assign(pb->cmin, pb->cmin + pb->m);
    cs.cmax.assign(pb->cmax, pb->cmax + pb->m);
    if (pb->n_obj) {
        obj.var.assign(pb->obj_var, pb->obj_var + pb->n_obj);
        obj.factor.assign(pb->obj_factor, pb->obj_factor + pb->n_obj);
    }
    if (pb->n_qobj) {
        obj.qa.assign(pb->q_a, pb->q_a + pb->n_qobj);
        obj.qb.assign(pb->q_b, pb->q_b + pb->n_qobj);
        obj.qfactor.assign(pb->q_factor, pb->q_factor + pb->n_qobj);
    }
    obj.constant = pb->obj_constant;
}

orc::Params
to_params(const orc_params* p)
{
    orc::Params q;
    q.theta = p->theta;
    q.delta = p->delta;
    q.kappa_min = p->kappa_min;
    q.kappa_step = p->kappa_step;
    q.kappa_max = p->kappa_max;
    q.alpha = p->alpha;
    q.pushing_k_factor = p->pushing_k_factor;
    q.pushing_objective_amplifier = p->pushing_objective_amplifier;
    q.init_policy_random = p->init_policy_random;
    q.init_crossover_bastert_insertion = p->init_crossover_bastert_insertion;
    q.init_crossover_solution_selection_mean =
      p->init_crossover_solution_selection_mean;
    q.init_crossover_solution_selection_stddev =
      p->init_crossover_solution_selection_stddev;
    q.init_mutation_variable_mean = p->init_mutation_variable_mean;
    q.init_mutation_variable_stddev = p->init_mutation_variable_stddev;
    q.init_mutation_value_mean = p->init_mutation_value_mean;
    q.init_mutation_value_stddev = p->init_mutation_value_stddev;
    q.init_kappa_improve_start = p->init_kappa_improve_start;
    q.init_kappa_improve_increase = p->init_kappa_improve_increase;
    q.init_kappa_improve_stop = p->init_kappa_improve_stop;
    q.w = p->w;
    q.limit = static_cast<long>(p->limit);
    q.seed = p->seed;
    q.pushes_limit = p->pushes_limit;
    q.pushing_iteration_limit = p->pushing_iteration_limit;
    q.init_population_size = p->init_population_size;
    q.order = p->order;
    q.init_policy = p->init_policy;
    q.cost_norm = p->cost_norm;
    q.sort_mode = p->sort_mode;
    return q;
}

struct HandleBase
{
    virtual ~HandleBase() = default;
    virtual int nwords() = 0;
    virtual int rmax() = 0;
    virtual double compute_delta(double theta) = 0;
    virtual void get_cost(double* out) = 0;
    virtual void reset() = 0;
    virtual int get_qcost(std::int32_t* qptr, std::int32_t* qid, double* qfactor) = 0;
    virtual void get_x(std::uint64_t* w) = 0;
    virtual void set_x(const std::uint64_t* w) = 0;
    virtual void init_x(int policy, double random) = 0;
    virtual void get_pi(double* o) = 0;
    virtual void set_pi(const double* o) = 0;
    virtual void get_P(double* o) = 0;
    virtual void set_P(const double* o) = 0;
    virtual std::uint32_t get_rng() = 0;
    virtual void set_rng(std::uint32_t s) = 0;
    virtual std::uint64_t row_updates() = 0;
    virtual std::uint64_t tie_rows() = 0;
    virtual std::uint64_t contract_violations() = 0;
    virtual void set_contract_checks(bool on) = 0;
    virtual double objective() = 0;
    virtual void order_init(int order) = 0;
    virtual int violated(std::int32_t* out) = 0;
    virtual int sweep(double, double, double, int, double) = 0;
    virtual int update_rows(const std::int32_t*,
                            int,
                            double,
                            double,
                            double,
                            int,
                            double) = 0;
};

template<typename Float>
struct Handle : HandleBase
{
    orc::Constraints cs;
    orc::Objective obj;
    orc::random_engine rng;
    bool maximize;
    int sort_mode;
    orc::Cost<Float> original;
    orc::Cost<Float> norm;
    std::unique_ptr<orc::Solver<Float>> slv;
    std::unique_ptr<orc::ComputeOrder> compute;
    orc::BitArray x;

    Handle(const orc_problem* pb,
           int kind,
           bool maximize_,
           int sort_mode_,
           int cost_norm,
           std::uint32_t seed)
      : rng(seed)
      , maximize(maximize_)
      , sort_mode(sort_mode_)
    {
        fill(cs, obj, pb);
        original = orc::Cost<Float>(obj, cs.n);
        norm = orc::normalize_costs(original, cost_norm, rng);
        slv = std::make_unique<orc::Solver<Float>>(
          rng, cs, norm, kind, maximize, sort_mode);
        x = orc::BitArray(cs.n);
    }

    int nwords() override { return x.block_size(); }
    int rmax() override { return static_cast<int>(slv->R.size()) - 1; }
    double compute_delta(double theta) override
    {
        return static_cast<double>(
          orc::compute_delta<Float>(norm, static_cast<Float>(theta)));
    }
    void reset() override { slv->reset(); }
    void get_cost(double* out) override
    {
        for (int i = 0; i != cs.n; ++i)
            out[i] = static_cast<double>(norm.linear[i]);
    }
    // the normalised quadratic part: adjacency of every variable
    int get_qcost(std::int32_t* qptr, std::int32_t* qid, double* qfactor) override
    {
        const int count = norm.qsize();
        if (qptr && norm.quadratic) {
            for (int i = 0; i <= cs.n; ++i)
                qptr[i] = norm.qindices[i];
            for (int i = 0; i != count; ++i) {
                qid[i] = norm.qid[i];
                qfactor[i] = static_cast<double>(norm.qfactor[i]);
            }
        }
        return count;
    }
    void get_x(std::uint64_t* w) override
    {
        std::copy(x.w.begin(), x.w.end(), w);
    }
    void set_x(const std::uint64_t* w) override
    {
        std::copy(w, w + x.w.size(), x.w.begin());
    }
    void init_x(int policy, double random) override
    {
        // itm-solver-common.hpp:99-123
        std::bernoulli_distribution choose_mutation(random);
        orc::BitArray empty_x;
        switch (policy) {
        case 1:
            orc::init_with_pre_solve(
              x, empty_x, rng, original, maximize, cs, 1.0);
            break;
        case 2:
            orc::init_with_pre_solve(
              empty_x, x, rng, original, maximize, cs, 1.0);
            break;
        default:
            orc::init_with_bastert(x, original, maximize, cs.n, 0);
            break;
        }
        for (int i = 0, e = x.size(); i != e; ++i)
            if (choose_mutation(rng))
                x.invert(i);
    }
    void get_pi(double* o) override
    {
        for (int i = 0; i != cs.m; ++i)
            o[i] = static_cast<double>(slv->pi[i]);
    }
    void set_pi(const double* o) override
    {
        for (int i = 0; i != cs.m; ++i)
            slv->pi[i] = static_cast<Float>(o[i]);
    }
    void get_P(double* o) override
    {
        for (int i = 0, e = cs.nnz(); i != e; ++i)
            o[i] = static_cast<double>(slv->P[i]);
    }
    void set_P(const double* o) override
    {
        for (int i = 0, e = cs.nnz(); i != e; ++i)
            slv->P[i] = static_cast<Float>(o[i]);
    }
    std::uint32_t get_rng() override
    {
        std::ostringstream os;
        os << rng;
        return static_cast<std::uint32_t>(std::stoul(os.str()));
    }
    void set_rng(std::uint32_t s) override
    {
        std::istringstream is(std::to_string(s));
        is >> rng;
    }
    std::uint64_t row_updates() override { return slv->row_updates; }
    std::uint64_t tie_rows() override { return slv->tie_rows; }
    std::uint64_t contract_violations() override { return slv->contract_violations; }
    void set_contract_checks(bool on) override { slv->check_contracts = on; }
    double objective() override { return original.results(x, obj.constant); }
    void order_init(int order) override
    {
        compute = std::make_unique<orc::ComputeOrder>(order, cs.n, sort_mode);
        compute->init(*slv, x);
    }
    int violated(std::int32_t* out) override
    {
        if (!compute)
            return -1;
        if (compute->order == orc::ORDER_INFEASIBILITY_DECR ||
            compute->order == orc::ORDER_INFEASIBILITY_INCR) {
            if (out)
                for (size_t i = 0; i != compute->m_order.size(); ++i)
                    out[i] = compute->m_order[i].first;
            return static_cast<int>(compute->m_order.size());
        }
        if (out)
            std::copy(compute->R.begin(), compute->R.end(), out);
        return static_cast<int>(compute->R.size());
    }
    int sweep(double kappa,
              double delta,
              double theta,
              int is_push,
              double obj_amp) override
    {
        return compute->run(*slv,
                            x,
                            rng,
                            static_cast<Float>(kappa),
                            static_cast<Float>(delta),
                            static_cast<Float>(theta),
                            is_push != 0,
                            static_cast<Float>(obj_amp));
    }
    int update_rows(const std::int32_t* rows,
                    int nrows,
                    double kappa,
                    double delta,
                    double theta,
                    int is_push,
                    double obj_amp) override
    {
        return slv->run_rows(
                 x,
                 rows,
                 rows + nrows,
                 [](const std::int32_t* it) { return static_cast<int>(*it); },
                 static_cast<Float>(kappa),
                 static_cast<Float>(delta),
                 static_cast<Float>(theta),
                 is_push != 0,
                 static_cast<Float>(obj_amp))
                 ? 1
                 : 0;
    }
};

inline HandleBase*
H(void* h)
{
    return static_cast<HandleBase*>(h);
}

} // namespace

extern "C" {

void
orc_default_params(orc_params* p)
{
    const orc::Params d;
    p->theta = d.theta;
    p->delta = d.delta;
    p->kappa_min = d.kappa_min;
    p->kappa_step = d.kappa_step;
    p->kappa_max = d.kappa_max;
    p->alpha = d.alpha;
    p->pushing_k_factor = d.pushing_k_factor;
    p->pushing_objective_amplifier = d.pushing_objective_amplifier;
    p->init_policy_random = d.init_policy_random;
    p->init_crossover_bastert_insertion = d.init_crossover_bastert_insertion;
    p->init_crossover_solution_selection_mean =
      d.init_crossover_solution_selection_mean;
    p->init_crossover_solution_selection_stddev =
      d.init_crossover_solution_selection_stddev;
    p->init_mutation_variable_mean = d.init_mutation_variable_mean;
    p->init_mutation_variable_stddev = d.init_mutation_variable_stddev;
    p->init_mutation_value_mean = d.init_mutation_value_mean;
    p->init_mutation_value_stddev = d.init_mutation_value_stddev;
    p->init_kappa_improve_start = d.init_kappa_improve_start;
    p->init_kappa_improve_increase = d.init_kappa_improve_increase;
    p->init_kappa_improve_stop = d.init_kappa_improve_stop;
    p->w = d.w;
    p->limit = d.limit;
    p->seed = d.seed;
    p->pushes_limit = d.pushes_limit;
    p->pushing_iteration_limit = d.pushing_iteration_limit;
    p->init_population_size = d.init_population_size;
    p->order = d.order;
    p->init_policy = d.init_policy;
    p->cost_norm = d.cost_norm;
    p->sort_mode = d.sort_mode;
}

const char*
orc_last_error(void)
{
    return g_error.c_str();
}

void*
orc_solver_create(const orc_problem* pb,
                  int kind,
                  int float_type,
                  int maximize,
                  int sort_mode,
                  int cost_norm,
                  uint32_t seed)
{
    try {
        if (float_type == 0)
            return new Handle<float>(
              pb, kind, maximize != 0, sort_mode, cost_norm, seed);
        if (float_type == 1)
            return new Handle<double>(
              pb, kind, maximize != 0, sort_mode, cost_norm, seed);
        g_error = "float_type must be 0 (float) or 1 (double)";
    } catch (const std::exception& e) {
        g_error = e.what();
    }
    return nullptr;
}

void
orc_solver_destroy(void* h)
{
    delete H(h);
}

int
orc_solver_nwords(void* h)
{
    return H(h)->nwords();
}
int
orc_solver_rmax(void* h)
{
    return H(h)->rmax();
}
double
orc_solver_compute_delta(void* h, double theta)
{
    return H(h)->compute_delta(theta);
}
void
orc_solver_get_cost(void* h, double* out)
{
    H(h)->get_cost(out);
}
void
orc_solver_reset(void* h)
{
    H(h)->reset();
}
int
orc_solver_get_qcost(void* h, int32_t* qptr, int32_t* qid, double* qfactor)
{
    return H(h)->get_qcost(qptr, qid, qfactor);
}
void
orc_solver_get_x(void* h, uint64_t* w)
{
    H(h)->get_x(w);
}
void
orc_solver_set_x(void* h, const uint64_t* w)
{
    H(h)->set_x(w);
}
void
orc_solver_init_x(void* h, int init_policy, double init_policy_random)
{
    H(h)->init_x(init_policy, init_policy_random);
}
void
orc_solver_get_pi(void* h, double* o)
{
    H(h)->get_pi(o);
}
void
orc_solver_set_pi(void* h, const double* o)
{
    H(h)->set_pi(o);
}
void
orc_solver_get_P(void* h, double* o)
{
    H(h)->get_P(o);
}
void
orc_solver_set_P(void* h, const double* o)
{
    H(h)->set_P(o);
}
uint32_t
orc_solver_get_rng(void* h)
{
    return H(h)->get_rng();
}
void
orc_solver_set_rng(void* h, uint32_t s)
{
    H(h)->set_rng(s);
}
uint64_t
orc_solver_row_updates(void* h)
{
    return H(h)->row_updates();
}
uint64_t
orc_solver_tie_rows(void* h)
{
    return H(h)->tie_rows();
}
uint64_t
orc_solver_contract_violations(void* h)
{
    return H(h)->contract_violations();
}
uint64_t
orc_contract_violations_total(void)
{
    return orc::g_contract_violations.load();
}
void
orc_solver_set_contract_checks(void* h, int on)
{
    H(h)->set_contract_checks(on != 0);
}
double
orc_solver_objective(void* h)
{
    return H(h)->objective();
}
void
orc_solver_order_init(void* h, int order)
{
    H(h)->order_init(order);
}
int
orc_solver_violated(void* h, int32_t* out)
{
    return H(h)->violated(out);
}
int
orc_solver_sweep(void* h,
                 double kappa,
                 double delta,
                 double theta,
                 int is_push,
                 double obj_amp)
{
    try {
        return H(h)->sweep(kappa, delta, theta, is_push, obj_amp);
    } catch (const std::exception& e) {
        g_error = e.what();
        return -1;
    }
}
int
orc_solver_update_rows(void* h,
                       const int32_t* rows,
                       int nrows,
                       double kappa,
                       double delta,
                       double theta,
                       int is_push,
                       double obj_amp)
{
    try {
        return H(h)->update_rows(
          rows, nrows, kappa, delta, theta, is_push, obj_amp);
    } catch (const std::exception& e) {
        g_error = e.what();
        return -1;
    }
}

int
orc_solve(const orc_problem* pb,
          int kind,
          int float_type,
          int maximize,
          const orc_params* params,
          orc_result* out,
          uint64_t* out_x_words)
{
    try {
        orc::Constraints cs;
        orc::Objective obj;
        fill(cs, obj, pb);
        const orc::Params p = to_params(params);
        if (cs.m == 0 || cs.n == 0) { // solve_problem's guard, itm-solver-common.hpp:269, :293-297
            out->status = orc::ST_SUCCESS;
            out->remaining_constraints = 0;
            out->has_solution_vector = 0;
            out->contract_violations = 0;
            out->value = obj.constant;
            out->loop = 0;
            out->row_updates = 0;
            out->sweeps = 0;
            return 0;
        }
        // init_random_generator_seed, itm-common.hpp:1442-1452 (seed > 0)
        orc::random_engine rng(p.seed);
        orc::SolveResult r =
          float_type == 0
            ? orc::solve<float>(cs, obj, kind, maximize != 0, p, rng)
            : orc::solve<double>(cs, obj, kind, maximize != 0, p, rng);
        out->status = r.status;
        out->remaining_constraints = r.best.remaining_constraints;
        out->has_solution_vector = r.has_solution_vector ? 1 : 0;
        out->value = r.best.value;
        out->loop = r.best.loop;
        out->row_updates = r.row_updates;
        out->sweeps = r.sweeps;
        out->contract_violations = static_cast<int32_t>(std::min<std::uint64_t>(r.contract_violations, INT32_MAX));
        if (out_x_words && r.has_solution_vector)
            std::copy(r.best.x.w.begin(), r.best.x.w.end(), out_x_words);
        return 0;
    } catch (const std::exception& e) {
        g_error = e.what();
        return -1;
    }
}

int
orc_optimize(const orc_problem* pb,
             int kind,
             int float_type,
             int maximize,
             const orc_params* params,
             int thread,
             int64_t rounds,
             orc_result* out,
             int32_t* pop_remaining,
             double* pop_value,
             uint64_t* pop_hash,
             uint64_t* pop_x,
             uint32_t* worker_rng,
             int64_t* worker_calls,
             double* sweep_seconds)
{
    try {
        orc::Constraints cs;
        orc::Objective obj;
        fill(cs, obj, pb);
        const orc::Params p = to_params(params);
        if (thread < 1 || rounds < 0 || p.seed == 0)
            throw std::invalid_argument("orc_optimize: thread >= 1, rounds >= 0, seed > 0");
        if (cs.m == 0 || cs.n == 0) { // optimize_problem's guard, itm-optimizer-common.hpp:785-792
            out->status = orc::ST_SUCCESS;
            out->remaining_constraints = 0;
            out->has_solution_vector = 0;
            out->contract_violations = 0;
            out->value = obj.constant;
            out->loop = 0;
            out->row_updates = 0;
            out->sweeps = 0;
            if (sweep_seconds)
                *sweep_seconds = 0.0;
            return 0; // (the population arrays are left as they came)
        }
        const orc::OptimizeResult r =
          float_type == 0
            ? orc::optimize<float>(cs, obj, kind, maximize != 0, p, thread, static_cast<long>(rounds))
            : orc::optimize<double>(cs, obj, kind, maximize != 0, p, thread, static_cast<long>(rounds));
        out->status = r.status;
        out->remaining_constraints = r.remaining_constraints;
        out->has_solution_vector = 1;
        out->value = r.population.front().value;
        out->loop = r.loop;
        out->row_updates = r.row_updates;
        out->contract_violations = static_cast<int32_t>(std::min<std::uint64_t>(r.contract_violations, INT32_MAX));
        out->sweeps = static_cast<uint64_t>(r.rounds);
        if (sweep_seconds)
            *sweep_seconds = r.sweep_seconds;
        const size_t nwords = static_cast<size_t>(cs.n / 64 + 1);
        for (size_t i = 0; i != r.population.size(); ++i) {
            if (pop_remaining)
                pop_remaining[i] = r.population[i].remaining_constraints;
            if (pop_value)
                pop_value[i] = r.population[i].value;
            if (pop_hash)
                pop_hash[i] = r.population[i].hash;
            if (pop_x)
                std::copy(r.population[i].x.w.begin(), r.population[i].x.w.end(), pop_x + i * nwords);
        }
        for (int i = 0; i != thread; ++i) {
            if (worker_rng)
                worker_rng[i] = r.worker_rng[static_cast<size_t>(i)];
            if (worker_calls)
                worker_calls[i] = r.worker_calls[static_cast<size_t>(i)];
        }
        return 0;
    } catch (const std::exception& e) {
        g_error = e.what();
        return -1;
    }
}

int
orc_bb_solve(int maximize,
             int n,
             double* values,
             int32_t* ids,
             int32_t* factors,
             int bk_min,
             int bk_max)
{
    try {
        std::vector<orc::RC<double>> R(static_cast<size_t>(n) + 1);
        for (int i = 0; i != n; ++i)
            R[i] = { values[i], ids[i], factors[i] };
        orc::BranchAndBound<double> bb;
        const int ret = bb.solve(maximize != 0, R.data(), n, bk_min, bk_max);
        for (int i = 0; i != n; ++i) {
            values[i] = R[i].value;
            ids[i] = R[i].id;
            factors[i] = R[i].f;
        }
        return ret;
    } catch (const std::exception& e) {
        g_error = e.what();
        return -1000;
    }
}

int
orc_exhaustive_solve(int maximize,
                     int n,
                     const int32_t* row_factors,
                     int bk_min,
                     int bk_max,
                     double* values,
                     int32_t* ids,
                     int32_t* factors)
{
    try {
        std::vector<orc::RC<double>> R(static_cast<size_t>(n) + 1);
        for (int i = 0; i != n; ++i)
            R[i] = { values[i], ids[i], factors[i] };
        orc::Exhaustive<double> ex;
        std::vector<int> f(row_factors, row_factors + n);
        ex.build(0, f.data(), n, bk_min, bk_max);
        const int ret = ex.solve(maximize != 0, 0, R.data(), n);
        for (int i = 0; i != n; ++i) {
            values[i] = R[i].value;
            ids[i] = R[i].id;
            factors[i] = R[i].f;
        }
        return ret;
    } catch (const std::exception& e) {
        g_error = e.what();
        return -1000;
    }
}

void
orc_bitarray_set_value(uint64_t* words, int index, int val)
{
    // `words` holds index / 64 + 1 blocks (oracle_capi.h); BitArray(index + 1)
    // would hold one more when (index + 1) % 64 == 0
    orc::BitArray a(index);
    std::copy(words, words + a.block_size(), a.w.begin());
    a.set_as_reference(index, val != 0);
    std::copy(a.w.begin(), a.w.end(), words);
}

int
orc_bitarray_script(int size, const int* ops, int nops, uint64_t* words, int* gets)
{
    orc::BitArray a(size), other(size);
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

} // extern "C"
