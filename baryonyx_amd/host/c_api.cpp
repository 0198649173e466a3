// baryonyx_amd/host/c_api.cpp -- a small C surface over the C++ API so the
// Python test / bench plumbing (ctypes) can drive baryonyx::make_problem,
// solve and optimize without a C++ test runner.
#include "private.hpp"

#include <cstdlib>
#include <cstring>
#include <sstream>

using namespace baryonyx;

extern "C" {

struct bxc_params
{
    double time_limit, theta, delta, kappa_min, kappa_step, kappa_max, alpha;
    double pushing_k_factor, pushing_objective_amplifier, init_policy_random, w;
    long long limit;
    unsigned seed;
    int thread, pushes_limit, pushing_iteration_limit, init_population_size;
    int order, float_type, init_policy, cost_norm, pre_order;
    int device;
    long long sweep_budget;
    int preprocessor; // solver_parameters::preprocessor_options: 0 none, 1 all
    int pad_;
    // the evolutionary layer of optimize() (core:216-229)
    double init_crossover_bastert_insertion, init_crossover_solution_selection_mean,
      init_crossover_solution_selection_stddev, init_mutation_variable_mean,
      init_mutation_variable_stddev, init_mutation_value_mean, init_mutation_value_stddev,
      init_kappa_improve_start, init_kappa_improve_increase, init_kappa_improve_stop;
};

struct bxc_problem_info
{
    int status; // file_format_error_tag
    int variables, equal, greater, less, nnz;
    int maximize, problem_type, n_objective, n_quadratic, n_binary;
    double objective_constant;
};

struct bxc_result
{
    int status; // result_status
    int remaining_constraints, variables, constraints, n_solutions, valid;
    long long loop;
    double value, recomputed_value, duration;
};

// wall-clock phases of the calling thread's last optimize(): host set-up,
// device set-up, rounds (seconds)
// ... and, of the rounds, the host's share outside its wait for the device
// (seconds) with the number of rounds
BX_EXPORT void
bxc_last_optimize_host_rounds(double* seconds, long long* rounds)
{
    const optimize_timing& t = last_optimize_timing();
    *seconds = t.rounds_host;
    *rounds = t.round_count;
}

BX_EXPORT void
bxc_last_optimize_timing(double* out3)
{
    const optimize_timing& t = last_optimize_timing();
    out3[0] = t.host_setup;
    out3[1] = t.device_setup;
    out3[2] = t.rounds;
}

BX_EXPORT void
bxc_default_params(bxc_params* p)
{
    const solver_parameters d;
    std::memset(p, 0, sizeof *p);
    p->time_limit = d.time_limit;
    p->theta = d.theta;
    p->delta = d.delta;
    p->kappa_min = d.kappa_min;
    p->kappa_step = d.kappa_step;
    p->kappa_max = d.kappa_max;
    p->alpha = d.alpha;
    p->pushing_k_factor = d.pushing_k_factor;
    p->pushing_objective_amplifier = d.pushing_objective_amplifier;
    p->init_policy_random = d.init_policy_random;
    p->w = d.w;
    p->limit = d.limit;
    p->seed = d.seed;
    p->thread = d.thread;
    p->pushes_limit = d.pushes_limit;
    p->pushing_iteration_limit = d.pushing_iteration_limit;
    p->init_population_size = d.init_population_size;
    p->order = static_cast<int>(d.order);
    p->float_type = static_cast<int>(d.float_type);
    p->init_policy = static_cast<int>(d.init_policy);
    p->cost_norm = static_cast<int>(d.cost_norm);
    p->pre_order = static_cast<int>(d.pre_order);
    p->preprocessor = static_cast<int>(d.preprocessor);
    p->init_crossover_bastert_insertion = d.init_crossover_bastert_insertion;
    p->init_crossover_solution_selection_mean = d.init_crossover_solution_selection_mean;
    p->init_crossover_solution_selection_stddev = d.init_crossover_solution_selection_stddev;
    p->init_mutation_variable_mean = d.init_mutation_variable_mean;
    p->init_mutation_variable_stddev = d.init_mutation_variable_stddev;
    p->init_mutation_value_mean = d.init_mutation_value_mean;
    p->init_mutation_value_stddev = d.init_mutation_value_stddev;
    p->init_kappa_improve_start = d.init_kappa_improve_start;
    p->init_kappa_improve_increase = d.init_kappa_improve_increase;
    p->init_kappa_improve_stop = d.init_kappa_improve_stop;
}

static context_ptr
make_ctx(const bxc_params* p)
{
    auto ctx = make_context(3);
    solver_parameters q;
    q.time_limit = p->time_limit;
    q.theta = p->theta;
    q.delta = p->delta;
    q.kappa_min = p->kappa_min;
    q.kappa_step = p->kappa_step;
    q.kappa_max = p->kappa_max;
    q.alpha = p->alpha;
    q.pushing_k_factor = p->pushing_k_factor;
    q.pushing_objective_amplifier = p->pushing_objective_amplifier;
    q.init_policy_random = p->init_policy_random;
    q.w = p->w;
    q.limit = static_cast<long>(p->limit);
    q.seed = p->seed;
    q.thread = p->thread;
    q.pushes_limit = p->pushes_limit;
    q.pushing_iteration_limit = p->pushing_iteration_limit;
    q.init_population_size = p->init_population_size;
    q.order = static_cast<solver_parameters::constraint_order>(p->order);
    q.float_type = static_cast<solver_parameters::floating_point_type>(p->float_type);
    q.init_policy = static_cast<solver_parameters::init_policy_type>(p->init_policy);
    q.cost_norm = static_cast<solver_parameters::cost_norm_type>(p->cost_norm);
    q.pre_order = static_cast<solver_parameters::pre_constraint_order>(p->pre_order);
    q.preprocessor = static_cast<solver_parameters::preprocessor_options>(p->preprocessor);
    q.init_crossover_bastert_insertion = p->init_crossover_bastert_insertion;
    q.init_crossover_solution_selection_mean = p->init_crossover_solution_selection_mean;
    q.init_crossover_solution_selection_stddev = p->init_crossover_solution_selection_stddev;
    q.init_mutation_variable_mean = p->init_mutation_variable_mean;
    q.init_mutation_variable_stddev = p->init_mutation_variable_stddev;
    q.init_mutation_value_mean = p->init_mutation_value_mean;
    q.init_mutation_value_stddev = p->init_mutation_value_stddev;
    q.init_kappa_improve_start = p->init_kappa_improve_start;
    q.init_kappa_improve_increase = p->init_kappa_improve_increase;
    q.init_kappa_improve_stop = p->init_kappa_improve_stop;
    context_set_solver_parameters(ctx, q);
    context_set_parameters(ctx, "device", std::to_string(p->device));
    if (const char* dl = std::getenv("BXC_DEVICES")) // e.g. "0,1,2,3": optimize() over several devices
        context_set_parameters(ctx, "devices", dl);
    if (p->sweep_budget > 0)
        context_set_parameters(ctx, "sweep-budget", std::to_string(p->sweep_budget));
    return ctx;
}

// context_set_solver_parameters + context_get_solver_parameters: what the solvers will see of
// `in` after the range checks of lib/src/private.cpp:50-172 (tests).
BX_EXPORT void
bxc_validated_params(const bxc_params* in, bxc_params* out)
{
    auto ctx = make_ctx(in);
    const solver_parameters q = context_get_solver_parameters(ctx);
    *out = *in;
    out->time_limit = q.time_limit;
    out->theta = q.theta;
    out->delta = q.delta;
    out->kappa_min = q.kappa_min;
    out->kappa_step = q.kappa_step;
    out->kappa_max = q.kappa_max;
    out->alpha = q.alpha;
    out->pushing_k_factor = q.pushing_k_factor;
    out->pushing_objective_amplifier = q.pushing_objective_amplifier;
    out->init_policy_random = q.init_policy_random;
    out->w = q.w;
    out->limit = q.limit;
    out->seed = q.seed;
    out->thread = q.thread;
    out->pushes_limit = q.pushes_limit;
    out->pushing_iteration_limit = q.pushing_iteration_limit;
    out->init_population_size = q.init_population_size;
    out->order = static_cast<int>(q.order);
    out->init_policy = static_cast<int>(q.init_policy);
    out->cost_norm = static_cast<int>(q.cost_norm);
}

// make_problem on a file; fills `info`.
BX_EXPORT int
bxc_parse(const char* path, bxc_problem_info* info)
{
    auto ctx = make_context(3);
    const raw_problem pb = make_problem(ctx, path);
    std::memset(info, 0, sizeof *info);
    info->status = static_cast<int>(pb.status);
    if (!pb)
        return 0;
    info->variables = static_cast<int>(pb.vars.values.size());
    info->equal = static_cast<int>(pb.equal_constraints.size());
    info->greater = static_cast<int>(pb.greater_constraints.size());
    info->less = static_cast<int>(pb.less_constraints.size());
    for (const auto* v : { &pb.equal_constraints, &pb.greater_constraints, &pb.less_constraints })
        for (const auto& c : *v)
            info->nnz += static_cast<int>(c.elements.size());
    info->maximize = pb.type == objective_function_type::maximize;
    info->problem_type = static_cast<int>(problem(pb).problem_type);
    info->n_objective = static_cast<int>(pb.objective.elements.size());
    info->n_quadratic = static_cast<int>(pb.objective.qelements.size());
    for (const auto& v : pb.vars.values)
        info->n_binary += v.type == variable_type::binary;
    info->objective_constant = pb.objective.value;
    return 0;
}

// make_problem on a file, written out in a canonical text (tests compare it with an
// independent restatement of the reference's reader, tests/lp_grammar_restatement.py:
// canonical()).  Returns the reader's status; `needed` receives the text's length.
BX_EXPORT int
bxc_dump_problem(const char* path, char* out, int cap, int* needed)
{
    auto ctx = make_context(3);
    const raw_problem pb = make_problem(ctx, path);
    std::string t;
    char line[256];
    if (pb) {
        std::snprintf(line, sizeof line, "type %d\nconstant %.17g", pb.type == objective_function_type::maximize ? 1 : 0,
                      pb.objective.value);
        t += line;
        for (size_t i = 0; i != pb.vars.values.size(); ++i) {
            t += "\nvar ";
            t.append(pb.vars.names[i].data(), pb.vars.names[i].size());
            std::snprintf(line, sizeof line, " %d %d %d", pb.vars.values[i].min, pb.vars.values[i].max,
                          static_cast<int>(pb.vars.values[i].type));
            t += line;
        }
        for (const auto& e : pb.objective.elements) {
            std::snprintf(line, sizeof line, "\nobj %d %.17g", e.variable_index, e.factor);
            t += line;
        }
        for (const auto& e : pb.objective.qelements) {
            std::snprintf(line, sizeof line, "\nqobj %d %d %.17g", e.variable_index_a, e.variable_index_b, e.factor);
            t += line;
        }
        const std::pair<const char*, const std::vector<constraint>*> kinds[] = {
            { "eq", &pb.equal_constraints }, { "ge", &pb.greater_constraints }, { "le", &pb.less_constraints }
        };
        for (const auto& kv : kinds)
            for (const auto& c : *kv.second) {
                std::snprintf(line, sizeof line, "\n%s %d %d [", kv.first, c.id, c.value);
                t += line;
                t.append(c.label.data(), c.label.size());
                t += "]";
                bool first = true;
                for (const auto& e : c.elements) {
                    std::snprintf(line, sizeof line, "%s%d*%d", first ? " " : " ", e.factor, e.variable_index);
                    t += line;
                    first = false;
                }
                if (c.elements.empty())
                    t += " ";
            }
    }
    if (needed)
        *needed = static_cast<int>(t.size());
    if (out && cap > 0) {
        const size_t n = std::min(t.size(), static_cast<size_t>(cap - 1));
        std::memcpy(out, t.data(), n);
        out[n] = '\0';
    }
    return static_cast<int>(pb.status);
}

// make_problem on a file; the variables' bounds and types as the reader left
// them (min, max, variable_type), `cap` entries at most.  Returns the status.
BX_EXPORT int
bxc_variables(const char* path, int* vmin, int* vmax, int* vtype, int cap)
{
    auto ctx = make_context(3);
    const raw_problem pb = make_problem(ctx, path);
    if (!pb)
        return static_cast<int>(pb.status);
    const int n = std::min(cap, static_cast<int>(pb.vars.values.size()));
    for (int i = 0; i != n; ++i) {
        vmin[i] = pb.vars.values[static_cast<size_t>(i)].min;
        vmax[i] = pb.vars.values[static_cast<size_t>(i)].max;
        vtype[i] = static_cast<int>(pb.vars.values[static_cast<size_t>(i)].type);
    }
    return 0;
}

// The merged constraints (CSR) and objective of a file, for feeding another
// solver with exactly what the device gets.  Two calls: buffers null -> sizes.
BX_EXPORT int
bxc_merged(const char* path,
           int pre_order,
           int preprocessor,
           int* m,
           int* n,
           int* nnz,
           int* n_obj,
           int* row_ptr,
           int* col,
           int* factor,
           int* cmin,
           int* cmax,
           int* obj_var,
           double* obj_factor,
           double* obj_constant,
           int* maximize,
           int* problem_type,
           int* n_quad,
           int* quad_a,
           int* quad_b,
           double* quad_factor)
{
    auto ctx = make_context(3);
    solver_parameters q;
    q.pre_order = static_cast<solver_parameters::pre_constraint_order>(pre_order);
    context_set_solver_parameters(ctx, q);
    const raw_problem raw = make_problem(ctx, path);
    if (!raw)
        return -1;
    const problem pb = preprocessor ? preprocess(*ctx, raw) : problem(raw);
    std::vector<merged_constraint> csts;
    try {
        csts = make_merged_constraints(*ctx, pb);
    } catch (...) {
        return -2;
    }
    int count = 0;
    for (const auto& c : csts)
        count += static_cast<int>(c.elements.size());
    *m = static_cast<int>(csts.size());
    *n = static_cast<int>(pb.vars.values.size());
    *nnz = count;
    *n_obj = static_cast<int>(pb.objective.elements.size());
    *obj_constant = pb.objective.value;
    *maximize = pb.type == objective_function_type::maximize;
    *problem_type = static_cast<int>(pb.problem_type);
    if (n_quad)
        *n_quad = static_cast<int>(pb.objective.qelements.size());
    if (!row_ptr)
        return 0;
    if (quad_a)
        for (size_t i = 0; i != pb.objective.qelements.size(); ++i) {
            quad_a[i] = pb.objective.qelements[i].variable_index_a;
            quad_b[i] = pb.objective.qelements[i].variable_index_b;
            quad_factor[i] = pb.objective.qelements[i].factor;
        }
    int e = 0;
    row_ptr[0] = 0;
    for (int k = 0; k != *m; ++k) {
        for (const auto& el : csts[k].elements) {
            col[e] = el.variable_index;
            factor[e] = el.factor;
            ++e;
        }
        row_ptr[k + 1] = e;
        cmin[k] = csts[k].min;
        cmax[k] = csts[k].max;
    }
    for (int i = 0; i != *n_obj; ++i) {
        obj_var[i] = pb.objective.elements[i].variable_index;
        obj_factor[i] = pb.objective.elements[i].factor;
    }
    return 0;
}

// baryonyx::optimize on a file with the recorder's final state (tests): the
// population best first ([population] entries, x as [population][nwords] 64-bit
// blocks of the PREPROCESSED variables), every worker's engine and restart count.
BX_EXPORT int
bxc_optimize_trace(const char* path, const bxc_params* p, bxc_result* out, int population, int nwords,
                   int workers, int* pop_remaining, double* pop_value, unsigned long long* pop_hash,
                   unsigned long long* pop_x, unsigned* worker_rng, long long* worker_calls,
                   long long* rounds, char* err, int errlen)
{
    std::memset(out, 0, sizeof *out);
    try {
        auto ctx = make_ctx(p);
        optimize_trace t;
        ctx->trace = &t;
        const raw_problem pb = make_problem(ctx, path);
        if (!pb) {
            std::snprintf(err, static_cast<size_t>(errlen), "parse error %d", static_cast<int>(pb.status));
            return -1;
        }
        const result r = optimize(ctx, pb);
        out->status = static_cast<int>(r.status);
        out->remaining_constraints = r.remaining_constraints;
        out->variables = r.variables;
        out->constraints = r.constraints;
        out->n_solutions = static_cast<int>(r.solutions.size());
        out->loop = r.loop;
        out->duration = r.duration;
        if (!r.solutions.empty())
            out->value = r.solutions.front().value;
        out->valid = is_valid_solution(pb, r) ? 1 : 0;
        if (r && !r.solutions.empty())
            out->recomputed_value = compute_solution(pb, r);
        if (static_cast<int>(t.pop_value.size()) != population ||
            (population > 0 && static_cast<int>(t.pop_x[0].size()) != nwords)) {
            std::snprintf(err, static_cast<size_t>(errlen), "trace shape: population %zu, words %zu",
                          t.pop_value.size(), t.pop_x.empty() ? size_t{ 0 } : t.pop_x[0].size());
            return -4;
        }
        if (static_cast<int>(t.worker_rng.size()) != workers) { // (thread <= 0 means 64 workers)
            std::snprintf(err, static_cast<size_t>(errlen), "trace shape: %zu workers, room for %d",
                          t.worker_rng.size(), workers);
            return -4;
        }
        for (int i = 0; i != population; ++i) {
            pop_remaining[i] = t.pop_remaining[i];
            pop_value[i] = t.pop_value[i];
            pop_hash[i] = t.pop_hash[i];
            std::copy(t.pop_x[i].begin(), t.pop_x[i].end(), pop_x + static_cast<size_t>(i) * nwords);
        }
        for (size_t b = 0; b != t.worker_rng.size(); ++b) {
            worker_rng[b] = t.worker_rng[b];
            worker_calls[b] = t.worker_calls[b];
        }
        *rounds = t.rounds;
        return 0;
    } catch (const solver_failure& e) {
        std::snprintf(err, static_cast<size_t>(errlen), "solver_failure %d", static_cast<int>(e.failure()));
        return -2;
    } catch (const std::exception& e) {
        std::snprintf(err, static_cast<size_t>(errlen), "%s", e.what());
        return -3;
    }
}

// baryonyx::solve (mode 0) or baryonyx::optimize (mode 1) on a file.
BX_EXPORT int
bxc_run(const char* path, int mode, const bxc_params* p, bxc_result* out, signed char* x_out, char* err, int errlen)
{
    std::memset(out, 0, sizeof *out);
    try {
        auto ctx = make_ctx(p);
        const raw_problem pb = make_problem(ctx, path);
        if (!pb) {
            std::snprintf(err, static_cast<size_t>(errlen), "parse error %d", static_cast<int>(pb.status));
            return -1;
        }
        const result r = mode == 0 ? solve(ctx, pb) : optimize(ctx, pb);
        out->status = static_cast<int>(r.status);
        out->remaining_constraints = r.remaining_constraints;
        out->variables = r.variables;
        out->constraints = r.constraints;
        out->n_solutions = static_cast<int>(r.solutions.size());
        out->loop = r.loop;
        out->duration = r.duration;
        if (!r.solutions.empty()) {
            out->value = r.solutions.front().value;
            if (x_out)
                for (size_t i = 0; i != r.solutions.front().variables.size(); ++i)
                    x_out[i] = r.solutions.front().variables[i];
        }
        out->valid = is_valid_solution(pb, r) ? 1 : 0;
        if (r && !r.solutions.empty())
            out->recomputed_value = compute_solution(pb, r);
        return 0;
    } catch (const solver_failure& e) {
        std::snprintf(err, static_cast<size_t>(errlen), "solver_failure %d", static_cast<int>(e.failure()));
        return -2;
    } catch (const std::exception& e) {
        std::snprintf(err, static_cast<size_t>(errlen), "%s", e.what());
        return -3;
    }
}

// baryonyx::preprocess / affect / split on a file (lib/test/preprocess.cpp):
// sizes of the reduced problem and, optionally, its objective.
struct bxc_pp_info
{
    int variables, affected, equal, greater, less, n_objective, n_quadratic, problem_type;
    double objective_constant;
};

static void
pp_fill(const problem& pb, bxc_pp_info* o)
{
    o->variables = static_cast<int>(pb.vars.names.size());
    o->affected = static_cast<int>(pb.affected_vars.names.size());
    o->equal = static_cast<int>(pb.equal_constraints.size());
    o->greater = static_cast<int>(pb.greater_constraints.size());
    o->less = static_cast<int>(pb.less_constraints.size());
    o->n_objective = static_cast<int>(pb.objective.elements.size());
    o->n_quadratic = static_cast<int>(pb.objective.qelements.size());
    o->problem_type = static_cast<int>(pb.problem_type);
    o->objective_constant = pb.objective.value;
}

// `fix_zero` (may be null): name of a variable whose bounds are set to 0 = 0
// before preprocessing.  `affect_var` >= 0: then affect() that variable of the
// reduced problem to `affect_value` -> out_affect; `split_var` >= 0: split()
// -> out_split1 (true) and out_split0 (false).
BX_EXPORT int
bxc_preprocess(const char* path,
               const char* fix_zero,
               bxc_pp_info* out,
               int affect_var,
               int affect_value,
               bxc_pp_info* out_affect,
               int split_var,
               bxc_pp_info* out_split1,
               bxc_pp_info* out_split0,
               double* lin_factor,
               int* lin_var,
               double* quad_factor,
               int* quad_a,
               int* quad_b)
{
    try {
        auto ctx = make_context(3);
        raw_problem raw = make_problem(ctx, path);
        if (!raw)
            return -1;
        if (fix_zero)
            for (size_t i = 0; i != raw.vars.names.size(); ++i)
                if (raw.vars.names[i] == fix_zero) {
                    raw.vars.values[i].min = 0;
                    raw.vars.values[i].max = 0;
                }
        const problem pp = preprocess(*ctx, raw);
        pp_fill(pp, out);
        if (lin_factor)
            for (size_t i = 0; i != pp.objective.elements.size(); ++i) {
                lin_factor[i] = pp.objective.elements[i].factor;
                lin_var[i] = pp.objective.elements[i].variable_index;
            }
        if (quad_factor)
            for (size_t i = 0; i != pp.objective.qelements.size(); ++i) {
                quad_factor[i] = pp.objective.qelements[i].factor;
                quad_a[i] = pp.objective.qelements[i].variable_index_a;
                quad_b[i] = pp.objective.qelements[i].variable_index_b;
            }
        if (affect_var >= 0 && out_affect)
            pp_fill(affect(*ctx, pp, affect_var, affect_value != 0), out_affect);
        if (split_var >= 0 && out_split1 && out_split0) {
            const auto both = split(*ctx, pp, split_var);
            pp_fill(std::get<0>(both), out_split1);
            pp_fill(std::get<1>(both), out_split0);
        }
        return 0;
    } catch (...) {
        return -2;
    }
}

// the host's bit_array through a list of operations (itm_gpu.cpp: bit_array_script)
BX_EXPORT int
bxc_bitarray_script(int size, const int* ops, int nops, unsigned long long* words, int* gets)
{
    static_assert(sizeof(unsigned long long) == sizeof(std::uint64_t), "64-bit blocks");
    return itm::bit_array_script(size, ops, nops, reinterpret_cast<std::uint64_t*>(words), gets);
}

} // extern "C"
