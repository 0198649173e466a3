// baryonyx_amd/host/api.cpp -- the exported functions of
// include/baryonyx/core (lib/src/lpcore.cpp:46-132,
// lib/src/private.cpp:39-180).
#include "private.hpp"

#include <algorithm>
#include <cmath>
#include <fstream>

namespace baryonyx {

optimize_timing&
last_optimize_timing()
{
    static thread_local optimize_timing t;
    return t;
}

static void
context_deleter(context* ctx)
{
    delete ctx;
}

context_ptr
make_context(int verbose_level)
{
    return context_ptr(new context(verbose_level), &context_deleter);
}

void
context_set_parameters(const context_ptr& ctx, const std::string& name, std::string value)
{
    if (!ctx)
        return;
    if (name == "method")
        ctx->method = std::move(value);
    else if (name == "device") // extension: HIP device ordinal of the batch
        ctx->device = std::stoi(value);
    else if (name == "devices") { // extension: optimize() over several HIP devices
        ctx->devices.clear();
        std::size_t at = 0;
        while (at < value.size()) {
            std::size_t used = 0;
            ctx->devices.push_back(std::stoi(value.substr(at), &used));
            at += used;
            while (at < value.size() && (value[at] == ',' || value[at] == ' '))
                ++at;
        }
    } else if (name == "sweep-budget") // extension: bounded optimize() for tests
        ctx->sweep_budget = std::stol(value);
}

// Range checks and derived values as lib/src/private.cpp:50-172.
void
context_set_solver_parameters(const context_ptr& ctx, const solver_parameters& params)
{
    if (!ctx)
        return;
    auto& q = ctx->parameters;
    q.time_limit = params.time_limit <= 0 ? std::numeric_limits<double>::infinity()
                                          : params.time_limit;
    if (params.theta >= 0 && params.theta <= 1)
        q.theta = params.theta;
    if (params.delta == -1 || params.delta >= 0)
        q.delta = params.delta;
    if (params.kappa_min < params.kappa_max && params.kappa_min >= 0 && params.kappa_min < 1 &&
        params.kappa_max <= 1 && params.kappa_step >= 0 && params.kappa_step < 1) {
        q.kappa_min = params.kappa_min;
        q.kappa_step = params.kappa_step;
        q.kappa_max = params.kappa_max;
    }
    if (params.alpha >= 0)
        q.alpha = params.alpha;
    if (std::isfinite(params.pushing_k_factor))
        q.pushing_k_factor = params.pushing_k_factor;
    if (std::isfinite(params.pushing_objective_amplifier))
        q.pushing_objective_amplifier = params.pushing_objective_amplifier;
    auto clamp01 = [](double v, double& out) {
        if (std::isfinite(v))
            out = std::clamp(v, 0.0, 1.0);
    };
    clamp01(params.init_crossover_bastert_insertion, q.init_crossover_bastert_insertion);
    clamp01(params.init_crossover_solution_selection_mean, q.init_crossover_solution_selection_mean);
    clamp01(params.init_crossover_solution_selection_stddev, q.init_crossover_solution_selection_stddev);
    clamp01(params.init_mutation_variable_mean, q.init_mutation_variable_mean);
    clamp01(params.init_mutation_variable_stddev, q.init_mutation_variable_stddev);
    clamp01(params.init_mutation_value_mean, q.init_mutation_value_mean);
    clamp01(params.init_mutation_value_stddev, q.init_mutation_value_stddev);
    if (params.init_policy_random >= 0 && params.init_policy_random <= 1)
        q.init_policy_random = params.init_policy_random;
    q.seed = params.seed;
    q.thread = params.thread;
    q.limit = params.limit <= 0 ? std::numeric_limits<long int>::max() : params.limit;
    if (params.print_level >= 0)
        q.print_level = params.print_level;
    if (params.w < 0) {
        q.w = 0;
    } else if (params.limit > 0) {
        const auto limit = static_cast<double>(params.limit);
        q.w = params.w < 1 ? limit * params.w : std::min(limit, params.w);
    } else {
        q.w = std::floor(params.w);
    }
    q.pushes_limit = params.pushes_limit <= 0 ? 0 : params.pushes_limit;
    q.pushing_iteration_limit = params.pushing_iteration_limit <= 0 ? 0 : params.pushing_iteration_limit;
    if (params.init_population_size >= 5)
        q.init_population_size = params.init_population_size;
    q.init_kappa_improve_start = std::clamp(params.init_kappa_improve_start, 0.0, 1.0);
    q.init_kappa_improve_increase = std::clamp(params.init_kappa_improve_increase, 0.0, 1.0);
    q.init_kappa_improve_stop =
      std::clamp(params.init_kappa_improve_stop, q.init_kappa_improve_start, 1.0);
    q.pre_order = params.pre_order;
    q.order = params.order;
    q.float_type = params.float_type;
    q.init_policy = params.init_policy;
    q.cost_norm = params.cost_norm;
    q.mode = params.mode;
    q.preprocessor = params.preprocessor;
    q.observer = params.observer;
    q.storage = params.storage;
    q.solver = params.solver;
    q.debug = params.debug;
}

solver_parameters
context_get_solver_parameters(const context_ptr& ctx)
{
    return ctx ? ctx->parameters : solver_parameters{};
}

void
context_register(const context_ptr& ctx,
                 solver_started_cb start,
                 solver_updated_cb update,
                 solver_finished_cb finish)
{
    if (ctx) {
        ctx->start = std::move(start);
        ctx->update = std::move(update);
        ctx->finish = std::move(finish);
    }
}

// lpcore.cpp:85-96: the preprocessor runs unless the parameters say `none`
static problem
prepared(const context& ctx, const raw_problem& rawpb)
{
    return ctx.parameters.preprocessor == solver_parameters::preprocessor_options::all
             ? preprocess(ctx, rawpb)
             : unpreprocess(ctx, rawpb);
}

result
solve(const context_ptr& ctx, const raw_problem& rawpb)
{
    if (!ctx)
        return result{ result_status::empty_context };
    return itm::solve(*ctx, prepared(*ctx, rawpb));
}

result
optimize(const context_ptr& ctx, const raw_problem& rawpb)
{
    if (!ctx)
        return result{ result_status::empty_context };
    // the branch / nlopt / manual meta-optimisers wrap this call in the
    // reference (lpcore.cpp:100-132); they are host code outside the path
    return itm::optimize(*ctx, prepared(*ctx, rawpb));
}

// `name=value` lines, as the reference writes its .sol files
// (lib/src/result.cpp:77-133).
result
make_result(const context_ptr& ctx, const std::string& filename)
{
    if (!ctx)
        return result{ result_status::empty_context };
    std::ifstream ifs(filename);
    result r;
    if (!ifs.is_open()) {
        r.status = result_status::internal_error;
        return r;
    }
    r.strings = std::make_shared<string_buffer>();
    r.solutions.emplace_back();
    std::string line;
    while (std::getline(ifs, line)) {
        if (line.empty() || line[0] == '\\')
            continue;
        const auto eq = line.find('=');
        if (eq == std::string::npos)
            continue;
        r.variable_name.emplace_back(r.strings->append(line.substr(0, eq)));
        r.solutions.back().variables.emplace_back(std::stoi(line.substr(eq + 1)) != 0);
    }
    r.variables = static_cast<index>(r.variable_name.size());
    r.status = result_status::success;
    return r;
}

} // namespace baryonyx
