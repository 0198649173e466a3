// baryonyx_amd/host/rsurface.cpp -- the rbaryonyx surface (include/rbaryonyx.h):
// the two exported functions of rbaryonyx/src/rbaryonyx.cpp over this library's
// baryonyx::solve / baryonyx::optimize.
//   code tables      rbaryonyx.cpp:119-238
//   convert_result   rbaryonyx.cpp:241-289
//   solve_01lp_problem :368-447, optimize_01lp_problem :519-618
#include "../../include/rbaryonyx.h"
#include <baryonyx/core>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>

namespace {

namespace bx = baryonyx;
using sp = bx::solver_parameters;

sp::pre_constraint_order
get_pre_constraint_order(int prep) // :119-149 (1 and every unknown code: memory)
{
    switch (prep) {
    case 0:
        return sp::pre_constraint_order::none;
    case 2:
        return sp::pre_constraint_order::less_greater_equal;
    case 3:
        return sp::pre_constraint_order::less_equal_greater;
    case 4:
        return sp::pre_constraint_order::greater_less_equal;
    case 5:
        return sp::pre_constraint_order::greater_equal_less;
    case 6:
        return sp::pre_constraint_order::equal_less_greater;
    case 7:
        return sp::pre_constraint_order::equal_greater_less;
    case 8:
        return sp::pre_constraint_order::p1;
    case 9:
        return sp::pre_constraint_order::p2;
    case 10:
        return sp::pre_constraint_order::p3;
    case 11:
        return sp::pre_constraint_order::p4;
    default:
        return sp::pre_constraint_order::memory;
    }
}

sp::constraint_order
get_constraint_order(int order) // :151-177
{
    switch (order) {
    case 1:
        return sp::constraint_order::reversing;
    case 2:
        return sp::constraint_order::random_sorting;
    case 3:
        return sp::constraint_order::infeasibility_decr;
    case 4:
        return sp::constraint_order::infeasibility_incr;
    case 5:
        return sp::constraint_order::lagrangian_decr;
    case 6:
        return sp::constraint_order::lagrangian_incr;
    case 7:
        return sp::constraint_order::pi_sign_change;
    case 8:
        return sp::constraint_order::cycle;
    default:
        return sp::constraint_order::none;
    }
}

sp::cost_norm_type
get_cost_norm(int norm) // :179-196
{
    switch (norm) {
    case 0:
        return sp::cost_norm_type::none;
    case 1:
        return sp::cost_norm_type::random;
    case 2:
        return sp::cost_norm_type::l1;
    case 3:
        return sp::cost_norm_type::l2;
    default:
        return sp::cost_norm_type::loo;
    }
}

sp::init_policy_type
get_init_policy(int type) // :198-213
{
    switch (type) {
    case 1:
        return sp::init_policy_type::pessimistic_solve;
    case 2:
        return sp::init_policy_type::optimistic_solve;
    default:
        return sp::init_policy_type::bastert;
    }
}

sp::floating_point_type
get_floating_point_type(int type) // :215-227
{
    switch (type) {
    case 0:
        return sp::floating_point_type::float_type;
    case 2:
        return sp::floating_point_type::longdouble_type;
    default:
        return sp::floating_point_type::double_type;
    }
}

sp::storage_type
get_storage_type(int type) // :229-238
{
    switch (type) {
    case 1:
        return sp::storage_type::bound;
    case 2:
        return sp::storage_type::five;
    default:
        return sp::storage_type::one;
    }
}

// the three callbacks of the R glue (:30-111) print through Rprintf; here: stdout
void
updated_cb(int remaining_constraints, double value, int loop, double duration, long int reinit_number)
{
    if (remaining_constraints > 0)
        std::printf("  - Constraints remaining: %d (loop: %d t: %fs reinit:%ld)\n", remaining_constraints, loop,
                    duration, reinit_number);
    else if (loop >= 0)
        std::printf("  - Solution found: %f (loop: %d t: %fs reinit:%ld)\n", value, loop, duration, reinit_number);
    else
        std::printf("  - Solution found via push: %f (loop: %d t: %fs reinit:%ld)\n", value, -loop, duration,
                    reinit_number);
    std::fflush(stdout);
}

void
started_cb(const bx::solver_parameters&) // :30-35 (flushes the R console)
{
    std::fflush(stdout);
}

void
finished_cb(const bx::result& r) // :70-111
{
    switch (r.status) {
    case bx::result_status::success:
        if (r.loop >= 0)
            std::printf("Best solution found: %f in %ld loop and %fs\n", r.solutions.back().value, r.loop,
                        r.duration);
        else
            std::printf("Best solution found via push: %f in %ld loop and %fs\n", r.solutions.back().value,
                        -r.loop, r.duration);
        break;
    case bx::result_status::internal_error:
        std::printf("No solution. Internal error\n");
        break;
    case bx::result_status::uninitialized:
        std::printf("No solution. Uninitialized error\n");
        break;
    case bx::result_status::kappa_max_reached:
        std::printf("No solution. Constraint remaining: %d. Kappa reached in %fs.\n", r.remaining_constraints,
                    r.duration);
        break;
    case bx::result_status::time_limit_reached:
        std::printf("No solution. Constraint remaining: %d. Time limit reached at %fs.\n",
                    r.remaining_constraints, r.duration);
        break;
    case bx::result_status::limit_reached:
        std::printf("No solution. Constraint remaining: %d. Loop limit reached in %fs.\n",
                    r.remaining_constraints, r.duration);
        break;
    default:
        break;
    }
    std::fflush(stdout);
}

void
convert_result(const bx::result& res, bool minimize, bxr_result* out) // :241-289
{
    std::memset(out, 0, sizeof *out);
    out->duration = res.duration;
    out->variables = res.variables;
    out->constraints = res.constraints;
    out->remaining_constraints = res.remaining_constraints;
    out->minimize = minimize ? 1 : 0;
    switch (res.status) {
    case bx::result_status::success:
        out->value = res.solutions.back().value;
        out->solution_found = 1;
        break;
    case bx::result_status::internal_error:
        out->error_found = 1;
        break;
    default:
        break;
    }
    out->n_solutions = static_cast<int>(res.solutions.size());
    if (out->n_solutions > 0) {
        out->solutions = static_cast<double*>(std::malloc(sizeof(double) * res.solutions.size()));
        if (!out->solutions) {
            out->n_solutions = 0;
            out->error_found = 1;
            out->solution_found = 0;
            return;
        }
        for (size_t i = 0; i != res.solutions.size(); ++i)
            out->solutions[i] = res.solutions[i].value;
    }
}

template<typename Fill, typename Run>
int
run_glue(const char* file_path, int verbose, bxr_result* out, Fill fill, Run run)
{
    if (!out)
        return -1;
    bx::result ret;
    bool minimize = true;
    try {
        auto ctx = bx::make_context(verbose ? 6 : 4);
        // all three callbacks whatever `verbose` is (:398-401, :556-559): it only sets the log level
        bx::context_register(ctx, started_cb, updated_cb, finished_cb);
        auto pb = bx::make_problem(ctx, file_path ? file_path : "");
        minimize = (pb.type == bx::objective_function_type::minimize);
        sp params;
        fill(params);
        bx::context_set_solver_parameters(ctx, params);
        ret = run(ctx, pb);
    } catch (const std::bad_alloc& e) {
        std::fprintf(stderr, "Baryonyx memory error: %s\n", e.what());
        ret.status = bx::result_status::internal_error;
    } catch (const std::exception& e) {
        std::fprintf(stderr, "Baryonyx error: %s\n", e.what());
        ret.status = bx::result_status::internal_error;
    } catch (...) {
        std::fprintf(stderr, "Baryonyx error: unknown error\n");
        ret.status = bx::result_status::internal_error;
    }
    convert_result(ret, minimize, out);
    return 0;
}

} // namespace

extern "C" {

BX_EXPORT int
bxr_solve_01lp_problem(const char* file_path, int limit, double theta, double delta, int pre_constraint_order,
                       int constraint_order, double kappa_min, double kappa_step, double kappa_max, double alpha,
                       double w, double time_limit, int seed, int thread, int norm, double pushing_k_factor,
                       double pushing_objective_amplifier, int pushes_limit, int pushing_iteration_limit,
                       int init_policy, double init_policy_random, int float_type, int storage_type, int verbose,
                       bxr_result* out)
{
    return run_glue(
      file_path, verbose, out,
      [&](sp& params) { // :404-427, in that order
          params.time_limit = time_limit;
          params.theta = theta;
          params.delta = delta;
          params.kappa_min = kappa_min;
          params.kappa_step = kappa_step;
          params.kappa_max = kappa_max;
          params.alpha = alpha;
          params.pushing_k_factor = pushing_k_factor;
          params.pushing_objective_amplifier = pushing_objective_amplifier;
          params.init_policy_random = init_policy_random;
          params.seed = seed;
          params.thread = thread;
          params.limit = limit;
          params.w = w;
          params.pushes_limit = pushes_limit;
          params.pushing_iteration_limit = pushing_iteration_limit;
          params.pre_order = get_pre_constraint_order(pre_constraint_order);
          params.order = get_constraint_order(constraint_order);
          params.cost_norm = get_cost_norm(norm);
          params.init_policy = get_init_policy(init_policy);
          params.float_type = get_floating_point_type(float_type);
          params.storage = get_storage_type(storage_type);
      },
      [](const bx::context_ptr& ctx, const bx::raw_problem& pb) { return bx::solve(ctx, pb); });
}

BX_EXPORT int
bxr_optimize_01lp_problem(const char* file_path, int limit, double theta, double delta, int pre_constraint_order,
                          int constraint_order, double kappa_min, double kappa_step, double kappa_max, double alpha,
                          double w, double time_limit, int seed, int thread, int norm, double pushing_k_factor,
                          double pushing_objective_amplifier, int pushes_limit, int pushing_iteration_limit,
                          double init_crossover_bastert_insertion, double init_crossover_solution_selection_mean,
                          double init_crossover_solution_selection_stddev, double init_mutation_variable_mean,
                          double init_mutation_variable_stddev, double init_mutation_value_mean,
                          double init_mutation_value_stddev, double init_kappa_improve_start,
                          double init_kappa_improve_increase, double init_kappa_improve_stop, int float_type,
                          int storage_type, int verbose, bxr_result* out)
{
    return run_glue(
      file_path, verbose, out,
      [&](sp& params) { // :563-598 (no init_policy / init_policy_random here, as there)
          params.time_limit = time_limit;
          params.theta = theta;
          params.delta = delta;
          params.kappa_min = kappa_min;
          params.kappa_step = kappa_step;
          params.kappa_max = kappa_max;
          params.alpha = alpha;
          params.pushing_k_factor = pushing_k_factor;
          params.pushing_objective_amplifier = pushing_objective_amplifier;
          params.init_crossover_bastert_insertion = init_crossover_bastert_insertion;
          params.init_crossover_solution_selection_mean = init_crossover_solution_selection_mean;
          params.init_crossover_solution_selection_stddev = init_crossover_solution_selection_stddev;
          params.init_mutation_variable_mean = init_mutation_variable_mean;
          params.init_mutation_variable_stddev = init_mutation_variable_stddev;
          params.init_mutation_value_mean = init_mutation_value_mean;
          params.init_mutation_value_stddev = init_mutation_value_stddev;
          params.init_kappa_improve_start = init_kappa_improve_start;
          params.init_kappa_improve_increase = init_kappa_improve_increase;
          params.init_kappa_improve_stop = init_kappa_improve_stop;
          params.seed = seed;
          params.thread = thread;
          params.limit = limit;
          params.w = w;
          params.pushes_limit = pushes_limit;
          params.pushing_iteration_limit = pushing_iteration_limit;
          params.pre_order = get_pre_constraint_order(pre_constraint_order);
          params.order = get_constraint_order(constraint_order);
          params.cost_norm = get_cost_norm(norm);
          params.float_type = get_floating_point_type(float_type);
          params.storage = get_storage_type(storage_type);
      },
      [](const bx::context_ptr& ctx, const bx::raw_problem& pb) { return bx::optimize(ctx, pb); });
}

BX_EXPORT void
bxr_free_result(bxr_result* r)
{
    if (r) {
        std::free(r->solutions);
        r->solutions = nullptr;
        r->n_solutions = 0;
    }
}

} // extern "C"
