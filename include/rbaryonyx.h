/* include/rbaryonyx.h -- the rbaryonyx surface over the MI355X path, as a C ABI.
 *
 * The reference's R package exports two functions (rbaryonyx/src/rbaryonyx.cpp:368-392
 * `solve_01lp_problem`, :519-551 `optimize_01lp_problem`, both [[Rcpp::export]]) that
 * translate integer codes into solver_parameters (:119-238), call baryonyx::solve /
 * baryonyx::optimize and return a named list of nine fields (:241-289).  There is no R
 * toolchain in this image, so the same two entry points are exported here with plain C
 * types: the same arguments in the same order with the same defaults and code tables, the
 * same nine result fields, every exception mapped to `error_found` as the R glue does
 * (:437-446).  The Rcpp stub a maintainer keeps in rbaryonyx.cpp is shown in
 * INTEGRATION.md section 3.
 */
#ifndef BARYONYX_AMD_RBARYONYX_H
#define BARYONYX_AMD_RBARYONYX_H

#ifdef __cplusplus
extern "C" {
#endif

/* convert_result's named list, rbaryonyx.cpp:280-288 */
typedef struct bxr_result
{
    int solution_found;        /* result.status == success */
    int error_found;           /* result.status == internal_error (incl. every exception) */
    double value;              /* solutions.back().value when solution_found, else 0 */
    double duration;
    int variables;
    int constraints;
    int remaining_constraints;
    int minimize;              /* the problem's objective type (TRUE until the file is read) */
    int n_solutions;           /* length of `solutions` */
    double* solutions;         /* every stored solution's value; bxr_free_result releases it */
} bxr_result;

/* rbaryonyx.cpp:368-392.  Codes: pre_constraint_order 0 none, 2-7 the six type orders,
 * 8-11 p1-p4, anything else memory (:119-149); constraint_order 0-8 (:151-177); norm 0 none,
 * 1 random, 2 l1, 3 l2, else loo (:179-196); init_policy 0-2 (:198-213); float_type 0 float,
 * 2 long double, else double (:215-227); storage_type 1 bound, 2 five, else one (:229-238).
 * Returns 0; the outcome is in `out` (never throws, as the `noexcept` exports). */
int bxr_solve_01lp_problem(const char* file_path,
                           int limit,                          /* = 1000 */
                           double theta,                       /* = 0.5 */
                           double delta,                       /* = -1 */
                           int pre_constraint_order,           /* = 1 */
                           int constraint_order,               /* = 0 */
                           double kappa_min,                   /* = 0.0 */
                           double kappa_step,                  /* = 1.0e-3 */
                           double kappa_max,                   /* = 0.6 */
                           double alpha,                       /* = 1.0 */
                           double w,                           /* = 0.05 */
                           double time_limit,                  /* = 10.0 */
                           int seed,                           /* = -1 */
                           int thread,                         /* = 1 */
                           int norm,                           /* = 4 */
                           double pushing_k_factor,            /* = 0.9 */
                           double pushing_objective_amplifier, /* = 5.0 */
                           int pushes_limit,                   /* = 100 */
                           int pushing_iteration_limit,        /* = 50 */
                           int init_policy,                    /* = 0 */
                           double init_policy_random,          /* = 0.5 */
                           int float_type,                     /* = 1 */
                           int storage_type,                   /* = 1 */
                           int verbose,                        /* = TRUE */
                           bxr_result* out);

/* rbaryonyx.cpp:519-551 */
int bxr_optimize_01lp_problem(const char* file_path,
                              int limit,
                              double theta,
                              double delta,
                              int pre_constraint_order,
                              int constraint_order,
                              double kappa_min,
                              double kappa_step,
                              double kappa_max,
                              double alpha,
                              double w,
                              double time_limit,
                              int seed,
                              int thread,
                              int norm,
                              double pushing_k_factor,
                              double pushing_objective_amplifier,
                              int pushes_limit,
                              int pushing_iteration_limit,
                              double init_crossover_bastert_insertion,         /* = 0.01 */
                              double init_crossover_solution_selection_mean,   /* = 0.0 */
                              double init_crossover_solution_selection_stddev, /* = 0.3 */
                              double init_mutation_variable_mean,              /* = 0.0001 */
                              double init_mutation_variable_stddev,            /* = 0.001 */
                              double init_mutation_value_mean,                 /* = 0.5 */
                              double init_mutation_value_stddev,               /* = 0.2 */
                              double init_kappa_improve_start,                 /* = 0 */
                              double init_kappa_improve_increase,              /* = 0.02 */
                              double init_kappa_improve_stop,                  /* = 0.2 */
                              int float_type,
                              int storage_type,
                              int verbose,
                              bxr_result* out);

void bxr_free_result(bxr_result* r);

#ifdef __cplusplus
}
#endif
#endif
