/* oracle/oracle_capi.h -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * C entry points of the CPU oracle (oracle/itm_oracle.hpp), bound with ctypes
 * by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg only.
 */
#ifndef ORACLE_CAPI_H
#define ORACLE_CAPI_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct orc_problem
{
    int32_t m;              /* merged constraints                          */
    int32_t n;              /* variables                                   */
    int64_t nnz;            /* elements                                    */
    const int32_t* row_ptr; /* [m+1]                                       */
    const int32_t* col;     /* [nnz] variable index, constraint order      */
    const int32_t* factor;  /* [nnz] coefficient, NULL = all +1            */
    const int32_t* cmin;    /* [m] merged_constraint::min                  */
    const int32_t* cmax;    /* [m] merged_constraint::max                  */
    int32_t n_obj;          /* linear objective elements                   */
    const int32_t* obj_var;
    const double* obj_factor;
    int32_t n_qobj;         /* quadratic objective elements                */
    const int32_t* q_a;
    const int32_t* q_b;
    const double* q_factor;
    double obj_constant;
} orc_problem;

typedef struct orc_params
{
    double theta, delta, kappa_min, kappa_step, kappa_max, alpha;
    double pushing_k_factor, pushing_objective_amplifier, init_policy_random;
    double init_crossover_bastert_insertion;
    double init_crossover_solution_selection_mean;
    double init_crossover_solution_selection_stddev;
    double init_mutation_variable_mean, init_mutation_variable_stddev;
    double init_mutation_value_mean, init_mutation_value_stddev;
    double init_kappa_improve_start, init_kappa_improve_increase;
    double init_kappa_improve_stop;
    double w;
    int64_t limit;
    uint32_t seed;
    int32_t pushes_limit, pushing_iteration_limit, init_population_size;
    int32_t order, init_policy, cost_norm, sort_mode;
} orc_params;

typedef struct orc_result
{
    int32_t status;
    int32_t remaining_constraints;
    int32_t has_solution_vector;
    int32_t contract_violations; /* row updates behind which the reference's default build stops
                                    (bx_expects(is_valid_constraint), itm-common.hpp:464,
                                    itm-solver-inequalities-Z.cpp:436); saturates at INT32_MAX */
    double value;
    int64_t loop;
    uint64_t row_updates;
    uint64_t sweeps;
} orc_result;

void orc_default_params(orc_params* p);

/* Step-wise Solver handle (kind: 0 eq01, 1 ineq01, 2 eq101, 3 ineq101, 4 Z;
 * float_type: 0 float, 1 double). NULL on error (message via orc_last_error). */
void* orc_solver_create(const orc_problem* pb,
                        int kind,
                        int float_type,
                        int maximize,
                        int sort_mode,
                        int cost_norm,
                        uint32_t seed);
void orc_solver_destroy(void* h);
const char* orc_last_error(void);

int orc_solver_nwords(void* h); /* n/64 + 1 */
int orc_solver_rmax(void* h);
double orc_solver_compute_delta(void* h, double theta);
void orc_solver_get_cost(void* h, double* out_n); /* normalised, as Float */
/* Solver::reset(): P = pi = 0 */
void orc_solver_reset(void* h);
/* normalised quadratic costs as adjacency lists (quadratic_cost_type::indices /
 * quadratic_elements); null arrays: only the count is returned */
int orc_solver_get_qcost(void* h, int32_t* qptr_n1, int32_t* qid, double* qfactor);
void orc_solver_get_x(void* h, uint64_t* words);
void orc_solver_set_x(void* h, const uint64_t* words);
void orc_solver_init_x(void* h, int init_policy, double init_policy_random);
void orc_solver_get_pi(void* h, double* out_m);
void orc_solver_set_pi(void* h, const double* in_m);
void orc_solver_get_P(void* h, double* out_nnz); /* CSR element order */
void orc_solver_set_P(void* h, const double* in_nnz);
uint32_t orc_solver_get_rng(void* h);
void orc_solver_set_rng(void* h, uint32_t state);
uint64_t orc_solver_row_updates(void* h);
uint64_t orc_solver_tie_rows(void* h); /* rows that met equal reduced costs */
/* The reference's working domain: row updates of this solver behind which the
 * reference's `bx_expects(is_valid_constraint(slv, k, x))` (itm-common.hpp:464,
 * itm-solver-inequalities-Z.cpp:436) fails -- std::terminate in its default
 * build, nothing in its BARYONYX_FULL_OPTIMIZATION build (debug.hpp:45-111).
 * The oracle behaves as the latter and counts.  orc_contract_violations_total:
 * the same over every solver, solve and optimize of the process. */
uint64_t orc_solver_contract_violations(void* h);
uint64_t orc_contract_violations_total(void);
/* on = 0: skip the check (and the count), as the full-optimisation build does */
void orc_solver_set_contract_checks(void* h, int on);
double orc_solver_objective(void* h); /* Cost::results(x, constant) */

/* compute_order::init / run / push_and_run */
void orc_solver_order_init(void* h, int order);
int orc_solver_violated(void* h, int32_t* out_rows); /* current list */
int orc_solver_sweep(void* h,
                     double kappa,
                     double delta,
                     double theta,
                     int is_push,
                     double obj_amp);
/* explicit row list, no violated-list recomputation; returns pi_changed */
int orc_solver_update_rows(void* h,
                           const int32_t* rows,
                           int nrows,
                           double kappa,
                           double delta,
                           double theta,
                           int is_push,
                           double obj_amp);

/* whole solver_functor (itm-solver-common.hpp:63-225) behind solve_problem's
 * guard (:269-297): no merged constraint or no variable -> success, value =
 * the objective's constant, no solution vector */
int orc_solve(const orc_problem* pb,
              int kind,
              int float_type,
              int maximize,
              const orc_params* params,
              orc_result* out,
              uint64_t* out_x_words);

/* optimize_problem (itm-optimizer-common.hpp:776-908) with `thread` workers for
 * `rounds` rounds of the deterministic schedule of oracle/itm_oracle_opt.hpp.
 * out: the incumbent; pop_*: the whole population, best first ([population]
 * entries; pop_x [population][nwords]); worker_*: [thread]. */
int orc_optimize(const orc_problem* pb,
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
                 double* sweep_seconds); /* time inside the sweeps, summed over the workers; may be NULL */

/* Z sub-solvers on their own (lib/test/lib.cpp:376-563) */
int orc_bb_solve(int maximize,
                 int n,
                 double* values,
                 int32_t* ids,
                 int32_t* factors,
                 int bk_min,
                 int bk_max);
int orc_exhaustive_solve(int maximize,
                         int n,
                         const int32_t* row_factors,
                         int bk_min,
                         int bk_max,
                         double* values,
                         int32_t* ids,
                         int32_t* factors);

/* bit_array_impl::set(int index, bool val) exactly as lib/src/bit-array.hpp:144-149
 * writes it (the mutation loop's only bit operation, itm-optimizer-common.hpp:523) */
/* `words`: index / 64 + 1 blocks, read and written */
void orc_bitarray_set_value(uint64_t* words, int index, int val);

/* A bit_array of `size` bits (zeros) taken through `nops` operations
 * (op, p, q): 0 zeros, 1 ones, 2 set(p), 3 unset(p), 4 invert(p), 5 get(p) -> next
 * slot of `gets`, 6 assign(all-zeros array, p, q), 7 assign(all-ones array, p, q),
 * 8 set(p, bool q) as the reference writes it.  `words` receives the size / 64 + 1
 * blocks; returns the block count.  (lib/test/lib.cpp:677-783) */
int orc_bitarray_script(int size, const int* ops, int nops, uint64_t* words, int* gets);

#ifdef __cplusplus
}
#endif
#endif
