/* include/bxh.h -- C ABI of the MI355X device layer for baryonyx's
 * in-the-middle row update.
 *
 * The reference has no FFI: the seam is its duck-typed C++ *Solver concept*
 * (SURVEY.md 8(b)-2), i.e. what `compute_order::run / push_and_run`
 * (lib/src/itm-common.hpp:682-880), `solver_functor`
 * (lib/src/itm-solver-common.hpp:63-225) and `optimize_functor`
 * (lib/src/itm-optimizer-common.hpp:603-753) call on a solver object.  Each
 * entry point below names the reference interface it stands in for.  One
 * handle == a batch of `batch` independent Solver instances over ONE
 * constraint matrix (the reference builds one Solver per worker thread,
 * itm-optimizer-common.hpp:649-653).
 *
 * Conventions: every function returns 0 or a negative bxh_error; no C++
 * exception crosses the ABI; a handle is used by one host thread at a time
 * (the rule `baryonyx::context` has, lib/include/baryonyx/core:483-489); host
 * buffers are owned by the caller; all device memory is owned by the handle.
 * Floating-point state crosses the boundary as double (exact for both device
 * float kinds).
 */
#ifndef BXH_H
#define BXH_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct bxh_handle_s* bxh_handle;

enum bxh_error
{
    BXH_OK = 0,
    BXH_ERR_INVALID = -1,     /* bad argument                                */
    BXH_ERR_OOM = -2,         /* -> solver_failure(not_enough_memory),
                                 lib/include/baryonyx/core:258-263            */
    BXH_ERR_UNSUPPORTED = -3, /* valid in the reference, not on the device yet */
    BXH_ERR_DEVICE = -4       /* HIP runtime error (see bxh_last_error), or a
                                 workgroup gave up waiting for its partners: the
                                 state of the handle's instances is undefined
                                 from then on, destroy the handle              */
};

/* Solver family == which itm-solver-*.cpp the handle replaces. */
enum bxh_kind
{
    BXH_EQ_01 = 0,    /* itm-solver-equalities-01.cpp    */
    BXH_INEQ_01 = 1,  /* itm-solver-inequalities-01.cpp  */
    BXH_EQ_101 = 2,   /* itm-solver-equalities-101.cpp   */
    BXH_INEQ_101 = 3, /* itm-solver-inequalities-101.cpp */
    BXH_INEQ_Z = 4    /* itm-solver-inequalities-Z.cpp   */
};

/* floating_point_type, lib/include/baryonyx/core:340-345 (long double has no
 * device form -> BXH_ERR_UNSUPPORTED). */
enum bxh_float
{
    BXH_FLOAT = 0,
    BXH_DOUBLE = 1
};

/* solver_parameters::constraint_order, lib/include/baryonyx/core:347-358 */
enum bxh_order
{
    BXH_ORDER_NONE = 0,
    BXH_ORDER_REVERSING = 1,
    BXH_ORDER_RANDOM_SORTING = 2,
    BXH_ORDER_INFEASIBILITY_DECR = 3,
    BXH_ORDER_INFEASIBILITY_INCR = 4,
    BXH_ORDER_LAGRANGIAN_DECR = 5,
    BXH_ORDER_LAGRANGIAN_INCR = 6,
    BXH_ORDER_PI_SIGN_CHANGE = 7,
    BXH_ORDER_CYCLE = 8
};

/* What a Solver constructor receives: `std::vector<merged_constraint>` in CSR
 * form (lib/src/itm-common.hpp:55-71) and the NORMALISED cost functor
 * (`norm_costs`, itm-solver-common.hpp:76-77).  The handle builds the CSC
 * view itself, as `sparse_matrix<int>` does (lib/src/sparse-matrix.hpp:86-186),
 * and clamps the bounds per solver kind as each constructor does. */
typedef struct bxh_problem
{
    int32_t m;              /* merged constraints                           */
    int32_t n;              /* variables                                    */
    int64_t nnz;            /* elements                                     */
    const int32_t* row_ptr; /* [m+1]                                        */
    const int32_t* col;     /* [nnz] variable_index, constraint order       */
    const int32_t* factor;  /* [nnz] coefficient; NULL = all +1             */
    const int32_t* cmin;    /* [m] merged_constraint::min                   */
    const int32_t* cmax;    /* [m] merged_constraint::max                   */
    const double* cost;     /* [n] normalised linear cost, already rounded
                                    to the device float kind                */
    int32_t n_quad;         /* entries of quadratic_cost_type's adjacency   */
    const int32_t* q_ptr;   /* [n+1]   quadratic_cost_type::indices         */
    const int32_t* q_id;    /* [n_quad]                                     */
    const double* q_factor; /* [n_quad]                                     */
} bxh_problem;

/* Arguments of one sweep of every instance == one call of
 * compute_order::run (itm-common.hpp:800-880) or, with is_push,
 * compute_order::push_and_run (:682-798). */
typedef struct bxh_sweep_args
{
    const double* kappa;   /* [batch] per-instance kappa                    */
    double delta;
    double theta;
    double obj_amp;        /* objective amplifier (push only)               */
    int32_t is_push;
    int32_t order;         /* bxh_order                                     */
    const uint8_t* active; /* [batch] 0 = skip this instance; NULL = all    */
} bxh_sweep_args;

/* Per-instance outcome of a sweep (`remaining` is what run() returns). */
typedef struct bxh_status
{
    int32_t remaining;    /* violated constraints after the sweep           */
    int32_t pi_changed;   /* OR of affect()'s sign-change results           */
    uint32_t rng;         /* minstd_rand0 state after the sweep             */
    uint32_t rows_in_sweep;
    uint64_t row_updates; /* cumulative count of row-loop bodies executed   */
    double alg_bytes;     /* cumulative algorithmic bytes of those row
                             updates (bxh_row_bytes summed over them)       */
} bxh_status;

const char* bxh_last_error(void);

/* Solver constructors (e.g. itm-solver-inequalities-01.cpp:72-107;
 * itm-solver-inequalities-Z.cpp:125-187 including the choice of the row
 * sub-solvers and exhaustive_solver::build_constraints,
 * exhaustive-solver.hpp:111-167), `batch` of them, on HIP device `device`.
 * P and pi start at zero, x at zero. */
int bxh_create(const bxh_problem* pb,
               int kind,
               int float_type,
               int maximize,
               int batch,
               int device,
               bxh_handle* out);
/* The same constructors for a constraint matrix that is already resident on
 * HIP device `device`: every pointer of `pb` is a DEVICE pointer.  The handle
 * builds the column view of `sparse_matrix<int>` (lib/src/sparse-matrix.hpp:
 * 86-186: stable by column) and the clamped bounds with kernels, once, shared
 * by all `batch` instances -- the reference rebuilds them per worker thread
 * (itm-optimizer-common.hpp:649-653); no host loop touches the elements
 * (2e9 of them at the 100k x 1M shape).  Kinds BXH_EQ_01 .. BXH_INEQ_101,
 * linear objective (n_quad == 0).  pb->factor may be NULL (all +1).  The call
 * first waits for all work queued on the device (the arrays may come from any
 * stream) and has copied what it needs when it returns. */
int bxh_create_device(const bxh_problem* pb_on_device,
                      int kind,
                      int float_type,
                      int maximize,
                      int batch,
                      int device,
                      bxh_handle* out);
int bxh_destroy(bxh_handle h);

/* Launch stream (a hipStream_t, e.g. torch's current stream); NULL = the
 * handle's own stream. */
int bxh_set_stream(bxh_handle h, void* hip_stream);
int bxh_batch(bxh_handle h);
int bxh_nwords(bxh_handle h); /* bit_array block count n/64 + 1, bit-array.hpp:60-66 */
int bxh_rmax(bxh_handle h);
/* Which form of the row-update kernel bxh_create chose for this handle
 * (diagnostics and bench labels; the results do not depend on it):
 *   1  lane-per-column walk, one workgroup per instance
 *   2  staged pipeline, one workgroup per instance
 *   3  inequalities-Z kernel
 *   5  short-row kernel (one or four waves per instance)
 *   6  inequalities-Z kernel, a wave per row: up to four rows of an instance that
 *      share no column run side by side (linear objective)
 *   7  01 / 101 solvers, every row of at most 63 elements: the same wave-per-row form
 *   20 + g  staged pipeline, g lock-step column owners per instance
 *   40 + w  pipelined sweep, w workgroups per instance (1 row chain + w - 1 producers)
 *   100 + g wide kernel (columns / row counts beyond the staged kernel: the XL
 *           shape), g workgroups per instance share every row */
int bxh_kernel_form(bxh_handle h);
/* Several sweeps of solver_functor's main loop (lib/src/itm-solver-common.hpp:
 * 135-213) in ONE launch, for the handles bxh_run_supported() says yes to (the
 * short-row kernels, forms 5 and 7) and the row orders none / reversing / random_sorting:
 *   for (i = first_i; ; ++i) { remaining = compute.run(kappa);
 *       if (remaining == 0) stop (1);
 *       if (remaining < best) { best = remaining; keep x; }
 *       if (i > w_limit) kappa += kappa_inc[remaining];
 *       if (kappa > kappa_max) stop (2);
 *       if (sweeps == max_sweeps) stop (0); }
 * kappa_inc[r] is the host's kappa_step * pow(r / m, alpha) evaluated in the
 * handle's float type for r = 0 .. m (so the device adds exactly what the host
 * would).  pi, x and the row list stay on chip between the sweeps; the status
 * blocks are those behind the last sweep, row_updates / alg_bytes cover all. */
typedef struct bxh_run_args
{
    const double* kappa;      /* [batch] kappa of the first sweep */
    double delta, theta;
    int order;
    int max_sweeps, first_i, w_limit;
    double kappa_max;
    const double* kappa_inc;  /* [m + 1] */
    const unsigned char* active;
} bxh_run_args;
typedef struct bxh_run_status
{
    int sweeps;         /* sweeps run */
    int best_remaining; /* fewest violated rows behind a sweep of this launch (INT_MAX: feasible at once) */
    int best_i;         /* ... its loop index */
    int stop;           /* 0 max_sweeps, 1 no row violated, 2 kappa > kappa_max */
    double kappa;       /* kappa behind the last sweep's update */
} bxh_run_status;
int bxh_run_supported(bxh_handle h);
int bxh_run(bxh_handle h, const bxh_run_args* a, bxh_status* status_batch, bxh_run_status* run_batch);
/* x behind the sweep best_remaining refers to (solver_functor::store_if_better) */
int bxh_get_best_x(bxh_handle h, int inst, uint64_t* words);

/* 1: compute_violated_constraints runs inside the sweep kernel (its bytes belong
 * to that kernel's roofline); 0: in launches of its own behind it (the wide
 * kernel with at most 32 instances: one pass over A for all of them) */
int bxh_sweep_includes_scan(bxh_handle h);

/* Solver fields x / pi / P (P in CSR element order, as `slv.P[e]`), and the
 * worker's random engine. `inst` in [0, batch). */
int bxh_set_x(bxh_handle h, int inst, const uint64_t* words);
int bxh_get_x(bxh_handle h, int inst, uint64_t* words);
/* x of every instance at once, [batch][nwords], stream-ordered (what the
 * restart loop does between two runs: reinit writes x,
 * itm-optimizer-common.hpp:660-663); host or device memory */
int bxh_set_x_all(bxh_handle h, const uint64_t* words_batch);
/* x / the engine state of every instance in one transfer each ([batch][nwords],
 * [batch]): what optimize()'s bookkeeping reads and writes once per round for
 * its whole worker pool */
int bxh_get_x_all(bxh_handle h, uint64_t* words_batch);
int bxh_set_rng_all(bxh_handle h, const uint32_t* state_batch);

/* The evolutionary layer of optimize() with the population on the device
 * (`storage` / `best_solution_recorder`, lib/src/itm-optimizer-common.hpp:93-601):
 * the bit vectors of the population live in device memory, a solution enters
 * it by a device copy of its instance's x (bxh_pop_store), and a restart builds
 * the instance's new x there -- block-wise crossover ((x1 ^ x2) & mask) ^ x1 with
 * the masks drawn from the worker's engine (:359-371) and the mutation loop
 * (:494-526: one Bernoulli draw per variable, a second where it hits) -- so
 * that no bit vector crosses PCIe at a restart.  The host keeps what it
 * decides: which solutions are crossed (choose_a_solution's normal draws) and
 * the two mutation probabilities; it passes the engine state in front of the
 * masks and in front of the mutation loop (the masks take a number of draws the
 * host can count without forming them), the device leaves the state behind
 * the loop in the instance's engine. */
typedef struct bxh_reinit_job
{
    int32_t inst;       /* instance whose x is rebuilt */
    int32_t cross;      /* 0: x stays; 1: crossover of slots first / second (the two extra slots,
                           indices slots and slots + 1, hold what the host puts there: the Bastert
                           vector, a random vector) */
    int32_t first, second;
    uint32_t rng_cross; /* engine state in front of the crossover's mask draws */
    int32_t mutate;     /* 0 / 1 */
    uint32_t rng_mut;   /* engine state in front of the mutation's draws (== behind the masks) */
    uint32_t pad_;
    double var_p, val_p; /* mutation: P(variable is redrawn), P(it becomes 1) */
} bxh_reinit_job;
/* `slots` vectors + two extra (indices `slots`, `slots + 1`) */
int bxh_pop_create(bxh_handle h, int slots);
int bxh_pop_set(bxh_handle h, int slot, const uint64_t* words);
int bxh_pop_get(bxh_handle h, int slot, uint64_t* words);
/* population[slot] <- x of instance `inst` (device copy, on the stream) */
int bxh_pop_store(bxh_handle h, int slot, int inst);
/* the jobs' crossovers and mutations, one workgroup per job, on the stream;
 * afterwards x and the engine state of every job's instance are the new ones */
int bxh_reinit(bxh_handle h, int njobs, const bxh_reinit_job* jobs);

int bxh_set_pi(bxh_handle h, int inst, const double* pi_m);
int bxh_get_pi(bxh_handle h, int inst, double* pi_m);
int bxh_set_P(bxh_handle h, int inst, const double* P_nnz);
int bxh_get_P(bxh_handle h, int inst, double* P_nnz);
int bxh_set_rng(bxh_handle h, int inst, uint32_t state);
int bxh_get_rng(bxh_handle h, int inst, uint32_t* state);
/* Solver::reset() (e.g. itm-solver-inequalities-01.cpp:109-113) */
int bxh_reset(bxh_handle h, int inst);
/* the same for every instance, enqueued on the stream (no synchronisation) */
int bxh_reset_all(bxh_handle h);

/* compute_order's constructor (itm-common.hpp:640-658) for every instance:
 * `order` is a bxh_order (solver_parameters::constraint_order, core:286-296);
 * BXH_ORDER_CYCLE starts as `none` and every instance steps to the next order
 * at each of its push sweeps (itm-common.hpp:686-687).  R and m_order start
 * as the reference's value-initialised lists of `n` entries.  pi_sign_change
 * (and cycle) walk row ids 0..n-1 (itm-common.hpp:672-674): BXH_ERR_UNSUPPORTED
 * when n > m, where the reference reads past its arrays. */
int bxh_order_create(bxh_handle h, int order);
/* compute_order::init (itm-common.hpp:660-680): rebuild every instance's
 * list (R, or m_order for the infeasibility orders, or iota for
 * pi_sign_change) from its current x; remaining[] may be NULL, in which
 * case the call only enqueues the kernel on the stream.  An `order` other than
 * the one last given to bxh_order_create first re-creates the object. */
int bxh_order_init(bxh_handle h, int order, int32_t* remaining_batch);
/* the same for the instances with active[i] != 0 only (a restart of some
 * workers, itm-optimizer-common.hpp:663-666); active may be NULL = all */
int bxh_order_init_masked(bxh_handle h, int order, const uint8_t* active_batch,
                          int32_t* remaining_batch);
/* the list one instance walks next (R, or the rows of m_order); rows needs
 * room for max(m, n) entries */
int bxh_get_violated(bxh_handle h, int inst, int32_t* rows, int32_t* count);

/* One sweep of all active instances; blocks until the status is on the host. */
int bxh_sweep(bxh_handle h, const bxh_sweep_args* args, bxh_status* status_batch);
/* Same without the final synchronisation / copy: for back-to-back launches. */
int bxh_sweep_async(bxh_handle h, const bxh_sweep_args* args);
int bxh_status_sync(bxh_handle h, bxh_status* status_batch);

/* The incumbent of the batch, found on the device: the instance that
 * best_solution_recorder would keep -- fewest violated rows (as the last sweep
 * or bxh_order_init left them), then the better objective (`storage`'s order,
 * lib/src/itm-optimizer-common.hpp:424-442; raw_result_compare, lib/src/result.hpp:96-107) -- for the exchange step of
 * the restart pool.  bxh_set_objective uploads the ORIGINAL
 * linear costs once (Cost::results sums those, itm-common.hpp:1140-1147).
 * bxh_incumbent launches two small kernels on the handle's stream and leaves a
 * 32-byte record and the winner's x ([nwords] bit_array blocks) in DEVICE memory:
 * `record_device` / `x_device` when given (e.g. tensors the caller all-reduces
 * over RCCL), the handle's own buffers otherwise.  `record_host` non-NULL: also
 * synchronise and copy the record back -- the ONE read-back of an exchange.
 * key_hi / key_lo: (violated rows << 44 | top 44 bits of the objective's ordered
 * image), (its low 20 bits << 6 | rank) -- compared lexicographically, both
 * non-negative; MIN names the best also when maximising.  The objective of an
 * instance with violated rows counts as +inf (-inf when maximising).  `value`
 * is summed in a fixed tree: exact for integer costs, within rounding of the
 * reference's sequential sum otherwise. */
typedef struct bxh_incumbent_record
{
    int64_t key_hi;
    int64_t key_lo;
    double value;
    int32_t remaining;
    int32_t instance;
} bxh_incumbent_record;
int bxh_set_objective(bxh_handle h, const double* cost_n, double constant, int maximize);
int bxh_incumbent(bxh_handle h, int rank, void* record_device, void* x_device, bxh_incumbent_record* record_host);

/* Device time of the sweep kernels launched through this handle, measured with
 * HIP events recorded on the launch stream around every kernel (the H2D copy
 * of kappa[] is outside the pair).  Synchronises the stream. */
int bxh_kernel_time(bxh_handle h, double* total_ms, uint64_t* launches, int reset);

/* Algorithmic bytes of one row update of row k:
 * sum_{j in row k} c_j (4 + 2F) + r_k (4 + 5F) + 2F + ceil(r_k / 8) 2
 * (inequalities-Z: twice the first term for the repair pass, + 4 r_k),
 * summed over the rows of the last sweep per instance is reported by bench.py;
 * this returns the per-row table so the host can do that sum. */
int bxh_row_bytes(bxh_handle h, double* bytes_m);

#ifdef __cplusplus
}
#endif
#endif
