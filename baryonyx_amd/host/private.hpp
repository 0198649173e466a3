// baryonyx_amd/host/private.hpp -- internals of the C++ host library that
// keeps baryonyx's public API (include/baryonyx/core) and drives the HIP
// device layer (include/bxh.h).  Host code is C++ like the reference's; the
// only native dependency is libbxh.so.
#ifndef BARYONYX_AMD_HOST_PRIVATE_HPP
#define BARYONYX_AMD_HOST_PRIVATE_HPP

#include <cstdint>
#include <baryonyx/core>

#include <chrono>
#include <cstdio>
#include <istream>
#include <random>
#include <string>
#include <tuple>
#include <vector>

namespace baryonyx {

// The engine every subsystem shares, lib/src/itm-common.hpp:45.
using random_engine = std::default_random_engine;

// lib/src/private.hpp: parameters, the `method` string, the three callbacks
// and a log level.
struct context
{
    explicit context(int verbose_level)
      : log_priority(verbose_level)
    {}
    solver_parameters parameters;
    std::string method;
    solver_started_cb start;
    solver_updated_cb update;
    solver_finished_cb finish;
    int log_priority = 6;
    int device = 0; // HIP device of the batch (context_set_parameters("device"))
    // optimize(): the HIP devices the workers are dealt to, round robin
    // (context_set_parameters("devices", "0,1,2,3"); empty: `device` alone)
    std::vector<int> devices;
    long sweep_budget = 0; // > 0: optimize() also stops after that many
                           // lock-step sweeps (context_set_parameters("sweep-budget"))
    struct optimize_trace* trace = nullptr; // tests: the recorder's final state
};

// What optimize() leaves behind for the parity tests (population best first,
// every worker's engine and restart count): filled when context::trace is set.
// wall-clock phases of the last optimize() of this thread (seconds): the host's
// set-up (merged constraints, costs, the initial population with its pre-solves
// and evaluations -- what the reference does in storage's constructor), the
// device set-up (handles, population upload) and the rounds themselves
struct optimize_timing
{
    double host_setup = 0, device_setup = 0, rounds = 0;
    // of `rounds`: what the host spends outside its wait for the device -- the
    // restarts' bookkeeping, the launches, the recorder's updates
    double rounds_host = 0;
    long round_count = 0;
};
optimize_timing& last_optimize_timing();

struct optimize_trace
{
    std::vector<int> pop_remaining;
    std::vector<double> pop_value;
    std::vector<std::uint64_t> pop_hash;
    std::vector<std::vector<std::uint64_t>> pop_x;
    std::vector<unsigned> worker_rng;
    std::vector<long> worker_calls;
    long rounds = 0;
};

template<typename... Args>
inline void
info(const context& ctx, const char* fmt, Args... args)
{
    if (ctx.log_priority >= 6)
        std::fprintf(stdout, fmt, args...);
}

// Internal problem: the raw problem plus what the preprocessor fixed
// (lib/src/problem.hpp:44-166).
struct problem
{
    problem() = default;
    explicit problem(const raw_problem& pb);

    string_buffer_ptr strings;
    objective_function objective;
    std::vector<constraint> equal_constraints;
    std::vector<constraint> greater_constraints;
    std::vector<constraint> less_constraints;
    variables vars;
    affected_variables affected_vars;
    objective_function_type type = objective_function_type::maximize;
    problem_solver_type problem_type = problem_solver_type::equalities_01;

    problem_solver_type which_problem_type() const noexcept;
};

raw_problem make_problem(const context& ctx, std::istream& is) noexcept;
problem unpreprocess(const context& ctx, const raw_problem& pb);
// lib/src/preprocessor.cpp:866-913 (baryonyx_amd/host/preprocessor.cpp)
problem preprocess(const context& ctx, const raw_problem& pb);
problem affect(const context& ctx, const problem& pb, int variable_index, bool variable_value);
std::tuple<problem, problem> split(const context& ctx, const problem& pb, int variable_index_to_affect);

// One merged row b_min <= sum a x <= b_max (lib/src/itm-common.hpp:55-71).
struct merged_constraint
{
    std::vector<function_element> elements;
    int min;
    int max;
    int id;
};
std::vector<merged_constraint> make_merged_constraints(const context& ctx,
                                                      const problem& pb);

bool is_valid_solution(const problem& pb, const std::vector<var_value>& x);
double compute_solution(const problem& pb, const std::vector<var_value>& x);

namespace itm {
result solve(const context& ctx, const problem& pb);
result optimize(const context& ctx, const problem& pb);
int bit_array_script(int size, const int* ops, int nops, std::uint64_t* words, int* gets); // tests
}

} // namespace baryonyx

#endif
