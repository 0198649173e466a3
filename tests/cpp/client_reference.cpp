// tests/cpp/client_reference.cpp -- a compiled client of <baryonyx/core>, built
// by tests/test_gpu_reference_held.py with
//     g++ -std=c++17 -I include client_reference.cpp -L baryonyx_amd/host -lbaryonyx_amd
// and run on the GPU box.  It drives the public C++ API exactly as a user of
// the reference library would (make_context -> context_set_solver_parameters ->
// make_problem -> solve / optimize -> is_valid_solution / compute_solution) on
// the cases the reference's own solver tests hold (lib/test/solve.cpp,
// lib/test/optimize.cpp), and prints one line per case:
//
//     case <name> key=value ...
//
// The pytest side holds the expectations (so that they sit next to the oracle's
// in tests/test_reference_held.py).  Usage:
//     client_reference <dir with the .lp fixtures> <scratch dir> [case-prefix]
#include <baryonyx/core>

#include <cstdio>
#include <cstring>
#include <fstream>
#include <functional>
#include <string>
#include <vector>

namespace bx = baryonyx;

static std::string g_lp, g_tmp, g_only;

// deterministic stand-in for the reference tests' std::random_device costs
// (uniform integers in 1..100); tests/test_reference_held.py uses the same rule
static int
drawn_cost(int draw, int i)
{
    return 1 + static_cast<int>((static_cast<long>(draw + 1) * 7919L + static_cast<long>(i) * 104729L +
                                 static_cast<long>(i) * i * 31L) %
                                100L);
}

static std::string
inline_file(const char* name, const char* text)
{
    const std::string path = g_tmp + "/" + name + ".lp";
    std::ofstream(path) << text;
    return path;
}

struct outcome
{
    bx::raw_problem pb;
    bx::result r;
};

static void
report(const std::string& name, const bx::raw_problem& pb, const bx::result& r)
{
    const bool have = !r.solutions.empty();
    std::printf("case %s parsed=%d status=%d ok=%d valid=%d nsol=%zu value=%.17g recomputed=%.17g "
                "affected=%zu names=%zu remaining=%d variables=%d constraints=%d",
                name.c_str(), static_cast<int>(static_cast<bool>(pb)), static_cast<int>(r.status),
                static_cast<int>(static_cast<bool>(r)), static_cast<int>(bx::is_valid_solution(pb, r)),
                r.solutions.size(), have ? r.solutions.back().value : 0.0,
                (r && have) ? bx::compute_solution(pb, r) : 0.0, r.affected_vars.names.size(),
                r.variable_name.size(), r.remaining_constraints, r.variables, r.constraints);
    if (have) {
        std::printf(" x=");
        for (auto v : r.solutions.back().variables)
            std::putchar(v ? '1' : '0');
    }
    std::printf("\n");
}

static void
run(const std::string& name,
    const std::string& file,
    const std::function<void(bx::solver_parameters&)>& set,
    bool do_optimize = false,
    const std::function<void(bx::raw_problem&)>& edit = nullptr,
    const std::function<void(const bx::result&)>& extra = nullptr,
    const char* sweep_budget = nullptr)
{
    if (!g_only.empty() && name.compare(0, g_only.size(), g_only) != 0)
        return;
    auto ctx = bx::make_context(3);
    bx::solver_parameters params;
    set(params);
    bx::context_set_solver_parameters(ctx, params);
    if (sweep_budget)
        bx::context_set_parameters(ctx, "sweep-budget", sweep_budget);
    auto pb = bx::make_problem(ctx, file);
    if (edit)
        edit(pb);
    try {
        const auto r = do_optimize ? bx::optimize(ctx, pb) : bx::solve(ctx, pb);
        report(name, pb, r);
        if (extra)
            extra(r);
    } catch (const std::exception& e) {
        std::printf("case %s exception=%s\n", name.c_str(), e.what());
    }
    std::fflush(stdout);
}

int
main(int argc, char** argv)
{
    if (argc < 3) {
        std::fprintf(stderr, "usage: %s <lp dir> <scratch dir> [case-prefix]\n", argv[0]);
        return 2;
    }
    g_lp = argv[1];
    g_tmp = argv[2];
    if (argc > 3)
        g_only = argv[3];
    using P = bx::solver_parameters;
    const unsigned seeds[] = { 1u, 2u, 123456789u };

    const char* real_cost = "minimize\n- 0.1 a - 0.5 b - 0.9 c - 1e-7 d\nSubject to:\n-a -b -c <= -1\n"
                            "-a -b -c >= -3\n-a -c >= -2\n-a -c <= -1\na + c >= 1\n+ b + c +d >= 2\n"
                            "Binaries\na b c d\nEnd\n";
    const char* negative5 = "minimize\na b c d\nSubject to:\n-a -b -c <= -1\n-a -b -c >= -3\n-a -c >= -2\n"
                            "-a -c <= -1\na + c >= 1\n+ b + c +d >= 2\nBinaries\na b c d\nEnd\n";
    const char* quad = "minimize\n-5a + [ 2a * b + 3c * d ] /2 + 10b\nSubject to:\na + b >= 0\n"
                       "b + c + d >= 0\nBinaries\na b c d\nEnd\n";
    const char* z_a = "minimize\n15 x1 + 19 x2 + 13 x3 + 12 x4\nSubject to:\n2 x1 + 1 x2 + 3 x3 + 2 x4 <= 3\n"
                      "Binaries\nx1 x2 x3 x4\nEnd\n";
    const char* z_b = "minimize\nSubject to:\n2a + 3b -5c + 7d <= 0\n-2b + 2c >= 1\n7a + 7d <= 7\n"
                      "Binaries\na b c d\nEnd\n";

    for (unsigned seed : seeds) {
        const std::string s = "-s" + std::to_string(seed);
        auto defaults = [seed](P& p) { p.seed = seed; };

        // lib/test/solve.cpp:77-193
        run("prepro" + s, g_lp + "/prepro.lp", [seed](P& p) {
            p.seed = seed;
            p.cost_norm = P::cost_norm_type::loo;
        }, false, nullptr, [&](const bx::result& r) {
            std::printf("affected prepro%s", s.c_str());
            for (size_t i = 0; i != r.affected_vars.names.size(); ++i)
                std::printf(" %.*s=%d", static_cast<int>(r.affected_vars.names[i].size()),
                            r.affected_vars.names[i].data(), static_cast<int>(r.affected_vars.values[i]));
            std::printf("\n");
        });
        run("capmo1_direct" + s, g_lp + "/capmo1_direct.lp", defaults);          // :197-240
        run("quadratic" + s, inline_file("quadratic", quad), defaults);        // :242-259
        run("real_cost" + s, inline_file("real_cost", real_cost), defaults);    // :261-290
        run("assignment" + s, g_lp + "/assignment_problem_1.lp", defaults);     // :292-303
        run("negative-coeff" + s, g_lp + "/negative-coeff.lp", [seed](P& p) {   // :337-351
            p.seed = seed;
            p.limit = 50;
        });
        run("negative-coeff2" + s, g_lp + "/negative-coeff2.lp", defaults);     // :353-369
        run("negative-coeff3" + s, g_lp + "/negative-coeff3.lp", [seed](P& p) { // :371-386
            p.seed = seed;
            p.limit = 10000;
        });
        run("negative-coeff4" + s, g_lp + "/negative-coeff4.lp", [seed](P& p) { // :388-402
            p.seed = seed;
            p.limit = 50;
        });
        run("negative-coeff5" + s, inline_file("negative5", negative5), defaults); // :404-429
        auto queens = [seed](P& p) { // :437-445
            p.seed = seed;
            p.limit = -1;
            p.theta = 0.5;
            p.delta = 0.02;
            p.kappa_step = 0.01;
            p.kappa_max = 60.0;
            p.alpha = 1.0;
            p.w = 40;
        };
        run("queens_fixed" + s, g_lp + "/8_queens_puzzle.lp", queens, false, [](bx::raw_problem& pb) {
            static const int cost[] = { 25, 89, 12, 22, 84, 3,  61, 14, 93, 97, 68, 5,  51, 72, 96, 80,
                                        13, 38, 81, 48, 70, 50, 66, 68, 30, 97, 79, 4,  41, 44, 47, 62,
                                        60, 11, 18, 44, 57, 24, 7,  11, 66, 87, 9,  17, 27, 60, 95, 45,
                                        94, 47, 60, 87, 79, 53, 81, 52, 91, 53, 57, 8,  63, 78, 1,  8 };
            int i = 0;
            for (auto& e : pb.objective.elements)
                e.factor = cost[i++];
        }); // :431-474
        run("qap_solve" + s, g_lp + "/small4.lp", [seed](P& p) { // :510-529 (sweeps capped)
            p.seed = seed;
            p.limit = 20000;
            p.theta = 0.5;
            p.delta = 0.01;
            p.kappa_step = 10e-4;
            p.kappa_max = 10.0;
            p.alpha = 0.0;
            p.w = 20;
        });
        run("flat30-7" + s, g_lp + "/flat30-7.lp", [seed](P& p) { // :531-549
            p.seed = seed;
            p.limit = 200000; // (-1 in the reference: capped)
            p.delta = 0.001;
            p.kappa_min = 0.3;
            p.kappa_step = 1e-10;
            p.kappa_max = 1.0;
            p.order = P::constraint_order::reversing;
        });
        run("uf50-0448" + s, g_lp + "/uf50-0448.lp", [seed](P& p) { // :551-572
            p.seed = seed;
            p.limit = 200000;
            p.theta = 0.5;
            p.delta = 1.0;
            p.kappa_min = 0.1;
            p.kappa_step = 1e-17;
            p.kappa_max = 1.0;
            p.alpha = 2.0;
            p.w = 60;
            p.order = P::constraint_order::random_sorting;
        });
        run("aim-50" + s, g_lp + "/aim-50-1_6-yes1-2.lp", [seed](P& p) { // :574-593
            p.seed = seed;
            p.limit = 200000;
            p.theta = 0.6;
            p.delta = 0.01;
            p.kappa_step = 2 * 10e-4;
            p.kappa_max = 100.0;
            p.alpha = 1.0;
            p.w = 20;
        });
        run("z_coefficient_a" + s, inline_file("z_a", z_a), defaults); // :596-616
        run("z_coefficient_b" + s, inline_file("z_b", z_b), defaults); // :618-638
    }

    for (int draw = 0; draw != 10; ++draw) {
        const std::string d = "-d" + std::to_string(draw);
        auto redraw = [draw](bx::raw_problem& pb) {
            int i = 0;
            for (auto& e : pb.objective.elements)
                e.factor = drawn_cost(draw, i++);
        };
        run("assignment_random" + d, g_lp + "/assignment_problem_1.lp", [draw](P& p) { // :305-335
            p.seed = 1u + static_cast<unsigned>(draw);
            p.limit = 1000000;
            p.theta = 0.5;
            p.delta = 0.2;
            p.kappa_step = 10e-4;
            p.kappa_max = 10.0;
            p.alpha = 0.0;
            p.w = 20;
        }, false, redraw);
        run("queens_random" + d, g_lp + "/8_queens_puzzle.lp", [draw](P& p) { // :476-508
            p.seed = 1u + static_cast<unsigned>(draw);
            p.limit = -1;
            p.theta = 0.5;
            p.delta = 0.02;
            p.kappa_step = 0.01;
            p.kappa_max = 60.0;
            p.alpha = 1.0;
            p.w = 40;
            p.order = P::constraint_order::infeasibility_decr;
        }, false, redraw);
    }

    // lib/test/optimize.cpp:89-165: optimize() on the 30 n-queens files
    for (int i = 0; i != 30; ++i)
        run("n-queens-" + std::to_string(i), g_lp + "/n-queens/n-queens-problem-" + std::to_string(i) + ".lp",
            [i](P& p) {
                p.seed = 100u + static_cast<unsigned>(i);
                p.limit = 100000;
                p.theta = 0.5;
                p.delta = 1.0;
                p.kappa_min = 0.30;
                p.kappa_step = 1e-2;
                p.kappa_max = 100.0;
                p.alpha = 1.0;
                p.w = 60;
                p.time_limit = 20.0;
                p.pushing_k_factor = 0.9;
                p.pushes_limit = 50;
                p.pushing_objective_amplifier = 10;
                p.pushing_iteration_limit = 10;
                p.order = P::constraint_order::random_sorting;
                p.thread = 2;
            },
            true, nullptr, nullptr, "1500");

    // lib/test/optimize.cpp:55-87 (small4.lp, 2 threads; bounded by sweeps here)
    run("qap_optimize", g_lp + "/small4.lp", [](P& p) {
        p.seed = 1;
        p.limit = -1;
        p.theta = 0.5;
        p.delta = 0.2;
        p.kappa_step = 10e-4;
        p.kappa_max = 10.0;
        p.alpha = 0.0;
        p.w = 20;
        p.time_limit = 40.0;
        p.pushing_k_factor = 0.9;
        p.pushes_limit = 50;
        p.pushing_objective_amplifier = 10;
        p.pushing_iteration_limit = 50;
        p.thread = 2;
    }, true, nullptr, nullptr, "12000");

    std::printf("done\n");
    return 0;
}
