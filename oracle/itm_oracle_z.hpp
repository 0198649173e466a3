// oracle/itm_oracle_z.hpp -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
//
// CPU restatement of the general-integer-coefficient row update
// (lib/src/itm-solver-inequalities-Z.cpp) and of its two knapsack-row
// sub-solvers (lib/src/branch-and-bound-solver.hpp,
// lib/src/exhaustive-solver.hpp).  Included by itm_oracle.hpp.
//
// Pinned by the reference's own known-answer tests lib/test/lib.cpp:376-563
// (restated in tests/test_oracle_subsolvers.py).
#ifndef ORACLE_ITM_ORACLE_Z_HPP
#define ORACLE_ITM_ORACLE_Z_HPP

#include <map>

namespace orc {

// exhaustive_solver, exhaustive-solver.hpp:39-255
template<typename Float>
struct Exhaustive
{
    struct Item
    {
        Float r;
        int factor;
        int variable;
        int result;
    };
    struct Row
    {
        int start;
        int solutions;
    };

    std::vector<int> flat;  // feasible 0/factor patterns, row after row
    std::map<int, Row> rows; // constraint id -> patterns
    std::vector<Item> items;

    // build_constraints, :111-167.  The reference walks a binary counter whose
    // least significant digit is the LAST element, starting at 0..01, and it
    // also evaluates the intermediate states of every carry (so some patterns,
    // and the all-zero one, appear more than once); that order decides ties in
    // solve() and is kept.
    void build(int k, const int* factors, int size, int bk_min, int bk_max)
    {
        const int start = static_cast<int>(flat.size());
        int nb = 0;
        std::vector<std::int8_t> digit(static_cast<size_t>(size), 0);
        digit.back() = 1;

        auto visit = [&]() {
            int sum = 0;
            for (int j = 0; j != size; ++j)
                if (digit[j])
                    sum += factors[j];
            if (bk_min <= sum && sum <= bk_max) {
                ++nb;
                for (int j = 0; j != size; ++j)
                    flat.push_back(digit[j] ? factors[j] : 0);
            }
        };

        bool done = false;
        while (!done) {
            int pos = size - 1;
            for (;;) {
                visit();
                if (++digit[pos] <= 1)
                    break; // no carry: next outer step
                digit[pos] = 0;
                if (pos == 0) {
                    done = true;
                    break;
                }
                --pos; // carry: the state in between is visited too
            }
        }

        if (nb <= 0)
            throw std::logic_error("exhaustive: no feasible pattern");
        rows[k] = Row{ start, nb };
    }

    bool better(bool maximize, Float z, Float old_z) const noexcept
    {
        return maximize ? z > old_z : z < old_z;
    }

    // solve, :185-252
    int solve(bool maximize, int k, RC<Float>* R, int r_size)
    {
        const Row row = rows.at(k);
        items.resize(static_cast<size_t>(r_size));
        for (int i = 0; i != r_size; ++i)
            items[i] = Item{ R[i].value, R[i].f, R[i].id, 0 };

        Float z_best = 0;
        int best = 0;
        for (int j = 0; j != r_size; ++j)
            if (flat[row.start + j])
                z_best += R[j].value;

        for (int i = 1; i != row.solutions; ++i) {
            const int s = row.start + i * r_size;
            Float z = 0;
            for (int j = 0; j != r_size; ++j)
                if (flat[s + j])
                    z += R[j].value;
            if (better(maximize, z, z_best)) {
                z_best = z;
                best = i;
            }
        }

        const int s = row.start + best * r_size;
        for (int i = 0; i != r_size; ++i)
            items[i].result = flat[s + i] ? 1 : 0;

        std::sort(items.begin(),
                  items.end(),
                  [maximize](const Item& l, const Item& r) {
                      if (l.result == r.result)
                          return maximize ? l.r > r.r : l.r < r.r;
                      return l.result > r.result;
                  });

        auto middle = std::find_if(items.begin(), items.end(), [](const Item& it) {
            return it.result == 0;
        });

        for (int i = 0; i != r_size; ++i) {
            R[i].value = items[i].r;
            R[i].id = items[i].variable;
            R[i].f = items[i].factor;
        }

        if (middle == items.end())
            return items[0].result == 0 ? -1 : r_size;
        return static_cast<int>(middle - items.begin()) - 1;
    }
};

// branch_and_bound_solver, branch-and-bound-solver.hpp:195-536.  The
// reference keeps every node's 0/1 vector in a shared arena with a free list
// (:47-193); the arena only recycles memory, so plain per-node vectors give
// the same traversal.
template<typename Float>
struct BranchAndBound
{
    struct Item
    {
        Float r = 0;
        Float sum_z = 0;
        int factor = 0;
        int original_factor = 0;
        int variable = 0;
        int result = 0;
    };
    struct Node
    {
        std::vector<std::int8_t> vars;
        Float sumr;
        int sumfactor;
        unsigned level;
    };

    std::vector<Item> items;
    std::vector<Node> nodes;
    std::vector<std::int8_t> solution;
    Float lower_bound;
    int b_min, b_max;
    bool maximize;
    // SORT_CANONICAL: both sorts below are stable (ties keep their position),
    // the rule the HIP path implements; SORT_STD: std::sort as the reference
    int sort_mode = SORT_STD;

    template<typename It, typename Cmp>
    void do_sort(It b, It e, Cmp cmp)
    {
        if (sort_mode == SORT_STD)
            std::sort(b, e, cmp);
        else
            std::stable_sort(b, e, cmp);
    }

    Float init_z() const noexcept
    {
        return maximize ? -std::numeric_limits<Float>::infinity()
                        : +std::numeric_limits<Float>::infinity();
    }
    bool is_best(Float z) const noexcept
    {
        return maximize ? z > lower_bound : z < lower_bound;
    }
    unsigned esize() const noexcept
    {
        return static_cast<unsigned>(items.size());
    }

    void init_node_0() // :288-295
    {
        nodes.push_back(Node{ std::vector<std::int8_t>(items.size(), 0),
                              Float{ 0 },
                              0,
                              0u });
    }
    void init_node_1() // :297-305
    {
        nodes.push_back(Node{ std::vector<std::int8_t>(items.size(), 0),
                              items[0].r,
                              items[0].factor,
                              0u });
        nodes.back().vars[0] = 1;
    }

    // make_next_equality_node :307-344 / make_next_inequality_node :346-384
    void next_node(bool equality)
    {
        const unsigned level = nodes.back().level + 1;
        if (level >= esize()) {
            nodes.pop_back();
            return;
        }
        nodes.back().level = level;

        const Float sumr = nodes.back().sumr;
        const int sumfactor = nodes.back().sumfactor;
        std::vector<std::int8_t> vars = nodes.back().vars;
        vars[level] = 1;

        nodes.push_back(Node{ std::move(vars),
                              sumr + items[level].r,
                              sumfactor + items[level].factor,
                              level });
        const Node& n1 = nodes.back();

        if (equality) {
            if (n1.sumfactor == b_min) {
                if (is_best(n1.sumr)) {
                    solution = n1.vars;
                    lower_bound = n1.sumr;
                }
                nodes.pop_back();
            }
        } else {
            if (b_min <= n1.sumfactor && n1.sumfactor <= b_max) {
                if (is_best(n1.sumr)) {
                    solution = n1.vars;
                    lower_bound = n1.sumr;
                }
            } else if (n1.sumfactor > b_max) {
                nodes.pop_back();
            }
        }
    }

    void solve_equality() // :386-409
    {
        lower_bound = init_z();
        init_node_0();
        if (nodes.back().sumfactor == b_min) {
            solution = nodes.back().vars;
            return;
        }
        init_node_1();
        if (nodes.back().sumfactor == b_min) {
            solution = nodes.back().vars;
            nodes.pop_back();
        }
        while (!nodes.empty())
            next_node(true);
    }

    void solve_inequality() // :411-440
    {
        lower_bound = init_z();
        init_node_0();
        if (b_min <= nodes.back().sumfactor && nodes.back().sumfactor <= b_max) {
            solution = nodes.back().vars;
        } else if (nodes.back().sumfactor > b_max) {
            nodes.pop_back();
        }
        init_node_1();
        if (b_min <= nodes.back().sumfactor && nodes.back().sumfactor <= b_max) {
            solution = nodes.back().vars;
        } else if (nodes.back().sumfactor > b_max) {
            nodes.pop_back();
        }
        while (!nodes.empty())
            next_node(false);
    }

    // solve, :450-533
    int solve(bool maximize_, RC<Float>* R, int r_size, int bk_min, int bk_max)
    {
        maximize = maximize_;
        b_min = bk_min;
        b_max = bk_max;

        // shared_subvector::init fails when an element is shorter than its
        // free-list link (4 bytes) -> solver_failure(no_solver_available)
        if (r_size < 4)
            throw std::runtime_error("no_solver_available");

        items.assign(static_cast<size_t>(r_size), Item{});
        nodes.clear();
        for (int i = 0; i != r_size; ++i) {
            items[i].r = R[i].value;
            items[i].variable = R[i].id;
            items[i].original_factor = R[i].f;
            items[i].factor = std::abs(R[i].f);
            items[i].result = 0;
        }
        solution.assign(static_cast<size_t>(r_size), 0);

        // item::operator<, :212-220: both sides are divided by the LEFT
        // item's factor.
        if (maximize)
            do_sort(items.begin(), items.end(), [](const Item& a, const Item& o) {
                return (o.r / static_cast<Float>(a.factor)) <
                       (a.r / static_cast<Float>(a.factor));
            });
        else
            do_sort(items.begin(), items.end(), [](const Item& a, const Item& o) {
                return (a.r / static_cast<Float>(a.factor)) <
                       (o.r / static_cast<Float>(a.factor));
            });

        for (const auto& it : items)
            if (it.original_factor < 0) {
                b_min += it.factor;
                b_max += it.factor;
            }

        items.back().sum_z = items.back().r;
        for (int i = r_size - 2; i >= 0; --i)
            items[i].sum_z = items[i + 1].sum_z + items[i].r;

        if (b_max == b_min)
            solve_equality();
        else
            solve_inequality();

        for (int i = 0; i != r_size; ++i) {
            if (items[i].original_factor > 0)
                items[i].result = solution[i] == 0 ? 0 : 1;
            else
                items[i].result = solution[i] == 0 ? 1 : 0;
        }

        do_sort(items.begin(),
                  items.end(),
                  [maximize_](const Item& l, const Item& r) {
                      if (l.result == r.result)
                          return maximize_ ? l.r > r.r : l.r < r.r;
                      return l.result > r.result;
                  });

        auto middle = std::find_if(items.begin(), items.end(), [](const Item& it) {
            return it.result == 0;
        });

        for (int i = 0; i != r_size; ++i) {
            R[i].value = items[i].r;
            R[i].id = items[i].variable;
            R[i].f = items[i].original_factor;
        }

        if (middle == items.end())
            return items[0].result == 0 ? -1 : r_size;
        return static_cast<int>(middle - items.begin()) - 1;
    }
};

template<typename Float>
struct ZSubSolvers
{
    static constexpr int maximum_factor_exhaustive_solver = 10; // Z.cpp:39
    BranchAndBound<Float> bb;
    Exhaustive<Float> ex;

    // Sub-solver choice per row, inequalities-Z.cpp:125-187.
    explicit ZSubSolvers(Solver<Float>& s)
    {
        bb.sort_mode = s.sort_mode;
        const auto& cs = s.cs;
        s.Z.assign(static_cast<size_t>(s.m), Z_LINEAR);
        for (int k = 0; k != s.m; ++k) {
            int lower = 0, upper = 0;
            const int len = cs.row_len(k);
            char z = Z_LINEAR;
            for (int e = cs.row_ptr[k], ee = cs.row_ptr[k + 1]; e != ee; ++e) {
                const int f = cs.factor[e];
                if (f == 0)
                    throw std::invalid_argument("zero coefficient");
                if (f > 0)
                    upper += f;
                else
                    lower += f;
                if (f < -1 || f > 1)
                    z = Z_BB;
            }
            if (z == Z_BB && len < maximum_factor_exhaustive_solver)
                z = Z_EXHAUSTIVE;
            if (cs.cmin[k] != cs.cmax[k]) {
                if (lower >= cs.cmin[k] || upper <= cs.cmax[k]) {
                    if (z == Z_BB)
                        z = Z_UNCONSTRAINED_BB;
                    else if (z == Z_EXHAUSTIVE)
                        z = Z_UNCONSTRAINED_EXHAUSTIVE;
                }
            }
            s.Z[k] = z;
            if (z == Z_EXHAUSTIVE || z == Z_UNCONSTRAINED_EXHAUSTIVE)
                ex.build(k,
                         cs.factor.data() + cs.row_ptr[k],
                         len,
                         s.b[k].min,
                         s.b[k].max);
        }
    }
};

// select_variables / select_variables_101, inequalities-Z.cpp:308-344
// (identical bodies).  Reads R[i + 1] up to R[r_size].
template<typename Float>
inline int
z_select_variables(const Solver<Float>& s, int r_size, int bkmin, int bkmax)
{
    int sum = 0;
    int best = -2;
    for (int i = -1; i < r_size; ++i) {
        if (bkmin <= sum && sum <= bkmax)
            best = i;
        if (best != -2 && i - 1 < r_size &&
            stop_iterating(s.maximize, s.R[i + 1].value))
            break;
        sum += s.R[i + 1].f;
    }
    return best;
}

// Sub-solver dispatch (inequalities-Z.cpp:471-499 / :529-556) followed by
// local_affect (:346-439).
template<typename Float>
bool
Solver<Float>::z_update_row(BitArray& x,
                            int k,
                            int r_size,
                            Float kappa,
                            Float delta)
{
    int selected = -2;
    rc* Rp = R.data();

    switch (Z[k]) {
    case Z_BB:
        calculator_sort(Rp, Rp + r_size);
        selected = zsub->bb.solve(maximize, Rp, r_size, b[k].min, b[k].max);
        break;
    case Z_EXHAUSTIVE:
        selected = zsub->ex.solve(maximize, k, Rp, r_size);
        break;
    case Z_UNCONSTRAINED_BB:
        calculator_sort(Rp, Rp + r_size);
        selected = z_select_variables(*this, r_size, b[k].min, b[k].max);
        if (selected == -2)
            selected = zsub->bb.solve(maximize, Rp, r_size, b[k].min, b[k].max);
        break;
    case Z_UNCONSTRAINED_EXHAUSTIVE:
        calculator_sort(Rp, Rp + r_size);
        selected = z_select_variables(*this, r_size, b[k].min, b[k].max);
        if (selected == -2)
            selected = zsub->ex.solve(maximize, k, Rp, r_size);
        break;
    default:
        calculator_sort(Rp, Rp + r_size);
        selected = z_select_variables(*this, r_size, b[k].min, b[k].max);
        break;
    }

    constexpr Float one{ 1 };
    constexpr Float two{ 2 };
    constexpr Float middle{ (two + one) / two };

    const int rs = cs.row_ptr[k];
    const auto old_pi = pi[k];
    const auto d = (kappa / (one - kappa)) + delta;

    auto to_zero = [&](int i) {
        const int e = rs + R[i].id;
        x.unset(cs.col[e]);
        P[e] -= d;
        const auto repair = local_compute_reduced_cost(cs.col[e], x);
        if (repair <= 0)
            P[e] = P[e] + repair - d;
    };
    auto to_one = [&](int i) {
        const int e = rs + R[i].id;
        x.set(cs.col[e]);
        P[e] += d;
        const auto repair = local_compute_reduced_cost(cs.col[e], x);
        if (repair >= 0)
            P[e] = P[e] - repair + d;
    };

    if (selected < 0) {
        pi[k] += R[0].value / two;
        for (int i = 0; i != r_size; ++i)
            to_zero(i);
    } else if (selected + 1 >= r_size) {
        pi[k] += R[selected].value * middle;
        for (int i = 0; i != r_size; ++i)
            to_one(i);
    } else {
        pi[k] += ((R[selected].value + R[selected + 1].value) / two);
        int i = 0;
        for (; i <= selected; ++i)
            to_one(i);
        for (; i != r_size; ++i)
            to_zero(i);
    }

    return std::signbit(old_pi) != std::signbit(pi[k]);
}

} // namespace orc

#endif
