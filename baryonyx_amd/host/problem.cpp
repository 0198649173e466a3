// baryonyx_amd/host/problem.cpp -- problem classification, merged
// constraints and solution checks (host side of the path).
//   problem / which_problem_type : lib/src/problem.hpp:44-134
//   make_merged_constraints      : lib/src/merged-constraint.cpp:37-343
//   is_valid_solution / compute_solution : lib/src/lpcore.cpp:134-257
#include "private.hpp"

#include <algorithm>
#include <climits>
#include <unordered_map>

namespace baryonyx {

namespace {

int
coefficient_type(const std::vector<constraint>& csts) noexcept
{
    int ret = 0;
    for (const auto& cst : csts)
        for (const auto& elem : cst.elements) {
            if (elem.factor == -1 && ret == 0)
                ret = 1;
            else if (elem.factor <= -2 || elem.factor >= 2)
                return 2;
        }
    return ret;
}

struct row_hash // merged-constraint.cpp:37-50
{
    size_t operator()(const std::vector<function_element>& fct) const noexcept
    {
        std::size_t seed{ fct.size() };
        for (auto& elem : fct)
            seed ^= elem.variable_index + 0x9e3779b9 + (seed << 6) + (seed >> 2);
        return seed;
    }
};

struct row_equal
{
    bool operator()(const std::vector<function_element>& a,
                    const std::vector<function_element>& b) const noexcept
    {
        if (a.size() != b.size())
            return false;
        for (size_t i = 0; i != a.size(); ++i)
            if (a[i].factor != b[i].factor || a[i].variable_index != b[i].variable_index)
                return false;
        return true;
    }
};

using row_cache = std::unordered_map<std::vector<function_element>, std::size_t, row_hash, row_equal>;

void
fill(row_cache& cache, operator_type op, const problem& pb, std::vector<merged_constraint>& ret)
{
    switch (op) {
    case operator_type::equal:
        for (const auto& elem : pb.equal_constraints) {
            auto it = cache.find(elem.elements);
            if (it == cache.end()) {
                cache.emplace(elem.elements, ret.size());
                ret.push_back({ elem.elements, elem.value, elem.value, elem.id });
            } else if (ret[it->second].min <= elem.value && elem.value <= ret[it->second].max) {
                ret[it->second].min = elem.value;
                ret[it->second].max = elem.value;
            } else {
                throw problem_definition_failure("equal", problem_definition_error_tag::multiple_constraint);
            }
        }
        break;
    case operator_type::less:
        for (const auto& elem : pb.less_constraints) {
            auto it = cache.find(elem.elements);
            if (it == cache.end()) {
                cache.emplace(elem.elements, ret.size());
                ret.push_back({ elem.elements, std::numeric_limits<int>::min(), elem.value, elem.id });
            } else {
                ret[it->second].max = std::min(ret[it->second].max, elem.value);
            }
        }
        break;
    case operator_type::greater:
        for (const auto& elem : pb.greater_constraints) {
            auto it = cache.find(elem.elements);
            if (it == cache.end()) {
                cache.emplace(elem.elements, ret.size());
                ret.push_back({ elem.elements, elem.value, std::numeric_limits<int>::max(), elem.id });
            } else {
                ret[it->second].min = std::max(ret[it->second].min, elem.value);
            }
        }
        break;
    }
}

int
compute_function(const std::vector<function_element>& fct, const std::vector<var_value>& x)
{
    int v = 0;
    for (const auto& f : fct)
        v += f.factor * x[f.variable_index];
    return v;
}

std::vector<var_value>
variable_value_of(const raw_problem& pb, const result& r)
{
    if (!r || r.solutions.empty())
        return {};
    std::unordered_map<std::string_view, bool> cache;
    for (size_t i = 0, e = r.affected_vars.names.size(); i != e; ++i)
        cache[r.affected_vars.names[i]] = r.affected_vars.values[i];
    for (size_t i = 0, e = r.variable_name.size(); i != e; ++i)
        cache[r.variable_name[i]] = r.solutions.back().variables[i];
    std::vector<var_value> ret(pb.vars.names.size(), false);
    for (size_t i = 0, e = pb.vars.names.size(); i != e; ++i)
        ret[i] = cache[pb.vars.names[i]];
    return ret;
}

} // namespace

problem::problem(const raw_problem& pb)
  : strings(pb.strings)
  , objective(pb.objective)
  , equal_constraints(pb.equal_constraints)
  , greater_constraints(pb.greater_constraints)
  , less_constraints(pb.less_constraints)
  , vars(pb.vars)
  , type(pb.type)
{
    problem_type = which_problem_type();
}

problem_solver_type
problem::which_problem_type() const noexcept
{
    const int c = std::max(coefficient_type(equal_constraints),
                           std::max(coefficient_type(less_constraints),
                                    coefficient_type(greater_constraints)));
    if (greater_constraints.empty() && less_constraints.empty())
        return c == 0 ? problem_solver_type::equalities_01
                      : (c == 1 ? problem_solver_type::equalities_101
                                : problem_solver_type::equalities_Z);
    return c == 0 ? problem_solver_type::inequalities_01
                  : (c == 1 ? problem_solver_type::inequalities_101
                            : problem_solver_type::inequalities_Z);
}

problem
unpreprocess(const context& /*ctx*/, const raw_problem& pb)
{
    return problem(pb);
}

std::vector<merged_constraint>
make_merged_constraints(const context& ctx, const problem& pb)
{
    using po = solver_parameters::pre_constraint_order;
    row_cache cache;
    std::vector<merged_constraint> ret;
    ret.reserve(pb.equal_constraints.size() + pb.less_constraints.size() + pb.greater_constraints.size());

    const operator_type L = operator_type::less, G = operator_type::greater, E = operator_type::equal;
    auto ordered = [&](operator_type a, operator_type b, operator_type c) {
        fill(cache, a, pb, ret);
        fill(cache, b, pb, ret);
        fill(cache, c, pb, ret);
    };
    auto memory = [&]() { // improve_memory_usage, merged-constraint.cpp:286-295
        for (auto& cst : ret)
            std::sort(cst.elements.begin(), cst.elements.end(), [](const auto& l, const auto& r) {
                return l.variable_index < r.variable_index;
            });
    };

    switch (ctx.parameters.pre_order) {
    case po::less_greater_equal:
        ordered(L, G, E);
        memory();
        break;
    case po::less_equal_greater:
        ordered(L, E, G);
        memory();
        break;
    case po::greater_less_equal:
        ordered(G, L, E);
        memory();
        break;
    case po::greater_equal_less:
        ordered(G, E, L);
        memory();
        break;
    case po::equal_less_greater:
        ordered(E, L, G);
        memory();
        break;
    case po::equal_greater_less:
        ordered(E, G, L);
        memory();
        break;
    default: // none, memory, p1 .. p4 start from the file order
        ordered(E, L, G);
        std::sort(ret.begin(), ret.end(), [](const auto& l, const auto& r) { return l.id < r.id; });
        if (ctx.parameters.pre_order == po::memory)
            memory();
        break;
    }

    // make_special_merged_constraints, merged-constraint.cpp:211-284: the rows take
    // the rank of their cheapest variable by cost / degree (p1: dearest first; p2 - p4
    // all take the other branch there: cheapest first), the elements of a row go by
    // falling cost / degree.  The same std::sort calls on the same sequences, so that
    // ties fall as libstdc++ lets them fall in the reference.
    if (ctx.parameters.pre_order == po::p1 || ctx.parameters.pre_order == po::p2 ||
        ctx.parameters.pre_order == po::p3 || ctx.parameters.pre_order == po::p4) {
        std::vector<int> degree(pb.vars.values.size(), 0);
        for (const auto& cst : ret)
            for (const auto& elem : cst.elements)
                ++degree[elem.variable_index];
        std::vector<double> variable_cost(pb.vars.values.size(), 0.0);
        for (const auto& elem : pb.objective.elements)
            variable_cost[elem.variable_index] = elem.factor / static_cast<double>(degree[elem.variable_index]);
        std::vector<std::pair<int, double>> ratio_min(ret.size());
        for (int i = 0, ie = static_cast<int>(ret.size()); i != ie; ++i) {
            ratio_min[i] = { i, variable_cost[ret[i].elements[0].variable_index] };
            for (int j = 1, ej = static_cast<int>(ret[i].elements.size()); j != ej; ++j)
                ratio_min[i].second = std::min(ratio_min[i].second, variable_cost[ret[i].elements[j].variable_index]);
        }
        if (ctx.parameters.pre_order == po::p1)
            std::sort(ratio_min.begin(), ratio_min.end(), [](const auto& l, const auto& r) { return l.second > r.second; });
        else
            std::sort(ratio_min.begin(), ratio_min.end(), [](const auto& l, const auto& r) { return l.second < r.second; });
        for (int i = 0, e = static_cast<int>(ret.size()); i != e; ++i)
            ret[i].id = ratio_min[i].first;
        std::sort(ret.begin(), ret.end(), [](const auto& l, const auto& r) { return l.id < r.id; });
        for (auto& cst : ret)
            std::sort(cst.elements.begin(), cst.elements.end(), [&variable_cost](const auto& l, const auto& r) {
                return variable_cost[l.variable_index] > variable_cost[r.variable_index];
            });
    }
    return ret;
}

bool
is_valid_solution(const problem& pb, const std::vector<var_value>& x)
{
    for (const auto& c : pb.equal_constraints)
        if (compute_function(c.elements, x) != c.value)
            return false;
    for (const auto& c : pb.less_constraints)
        if (compute_function(c.elements, x) > c.value)
            return false;
    for (const auto& c : pb.greater_constraints)
        if (compute_function(c.elements, x) < c.value)
            return false;
    return true;
}

double
compute_solution(const problem& pb, const std::vector<var_value>& x)
{
    double ret = pb.objective.value;
    for (auto& elem : pb.objective.elements)
        ret += elem.factor * x[elem.variable_index];
    for (auto& elem : pb.objective.qelements)
        ret += elem.factor * x[elem.variable_index_a] * x[elem.variable_index_b];
    return ret;
}

bool
is_valid_solution(const raw_problem& pb, const result& r)
{
    if (!r || r.solutions.empty())
        return false;
    return is_valid_solution(problem(pb), variable_value_of(pb, r));
}

double
compute_solution(const raw_problem& pb, const result& r)
{
    return compute_solution(problem(pb), variable_value_of(pb, r));
}

} // namespace baryonyx
