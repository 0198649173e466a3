// baryonyx_amd/host/preprocessor.cpp -- unit propagation ahead of the solver:
// the behaviour of lib/src/preprocessor.cpp (preprocess :903-913, affect
// :883-901, split :866-881), written for this library's own data structures.
//
// What it does, in the reference's order:
//   1. variables whose bounds coincide are fixed (try_affect_bounded_variable,
//      :443-455);
//   2. every constraint is looked at once (try_affect_variable, :457-556): a
//      single-element constraint forces (or frees) its variable; constraints
//      with only positive coefficients whose right-hand side is 0 or the sum of
//      the coefficients fix all their variables or are dropped;
//   3. fixed variables are propagated depth first (affects, :366-434): a
//      constraint left with one free variable forces it and disappears;
//   4. variables no live constraint mentions take the value the objective
//      prefers, `true` when the objective does not mention them either
//      (try_remove_unused_variable, :337-364);
//   5. the reduced problem is rebuilt with the free variables renumbered in
//      order (make_problem, :651-727).
// Pending assignments sit on a stack that refuses a variable already waiting
// on it (pp_lifo, :94-133), so of two conflicting pending values the first
// pushed wins.
//
// Quadratic objectives: fixed variables are folded into linear terms and the
// constant as objective_function_exclude_copy does (:729-785).  The
// reference's make_lp_from_bqp_objective (:281-331) only contributes its
// constant part here: the loop that should merge its linear part (:772-782)
// indexes the wrong container and dereferences end(), i.e. has no defined
// result to reproduce.
#include "private.hpp"

#include <algorithm>
#include <stdexcept>
#include <tuple>

namespace baryonyx {

namespace {

enum cst_class
{
    EQ = 0,
    GE = 1,
    LE = 2
};

const affected_variables*
already_affected(const raw_problem&)
{
    return nullptr;
}
const affected_variables*
already_affected(const problem& pb)
{
    return &pb.affected_vars;
}

template<typename Pb>
class propagation
{
    const Pb& pb;
    const std::vector<constraint>* rows[3];
    std::vector<signed char> fixed;        // -1 free, else the value
    std::vector<int> live[3];              // free elements left; 0 = dropped
    std::vector<std::vector<int>> uses[3]; // per variable: rows mentioning it
    std::vector<std::pair<int, bool>> pending;
    std::vector<char> is_pending;
    double extra_constant = 0;

    bool push(int var, bool value)
    {
        if (is_pending[static_cast<size_t>(var)])
            return false;
        is_pending[static_cast<size_t>(var)] = 1;
        pending.emplace_back(var, value);
        return true;
    }

    // the one free variable of a row and what is left of its right-hand side
    std::tuple<int, int, int> last_free(const constraint& c) const
    {
        int rhs = c.value, at = -1;
        for (int i = 0, e = static_cast<int>(c.elements.size()); i != e; ++i) {
            const auto v = fixed[static_cast<size_t>(c.elements[i].variable_index)];
            if (v < 0)
                at = i;
            else
                rhs -= c.elements[i].factor * v;
        }
        if (at < 0)
            return { -1, -1, rhs };
        return { c.elements[at].factor, c.elements[at].variable_index, rhs };
    }

    // a row with one free variable: force it when only one value satisfies it
    void close_row(int cls, int id)
    {
        const auto [factor, var, rhs] = last_free((*rows[cls])[static_cast<size_t>(id)]);
        live[cls][static_cast<size_t>(id)] = 0;
        if (var < 0)
            return;
        auto holds = [&](int lhs) {
            return cls == EQ ? lhs == rhs : (cls == GE ? lhs >= rhs : lhs <= rhs);
        };
        const bool zero_ok = holds(0), one_ok = holds(factor);
        if (zero_ok && one_ok)
            return;
        if (!zero_ok && !one_ok) // bx_reach() in the reference
            throw solver_failure(solver_error_tag::unrealisable_constraint);
        push(var, one_ok);
    }

    void propagate()
    {
        while (!pending.empty()) {
            const auto [var, value] = pending.back();
            pending.pop_back();
            is_pending[static_cast<size_t>(var)] = 0;
            if (fixed[static_cast<size_t>(var)] < 0)
                fixed[static_cast<size_t>(var)] = value ? 1 : 0;
            for (int cls = 0; cls != 3; ++cls)
                for (int id : uses[cls][static_cast<size_t>(var)]) {
                    int& n = live[cls][static_cast<size_t>(id)];
                    if (n <= 0)
                        continue;
                    if (--n == 1)
                        close_row(cls, id);
                }
        }
    }

    void reset()
    {
        std::fill(fixed.begin(), fixed.end(), static_cast<signed char>(-1));
        std::fill(is_pending.begin(), is_pending.end(), 0);
        pending.clear();
        extra_constant = 0;
        for (int cls = 0; cls != 3; ++cls)
            for (size_t i = 0; i != rows[cls]->size(); ++i)
                live[cls][i] = static_cast<int>((*rows[cls])[i].elements.size());
    }

    void fix_bounded_variables()
    {
        for (int i = 0, e = static_cast<int>(pb.vars.values.size()); i != e; ++i)
            if (pb.vars.values[static_cast<size_t>(i)].min == pb.vars.values[static_cast<size_t>(i)].max)
                push(i, pb.vars.values[static_cast<size_t>(i)].max != 0);
    }

    void first_look()
    {
        for (int cls = 0; cls != 3; ++cls)
            for (int id = 0, e = static_cast<int>(rows[cls]->size()); id != e; ++id) {
                const constraint& c = (*rows[cls])[static_cast<size_t>(id)];
                if (live[cls][static_cast<size_t>(id)] == 1) {
                    close_row(cls, id);
                    continue;
                }
                int total = 0;
                bool positive = true;
                for (const auto& el : c.elements) {
                    positive = positive && el.factor > 0;
                    total += el.factor;
                }
                if (!positive)
                    continue;
                auto fix_all = [&](bool value) {
                    live[cls][static_cast<size_t>(id)] = 0;
                    for (const auto& el : c.elements)
                        push(el.variable_index, value);
                };
                if (cls == EQ) {
                    if (total == c.value || c.value == 0)
                        fix_all(c.value != 0);
                } else if (cls == GE) {
                    if (total == c.value)
                        fix_all(true);
                    else if (c.value == 0)
                        live[cls][static_cast<size_t>(id)] = 0;
                } else {
                    if (c.value <= 0)
                        fix_all(false);
                    else if (total == c.value)
                        live[cls][static_cast<size_t>(id)] = 0;
                }
            }
    }

    bool unused(int var) const
    {
        for (int cls = 0; cls != 3; ++cls)
            for (int id : uses[cls][static_cast<size_t>(var)])
                if (live[cls][static_cast<size_t>(id)] > 0)
                    return false;
        return true;
    }

    // constant part of make_lp_from_bqp_objective (see the file header)
    void quadratic_constant()
    {
        const bool maximize = pb.type == objective_function_type::maximize;
        for (int i = 0, e = static_cast<int>(fixed.size()); i != e; ++i) {
            if (fixed[static_cast<size_t>(i)] >= 0 || !unused(i))
                continue;
            for (const auto& q : pb.objective.qelements) {
                if (q.variable_index_a != i && q.variable_index_b != i)
                    continue;
                const bool value = maximize ? q.factor > 0 : q.factor < 0;
                if (value && q.variable_index_a == q.variable_index_b)
                    extra_constant += q.factor / 2;
            }
        }
    }

    void settle_unused_variables()
    {
        const bool maximize = pb.type == objective_function_type::maximize;
        for (int i = 0, e = static_cast<int>(fixed.size()); i != e; ++i) {
            if (fixed[static_cast<size_t>(i)] >= 0 || !unused(i))
                continue;
            bool value = true;
            for (const auto& el : pb.objective.elements)
                if (el.variable_index == i) {
                    value = maximize ? el.factor > 0 : el.factor < 0;
                    break;
                }
            fixed[static_cast<size_t>(i)] = value ? 1 : 0;
        }
    }

    problem rebuild() const
    {
        problem out;
        out.strings = pb.strings;
        out.type = pb.type;

        const int n = static_cast<int>(fixed.size());
        std::vector<int> renum(static_cast<size_t>(n), -1);
        int next = 0;
        for (int i = 0; i != n; ++i)
            if (fixed[static_cast<size_t>(i)] < 0)
                renum[static_cast<size_t>(i)] = next++;
        auto value_of = [&](int i) { return fixed[static_cast<size_t>(i)] > 0; };

        out.objective.value = pb.objective.value + extra_constant;
        for (const auto& el : pb.objective.elements) {
            const int to = renum[static_cast<size_t>(el.variable_index)];
            if (to >= 0)
                out.objective.elements.emplace_back(el.factor, to);
            else if (value_of(el.variable_index))
                out.objective.value += el.factor;
        }
        for (const auto& q : pb.objective.qelements) {
            const int a = renum[static_cast<size_t>(q.variable_index_a)];
            const int b = renum[static_cast<size_t>(q.variable_index_b)];
            if (a < 0 && b < 0) {
                if (value_of(q.variable_index_a) && value_of(q.variable_index_b))
                    out.objective.value += q.factor;
            } else if (a < 0) { // (the reference does not look at the fixed value here)
                out.objective.elements.emplace_back(q.factor, b);
            } else if (b < 0) {
                out.objective.elements.emplace_back(q.factor, a);
            } else {
                out.objective.qelements.emplace_back(q.factor, a, b);
            }
        }

        if (const affected_variables* before = already_affected(pb))
            out.affected_vars = *before;
        for (int i = 0; i != n; ++i) {
            if (fixed[static_cast<size_t>(i)] < 0) {
                out.vars.names.emplace_back(pb.vars.names[static_cast<size_t>(i)]);
                out.vars.values.emplace_back(pb.vars.values[static_cast<size_t>(i)]);
            } else {
                out.affected_vars.push_back(pb.vars.names[static_cast<size_t>(i)], value_of(i));
            }
        }

        std::vector<constraint>* dst[3] = { &out.equal_constraints, &out.greater_constraints,
                                            &out.less_constraints };
        for (int cls = 0; cls != 3; ++cls)
            for (size_t id = 0; id != rows[cls]->size(); ++id) {
                if (live[cls][id] <= 1) // dropped, or one free variable left: no information
                    continue;
                const constraint& c = (*rows[cls])[id];
                constraint k;
                k.id = c.id;
                k.label = c.label;
                k.value = c.value;
                for (const auto& el : c.elements) {
                    const int to = renum[static_cast<size_t>(el.variable_index)];
                    if (to >= 0)
                        k.elements.emplace_back(el.factor, to);
                    else if (value_of(el.variable_index))
                        k.value -= el.factor;
                }
                dst[cls]->emplace_back(std::move(k));
            }
        out.problem_type = out.which_problem_type();
        return out;
    }

public:
    explicit propagation(const Pb& pb_)
      : pb(pb_)
      , rows{ &pb_.equal_constraints, &pb_.greater_constraints, &pb_.less_constraints }
      , fixed(pb_.vars.values.size(), static_cast<signed char>(-1))
      , is_pending(pb_.vars.values.size(), 0)
    {
        for (int cls = 0; cls != 3; ++cls) {
            live[cls].resize(rows[cls]->size());
            uses[cls].resize(pb.vars.values.size());
            for (int id = 0, e = static_cast<int>(rows[cls]->size()); id != e; ++id)
                for (const auto& el : (*rows[cls])[static_cast<size_t>(id)].elements)
                    uses[cls][static_cast<size_t>(el.variable_index)].push_back(id);
        }
    }

    problem whole() // preprocessor::operator()(), :617-640
    {
        reset();
        fix_bounded_variables();
        first_look();
        propagate();
        quadratic_constant();
        settle_unused_variables();
        propagate();
        return rebuild();
    }

    problem with(int var, bool value) // preprocessor::operator()(int, bool), :594-615
    {
        reset();
        push(var, value);
        propagate();
        quadratic_constant();
        settle_unused_variables();
        propagate();
        return rebuild();
    }
};

} // namespace

problem
preprocess(const context& /*ctx*/, const raw_problem& pb)
{
    return propagation<raw_problem>(pb).whole();
}

problem
affect(const context& /*ctx*/, const problem& pb, int variable_index, bool variable_value)
{
    if (variable_index < 0 || variable_index >= static_cast<int>(pb.vars.values.size()))
        throw std::out_of_range("affect: variable index");
    return propagation<problem>(pb).with(variable_index, variable_value);
}

std::tuple<problem, problem>
split(const context& /*ctx*/, const problem& pb, int variable_index_to_affect)
{
    if (variable_index_to_affect < 0 ||
        variable_index_to_affect >= static_cast<int>(pb.vars.values.size()))
        throw std::out_of_range("split: variable index");
    propagation<problem> pp(pb);
    return { pp.with(variable_index_to_affect, true), pp.with(variable_index_to_affect, false) };
}

} // namespace baryonyx
