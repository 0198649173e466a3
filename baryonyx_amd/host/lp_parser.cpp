// baryonyx_amd/host/lp_parser.cpp -- the `.lp` reader, with the grammar the reference
// accepts and its readings of it (lib/src/parser.cpp:268-1257): the tokens of
// stream_buffer::next_token (:385-447), read_real (:565-611), a function element = [sign]
// [number] [name] (:789-822), the `[ ... ] / 2` quadratic block (:692-786), `subject to`
// constraints `[label:] elements op [sign] number` (:1115-1210), then `bounds` (:907-965),
// `binary`, `general`, `end`.  Variables are numbered by first appearance; a variable named
// twice in one function adds up.  Round 5: token for token the reference's (checked against
// an independent Python restatement on every fixture and on generated texts,
// tests/test_host_parser_grammar.py), quirks included -- a number token runs over `e` / `E`
// (`3ex` is 3 times `x`), a second number in a row is a NAME, a constraint element without
// a name names the empty variable, the operator of a bound line is not looked at.
#include "private.hpp"

#include <algorithm>
#include <cctype>
#include <climits>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <initializer_list>
#include <iterator>
#include <optional>
#include <unordered_map>

namespace baryonyx {
namespace {

bool
ieq(std::string_view a, const char* b)
{
    size_t i = 0;
    for (; i < a.size() && b[i]; ++i)
        if (std::tolower(static_cast<unsigned char>(a[i])) != b[i])
            return false;
    return i == a.size() && b[i] == '\0';
}

// lib/src/parser.cpp:131-150 (is_separator)
bool
is_sep(char c)
{
    return c == '<' || c == '=' || c == '>' || c == ':' || c == '-' || c == '+' || c == '[' ||
           c == ']' || c == '*' || c == '^';
}

bool
number_char(char c) // :103-108 (is_number)
{
    return (c >= '0' && c <= '9') || c == '.' || c == 'e' || c == 'E' || c == '-' || c == '+';
}

// Token stream of stream_buffer::next_token (:385-447): words between white space; a word
// that STARTS with a backslash drops the rest of its line; inside a word every separator is a
// token of its own, and a token that starts with a digit, `.`, `-` or `+` ends in front of the
// first character that is not one of 0-9 . e E - +.
struct lexer
{
    std::vector<std::string> tok;
    size_t pos = 0;
    bool overlong = false; // a word beyond the reference's 512-byte token buffer

    explicit lexer(std::istream& is)
    {
        std::string text((std::istreambuf_iterator<char>(is)), std::istreambuf_iterator<char>());
        const size_t n = text.size();
        size_t i = 0;
        auto space = [](char c) {
            return c == ' ' || c == '\t' || c == '\n' || c == '\v' || c == '\f' || c == '\r';
        };
        while (i < n) {
            while (i < n && space(text[i]))
                ++i;
            if (i >= n)
                break;
            size_t j = i;
            while (j < n && !space(text[j]))
                ++j;
            if (text[i] == '\\') { // comment
                while (j < n && text[j] != '\n')
                    ++j;
                i = j;
                continue;
            }
            if (j - i > 511)
                overlong = true;
            size_t cur = i;
            while (cur < j) {
                const char c0 = text[cur];
                const bool num = (c0 >= '0' && c0 <= '9') || c0 == '.' || c0 == '-' || c0 == '+';
                const size_t s0 = cur++;
                if (is_sep(c0) || cur == j) {
                    tok.emplace_back(1, c0);
                    continue;
                }
                while (cur < j && !is_sep(text[cur]) && !(num && !number_char(text[cur])))
                    ++cur;
                tok.emplace_back(text.substr(s0, cur - s0));
            }
            i = j;
        }
    }

    std::string_view peek(size_t k = 0) const
    {
        return pos + k < tok.size() ? std::string_view(tok[pos + k])
                                    : std::string_view();
    }
    void pop(size_t k = 1) { pos += k; }
    bool end() const { return pos >= tok.size(); }
};

struct builder
{
    raw_problem pb;
    std::unordered_map<std::string, int> cache;

    builder() { pb.strings = std::make_shared<string_buffer>(); }

    int variable(std::string_view name)
    {
        auto it = cache.find(std::string(name));
        if (it != cache.end())
            return it->second;
        const int id = static_cast<int>(cache.size());
        cache.emplace(std::string(name), id);
        pb.vars.names.emplace_back(pb.strings->append(name));
        pb.vars.values.emplace_back(0, std::numeric_limits<int>::max(), variable_type::real);
        return id;
    }
    int find(std::string_view name) const
    {
        auto it = cache.find(std::string(name));
        return it == cache.end() ? -1 : it->second;
    }
};

// static_cast<int>(double) as the reference's build performs it (x86-64
// cvttsd2si: anything outside int's range, NaN included, gives INT_MIN).
int
to_int(double v)
{
    if (!(v > -2147483649.0 && v < 2147483648.0))
        return std::numeric_limits<int>::min();
    return static_cast<int>(v);
}

// lib/src/parser.cpp:540-561 (set_bound_variable): only -inf on the left and
// +inf on the right are replaced, everything else is cast.
int
to_bound_min(double v)
{
    return v == -std::numeric_limits<double>::infinity() ? std::numeric_limits<int>::min()
                                                         : to_int(v);
}

int
to_bound_max(double v)
{
    return v == std::numeric_limits<double>::infinity() ? std::numeric_limits<int>::max()
                                                        : to_int(v);
}

// ---- the bound section, as the reference reads it ---------------------------
// A bound line is one of (lib/src/parser.cpp:907-965)
//   l op name op r   -> [l, r]   (l > r is an error)
//   l op name        -> [l, +inf], and ONE MORE token is consumed (:924-928
//                       returns read + 1 after read already counts the name)
//   name op r        -> [0, r]   whatever op is (:944-956): `t = 1` is [0, 1]
//   name             -> [-inf, +inf]
// where op is any of < <= =< > >= => = == and is not looked at.  A token is
// taken for a number by its first character alone -- digits . e E - + or the
// words inf / infinity (:118-129) -- so a bound on a variable whose name
// starts with e or E is a bad_bound; `free` is not a keyword of the reference
// (a variable of that name would be), and the section must be closed by one
// of the keywords (:152-153): the end of the file is a bad_bound.
bool
bound_char_name(int c)
{
    if ((c >= 'a' && c <= 'z') || (c >= 'A' && c <= 'Z') || (c >= '0' && c <= '9'))
        return true;
    switch (c) {
    case '!': case '"': case '#': case '$': case '%': case '&': case '(': case ')':
    case ',': case '.': case ';': case '?': case '@': case '_': case '{': case '}':
    case '~':
        return true;
    default:
        return false;
    }
}

bool
bound_keyword(std::string_view t)
{
    static const char* const words[] = { "binary", "binaries", "bin",  "bound", "bounds",
                                         "general", "generals", "gen", "end",   "st",
                                         "subject", "sush",     "s.t.", "st." };
    for (auto w : words)
        if (ieq(t, w))
            return true;
    return false;
}

bool
bound_starts_with_number(std::string_view t)
{
    if (t.empty())
        return false;
    if (ieq(t, "infinity") || ieq(t, "inf"))
        return true;
    const char c = t[0];
    return (c >= '0' && c <= '9') || c == '.' || c == 'e' || c == 'E' || c == '-' || c == '+';
}

bool
bound_is_name(std::string_view t) // read_name, :612-619: every character a name character
{
    for (char c : t)
        if (!bound_char_name(static_cast<unsigned char>(c)))
            return false;
    return true;
}

std::optional<double>
bound_real(std::string_view t) // read_real, :565-584: `inf`, else what sscanf("%lf") reads
{
    if (ieq(t, "inf"))
        return std::numeric_limits<double>::infinity();
    if (t.empty() || t.size() > 319)
        return std::nullopt;
    const std::string z(t);
    double v = 0.0;
    if (std::sscanf(z.c_str(), "%lf", &v) > 0)
        return v;
    return std::nullopt;
}

int
bound_operator(std::string_view a, std::string_view b) // tokens of an operator, 0: none (:622-649)
{
    if (a == "<" || a == ">")
        return b == "=" ? 2 : 1;
    if (a == "=")
        return (b == "<" || b == "=" || b == ">") ? 2 : 1;
    return 0;
}

struct bound_line
{
    double min, max;
    std::string_view name;
    int read;
};

struct sub_bound
{
    double value;
    int read;
};

std::optional<sub_bound>
read_left_bound(const lexer& lx) // :825-860: [+|-] number op
{
    const bool sign = lx.peek(0) == "+" || lx.peek(0) == "-";
    const size_t at = sign ? 1 : 0;
    auto v = bound_real(lx.peek(at));
    if (!v)
        return std::nullopt;
    const int op = bound_operator(lx.peek(at + 1), lx.peek(at + 2));
    if (!op)
        return std::nullopt;
    return sub_bound{ lx.peek(0) == "-" ? -*v : *v, static_cast<int>(at) + 1 + op };
}

std::optional<sub_bound>
read_right_bound(const lexer& lx, size_t from) // :862-905: op [+|-] number
{
    const int op = bound_operator(lx.peek(from), lx.peek(from + 1));
    if (!op)
        return std::nullopt;
    size_t at = from + static_cast<size_t>(op);
    double negative = 1.0;
    if (lx.peek(at) == "+" || lx.peek(at) == "-") {
        if (lx.peek(at) == "-")
            negative = -1.0;
        ++at;
    }
    auto v = bound_real(lx.peek(at));
    if (!v)
        return std::nullopt;
    return sub_bound{ negative * *v, static_cast<int>(at + 1 - from) };
}

std::optional<bound_line>
read_bound(const lexer& lx)
{
    constexpr double inf = std::numeric_limits<double>::infinity();
    const auto first = lx.peek(0);
    if (bound_starts_with_number(first)) {
        auto left = read_left_bound(lx);
        if (!left)
            return std::nullopt;
        int read = left->read;
        const auto name = lx.peek(static_cast<size_t>(read));
        if (!bound_is_name(name))
            return std::nullopt;
        ++read;
        const auto next = lx.peek(static_cast<size_t>(read));
        if (next.empty() || !(next[0] == '<' || next[0] == '>' || next[0] == '='))
            return bound_line{ left->value, inf, name, read + 1 };
        auto right = read_right_bound(lx, static_cast<size_t>(read));
        if (!right)
            return std::nullopt;
        if (left->value > right->value)
            return std::nullopt;
        return bound_line{ left->value, right->value, name, read + right->read };
    }
    if (!first.empty() && bound_char_name(static_cast<unsigned char>(first[0]))) {
        if (!bound_is_name(first))
            return std::nullopt;
        const auto next = lx.peek(1);
        if (!next.empty() && (next[0] == '<' || next[0] == '>' || next[0] == '=')) {
            auto right = read_right_bound(lx, 1);
            if (!right)
                return std::nullopt;
            return bound_line{ 0.0, right->value, first, 1 + right->read };
        }
        return bound_line{ -inf, inf, first, 1 };
    }
    return std::nullopt;
}

// Where a variable already sits in the function being read: (stamp of that
// function, position) per variable.
struct seen_table
{
    std::vector<std::pair<int, int>> at;
    int find(int id, int stamp) const
    {
        return static_cast<size_t>(id) < at.size() && at[static_cast<size_t>(id)].first == stamp
                 ? at[static_cast<size_t>(id)].second
                 : -1;
    }
    void mark(int id, int stamp, int pos)
    {
        if (static_cast<size_t>(id) >= at.size())
            at.resize(static_cast<size_t>(id) + 1 + at.size() / 2, std::make_pair(-1, -1));
        at[static_cast<size_t>(id)] = std::make_pair(stamp, pos);
    }
};

// read_real(buf_1, buf_2), :587-611: [+|-] number -> {value, tokens read}; a sign without a
// number is +-1 and one token, no sign and no number is 1 and NO token
struct real_read
{
    double value;
    int read;
};

real_read
read_real2(std::string_view b1, std::string_view b2)
{
    if (b1 == "-") {
        auto v = bound_real(b2);
        return v ? real_read{ -1.0 * *v, 2 } : real_read{ -1.0, 1 };
    }
    if (b1 == "+") {
        auto v = bound_real(b2);
        return v ? real_read{ *v, 2 } : real_read{ 1.0, 1 };
    }
    auto v = bound_real(b1);
    return v ? real_read{ *v, 1 } : real_read{ 1.0, 0 };
}

// read_function_element, :789-822: [sign] [number] [name]; the name is whatever token comes
// next when it starts with a name character (digits are name characters: a second number in
// a row is a name) and is no keyword
struct element_read
{
    double factor;
    std::string_view name;
    int read;
};

std::optional<element_read>
read_element(const lexer& lx)
{
    const real_read v = read_real2(lx.peek(0), lx.peek(1));
    const std::string_view to_read = lx.peek(static_cast<size_t>(std::min(v.read, 2)));
    if (!bound_keyword(to_read) && !to_read.empty() && bound_char_name(static_cast<unsigned char>(to_read[0]))) {
        if (!bound_is_name(to_read))
            return std::nullopt;
        return element_read{ v.value, to_read, v.read + 1 };
    }
    return element_read{ v.value, std::string_view(), v.read };
}

// (the reference's buffer advances by 1 .. 9 tokens; anything else trips its contract check:
// here the nearest error of the section being read)
bool
advance(lexer& lx, int read)
{
    if (read < 1 || read > 9)
        return false;
    lx.pop(static_cast<size_t>(read));
    return true;
}

int
section_len(const lexer& lx, std::initializer_list<const char*> words) // keyword [:] -> tokens, 0: not here
{
    for (auto w : words)
        if (ieq(lx.peek(0), w))
            return ieq(lx.peek(1), ":") ? 2 : 1;
    return 0;
}

file_format_error_tag
parse(lexer& lx, builder& b)
{
    seen_table seen;
    int stamp_counter = 0; // 0: the objective, then one per constraint
    using tag = file_format_error_tag;
    if (lx.overlong)
        return tag::bad_name;

    if (ieq(lx.peek(), "maximize") || ieq(lx.peek(), "maximum") || ieq(lx.peek(), "max"))
        b.pb.type = objective_function_type::maximize;
    else if (ieq(lx.peek(), "minimize") || ieq(lx.peek(), "minimum") || ieq(lx.peek(), "min"))
        b.pb.type = objective_function_type::minimize;
    else
        return tag::bad_objective_function_type;
    lx.pop();

    if (!bound_keyword(lx.peek()) && ieq(lx.peek(1), ":"))
        lx.pop(2); // objective label (:982-993)

    // ---- objective (:1086-1113) ------------------------------------------------
    while (!bound_keyword(lx.peek())) {
        const bool quad = lx.peek() == "[" ||
                          ((lx.peek() == "+" || lx.peek() == "-") && lx.peek(1) == "[");
        if (quad) { // (:692-786)
            double sign = 1.0;
            if (lx.peek() == "-")
                sign = -1.0;
            if (lx.peek() != "[")
                lx.pop();
            lx.pop(); // [
            while (!lx.peek().empty() && lx.peek() != "]") {
                const real_read v = read_real2(lx.peek(0), lx.peek(1));
                const std::string_view name = lx.peek(static_cast<size_t>(std::min(v.read, 2)));
                if (bound_keyword(name) || name.empty() || !bound_char_name(static_cast<unsigned char>(name[0])) ||
                    !bound_is_name(name))
                    return tag::bad_objective_quadratic;
                lx.pop(static_cast<size_t>(v.read) + 1);
                int a = -1, c = -1;
                if (lx.peek() == "*") {
                    if (!bound_is_name(lx.peek(1)))
                        return tag::bad_objective_quadratic;
                    a = b.variable(name);
                    c = b.variable(lx.peek(1));
                    lx.pop(2);
                } else if (lx.peek() == "^") { // (`^` is a separator: `^2` is always two tokens)
                    lx.pop(lx.peek(1) == "2" ? 2 : 1);
                    a = c = b.variable(name);
                } else
                    continue; // (neither: the element is dropped, as the reference drops it)
                auto& q = b.pb.objective.qelements;
                auto it = std::find_if(q.begin(), q.end(), [a, c](const auto& e) {
                    return (e.variable_index_a == a && e.variable_index_b == c) ||
                           (e.variable_index_a == c && e.variable_index_b == a);
                });
                if (it == q.end())
                    q.emplace_back(v.value * sign / 2.0, a, c);
                else
                    it->factor += v.value * sign / 2.0;
            }
            lx.pop(); // ]
            if (lx.peek() == "/" && lx.peek(1) == "2")
                lx.pop(2);
            else if (lx.peek() == "/2")
                lx.pop(1);
            else
                return tag::bad_objective_quadratic;
            continue;
        }
        auto t = read_element(lx);
        if (!t)
            return tag::bad_objective;
        if (t->name.empty()) {
            b.pb.objective.value += t->factor;
        } else {
            // (a variable named twice adds up, lib/src/parser.cpp:492-511 -- found there by a
            // scan of the elements read so far, O(r^2) per function; here by a stamp per
            // variable: the set-covering files of 1e6 columns parse in linear time)
            const int id = b.variable(t->name);
            auto& el = b.pb.objective.elements;
            const int pos = seen.find(id, 0);
            if (pos < 0) {
                seen.mark(id, 0, static_cast<int>(el.size()));
                el.emplace_back(t->factor, id);
            } else
                el[static_cast<size_t>(pos)].factor += t->factor;
        }
        if (!advance(lx, t->read))
            return tag::bad_objective;
    }

    // ---- constraints (:1115-1210) ----------------------------------------------
    {
        int st = 0;
        const auto a0 = lx.peek(0), a1 = lx.peek(1), a2 = lx.peek(2);
        if (ieq(a0, "st") || ieq(a0, "st.") || ieq(a0, "s.t") || ieq(a0, "s.t."))
            st = ieq(a1, ":") ? 2 : 1;
        else if ((ieq(a0, "subject") && ieq(a1, "to")) || (ieq(a0, "sush") && ieq(a1, "that")))
            st = ieq(a2, ":") ? 3 : 2;
        if (st) {
            lx.pop(static_cast<size_t>(st));
            int next_id = 0;
            while (!lx.peek().empty() && !bound_keyword(lx.peek())) {
                std::string_view label;
                if (bound_char_name(static_cast<unsigned char>(lx.peek()[0])) && lx.peek(1) == ":") {
                    label = b.pb.strings->append(lx.peek());
                    lx.pop(2);
                }
                std::vector<function_element> elements;
                const int stamp = ++stamp_counter;
                bool first = true;
                // (at least one element; then elements until a token starts with an operator
                // character.  An element without a name names the EMPTY variable, :1143.)
                while (first || (!lx.peek().empty() && !(lx.peek()[0] == '<' || lx.peek()[0] == '>' ||
                                                         lx.peek()[0] == '='))) {
                    auto t = read_element(lx);
                    if (!t)
                        return tag::bad_constraint;
                    const int id = b.variable(t->name);
                    const int pos = seen.find(id, stamp); // (:1146-1155, in linear time)
                    if (pos < 0) {
                        seen.mark(id, stamp, static_cast<int>(elements.size()));
                        elements.emplace_back(to_int(t->factor), id);
                    } else
                        elements[static_cast<size_t>(pos)].factor += to_int(t->factor);
                    if (!advance(lx, t->read))
                        return tag::bad_constraint;
                    first = false;
                }
                const int op_len = bound_operator(lx.peek(0), lx.peek(1));
                if (!op_len)
                    return tag::bad_constraint;
                operator_type op = operator_type::equal;
                if (lx.peek(0) == "<" || (lx.peek(0) == "=" && op_len == 2 && lx.peek(1) == "<"))
                    op = operator_type::less;
                else if (lx.peek(0) == ">" || (lx.peek(0) == "=" && op_len == 2 && lx.peek(1) == ">"))
                    op = operator_type::greater;
                lx.pop(static_cast<size_t>(op_len));
                const real_read rhs = read_real2(lx.peek(0), lx.peek(1));
                const int value = to_int(rhs.value);
                if (!advance(lx, rhs.read))
                    return tag::bad_constraint;
                switch (op) {
                case operator_type::equal:
                    b.pb.equal_constraints.emplace_back(label, std::move(elements), value, next_id++);
                    break;
                case operator_type::greater:
                    b.pb.greater_constraints.emplace_back(label, std::move(elements), value, next_id++);
                    break;
                case operator_type::less:
                    b.pb.less_constraints.emplace_back(label, std::move(elements), value, next_id++);
                    break;
                }
            }
        }
    }

    // ---- bounds ------------------------------------------------------------
    // The reference's bound rules, token for token (lib/src/parser.cpp:1213-1226
    // around read_bound :907-965): the operators of a bound line are read and
    // ignored.
    if (const int n = section_len(lx, { "bounds", "bound" })) {
        lx.pop(static_cast<size_t>(n));
        while (!bound_keyword(lx.peek())) {
            auto bound = read_bound(lx);
            if (!bound)
                return tag::bad_bound;
            const int id = b.find(bound->name);
            if (id < 0)
                return tag::bad_bound;
            b.pb.vars.values[id].min = to_bound_min(bound->min);
            b.pb.vars.values[id].max = to_bound_max(bound->max);
            lx.pop(static_cast<size_t>(bound->read));
        }
    }

    // ---- binary / general / end (:1228-1256) -------------------------------------
    if (const int n = section_len(lx, { "binary", "binaries", "bin" })) {
        lx.pop(static_cast<size_t>(n));
        while (!bound_keyword(lx.peek())) {
            const int id = b.find(lx.peek());
            if (id < 0)
                return tag::bad_binary;
            b.pb.vars.values[id] = variable_value(0, 1, variable_type::binary);
            lx.pop();
        }
    }
    if (const int n = section_len(lx, { "general", "generals", "gen" })) {
        lx.pop(static_cast<size_t>(n));
        while (!bound_keyword(lx.peek())) {
            const int id = b.find(lx.peek());
            if (id < 0)
                return tag::bad_general;
            b.pb.vars.values[id].type = variable_type::general;
            lx.pop();
        }
    }
    if (const int n = section_len(lx, { "end" })) {
        lx.pop(static_cast<size_t>(n));
        return lx.peek().empty() ? tag::success : tag::bad_end_of_file;
    }
    return tag::bad_end;
}

} // namespace

raw_problem
make_problem(const context& /*ctx*/, std::istream& is) noexcept
{
    try {
        lexer lx(is);
        builder b;
        const auto st = parse(lx, b);
        if (st != file_format_error_tag::success)
            return raw_problem(st);
        return std::move(b.pb);
    } catch (...) {
        return raw_problem(file_format_error_tag::bad_end_of_file);
    }
}

namespace {

// OR-Library set-covering format (scp4x .. scpnrh, scpclr, ...): `m n`, the n
// column costs, then per row the number of columns covering it followed by
// their 1-based indices.  The reference reads such instances only after they
// have been rewritten as CPLEX-LP files (benchmarks/download.sh); this reads
// them as they are, in one pass over the numbers: min sum c_j x_j, every row
// sum_{j covers i} x_j >= 1, x binary (variables x1 .. xn, constraints r1 .. rm).
raw_problem
read_scp(std::istream& is)
{
    using tag = file_format_error_tag;
    raw_problem pb;
    pb.strings = std::make_shared<string_buffer>();
    pb.type = objective_function_type::minimize;
    long m = 0, n = 0;
    if (!(is >> m >> n) || m <= 0 || n <= 0 || m > 0x7fffffffL || n > 0x7fffffffL)
        return raw_problem(tag::bad_objective);
    pb.vars.names.reserve(static_cast<size_t>(n));
    pb.vars.values.reserve(static_cast<size_t>(n));
    pb.objective.elements.reserve(static_cast<size_t>(n));
    for (long j = 0; j != n; ++j) {
        double c;
        if (!(is >> c))
            return raw_problem(tag::bad_objective);
        pb.vars.names.emplace_back(pb.strings->append("x" + std::to_string(j + 1)));
        pb.vars.values.emplace_back(0, 1, variable_type::binary);
        pb.objective.elements.emplace_back(c, static_cast<int>(j));
    }
    pb.greater_constraints.reserve(static_cast<size_t>(m));
    // A row is the SET of the columns that cover it: an index listed twice stays
    // one element with factor 1 (the LP reader would add the factors up to 2 and
    // send the instance to the Z solver).  seen[j] = last row that listed j.
    std::vector<long> seen(static_cast<size_t>(n), -1);
    for (long i = 0; i != m; ++i) {
        long k;
        if (!(is >> k) || k <= 0)
            return raw_problem(tag::bad_constraint);
        std::vector<function_element> elements;
        elements.reserve(static_cast<size_t>(std::min(k, n)));
        for (long t = 0; t != k; ++t) {
            long j;
            if (!(is >> j) || j < 1 || j > n)
                return raw_problem(tag::bad_constraint);
            if (seen[static_cast<size_t>(j - 1)] == i)
                continue;
            seen[static_cast<size_t>(j - 1)] = i;
            elements.emplace_back(1, static_cast<int>(j - 1));
        }
        pb.greater_constraints.emplace_back(pb.strings->append("r" + std::to_string(i + 1)),
                                            std::move(elements), 1, static_cast<int>(i));
    }
    return pb;
}

bool
ends_with(const std::string& s, const char* suffix)
{
    const std::string t(suffix);
    return s.size() >= t.size() && s.compare(s.size() - t.size(), t.size(), t) == 0;
}

} // namespace

raw_problem
make_problem(const context_ptr& ctx, const std::string& filename)
{
    if (!ctx)
        return raw_problem(file_format_error_tag::empty_context);
    std::ifstream ifs(filename);
    if (!ifs.is_open())
        return raw_problem(file_format_error_tag::file_not_found);
    if (ends_with(filename, ".scp")) {
        try {
            return read_scp(ifs);
        } catch (const std::bad_alloc&) {
            throw; // not a format error: solve / optimize report it as not_enough_memory
        } catch (...) {
            return raw_problem(file_format_error_tag::bad_end_of_file);
        }
    }
    return make_problem(*ctx, ifs);
}

} // namespace baryonyx
