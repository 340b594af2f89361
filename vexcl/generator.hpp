#ifndef VEXCL_GENERATOR_HPP
#define VEXCL_GENERATOR_HPP
// Kernel generator: a generic C++ algorithm is run once on vex::symbolic<T> values, which
// RECORD the operations applied to them as kernel source; the recorded sequence becomes one
// fused kernel or one device function (reference: vexcl/generator.hpp:68-851).
//
//   std::ostringstream body;
//   vex::generator::set_recorder(body);
//   vex::symbolic<double> x(vex::symbolic<double>::VectorParameter);
//   runge_kutta_4(system, x, dt);                                     // any generic code
//   auto kernel = vex::generator::build_kernel(ctx, "rk4", body.str(), x);
//   kernel(X);                                                        // one pass over HBM for the whole algorithm
//
// symbolic<T> is a terminal of the same expression templates as vex::vector (operators,
// builtin and user functions, if_else); what differs is the traversal: recording emits the
// expression text with scalar values as LITERALS and without kernel parameters.  The kernel is
// launched like the fused elementwise kernels (hiprtc, ~2 elements per lane on a streaming grid).
#include <iostream>
#include <map>
#include <memory>
#include <set>
#include <sstream>
#include <string>
#include <tuple>
#include <vector>
#include "function.hpp"
#include "vector.hpp"

namespace vex {

template <typename T> class symbolic;

namespace generator {

namespace detail {
/// Where the recorded statements go, the device functions they use, and the variable counter
/// (generator.hpp:75-117 of the reference: one recorder per process).
struct recorder_state {
    std::ostream *os = nullptr;
    std::unique_ptr<backend::source_generator> preamble{new backend::source_generator};
    std::set<std::string> seen;
    size_t index = 0;
    static recorder_state &get() { static recorder_state s; return s; }
};
}

inline size_t var_id() { return ++detail::recorder_state::get().index; }
inline std::ostream &get_recorder() { auto &s = detail::recorder_state::get(); return s.os ? *s.os : std::cout; }
inline backend::source_generator &get_preamble() { return *detail::recorder_state::get().preamble; }

/// Directs the recording to `os`; forgets the device functions collected so far.
inline void set_recorder(std::ostream &os) {
    auto &s = detail::recorder_state::get();
    s.os = &os;
    s.preamble.reset(new backend::source_generator);
    s.seen.clear();
}

namespace detail {
/// Writes the text of `expr` to the recorder; device functions it calls go to the preamble once.
template <class Expr>
void record(const Expr &expr) {
    auto &st = recorder_state::get();
    static const backend::command_queue none;
    const auto &e = vex::detail::as_expr<Expr>::get(expr);
    { vex::detail::gen_context c(*st.preamble, none, st.seen); e.preamble(c); }
    backend::source_generator text;
    { vex::detail::gen_context c(text, none, st.seen); e.emit(c); }
    get_recorder() << text.str();
}

/// generator::index(): the position of the element being processed.
struct index_expr : vex::detail::expression_base {
    typedef size_t value_type;
    void preamble(vex::detail::gen_context &c) const { c.next(); }
    void params(vex::detail::gen_context &c) const { c.next(); }
    void local_init(vex::detail::gen_context &c) const { c.next(); }
    void emit(vex::detail::gen_context &c) const { c.next(); c.src << "idx"; }
    void set_args(vex::detail::arg_context &a) const { a.next(); }
    void get_props(vex::detail::prop_context &) const {}
};

/// How a symbolic variable appears inside expression nodes: by number (copying the
/// variable itself would record a declaration).
template <class T>
struct symbolic_ref : vex::detail::expression_base {
    typedef T value_type;
    size_t id;
    symbolic_ref(const symbolic<T> &s) : id(s.id()) {}
    void preamble(vex::detail::gen_context &c) const { c.next(); }
    void params(vex::detail::gen_context &c) const { c.next(); }
    void local_init(vex::detail::gen_context &c) const { c.next(); }
    void emit(vex::detail::gen_context &c) const { c.next(); c.src << "var" << id; }
    void set_args(vex::detail::arg_context &a) const { a.next(); }
    void get_props(vex::detail::prop_context &) const {}
};
} // namespace detail

inline detail::index_expr index() { return detail::index_expr(); }

} // namespace generator

/// Symbolic variable.
template <typename T>
class symbolic : public detail::expression_base {
    public:
        typedef T value_type;
        typedef generator::detail::symbolic_ref<T> expr_ref_type;

        enum scope_type { LocalVar = 0, VectorParameter = 1, ScalarParameter = 2 };
        /// NonConst vector parameters are written back when the kernel ends.
        enum constness_type { NonConst = 0, Const = 1 };

        /// A local variable, value-initialized.
        symbolic() : num(generator::var_id()), scope(LocalVar), constness(NonConst) {
            generator::get_recorder() << "\t\t" << type_name<T>() << " " << *this << " = " << T() << ";\n";
        }
        explicit symbolic(scope_type scope, constness_type constness = NonConst)
            : num(generator::var_id()), scope(scope), constness(constness)
        {
            if (scope == LocalVar) generator::get_recorder() << "\t\t" << type_name<T>() << " " << *this << ";\n";
        }
        /// A copy is a NEW local variable initialized with the other one.
        symbolic(const symbolic &other) : num(generator::var_id()), scope(LocalVar), constness(NonConst) { declare(other); }
        /// A new local variable initialized with the expression.
        template <class Expr, class = typename std::enable_if<detail::is_operand<Expr>::value>::type>
        symbolic(const Expr &expr) : num(generator::var_id()), scope(LocalVar), constness(NonConst) { declare(expr); }

        const symbolic &operator=(const symbolic &other) const { return assign("=", other); }
#define VEXCL_SYMBOLIC_ASSIGNMENT(op)                                                                    \
        template <class Expr>                                                                            \
        typename std::enable_if<detail::is_operand<Expr>::value, const symbolic &>::type                 \
        operator op(const Expr &expr) const { return assign(#op, expr); }
        VEXCL_SYMBOLIC_ASSIGNMENT(=)  VEXCL_SYMBOLIC_ASSIGNMENT(+=) VEXCL_SYMBOLIC_ASSIGNMENT(-=)
        VEXCL_SYMBOLIC_ASSIGNMENT(*=) VEXCL_SYMBOLIC_ASSIGNMENT(/=) VEXCL_SYMBOLIC_ASSIGNMENT(%=)
        VEXCL_SYMBOLIC_ASSIGNMENT(&=) VEXCL_SYMBOLIC_ASSIGNMENT(|=) VEXCL_SYMBOLIC_ASSIGNMENT(^=)
        VEXCL_SYMBOLIC_ASSIGNMENT(<<=) VEXCL_SYMBOLIC_ASSIGNMENT(>>=)
#undef VEXCL_SYMBOLIC_ASSIGNMENT

        size_t id() const { return num; }

        /// Statement that loads the parameter into the local variable when the kernel starts.
        std::string init() const {
            std::ostringstream s;
            if (scope == VectorParameter) s << "\t\t" << type_name<T>() << " " << *this << " = p_" << *this << "[idx];\n";
            else if (scope == ScalarParameter) s << "\t\t" << type_name<T>() << " " << *this << " = p_" << *this << ";\n";
            return s.str();
        }
        /// Statement that stores the local variable back when the kernel ends.
        std::string write() const {
            std::ostringstream s;
            if (scope == VectorParameter && constness == NonConst) s << "\t\tp_" << *this << "[idx] = " << *this << ";\n";
            return s.str();
        }
        /// (type, name) of the kernel parameter.
        std::tuple<std::string, std::string> prmdecl() const {
            std::ostringstream name; name << "p_" << *this;
            std::string type = scope == VectorParameter
                ? (constness == Const ? type_name<global_ptr<const T>>() : type_name<global_ptr<T>>())
                : type_name<T>();
            return std::make_tuple(type, name.str());
        }

        friend std::ostream &operator<<(std::ostream &os, const symbolic &s) { return os << "var" << s.num; }

    private:
        size_t num;
        scope_type scope;
        constness_type constness;

        template <class Expr> void declare(const Expr &expr) const {
            generator::get_recorder() << "\t\t" << type_name<T>() << " " << *this << " = ";
            generator::detail::record(expr);
            generator::get_recorder() << ";\n";
        }
        template <class Expr> const symbolic &assign(const char *op, const Expr &expr) const {
            generator::get_recorder() << "\t\t" << *this << " " << op << " ";
            generator::detail::record(expr);
            generator::get_recorder() << ";\n";
            return *this;
        }
};

namespace generator {

/// A kernel built from a recorded sequence; call it with one vex::vector (or scalar) per symbolic parameter.
class kernel {
    public:
        kernel(const std::vector<backend::command_queue> &queue, const std::string &name)
            : queue(queue), name(name), psize(queue.size(), 0), impl(new std::vector<backend::kernel>()) {}

        template <class SymVar> void add_param(const SymVar &var) {
            prm_decl.push_back(var.prmdecl());
            prm_read += var.init();
            prm_save += var.write();
        }

        void build(const std::string &body) {
            for (const auto &q : queue) {
                backend::source_generator source(q);
                source << get_preamble().str();
                source.begin_kernel(name);
                source.begin_kernel_parameters();
                for (const auto &p : prm_decl) source.parameter(std::get<0>(p), std::get<1>(p));
                source.template parameter<size_t>("n");
                source.end_kernel_parameters();
                source.grid_stride_loop().open("{");
                source.new_line() << prm_read << body << prm_save;
                source.close("}");
                source.end_kernel();
                impl->push_back(backend::kernel(q, source.str(), name));
            }
        }

        template <class T> void push_arg(const T &v) { for (auto &k : *impl) k.push_arg(v); }
        template <class T> void push_arg(const vector<T> &v) {
            for (unsigned d = 0; d < queue.size(); ++d) {
                (*impl)[d].push_arg(v(d));
                psize[d] = std::max(psize[d], v.part_size(d));
            }
        }
        /// One value per device.
        template <class T> void push_arg(const std::vector<T> &args) {
            for (unsigned d = 0; d < queue.size(); ++d) (*impl)[d].push_arg(args[d]);
        }

        void operator()() {
            for (unsigned d = 0; d < queue.size(); ++d) {
                backend::kernel &K = (*impl)[d];
                if (psize[d]) {
                    K.push_arg(psize[d]);
                    K.config_streaming(queue[d], psize[d]);
                    K(queue[d]);
                    psize[d] = 0;
                } else {
                    K.reset();
                }
            }
        }
        template <class Head, class... Tail>
        void operator()(const Head &head, const Tail &...tail) { push_arg(head); (*this)(tail...); }

        static void add_params(kernel &) {}
        template <class Head, class... Tail>
        static void add_params(kernel &K, const Head &head, const Tail &...tail) { K.add_param(head); add_params(K, tail...); }

    private:
        std::vector<backend::command_queue> queue;
        std::string name;
        std::vector<size_t> psize;
        std::vector<std::tuple<std::string, std::string>> prm_decl;
        std::string prm_read, prm_save;
        std::shared_ptr<std::vector<backend::kernel>> impl;      // one per queue
};

/// Builds the kernel `name` from the recorded `body` and the symbolic parameters, in call order.
template <class... Args>
kernel build_kernel(const std::vector<backend::command_queue> &queue, const std::string &name,
                    const std::string &body, const Args &...args)
{
    kernel K(queue, name);
    kernel::add_params(K, args...);
    K.build(body);
    return K;
}

/// Body of a device function `ret f(args...)` from the recorded sequence (parameters are called prm1, prm2, ...:
/// use with VEX_FUNCTION_S(type, name, (type, prm1)..., body)).
template <class Ret, class... Args>
std::string make_function(std::string body, const Ret &ret, const Args &...args) {
    std::ostringstream source;
    int k = 0;
    int dummy[] = {0, ((source << "\t\t" << type_name<typename Args::value_type>() << " " << args << " = prm" << ++k << ";\n"), 0)...};
    (void)dummy;
    source << body << "\t\treturn " << ret << ";\n";
    return source.str();
}

/// A user function whose body is recorded from a generic functor.
template <class Signature, class Functor> struct FunctorAdapter;
template <class R, class... A, class Functor>
struct FunctorAdapter<R(A...), Functor> : UserFunction<FunctorAdapter<R(A...), Functor>, R> {
    FunctorAdapter() {}
    template <class F>
    FunctorAdapter(F &&f, const std::string &fname) {
        name_string() = fname;
        std::ostringstream source;
        set_recorder(source);
        record_body(std::forward<F>(f), source, std::index_sequence_for<A...>());
        body_string() = source.str();
        deps_string() = get_preamble().str();
    }
    static std::string name() { return name_string(); }
    static std::string body() { return body_string(); }
    static void params(std::vector<std::pair<std::string, std::string>> &p) { vex::detail::signature_params<R(A...)>::get(p); }
    /// device functions the recorded body calls
    static void dependencies(vex::detail::gen_context &c) { c.src << deps_string(); }

    private:
        static std::string &name_string() { static std::string s; return s; }
        static std::string &body_string() { static std::string s; return s; }
        static std::string &deps_string() { static std::string s; return s; }

        template <class F, size_t... I>
        static void record_body(F &&f, std::ostream &source, std::index_sequence<I...>) {
            std::tuple<symbolic<A>...> prm{(void(I), symbolic<A>::ScalarParameter)...};       // constructed in place: a copy would record a declaration
            int dummy[] = {0, ((source << "\t\t" << type_name<A>() << " " << std::get<I>(prm) << " = prm" << I + 1 << ";\n"), 0)...};
            (void)dummy;
            symbolic<R> ret = f(std::get<I>(prm)...);
            source << "\t\treturn " << ret << ";\n";
        }
};

inline size_t get_gen_fun_id() { static size_t id = 0; return id++; }

/// make_function<double(double, double)>(functor): a device function usable in vector expressions.
template <class Signature, class Functor>
FunctorAdapter<Signature, typename std::decay<Functor>::type> make_function(Functor &&f) {
    std::ostringstream name;
    name << "generated_function_" << get_gen_fun_id();
    return FunctorAdapter<Signature, typename std::decay<Functor>::type>(std::forward<Functor>(f), name.str());
}

} // namespace generator
} // namespace vex
#endif
