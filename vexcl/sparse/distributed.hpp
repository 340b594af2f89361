#ifndef VEXCL_SPARSE_DISTRIBUTED_HPP
#define VEXCL_SPARSE_DISTRIBUTED_HPP
// vex::sparse::distributed<Matrix>: a row-partitioned matrix over all devices of
// a context whose product is still an inlinable terminal (reference:
// vexcl/sparse/distributed.hpp:23-432).  Per device: a local matrix over the
// owned columns (renumbered c - col_begin) and a remote matrix over the ghost
// columns (renumbered to their rank in the sorted ghost list, :49-131); A * x
// runs the ghost exchange eagerly (:227-230) and returns a terminal that prints
// loc_sum + rem_sum (:244-262).
#include <memory>
#include <set>
#include "matrix.hpp"
#include "../exchange.hpp"

namespace vex {
namespace sparse {

namespace detail {
    /// One device buffer per device, used as the x operand of the remote part.
    template <class T>
    struct buffer_ref : vex::detail::expression_base {
        typedef T value_type;
        const vex::detail::ghost_exchange<T> *ex;
        explicit buffer_ref(const vex::detail::ghost_exchange<T> &e) : ex(&e) {}
        void preamble(vex::detail::gen_context &c) const { c.next(); }
        void params(vex::detail::gen_context &c) const { c.src.template parameter<global_ptr<T>>(c.next()); }
        void local_init(vex::detail::gen_context &c) const { c.next(); }
        void emit(vex::detail::gen_context &c) const { c.src << c.next() << "[idx]"; }
        void set_args(vex::detail::arg_context &a) const { a.next(); a.krn.push_arg(ex->ghost_buffer(a.device)); }
        void get_props(vex::detail::prop_context &) const {}
    };
}

template <class Matrix, typename rhs_type = typename Matrix::value_type>
class distributed {
    public:
        typedef typename Matrix::value_type value_type;
        typedef typename Matrix::col_type col_type;
        typedef typename Matrix::ptr_type ptr_type;

        template <class PtrRange, class ColRange, class ValRange>
        distributed(const std::vector<backend::command_queue> &q, size_t nrows, size_t ncols,
                const PtrRange &ptr, const ColRange &col, const ValRange &val, bool fast_setup = true)
            : q(q), n(nrows), m(ncols), nnz(val.size()),
              row_part(vex::partition(nrows, q)), col_part(vex::partition(ncols, q))
        {
            const unsigned nd = static_cast<unsigned>(q.size());
            std::vector<std::vector<col_type>> ghosts(nd);
            for (unsigned d = 0; d < nd; ++d) {
                std::vector<backend::command_queue> qd(1, q[d]);
                const size_t r0 = row_part[d], r1 = row_part[d + 1], c0 = col_part[d], c1 = col_part[d + 1];
                std::set<col_type> gset;
                for (size_t i = r0; i < r1; ++i)
                    for (auto j = ptr[i]; j < ptr[i + 1]; ++j)
                        if (static_cast<size_t>(col[j]) < c0 || static_cast<size_t>(col[j]) >= c1) gset.insert(col[j]);
                ghosts[d].assign(gset.begin(), gset.end());
                std::vector<ptr_type> lp(1, 0), rp(1, 0); std::vector<col_type> lc, rc; std::vector<value_type> lv, rv;
                for (size_t i = r0; i < r1; ++i) {
                    for (auto j = ptr[i]; j < ptr[i + 1]; ++j) {
                        size_t c = static_cast<size_t>(col[j]);
                        if (c >= c0 && c < c1) { lc.push_back(static_cast<col_type>(c - c0)); lv.push_back(val[j]); }
                        else {
                            rc.push_back(static_cast<col_type>(std::lower_bound(ghosts[d].begin(), ghosts[d].end(), col[j]) - ghosts[d].begin()));
                            rv.push_back(val[j]);
                        }
                    }
                    lp.push_back(static_cast<ptr_type>(lc.size()));
                    rp.push_back(static_cast<ptr_type>(rc.size()));
                }
                loc.push_back(lc.empty() ? std::make_shared<Matrix>(q[d])
                                         : std::make_shared<Matrix>(qd, r1 - r0, c1 - c0, lp, lc, lv, fast_setup));
                rem.push_back(rc.empty() ? std::make_shared<Matrix>(q[d])
                                         : std::make_shared<Matrix>(qd, r1 - r0, ghosts[d].size(), rp, rc, rv, fast_setup));
            }
            exchange.setup(q, col_part, ghosts);
        }

        size_t rows() const { return n; }
        size_t cols() const { return m; }
        size_t nonzeros() const { return nnz; }
        const std::vector<backend::command_queue> &queue_list() const { return q; }
        const std::vector<size_t> &row_partition() const { return row_part; }

        /// The product terminal: (loc * x(d)) + (rem * ghosts(d)) inside the fused kernel.
        struct product : vex::detail::expression_base {
            typedef typename std::common_type<typename Matrix::value_type, rhs_type>::type value_type;
            const distributed &A; vex::detail::vector_ref<rhs_type> x; detail::buffer_ref<rhs_type> g;
            std::shared_ptr<vex::vector<rhs_type>> owned;     // set when x was an expression
            product(const distributed &A, const vex::vector<rhs_type> &xv) : A(A), x(xv), g(A.exchange) {}

            void preamble(vex::detail::gen_context &c) const { std::string n = c.next(); Matrix::product_preamble(x, c, n + "_loc"); Matrix::product_preamble(g, c, n + "_rem"); }
            void params(vex::detail::gen_context &c) const { std::string n = c.next(); Matrix::product_params(x, c, n + "_loc"); Matrix::product_params(g, c, n + "_rem"); }
            void local_init(vex::detail::gen_context &c) const {
                std::string n = c.next();
                Matrix::template product_local_init<value_type>(x, c, n + "_loc");
                Matrix::template product_local_init<value_type>(g, c, n + "_rem");
            }
            void emit(vex::detail::gen_context &c) const { std::string n = c.next(); c.src << "( " << n << "_loc_sum + " << n << "_rem_sum )"; }
            void set_args(vex::detail::arg_context &a) const {
                a.next();
                A.loc[a.device]->product_args(x, a);
                A.rem[a.device]->product_args(g, a);
            }
            void get_props(vex::detail::prop_context &p) const {
                if (p.empty()) { p.queue = A.q; p.part = A.row_part; p.size = A.n; }
            }
        };

        friend product operator*(const distributed &A, const vex::vector<rhs_type> &x) {
            precondition(x.size() == A.m, "distributed product: incompatible sizes");
            if (A.exchange.active()) A.exchange.run(x);                  // eager, distributed.hpp:227-230
            return product(A, x);
        }
        /// x given as an expression: evaluated once into a temporary that the terminal keeps alive.
        template <class Expr>
        friend typename std::enable_if<vex::detail::is_expr<Expr>::value && !std::is_same<Expr, vex::vector<rhs_type>>::value, product>::type
        operator*(const distributed &A, const Expr &x) {
            auto tmp = std::make_shared<vex::vector<rhs_type>>(x);
            if (A.exchange.active()) A.exchange.run(*tmp);
            product p(A, *tmp);
            p.owned = tmp;
            return p;
        }
    private:
        std::vector<backend::command_queue> q;
        size_t n, m, nnz;
        std::vector<size_t> row_part, col_part;
        std::vector<std::shared_ptr<Matrix>> loc, rem;
        vex::detail::ghost_exchange<rhs_type> exchange;
};

} // namespace sparse
} // namespace vex
#endif
