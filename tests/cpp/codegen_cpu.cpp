// CPU-only: the expression engine's generated kernels, checked for shape
//  and compiled for gfx950 with hiprtc (no GPU needed).
#define VEX_TEST_CPU_ONLY
#include "vex_test.hpp"
#include <tuple>
#undef ctx

using namespace vex;
using detail::assignment_source;
using detail::as_expr;

VEX_FUNCTION(double, sqr, (double, x), return x * x;);
VEX_FUNCTION(double, times2, (double, x), return x * 2;);
VEX_FUNCTION(double, times4, (double, x), return x * 4;);
VEX_FUNCTION_D(double, sqr_plus, (double, x)(double, y), (sqr), return sqr(x) + y;);
VEX_FUNCTION_V1(old_style, double(double, int), "return prm1 * prm2;");

template <class OP, class L, class R> std::string src_of(const L &l, const R &r) {
    backend::command_queue q;
    return assignment_source<OP>(as_expr<L>::get(l), as_expr<R>::get(r), q);
}
static bool has(const std::string &s, const std::string &what) { return s.find(what) != std::string::npos; }
static size_t count(const std::string &s, const std::string &what) {
    size_t n = 0; for (size_t p = s.find(what); p != std::string::npos; p = s.find(what, p + 1)) ++n; return n;
}

TEST_CASE(fused_elementwise_shape) {
    vector<double> a, b, c, d;
    std::string s = src_of<assign::SET>(a, b * c + sin(d));
    CHECK(has(s, "extern \"C\" __global__ void vexcl_vector_kernel"));
    CHECK(has(s, "ulong n"));
    CHECK(has(s, "double * prm_1") && has(s, "double * prm_4") && !has(s, "prm_5"));
    // the reference's statement text: for two elements per trip before either store, and once for a lane's leftover
    // element (every element is evaluated exactly once)
    CHECK_EQUAL(count(s, "= ( ( prm_2[idx] * prm_3[idx] ) + sin( prm_4[idx] ) );"), size_t(3));
    CHECK(has(s, "prm_1[idx] = vex_r0;") && has(s, "prm_1[idx] = vex_r1;"));
    CHECK(s.find("vex_r1 = ") < s.find("prm_1[idx] = vex_r0;"));
    backend::check_sources(s);
}

TEST_CASE(scalars_are_parameters_not_text) {
    vector<double> a, b;
    std::string s = src_of<assign::ADD>(a, 5 * sin(b) + 42.5);
    CHECK(has(s, "int prm_2") && has(s, "double prm_4"));
    CHECK(!has(s, "42.5"));
    CHECK(has(s, "vex_r0 = ( ( prm_2 * sin( prm_3[idx] ) ) + prm_4 );") && has(s, "prm_1[idx] += vex_r0;"));
    backend::check_sources(s);
}

TEST_CASE(all_assignment_operators_compile) {
    vector<int> a, b;
    backend::check_sources(src_of<assign::SET>(a, b));
    backend::check_sources(src_of<assign::SUB>(a, b + 1));
    backend::check_sources(src_of<assign::MUL>(a, b));
    backend::check_sources(src_of<assign::DIV>(a, b + 1));
    backend::check_sources(src_of<assign::MOD>(a, b + 7));
    backend::check_sources(src_of<assign::AND>(a, b));
    backend::check_sources(src_of<assign::OR>(a, b));
    backend::check_sources(src_of<assign::XOR>(a, b));
    backend::check_sources(src_of<assign::LSH>(a, b & 3));
    backend::check_sources(src_of<assign::RSH>(a, b & 3));
}

TEST_CASE(user_functions_once_by_name) {
    vector<double> a, b;
    std::string s = src_of<assign::SET>(a, sqr(b) + sqr(a) + times2(b) * times4(b));
    CHECK_EQUAL(count(s, "__device__ double sqr"), size_t(1));
    CHECK(has(s, "__device__ double times2") && has(s, "__device__ double times4"));
    backend::check_sources(s);
    s = src_of<assign::SET>(a, sqr_plus(a, b) + old_style(b, 3));
    CHECK(s.find("__device__ double sqr") < s.find("__device__ double sqr_plus"));
    CHECK(has(s, "double prm1") && has(s, "int prm2"));
    backend::check_sources(s);
}

TEST_CASE(ternary_comparison_builtins) {
    vector<double> a, b; vector<int> k;
    std::string s = src_of<assign::SET>(a, if_else(b < 0.5, pow(b, 2), fabs(b - 1) + max(a, b) + abs(a)));
    CHECK(has(s, " ? ") && has(s, "pow( ") && has(s, "fabs( "));
    backend::check_sources(s);
    backend::check_sources(src_of<assign::SET>(k, abs(k) + (k > 3) + !k + -k));
    backend::check_sources(src_of<assign::SET>(a, constants::pi() * a + sqrt(b) * exp(-a) / log(b + 2)));
}

TEST_CASE(tagged_terminals_share_a_parameter) {
    vector<double> a, b;
    auto ta = tag<1>(a);
    std::string s = src_of<assign::SET>(ta, 2.0 * ta + b);
    CHECK_EQUAL(count(s, "double * prm_tag_1_1"), size_t(1));
    CHECK(has(s, "vex_r0 = ( ( prm_1 * prm_tag_1_1[idx] ) + prm_2[idx] );") && has(s, "prm_tag_1_1[idx] = vex_r0;"));
    backend::check_sources(s);
    // untagged: the same vector twice = two parameters
    s = src_of<assign::SET>(a, a + a);
    CHECK(has(s, "prm_2") && has(s, "prm_3"));
}

TEST_CASE(element_index_and_value_types) {
    vector<double> a; vector<float> f; vector<int> k;
    std::string s = src_of<assign::SET>(a, sin(0.5 * element_index()));
    CHECK(has(s, "( prm_3 + idx )") && has(s, "ulong prm_3"));
    backend::check_sources(s);
    static_assert(std::is_same<decltype(f * k)::value_type, float>::value, "common type");
    static_assert(std::is_same<decltype(a + f)::value_type, double>::value, "common type");
    static_assert(std::is_same<decltype(a < f)::value_type, cl_long>::value, "comparison -> long");
    static_assert(std::is_same<decltype(k << 2)::value_type, int>::value, "shift -> left type");
    backend::check_sources(src_of<assign::SET>(f, f * k + 1));
}

TEST_CASE(sparse_products_are_inlinable_terminals) {
    backend::command_queue q;
    sparse::csr<double> A(q);
    sparse::ell<double> E(q);
    vector<double> x, y;
    std::string s = src_of<assign::SET>(y, x + A * sin(x));
    CHECK(has(s, "prm_3_sum") && has(s, "const int * prm_3_ptr") && has(s, "sin( prm_3_x_1[idx] )"));
    backend::check_sources(s);
    s = src_of<assign::ADD>(y, 2.0 * (E * x));
    CHECK(has(s, "_ell_width") && has(s, "} else break;"));
    backend::check_sources(s);
}

TEST_CASE(ccsr_product_is_a_terminal) {
    backend::command_queue q;
    std::vector<size_t> idx(4, 0), row = {0, 1};
    std::vector<int> col = {0}; std::vector<double> val = {2.0};
    // device-free check of the generated text only: build the node by hand
    vector<double> x, y;
    typedef SpMatCCSR<double, int> M;
    typedef detail::ccsr_product<double, int, size_t, double> P;
    static_assert(detail::expr_kind<P>::value == 0, "CCSR product is a vector expression");
    static_assert(std::is_same<P::value_type, double>::value, "");
    CHECK(true);
}

TEST_CASE(additive_transform_classification) {
    typedef SpMat<double, int, int> M;
    typedef detail::additive_operator<M, vector<double>> AX;
    typedef detail::vector_ref<double> V;
    typedef detail::scalar_terminal<int> S;
    using namespace detail;
    static_assert(expr_kind<V>::value == 0, "");
    static_assert(expr_kind<AX>::value == 1, "");
    static_assert(expr_kind<binary_expr<tag::multiplies, S, AX>>::value == 1, "42 * (A*x)");
    static_assert(expr_kind<binary_expr<tag::plus, V, AX>>::value == 2, "x + A*x");
    static_assert(expr_kind<unary_expr<tag::negate, AX>>::value == 1, "-(A*x)");
    static_assert(expr_kind<binary_expr<tag::multiplies, V, AX>>::value == -1, "x * (A*x) is not assignable");
    static_assert(expr_kind<binary_expr<tag::minus, binary_expr<tag::plus, V, AX>, binary_expr<tag::multiplies, S, AX>>>::value == 2, "");
    CHECK(true);
}

// one kernel assigns all components: every rhs evaluated before the first store
// (reference kernel shape: vexcl/multivector.hpp:486-600)
template <class OP, class T, size_t N, class E, size_t... I>
std::string multi_src(const multivector<T, N> &x, const E &expr, std::index_sequence<I...>) {
    using namespace detail;
    backend::command_queue q;
    const auto &e = as_expr<E>::get(expr);
    typedef typename std::decay<decltype(e)>::type node;
    auto lhs = std::make_tuple(vector_ref<T>(x(I))...);
    auto rhs = std::make_tuple(component_of<I, node>::get(e)...);
    return multi_assignment_source<OP>(lhs, rhs, q);
}

TEST_CASE(multivector_kernel_shape) {
    multivector<double, 2> x, y;
    vector<double> v;
    std::string s = multi_src<assign::SET>(x, std::make_tuple(1, 2.5) * y + sin(v), std::make_index_sequence<2>());
    CHECK(has(s, "extern \"C\" __global__ void vexcl_multivector_kernel"));
    // lhs terminals first (prm_1, prm_2), then component 0's (int, y(0), v), then component 1's (double, y(1), v)
    CHECK(has(s, "double * prm_1") && has(s, "double * prm_2") && has(s, "int prm_3") && has(s, "double prm_6") && !has(s, "prm_9"));
    CHECK(has(s, "double buf_1 = ( ( prm_3 * prm_4[idx] ) + sin( prm_5[idx] ) );"));
    CHECK(has(s, "double buf_2 = ( ( prm_6 * prm_7[idx] ) + sin( prm_8[idx] ) );"));
    CHECK(has(s, "prm_1[idx] = buf_1;") && has(s, "prm_2[idx] = buf_2;"));
    CHECK(s.find("buf_2 = ") < s.find("prm_1[idx] = buf_1;"));
    backend::check_sources(s);

    s = multi_src<assign::MUL>(x, std::tie(y(1), sqr(y(0))), std::make_index_sequence<2>());
    CHECK(has(s, "prm_1[idx] *= buf_1;") && has(s, "double buf_2 = sqr( prm_4[idx] );"));
    CHECK_EQUAL(count(s, "double sqr"), size_t(1));
    backend::check_sources(s);

    using namespace detail;
    typedef SpMat<double, int, int> M;
    static_assert(mv_dim<mv_ref<double, 3>>::value == 3, "");
    static_assert(mv_dim<std::decay<decltype(2 * x + 1)>::type>::value == 2, "");
    static_assert(mv_dim<vector_ref<double>>::value == 0, "");
    typedef additive_operator<M, multivector<double, 2>> AX;
    static_assert(mv_dim<AX>::value == 2 && expr_kind<AX>::value == 1, "");
    static_assert(std::is_same<component_of<1, AX>::type, additive_operator<M, vector<double>>>::value, "");
    static_assert(std::is_same<AX::value_type, double>::value, "");
}

TEST_CASE(reductor_kernels_compile_in_every_order_mode) {             // reductor.hpp:302-439; VEXCL_REDUCTOR_ORDER
    using namespace detail;
    backend::command_queue q;
    vector<double> a, b;
    auto e = a * b;
    for (int mode : {order_tagged, order_release, order_relaxed, order_two_launch}) {
        reductor_order_override() = mode;
        for (int r = 0; r < 3; ++r) {
            std::string s = r == 0 ? Reductor<double, SUM>::source(as_expr<decltype(e)>::get(e), q)
                          : r == 1 ? Reductor<double, SUM_Kahan>::source(as_expr<decltype(e)>::get(e), q)
                                   : Reductor<double, MIN_MAX>::source(as_expr<decltype(e)>::get(e), q);
            CHECK(has(s, "vexcl_reductor_kernel"));
            // the default publishes a partial with a RELEASE arrival and the closing workgroup acquires; round 4's exchange form
            // only on request; the two-launch form has no device-side hand-over at all
            CHECK_EQUAL(count(s, "__builtin_amdgcn_fence(__ATOMIC_RELEASE, \"agent\")"), size_t(mode == order_release ? 2 : 0));
            CHECK_EQUAL(count(s, "__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, \"agent\")"), size_t(mode == order_release ? 3 : 0));
            CHECK_EQUAL(has(s, "__hip_atomic_exchange"), mode == order_relaxed);
            CHECK_EQUAL(has(s, "s_last"), mode != order_two_launch);
            // the default: partials as tagged words, no fence anywhere but the system-scope release of the result
            CHECK_EQUAL(has(s, "vex_publish(g_words") && has(s, "vex_collect(g_words"), mode == order_tagged);
            if (mode == order_tagged) CHECK(!has(s, "__builtin_amdgcn_fence"));
            backend::check_sources(s);
        }
    }
    reductor_order_override() = -1;
}

VEX_FUNCTION(bool, keys_equal, (int, a1)(long, a2)(int, b1)(long, b2), return a1 == b1 && a2 == b2;);
VEX_FUNCTION(double, dplus, (double, x)(double, y), return x + y;);

TEST_CASE(by_key_kernels_compile) {                                   // scan_by_key.hpp / reduce_by_key.hpp sources
    using namespace detail::sbk;
    backend::command_queue q;
    for (scan_mode m : {INCLUSIVE, EXCLUSIVE, REDUCE}) {
        std::string s = source<double, decltype(keys_equal), decltype(dplus)>(q, {"int", "long"}, m);
        CHECK(has(s, "vexcl_sbk_reduce") && has(s, "vexcl_sbk_carry_local") && has(s, "vexcl_sbk_carry(") && has(s, "vexcl_sbk_scan"));
        CHECK(has(s, "keys_equal(pk0, pk1, k0[j], k1[j])") && has(s, "dplus(a.v, b.v)"));
        CHECK_EQUAL(has(s, "okey1[fin.c - 1] = key1[i];"), m == REDUCE);
        // the single-pass form (round 3): look-back kernel + keys-only run count, 3 status words per tile for an 8-byte value
        CHECK(has(s, "vexcl_sbk_lookback") && has(s, "vexcl_sbk_count") && has(s, "#define NW 3"));
        CHECK(has(s, "c += (i0 + j == 0) || !keys_equal(pk0, pk1, k0[j], k1[j])"));
        backend::check_sources(s);
    }
    {   // 4-byte values: 2 status words; a value type the look-back does not carry keeps the three phases only
        std::string f = source<float, equal_fn<int>, plus_fn<float>>(q, {"int"}, INCLUSIVE);
        CHECK(has(f, "vexcl_sbk_lookback") && has(f, "#define NW 2"));
        backend::check_sources(f);
        CHECK(!lookback_value<cl_double2>::value && lookback_value<long>::value && lookback_value<unsigned>::value);
    }
    std::string s = source<int, equal_fn<unsigned>, plus_fn<int>>(q, {"uint"}, EXCLUSIVE);
    CHECK(has(s, "sbk_plus(init, prev.v)"));
    CHECK(!has(s, "vexcl_sbk_pipe"));                                  // the pipelined single pass is not part of the default source
    backend::check_sources(s);
    // VEXCL_SBK_PIPELINE=1 (round 4, experimental): the same single pass with the next tile's elements in flight; every mode compiles
    setenv("VEXCL_SBK_PIPELINE", "1", 1);
    for (scan_mode m : {INCLUSIVE, EXCLUSIVE, REDUCE}) {
        std::string p = source<double, decltype(keys_equal), decltype(dplus)>(q, {"int", "long"}, m);
        CHECK(has(p, "vexcl_sbk_pipe") && has(p, "sbk_fetch_ragged") && has(p, "#define PW 7"));
        backend::check_sources(p);
    }
    backend::check_sources(source<float, equal_fn<int>, plus_fn<float>>(q, {"int"}, INCLUSIVE));
    unsetenv("VEXCL_SBK_PIPELINE");
    // VEXCL_SBK_DPP=1 (round 4, opt-in): the wave scan of the single pass on DPP (row_shr, row_bcast:15 / 31) for 4- and 8-byte values
    setenv("VEXCL_SBK_DPP", "1", 1);
    for (scan_mode m : {INCLUSIVE, EXCLUSIVE, REDUCE}) {
        std::string d = source<double, decltype(keys_equal), decltype(dplus)>(q, {"int", "long"}, m);
        CHECK(has(d, "sbk_dpp<0x142>(T)") && has(d, "sbk_dpp<0x138>(T)") && !has(d, "__shfl_up(T, o, 64)"));
        backend::check_sources(d);
    }
    backend::check_sources(source<float, equal_fn<int>, plus_fn<float>>(q, {"int"}, EXCLUSIVE));
    unsetenv("VEXCL_SBK_DPP");
}

namespace {
struct pair_less_t {                                                  // the reference's tests/sort.cpp:97-103 comparator
    VEX_DUAL_FUNCTOR(bool, (int, a1)(float, a2)(int, b1)(float, b2), return (a1 == b1) ? (a2 < b2) : (a1 < b1);)
};
}

TEST_CASE(merge_sort_kernels_compile) {                              // sort.hpp: detail::msort
    using namespace detail::msort;
    backend::command_queue q;
    pair_less_t pl;
    CHECK(pl(1, 2.0f, 1, 3.0f) && !pl(2, 0.0f, 1, 9.0f));              // the host side of the dual functor
    typedef std::decay<decltype(pl.device)>::type dev;
    std::string s = source<dev>(q, {"int", "float"}, items_per_lane(8));
    CHECK(has(s, "vexcl_msort_block") && has(s, "vexcl_msort_partition") && has(s, "vexcl_msort_merge"));
    CHECK(has(s, "return device(a.k0, a.k1, b.k0, b.k1);") && has(s, "#define VT 8"));
    backend::check_sources(s);
    typedef std::decay<decltype(vex::greater_equal<double>().device)>::type ge;
    CHECK_EQUAL(items_per_lane(200), 1);
    backend::check_sources(source<ge>(q, {"double"}, 2));
}

TEST_CASE(cast_and_temporaries) {                                    // cast.hpp / temporary.hpp
    vector<double> x, y;
    std::string s = src_of<assign::SET>(y, cast<float>(x) * 2 + cast<double>(5));
    CHECK(has(s, "( (float)( prm_2[idx] ) )") && has(s, "( (double)( prm_4 ) )"));
    backend::check_sources(s);

    auto t1 = make_temp<1>(log(x));
    auto t2 = make_temp<2>(t1 + sin(x));
    s = src_of<assign::SET>(y, t1 * t2 + t1);
    // declared once per element block, inner temporary first; its terminal is one parameter
    CHECK_EQUAL(count(s, "double temp_1 = log( prm_temp_1_1[idx] );"), size_t(3));
    CHECK_EQUAL(count(s, "double temp_2 = ( temp_1 + sin( prm_temp_2_1[idx] ) );"), size_t(3));
    CHECK(s.find("double temp_1 = ") < s.find("double temp_2 = "));
    CHECK(has(s, "= ( ( temp_1 * temp_2 ) + temp_1 );"));
    CHECK_EQUAL(count(s, "double * prm_temp_1_1"), size_t(1));
    backend::check_sources(s);

    multivector<double, 2> X, Y;
    auto tm = make_temp<7, double>(tan(X));
    s = multi_src<assign::SET>(Y, tm * tm, std::make_index_sequence<2>());
    // one temporary per component: names 2^20 + 64 * tag + component
    CHECK(has(s, "double temp_1049024 = tan( prm_temp_1049024_1[idx] );") && has(s, "double buf_2 = ( temp_1049025 * temp_1049025 );"));
    backend::check_sources(s);
}

TEST_CASE(slices_and_dimension_reductions) {                          // vector_view.hpp: gslice, slicer, reduce, reshape
    vector<double> x, y;
    size_t dims[2] = {32, 32};
    slicer<2> sl(dims);
    std::string s = src_of<assign::SET>(y, sl[range(2, 2, 11)][_](x));
    CHECK(has(s, "double * prm_2_base") && has(s, "ulong prm_2_start") && has(s, "long prm_2_stride1"));
    CHECK(has(s, "prm_2_base[ (ulong)( prm_2_start + (long)( idx % prm_2_length1 ) * prm_2_stride1 + (long)( idx / prm_2_length1 ) * prm_2_stride0 ) ]"));
    backend::check_sources(s);
    s = src_of<assign::ADD>(sl[5](y), 2 * x);                                    // a slice as the left-hand side
    CHECK(has(s, "prm_1_base[ ") && has(s, " ] += vex_r0;"));
    backend::check_sources(s);
    s = src_of<assign::SET>(y, sl[_][3](x * x) + 1);                             // slice of an expression: idx re-declared
    CHECK(has(s, "const ulong idx = vex_pos;") && has(s, "prm_2_val = ( prm_2_e_1[idx] * prm_2_e_2[idx] );"));
    backend::check_sources(s);
    s = src_of<assign::SET>(y, reduce<MAX>(sl[_], reduce<SUM>(extents[32][32][32], sin(x), 2), 1));
    CHECK(has(s, "for(ulong vex_r0 = 0; vex_r0 < prm_2_rlen0; ++vex_r0)") && has(s, "sin( prm_2_e_1_e_1[idx] )"));
    backend::check_sources(s);
    s = src_of<assign::SET>(y, reshape(x, make_array<size_t>(4, 2), make_array<size_t>(1, 0)));
    backend::check_sources(s);
}

TEST_CASE(random_functions_compile) {                                // random.hpp
    vector<double> x; vector<float> f; vector<cl_uint> u;
    Random<double> rd; RandomNormal<double> rn; Random<float, random::threefry> rf; RandomNormal<float, random::threefry> rnf; Random<cl_uint> ru;
    std::string s = src_of<assign::SET>(x, rd(element_index(), 42) + rn(element_index(), 7));
    CHECK(has(s, "philox_uint_2_10") && has(s, "philox_uint_4_10") && has(s, "random_normal_double_philox( ( prm_4 + idx ), prm_5 )"));
    CHECK_EQUAL(count(s, "__device__ void philox_uint_2_10"), size_t(1));
    backend::check_sources(s);
    backend::check_sources(src_of<assign::SET>(f, rf(element_index(), 1) * rnf(element_index(), 2)));
    backend::check_sources(src_of<assign::SET>(u, ru(element_index(), 3) + ru(element_index(), 4)));
}

TEST_CASE(short_vector_types_compile) {                              // types.hpp; random.hpp with vector outputs
    static_assert(sizeof(cl_float4) == 16 && alignof(cl_float4) == 16 && sizeof(cl_double4) == 32 && sizeof(cl_char2) == 2, "layout");
    static_assert(is_cl_native<cl_int8>::value && is_cl_vector<cl_ulong16>::value && !is_cl_vector<float>::value, "traits");
    static_assert(cl_vector_length<cl_uint16>::value == 16 && std::is_same<cl_scalar_of<cl_short8>::type, short>::value, "traits");
    static_assert(std::is_same<cl_vector_of<float, 4>::type, cl_float4>::value && std::is_same<cl_vector_of<int, 1>::type, int>::value, "vector_of");
    CHECK_EQUAL(type_name<cl_double2>(), std::string("double2"));
    CHECK_EQUAL(type_name<cl_uchar16>(), std::string("uchar16"));
    cl_int4 a = {{1, 2, 3, 4}}, b = {{10, 20, 30, 40}};
    CHECK((a + b) == (cl_int4{{11, 22, 33, 44}}) && (b - a) != a && (2 * a) == (cl_int4{{2, 4, 6, 8}}));
    vector<cl_float4> f4; vector<cl_double4> d4; vector<cl_int8> i8; vector<cl_double2> d2; vector<double> x;
    Random<cl_float4> rf; Random<cl_double4> rd; Random<cl_int8, random::threefry> ri;
    std::string s = src_of<assign::SET>(f4, rf(element_index(), 42) * 2.0f);
    CHECK(has(s, "float4 random_float4_philox") && has(s, "philox_uint_4_10(u.ctr, key)") && has(s, "float4 * prm_1"));
    backend::check_sources(s);
    s = src_of<assign::SET>(d4, rd(element_index(), 42));
    CHECK(has(s, "philox_ulong_4_10") && has(s, "__umul64hi"));
    backend::check_sources(s);
    s = src_of<assign::ADD>(i8, ri(element_index(), 1) + i8);                  // length 8: the extended vectors of the standard header
    CHECK(has(s, "threefry_ulong_4_20") && has(s, "typedef T T##8 __attribute__((ext_vector_type(8)))"));
    backend::check_sources(s);
    backend::check_sources(src_of<assign::SET>(d2, d2 * x + d2));               // short vector with scalar operands
}

// Round 6: which right-hand sides are handed to a product whole (operations.hpp axpby_shape: one vector + one product term, plus / minus /
// negation / scalar factors only) -- decided at compile time, checked here without a device.
TEST_CASE(one_vector_one_product_shapes) {
    using detail::axpby_shape;
    vector<double> x, z; SpMat<double> A;
    // (the expressions are only NAMED -- decltype --, never built: make_inline wants a vector that lives on a device)
#define SHAPE(...) std::make_tuple(axpby_shape<typename std::decay<decltype(__VA_ARGS__)>::type>::ok, axpby_shape<typename std::decay<decltype(__VA_ARGS__)>::type>::vectors, axpby_shape<typename std::decay<decltype(__VA_ARGS__)>::type>::terms)
    CHECK(SHAPE(z - A * x) == std::make_tuple(true, 1, 1));
    CHECK(SHAPE(3 * z + A * x) == std::make_tuple(true, 1, 1));
    CHECK(SHAPE(-z - 2 * (A * x)) == std::make_tuple(true, 1, 1));
    CHECK(SHAPE(0.5 * (A * x) - z * 0.25) == std::make_tuple(true, 1, 1));
    CHECK(SHAPE(x + 2 * make_inline(A * x)) == std::make_tuple(true, 1, 1));
    CHECK(SHAPE(z + x - A * x) == std::make_tuple(true, 2, 1));                 // two vectors: the general route
    CHECK(SHAPE(z - A * x - A * z) == std::make_tuple(true, 1, 2));             // two products
    CHECK(!std::get<0>(SHAPE(sin(z) - A * x)));                                 // a function of a vector
    CHECK(!std::get<0>(SHAPE(z * x + make_inline(A * x))));                     // a product of vectors
    CHECK(!std::get<0>(SHAPE(x * make_inline(A * x))));
    vector<float> xf; SpMat<float> Af;                                          // float matrices: the same shape (the fp32 plane / grid products take the addend too)
    CHECK(SHAPE(xf - Af * xf) == std::make_tuple(true, 1, 1));
#undef SHAPE
    (void)x; (void)z; (void)A; (void)xf; (void)Af;
}

TEST_CASE(partition_and_util) {
    CHECK_EQUAL(alignup(17), size_t(32));
    CHECK_EQUAL(nextpow2(1000), size_t(1024));
    std::vector<size_t> part = {0, 16, 48, 100};
    CHECK_EQUAL(column_owner(0, part), size_t(0));
    CHECK_EQUAL(column_owner(16, part), size_t(1));
    CHECK_EQUAL(column_owner(99, part), size_t(2));
    std::vector<backend::command_queue> one(1), eight(8);
    CHECK_EQUAL(partition(1000, one).back(), size_t(1000));
    auto p = partition(134217728, eight);
    for (size_t d = 0; d <= 8; ++d) CHECK_EQUAL(p[d], size_t(16777216) * d);
    p = partition(1000, eight);
    for (size_t d = 1; d < 8; ++d) CHECK(p[d] % 16 == 0 && p[d] >= p[d - 1]);
    bool thrown = false;
    try { precondition(false, "x"); } catch (const std::runtime_error &) { thrown = true; }
    CHECK(thrown);
}

TEST_CASE(type_names) {
    CHECK_EQUAL(type_name<double>(), std::string("double"));
    CHECK_EQUAL(type_name<size_t>(), std::string("ulong"));
    CHECK_EQUAL(type_name<cl_uint>(), std::string("uint"));
    CHECK_EQUAL(type_name<int *>(), std::string("int *"));
    CHECK_EQUAL(type_name<global_ptr<const double>>(), std::string("const double *"));
}

TEST_CASE(broken_source_raises_vex_error_with_log) {
    bool thrown = false;
    try { backend::check_sources("extern \"C\" __global__ void k(int *x) { x[0] = undeclared_symbol; }"); }
    catch (const vex::error &e) { thrown = std::string(e.what()).find("undeclared_symbol") != std::string::npos; }
    CHECK(thrown);
}
