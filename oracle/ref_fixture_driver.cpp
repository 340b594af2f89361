// Test infrastructure: pins oracle/vex_oracle.c to REFERENCE-EXECUTED code.
//
// This program includes the reference's own generators where they lie -- /root/reference/tests/random_matrix.hpp and
// random_vector.hpp (compiled by oracle/build_ref.sh with -I /root/reference/tests; nothing is copied) -- seeds them with a
// fixed std::srand(), and evaluates every matrix with the host loop the reference's test asserts against
// (/root/reference/tests/spmv.cpp:28-32, :83-85, :108-110, :139-141:  sum += val[j] * x[col[j]]  over j = row[i] .. row[i+1]).
// It writes (row, col, val, x, y, 42*y + y0 ...) for the four matrix shapes of tests/spmv.cpp to one binary file, which
// oracle/ref_fixtures_to_npz.py turns into tests/golden/ref_fixtures.npz (committed).  tests/test_oracle.py then checks
//   * oracle.spmv_csr (and its alpha / append forms) against the y the reference's loop produced, bit for bit, and
//   * that the reference generator's output has the layout properties oracle.random_matrix promises (widths, sorted
//     distinct columns, value range) -- the restatement's generator can not reproduce the stream (std::default_random_engine
//     seeded from rand()), its SHAPE is what the tests rely on.
// usage: ref_fixture_driver out.bin
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include "random_matrix.hpp"        // the reference's (pulls in its random_vector.hpp)

template <typename T> static void put(FILE *f, const std::vector<T> &v, const char *name, const char *type) {
    const uint64_t n = v.size();
    char tag[32] = {0}, ty[8] = {0};
    std::snprintf(tag, sizeof(tag), "%s", name); std::snprintf(ty, sizeof(ty), "%s", type);
    std::fwrite(tag, 1, 32, f); std::fwrite(ty, 1, 8, f); std::fwrite(&n, 8, 1, f);
    if (n) std::fwrite(v.data(), sizeof(T), n, f);
}

template <typename RT, typename CT>
static void one_case(FILE *f, const char *name, size_t n, size_t m, size_t rows_filled) {
    std::vector<RT> row; std::vector<CT> col; std::vector<double> val;
    row.reserve(n + 1);
    random_matrix(rows_filled, m, 16, row, col, val);                  // tests/spmv.cpp:18, :70, :97, :127
    while (row.size() < n + 1) row.push_back(static_cast<RT>(col.size()));   // tests/spmv.cpp:129 (empty_rows)
    std::vector<double> x = random_vector<double>(m), y0 = random_vector<double>(n);
    std::vector<double> y(n), y42(n), yx(n);
    for (size_t idx = 0; idx < n; ++idx) {
        double sum = 0;                                               // tests/spmv.cpp:28-30
        for (size_t j = row[idx]; j < row[idx + 1]; j++)
            sum += val[j] * x[col[j]];
        y[idx] = sum;
        y42[idx] = y0[idx] + 42 * sum;                                // the value Y += 42 * (A * X) is checked against (:44-52), on top of y0
        yx[idx] = (idx < m ? x[idx] : 0.0) + sum;                     // Y = X + A * X (:54-58), square cases
    }
    std::vector<int64_t> r64(row.begin(), row.end()), c64(col.begin(), col.end());
    std::vector<int64_t> shape = {(int64_t)n, (int64_t)m, (int64_t)rows_filled, (int64_t)sizeof(RT), (int64_t)sizeof(CT)};
    char key[32];
#define PUT(vec, suffix, type) std::snprintf(key, sizeof(key), "%s_%s", name, suffix); put(f, vec, key, type)
    PUT(shape, "shape", "i8"); PUT(r64, "row", "i8"); PUT(c64, "col", "i8"); PUT(val, "val", "f8");
    PUT(x, "x", "f8"); PUT(y0, "y0", "f8"); PUT(y, "y", "f8"); PUT(y42, "y42", "f8"); PUT(yx, "yx", "f8");
#undef PUT
}

int main(int argc, char **argv) {
    if (argc < 2) { std::fprintf(stderr, "usage: %s out.bin\n", argv[0]); return 2; }
    std::srand(20250711);                                             // the reference seeds from time(0) (tests/context_setup.hpp:19-22)
    FILE *f = std::fopen(argv[1], "wb");
    if (!f) return 1;
    one_case<size_t, size_t>(f, "square", 1024, 1024, 1024);          // vector_product, tests/spmv.cpp:10-59
    one_case<size_t, size_t>(f, "nonsquare", 1024, 2048, 1024);       // non_square_matrix, :61-87
    one_case<unsigned, int>(f, "types", 1024, 1024, 1024);            // non_default_types, :89-114
    one_case<size_t, size_t>(f, "emptyrows", 1024, 1024, 256);        // empty_rows, :116-146
    std::fclose(f);
    return 0;
}
