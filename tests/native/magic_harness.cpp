// CPU harness for the dense kernel's row / G by a 32-bit reciprocal (abd_types.hpp: abd_div_magic, abd_div_by_magic,
// abd_div_magic_exact; used by abd_dense.hpp: range_of and abd_create's choice of the dense path).
#include "abd_types.hpp"

// rows near every multiple of G up to n_rows (where a wrong quotient shows first): number of wrong quotients
extern "C" long long magic_sweep(unsigned long long n_rows, unsigned G) {
  const uint32_t magic = abd_div_magic(G);
  long long bad = 0;
  for (unsigned long long q = 0; q * G <= n_rows; ++q)
    for (long long d = -1; d <= 1; ++d) {
      const long long row = (long long)(q * G) + d;
      if (row < 0 || (unsigned long long)row > n_rows) continue;
      if (abd_div_by_magic((uint32_t)row, magic) != (uint32_t)(row / G)) ++bad;
    }
  return bad;
}
extern "C" int magic_exact(unsigned long long n_rows, unsigned G) { return abd_div_magic_exact(n_rows, G) ? 1 : 0; }
