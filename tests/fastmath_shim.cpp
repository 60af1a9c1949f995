// host build of lart_amd/csrc/lart_math.hpp for tests/test_fastmath_cpu.py (g++, -ffp-contract=off: only the explicit fma()s fuse)
#define LART_HD static inline
#include "../lart_amd/csrc/lart_math.hpp"
extern "C" {
void fm_tan(int n, const double *x, double *y) { for (int i = 0; i < n; i++) y[i] = lartm::tan_halfpi(x[i]); }
void fm_log(int n, const double *x, double *y) { for (int i = 0; i < n; i++) y[i] = lartm::log_pos(x[i]); }
void fm_exp_neg(int n, const double *x, double *y) { for (int i = 0; i < n; i++) y[i] = lartm::exp_neg(x[i]); }
void fm_sincos_2pi(int n, const double *u, double *s, double *c) { for (int i = 0; i < n; i++) lartm::sincos_2pi(u[i], &s[i], &c[i]); }
}
