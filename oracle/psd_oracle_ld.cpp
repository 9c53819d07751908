// ORACLE — TEST INFRASTRUCTURE ONLY.  Extended-precision build of the CPU restatement (the
// reference runs its own tests for BigFloat as well — test/runtests.jl:89, test/generalized.jl:68,201 — because its code is
// generic in the element type).  The restatement is plain C++ on `double`; this translation unit compiles the SAME
// sources once more with `long double` (x87 extended precision, eps = 2^-63 = 1.08e-19) substituted for `double`, in a
// namespace of its own, and exports every entry point of psd_oracle.cpp under the prefix psdo_ld_.  Matrix arguments
// are numpy `longdouble` arrays (16-byte storage per element, complex = two of them).  It checks that the restated
// algorithm is generic and that its accuracy follows the working precision (tests/test_oracle_extended.py); nothing in
// the product is built from it.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <complex>
#include <cstdint>
#include <cstring>
#include <limits>
#include <vector>
#ifdef _OPENMP
#include <omp.h>
#endif

#define double long double
#define psdo psdo_ld
#define psdo_d_gpschur_hess psdo_ld_d_gpschur_hess
#define psdo_d_hessenberg_q psdo_ld_d_hessenberg_q
#define psdo_d_ordschur psdo_ld_d_ordschur
#define psdo_d_ordschur_real1x1 psdo_ld_d_ordschur_real1x1
#define psdo_d_phessenberg psdo_ld_d_phessenberg
#define psdo_d_phessenberg_cols psdo_ld_d_phessenberg_cols
#define psdo_d_pschur psdo_ld_d_pschur
#define psdo_d_pschur_hess psdo_ld_d_pschur_hess
#define psdo_dbg_dustpos psdo_ld_dbg_dustpos
#define psdo_dbg_maxdust psdo_ld_dbg_maxdust
#define psdo_get_max_threads psdo_ld_get_max_threads
#define psdo_gordschur psdo_ld_gordschur
#define psdo_gpschur psdo_ld_gpschur
#define psdo_rphessenberg psdo_ld_rphessenberg
#define psdo_set_sweep_cap psdo_ld_set_sweep_cap
#define psdo_set_threads psdo_ld_set_threads
#define psdo_sg_phessenberg psdo_ld_sg_phessenberg
#define psdo_z_ordschur psdo_ld_z_ordschur
#define psdo_z_phessenberg psdo_ld_z_phessenberg
#define psdo_z_pschur psdo_ld_z_pschur
#define psdo_z_pschur_hess psdo_ld_z_pschur_hess

#include "psd_oracle.cpp"
