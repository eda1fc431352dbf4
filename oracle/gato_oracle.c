/* ORACLE - test infrastructure, not product code.
 * C restatement of the reference hot path in fp32 and fp64 (see gato_oracle_impl.h).
 * Built by oracle/Makefile into oracle/libgato_oracle.so; used by tests/ as the checker and
 * by bench.py's cpu_baseline leg ("kind": "port") only. */
#include <stdlib.h>
#include <string.h>

#define REAL float
#define SUF f32
#include "gato_oracle_impl.h"
#undef REAL
#undef SUF

#define REAL double
#define SUF f64
#include "gato_oracle_impl.h"
#undef REAL
#undef SUF

#ifdef _OPENMP
#include <omp.h>
int gato_oracle_max_threads(void) { return omp_get_max_threads(); }
void gato_oracle_set_threads(int n) { omp_set_num_threads(n); }
#else
int gato_oracle_max_threads(void) { return 1; }
void gato_oracle_set_threads(int n) { (void)n; }
#endif
