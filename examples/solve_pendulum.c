/* The reference's pendulum test case (test_pendulum_5.py:9-25) solved through the C ABI of libgato_hip.so from
 * plain C - the boundary a compiled-language host binds (include/gato_hip.h), no Python anywhere.
 *   make -C examples && ./examples/solve_pendulum
 */
#include <math.h>
#include <stdio.h>

#include "gato_hip.h"

int main(void)
{
    const int G_row[] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14};
    const int G_col[] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13};
    const float G_val[] = {1.f, 1.f, 0.1f, 1.f, 1.f, 0.1f, 1.f, 1.f, 0.1f, 1.f, 1.f, 0.1f, 100.f, 100.f};
    const int C_row[] = {0, 1, 2, 5, 9, 12, 16, 19, 23, 26, 30};
    const int C_col[] = {0, 1, 0, 1, 3, 0, 1, 2, 4, 3, 4, 6, 3, 4, 5, 7, 6, 7, 9, 6, 7, 8, 10, 9, 10, 12, 9, 10, 11, 13};
    const float C_val[] = {1.f, 1.f, -1.f, -0.1f, 1.f, 0.981f, -1.f, -0.1f, 1.f, -1.f, -0.1f, 1.f, 0.981f, -1.f, -0.1f,
                           1.f, -1.f, -0.1f, 1.f, 0.981f, -1.f, -0.1f, 1.f, -1.f, -0.1f, 1.f, 0.981f, -1.f, -0.1f, 1.f};
    const float g[] = {-3.1416f, 0.f, 0.f, -3.1416f, 0.f, 0.f, -3.1416f, 0.f, 0.f, -3.1416f, 0.f, 0.f, -314.159f, 0.f};
    const float c[10] = {0};
    const float lambda_in[10] = {0};
    int S, C, K;
    if (gato_infer_shape(C_row, 11, 14, 10, &S, &C, &K)) { fprintf(stderr, "%s\n", gato_last_error()); return 1; }
    float lambda[10], dz[14], ms[10];
    int iters = -1;
    int rc = gato_linsys_solve_f32(G_row, 15, G_col, G_val, 14, C_row, 11, C_col, C_val, 30, g, 14, c, 10, lambda_in,
                                   S, C, K, /*testiters*/ 10, /*exit_tol*/ 1e-6f, /*max_iters*/ 10, /*warm_start*/ 0,
                                   /*rho*/ 1e-3f, lambda, dz, &iters, ms);
    if (rc) { fprintf(stderr, "gato_linsys_solve_f32: %d %s\n", rc, gato_last_error()); return 2; }
    printf("S=%d C=%d K=%d  first run PCG terminated in %d iterations, time: %f ms\n", S, C, K, iters, ms[0]);
    /* known answers (dense KKT solve with rho) */
    const double lam0 = -203.147040572, dz2 = -32.17244716;
    printf("lambda[0] = %.6f (expected %.6f)   dz[2] = %.6f (expected %.6f)\n", lambda[0], lam0, dz[2], dz2);
    if (fabs(lambda[0] - lam0) > 2e-2 || fabs(dz[2] - dz2) > 1e-2) { fprintf(stderr, "MISMATCH\n"); return 3; }
    printf("Test passed\n");
    return 0;
}
