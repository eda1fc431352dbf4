/* ORACLE - test infrastructure, not product code.  Included twice by gato_oracle.c with
 * REAL = float / double and SUF = f32 / f64.
 *
 * Plain-C restatement of the reference hot path, loop for loop in the reference's
 * accumulation order (sequential "res += a*b" over the same index order), so that an fp32
 * run tracks the CUDA path as closely as a CPU can.  File:line citations are relative to
 * /root/reference.  Deviations D1-D4 of SURVEY.md section 2.3 are applied and marked.
 * Parity pin: see oracle/gato_oracle.py header (pendulum fixture + dense KKT solve).
 */

#define CAT_(a, b) a##_##b
#define CAT(a, b) CAT_(a, b)
#define FN(name) CAT(name, SUF)

/* ---- A10: Gauss-Jordan, no pivoting (src/gato_utils.cuh:468-586).  A is n x n col-major,
 * overwritten; Ainv receives the inverse.  Update rule of the 3-matrix overload (:551-585):
 * pivot row /= pv ; other rows -= (col[row]/pv) * rowOld[col]. */
static void FN(gj_inverse)(REAL *A, REAL *Ainv, int n, REAL *tmp /* 2n */)
{
    for (int i = 0; i < n * n; ++i) Ainv[i] = (REAL)((i % n) == (i / n));
    REAL *colv = tmp, *rowA = tmp + n;
    REAL rowI[64];
    for (int p = 0; p < n; ++p) {
        REAL pv = A[p + p * n];
        for (int r = 0; r < n; ++r) colv[r] = A[r + p * n];
        for (int c = 0; c < n; ++c) { rowA[c] = A[p + c * n]; rowI[c] = Ainv[p + c * n]; }
        for (int c = 0; c < n; ++c) {
            for (int r = 0; r < n; ++r) {
                if (r == p) { A[r + c * n] /= pv; Ainv[r + c * n] /= pv; }
                else {
                    REAL f = colv[r] / pv;
                    A[r + c * n] -= f * rowA[c];
                    Ainv[r + c * n] -= f * rowI[c];
                }
            }
        }
    }
}

/* out(m x n) = A(m x k) * B(k x n), all col-major (mat_mat_prod, src/gato_utils.cuh:609-633) */
static void FN(mm)(REAL *out, const REAL *A, const REAL *B, int m, int k, int n)
{
    for (int c = 0; c < n; ++c)
        for (int r = 0; r < m; ++r) {
            REAL res = 0;
            for (int t = 0; t < k; ++t) res += A[t * m + r] * B[c * k + t];
            out[c * m + r] = res;
        }
}
/* out(m x n) = A(m x k) * B(n x k)^T (transposeB branch, :635-658) */
static void FN(mmT)(REAL *out, const REAL *A, const REAL *B, int m, int k, int n)
{
    for (int c = 0; c < n; ++c)
        for (int r = 0; r < m; ++r) {
            REAL res = 0;
            for (int t = 0; t < k; ++t) res += A[t * m + r] * B[t * n + c];
            out[c * m + r] = res;
        }
}
/* out(m) = A(m x n) * x(n)  (mat_vec_prod, :595-606) */
static void FN(mv)(REAL *out, const REAL *A, const REAL *x, int m, int n)
{
    for (int r = 0; r < m; ++r) {
        REAL res = 0;
        for (int c = 0; c < n; ++c) res += A[r + c * m] * x[c];
        out[r] = res;
    }
}
/* out(n) = A(m x n)^T * x(m)  (gato_ATx, :664-679) */
static void FN(mTv)(REAL *out, const REAL *A, const REAL *x, int m, int n)
{
    for (int i = 0; i < n; ++i) {
        REAL res = 0;
        for (int t = 0; t < m; ++t) res += A[i * m + t] * x[t];
        out[i] = res;
    }
}

/* ---- A1: CSR -> dense (src/gato_schur.cuh:674-743).  Outputs must be pre-zeroed. */
void FN(gato_oracle_convert)(const int *G_row, const int *G_col, const REAL *G_val,
                             const int *C_row, const int *C_col, const REAL *C_val,
                             int S, int C, int K, REAL rho, REAL *Gd, REAL *Cd)
{
    const int n = S + C, N = n * K - C, SS = S * S, CC = C * C, SC = S * C;
    memset(Gd, 0, sizeof(REAL) * ((size_t)(SS + CC) * K - CC));
    memset(Cd, 0, sizeof(REAL) * ((size_t)(SS + SC) * (K - 1)));
    for (int row = 0; row < N; ++row) {
        int in_set_row = row % n;
        size_t set_offset = (size_t)(row / n) * (SS + CC);
        for (int it = G_row[row]; it < G_row[row + 1]; ++it) {
            int col = G_col[it], in_set_col = col % n;
            REAL v = G_val[it] + (REAL)(col == row) * rho;             /* :697,:700 */
            if (in_set_col < S) Gd[set_offset + in_set_col * S + in_set_row] = v;
            else Gd[set_offset + SS + (in_set_col - S) * C + (in_set_row - S)] = v;
        }
    }
    for (int row = S; row < S * K; ++row) {                            /* :723-725 */
        int block_row = row / S - 1;
        for (int it = C_row[row]; it < C_row[row + 1]; ++it) {
            int col = C_col[it];
            if (col / n > block_row) continue;                         /* :735 */
            Cd[(size_t)block_row * (SS + SC) + (col % n) * S + row % S] = C_val[it];
        }
    }
}

/* ---- A2: Schur + block-Jacobi (src/gato_schur.cuh:13-460).  Ginv gets Q^-1/R^-1 in G_dense
 * layout (separate buffer: D3). */
void FN(gato_oracle_form_schur)(const REAL *Gd, const REAL *Cd, const REAL *g, const REAL *c,
                                int S, int C, int K, REAL *Sbd, REAL *Pbd, REAL *gamma, REAL *Ginv)
{
    const int n = S + C, SS = S * S, CC = C * C, SC = S * C;
    const size_t gst = SS + CC, cst = SS + SC;
    memset(Sbd, 0, sizeof(REAL) * 3 * SS * (size_t)K);
    memset(Pbd, 0, sizeof(REAL) * 3 * SS * (size_t)K);
    /* every Q_k, R_k inverted exactly once */
#pragma omp parallel
    {
        REAL *w = (REAL *)malloc(sizeof(REAL) * (SS + 2 * S + 2));
#pragma omp for schedule(static)
        for (int k = 0; k < K; ++k) {
            memcpy(w, Gd + k * gst, sizeof(REAL) * SS);
            FN(gj_inverse)(w, Ginv + k * gst, S, w + SS);
            if (k < K - 1) {
                memcpy(w, Gd + k * gst + SS, sizeof(REAL) * CC);
                FN(gj_inverse)(w, Ginv + k * gst + SS, C, w + SS);
            }
        }
        free(w);
    }
    /* k = 0 (:26-147) */
    for (int i = 0; i < SS; ++i) {
        Pbd[SS + i] = -Gd[i];                                          /* :75-81 */
        Sbd[SS + i] = -Ginv[i];                                        /* :120-126 */
    }
    {
        REAL t[64];
        FN(mv)(t, Ginv, g, S, S);
        for (int i = 0; i < S; ++i) gamma[i] = c[i] - t[i];            /* :131-146 (+c_0: D4) */
    }
#pragma omp parallel
    {
        REAL *phi = (REAL *)malloc(sizeof(REAL) * (5 * SS + 6 * S + 2));
        REAL *BR = phi + SS, *theta = BR + SS, *tmp = theta + SS, *thinv = tmp + SS, *v = thinv + SS;
#pragma omp for schedule(static)
        for (int k = 1; k < K; ++k) {
            const REAL *A = Cd + (k - 1) * cst, *B = A + SS;
            const REAL *Qim = Ginv + (k - 1) * gst, *Rim = Qim + SS, *Qik = Ginv + k * gst;
            const REAL *qm = g + (size_t)(k - 1) * n, *rm = qm + S, *qk = g + (size_t)k * n;
            FN(mm)(phi, A, Qim, S, S, S);                              /* :277-285 */
            FN(mm)(BR, B, Rim, S, C, C);                               /* :293-301 */
            REAL *gt = v, *t1 = v + S, *t2 = v + 2 * S;
            FN(mv)(gt, Qik, qk, S, S);                                 /* :306-310 */
            for (int i = 0; i < S; ++i) gt[i] -= c[k * S + i];         /* :311-313 */
            FN(mv)(t1, phi, qm, S, S);                                 /* :316-320 */
            FN(mv)(t2, BR, rm, S, C);                                  /* :324-328 */
            for (int i = 0; i < S; ++i) gt[i] += t2[i] + t1[i];        /* :336-338 */
            FN(mmT)(theta, phi, A, S, S, S);                           /* :342-351 */
            for (int i = 0; i < SS; ++i) theta[i] += Qik[i];           /* :362-364 */
            FN(mmT)(tmp, BR, B, S, C, S);                              /* :368-377 */
            for (int i = 0; i < SS; ++i) theta[i] += tmp[i];           /* :382-384 */
            REAL *Sk = Sbd + (size_t)k * 3 * SS, *Pk = Pbd + (size_t)k * 3 * SS;
            for (int i = 0; i < SS; ++i) { Sk[i] = -phi[i]; Sk[SS + i] = -theta[i]; }  /* :388-404 */
            memcpy(tmp, theta, sizeof(REAL) * SS);
            FN(gj_inverse)(tmp, thinv, S, v + 3 * S);                  /* :407-414 */
            for (int i = 0; i < SS; ++i) Pk[SS + i] = -thinv[i];       /* :415-422 */
            for (int i = 0; i < S; ++i) gamma[k * S + i] = -gt[i];     /* :435-438 */
            REAL *Srm = Sbd + (size_t)(k - 1) * 3 * SS + 2 * SS;       /* :443-455  S[k-1].right = -phi^T */
            for (int cc = 0; cc < S; ++cc)
                for (int rr = 0; rr < S; ++rr) Srm[cc * S + rr] = -phi[rr * S + cc];
        }
        free(phi);
    }
}

/* ---- A3: symmetric stair (src/gato_schur.cuh:497-649) */
void FN(gato_oracle_form_ss)(const REAL *Sbd, REAL *Pbd, int S, int K)
{
    const int SS = S * S;
#pragma omp parallel
    {
        REAL *t = (REAL *)malloc(sizeof(REAL) * 3 * SS);
        REAL *phT = t + SS, *o = phT + SS;
#pragma omp for schedule(static)
        for (int k = 0; k < K; ++k) {
            const REAL *Pm = Pbd + (size_t)k * 3 * SS + SS;
            REAL *Pk = Pbd + (size_t)k * 3 * SS;
            if (k > 0) {                                               /* :578-611 */
                FN(mm)(t, Pm, Sbd + (size_t)k * 3 * SS, S, S, S);
                FN(mm)(o, t, Pbd + (size_t)(k - 1) * 3 * SS + SS, S, S, S);
                for (int i = 0; i < SS; ++i) Pk[i] = -o[i];
            }
            if (k < K - 1) {                                           /* :614-648 (D1: k<K-1 only) */
                const REAL *Sl1 = Sbd + (size_t)(k + 1) * 3 * SS;
                for (int cc = 0; cc < S; ++cc)
                    for (int rr = 0; rr < S; ++rr) phT[cc * S + rr] = Sl1[rr * S + cc];
                FN(mm)(t, Pm, phT, S, S, S);
                FN(mm)(o, t, Pbd + (size_t)(k + 1) * 3 * SS + SS, S, S, S);
                for (int i = 0; i < SS; ++i) Pk[2 * SS + i] = -o[i];
            }
        }
        free(t);
    }
}

/* ---- A7: y_k = [L M R]_k [x_{k-1}; x_k; x_{k+1}] with the K4 boundary rule
 * (src/gato_utils.cuh:153-185): first row main,right; last row left,main. */
static inline void FN(bt_row)(REAL *y, const REAL *Mk, const REAL *x, int S, int k, int K)
{
    const int c0 = (k == 0) ? S : 0, c1 = (k == K - 1) ? 2 * S : 3 * S;
    const REAL *xw = x + (size_t)(k - 1) * S;
    for (int r = 0; r < S; ++r) {
        REAL val = 0;
        for (int cc = c0; cc < c1; ++cc) val += Mk[S * cc + r] * xw[cc];
        y[r] = val;
    }
}

/* ---- A5/A8: PCG, K4 semantics (src/gato_pcg.cuh:270-439).  Returns iters.  Dots are summed
 * per knot with reducePlus' tree (src/gato_utils.cuh:253-275) then over knots in index order
 * (the reference's atomicAdd order is unspecified). */
static REAL FN(knot_dot)(const REAL *a, const REAL *b, int S)
{
    REAL t[64];
    for (int i = 0; i < S; ++i) t[i] = a[i] * b[i];
    int left = S;
    while (left > 3) {
        int odd = left % 2;
        left = (left - odd) / 2;
        for (int i = 0; i < left; ++i) t[i] += t[i + left];
        if (odd) t[0] += t[2 * left];
    }
    for (int i = 1; i < left; ++i) t[0] += t[i];
    return t[0];
}

int FN(gato_oracle_pcg)(const REAL *Sbd, const REAL *Pbd, const REAL *gamma, int S, int K,
                        REAL exit_tol, int max_iters, REAL *lambda, REAL *eta_hist /* max_iters+1 or NULL */)
{
    const size_t SK = (size_t)S * K;
    const int SS3 = 3 * S * S;
    REAL *r = (REAL *)malloc(sizeof(REAL) * SK * 4), *p = r + SK, *rt = p + SK, *ups = rt + SK;
    REAL *kd = (REAL *)malloc(sizeof(REAL) * K);
    REAL eta = 0, eta_new = 0;
    int iters = max_iters;
    for (size_t i = 0; i < SK; ++i) { r[i] = gamma[i]; lambda[i] = 0; }
#pragma omp parallel for schedule(static)
    for (int k = 0; k < K; ++k) {
        FN(bt_row)(rt + (size_t)k * S, Pbd + (size_t)k * SS3, r, S, k, K);
        for (int i = 0; i < S; ++i) p[(size_t)k * S + i] = rt[(size_t)k * S + i];
        kd[k] = FN(knot_dot)(r + (size_t)k * S, rt + (size_t)k * S, S);
    }
    for (int k = 0; k < K; ++k) eta += kd[k];
    if (eta_hist) eta_hist[0] = eta;
    for (int it = 0; it < max_iters; ++it) {
        REAL v = 0;
#pragma omp parallel for schedule(static)
        for (int k = 0; k < K; ++k) {
            FN(bt_row)(ups + (size_t)k * S, Sbd + (size_t)k * SS3, p, S, k, K);
            kd[k] = FN(knot_dot)(p + (size_t)k * S, ups + (size_t)k * S, S);
        }
        for (int k = 0; k < K; ++k) v += kd[k];
        REAL alpha = eta / v;
#pragma omp parallel for schedule(static)
        for (size_t i = 0; i < SK; ++i) { lambda[i] += alpha * p[i]; r[i] -= alpha * ups[i]; }
        eta_new = 0;
#pragma omp parallel for schedule(static)
        for (int k = 0; k < K; ++k) {
            FN(bt_row)(rt + (size_t)k * S, Pbd + (size_t)k * SS3, r, S, k, K);
            kd[k] = FN(knot_dot)(r + (size_t)k * S, rt + (size_t)k * S, S);
        }
        for (int k = 0; k < K; ++k) eta_new += kd[k];
        if (eta_hist) eta_hist[it + 1] = eta_new;
        if ((eta_new < 0 ? -eta_new : eta_new) < exit_tol) { iters = it; break; }   /* :404-411 */
        REAL beta = eta_new / eta;
#pragma omp parallel for schedule(static)
        for (size_t i = 0; i < SK; ++i) p[i] = rt[i] + beta * p[i];
        eta = eta_new;
    }
    free(r); free(kd);
    return iters;
}

/* ---- A9: dz back-substitution (src/gato_schur.cuh:758-867; D2: last state row has no A / lambda_K) */
void FN(gato_oracle_compute_dz)(const REAL *Ginv, const REAL *Cd, const REAL *g, const REAL *lambda,
                                int S, int C, int K, REAL *dz)
{
    const int n = S + C, SS = S * S, CC = C * C, SC = S * C;
#pragma omp parallel for schedule(static)
    for (int k = 0; k < K; ++k) {
        REAL t[64], u[64];
        const REAL *Qi = Ginv + (size_t)k * (SS + CC);
        if (k < K - 1) {
            const REAL *A = Cd + (size_t)k * (SS + SC), *B = A + SS;
            FN(mTv)(t, A, lambda + (size_t)(k + 1) * S, S, S);          /* :833-838 */
            for (int i = 0; i < S; ++i) t[i] = lambda[(size_t)k * S + i] + t[i];       /* :841-845 */
            for (int i = 0; i < S; ++i) t[i] = g[(size_t)k * n + i] - t[i];            /* :848-852 */
            FN(mTv)(u, B, lambda + (size_t)(k + 1) * S, S, C);          /* :784-789 */
            for (int i = 0; i < C; ++i) u[i] = g[(size_t)k * n + S + i] - u[i];        /* :792-796 */
            FN(mv)(dz + (size_t)k * n + S, Qi + SS, u, C, C);           /* :799-808 */
        } else {
            for (int i = 0; i < S; ++i) t[i] = g[(size_t)k * n + i] - lambda[(size_t)k * S + i];
        }
        FN(mv)(dz + (size_t)k * n, Qi, t, S, S);                        /* :856-865 */
    }
}

/* ---- L3: whole solve (gpu_library.cu:25-83).  Returns iters. */
int FN(gato_oracle_linsys)(const int *G_row, const int *G_col, const REAL *G_val,
                           const int *C_row, const int *C_col, const REAL *C_val,
                           const REAL *g, const REAL *c, int S, int C, int K,
                           REAL exit_tol, int max_iters, REAL rho, REAL *lambda, REAL *dz)
{
    const size_t SS = (size_t)S * S, CC = (size_t)C * C, SC = (size_t)S * C;
    const size_t gsz = (SS + CC) * K - CC, csz = (SS + SC) * (K - 1);
    REAL *Gd = (REAL *)malloc(sizeof(REAL) * (2 * gsz + csz + 6 * SS * K + (size_t)S * K + 16));
    REAL *Gi = Gd + gsz, *Cd = Gi + gsz, *Sbd = Cd + csz, *Pbd = Sbd + 3 * SS * K, *gam = Pbd + 3 * SS * K;
    FN(gato_oracle_convert)(G_row, G_col, G_val, C_row, C_col, C_val, S, C, K, rho, Gd, Cd);
    FN(gato_oracle_form_schur)(Gd, Cd, g, c, S, C, K, Sbd, Pbd, gam, Gi);
    FN(gato_oracle_form_ss)(Sbd, Pbd, S, K);
    int iters = FN(gato_oracle_pcg)(Sbd, Pbd, gam, S, K, exit_tol, max_iters, lambda, NULL);
    FN(gato_oracle_compute_dz)(Gi, Cd, g, lambda, S, C, K, dz);
    free(Gd);
    return iters;
}

#undef FN
#undef CAT
#undef CAT_
