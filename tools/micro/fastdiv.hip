// Accuracy of reciprocal-based quotients against IEEE division on gfx950 (v_rcp_f64 / v_rcp_f32 + Newton steps + one residual
// correction): max relative error in ulps over random operands.  Build: hipcc --offload-arch=gfx950 -O3 fastdiv.hip -o fastdiv
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>

template <int NEWTON> __device__ double qd(double n, double d)
{
    if (NEWTON == 3) {           // the product's version: mantissa quotient, exponents by ldexp (gato_pcg_device.h: quotient)
        const double mn = __builtin_amdgcn_frexp_mant(n), md = __builtin_amdgcn_frexp_mant(d);
        const int ex = __builtin_amdgcn_frexp_exp(n) - __builtin_amdgcn_frexp_exp(d);
        double x = __builtin_amdgcn_rcp(md);
        x = fma(x, fma(-md, x, 1.0), x);
        const double q = mn * x;
        return __builtin_amdgcn_ldexp(fma(fma(-md, q, mn), x, q), ex);
    }
    double x = __builtin_amdgcn_rcp(d);
#pragma unroll
    for (int i = 0; i < NEWTON; ++i) { const double e = fma(-d, x, 1.0); x = fma(x, e, x); }
    double q = n * x;
    const double r = fma(-d, q, n);
    return fma(r, x, q);
}
template <int NEWTON> __device__ float qf(float n, float d)
{
    if (NEWTON == 3) {
        const float mn = __builtin_amdgcn_frexp_mantf(n), md = __builtin_amdgcn_frexp_mantf(d);
        const int ex = __builtin_amdgcn_frexp_expf(n) - __builtin_amdgcn_frexp_expf(d);
        const float x = __builtin_amdgcn_rcpf(md);
        const float q = mn * x;
        return __builtin_amdgcn_ldexpf(fmaf(fmaf(-md, q, mn), x, q), ex);
    }
    float x = __builtin_amdgcn_rcpf(d);
#pragma unroll
    for (int i = 0; i < NEWTON; ++i) { const float e = fmaf(-d, x, 1.0f); x = fmaf(x, e, x); }
    float q = n * x;
    const float r = fmaf(-d, q, n);
    return fmaf(r, x, q);
}
__global__ void kd(const double *n, const double *d, double *o0, double *o1, double *o2, int N)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < N) { o0[i] = qd<0>(n[i], d[i]); o1[i] = qd<1>(n[i], d[i]); o2[i] = qd<3>(n[i], d[i]); }
}
__global__ void kf(const float *n, const float *d, float *o0, float *o1, int N)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < N) { o0[i] = qf<0>(n[i], d[i]); o1[i] = qf<3>(n[i], d[i]); }
}
int main()
{
    const int N = 1 << 22;
    std::vector<double> n(N), d(N);
    std::vector<float> nf(N), df(N);
    srand(1);
    for (int i = 0; i < N; ++i) {
        const double sn = (rand() & 1) ? 1 : -1, sd = (rand() & 1) ? 1 : -1;
        n[i] = sn * ldexp(1.0 + rand() / (double)RAND_MAX + rand() / ((double)RAND_MAX * RAND_MAX), rand() % 2040 - 1020);
        d[i] = sd * ldexp(1.0 + rand() / (double)RAND_MAX + rand() / ((double)RAND_MAX * RAND_MAX), rand() % 2040 - 1020);
        nf[i] = (float)(sn * ldexp(1.0 + rand() / (double)RAND_MAX, rand() % 250 - 125));
        df[i] = (float)(sd * ldexp(1.0 + rand() / (double)RAND_MAX, rand() % 250 - 125));
    }
    double *dn, *dd, *o[3]; float *fn, *fd, *of[2];
    hipMalloc(&dn, N * 8); hipMalloc(&dd, N * 8); for (auto &p : o) hipMalloc(&p, N * 8);
    hipMalloc(&fn, N * 4); hipMalloc(&fd, N * 4); for (auto &p : of) hipMalloc(&p, N * 4);
    hipMemcpy(dn, n.data(), N * 8, hipMemcpyHostToDevice); hipMemcpy(dd, d.data(), N * 8, hipMemcpyHostToDevice);
    hipMemcpy(fn, nf.data(), N * 4, hipMemcpyHostToDevice); hipMemcpy(fd, df.data(), N * 4, hipMemcpyHostToDevice);
    kd<<<N / 256, 256>>>(dn, dd, o[0], o[1], o[2], N);
    kf<<<N / 256, 256>>>(fn, fd, of[0], of[1], N);
    std::vector<double> r(N); std::vector<float> rf(N);
    for (int v = 0; v < 3; ++v) {
        hipMemcpy(r.data(), o[v], N * 8, hipMemcpyDeviceToHost);
        double worst = 0; long wrong = 0;
        for (int i = 0; i < N; ++i) {
            const double ex = n[i] / d[i];
            if (!(fabs(ex) > 2.3e-308 && fabs(ex) < 1.7e308)) continue;      // denormal / overflowed quotients: not compared
            const double ulp = fabs(r[i] - ex) / (fabs(ex) * 2.220446049250313e-16);
            if (ulp > worst) worst = ulp;
            if (r[i] != ex) ++wrong;
        }
        printf("f64 variant %d (0/1: rcp + 0/1 Newton + correction on the raw operands; 2: mantissa quotient + ldexp): worst %.3g ulp, %ld of %d differ from IEEE\n", v, worst, wrong, N);
    }
    for (int v = 0; v < 2; ++v) {
        hipMemcpy(rf.data(), of[v], N * 4, hipMemcpyDeviceToHost);
        double worst = 0; long wrong = 0;
        for (int i = 0; i < N; ++i) {
            const float ex = nf[i] / df[i];
            if (!(fabsf(ex) > 1.2e-38f && fabsf(ex) < 3.4e38f)) continue;
            const double ulp = fabs((double)rf[i] - (double)ex) / (fabs((double)ex) * 1.1920929e-7);
            if (ulp > worst) worst = ulp;
            if (rf[i] != ex) ++wrong;
        }
        printf("f32 variant %d (0: rcp + correction on the raw operands; 1: mantissa quotient + ldexp): worst %.3g ulp, %ld of %d differ from IEEE\n", v, worst, wrong, N);
    }
    return 0;
}
