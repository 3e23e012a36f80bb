// refine_accuracy.hip — how far are gfx950's v_rcp_f64 / v_rsq_f64 seeds, and the refinements the trace kernel
// builds on them, from the correctly rounded 1/x, 1/sqrt(x), sqrt(x)?  Errors in ulps of the result, against
// long double on the host.
//   hipcc --offload-arch=gfx950 -O2 -o racer-tracer_amd/build/refine_accuracy tools/microbench/refine_accuracy.hip
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

enum { RCP_SEED, RCP_NEWTON2, RCP_CUBIC, RSQ_SEED, RSQ_GOLD2, RSQ_CUBIC, SQRT_GOLD, SQRT_CUBIC, SQRT_CUBIC_FIX, N_SCHEMES };

__global__ void k(const double *x, double *out, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double v = x[i];
    { // reciprocal
        const double r0 = __builtin_amdgcn_rcp(v);
        out[RCP_SEED * n + i] = r0;
        double r = fma(r0, fma(-v, r0, 1.0), r0);
        r = fma(r, fma(-v, r, 1.0), r);
        out[RCP_NEWTON2 * n + i] = r;
        const double e = fma(-v, r0, 1.0);
        out[RCP_CUBIC * n + i] = fma(r0, fma(e, e, e), r0);
    }
    { // reciprocal square root
        const double y = __builtin_amdgcn_rsq(v);
        out[RSQ_SEED * n + i] = y;
        double g = v * y, h = 0.5 * y;
        double r = fma(-h, g, 0.5);
        g = fma(g, r, g);
        h = fma(h, r, h);
        r = fma(-h, g, 0.5);
        h = fma(h, r, h);
        out[RSQ_GOLD2 * n + i] = h + h;
        const double g0 = v * y;
        const double e = fma(-y, g0, 1.0);       // 1 - x y^2
        const double q = e * fma(e, 0.375, 0.5); // e/2 + 3e^2/8
        out[RSQ_CUBIC * n + i] = fma(y, q, y);
    }
    { // square root
        const double y = __builtin_amdgcn_rsq(v);
        double g = v * y, h = 0.5 * y;
        const double r = fma(-h, g, 0.5);
        g = fma(g, r, g);
        h = fma(h, r, h);
        out[SQRT_GOLD * n + i] = fma(fma(-g, g, v), h, g);
        const double g0 = v * y;
        const double e = fma(-y, g0, 1.0);
        const double q = e * fma(e, 0.375, 0.5);
        const double s = fma(g0, q, g0);
        out[SQRT_CUBIC * n + i] = s;
        out[SQRT_CUBIC_FIX * n + i] = fma(fma(-s, s, v), 0.5 * y, s); // one correction with the unrefined seed
    }
}

__global__ void k_fixup(const double *x, double *out, int n) { // rt_trace_common.h: rcp_f64
    const int i = threadIdx.x;
    if (i >= n) return;
    const double v = x[i];
    const double r0 = __builtin_amdgcn_rcp(v);
    const double e = fma(-v, r0, 1.0);
    out[i] = __builtin_amdgcn_div_fixup(fma(r0, fma(e, e, e), r0), v, 1.0);
}

static double ulps(double got, long double want) {
    int ex;
    frexpl(want, &ex);
    const long double ulp = ldexpl(1.0L, ex - 53);
    return (double)(fabsl((long double)got - want) / ulp);
}

int main() {
    const int n = 1 << 22;
    std::vector<double> x(n);
    srand48(12345);
    for (int i = 0; i < n; ++i) { // mantissas uniform, exponents over the range the kernel sees
        const double m = 1.0 + drand48();
        const int e = (int)(drand48() * 80) - 40;
        x[i] = ldexp(m, e);
    }
    double *dx, *dout;
    CHECK(hipMalloc(&dx, n * sizeof(double)));
    CHECK(hipMalloc(&dout, (size_t)N_SCHEMES * n * sizeof(double)));
    CHECK(hipMemcpy(dx, x.data(), n * sizeof(double), hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k, dim3((n + 255) / 256), dim3(256), 0, 0, dx, dout, n);
    CHECK(hipDeviceSynchronize());
    std::vector<double> out((size_t)N_SCHEMES * n);
    CHECK(hipMemcpy(out.data(), dout, out.size() * sizeof(double), hipMemcpyDeviceToHost));
    const char *names[N_SCHEMES] = {"v_rcp_f64 seed", "rcp: two Newton steps (4 fma)", "rcp: one cubic step (3 fma)",
                                    "v_rsq_f64 seed", "rsqrt: Goldschmidt x2 (8 ops)", "rsqrt: one cubic step (5 ops)",
                                    "sqrt: Goldschmidt + correction (7 ops)", "sqrt: cubic step (5 ops)", "sqrt: cubic step + correction (8 ops)"};
    { // the special arguments rcp_f64's v_div_fixup_f64 is there for: 1/x of +-0, +-inf, NaN, a denormal
        const double special[8] = {0.0, -0.0, INFINITY, -INFINITY, NAN, 4.9406564584124654e-324, 1e-310, 1.7976931348623157e308};
        CHECK(hipMemcpy(dx, special, sizeof special, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(k_fixup, dim3(1), dim3(64), 0, 0, dx, dout, 8);
        CHECK(hipDeviceSynchronize());
        double got[8];
        CHECK(hipMemcpy(got, dout, sizeof got, hipMemcpyDeviceToHost));
        for (int i = 0; i < 8; ++i) printf("rcp with v_div_fixup_f64: 1 / %-24.17g = %-24.17g (IEEE division: %.17g)\n", special[i], got[i], 1.0 / special[i]);
    }
    for (int s = 0; s < N_SCHEMES; ++s) {
        double worst = 0, sum = 0;
        long exact = 0;
        for (int i = 0; i < n; ++i) {
            const long double v = x[i];
            const long double want = s <= RCP_CUBIC ? 1.0L / v : (s <= RSQ_CUBIC ? 1.0L / sqrtl(v) : sqrtl(v));
            const double u = ulps(out[(size_t)s * n + i], want);
            worst = u > worst ? u : worst;
            sum += u;
            exact += out[(size_t)s * n + i] == (double)want;
        }
        printf("%-42s max %12.3f ulp   mean %10.4f ulp   correctly rounded %7.3f %%\n", names[s], worst, sum / n, 100.0 * exact / n);
    }
    return 0;
}
