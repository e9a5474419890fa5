// Host accuracy check of mcmc_amd/csrc/mhd_math.hpp against libm (long double).
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include "../../mcmc_amd/csrc/mhd_math.hpp"
static double ulp_err(double got, long double want) {
    if (want == 0) return std::fabs(got) == 0 ? 0 : 1e9;
    int e; std::frexp((double)want, &e);
    long double ulp = std::ldexp(1.0L, e - 53);
    return (double)(std::fabs((long double)got - want) / ulp);
}
int main() {
    srand48(12345);
    double ms = 0, mc = 0, me = 0, mr = 0, mabs = 0;
    for (int i = 0; i < 2000000; ++i) {
        double x = (drand48() * 2 - 1) * std::pow(10.0, drand48() * 7 - 2);  // |x| in [1e-2, 1e5]
        double s, c; mhd::sincos_pio2(x, s, c);
        long double sl = sinl((long double)x), cl = cosl((long double)x);
        // near zeros of sin/cos judge absolute error instead of ulps
        double es = std::fabs((double)((long double)s - sl)), ec = std::fabs((double)((long double)c - cl));
        if (es > mabs) mabs = es; if (ec > mabs) mabs = ec;
        if (std::fabs((double)sl) > 0.1) { double u = ulp_err(s, sl); if (u > ms) ms = u; }
        if (std::fabs((double)cl) > 0.1) { double u = ulp_err(c, cl); if (u > mc) mc = u; }
        double y = drand48() * 700.0;
        double u = ulp_err(mhd::exp_neg(y), expl(-(long double)y)); if (u > me) me = u;
        double p = std::pow(10.0, drand48() * 20 - 10);
        u = ulp_err(mhd::rsqrt_pos(p), 1.0L / sqrtl((long double)p)); if (u > mr) mr = u;
    }
    printf("sin_ulp %.3f cos_ulp %.3f sincos_abs %.3e exp_ulp %.3f rsqrt_ulp %.3f\n", ms, mc, mabs, me, mr);
    return 0;
}
