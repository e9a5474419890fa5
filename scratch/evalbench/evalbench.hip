// evalbench.hip -- what one secular evaluation costs on gfx950, in isolation: one or two
// abscissae per lane, one to four waves per SIMD.  Builds against the library's own device code.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o evalbench evalbench.hip && ./evalbench
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

#include "../../mcmc_amd/csrc/mhd_device.hpp"

using namespace mhd;

template <int MODE>
__global__ __launch_bounds__(64) MHD_REGS void bench_kernel(const double* m, int nl, int reps,
                                                            double* out, long long* cycles)
{
    LayerTable& T = g_T;
    __shared__ double mm[MAXNP];
    const int lane = threadIdx.x;
    if (lane < 2 * nl + 1) mm[lane] = m[lane];
    __syncthreads();
    build_layer_table(T, mm, nl, 1.75, lane, nl);
    const double omega = TWOPI / (0.05 + 0.005 * (lane % 40));
    double c = 0.25 + 0.01 * lane, acc = 0.0;
    const long long t0 = clock64();
    for (int r = 0; r < reps; ++r) {
        if (MODE == 1) {
            const SecVal v = eval_secular(MHD_RAYLEIGH, omega, c);
            acc += v.m;
            c += 1.0e-4;
        } else {
            const SecVal2 v = eval_secular2(MHD_RAYLEIGH, omega, c, omega, c + 5.0e-5);
            acc += v.m0 + v.m1;
            c += 1.0e-4;
        }
    }
    const long long t1 = clock64();
    out[blockIdx.x * 64 + lane] = acc;
    if (lane == 0) cycles[blockIdx.x] = t1 - t0;
}

int main()
{
    const int nl = 6;
    const double hm[13] = {0.01, 0.02, 0.04, 0.06, 0.08, 0.10, 0.2, 0.3, 0.45, 0.6, 0.8, 1.0, 1.5};
    double *m, *out;
    long long* cyc;
    const int maxblocks = 4096;
    hipMalloc(&m, sizeof(hm));
    hipMalloc(&out, sizeof(double) * 64 * maxblocks);
    hipMalloc(&cyc, sizeof(long long) * maxblocks);
    hipMemcpy(m, hm, sizeof(hm), hipMemcpyHostToDevice);
    const int reps = 400;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int mode = 1; mode <= 2; ++mode) {
        for (int blocks : {256, 1024, 2048, 4096}) {
            for (int it = 0; it < 2; ++it) {
                hipEventRecord(e0, 0);
                if (mode == 1)
                    hipLaunchKernelGGL(bench_kernel<1>, dim3(blocks), dim3(64), WAVE_LDS_BYTES, 0, m,
                                       nl, reps, out, cyc);
                else
                    hipLaunchKernelGGL(bench_kernel<2>, dim3(blocks), dim3(64), WAVE_LDS_BYTES, 0, m,
                                       nl, reps, out, cyc);
                hipEventRecord(e1, 0);
                hipEventSynchronize(e1);
                float ms = 0;
                hipEventElapsedTime(&ms, e0, e1);
                std::vector<long long> h(blocks);
                hipMemcpy(h.data(), cyc, sizeof(long long) * blocks, hipMemcpyDeviceToHost);
                double mean = 0;
                for (long long v : h) mean += (double)v;
                mean /= blocks;
                const double evals = (double)blocks * 64 * reps * mode;
                if (it == 1)
                    printf("abscissae/lane %d waves %4d: %.3f ms, %.0f cycles per call per wave, "
                           "%.1f G evals/s, %.2f TFLOP/s algorithmic (1063 flop)\n",
                           mode, blocks, ms, mean / reps, evals / ms / 1e6,
                           evals * 1063 / ms / 1e9);
            }
        }
    }
    return 0;
}
