#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void one(unsigned long long* out) { if (threadIdx.x == 0) { out[0] = wall_clock64(); out[1] = __builtin_amdgcn_s_getreg((20) | (0 << 6) | (31 << 11)); } }
__global__ void many(unsigned long long* out) { if (threadIdx.x == 0) { out[2 * blockIdx.x] = wall_clock64(); out[2 * blockIdx.x + 1] = __builtin_amdgcn_s_getreg((20) | (0 << 6) | (31 << 11)); } }
int main() {
    unsigned long long *a, *b, *c;
    hipMalloc(&a, 16); hipMalloc(&c, 16); hipMalloc(&b, 16 * 2048);
    for (int rep = 0; rep < 3; ++rep) {
        one<<<1, 1024>>>(a); many<<<2048, 128>>>(b); one<<<1, 1024>>>(c);
        hipDeviceSynchronize();
        unsigned long long ha[2], hc[2]; std::vector<unsigned long long> hb(4096);
        hipMemcpy(ha, a, 16, hipMemcpyDeviceToHost); hipMemcpy(hc, c, 16, hipMemcpyDeviceToHost); hipMemcpy(hb.data(), b, 16 * 2048, hipMemcpyDeviceToHost);
        printf("before: %llu (xcc %llu)  after: %llu (xcc %llu)  after-before %lld ticks\n", ha[0], ha[1] & 15, hc[0], hc[1] & 15, (long long)(hc[0] - ha[0]));
        for (int x = 0; x < 8; ++x) {
            unsigned long long mn = ~0ull, mx = 0; int n = 0;
            for (int i = 0; i < 2048; ++i) if ((hb[2 * i + 1] & 15) == (unsigned)x) { mn = hb[2 * i] < mn ? hb[2 * i] : mn; mx = hb[2 * i] > mx ? hb[2 * i] : mx; ++n; }
            if (n) printf("  xcc %d: n %d  min-before %lld  max-before %lld\n", x, n, (long long)(mn - ha[0]), (long long)(mx - ha[0]));
        }
    }
    return 0;
}
