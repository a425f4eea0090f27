// Developer check: the device's log10 (f64) against the host C library's over the unit interval, in units in the last
// place of the host's result; densest near 1, where log10(u) goes through zero.   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

__global__ void k_log10(const double* u, double* out, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = log10(u[i]);
}

static int64_t ulps(double a, double b) {
    int64_t x, y;
    std::memcpy(&x, &a, 8); std::memcpy(&y, &b, 8);
    return x > y ? x - y : y - x;
}

int main() {
    std::vector<double> u;
    uint64_t s = 0x9E3779B97F4A7C15ull;
    auto next = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return (double)(s >> 11) * 0x1.0p-53; };
    for (int k = 0; k < 4000000; ++k) u.push_back(next());                                 // uniform
    for (int k = 0; k < 2000000; ++k) u.push_back(1.0 - next() * std::pow(2.0, -(k % 50))); // towards 1 from below
    for (int k = 0; k < 1000000; ++k) u.push_back(next() * std::pow(2.0, -(k % 60)));        // towards 0
    const int n = (int)u.size();
    double *du, *dout;
    hipMalloc(&du, n * 8); hipMalloc(&dout, n * 8);
    hipMemcpy(du, u.data(), n * 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_log10, dim3((n + 255) / 256), dim3(256), 0, 0, du, dout, n);
    std::vector<double> g(n);
    hipMemcpy(g.data(), dout, n * 8, hipMemcpyDeviceToHost);
    int64_t worst = 0; int worst_i = 0; long differ = 0, over4 = 0;
    for (int i = 0; i < n; ++i) {
        if (!(u[i] > 0.0)) continue;
        const int64_t d = ulps(std::log10(u[i]), g[i]);
        differ += d != 0; over4 += d > 4;
        if (d > worst) { worst = d; worst_i = i; }
    }
    std::printf("n=%d differ=%ld over4=%ld worst=%lld ulp at u=%.17g host=%.17g device=%.17g\n", n, differ, over4, (long long)worst,
                u[worst_i], std::log10(u[worst_i]), g[worst_i]);
    return 0;
}
