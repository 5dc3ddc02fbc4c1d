// Shader cycles per wave64 float64 instruction per SIMD on gfx950, measured inside the kernel (s_memtime around an
// unrolled block of independent chains, as tools/microbench/valu_cycles.hip does for FP32), and the cost of the IEEE
// float64 division and square root as the compiler expands them (what the ray-tree kernel executes).
// Build: hipcc -O3 --offload-arch=gfx950 f64_cycles.hip -o f64_cycles
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <vector>

#define CHAINS 8
#define ITERS 4096

template <int OP>
__global__ __launch_bounds__(256) void probe(unsigned long long* out, double seed, int iters) {
    double a[CHAINS];
#pragma unroll
    for (int i = 0; i < CHAINS; ++i) a[i] = seed + threadIdx.x * 1e-3 + i;
    const double m = 0.999 + seed * 1e-12, c = 1e-3 + seed * 1e-12;
    unsigned long long t0, t1;
    __syncthreads();
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < CHAINS; ++i) {
            if (OP == 0) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c));
            if (OP == 1) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a[i]) : "v"(m));
            if (OP == 2) asm volatile("v_add_f64 %0, %0, %1" : "+v"(a[i]) : "v"(c));
            if (OP == 3) asm volatile("v_rcp_f64 %0, %0" : "+v"(a[i]));
            if (OP == 4) asm volatile("v_rsq_f64 %0, %0" : "+v"(a[i]));
            if (OP == 5) { a[i] = c / a[i]; asm volatile("" : "+v"(a[i])); }        // IEEE division, compiler's expansion
            if (OP == 6) { a[i] = sqrt(a[i] + 2.0); asm volatile("" : "+v"(a[i])); }  // IEEE square root (+ one add)
            if (OP == 7) asm volatile("v_max_f64 %0, %0, %1" : "+v"(a[i]) : "v"(m));
        }
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory");
    double r = 0;
#pragma unroll
    for (int i = 0; i < CHAINS; ++i) r += a[i];
    if ((threadIdx.x & 63) == 0) out[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = (t1 - t0) | (r == 1.2345 ? 1ull << 63 : 0ull);
}

static const char* names[] = {"v_fma_f64", "v_mul_f64", "v_add_f64", "v_rcp_f64", "v_rsq_f64", "a / b (IEEE, expanded)", "sqrt(a) (IEEE, expanded) + add", "v_max_f64"};

template <int OP>
static void run(int cus) {
    for (int w : {1, 2, 4}) {
        const int blocks = cus * w, waves = blocks * 4;
        unsigned long long* d; (void)hipMalloc(&d, 8 * waves);
        std::vector<unsigned long long> h(waves);
        probe<OP><<<blocks, 256>>>(d, 1.0, ITERS);
        probe<OP><<<blocks, 256>>>(d, 1.0, ITERS);
        (void)hipDeviceSynchronize();
        (void)hipMemcpy(h.data(), d, 8 * waves, hipMemcpyDeviceToHost);
        std::vector<double> per;
        for (auto v : h) per.push_back((double)(v & ~(1ull << 63)) / ((double)ITERS * CHAINS));
        std::sort(per.begin(), per.end());
        printf("%-32s waves/SIMD=%d  per wave: median %.1f cycles per operation -> per SIMD %.1f\n", names[OP], w, per[per.size() / 2], per[per.size() / 2] / w);
        (void)hipFree(d);
    }
}

int main() {
    hipDeviceProp_t p; (void)hipGetDeviceProperties(&p, 0);
    printf("device %s  CUs=%d\n", p.name, p.multiProcessorCount);
    const int cus = p.multiProcessorCount;
    run<0>(cus); run<1>(cus); run<2>(cus); run<7>(cus); run<3>(cus); run<4>(cus); run<5>(cus); run<6>(cus);
    return 0;
}
