// VALU issue-rate probe for gfx950: measures the sustained rate of the vector
// instructions the renderer's inner loops are built from, so roofline
// denominators come from the chip rather than from a datasheet.
// Build: hipcc -O3 --offload-arch=gfx950 valu_peak.hip -o valu_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>

typedef float f2 __attribute__((ext_vector_type(2)));

#define CHAINS 16
#define ITERS 4096

template <int OP>
__global__ __launch_bounds__(256) void probe(float* out, float seed, int iters) {
    float a[CHAINS]; f2 p[CHAINS]; double d[CHAINS]; uint32_t u[CHAINS];
    float s = seed + threadIdx.x * 1e-3f;
#pragma unroll
    for (int i = 0; i < CHAINS; ++i) { a[i] = s + i; p[i] = f2{s + i, s - i}; d[i] = (double)s + i; u[i] = (uint32_t)(threadIdx.x * 2654435761u + i); }
    float m = 0.999f + seed * 1e-9f, c = 1e-3f + seed * 1e-9f;
    f2 pm = f2{m, m}, pc = f2{c, c};
    double dm = 0.999 + seed * 1e-12, dc = 1e-3;
    uint32_t um = 0x7feb352du + (uint32_t)seed;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < CHAINS; ++i) {
            if (OP == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c));
            if (OP == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(pm), "v"(pc));
            if (OP == 2) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d[i]) : "v"(dm), "v"(dc));
            if (OP == 3) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(u[i]) : "v"(um));
            if (OP == 4) asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(u[i]) : "v"(um), "v"(um));
            if (OP == 5) asm volatile("v_sqrt_f32 %0, %0" : "+v"(a[i]));
            if (OP == 6) asm volatile("v_rsq_f32 %0, %0" : "+v"(a[i]));
            if (OP == 7) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i]));
            if (OP == 8) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(u[i]) : "v"(um));
            if (OP == 9) asm volatile("v_max_f32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
            if (OP == 10) asm volatile("v_cmp_gt_f32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(m) : "vcc");
            if (OP == 11) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(pm));
            if (OP == 12) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(pc));
            if (OP == 13) d[i] = d[i] / dm + dc;           // IEEE f64 divide sequence
            if (OP == 14) d[i] = __builtin_sqrt(d[i]) + dc; // IEEE f64 sqrt sequence
            if (OP == 15) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d[i]) : "v"(dm));
            if (OP == 16) asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[i]) : "v"(dc));
            if (OP == 17) asm volatile("v_lshl_add_u32 %0, %0, 3, %1" : "+v"(u[i]) : "v"(um));
            if (OP == 18) asm volatile("v_cvt_f32_u32 %0, %1" : "=v"(a[i]) : "v"(u[i]));
            if (OP == 19) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(u[i]) : "v"(um));
        }
    }
    float r = 0; double rd = 0; uint32_t ru = 0;
#pragma unroll
    for (int i = 0; i < CHAINS; ++i) { r += a[i] + p[i].x + p[i].y; rd += d[i]; ru ^= u[i]; }
    if (r + (float)rd + (float)ru == 1.2345f) out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

struct Op { const char* name; int flop_per_lane_inst; int inst_per_item; };
static const Op ops[] = {
    {"v_fma_f32", 2, 1}, {"v_pk_fma_f32", 4, 1}, {"v_fma_f64", 2, 1}, {"v_mul_lo_u32", 1, 1},
    {"v_mad_u32_u24", 1, 1}, {"v_sqrt_f32", 1, 1}, {"v_rsq_f32", 1, 1}, {"v_rcp_f32", 1, 1},
    {"v_xor_b32", 1, 1}, {"v_max_f32", 1, 1}, {"v_cmp+v_cndmask", 1, 2}, {"v_pk_mul_f32", 2, 1},
    {"v_pk_add_f32", 2, 1}, {"f64 divide (IEEE)", 1, 1}, {"f64 sqrt (IEEE)", 1, 1}, {"v_mul_f64", 1, 1},
    {"v_add_f64", 1, 1}, {"v_lshl_add_u32", 1, 1}, {"v_cvt_f32_u32", 1, 1}, {"v_mul_hi_u32", 1, 1}};

template <int OP>
static void run(int cus, int waves_per_simd, double clk_ghz) {
    float* out; hipMalloc(&out, 1 << 26);
    int blocks = cus * waves_per_simd;  // 256 threads = 4 waves = 1 per SIMD per block
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    int iters = (OP == 13 || OP == 14) ? ITERS / 8 : ITERS;
    probe<OP><<<blocks, 256>>>(out, 1.0f, iters);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    probe<OP><<<blocks, 256>>>(out, 1.0f, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double lane_items = (double)blocks * 256 * iters * CHAINS;
    double wave_inst = lane_items / 64.0 * ops[OP].inst_per_item;
    double per_simd_cyc = ms * 1e-3 * clk_ghz * 1e9 / (wave_inst / (cus * 4.0));
    printf("%-20s waves/SIMD=%d  %8.3f ms  %9.2f Gitem/s  %8.2f TFLOP/s-equiv  %.2f cyc/wave-inst/SIMD(@%.1fGHz)\n",
           ops[OP].name, waves_per_simd, ms, lane_items / ms * 1e-6, lane_items * ops[OP].flop_per_lane_inst / ms * 1e-9,
           per_simd_cyc, clk_ghz);
    hipFree(out);
}

template <int OP> static void sweep(int cus, double clk) { run<OP>(cus, 1, clk); run<OP>(cus, 2, clk); run<OP>(cus, 4, clk); run<OP>(cus, 8, clk); }

int main() {
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    printf("device %s  CUs=%d  clock=%d kHz  wave=%d  LDS/block=%zu\n", p.name, p.multiProcessorCount, p.clockRate, p.warpSize, p.sharedMemPerBlock);
    int cus = p.multiProcessorCount; double clk = p.clockRate * 1e-6;
    sweep<0>(cus, clk); sweep<1>(cus, clk); sweep<2>(cus, clk); sweep<3>(cus, clk); sweep<4>(cus, clk);
    sweep<5>(cus, clk); sweep<6>(cus, clk); sweep<7>(cus, clk); sweep<8>(cus, clk); sweep<9>(cus, clk);
    sweep<10>(cus, clk); sweep<11>(cus, clk); sweep<12>(cus, clk); sweep<13>(cus, clk); sweep<14>(cus, clk);
    sweep<15>(cus, clk); sweep<16>(cus, clk); sweep<17>(cus, clk); sweep<18>(cus, clk); sweep<19>(cus, clk);
    return 0;
}
