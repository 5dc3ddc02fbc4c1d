// In-kernel cycle probe for gfx950: how many shader cycles does one SIMD need per wave64 vector
// instruction?  Each wave stamps s_memtime (shader clock) and s_memrealtime (100 MHz) around an unrolled
// block of independent instructions, so the answer does not depend on host timing, launch overhead or an
// assumed clock.  With w waves resident per SIMD, all running the same block over the same interval, the
// SIMD issued w * INSTS instructions in (last end - first start) cycles.
// Build: hipcc -O3 --offload-arch=gfx950 valu_cycles.hip -o valu_cycles
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <map>
#include <vector>

typedef float f2 __attribute__((ext_vector_type(2)));

#define CHAINS 16
#define ITERS 8192

struct Stamp { unsigned long long t0, t1, r0, r1; uint32_t hw_id, xcc; };

template <int OP>
__global__ __launch_bounds__(256) void probe(Stamp* out, float seed, int iters) {
    float a[CHAINS]; f2 p[CHAINS]; uint32_t u[CHAINS];
    float s = seed + threadIdx.x * 1e-3f;
#pragma unroll
    for (int i = 0; i < CHAINS; ++i) { a[i] = s + i; p[i] = f2{s + i, s - i}; u[i] = (uint32_t)(threadIdx.x * 2654435761u + i); }
    float m = 0.999f + seed * 1e-9f, c = 1e-3f + seed * 1e-9f;
    f2 pm = f2{m, m}, pc = f2{c, c};
    uint32_t um = 0x7feb352du + (uint32_t)seed;
    unsigned long long t0, t1, r0, r1;
    __syncthreads();
    asm volatile("s_memrealtime %0\n\ts_memtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(r0), "=s"(t0) :: "memory");
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < CHAINS; ++i) {
            if (OP == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c));
            if (OP == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(pm), "v"(pc));
            if (OP == 2) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(u[i]) : "v"(um));
            if (OP == 3) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(u[i]) : "v"(um));
            if (OP == 4) asm volatile("v_cvt_f32_u32 %0, %1" : "=v"(a[i]) : "v"(u[i]));
            if (OP == 5) asm volatile("v_max_f32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
            if (OP == 6) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
            if (OP == 7) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
            if (OP == 8) asm volatile("v_lshrrev_b32 %0, 1, %0" : "+v"(u[i]));
            if (OP == 9) asm volatile("v_alignbit_b32 %0, %0, %1, 31" : "+v"(u[i]) : "v"(um));
            if (OP == 10) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(u[i]) : "v"(um), "v"(um));
            if (OP == 11) asm volatile("v_rsq_f32 %0, %0" : "+v"(a[i]));
            if (OP == 12) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c));
            if (OP == 13) asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(u[i]) : "v"(um), "v"(um));
            if (OP == 14) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(pm));
            if (OP == 15) asm volatile("v_mov_b32 %0, %1" : "=v"(u[i]) : "v"(um));
            if (OP == 16) asm volatile("v_cvt_f32_ubyte1 %0, %1" : "=v"(a[i]) : "v"(u[i]));
            if (OP == 17) asm volatile("v_cvt_f32_ubyte3 %0, %1" : "=v"(a[i]) : "v"(u[i]));
            if (OP == 18) asm volatile("v_xor_b32_sdwa %0, %0, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD" : "+v"(u[i]));
            if (OP == 19) asm volatile("v_pk_fma_f32 %0, %0, %1, %2 op_sel:[0,0,1] op_sel_hi:[1,0,1]" : "+v"(p[i]) : "v"(pm), "v"(pc));
            if (OP == 20) asm volatile("v_pk_add_f32 %0, %0, %1 op_sel:[0,1] op_sel_hi:[1,1]" : "+v"(p[i]) : "v"(pc));
            if (OP == 21) asm volatile("v_fma_f32 %0, %0, |%0|, -%1" : "+v"(a[i]) : "v"(c));
            if (OP == 22) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c));
            if (OP == 23) asm volatile("v_bcnt_u32_b32 %0, %0, 0" : "+v"(u[i]));
        }
    }
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1), "=s"(r1) :: "memory");
    float r = 0; uint32_t ru = 0;
#pragma unroll
    for (int i = 0; i < CHAINS; ++i) { r += a[i] + p[i].x + p[i].y; ru ^= u[i]; }
    if ((threadIdx.x & 63) == 0) {
        Stamp st; st.t0 = t0; st.t1 = t1; st.r0 = r0; st.r1 = r1;
        st.hw_id = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));  // HW_REG_HW_ID
        st.xcc = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (3 << 11)) | ((r + (float)ru == 1.2345f) ? 0x100u : 0u);  // HW_REG_XCC_ID
        out[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = st;
    }
}

static const char* names[] = {"v_fma_f32", "v_pk_fma_f32", "v_mul_lo_u32", "v_xor_b32", "v_cvt_f32_u32", "v_max_f32",
                              "v_add_f32", "v_mul_f32", "v_lshrrev_b32", "v_alignbit_b32", "v_and_or_b32", "v_rsq_f32",
                              "v_fmac_f32", "v_mad_u32_u24", "v_pk_mul_f32", "v_mov_b32", "v_cvt_f32_ubyte1", "v_cvt_f32_ubyte3",
                              "v_xor_b32_sdwa", "v_pk_fma_f32 op_sel", "v_pk_add_f32 op_sel", "v_fma_f32 |x| -y", "v_max3_f32", "v_bcnt_u32_b32"};

template <int OP>
static void run(int cus, int w) {
    const int blocks = cus * w, waves = blocks * 4;
    Stamp* d; hipMalloc(&d, sizeof(Stamp) * waves);
    std::vector<Stamp> h(waves);
    probe<OP><<<blocks, 256>>>(d, 1.0f, ITERS);   // warm-up (clock ramp, code fetch)
    probe<OP><<<blocks, 256>>>(d, 1.0f, ITERS);
    hipDeviceSynchronize();
    hipMemcpy(h.data(), d, sizeof(Stamp) * waves, hipMemcpyDeviceToHost);
    const double insts = (double)ITERS * CHAINS;
    std::vector<double> per_wave, clk, per_simd;
    // group the waves by the SIMD they ran on: (xcc, se, sh, cu, simd); one CU's waves share one s_memtime counter
    std::map<uint32_t, std::vector<const Stamp*>> simd;
    for (auto& s : h) {
        per_wave.push_back((double)(s.t1 - s.t0) / insts);
        if (s.r1 > s.r0) clk.push_back((double)(s.t1 - s.t0) / (double)(s.r1 - s.r0) * 0.1);  // GHz (100 MHz reference)
        const uint32_t key = ((s.xcc & 0xFu) << 16) | (s.hw_id & 0xFF30u);   // se_id, sh_id, cu_id, simd_id
        simd[key].push_back(&s);
    }
    double waves_seen = 0;
    for (auto& kv : simd) {
        unsigned long long lo = ~0ull, hi = 0;
        for (auto* s : kv.second) { lo = std::min(lo, s->t0); hi = std::max(hi, s->t1); }
        per_simd.push_back((double)(hi - lo) / (insts * kv.second.size()));
        waves_seen += kv.second.size();
    }
    std::sort(per_wave.begin(), per_wave.end());
    std::sort(clk.begin(), clk.end());
    std::sort(per_simd.begin(), per_simd.end());
    printf("%-16s waves/SIMD=%d (%zu SIMDs seen, %.2f waves each)  per wave: %.2f cyc/inst  per SIMD (first start..last end): median %.2f min %.2f max %.2f cyc/inst  clock %.2f GHz\n",
           names[OP], w, simd.size(), waves_seen / simd.size(), per_wave[per_wave.size() / 2], per_simd[per_simd.size() / 2],
           per_simd.front(), per_simd.back(), clk.empty() ? 0.0 : clk[clk.size() / 2]);
    hipFree(d);
}
template <int OP> static void sweep(int cus) { run<OP>(cus, 1); run<OP>(cus, 2); run<OP>(cus, 4); run<OP>(cus, 8); }

int main() {
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    printf("device %s  CUs=%d  nominal clock=%d kHz\n", p.name, p.multiProcessorCount, p.clockRate);
    const int cus = p.multiProcessorCount;
    sweep<0>(cus); sweep<1>(cus); sweep<2>(cus); sweep<3>(cus); sweep<4>(cus); sweep<5>(cus); sweep<6>(cus); sweep<7>(cus);
    sweep<8>(cus); sweep<9>(cus); sweep<10>(cus); sweep<11>(cus); sweep<12>(cus); sweep<13>(cus); sweep<14>(cus); sweep<15>(cus);
    sweep<16>(cus); sweep<17>(cus); sweep<18>(cus); sweep<19>(cus); sweep<20>(cus); sweep<21>(cus); sweep<22>(cus); sweep<23>(cus);
    return 0;
}
