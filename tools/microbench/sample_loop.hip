// Cycle probe of the soft-shadow sample loop (k_shade_fast, K linear rows, opaque scene) on gfx950: the loop body as
// the kernel has it -- one hash per two samples, byte jitter, packed FP32 -- against the one-hash-per-sample body it
// replaced, timed in-kernel (s_memtime) at the kernel's occupancy of 4 waves per SIMD.  Prints cycles per sample per
// SIMD.  Build: hipcc -O3 --offload-arch=gfx950 -ffp-contract=fast sample_loop.hip -o sample_loop
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <map>
#include <vector>

typedef float f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f2 F2(float x, float y) { f2 r; r.x = x; r.y = y; return r; }
__device__ __forceinline__ f2 pk_fma(f2 a, f2 b, f2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ f2 pk_fma_bLcL(f2 a, f2 b, f2 c) { f2 d; asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(d) : "v"(a), "v"(b), "v"(c)); return d; }
__device__ __forceinline__ f2 pk_fma_bLcH(f2 a, f2 b, f2 c) { f2 d; asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,1] op_sel_hi:[1,0,1]" : "=v"(d) : "v"(a), "v"(b), "v"(c)); return d; }
__device__ __forceinline__ f2 pk_fma_bL(f2 a, f2 b, f2 c) { f2 d; asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[1,0,1]" : "=v"(d) : "v"(a), "v"(b), "v"(c)); return d; }
__device__ __forceinline__ f2 pk_fma_bH(f2 a, f2 b, f2 c) { f2 d; asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "=v"(d) : "v"(a), "v"(b), "v"(c)); return d; }
__device__ __forceinline__ f2 pk_fma_cH(f2 a, f2 b, f2 c) { f2 d; asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,1] op_sel_hi:[1,1,1]" : "=v"(d) : "v"(a), "v"(b), "v"(c)); return d; }
__device__ __forceinline__ f2 pk_add_bH(f2 a, f2 b) { f2 d; asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,1]" : "=v"(d) : "v"(a), "v"(b)); return d; }
template <int B> __device__ __forceinline__ float ubyte_f32(uint32_t m) { return (float)((m >> (8 * B)) & 0xFFu); }
__device__ __forceinline__ uint32_t mix32(uint32_t x) { x ^= x >> 16; x *= 0x7FEB352Du; x ^= x >> 15; x *= 0x846CA68Bu; x ^= x >> 16; return x; }

struct Stamp { unsigned long long t0, t1; uint32_t hw_id, xcc; };

template <int K, int MODE>
__global__ __launch_bounds__(256, 4) void probe(Stamp* out, const float* coef, uint32_t* sink, int tiles, int N) {
    const int lane = threadIdx.x;
    float Ad[K], Bd[K], Cd[K];
#pragma unroll
    for (int e = 0; e < K; ++e) { Ad[e] = coef[lane * 32 + e]; Bd[e] = coef[lane * 32 + 8 + e]; Cd[e] = coef[lane * 32 + 16 + e]; }
    const float ku = coef[lane * 32 + 24] * (MODE == 0 ? 1.f / 256.f : 1.f), koff = coef[lane * 32 + 25], lc2 = coef[lane * 32 + 26];
    const uint32_t key = (uint32_t)(blockIdx.x * 256 + lane) * 2654435761u;
    uint32_t open = 0;
    unsigned long long t0, t1;
    __syncthreads();
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");
    for (int t = 0; t < tiles; ++t) {
        const int i0 = (t & 1) * 8, j0 = ((t >> 1) & 1) * 8, i1 = i0 + 8, j1 = j0 + 8;
        if (MODE == 0) {   // 16-bit jitter, one hash per sample, scalar FP32
            const float cell = 65536.0f * ku;
            for (int i = i0; i < i1; ++i) {
                uint32_t ctr = key + (uint32_t)(i * N + j0) * 0x9E3779B9u;
                uint32_t mask = 0;
                const float urow = fmaf((float)i, cell, koff);
                for (int j = j0; j < j1; ++j) {
                    const uint32_t mm = mix32(ctr);
                    ctr += 0x9E3779B9u;
                    const float ur = fmaf((float)(mm >> 16), ku, urow);
                    const float vr = fmaf((float)((mm & 0xFFFFu) | ((uint32_t)j << 16)), ku, koff);
                    const float len2 = fmaf(ur, ur, fmaf(vr, vr, lc2));
                    float vmax = fmaf(vr, Cd[0], fmaf(ur, Bd[0], Ad[0]));
#pragma unroll
                    for (int e = 1; e < K; ++e) vmax = __builtin_fmaxf(vmax, fmaf(vr, Cd[e], fmaf(ur, Bd[e], Ad[e])));
                    const float res = fmaf(vmax, fabsf(vmax), -len2);
                    mask = __builtin_amdgcn_alignbit(mask, __float_as_uint(res), 31);
                }
                open += (uint32_t)__popc(mask);
            }
        } else {           // byte jitter, one hash per two samples, packed FP32
            const float cell = 256.0f * ku;
            f2 BC[K], AA[(K + 1) / 2];
#pragma unroll
            for (int e = 0; e < K; ++e) BC[e] = F2(Bd[e], Cd[e]);
#pragma unroll
            for (int e = 0; e < (K + 1) / 2; ++e) AA[e] = F2(Ad[2 * e], 2 * e + 1 < K ? Ad[2 * e + 1] : -INFINITY);
            const f2 KU = F2(ku, 2.f * cell);
            const f2 vcol0 = F2(fmaf((float)j0, cell, koff), fmaf((float)(j0 + 1), cell, koff));
            const uint32_t Np = (uint32_t)(N + 1) >> 1;
            for (int i = i0; i < i1; ++i) {
                uint32_t ctr = key + ((uint32_t)i * Np + ((uint32_t)j0 >> 1)) * 0x9E3779B9u;
                uint32_t mask = 0;
                const f2 UL = F2(fmaf((float)i, cell, koff), lc2);
                f2 vcol = vcol0;
                for (int j = j0; j < j1; j += 2) {
                    const uint32_t mm = mix32(ctr);
                    ctr += 0x9E3779B9u;
                    const f2 ub = F2(ubyte_f32<3>(mm), ubyte_f32<1>(mm));
                    const f2 vb = F2(ubyte_f32<2>(mm), ubyte_f32<0>(mm));
                    const f2 ur = pk_fma_bLcL(ub, KU, UL);
                    const f2 vr = pk_fma_bL(vb, KU, vcol);
                    vcol = pk_add_bH(vcol, KU);
                    const f2 len2 = pk_fma(ur, ur, pk_fma_cH(vr, vr, UL));
                    f2 vmax = pk_fma_bH(vr, BC[0], pk_fma_bLcL(ur, BC[0], AA[0]));
#pragma unroll
                    for (int e = 1; e < K; ++e) {
                        const f2 tt = (e & 1) ? pk_fma_bLcH(ur, BC[e], AA[e >> 1]) : pk_fma_bLcL(ur, BC[e], AA[e >> 1]);
                        const f2 lin = pk_fma_bH(vr, BC[e], tt);
                        vmax.x = __builtin_fmaxf(vmax.x, lin.x); vmax.y = __builtin_fmaxf(vmax.y, lin.y);
                    }
                    const float r0 = fmaf(vmax.x, fabsf(vmax.x), -len2.x), r1 = fmaf(vmax.y, fabsf(vmax.y), -len2.y);
                    mask = __builtin_amdgcn_alignbit(mask, __float_as_uint(r0), 31);
                    mask = __builtin_amdgcn_alignbit(mask, __float_as_uint(r1), 31);
                }
                open += (uint32_t)__popc(mask);
            }
        }
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory");
    sink[blockIdx.x * 256 + lane] = open;
    if ((lane & 63) == 0) {
        Stamp st; st.t0 = t0; st.t1 = t1;
        st.hw_id = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));
        st.xcc = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (3 << 11));
        out[(blockIdx.x * 256 + lane) >> 6] = st;
    }
}

template <int K, int MODE>
static void run(int cus, const float* d_coef) {
    const int w = 4, blocks = cus * w, waves = blocks * 4, tiles = 256;
    Stamp* d; hipMalloc(&d, sizeof(Stamp) * waves);
    uint32_t* sink; hipMalloc(&sink, 4 * blocks * 256);
    std::vector<Stamp> h(waves);
    for (int r = 0; r < 2; ++r) probe<K, MODE><<<blocks, 256>>>(d, d_coef, sink, tiles, 16);
    hipDeviceSynchronize();
    hipMemcpy(h.data(), d, sizeof(Stamp) * waves, hipMemcpyDeviceToHost);
    std::map<uint32_t, std::vector<const Stamp*>> simd;
    for (auto& s : h) simd[((s.xcc & 0xFu) << 16) | (s.hw_id & 0xFF30u)].push_back(&s);
    std::vector<double> per;
    const double samples = (double)tiles * 64;
    for (auto& kv : simd) {
        unsigned long long lo = ~0ull, hi = 0;
        for (auto* s : kv.second) { lo = std::min(lo, s->t0); hi = std::max(hi, s->t1); }
        per.push_back((double)(hi - lo) / (samples * kv.second.size()));
    }
    std::sort(per.begin(), per.end());
    printf("K=%d %-34s %.1f cycles per wave-sample per SIMD (median of %zu SIMDs, %d waves each)\n", K,
           MODE == 0 ? "one hash per sample, scalar FP32:" : "one hash per pair, packed FP32:", per[per.size() / 2], simd.size(), w);
    hipFree(d); hipFree(sink);
}

int main() {
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    std::vector<float> coef(256 * 32);
    for (size_t i = 0; i < coef.size(); ++i) coef[i] = 0.01f * (float)((i * 7919u) % 997u) - 3.f;
    float* d_coef; hipMalloc(&d_coef, coef.size() * 4);
    hipMemcpy(d_coef, coef.data(), coef.size() * 4, hipMemcpyHostToDevice);
    const int cus = p.multiProcessorCount;
    run<1, 0>(cus, d_coef); run<1, 1>(cus, d_coef);
    run<2, 0>(cus, d_coef); run<2, 1>(cus, d_coef);
    run<4, 0>(cus, d_coef); run<4, 1>(cus, d_coef);
    run<8, 0>(cus, d_coef); run<8, 1>(cus, d_coef);
    return 0;
}
