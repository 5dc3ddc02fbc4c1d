// How fast does this GPU take returning atomic adds, from one lane per workgroup, (a) all on ONE word, (b) on one word per
// workgroup (128 bytes apart), (c) on 16 words shared round-robin?  Each workgroup issues its atomics one after the other
// (the next needs the last one's result, as a work counter does).  Prints nanoseconds per atomic, aggregate, and the
// latency one workgroup sees.
// Build: hipcc -O3 --offload-arch=gfx950 atomic_rate.hip -o atomic_rate
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

__global__ __launch_bounds__(64) void k(uint32_t* words, int stride_words, int n_words, int iters, uint32_t* sink) {
    if (threadIdx.x != 0) return;
    uint32_t* w = words + (size_t)(blockIdx.x % n_words) * stride_words;
    uint32_t v = 0;
    for (int i = 0; i < iters; ++i) v += atomicAdd(w + (v & 0u), 1u);   // (address depends on the last result: no overlap)
    sink[blockIdx.x] = v;
}

int main() {
    const int blocks = 1024, iters = 200;
    uint32_t *words, *sink;
    hipMalloc(&words, 4096 * 128); hipMalloc(&sink, blocks * 4);
    hipMemset(words, 0, 4096 * 128);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int cases[][2] = {{1, 1}, {16, 32}, {64, 32}, {1024, 32}};
    for (auto& c : cases) {
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(k, dim3(blocks), dim3(64), 0, 0, words, c[1], c[0], iters, sink);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (rep == 2) printf("%4d word(s): %d workgroups x %d dependent atomics in %.3f ms = %.2f ns per atomic aggregate, %.0f ns per atomic as one workgroup sees it\n",
                                 c[0], blocks, iters, ms, ms * 1e6 / ((double)blocks * iters), ms * 1e6 / iters);
        }
    }
    return 0;
}
