// rt_device.h -- structures shared by the host orchestration (rt_api.cpp) and the
// gfx950 kernels (rt_trace.hip, rt_shade.hip).  Not part of the public ABI.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define RT_EPS 1e-6   // /root/reference/utils.py:6
#define RT_EPSF 1e-6f
#define RT_TRACE_BLOCK 512  // threads per workgroup of the ray-tree kernel
// its run-to-completion variant: threads per workgroup, workgroups per CU and waves per SIMD its registers are budgeted
// for -- scenes whose surface list is staged in LDS / scenes walked through the grid (a wave takes its tiles alone there:
// single-wave workgroups wait for nobody, and three waves per SIMD leave the walk its registers)
#define RT_TAIL_BLOCK 512
#define RT_TAIL_PER_CU 2
#define RT_TAIL_WAVES 4
#define RT_TAIL_BLOCK_GRID 64
#define RT_TAIL_PER_CU_GRID 12
#define RT_TAIL_WAVES_GRID 3
#define RT_TAIL_RESERVE 1024u   // shade-queue slots one of its workgroups reserves at a time, at most
#define RT_SURF_TILE 512    // surfaces staged in LDS at a time (512 x 64 B = 32 KiB)

// ---- float64 surface record used by the ray-tree kernel -------------------------
// One 64-byte record per surface, file order.  All lanes of a wave test the same
// surface at the same time, so LDS reads of a record are broadcasts.
//   sphere: d[0..2] centre, d[3] radius, d[4] r*r, d[5] pow(r,2)
//   plane : d[0..2] unit normal, d[3] offset
//   box   : d[0..2] min corner, d[3..5] max corner
struct SurfD {
    double d[7];
    int32_t type;  // RT_SURF_SPHERE / PLANE / BOX (PLANE_RAW is resolved on upload)
    int32_t mat;
};

// ---- float32 surface record used by the shading kernel --------------------------
//   sphere: a = (cx, cy, cz, r*r)
//   plane : a = (nx, ny, nz, offset)
//   box   : a = (minx, miny, minz, -), b = (maxx, maxy, maxz, -)
// Surfaces with identical geometry ("twins": a line of the scene file repeated) form a group; a shadow ray that
// starts on one of them starts on all, so the own-surface rules apply to the whole group.
//   pad = index of the group's first surface, b.w = product of the group's transparencies
struct __attribute__((aligned(16))) SurfF {
    float4 a;
    float4 b;       // box: max corner; .w (all types): see above
    float4 c;       // sphere: (radius, -, -, -); box: bounding sphere (centre, radius)
    int32_t type;
    float transp;   // transparency of the surface's material (shadow attenuation)
    int32_t mat;
    int32_t pad;    // first surface of its twin group (itself when it has no twin)
};

// ---- uniform grid over the bounded surfaces (spheres, boxes), built on upload for large scenes ----
// Cell (ix, iy, iz) lists, in ascending order, the surfaces whose slightly inflated bounding box
// touches it: items[cell_start[c] .. cell_start[c + 1]) with c = (iz * n[1] + iy) * n[0] + ix.
// Unbounded surfaces (planes) are in global_items and tested for every ray.  cell_start == nullptr:
// no grid, every ray is tested against every surface.
struct SceneGrid {
    double lo[3], cell[3], inv_cell[3];
    int32_t n[3];
    int32_t n_global;
    const uint32_t* cell_start;
    const uint32_t* items;
    const uint32_t* global_items;
    // one bounding sphere per entry of `items` (centre xyz, radius; FP32, the radius grown by the rounding of the
    // centre): a ray that stays clear of it cannot meet the surface, which the walks test before they load the surface
    const float4* item_bounds;
    // one 64-byte record per cell: {entries in the cell, index of entry 0, 1, 2} and the bounds of those three entries --
    // what a walk needs of most cells in ONE round trip (cell_start -> item_bounds -> items are three dependent ones);
    // entries from the fourth on are read from the lists above
    const uint4* cell_rec;
};

struct MatF {
    float kd[3], ks[3], kr[3];
    float shin, transp;
    int32_t has_refl;
};

struct LightF {
    float pos[3], col[3];
    float spec, shadow, radius;
    int32_t point_path;  // 1: single-ray path (N == 1 or shadow_intensity == 0, ray_tracer.py:36)
};

struct CameraD {
    double pos[3], fwd[3], right[3], up[3];
    double sdist, swidth;
};

// ---- queues (structure of arrays in HBM; one slot per ray / per shaded hit) -----
// Every array holds one 16-byte element per slot, so a wave reads / writes 1 KiB per instruction.
struct RayQueue {
    double2 *oa, *ob, *oc;  // (ox,oy) (oz,dx) (dy,dz): float64 origin / direction (the bit-faithful chain)
    uint4 *tp;              // throughput r,g,b (float bits: product of kr / transparency up the tree), LOCAL pixel index
    uint32_t *node;         // ray-tree node id (1 = camera ray, 2k reflect, 2k+1 transmit)
    uint32_t* count;        // device counter
    uint32_t capacity;
};

// flags of a shaded hit: how the reference's float64 arithmetic resolves shadow rays that
// start on the hit primitive and point into it (decided in float64 by the ray-tree kernel)
#define HF_CBATCH_POS 1u   // sphere: sum(oc^2) - r**2 > 0   (sphere.py:93)  -> inward samples NOT blocked
#define HF_CDOT_POS 2u     // sphere: dot(oc,oc) - r*r > 0    (sphere.py:38)  -> same, single-ray path
#define HF_BOX_OUTSIDE 4u  // box: hit point lies strictly outside its face plane -> single-ray path not blocked
#define HF_NO_OWN 8u       // degenerate hit (zero normal): no own-surface rule
// A hit on a sphere with c > 0 (either form above): a shadow ray that points INTO that sphere has its near root at
// t1 = c/|b| ~ 1e-16.  The reference keeps it (and so sees no blocker) only if its float64 quadratic still resolves
// it: sqrt(fl(b*b - 4ac)) < |b|, which depends on the last bits of b once 4ac is within a few ulps of b*b -- the
// normal case, since c is a rounding residue of one or two quanta.  Such hits carry their float64 hit point
// (ShadeQueue::p64) and the shading kernels evaluate the own-sphere quadratic of inward samples in float64.
#define HF_EXACT 16u

// Slots [0, base) belong to the camera rays: the ray of enumeration index i (8 x 8 pixel tiles, see
// rt_trace.hip) owns slot i, so a 64-slot chunk is one pixel tile; a slot whose ray left the scene or
// lies outside the band holds RT_NULL_HIT in r3.w.  Hits of deeper rays are appended from `base` on;
// *count counts only those.
#define RT_NULL_HIT 0xFFFFFFFFu
#define RT_PAIR_FULL 0x80000000u   // exact-list entry: shade the whole pair (light word without this bit = light index)
struct ShadeQueue {
    float4 *r0;   // hit point xyz, normal x
    float4 *r1;   // normal yz (stored outward normal), view xy (direction towards the ray origin)
    float4 *r2;   // view z, weight rgb = throughput * (1 - transparency)
    uint4 *r3;    // LOCAL pixel index inside the band, GLOBAL pixel index y*W+x (jitter key), node, surface | flags << 24
    double *p64;  // float64 hit point, 3 per slot; written only for HF_EXACT hits
    uint32_t* count;
    uint32_t capacity;
    uint32_t base;
};

// optional float64 record of every nearest-hit query (debug / parity tests)
struct HitDebug {
    uint32_t *gpixel, *node;
    int32_t* surf;
    double *t, *P, *n;  // P, n: 3 per record
    uint32_t* flags;
    uint32_t* count;
    uint32_t capacity;
};

// device counters, one block of u64
enum { CNT_PRIMARY = 0, CNT_SECONDARY, CNT_DEPTH_CUT, CNT_SHADED, CNT_OVERFLOW_RAY, CNT_OVERFLOW_SHADE, CNT_WORK,
       // how the shading kernel resolved each (wave, light): ratio = 1 without sampling, compact loop
       // with 1..4 linear tests, generic loop (some lane needs it / more than 4 candidates), single-ray path
       CNT_WL_NONE, CNT_WL_K1, CNT_WL_K2, CNT_WL_K3, CNT_WL_K4, CNT_WL_GENERIC, CNT_WL_POINT,
       CNT_EXACT,  // (hit, light) pairs evaluated by k_shade_exact
       CNT_SAMPLES,  // shadow-ray samples generated and tested by the shading kernels (lanes holding a hit x samples)
       CNT_N };
// The counters exist RT_CNT_BANKS times, one 128-byte line each; a workgroup adds into the bank of its index and the host
// sums the banks.  Atomics on one line complete one after the other, ~11 ns each on this chip: with one copy, the ~20 000
// additions of the ~3 000 waves of a persistent kernel -- all arriving as the launch ends -- held it open for ~0.1 ms.
#define RT_CNT_BANKS 64
static_assert(CNT_N == 16, "one bank of counters is one 128-byte line");
__device__ __forceinline__ unsigned long long* rt_counter(unsigned long long* counters, int which) {
    return counters + (blockIdx.x & (RT_CNT_BANKS - 1)) * CNT_N + which;
}

struct FrameParams {
    int32_t width, height;
    int32_t band_rows;       // rows in this band
    int32_t max_rec;
    int32_t n_shadow;
    int32_t n_surfaces, n_lights;
    uint32_t seed_key;       // mix32(seed + GOLDEN), the per-frame part of the jitter key
    float bg[3];
    const int32_t* row_y;    // image row of each band row
    int32_t all_p64;         // 1: every shaded hit carries its float64 hit point (scene with a light lying on a surface)
};

// ---- jitter stream (contract: oracle/jitter.py) ----------------------------------
__host__ __device__ inline uint32_t rt_mix32(uint32_t x) {
    x ^= x >> 16; x *= 0x7FEB352Du; x ^= x >> 15; x *= 0x846CA68Bu; x ^= x >> 16; return x;
}
__host__ __device__ inline uint32_t rt_seed_key(uint32_t seed) { return rt_mix32(seed + 0x9E3779B9u); }
__host__ __device__ inline uint32_t rt_grid_key(uint32_t seed_key, uint32_t pixel, uint32_t node, uint32_t light) {
    uint32_t k = rt_mix32(seed_key ^ pixel);
    k = rt_mix32(k ^ (node * 0x9E3779B1u));
    k = rt_mix32(k ^ ((light + 1u) * 0x85EBCA6Bu));
    return k;
}
// Sample (i, j) of an n x n grid: samples (i, 2q) and (i, 2q+1) share one 32-bit word, one byte per coordinate -- the
// even sample takes bytes 3 (u) and 2 (v), the odd one bytes 1 (u) and 0 (v); the jitter is byte / 256 of a grid cell.
// The word is  rt_pair_word(i * ceil(n/2) + q) ^ key : a fixed table of well-mixed words indexed by the pair alone,
// scrambled per (pixel, node, light) by XOR with the grid's key.  Every byte of the result is uniform whatever the key,
// grids of different pixels differ by the XOR of their keys, and all lanes of a wavefront walk the table at the same
// index: the shading kernel reads the table word through the scalar unit and spends ONE vector instruction per pair.
#define RT_JITTER_INV 0.00390625f   // 2^-8
__host__ __device__ inline uint32_t rt_pair_stride(uint32_t n) { return (n + 1u) >> 1; }
__host__ __device__ inline uint32_t rt_pair_word(uint32_t pair) { return rt_mix32(pair * 0x9E3779B9u + 0x5BF03635u); }
// words of the table a scene with an n x n grid reads (rows i = 0..n, so that the row after the last may be prefetched,
// plus one 4-word load past the end)
__host__ __device__ inline uint32_t rt_pair_table_words(uint32_t n) { return (n + 1u) * rt_pair_stride(n) + 8u; }
__host__ __device__ inline void rt_sample_bits(uint32_t key, uint32_t n, uint32_t i, uint32_t j, uint32_t& u, uint32_t& v) {
    const uint32_t m = rt_pair_word(i * rt_pair_stride(n) + (j >> 1)) ^ key;
    if (j & 1u) { u = (m >> 8) & 0xFFu; v = m & 0xFFu; }
    else { u = m >> 24; v = (m >> 16) & 0xFFu; }
}

// ---- launch wrappers implemented by the .hip files -------------------------------
struct TraceArgs {
    FrameParams fp;
    CameraD cam;
    const SurfD* surf;
    const MatF* mat;
    SceneGrid grid;
    RayQueue in, out;
    ShadeQueue shade;
    HitDebug dbg;          // capacity 0 = off
    double* accum;         // planar [3][npix]
    size_t npix;           // pixels of the band
    unsigned long long* counters;
    int depth;
    RayQueue stack;        // tail kernel only: parked children, [level][thread]; tp.w = depth of the parked ray
    uint32_t stack_levels; // slots per thread (0: the scene has no material with both children)
    uint32_t tail_reserve; // tail kernel only: shade-queue slots a workgroup reserves at a time (a multiple of 64)
    uint32_t* work;        // tail kernel only: tiles handed out so far beyond each wave's (workgroup's) first one; zero at launch
};
void rt_launch_trace(const TraceArgs& a, int blocks, hipStream_t s);
// follows every ray of a.in (or every camera ray when a.depth == 0) to the recursion limit; needs a.grid, or n_surfaces <= RT_SURF_TILE
void rt_launch_trace_tail(const TraceArgs& a, int blocks, hipStream_t s);

enum { SHADE_AUTO = 0, SHADE_GENERAL = 1 };
#define RT_WORK_QUEUES 16u
#define RT_WORK_QUEUE_STRIDE 32u   // words: one 128-byte line per counter
struct ShadeArgs {
    FrameParams fp;
    int32_t variant;     // SHADE_AUTO picks the register-resident kernel for small scenes
    int32_t has_box;     // scene contains at least one box
    int32_t all_opaque;  // every surface's material has transparency == 0
    int32_t no_cull;     // 1: always run the generic per-sample loop (no per-light classification)
    uint32_t* work;      // work counters of the fast kernel: RT_WORK_QUEUES of them, RT_WORK_QUEUE_STRIDE words apart (zeroed per band)
    uint2* deferred;     // (first hit of a 64-hit chunk, light) pairs left to the generic loop
    uint32_t* deferred_count;
    uint2* exact;        // (shade slot, light) pairs whose own-sphere samples need float64 (k_shade_exact), RT_NULL_HIT = void
    uint32_t* exact_count;
    uint32_t exact_capacity;
    uint32_t* work3;     // shared work counter of k_shade_exact
    uint32_t* full_count; // RT_PAIR_FULL entries: they fill the same list from its END downwards (one lane each in k_shade_exact)
    uint32_t* work4;     // ... and their work counter
    float exact_min;     // a back-lit pair goes to k_shade_exact only if its specular term can exceed this (full scale = 1)
    // Lights whose centre lies on (within 1e-3 of) a surface (null: the scene has none): the surfaces of light l are
    // special[L + 1 + k] for k in [special[l], special[l + 1]).  A shadow ray towards such a light meets that surface at
    // t = dist up to rounding, and the reference decides t < dist by the last bit of its float64 arithmetic: every pair
    // with such a light is shaded by k_shade_exact, with those surfaces tested in float64 (list entries with
    // RT_PAIR_FULL set).
    const uint32_t* special;
    const uint32_t* jitter_words;   // rt_pair_word(0 .. rt_pair_table_words(N) - 1), read at wave-uniform indices
    const SurfF* surf;
    const MatF* mat;
    const LightF* light;
    const SurfD* surf64; // float64 surfaces and lights (position xyz, radius: 4 per light) for HF_EXACT hits
    const double* light64;
    SceneGrid grid;      // used by the any-size kernel
    ShadeQueue q;
    double* accum;       // planar [3][npix]
    size_t npix;
    unsigned long long* counters;
};
// after_main (may be null) is recorded behind the first, dominant launch
void rt_launch_shade(const ShadeArgs& a, int n_cu, hipStream_t s, hipEvent_t after_main);

void rt_launch_resolve(double* accum, int64_t n_pixels, float* out_f32, uint8_t* out_u8, hipStream_t s);

// rows of uint8 pixels from device memory to (pinned, device-visible) host memory; src_row (device, may be null)
// names the source row of every destination row (rt_deliver.hip)
void rt_launch_deliver(void* dst, const void* src, const int32_t* src_row, uint32_t rows, size_t row_bytes, int workgroups, hipStream_t s);
