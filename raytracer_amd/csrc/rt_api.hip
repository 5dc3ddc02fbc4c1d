// rt_api.hip -- host side of librt_amd.so: the C ABI of include/rt_amd.h.
//
// Owns the per-GPU context, derives the scene constants in float64 with the
// reference's rounding (camera basis camera.py:14-16, unit plane normals
// infinite_plane.py:8, box corners cube.py:32-34, r*r and r**2 sphere.py:38/:93),
// and drives the wavefront pipeline for a frame:
//
//   for each row band:   depth 0: camera rays + nearest hit  -> shade queue, ray queue 1
//                        depth d: ray queue d -> nearest hit -> shade queue, ray queue d+1
//                        shade : every shaded hit of the band, all depths, ONE launch
//                        resolve: accumulators -> clip*255 -> float32 / uint8 rows
//
// The launch sequence is fixed (max_recursions trace launches per band whatever the
// queues hold; kernels read their item counts from device memory), so there is no
// host round trip inside a band; the only synchronisation is at the end of a band,
// where the overflow counters are read.  A band whose ray tree outgrows the queues
// is retried with fewer rows.
#include "rt_device.h"
#include "../../include/rt_amd.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <string>
#include <array>
#include <map>
#include <vector>

#pragma clang fp contract(off)

namespace {

struct v3 { double x, y, z; };
inline v3 sub(v3 a, v3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline v3 divs(v3 a, double s) { return {a.x / s, a.y / s, a.z / s}; }
inline double dotb(v3 a, v3 b) { return std::fma(a.z, b.z, std::fma(a.y, b.y, a.x * b.x)); }
inline v3 cross(v3 a, v3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
inline v3 normalize(v3 v) { double n = std::sqrt(dotb(v, v)); return n == 0 ? v : divs(v, n); }

constexpr int kRayFactor = 2;    // ray-queue slots per band pixel
constexpr int kShadeFactor = 4;  // shade-queue slots per band pixel
constexpr size_t kMinArenaPixels = 65536;
constexpr int kAsyncFirst = 32;        // bookkeeping entries (events + pinned counters) made at the first rt_render_enqueue
constexpr int kAsyncMax = 1024;        // frames rt_render_enqueue keeps in flight before it waits for the oldest one
constexpr int kGridMinSurfaces = 64;   // smaller scenes are tested surface by surface (staged in LDS)
constexpr size_t kCounterWords = (size_t)CNT_N * RT_CNT_BANKS;   // statistics: RT_CNT_BANKS banks of CNT_N words
constexpr int kTailWork = 64 + (int)(RT_WORK_QUEUES * RT_WORK_QUEUE_STRIDE);   // tile counter of k_trace_tail, on a line of its own
// list counters the shading kernels append through with returning atomics -- a (wave, light) pair at a time, tens of
// thousands per frame on scenes with touching or transparent geometry -- each on a line of its own: atomics on one
// line complete one after the other, whichever word they address
constexpr int kLineDeferred = kTailWork + 1 * (int)RT_WORK_QUEUE_STRIDE, kLineExact = kTailWork + 2 * (int)RT_WORK_QUEUE_STRIDE;
constexpr int kLineFull = kTailWork + 3 * (int)RT_WORK_QUEUE_STRIDE, kLineWork3 = kTailWork + 4 * (int)RT_WORK_QUEUE_STRIDE;
constexpr int kLineWork4 = kTailWork + 5 * (int)RT_WORK_QUEUE_STRIDE;
constexpr int kQCounts = kTailWork + 6 * (int)RT_WORK_QUEUE_STRIDE;  // [0..7] fixed roles, [8 + d] item count of the ray queue read at depth d, [64..] work counters of k_shade_fast
constexpr size_t kMaxArenaPixels = size_t(1) << 26;
constexpr size_t kHostRows = 16384;    // rows of the pinned row-list staging buffer made at rt_create (grows on demand)

}  // namespace

// Frame memory of rt_render_device_noalloc: one array per GPU that loads this library, held until the process ends.
// In host code the name is only a stand-in for the symbol: rt_create resolves this device's copy once
// (hipGetSymbolAddress, kept in g_scratch_ptr) and everything that touches the memory uses that pointer.
static __device__ __align__(256) unsigned char scratch[RT_NOALLOC_SCRATCH_BYTES];
static void* g_scratch_ptr[64] = {};   // per device ordinal

// Device memory of one row band: the ray / shade queues, the accumulators, the work lists.  Normally carved out of one
// hipMalloc'd arena that grows on demand; rt_render_device_noalloc points the same fields into the static scratch array.
struct FrameMem {
    size_t band_capacity = 0;        // pixels
    RayQueue rq[2]{};
    ShadeQueue sq{};
    double* d_accum = nullptr;
    size_t accum_zero_prefix = 0;    // accumulator values [0, prefix) are known to be zero (k_resolve clears what it reads)
    uint2* d_deferred = nullptr;
    uint2* d_exact = nullptr;        // (slot, light) list of k_shade_exact
    uint32_t exact_capacity = 0;
    int arena_lights = 0;
    bool arena_has_rays = false;     // the arena holds full-size ray queues
    int32_t* d_row_y = nullptr;
    size_t row_y_cap = 0;
    void* arena = nullptr;           // one allocation behind all queue arrays
    RayQueue stack{};                // run-to-completion ray tree: parked children, [level][thread]
    void* stack_mem = nullptr;
    size_t stack_slots = 0;
};

struct rt_handle {
    int device = 0;
    hipStream_t stream = nullptr;
    std::string err;
    int n_cu = 256;

    // scene
    bool has_scene = false;
    int S = 0, M = 0, L = 0, N = 1, R = 0;
    float bg[3] = {0, 0, 0};
    CameraD cam{};
    SurfD* d_surf = nullptr;
    SurfF* d_surff = nullptr;
    MatF* d_mat = nullptr;
    LightF* d_light = nullptr;
    double* d_light64 = nullptr;         // float64 position xyz + radius per light (exact own-sphere evaluation)
    uint32_t* d_special = nullptr;  // per light: surfaces its centre lies on (ShadeArgs::special); null if none
    uint32_t* d_jitter = nullptr;   // table of jitter pair words for the scene's N x N grid (ShadeArgs::jitter_words)
    int has_box = 0, all_opaque = 1;
    double shadow_rays_per_hit = 0;  // sum over lights of 1 or N*N
    double flop_per_hit = 0;         // algorithmic FP32 flop per shaded hit

    FrameMem fm{};                   // queues and accumulators of a band (rt_render_device_noalloc swaps in a view of the static scratch)
    bool no_alloc = false;           // set while rt_render_device_noalloc runs: nothing may be allocated or grown
    size_t max_band_pixels = 3840u * 2160u;
    HitDebug dbg{};
    unsigned long long* d_counters = nullptr;  // RT_CNT_BANKS banks of CNT_N statistics words, directly followed by d_qcounts (one memset per band)
    uint32_t* d_qcounts = nullptr;   // [0],[1] ray queues, [2] shade queue, [3] debug, [8 + d] ray queue of depth d, [64..] one line per busy counter (kTailWork, kLine*)
    unsigned long long* host_counters = nullptr;  // pinned: the band's counters land here (no staging by the runtime)
    int32_t* host_rows = nullptr;    // pinned copy of the row list on the device (the upload reads it in stream order)
    size_t host_rows_cap = 0;
    // The row list on the device is read by every band's kernels and rewritten when the selection changes.  rows_ev (owned by
    // the handle) is recorded behind the last kernel of every band; a band on another stream than the previous one first
    // waits for it, so the event always stands for ALL earlier readers, and the list is rewritten only after it has fired.
    // rows_stream_id is compared, never passed to the runtime: the caller may have destroyed that stream since.
    hipEvent_t rows_ev = nullptr;
    bool rows_ev_recorded = false;
    const void* rows_stream_id = nullptr;
    void* scratch = nullptr;         // this device's copy of the static scratch array (rt_render_device_noalloc)
    // uniform grid over the bounded surfaces (large scenes)
    SceneGrid grid{};
    uint32_t* d_grid_mem = nullptr;
    float exact_min = 1e-4f;         // RT_EXACT_MIN: see ShadeArgs::exact_min
    size_t exact_list = 0;           // RT_EXACT_LIST: initial length of the float64 pair list (test hook; 0 = two per pixel)
    int grid_override = -1;          // RT_GRID: -1 automatic (more than kGridMinSurfaces surfaces), 0 never, 1 always
    double grid_density = 4.0;       // RT_GRID_DENSITY: cells per bounded surface (measurement hook)
    // run-to-completion tail of the ray tree (k_trace_tail)
    bool both_children = false;      // some material spawns a reflection and a transparency child
    int tail_override = -1;          // RT_TAIL_FROM: -1 automatic, 0 never, d >= 1 from depth d on
    std::vector<int32_t> rows_on_device;  // what d_row_y holds
    size_t exact_want = 0;               // pairs a frame asked for when the list was too short (it is then rebuilt that large)
    float* d_out_f32 = nullptr;
    uint8_t* d_out_u8 = nullptr;
    size_t out_cap = 0;              // values
    hipEvent_t ev[7]{};              // [0..3] band stages, [4] behind the main shading launch, [5], [6] begin / end of a synchronous call
    // frames enqueued by rt_render_enqueue and not yet collected
    struct Pending { hipEvent_t ev[5]{}; unsigned long long* counters = nullptr; size_t rows = 0; };
    std::vector<Pending*> pending_free;  // entries ready for use
    std::deque<Pending*> in_flight;      // oldest first
    int async_max = kAsyncMax;           // RT_ASYNC_DEPTH (measurement hook)
    int deliver_workgroups = 0;          // RT_DELIVER_WGS: overrides the workgroup count of rt_deliver_rows (measurement hook)
    rt_stats async_sum{};                // statistics of collected frames since the last rt_render_wait
    bool async_overflow = false;
};

namespace {

int fail(rt_handle* h, int code, const std::string& msg) {
    if (h) h->err = msg;
    return code;
}
#define HIP_TRY(h, expr)                                                                          \
    do {                                                                                          \
        hipError_t e_ = (expr);                                                                   \
        if (e_ != hipSuccess)                                                                     \
            return fail(h, e_ == hipErrorOutOfMemory ? RT_ERR_OOM : RT_ERR_DEVICE,                \
                        std::string(#expr) + ": " + hipGetErrorString(e_));                       \
    } while (0)

void free_frame_buffers(rt_handle* h) {
    if (h->fm.arena) (void)hipFree(h->fm.arena);
    h->fm.arena = nullptr;
    h->fm.band_capacity = 0;
    h->fm.arena_lights = 0;
    h->fm.arena_has_rays = false;
    h->fm.accum_zero_prefix = 0;
}

// Depth from which one k_trace_tail launch replaces the per-level launches (>= depths: never).
// Whenever the surface list fits one LDS tile the whole tree, camera rays included, runs in one
// launch: no ray queue traffic, no launch per level (measured 0.45 ms against 0.92 ms on the 4K
// pool frame, 0.55 against 0.89 ms on the 1080p transparency scene).  The choice never changes
// the image.
int choose_tail_depth(const rt_handle* h, int depths) {
    if ((h->S > RT_SURF_TILE && !h->grid.cell_start) || depths < 2) return depths;
    if (h->tail_override == 0) return depths;
    if (h->tail_override > 0) return std::min(h->tail_override, depths);
    return 0;
}

// camera rays are enumerated in 8 x 8 pixel tiles; each owns one shade slot (rt_trace.hip: primary_slots)
size_t primary_slots(size_t width, size_t rows) { return ((width + 7) / 8) * ((rows + 7) / 8) * 64; }

inline size_t al256(size_t b) { return (b + 255) / 256 * 256; }

// Sizes of the queue arrays for a band of `pixels` pixels (one layout for the arena and for the static scratch).
struct BandLayout { size_t rcap, scap, dcap, ecap, bytes; };
BandLayout band_layout(size_t pixels, int lights, bool need_rays, size_t ecap_min, size_t ecap_fixed) {
    BandLayout b{};
    b.rcap = need_rays ? pixels * kRayFactor : 64;
    b.scap = pixels * kShadeFactor;
    b.dcap = (b.scap / 64 + 1) * (size_t)lights;   // every (64-hit chunk, light) pair at most once
    b.ecap = std::max(ecap_fixed ? ecap_fixed : b.scap / 2, ecap_min);  // pairs needing float64 are rare; a frame that needs more says so
    b.bytes = 2 * (4 * al256(b.rcap * 16) + al256(b.rcap * 4))     // two ray queues
            + 4 * al256(b.scap * 16) + al256(b.scap * 24)          // shade queue (+ float64 hit points)
            + al256(pixels * 3 * 8)                                // accumulators
            + al256(b.dcap * sizeof(uint2)) + al256(b.ecap * sizeof(uint2));
    return b;
}

// carve the SoA queues out of `base` (b.bytes bytes); returns the first byte behind them
char* carve_band(rt_handle* h, void* base, const BandLayout& b, size_t pixels, int lights, bool need_rays) {
    char* p = (char*)base;
    auto take = [&](size_t n) { void* r = p; p += al256(n); return r; };
    for (int q = 0; q < 2; ++q) {
        RayQueue& r = h->fm.rq[q];
        r.oa = (double2*)take(b.rcap * 16); r.ob = (double2*)take(b.rcap * 16); r.oc = (double2*)take(b.rcap * 16);
        r.tp = (uint4*)take(b.rcap * 16); r.node = (uint32_t*)take(b.rcap * 4);
        r.count = h->d_qcounts + q;
        r.capacity = (uint32_t)b.rcap;
    }
    ShadeQueue& s = h->fm.sq;
    s.r0 = (float4*)take(b.scap * 16); s.r1 = (float4*)take(b.scap * 16); s.r2 = (float4*)take(b.scap * 16);
    s.r3 = (uint4*)take(b.scap * 16);
    s.p64 = (double*)take(b.scap * 24);
    s.count = h->d_qcounts + 2;
    s.capacity = (uint32_t)b.scap;
    h->fm.d_accum = (double*)take(pixels * 3 * 8);
    h->fm.d_deferred = (uint2*)take(b.dcap * sizeof(uint2));
    h->fm.d_exact = (uint2*)take(b.ecap * sizeof(uint2));
    h->fm.exact_capacity = (uint32_t)std::min<size_t>(b.ecap, 0x7FFFFFFFu);
    h->fm.band_capacity = pixels;
    h->fm.arena_lights = lights;
    h->fm.arena_has_rays = need_rays;
    h->fm.accum_zero_prefix = 0;
    return p;
}

int ensure_band_capacity(rt_handle* h, size_t pixels) {
    const int lights = h->L > 0 ? h->L : 1;
    // the ping-pong ray queues are only needed when some level has a launch of its own
    const bool need_rays = choose_tail_depth(h, h->R > 0 ? h->R : 1) > 0;
    if (pixels <= h->fm.band_capacity && lights <= h->fm.arena_lights && (!need_rays || h->fm.arena_has_rays) &&
        h->exact_want <= h->fm.exact_capacity) return RT_OK;
    if (h->no_alloc)
        return fail(h, RT_ERR_QUEUE, "rt_render_device_noalloc: the frame's ray tree needs larger queues than the static scratch holds "
                                     "(they cannot grow here): render it with rt_render_device");
    if (pixels < h->fm.band_capacity) pixels = h->fm.band_capacity;
    if (pixels < kMinArenaPixels) pixels = kMinArenaPixels;  // head-room for deep ray trees of small frames
    free_frame_buffers(h);
    const BandLayout b = band_layout(pixels, lights, need_rays, h->exact_want, h->exact_list);
    void* base = nullptr;
    hipError_t e = hipMalloc(&base, b.bytes);
    if (e != hipSuccess) return fail(h, RT_ERR_OOM, "hipMalloc of " + std::to_string(b.bytes) + " bytes for frame queues failed");
    h->fm.arena = base;
    carve_band(h, base, b, pixels, lights, need_rays);
    return RT_OK;
}

int validate_params(rt_handle* h, const rt_render_params* p) {
    if (!p || p->struct_size != sizeof(rt_render_params)) return fail(h, RT_ERR_INVALID, "rt_render_params: bad struct_size");
    if (p->width < 1 || p->height < 1) return fail(h, RT_ERR_INVALID, "width and height must be >= 1");
    if ((int64_t)p->width * p->height > 0x7FFFFFFFll) return fail(h, RT_ERR_INVALID, "frame has more than 2^31 pixels");
    if (p->row_begin < 0 || p->row_end > p->height || p->row_begin > p->row_end)
        return fail(h, RT_ERR_INVALID, "row range outside the frame");
    if (p->stripe_rows < 1 || p->stripe_count < 1 || p->stripe_index < 0 || p->stripe_index >= p->stripe_count)
        return fail(h, RT_ERR_INVALID, "bad stripe selection");
    return RT_OK;
}

void selected_rows(const rt_render_params* p, std::vector<int32_t>& rows) {
    rows.clear();
    for (int y = p->row_begin; y < p->row_end; ++y)
        if ((y / p->stripe_rows) % p->stripe_count == p->stripe_index) rows.push_back(y);
}

struct BandResult { unsigned long long c[CNT_N]; float ms_trace, ms_shade, ms_shade_main, ms_resolve; };
// the device keeps RT_CNT_BANKS copies of the counters (rt_device.h: rt_counter); their sum is the count
void sum_counter_banks(unsigned long long (&c)[CNT_N], const unsigned long long* banks) {
    for (int k = 0; k < CNT_N; ++k) c[k] = 0;
    for (int b = 0; b < RT_CNT_BANKS; ++b)
        for (int k = 0; k < CNT_N; ++k) c[k] += banks[(size_t)b * CNT_N + k];
}

void point_tail_stack(rt_handle* h, void* mem, size_t slots) {
    char* p = (char*)mem;
    h->fm.stack.oa = (double2*)p; p += slots * 16;
    h->fm.stack.ob = (double2*)p; p += slots * 16;
    h->fm.stack.oc = (double2*)p; p += slots * 16;
    h->fm.stack.tp = (uint4*)p; p += slots * 16;
    h->fm.stack.node = (uint32_t*)p;
    h->fm.stack.count = nullptr; h->fm.stack.capacity = (uint32_t)std::min<size_t>(slots, 0xFFFFFFFFu);
    h->fm.stack_slots = slots;
}

// Parked children of the run-to-completion ray tree: max_rec slots per thread of the launch.  `blocks` may come back
// smaller when nothing may be allocated (the launch then covers the band in more passes).
int ensure_tail_stack(rt_handle* h, int& blocks) {
    if (!h->both_children) return RT_OK;  // no hit can park a child: the kernel gets stack_levels = 0
    const size_t per_block = (size_t)(h->grid.cell_start ? RT_TAIL_BLOCK_GRID : RT_TAIL_BLOCK) * (size_t)std::max(h->R, 1);
    if ((size_t)blocks * per_block <= h->fm.stack_slots) return RT_OK;
    if (h->no_alloc) {
        blocks = (int)std::min<size_t>((size_t)blocks, h->fm.stack_slots / per_block);
        if (blocks < 1) return fail(h, RT_ERR_UNSUPPORTED, "rt_render_device_noalloc: no room in the static scratch for the ray tree's stack "
                                                           "(scene with reflecting and transparent materials): render it with rt_render_device");
        return RT_OK;
    }
    const size_t slots = (size_t)blocks * per_block;
    if (h->fm.stack_mem) (void)hipFree(h->fm.stack_mem);
    h->fm.stack_mem = nullptr; h->fm.stack_slots = 0;
    HIP_TRY(h, hipMalloc(&h->fm.stack_mem, slots * 68));
    point_tail_stack(h, h->fm.stack_mem, slots);
    return RT_OK;
}

// every reader of the device row list so far has finished (see rt_handle::rows_ev)
int wait_for_row_readers(rt_handle* h) {
    if (!h->rows_ev_recorded) return RT_OK;
    hipError_t e = hipEventSynchronize(h->rows_ev);
    if (e != hipSuccess) {
        h->rows_on_device.clear();
        return fail(h, RT_ERR_DEVICE, std::string("hipEventSynchronize(rows_ev): ") + hipGetErrorString(e));
    }
    return RT_OK;
}

// The device copy of the band's row list (FrameParams::row_y).  The upload reads a pinned host buffer in stream order and
// earlier bands' kernels -- on this or on another stream -- may still read the device copy, so both are rewritten only
// after every earlier band has finished (the list changes rarely: another selection of rows, or a frame split into bands).
int upload_rows(rt_handle* h, const int32_t* rows, int n_rows, hipStream_t st) {
    if (h->rows_on_device.size() == (size_t)n_rows && !std::memcmp(h->rows_on_device.data(), rows, sizeof(int32_t) * n_rows)) return RT_OK;
    if ((size_t)n_rows > h->host_rows_cap) return fail(h, RT_ERR_INVALID, "row list longer than its staging buffer");
    { int rc = wait_for_row_readers(h); if (rc != RT_OK) return rc; }
    h->rows_on_device.clear();
    std::memcpy(h->host_rows, rows, sizeof(int32_t) * n_rows);
    HIP_TRY(h, hipMemcpyAsync(h->fm.d_row_y, h->host_rows, sizeof(int32_t) * n_rows, hipMemcpyHostToDevice, st));
    h->rows_on_device.assign(rows, rows + n_rows);
    return RT_OK;
}

// room for `n` rows in the pinned staging buffer and (unless the frame memory is the static scratch) on the device
int ensure_rows(rt_handle* h, size_t n) {
    if (n > h->host_rows_cap) {
        if (h->no_alloc) return fail(h, RT_ERR_UNSUPPORTED, "rt_render_device_noalloc: more rows than the row staging buffer holds");
        { int rc = wait_for_row_readers(h); if (rc != RT_OK) return rc; }
        if (h->host_rows) (void)hipHostFree(h->host_rows);
        h->host_rows = nullptr; h->host_rows_cap = 0; h->rows_on_device.clear();
        HIP_TRY(h, hipHostMalloc((void**)&h->host_rows, sizeof(int32_t) * n));
        h->host_rows_cap = n;
    }
    if (n > h->fm.row_y_cap) {
        if (h->no_alloc) return fail(h, RT_ERR_UNSUPPORTED, "rt_render_device_noalloc: more rows than the static scratch has room for");
        { int rc = wait_for_row_readers(h); if (rc != RT_OK) return rc; }
        if (h->fm.d_row_y) (void)hipFree(h->fm.d_row_y);
        h->fm.d_row_y = nullptr; h->fm.row_y_cap = 0; h->rows_on_device.clear();
        HIP_TRY(h, hipMalloc(&h->fm.d_row_y, sizeof(int32_t) * n));
        h->fm.row_y_cap = n;
    }
    return RT_OK;
}

// enqueue one band; rows[r0..r1) ; outputs are device pointers to the band's first value (may be null)
// With `ev` / `host_counters` given (a ring entry: events of its own, pinned counters) the band is only
// enqueued; otherwise the handle's events are used and the call returns after the stream has drained.
int run_band(rt_handle* h, const rt_render_params* p, const int32_t* rows, int n_rows, float* d_f32, uint8_t* d_u8,
             hipStream_t st, bool debug, BandResult* res, hipEvent_t* ev = nullptr, unsigned long long* host_counters = nullptr) {
    const bool enqueue_only = ev != nullptr;
    if (!ev) ev = h->ev;
    const size_t pixels = (size_t)n_rows * p->width;
    const size_t slots0 = primary_slots((size_t)p->width, (size_t)n_rows);
    // thin bands pad their tiles up to 8 rows: make room for the camera rays' slots plus as many appended hits
    if (slots0 * 2 > (size_t)h->fm.sq.capacity) {
        int rc = ensure_band_capacity(h, (slots0 * 2 + kShadeFactor - 1) / kShadeFactor);
        if (rc != RT_OK) return rc;
    }
    h->fm.sq.base = (uint32_t)slots0;
    if (h->rows_ev_recorded && h->rows_stream_id != (const void*)st) HIP_TRY(h, hipStreamWaitEvent(st, h->rows_ev, 0));
    h->rows_stream_id = (const void*)st;
    { int rc = upload_rows(h, rows, n_rows, st); if (rc != RT_OK) return rc; }
    if (h->no_alloc) h->fm.accum_zero_prefix = 0;   // the scratch is shared by every handle of the device: clear what this band adds into
    HIP_TRY(h, hipMemsetAsync(h->d_counters, 0, sizeof(unsigned long long) * kCounterWords + sizeof(uint32_t) * kQCounts, st));
    if (h->fm.accum_zero_prefix < 3 * pixels) {
        HIP_TRY(h, hipMemsetAsync(h->fm.d_accum + h->fm.accum_zero_prefix, 0, sizeof(double) * (3 * pixels - h->fm.accum_zero_prefix), st));
    }
    const size_t zero_before = std::max(h->fm.accum_zero_prefix, 3 * pixels);
    h->fm.accum_zero_prefix = 0;  // dirty until k_resolve has run over this band

    FrameParams fp{};
    fp.width = p->width; fp.height = p->height; fp.band_rows = n_rows;
    fp.max_rec = h->R; fp.n_shadow = h->N; fp.n_surfaces = h->S; fp.n_lights = h->L;
    fp.seed_key = rt_seed_key(p->seed);
    fp.bg[0] = h->bg[0]; fp.bg[1] = h->bg[1]; fp.bg[2] = h->bg[2];
    fp.row_y = h->fm.d_row_y;
    fp.all_p64 = h->d_special ? 1 : 0;

    HIP_TRY(h, hipEventRecord(ev[0], st));
    const int max_blocks = h->n_cu * 2;  // 512-thread workgroups: two are resident per CU, the loop is grid-stride
    const int depths = h->R > 0 ? h->R : 1;  // R == 0: one launch that writes the background
    const int tail_from = choose_tail_depth(h, depths);
    int tail_blocks = h->n_cu * (h->grid.cell_start ? RT_TAIL_PER_CU_GRID : RT_TAIL_PER_CU);
    if (tail_from < depths) { int rc = ensure_tail_stack(h, tail_blocks); if (rc != RT_OK) return rc; }
    for (int d = 0; d < depths && d <= tail_from; ++d) {
        TraceArgs ta{};
        ta.fp = fp; ta.cam = h->cam; ta.surf = h->d_surf; ta.mat = h->d_mat; ta.grid = h->grid;
        // every depth has its own counter word (zeroed with the others, once per band): no memset per depth
        ta.in = h->fm.rq[d & 1]; ta.out = h->fm.rq[(d + 1) & 1];
        ta.in.count = h->d_qcounts + 8 + d; ta.out.count = h->d_qcounts + 8 + d + 1;
        ta.shade = h->fm.sq;
        ta.dbg = debug ? h->dbg : HitDebug{};
        ta.accum = h->fm.d_accum; ta.npix = pixels; ta.counters = h->d_counters; ta.depth = d;
        if (d == tail_from) {
            ta.stack = h->fm.stack; ta.stack_levels = h->both_children ? (uint32_t)h->R : 0u;
            {   // what the workgroups reserve without using (less than two reservations each) stays below an eighth of the room
                const size_t room = (size_t)h->fm.sq.capacity > slots0 ? (size_t)h->fm.sq.capacity - slots0 : 0;
                size_t rsv = RT_TAIL_RESERVE;
                while (rsv > 64 && rsv * 16 * (size_t)tail_blocks > room) rsv /= 2;
                ta.tail_reserve = (uint32_t)rsv;
            }
            // tiles from a shared counter when a workgroup's equal share would be fewer than 16 eights of tiles (measured:
            // 1080p frames gain up to 12 %, 4K frames lose as much to the counter's round trips)
            // (a launch that starts below the camera rays -- a test hook -- does not know its item count here)
            const bool few = d > 0 || slots0 / ((size_t)RT_TAIL_BLOCK * 16) < (size_t)tail_blocks;   // (the grid variant ignores it)
            ta.work = few ? h->d_qcounts + kTailWork : nullptr;
            rt_launch_trace_tail(ta, tail_blocks, st);
            break;
        }
        int blocks = max_blocks;
        if (d == 0) blocks = (int)std::min<size_t>((slots0 + RT_TRACE_BLOCK - 1) / RT_TRACE_BLOCK, (size_t)max_blocks);
        rt_launch_trace(ta, std::max(blocks, 1), st);
    }
    HIP_TRY(h, hipEventRecord(ev[1], st));
    if (!debug) {
        ShadeArgs sa{};
        sa.fp = fp; sa.surf = h->d_surff; sa.mat = h->d_mat; sa.light = h->d_light; sa.q = h->fm.sq; sa.grid = h->grid;
        sa.surf64 = h->d_surf; sa.light64 = h->d_light64;
        sa.accum = h->fm.d_accum; sa.npix = pixels; sa.counters = h->d_counters;
        sa.variant = (p->flags & RT_FLAG_GENERAL_SHADE) ? SHADE_GENERAL : SHADE_AUTO;
        sa.has_box = h->has_box; sa.all_opaque = h->all_opaque;
        sa.no_cull = (p->flags & RT_FLAG_NO_CULL) ? 1 : 0;
        sa.work = h->d_qcounts + 64;
        sa.deferred = h->fm.d_deferred; sa.deferred_count = h->d_qcounts + kLineDeferred;
        sa.exact_min = h->exact_min;
        sa.special = h->d_special;
        sa.jitter_words = h->d_jitter;
        sa.exact = h->fm.d_exact; sa.exact_count = h->d_qcounts + kLineExact; sa.exact_capacity = h->fm.exact_capacity; sa.work3 = h->d_qcounts + kLineWork3;
        sa.full_count = h->d_qcounts + kLineFull; sa.work4 = h->d_qcounts + kLineWork4;
        rt_launch_shade(sa, h->n_cu, st, ev[4]);
    }
    HIP_TRY(h, hipEventRecord(ev[2], st));
    if (!debug && (d_f32 || d_u8)) rt_launch_resolve(h->fm.d_accum, (int64_t)pixels, d_f32, d_u8, st);
    HIP_TRY(h, hipGetLastError());
    HIP_TRY(h, hipEventRecord(h->rows_ev, st));   // behind the last kernel that reads the row list
    h->rows_ev_recorded = true;
    HIP_TRY(h, hipMemcpyAsync(enqueue_only ? host_counters : h->host_counters, h->d_counters, sizeof(unsigned long long) * kCounterWords, hipMemcpyDeviceToHost, st));
    HIP_TRY(h, hipEventRecord(ev[3], st));  // after the counters have landed: waiting for it is enough
    if (enqueue_only) {
        if (!debug && (d_f32 || d_u8)) h->fm.accum_zero_prefix = zero_before;  // stream order: later bands run after this resolve
        return RT_OK;
    }
    HIP_TRY(h, hipStreamSynchronize(st));
    sum_counter_banks(res->c, h->host_counters);
    HIP_TRY(h, hipEventElapsedTime(&res->ms_trace, ev[0], ev[1]));
    HIP_TRY(h, hipEventElapsedTime(&res->ms_shade, ev[1], ev[2]));
    res->ms_shade_main = 0.f;
    if (!debug) HIP_TRY(h, hipEventElapsedTime(&res->ms_shade_main, ev[1], ev[4]));
    HIP_TRY(h, hipEventElapsedTime(&res->ms_resolve, ev[2], ev[3]));
    if (!debug && (d_f32 || d_u8)) h->fm.accum_zero_prefix = zero_before;
    return RT_OK;
}

void add_band_stats(rt_handle* h, rt_stats& s, const unsigned long long* c, float ms_trace, float ms_shade, float ms_shade_main, float ms_resolve, size_t rows) {
    s.primary_rays += c[CNT_PRIMARY]; s.secondary_rays += c[CNT_SECONDARY];
    s.depth_cut += c[CNT_DEPTH_CUT]; s.shaded_points += c[CNT_SHADED];
    s.ms_trace += ms_trace; s.ms_shade += ms_shade; s.ms_shade_main += ms_shade_main; s.ms_resolve += ms_resolve;
    s.shadow_samples += c[CNT_SAMPLES];
    for (int k = 0; k < 7; ++k) s.wave_light[k] += c[CNT_WL_NONE + k];
    s.wave_light[7] += c[CNT_EXACT];
    s.bands += 1;
    s.rows_rendered += rows;
    s.shadow_rays += (uint64_t)std::llround((double)c[CNT_SHADED] * h->shadow_rays_per_hit);
    s.flop_shade += (double)c[CNT_SHADED] * h->flop_per_hit;
}

// wait for the oldest enqueued frame and add its statistics to h->async_sum
int collect_oldest(rt_handle* h) {
    rt_handle::Pending& e = *h->in_flight.front();
    h->pending_free.push_back(&e);
    h->in_flight.pop_front();
    HIP_TRY(h, hipEventSynchronize(e.ev[3]));
    float t0 = 0, t1 = 0, t1m = 0, t2 = 0, all = 0;
    HIP_TRY(h, hipEventElapsedTime(&t0, e.ev[0], e.ev[1]));
    HIP_TRY(h, hipEventElapsedTime(&t1, e.ev[1], e.ev[2]));
    HIP_TRY(h, hipEventElapsedTime(&t1m, e.ev[1], e.ev[4]));
    HIP_TRY(h, hipEventElapsedTime(&t2, e.ev[2], e.ev[3]));
    HIP_TRY(h, hipEventElapsedTime(&all, e.ev[0], e.ev[3]));
    unsigned long long c[CNT_N];
    sum_counter_banks(c, e.counters);
    if (c[CNT_OVERFLOW_RAY] || c[CNT_OVERFLOW_SHADE] || c[CNT_EXACT] > h->fm.exact_capacity) h->async_overflow = true;
    add_band_stats(h, h->async_sum, c, t0, t1, t1m, t2, e.rows);
    h->async_sum.ms_total += all;
    return RT_OK;
}

int drain(rt_handle* h) {
    while (!h->in_flight.empty()) {
        int rc = collect_oldest(h);
        if (rc != RT_OK) {
            for (auto* e : h->in_flight) h->pending_free.push_back(e);
            h->in_flight.clear();
            return rc;
        }
    }
    return RT_OK;
}

// An entry for the next enqueued frame.  Frames that have finished are collected on the way (a query, no wait); the
// call waits for the oldest frame only when async_max frames are in flight, so a caller that queues a long sequence is
// never paced by the completion of its first frames.
int take_pending(rt_handle* h, rt_handle::Pending** out) {
    while (!h->in_flight.empty() && hipEventQuery(h->in_flight.front()->ev[3]) == hipSuccess) {
        int rc = collect_oldest(h);
        if (rc != RT_OK) return rc;
    }
    if (h->pending_free.empty() && (int)h->in_flight.size() >= h->async_max) {
        int rc = collect_oldest(h);
        if (rc != RT_OK) return rc;
    }
    if (h->pending_free.empty()) {
        const int n = h->in_flight.empty() ? kAsyncFirst : (int)std::min<size_t>(h->in_flight.size(), 256);
        for (int k = 0; k < n; ++k) {
            auto* e = new rt_handle::Pending();
            h->pending_free.push_back(e);   // owned by the handle from here on (rt_destroy releases partly built entries too)
            for (auto& v : e->ev) HIP_TRY(h, hipEventCreate(&v));
            HIP_TRY(h, hipHostMalloc((void**)&e->counters, sizeof(unsigned long long) * kCounterWords));
        }
    }
    *out = h->pending_free.back();
    h->pending_free.pop_back();
    return RT_OK;
}

int render_impl(rt_handle* h, const rt_render_params* p, float* d_f32, uint8_t* d_u8, hipStream_t st, rt_stats* stats) {
    { int rc0 = drain(h); if (rc0 != RT_OK) return rc0; }
    std::vector<int32_t> rows;
    selected_rows(p, rows);
    rt_stats s{};
    s.rows_rendered = rows.size();
    if (rows.empty()) { if (stats) *stats = s; return RT_OK; }
    { int rc0 = ensure_rows(h, rows.size()); if (rc0 != RT_OK) return rc0; }
    const size_t W = (size_t)p->width;
    size_t band_rows = std::max<size_t>(1, std::min(rows.size(), h->max_band_pixels / W));
    int rc = ensure_band_capacity(h, band_rows * W);
    if (rc != RT_OK) return rc;

    hipEvent_t e_begin = h->ev[5];
    HIP_TRY(h, hipEventRecord(e_begin, st));
    hipEvent_t e_end = h->ev[6];
    size_t r0 = 0;
    while (r0 < rows.size()) {
        size_t n = std::min(band_rows, rows.size() - r0);
        BandResult br{};
        rc = run_band(h, p, rows.data() + r0, (int)n, d_f32 ? d_f32 + r0 * W * 3 : nullptr,
                      d_u8 ? d_u8 + r0 * W * 3 : nullptr, st, false, &br);
        if (rc != RT_OK) return rc;
        if (br.c[CNT_EXACT] > h->fm.exact_capacity && !br.c[CNT_OVERFLOW_RAY] && !br.c[CNT_OVERFLOW_SHADE]) {
            // more (hit, light) pairs needed their own sphere in float64 than the list holds (a camera inside a glass
            // ball): the counter says how many; rebuild the arena with a list of that size and render the band again
            h->exact_want = (size_t)br.c[CNT_EXACT] + (size_t)br.c[CNT_EXACT] / 4 + 1024;
            if (h->exact_want > 0x7FFFFFFFu) return fail(h, RT_ERR_QUEUE, "float64 pair list beyond 2^31 entries");
            rc = ensure_band_capacity(h, h->fm.band_capacity);
            if (rc != RT_OK) return rc;
            continue;
        }
        if (br.c[CNT_OVERFLOW_RAY] || br.c[CNT_OVERFLOW_SHADE]) {
            if (!br.c[CNT_OVERFLOW_RAY]) {
                // Only the shade queue overflowed, and the counters say by how much (records beyond the capacity are
                // counted, not written): enlarge the arena to what this band needs and render it again in one piece.
                // The arena stays that large, so the following frames of the scene do not pay for the discovery again
                // (a deep ray tree over a 4K band needs a few GB out of 288).
                // (+ what the ray-tree kernel's workgroups reserve without using: at most twice RT_TAIL_RESERVE slots each)
                const size_t slots = (size_t)h->fm.sq.base + (size_t)br.c[CNT_SHADED] + (size_t)br.c[CNT_OVERFLOW_SHADE] +
                                     (size_t)h->n_cu * (h->grid.cell_start ? RT_TAIL_PER_CU_GRID : RT_TAIL_PER_CU) * 2 * RT_TAIL_RESERVE;
                const size_t need = (slots + slots / 8 + kShadeFactor - 1) / kShadeFactor;
                if (need <= kMaxArenaPixels && need > h->fm.band_capacity) {
                    rc = ensure_band_capacity(h, need);
                    if (rc != RT_OK) return rc;
                    continue;
                }
            }
            if (n == 1) {
                // a single row outgrew the queues: enlarge them (the ray tree can double per level)
                if (h->fm.band_capacity * 2 > kMaxArenaPixels) {
                    return fail(h, RT_ERR_QUEUE, "ray tree of a single row exceeds the largest queue capacity");
                }
                rc = ensure_band_capacity(h, h->fm.band_capacity * 2);
                if (rc != RT_OK) return rc;
                continue;
            }
            band_rows = std::max<size_t>(1, n / 2);  // retry this band with fewer rows
            continue;
        }
        add_band_stats(h, s, br.c, br.ms_trace, br.ms_shade, br.ms_shade_main, br.ms_resolve, 0);
        r0 += n;
    }
    HIP_TRY(h, hipEventRecord(e_end, st));
    HIP_TRY(h, hipEventSynchronize(e_end));
    float ms = 0;
    HIP_TRY(h, hipEventElapsedTime(&ms, e_begin, e_end));
    s.ms_total = ms;
    if (stats) *stats = s;
    return RT_OK;
}

// Uniform grid over the bounding boxes of spheres and boxes (planes go to the global list).  About four
// cells per surface, cubic cells; a surface is listed in every cell its box touches after growing it by
// 1e-6 of the scene extent, which makes the lists robust against rounding at cell borders in the walks.
int build_grid(rt_handle* h, const std::vector<SurfD>& sd, int S) {
    if (h->d_grid_mem) (void)hipFree(h->d_grid_mem);
    h->d_grid_mem = nullptr;
    h->grid = SceneGrid{};
    const bool want = h->grid_override == 1 || (h->grid_override != 0 && S > kGridMinSurfaces);
    if (!want) return RT_OK;
    std::vector<uint32_t> global_items;
    std::vector<int> bounded;
    double lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    auto box_of = [&](const SurfD& a, double* bl, double* bh) {
        if (a.type == RT_SURF_SPHERE) { const double r = std::fabs(a.d[3]); for (int k = 0; k < 3; ++k) { bl[k] = a.d[k] - r; bh[k] = a.d[k] + r; } }
        else for (int k = 0; k < 3; ++k) { bl[k] = std::min(a.d[k], a.d[3 + k]); bh[k] = std::max(a.d[k], a.d[3 + k]); }
    };
    for (int s = 0; s < S; ++s) {
        if (sd[s].type == RT_SURF_PLANE) { global_items.push_back((uint32_t)s); continue; }
        double bl[3], bh[3];
        box_of(sd[s], bl, bh);
        bool finite = true;
        for (int k = 0; k < 3; ++k) finite = finite && std::isfinite(bl[k]) && std::isfinite(bh[k]);
        if (!finite) { global_items.push_back((uint32_t)s); continue; }
        bounded.push_back(s);
        for (int k = 0; k < 3; ++k) { lo[k] = std::min(lo[k], bl[k]); hi[k] = std::max(hi[k], bh[k]); }
    }
    if (bounded.empty()) return RT_OK;
    double ext[3], max_ext = 0;
    for (int k = 0; k < 3; ++k) { ext[k] = hi[k] - lo[k]; max_ext = std::max(max_ext, ext[k]); }
    if (!(max_ext > 0)) max_ext = 1.0;
    const double grow = 1e-6 * max_ext + 1e-9;
    for (int k = 0; k < 3; ++k) { lo[k] -= 2 * grow; hi[k] += 2 * grow; ext[k] = hi[k] - lo[k]; }
    const double target = std::min(h->grid_density * (double)bounded.size(), 2097152.0);
    const double edge = std::cbrt(ext[0] * ext[1] * ext[2] / target);
    SceneGrid g{};
    size_t cells = 1;
    for (int k = 0; k < 3; ++k) {
        int n = edge > 0 ? (int)std::ceil(ext[k] / edge) : 1;
        n = std::max(1, std::min(n, 512));
        g.n[k] = n; g.lo[k] = lo[k]; g.cell[k] = ext[k] / n; g.inv_cell[k] = 1.0 / g.cell[k];
        cells *= (size_t)n;
    }
    auto range_of = [&](int s, int* c0, int* c1) {
        double bl[3], bh[3];
        box_of(sd[s], bl, bh);
        for (int k = 0; k < 3; ++k) {
            c0[k] = std::max(0, std::min(g.n[k] - 1, (int)std::floor((bl[k] - grow - g.lo[k]) * g.inv_cell[k])));
            c1[k] = std::max(0, std::min(g.n[k] - 1, (int)std::floor((bh[k] + grow - g.lo[k]) * g.inv_cell[k])));
        }
    };
    std::vector<uint32_t> start(cells + 1, 0);
    for (int s : bounded) {
        int c0[3], c1[3];
        range_of(s, c0, c1);
        for (int z = c0[2]; z <= c1[2]; ++z) for (int y = c0[1]; y <= c1[1]; ++y) for (int x = c0[0]; x <= c1[0]; ++x)
            start[((size_t)z * g.n[1] + y) * g.n[0] + x + 1] += 1;
    }
    for (size_t c = 0; c < cells; ++c) start[c + 1] += start[c];
    std::vector<uint32_t> items(std::max<size_t>(start[cells], 1)), fill(start.begin(), start.end() - 1);
    for (int s : bounded) {   // ascending surface index => every cell list is ascending
        int c0[3], c1[3];
        range_of(s, c0, c1);
        for (int z = c0[2]; z <= c1[2]; ++z) for (int y = c0[1]; y <= c1[1]; ++y) for (int x = c0[0]; x <= c1[0]; ++x)
            items[fill[((size_t)z * g.n[1] + y) * g.n[0] + x]++] = (uint32_t)s;
    }
    // bounding sphere of every list entry, in FP32 (radius grown by what the FP32 centre may be off)
    std::vector<float> bounds(items.size() * 4, 0.f);
    for (size_t k = 0; k < (size_t)start[cells]; ++k) {
        const SurfD& a = sd[items[k]];
        double c[3], r;
        if (a.type == RT_SURF_SPHERE) { for (int q = 0; q < 3; ++q) c[q] = a.d[q]; r = std::fabs(a.d[3]); }
        else {
            double dd2 = 0;
            for (int q = 0; q < 3; ++q) { c[q] = 0.5 * (a.d[q] + a.d[3 + q]); const double e = 0.5 * (a.d[3 + q] - a.d[q]); dd2 += e * e; }
            r = std::sqrt(dd2);
        }
        const double off = 2e-7 * (std::fabs(c[0]) + std::fabs(c[1]) + std::fabs(c[2]));
        for (int q = 0; q < 3; ++q) bounds[k * 4 + q] = (float)c[q];
        bounds[k * 4 + 3] = (float)((r + off) * (1.0 + 1e-6) + 1e-9) * 1.000001f;
    }
    // per-cell records (SceneGrid::cell_rec)
    std::vector<uint32_t> rec(cells * 16, 0u);
    for (size_t c = 0; c < cells; ++c) {
        const uint32_t k0 = start[c], n = start[c + 1] - start[c];
        uint32_t* r = rec.data() + c * 16;
        r[0] = n;
        for (uint32_t e = 0; e < 3; ++e) {
            r[1 + e] = e < n ? items[k0 + e] : 0xFFFFFFFFu;
            if (e < n) std::memcpy(r + 4 + 4 * e, bounds.data() + (size_t)(k0 + e) * 4, 16);
        }
    }
    const size_t words = rec.size() + bounds.size() + start.size() + items.size() + std::max<size_t>(global_items.size(), 1);
    HIP_TRY(h, hipMalloc(&h->d_grid_mem, words * sizeof(uint32_t)));
    uint32_t* p = h->d_grid_mem;
    HIP_TRY(h, hipMemcpy(p, rec.data(), rec.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
    g.cell_rec = (const uint4*)p; p += rec.size();   // (first: 64-byte records on 64-byte boundaries)
    HIP_TRY(h, hipMemcpy(p, bounds.data(), bounds.size() * sizeof(float), hipMemcpyHostToDevice));
    g.item_bounds = (const float4*)p; p += bounds.size();   // (first: the allocation is 16-byte aligned)
    HIP_TRY(h, hipMemcpy(p, start.data(), start.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
    g.cell_start = p; p += start.size();
    HIP_TRY(h, hipMemcpy(p, items.data(), items.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
    g.items = p; p += items.size();
    if (!global_items.empty())
        HIP_TRY(h, hipMemcpy(p, global_items.data(), global_items.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
    g.global_items = p;
    g.n_global = (int)global_items.size();
    h->grid = g;
    return RT_OK;
}

}  // namespace

extern "C" {

int rt_create(int device, rt_handle** out) {
    if (!out) return RT_ERR_INVALID;
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || device < 0 || device >= n) return RT_ERR_DEVICE;
    rt_handle* h = new rt_handle();
    h->device = device;
    if (hipSetDevice(device) != hipSuccess || hipStreamCreate(&h->stream) != hipSuccess) { delete h; return RT_ERR_DEVICE; }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess) h->n_cu = prop.multiProcessorCount;
    // test / measurement hook: 0 = one launch per level, d >= 1 = run-to-completion kernel from depth d on (default: from the camera rays)
    if (const char* e = std::getenv("RT_TAIL_FROM")) h->tail_override = std::atoi(e);
    if (const char* e = std::getenv("RT_GRID")) h->grid_override = std::atoi(e);  // test hook: force the grid on / off
    if (const char* e = std::getenv("RT_GRID_DENSITY")) { const double v = std::atof(e); if (v >= 0.125 && v <= 64.0) h->grid_density = v; }
    if (const char* e = std::getenv("RT_DELIVER_WGS")) h->deliver_workgroups = std::max(1, std::min(4096, std::atoi(e)));
    if (const char* e = std::getenv("RT_ASYNC_DEPTH")) h->async_max = std::max(1, std::atoi(e));
    if (const char* e = std::getenv("RT_EXACT_MIN")) h->exact_min = (float)std::atof(e);
    if (const char* e = std::getenv("RT_EXACT_LIST")) h->exact_list = (size_t)std::max(1ll, std::atoll(e));
    if (const char* e = std::getenv("RT_MAX_BAND_PIXELS")) {  // test hook: force row bands on small frames
        long v = std::atol(e);
        if (v > 0) h->max_band_pixels = (size_t)v;
    }
    bool ok = hipMalloc(&h->d_counters, sizeof(unsigned long long) * kCounterWords + sizeof(uint32_t) * kQCounts) == hipSuccess;
    if (ok) h->d_qcounts = (uint32_t*)(h->d_counters + kCounterWords);
    for (auto& e : h->ev) ok = ok && hipEventCreate(&e) == hipSuccess;
    ok = ok && hipEventCreateWithFlags(&h->rows_ev, hipEventDisableTiming) == hipSuccess;
    ok = ok && hipHostMalloc((void**)&h->host_counters, sizeof(unsigned long long) * kCounterWords) == hipSuccess;
    ok = ok && hipHostMalloc((void**)&h->host_rows, sizeof(int32_t) * kHostRows) == hipSuccess;
    if (ok) h->host_rows_cap = kHostRows;
    // this device's copy of the static scratch array (each GPU has its own, at its own address)
    if (ok && device < 64) {
        if (!g_scratch_ptr[device]) {
            void* ptr = nullptr;
            ok = hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch)) == hipSuccess && ptr != nullptr;
            if (ok) g_scratch_ptr[device] = ptr;
        }
        h->scratch = g_scratch_ptr[device];
    }
    if (!ok) { rt_destroy(h); return RT_ERR_DEVICE; }
    *out = h;
    return RT_OK;
}

void rt_destroy(rt_handle* h) {
    if (!h) return;
    (void)hipSetDevice(h->device);
    (void)drain(h);
    for (auto* e : h->pending_free) {
        for (auto& v : e->ev) if (v) (void)hipEventDestroy(v);
        if (e->counters) (void)hipHostFree(e->counters);
        delete e;
    }
    free_frame_buffers(h);
    for (void* p : {(void*)h->d_surf, (void*)h->d_surff, (void*)h->d_mat, (void*)h->d_light, (void*)h->d_light64, (void*)h->d_special, (void*)h->d_jitter, (void*)h->d_counters,
                    (void*)h->fm.stack_mem, (void*)h->d_grid_mem, (void*)h->fm.d_row_y, (void*)h->d_out_f32, (void*)h->d_out_u8,
                    (void*)h->dbg.gpixel})
        if (p) (void)hipFree(p);
    for (auto& e : h->ev) if (e) (void)hipEventDestroy(e);
    if (h->rows_ev) (void)hipEventDestroy(h->rows_ev);
    if (h->host_counters) (void)hipHostFree(h->host_counters);
    if (h->host_rows) (void)hipHostFree(h->host_rows);
    if (h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
}

const char* rt_last_error(const rt_handle* h) { return h ? h->err.c_str() : "null handle"; }

int rt_set_scene(rt_handle* h, const rt_scene_desc* d) {
    if (!h) return RT_ERR_INVALID;
    { int rc0 = drain(h); if (rc0 != RT_OK) return rc0; }
    if (!d || d->struct_size != sizeof(rt_scene_desc) || d->abi_version != RT_ABI_VERSION)
        return fail(h, RT_ERR_INVALID, "rt_scene_desc: bad struct_size / abi_version");
    if (d->n_surfaces < 0 || d->n_materials < 0 || d->n_lights < 0) return fail(h, RT_ERR_INVALID, "negative count");
    if (d->n_surfaces >= (1 << 24)) return fail(h, RT_ERR_UNSUPPORTED, "more than 2^24-1 surfaces");
    if (d->n_lights >= (1 << 24)) return fail(h, RT_ERR_UNSUPPORTED, "more than 2^24-1 lights");
    if (d->n_shadow < 1) return fail(h, RT_ERR_INVALID, "n_shadow must be >= 1");
    if (d->n_shadow > 1024) return fail(h, RT_ERR_UNSUPPORTED, "n_shadow above 1024");
    if (d->max_rec < 0 || d->max_rec > 31) return fail(h, RT_ERR_UNSUPPORTED, "max_rec must be in [0, 31] (32-bit ray-tree node ids)");
    if (d->n_surfaces > 0 && (!d->surf_type || !d->surf_param || !d->surf_material)) return fail(h, RT_ERR_INVALID, "null surface arrays");
    if (d->n_surfaces > 0 && d->n_materials < 1) return fail(h, RT_ERR_INVALID, "surfaces without materials");
    if ((d->n_materials > 0 && !d->mat_param) || (d->n_lights > 0 && !d->light_param)) return fail(h, RT_ERR_INVALID, "null material / light arrays");
    HIP_TRY(h, hipSetDevice(h->device));

    const int S = d->n_surfaces, M = d->n_materials, L = d->n_lights;
    std::vector<MatF> mats((size_t)std::max(M, 1));
    std::memset(mats.data(), 0, sizeof(MatF) * mats.size());
    for (int m = 0; m < M; ++m) {
        const double* q = d->mat_param;
        MatF& t = mats[m];
        for (int c = 0; c < 3; ++c) {
            t.kd[c] = (float)q[(0 + c) * M + m]; t.ks[c] = (float)q[(3 + c) * M + m]; t.kr[c] = (float)q[(6 + c) * M + m];
        }
        t.shin = (float)q[9 * M + m];
        t.transp = (float)q[10 * M + m];
        // gates are evaluated on the float64 inputs (ray_tracer.py:192, :226)
        t.has_refl = (q[6 * M + m] > 0 || q[7 * M + m] > 0 || q[8 * M + m] > 0) ? 1 : 0;
        if (q[10 * M + m] > 0 && t.transp == 0.f) t.transp = 1e-45f;
    }
    std::vector<SurfD> sd((size_t)std::max(S, 1));
    std::vector<SurfF> sf((size_t)std::max(S, 1));
    std::memset(sd.data(), 0, sizeof(SurfD) * sd.size());
    std::memset(sf.data(), 0, sizeof(SurfF) * sf.size());
    double test_flops = 0; bool any_box = false;
    for (int s = 0; s < S; ++s) {
        int type = d->surf_type[s], mat = d->surf_material[s];
        if (mat < 0 || mat >= M) return fail(h, RT_ERR_INVALID, "surface " + std::to_string(s) + ": material index out of range");
        v3 p = {d->surf_param[0 * S + s], d->surf_param[1 * S + s], d->surf_param[2 * S + s]};
        double p3 = d->surf_param[3 * S + s];
        SurfD& a = sd[s]; SurfF& f = sf[s];
        a.mat = f.mat = mat;
        f.transp = mats[mat].transp;
        if (type == RT_SURF_SPHERE) {
            a.type = f.type = RT_SURF_SPHERE;
            a.d[0] = p.x; a.d[1] = p.y; a.d[2] = p.z; a.d[3] = p3;
            a.d[4] = p3 * p3;            // sphere.py:38
            a.d[5] = std::pow(p3, 2.0);  // sphere.py:93 (Python float ** 2)
            f.a = make_float4((float)p.x, (float)p.y, (float)p.z, (float)(p3 * p3));
            f.c = make_float4((float)p3, 0.f, 0.f, 0.f);
            test_flops += 20;
        } else if (type == RT_SURF_PLANE || type == RT_SURF_PLANE_RAW) {
            if (type == RT_SURF_PLANE_RAW) p = divs(p, std::sqrt(dotb(p, p)));  // infinite_plane.py:8
            a.type = f.type = RT_SURF_PLANE;
            a.d[0] = p.x; a.d[1] = p.y; a.d[2] = p.z; a.d[3] = p3;
            f.a = make_float4((float)p.x, (float)p.y, (float)p.z, (float)p3);
            test_flops += 12;
        } else if (type == RT_SURF_BOX) {
            a.type = f.type = RT_SURF_BOX;
            double half = p3 / 2.0;  // cube.py:32-34
            a.d[0] = p.x - half; a.d[1] = p.y - half; a.d[2] = p.z - half;
            a.d[3] = p.x + half; a.d[4] = p.y + half; a.d[5] = p.z + half;
            f.a = make_float4((float)a.d[0], (float)a.d[1], (float)a.d[2], 0.f);
            f.b = make_float4((float)a.d[3], (float)a.d[4], (float)a.d[5], 0.f);
            f.c = make_float4((float)p.x, (float)p.y, (float)p.z, (float)(std::fabs(p3) * 0.8660254037844386 * 1.00001));
            test_flops += 24; any_box = true;
        } else {
            return fail(h, RT_ERR_INVALID, "surface " + std::to_string(s) + ": unknown type");
        }
    }
    if (any_box) test_flops += 3;
    {   // twin groups (SurfF::pad, SurfF::b.w): surfaces whose float64 records are equal
        std::map<std::array<double, 8>, int> first;
        for (int s = 0; s < S; ++s) {
            std::array<double, 8> key{};
            for (int k = 0; k < 7; ++k) key[k] = sd[s].d[k];
            key[7] = (double)sd[s].type;
            sf[s].pad = first.emplace(key, s).first->second;
            sf[s].b.w = 1.f;
        }
        for (int s = 0; s < S; ++s) sf[sf[s].pad].b.w *= sf[s].transp;
        for (int s = 0; s < S; ++s) sf[s].b.w = sf[sf[s].pad].b.w;
    }
    std::vector<LightF> lights((size_t)std::max(L, 1));
    std::memset(lights.data(), 0, sizeof(LightF) * lights.size());
    double rays_per_hit = 0;
    std::vector<double> light64((size_t)std::max(L, 1) * 4, 0.0);
    for (int l = 0; l < L; ++l) {
        const double* q = d->light_param;
        LightF& t = lights[l];
        for (int c = 0; c < 3; ++c) { t.pos[c] = (float)q[c * L + l]; t.col[c] = (float)q[(3 + c) * L + l]; }
        t.spec = (float)q[6 * L + l]; t.shadow = (float)q[7 * L + l]; t.radius = (float)q[8 * L + l];
        t.point_path = (q[7 * L + l] == 0 || d->n_shadow == 1) ? 1 : 0;  // ray_tracer.py:36
        for (int c = 0; c < 3; ++c) light64[(size_t)l * 4 + c] = q[c * L + l];
        light64[(size_t)l * 4 + 3] = q[8 * L + l];
        rays_per_hit += t.point_path ? 1.0 : (double)d->n_shadow * d->n_shadow;
    }

    // Camera.__init__ (camera.py:14-16)
    v3 pos = {d->cam_pos[0], d->cam_pos[1], d->cam_pos[2]};
    v3 look = {d->cam_look_at[0], d->cam_look_at[1], d->cam_look_at[2]};
    v3 uph = {d->cam_up[0], d->cam_up[1], d->cam_up[2]};
    v3 fwd = normalize(sub(look, pos));
    v3 right = normalize(cross(fwd, uph));
    v3 up = normalize(cross(right, fwd));
    CameraD cam{};
    cam.pos[0] = pos.x; cam.pos[1] = pos.y; cam.pos[2] = pos.z;
    cam.fwd[0] = fwd.x; cam.fwd[1] = fwd.y; cam.fwd[2] = fwd.z;
    cam.right[0] = right.x; cam.right[1] = right.y; cam.right[2] = right.z;
    cam.up[0] = up.x; cam.up[1] = up.y; cam.up[2] = up.z;
    cam.sdist = d->cam_screen_dist; cam.swidth = d->cam_screen_width;

    for (void** p : {(void**)&h->d_surf, (void**)&h->d_surff, (void**)&h->d_mat, (void**)&h->d_light, (void**)&h->d_light64,
                     (void**)&h->d_special, (void**)&h->d_jitter}) {
        if (*p) (void)hipFree(*p);
        *p = nullptr;
    }
    h->has_scene = false;
    HIP_TRY(h, hipMalloc(&h->d_surf, sizeof(SurfD) * sd.size()));
    HIP_TRY(h, hipMalloc(&h->d_surff, sizeof(SurfF) * sf.size()));
    HIP_TRY(h, hipMalloc(&h->d_mat, sizeof(MatF) * mats.size()));
    HIP_TRY(h, hipMalloc(&h->d_light, sizeof(LightF) * lights.size()));
    HIP_TRY(h, hipMemcpy(h->d_surf, sd.data(), sizeof(SurfD) * sd.size(), hipMemcpyHostToDevice));
    HIP_TRY(h, hipMemcpy(h->d_surff, sf.data(), sizeof(SurfF) * sf.size(), hipMemcpyHostToDevice));
    HIP_TRY(h, hipMemcpy(h->d_mat, mats.data(), sizeof(MatF) * mats.size(), hipMemcpyHostToDevice));
    HIP_TRY(h, hipMemcpy(h->d_light, lights.data(), sizeof(LightF) * lights.size(), hipMemcpyHostToDevice));
    HIP_TRY(h, hipMalloc(&h->d_light64, sizeof(double) * light64.size()));
    HIP_TRY(h, hipMemcpy(h->d_light64, light64.data(), sizeof(double) * light64.size(), hipMemcpyHostToDevice));
    {   // jitter table: the pair words of an N x N grid (rt_device.h), the same for every seed, pixel, node and light
        std::vector<uint32_t> words(rt_pair_table_words((uint32_t)d->n_shadow));
        for (size_t k = 0; k < words.size(); ++k) words[k] = rt_pair_word((uint32_t)k);
        HIP_TRY(h, hipMalloc(&h->d_jitter, sizeof(uint32_t) * words.size()));
        HIP_TRY(h, hipMemcpy(h->d_jitter, words.data(), sizeof(uint32_t) * words.size(), hipMemcpyHostToDevice));
    }
    int rc = build_grid(h, sd, S);
    if (rc != RT_OK) return rc;
    // lights whose centre lies on a surface (ShadeArgs::special): L + 1 offsets, then the surfaces
    {
        std::vector<uint32_t> start((size_t)L + 1, 0u), items;
        for (int l = 0; l < L; ++l) {
            const double* q = d->light_param;
            const double p[3] = {q[0 * L + l], q[1 * L + l], q[2 * L + l]};
            const double tol = 1e-3 * (1.0 + std::max(std::fabs(p[0]), std::max(std::fabs(p[1]), std::fabs(p[2]))));
            for (int s2 = 0; s2 < S; ++s2) {
                const SurfD& a = sd[s2];
                double dist;
                if (a.type == RT_SURF_SPHERE) {
                    const double dx = p[0] - a.d[0], dy = p[1] - a.d[1], dz = p[2] - a.d[2];
                    dist = std::fabs(std::sqrt(dx * dx + dy * dy + dz * dz) - std::fabs(a.d[3]));
                } else if (a.type == RT_SURF_PLANE) {
                    dist = std::fabs(p[0] * a.d[0] + p[1] * a.d[1] + p[2] * a.d[2] - a.d[3]);
                } else {
                    double out2 = 0, in_min = INFINITY;   // outside: distance to the box; inside: to the nearest face
                    for (int k = 0; k < 3; ++k) {
                        const double lo = std::min(a.d[k], a.d[3 + k]), hi = std::max(a.d[k], a.d[3 + k]);
                        const double o = std::max(std::max(lo - p[k], p[k] - hi), 0.0);
                        out2 += o * o;
                        in_min = std::min(in_min, std::min(std::fabs(p[k] - lo), std::fabs(hi - p[k])));
                    }
                    dist = out2 > 0 ? std::sqrt(out2) : in_min;
                }
                if (dist <= tol) items.push_back((uint32_t)s2);
            }
            start[(size_t)l + 1] = (uint32_t)items.size();
        }
        if (!items.empty()) {
            start.insert(start.end(), items.begin(), items.end());
            HIP_TRY(h, hipMalloc(&h->d_special, sizeof(uint32_t) * start.size()));
            HIP_TRY(h, hipMemcpy(h->d_special, start.data(), sizeof(uint32_t) * start.size(), hipMemcpyHostToDevice));
        }
    }
    h->S = S; h->M = M; h->L = L; h->N = d->n_shadow; h->R = d->max_rec;
    for (int c = 0; c < 3; ++c) h->bg[c] = (float)d->background[c];
    h->cam = cam;
    h->shadow_rays_per_hit = rays_per_hit;
    h->has_box = any_box ? 1 : 0;
    h->all_opaque = 1;
    for (int s = 0; s < S; ++s) if (sf[s].transp != 0.f) h->all_opaque = 0;
    h->both_children = false;
    for (int m = 0; m < M; ++m) if (mats[m].has_refl && mats[m].transp > 0.f) h->both_children = true;
    // algorithmic FP32 flop per shaded hit: per shadow ray the surface tests + 26 for building the
    // ray; per (hit, light) 30 for the sampling basis + 52 for the Phong terms; 12 for the blend
    h->flop_per_hit = rays_per_hit * (test_flops + 26.0) + L * 82.0 + 12.0;
    h->has_scene = true;
    return RT_OK;
}

int rt_render_enqueue(rt_handle* h, const rt_render_params* p, void* dev_rgb_f32, void* dev_rgb_u8, void* stream) {
    if (!h) return RT_ERR_INVALID;
    if (!h->has_scene) return fail(h, RT_ERR_NO_SCENE, "rt_render_enqueue before rt_set_scene");
    int rc = validate_params(h, p);
    if (rc != RT_OK) return rc;
    HIP_TRY(h, hipSetDevice(h->device));
    hipStream_t st = stream ? (hipStream_t)stream : h->stream;
    std::vector<int32_t> rows;
    selected_rows(p, rows);
    if (rows.empty()) return RT_OK;
    const size_t pixels = rows.size() * (size_t)p->width;
    if (pixels > h->max_band_pixels) return fail(h, RT_ERR_UNSUPPORTED, "rt_render_enqueue: the selection needs more than one row band");
    // anything that reallocates device memory waits for the frames in flight first
    const size_t need_px = std::max(pixels, (primary_slots((size_t)p->width, rows.size()) * 2 + kShadeFactor - 1) / kShadeFactor);
    const bool need_rays = choose_tail_depth(h, h->R > 0 ? h->R : 1) > 0;
    if (rows.size() > h->fm.row_y_cap || rows.size() > h->host_rows_cap || need_px > h->fm.band_capacity ||
        (h->L > 0 ? h->L : 1) > h->fm.arena_lights || (need_rays && !h->fm.arena_has_rays)) {
        rc = drain(h);
        if (rc != RT_OK) return rc;
        HIP_TRY(h, hipStreamSynchronize(st));
        rc = ensure_rows(h, rows.size());
        if (rc != RT_OK) return rc;
    }
    rc = ensure_band_capacity(h, need_px);  // no-op unless drained above (or the first frame)
    if (rc != RT_OK) return rc;
    rt_handle::Pending* e = nullptr;
    rc = take_pending(h, &e);
    if (rc != RT_OK) return rc;
    e->rows = rows.size();
    BandResult br{};
    rc = run_band(h, p, rows.data(), (int)rows.size(), (float*)dev_rgb_f32, (uint8_t*)dev_rgb_u8, st, false, &br, e->ev, e->counters);
    if (rc != RT_OK) { h->pending_free.push_back(e); return rc; }
    h->in_flight.push_back(e);
    return RT_OK;
}

int rt_render_wait(rt_handle* h, rt_stats* stats) {
    if (!h) return RT_ERR_INVALID;
    HIP_TRY(h, hipSetDevice(h->device));
    int rc = drain(h);
    if (stats) *stats = h->async_sum;
    const bool overflow = h->async_overflow;
    h->async_sum = rt_stats{};
    h->async_overflow = false;
    if (rc != RT_OK) return rc;
    if (overflow) return fail(h, RT_ERR_QUEUE, "an enqueued frame outgrew the ray / shade queues: render it again with rt_render_device");
    return RT_OK;
}

int rt_deliver_rows(rt_handle* h, void* host_dst, const void* dev_src, const int32_t* dev_src_row, int64_t rows, int64_t row_bytes,
                    int workgroups, void* stream) {
    if (!h) return RT_ERR_INVALID;
    if (rows < 0 || row_bytes < 0 || rows > 0x7FFFFFFFll || (rows && row_bytes && (!host_dst || !dev_src)))
        return fail(h, RT_ERR_INVALID, "rt_deliver_rows: bad rows / row_bytes / pointers");
    if (!rows || !row_bytes) return RT_OK;
    HIP_TRY(h, hipSetDevice(h->device));
    void* mapped = nullptr;
    if (hipHostGetDevicePointer(&mapped, host_dst, 0) != hipSuccess || !mapped) {
        (void)hipGetLastError();
        return fail(h, RT_ERR_INVALID, "rt_deliver_rows: host_dst is not pinned, device-visible host memory (hipHostMalloc / torch pin_memory)");
    }
    hipStream_t st = stream ? (hipStream_t)stream : h->stream;
    if (workgroups < 0 || workgroups > 4096) return fail(h, RT_ERR_INVALID, "rt_deliver_rows: workgroups outside [0, 4096]");
    const int wgs = h->deliver_workgroups ? h->deliver_workgroups : (workgroups ? workgroups : 8);
    rt_launch_deliver(mapped, dev_src, dev_src_row, (uint32_t)rows, (size_t)row_bytes, wgs, st);
    HIP_TRY(h, hipGetLastError());
    return RT_OK;
}

int rt_rows_selected(const rt_render_params* p) {
    if (!p || p->stripe_rows < 1 || p->stripe_count < 1) return RT_ERR_INVALID;
    int n = 0;
    for (int y = p->row_begin; y < p->row_end; ++y)
        if ((y / p->stripe_rows) % p->stripe_count == p->stripe_index) ++n;
    return n;
}

int rt_render_device(rt_handle* h, const rt_render_params* p, void* dev_rgb_f32, void* dev_rgb_u8, void* stream, rt_stats* stats) {
    if (!h) return RT_ERR_INVALID;
    if (!h->has_scene) return fail(h, RT_ERR_NO_SCENE, "rt_render before rt_set_scene");
    int rc = validate_params(h, p);
    if (rc != RT_OK) return rc;
    HIP_TRY(h, hipSetDevice(h->device));
    hipStream_t st = stream ? (hipStream_t)stream : h->stream;
    return render_impl(h, p, (float*)dev_rgb_f32, (uint8_t*)dev_rgb_u8, st, stats);
}

int rt_render_device_noalloc(rt_handle* h, const rt_render_params* p, void* dev_rgb_f32, void* dev_rgb_u8, void* stream, rt_stats* stats) {
    if (!h) return RT_ERR_INVALID;
    if (!h->has_scene) return fail(h, RT_ERR_NO_SCENE, "rt_render_device_noalloc before rt_set_scene");
    int rc = validate_params(h, p);
    if (rc != RT_OK) return rc;
    HIP_TRY(h, hipSetDevice(h->device));
    if (!h->scratch) return fail(h, RT_ERR_UNSUPPORTED, "rt_render_device_noalloc: no scratch array on this device");
    hipStream_t st = stream ? (hipStream_t)stream : h->stream;
    rc = drain(h);
    if (rc != RT_OK) return rc;
    const size_t n_rows = (size_t)rt_rows_selected(p), W = (size_t)p->width;
    const size_t pixels = n_rows * W;
    if (pixels > (size_t)RT_NOALLOC_MAX_PIXELS)
        return fail(h, RT_ERR_UNSUPPORTED, "rt_render_device_noalloc: " + std::to_string(pixels) + " pixels selected, the limit is " +
                                           std::to_string((size_t)RT_NOALLOC_MAX_PIXELS) + " (RT_NOALLOC_MAX_PIXELS): use rt_render_device");
    if (n_rows > h->host_rows_cap) return fail(h, RT_ERR_UNSUPPORTED, "rt_render_device_noalloc: more rows than the row staging buffer holds");
    if (pixels == 0) { if (stats) *stats = rt_stats{}; return RT_OK; }
    // lay the band's queues out in the scratch array: row list first, then the queues, the rest is the ray tree's stack
    const int lights = h->L > 0 ? h->L : 1;
    const bool need_rays = choose_tail_depth(h, h->R > 0 ? h->R : 1) > 0;
    const size_t band_px = std::max(pixels, (primary_slots(W, n_rows) * 2 + kShadeFactor - 1) / kShadeFactor);
    const BandLayout b = band_layout(band_px, lights, need_rays, 0, 0);
    const size_t rows_bytes = al256(n_rows * sizeof(int32_t));
    if (rows_bytes + b.bytes > (size_t)RT_NOALLOC_SCRATCH_BYTES)
        return fail(h, RT_ERR_UNSUPPORTED, "rt_render_device_noalloc: the queues of this frame need " + std::to_string(rows_bytes + b.bytes) +
                                           " bytes, the static scratch holds " + std::to_string((size_t)RT_NOALLOC_SCRATCH_BYTES) +
                                           ": use rt_render_device");
    const FrameMem saved = h->fm;
    const size_t exact_want_saved = h->exact_want;
    h->exact_want = 0;
    h->fm = FrameMem{};
    h->rows_on_device.clear();
    h->no_alloc = true;
    char* base = (char*)h->scratch;
    h->fm.d_row_y = (int32_t*)base; h->fm.row_y_cap = n_rows;
    char* end = carve_band(h, base + rows_bytes, b, band_px, lights, need_rays);
    point_tail_stack(h, end, (size_t)((char*)h->scratch + (size_t)RT_NOALLOC_SCRATCH_BYTES - end) / 68);
    const size_t max_band_saved = h->max_band_pixels;
    h->max_band_pixels = std::max(h->max_band_pixels, pixels);   // one band: the layout above is for the whole selection
    rc = render_impl(h, p, (float*)dev_rgb_f32, (uint8_t*)dev_rgb_u8, st, stats);
    h->max_band_pixels = max_band_saved;
    h->no_alloc = false;
    h->exact_want = exact_want_saved;
    h->fm = saved;                 // the handle's own arena (if any) serves the other entry points again
    h->rows_on_device.clear();
    return rc;
}

int rt_render(rt_handle* h, const rt_render_params* p, float* host_rgb_f32, uint8_t* host_rgb_u8, rt_stats* stats) {
    if (!h) return RT_ERR_INVALID;
    if (!h->has_scene) return fail(h, RT_ERR_NO_SCENE, "rt_render before rt_set_scene");
    int rc = validate_params(h, p);
    if (rc != RT_OK) return rc;
    HIP_TRY(h, hipSetDevice(h->device));
    const size_t values = (size_t)rt_rows_selected(p) * p->width * 3;
    if (values > h->out_cap) {
        if (h->d_out_f32) (void)hipFree(h->d_out_f32);
        if (h->d_out_u8) (void)hipFree(h->d_out_u8);
        h->d_out_f32 = nullptr; h->d_out_u8 = nullptr; h->out_cap = 0;
        HIP_TRY(h, hipMalloc(&h->d_out_f32, values * sizeof(float)));
        HIP_TRY(h, hipMalloc(&h->d_out_u8, values));
        h->out_cap = values;
    }
    rc = render_impl(h, p, host_rgb_f32 ? h->d_out_f32 : nullptr, host_rgb_u8 ? h->d_out_u8 : nullptr, h->stream, stats);
    if (rc != RT_OK) return rc;
    if (host_rgb_f32 && values) HIP_TRY(h, hipMemcpy(host_rgb_f32, h->d_out_f32, values * sizeof(float), hipMemcpyDeviceToHost));
    if (host_rgb_u8 && values) HIP_TRY(h, hipMemcpy(host_rgb_u8, h->d_out_u8, values, hipMemcpyDeviceToHost));
    return RT_OK;
}

int rt_debug_hits(rt_handle* h, const rt_render_params* p, uint32_t capacity, uint32_t* pixel, uint32_t* node, int32_t* surf,
                  double* t, double* P, double* n, uint32_t* flags, uint32_t* count) {
    if (!h) return RT_ERR_INVALID;
    if (!h->has_scene) return fail(h, RT_ERR_NO_SCENE, "rt_debug_hits before rt_set_scene");
    int rc = validate_params(h, p);
    if (rc != RT_OK) return rc;
    if (!capacity || !count) return fail(h, RT_ERR_INVALID, "capacity / count");
    HIP_TRY(h, hipSetDevice(h->device));
    std::vector<int32_t> rows;
    selected_rows(p, rows);
    *count = 0;
    if (rows.empty()) return RT_OK;
    const size_t pixels = rows.size() * (size_t)p->width;
    if (pixels > h->max_band_pixels) return fail(h, RT_ERR_UNSUPPORTED, "debug probe is limited to one band");
    { int rc0 = drain(h); if (rc0 != RT_OK) return rc0; }
    rc = ensure_rows(h, rows.size());
    if (rc != RT_OK) return rc;
    rc = ensure_band_capacity(h, pixels);
    if (rc != RT_OK) return rc;
    // debug arrays: one allocation, freed with the handle or on the next probe
    if (h->dbg.gpixel) (void)hipFree(h->dbg.gpixel);
    h->dbg = HitDebug{};
    const size_t c = capacity;
    char* base = nullptr;
    HIP_TRY(h, hipMalloc(&base, c * (4 + 4 + 4 + 4 + 8 + 24 + 24) + 1024));
    HitDebug g{};
    g.gpixel = (uint32_t*)base; g.node = g.gpixel + c; g.surf = (int32_t*)(g.node + c); g.flags = (uint32_t*)(g.surf + c);
    g.t = (double*)(base + (c * 16 + 255) / 256 * 256); g.P = g.t + c; g.n = g.P + 3 * c;
    g.count = h->d_qcounts + 3; g.capacity = capacity;
    h->dbg = g;
    BandResult br{};
    rc = run_band(h, p, rows.data(), (int)rows.size(), nullptr, nullptr, h->stream, true, &br);
    if (rc != RT_OK) return rc;
    if (br.c[CNT_OVERFLOW_RAY] || br.c[CNT_OVERFLOW_SHADE]) return fail(h, RT_ERR_QUEUE, "debug probe overflowed the queues");
    uint32_t got = 0;
    HIP_TRY(h, hipMemcpy(&got, g.count, 4, hipMemcpyDeviceToHost));
    *count = got;
    const size_t m = std::min<size_t>(got, c);
    if (m) {
        HIP_TRY(h, hipMemcpy(pixel, g.gpixel, m * 4, hipMemcpyDeviceToHost));
        HIP_TRY(h, hipMemcpy(node, g.node, m * 4, hipMemcpyDeviceToHost));
        HIP_TRY(h, hipMemcpy(surf, g.surf, m * 4, hipMemcpyDeviceToHost));
        HIP_TRY(h, hipMemcpy(flags, g.flags, m * 4, hipMemcpyDeviceToHost));
        HIP_TRY(h, hipMemcpy(t, g.t, m * 8, hipMemcpyDeviceToHost));
        HIP_TRY(h, hipMemcpy(P, g.P, m * 24, hipMemcpyDeviceToHost));
        HIP_TRY(h, hipMemcpy(n, g.n, m * 24, hipMemcpyDeviceToHost));
    }
    return RT_OK;
}

void rt_jitter_bits(uint32_t seed, uint32_t pixel, uint32_t node, uint32_t light, uint32_t n, uint32_t s, uint32_t* u8, uint32_t* v8) {
    uint32_t u = 0, v = 0;
    if (n) rt_sample_bits(rt_grid_key(rt_seed_key(seed), pixel, node, light), n, s / n, s % n, u, v);
    if (u8) *u8 = u;
    if (v8) *v8 = v;
}

}  // extern "C"
