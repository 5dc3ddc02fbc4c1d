/*
 * rt_amd.h -- C ABI of the MI355X renderer for the reference's per-pixel frame loop.
 *
 * The reference (a pure-Python program) has no FFI layer; its only seam is the
 * call `image_array = render(scene_settings, args.width, args.height)` at
 * /root/reference/ray_tracer.py:436 (definition: ray_tracer.py:297-368).  A
 * replacement body for that function packs `scene_settings` into the plain
 * arrays below and makes the calls declared here through ctypes.  Everything
 * is `extern "C"`, plain pointers and sizes; no C++ or torch types cross it.
 *
 * Conventions
 *   - every function returns RT_OK (0) or a negative rt_status; none throws,
 *     none exits.  rt_last_error() gives the text of the last failure.
 *   - the caller owns every host buffer for the duration of a call; the library
 *     owns all device memory and copies the scene in rt_set_scene().
 *   - a handle is bound to one device and one stream and is not re-entrant;
 *     distinct handles are independent (one per rank / GPU).
 *   - numbers in rt_scene_desc are the RAW parsed values of the scene file.  The
 *     library derives the camera basis, unit plane normals, r*r, pow(r,2) ...
 *     itself with the reference's rounding, so results do not depend on the
 *     NumPy/BLAS installed next to the caller.
 */
#ifndef RT_AMD_H
#define RT_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RT_ABI_VERSION 3

typedef enum rt_status {
    RT_OK = 0,
    RT_ERR_INVALID = -1,       /* bad argument / inconsistent descriptor            */
    RT_ERR_NO_SCENE = -2,      /* rt_render before rt_set_scene                     */
    RT_ERR_DEVICE = -3,        /* HIP runtime error (text in rt_last_error)         */
    RT_ERR_OOM = -4,           /* device allocation failed                          */
    RT_ERR_QUEUE = -5,         /* ray tree outgrew the queues even at minimum band  */
    RT_ERR_UNSUPPORTED = -6
} rt_status;

/* surface kinds, in the order of the reference's classes
 * (surfaces/sphere.py, surfaces/infinite_plane.py, surfaces/cube.py)            */
#define RT_SURF_SPHERE 0 /* p0..p2 centre, p3 radius                                */
#define RT_SURF_PLANE 1  /* p0..p2 UNIT normal as InfinitePlane stores it
                            (infinite_plane.py:8), used as is; p3 offset            */
#define RT_SURF_BOX 2    /* p0..p2 centre, p3 edge length                           */
#define RT_SURF_PLANE_RAW 3 /* p0..p2 normal as written in the scene file; the library
                            normalises it (offset untouched, infinite_plane.py:9)   */

#define RT_MAT_STRIDE 11  /* kd[3] ks[3] kr[3] shininess transparency  (ray_tracer.py:393) */
#define RT_LIGHT_STRIDE 9 /* pos[3] colour[3] specular shadow radius   (ray_tracer.py:405) */

/* Scene as parsed by parse_scene_file (ray_tracer.py:371-419), structure-of-arrays:
 * surf_param[c*n_surfaces + s], mat_param[c*n_materials + m], light_param[c*n_lights + l].
 * Surfaces keep FILE ORDER (ties in find_nearest_intersection go to the earlier
 * surface, scene_settings.py:35-40).  surf_material is 0-based.                   */
typedef struct rt_scene_desc {
    uint32_t struct_size; /* sizeof(rt_scene_desc), checked                        */
    uint32_t abi_version; /* RT_ABI_VERSION                                        */
    double cam_pos[3];
    double cam_look_at[3];
    double cam_up[3];
    double cam_screen_dist;
    double cam_screen_width;
    double background[3];
    int32_t n_shadow;  /* int(root_number_shadow_rays)        (ray_tracer.py:312)   */
    int32_t max_rec;   /* depth levels traced: ceil(max_recursions), >= 0 (:261)    */
    int32_t n_surfaces;
    int32_t n_materials;
    int32_t n_lights;
    int32_t reserved;
    const int32_t* surf_type;     /* [n_surfaces]                                   */
    const double* surf_param;     /* [4][n_surfaces]                                */
    const int32_t* surf_material; /* [n_surfaces], 0-based                          */
    const double* mat_param;      /* [RT_MAT_STRIDE][n_materials]                   */
    const double* light_param;    /* [RT_LIGHT_STRIDE][n_lights]                    */
} rt_scene_desc;

/* Which rows of the frame this call renders: rows y with
 *   row_begin <= y < row_end  and  ((y / stripe_rows) % stripe_count) == stripe_index.
 * stripe_count = 1 renders the whole range.  Ranks of a multi-GPU job use
 * stripe_count = world size, stripe_index = rank: interleaved stripes balance sky
 * against objects, and the keyed jitter stream makes every partition reproduce the
 * single-GPU image.  Output rows are packed in increasing y.                       */
typedef struct rt_render_params {
    uint32_t struct_size;
    int32_t width;
    int32_t height;
    int32_t row_begin;
    int32_t row_end;
    int32_t stripe_rows;
    int32_t stripe_count;
    int32_t stripe_index;
    uint32_t seed;    /* jitter stream seed (same value on every rank)              */
    uint32_t flags;   /* RT_FLAG_*                                                  */
} rt_render_params;

#define RT_FLAG_NONE 0u
#define RT_FLAG_GENERAL_SHADE 1u /* force the any-scene-size shading kernel (cross-check of the
                                    register-resident small-scene kernel)                  */
#define RT_FLAG_NO_CULL 2u       /* evaluate every shadow ray against every surface: no per-
                                    (hit, light) classification of surfaces (A/B measurements) */

/* Counters use the definitions both sides of the comparison share:
 * primary = pixels; secondary = reflection/transparency rays actually intersected
 * with the scene; depth_cut = children that hit the recursion limit (they return
 * the background without being traced, ray_tracer.py:261-262) ; shaded = hits that
 * were Phong-shaded; shadow = per (shaded hit, light) 1 if N==1 or
 * shadow_intensity==0 else N*N (ray_tracer.py:36, :52).                            */
typedef struct rt_stats {
    uint64_t primary_rays;
    uint64_t secondary_rays;
    uint64_t depth_cut;
    uint64_t shaded_points;
    uint64_t shadow_rays;
    uint64_t rows_rendered;
    uint32_t bands;           /* row bands the frame was split into (queue capacity) */
    uint32_t reserved;
    double ms_total;          /* stream time of the whole call incl. copies          */
    double ms_trace;          /* nearest-hit / ray-tree kernels                      */
    double ms_shade;          /* shading + soft-shadow kernel (the dominant one)     */
    double ms_resolve;        /* accumulate -> clip*255 -> f32/u8                    */
    double flop_shade;        /* algorithmic FP32 flop of the shade kernel launches  */
    uint64_t wave_light[8];   /* how the shading kernel resolved each (wavefront, light):
                                 [0] no surface in reach, ratio = 1 without sampling;
                                 [1..4] compact loop with 1..4 linear tests per sample;
                                 [5] generic loop (every ray against every surface);
                                 [6] single-ray path (N == 1 or shadow_intensity == 0);
                                 [7] not a wavefront class: (hit, light) pairs whose own-sphere samples were
                                     evaluated in float64 (hits on a sphere with c > 0; lights lying on a surface:
                                     raytracer_amd/csrc/rt_exact.h) */
    double ms_shade_main;     /* of ms_shade: the dominant kernel alone (k_shade_fast; k_shade_general for scenes of more
                                 than 16 surfaces), HIP events around that one launch                              */
    uint64_t shadow_samples;  /* shadow-ray samples the kernels really generated and tested, counted per lane that
                                 holds a hit: shadow_rays above is the reference's count (N*N per hit and light),
                                 this is what is left of it after whole (wavefront, light) pairs and whole sample
                                 tiles were decided without sampling                                               */
} rt_stats;

typedef struct rt_handle rt_handle;

/* Replaces nothing in the reference (it has no device state); creates the per-GPU
 * context that the body of render() (ray_tracer.py:297) uses.                      */
int rt_create(int device, rt_handle** out);
void rt_destroy(rt_handle* h);
const char* rt_last_error(const rt_handle* h);

/* Replaces the SceneSettings object handed to render() (scene_settings.py:6-17,
 * built at ray_tracer.py:415-418): uploads surfaces, materials, lights, camera.   */
int rt_set_scene(rt_handle* h, const rt_scene_desc* scene);

/* Replaces the pixel loop of render() (ray_tracer.py:317-331): camera ray
 * (camera.py:18-44), trace_ray (ray_tracer.py:242-294) and clip(c,0,1)*255.
 * host_rgb_f32 : rows*width*3 floats in [0,255] (the array render() returns), or NULL
 * host_rgb_u8  : the same after np.uint8 truncation (ray_tracer.py:438), or NULL
 * stats        : optional.                                                         */
int rt_render(rt_handle* h, const rt_render_params* p, float* host_rgb_f32, uint8_t* host_rgb_u8,
              rt_stats* stats);

/* Same, but the outputs are DEVICE pointers on h's device (e.g. torch tensors that
 * feed an RCCL gather) and nothing is copied to the host; work is enqueued on
 * `stream` (a hipStream_t, NULL = the handle's own stream) and the call returns
 * after the stream has drained so that stats are valid.                            */
int rt_render_device(rt_handle* h, const rt_render_params* p, void* dev_rgb_f32, void* dev_rgb_u8,
                     void* stream, rt_stats* stats);

/* No counterpart in the reference.  For callers that cannot allocate device memory: tight serving loops, harnesses
 * that forbid hipMalloc, memory budgets fixed at start-up.  The call still SYNCHRONISES (it waits for `stream` and reads
 * its counters on the host), so it must not be made on a stream that is being captured into a HIP graph: "allocates
 * nothing" is all it promises.  Same contract as rt_render_device (device
 * output pointers, work on `stream`, returns after the stream has drained), but the ray / shade queues, the
 * accumulators and the work lists live in a static __device__ array of RT_NOALLOC_SCRATCH_BYTES that the library
 * holds on every GPU it runs on until the process ends, and the call makes no hipMalloc / hipFree (rt_create
 * and rt_set_scene, the set-up calls, still allocate).  Limits: at most RT_NOALLOC_MAX_PIXELS selected pixels per
 * call, and the queues cannot grow -- a selection above the limit, a scene whose queues do not fit, or a ray
 * tree that outgrows them (a deep tree of reflecting AND transparent hits) fails with RT_ERR_UNSUPPORTED /
 * RT_ERR_QUEUE and a message that says so; render such frames with rt_render_device.  All handles of a GPU share
 * that one array, and it keeps the last call's contents (the call clears what has to start at zero): calls that
 * overlap on one GPU -- two handles, two threads, two streams -- race and must be serialised by the caller.     */
#define RT_NOALLOC_SCRATCH_BYTES (64u << 20)
#define RT_NOALLOC_MAX_PIXELS (1 << 17)
int rt_render_device_noalloc(rt_handle* h, const rt_render_params* p, void* dev_rgb_f32, void* dev_rgb_u8,
                             void* stream, rt_stats* stats);

/* Frame loops that must not stall on the host between frames (serving, the multi-GPU
 * gather): rt_render_enqueue puts one frame on `stream` like rt_render_device and returns
 * at once; rt_render_wait blocks until every enqueued frame is done, reports whether one
 * of them outgrew the ray / shade queues (RT_ERR_QUEUE: that frame's output is invalid and
 * has to be rendered again with rt_render_device, which resizes) and returns the SUMS of
 * the frames' statistics.  The selection must fit one row band (at most 3840x2160 pixels);
 * all frames between two waits must use the same stream; the outputs of frame k may be
 * consumed by later work on that stream without waiting.  Finished frames are collected
 * on the way (a query, not a wait), so a caller may queue long sequences: an enqueue waits
 * for the oldest frame only with 1024 frames in flight.  Every other call on the handle
 * first waits for the enqueued frames.  No counterpart in the reference (its render() is
 * synchronous by nature).                                                           */
int rt_render_enqueue(rt_handle* h, const rt_render_params* p, void* dev_rgb_f32, void* dev_rgb_u8,
                      void* stream);
int rt_render_wait(rt_handle* h, rt_stats* stats);

/* Hands finished rows to the host: the array the caller of the reference's render() receives and converts
 * (ray_tracer.py:436-439).  Puts ONE small kernel on `stream` (NULL = the handle's stream) that copies `rows` rows
 * of `row_bytes` bytes from device memory `dev_src` into `host_dst`, which must be pinned, device-visible host
 * memory (hipHostMalloc, torch pin_memory).  With `dev_src_row` (device array of `rows` int32, may be NULL) the
 * destination row y is source row dev_src_row[y]: rank 0 of a multi-GPU job assembles the frame out of the gathered
 * stripes on the way.  Returns at once; the rows are in host memory when `stream` has passed the kernel.
 * `workgroups` (0 = 8) sets the footprint: 6 vector registers per lane, so its waves fit beside the persistent
 * waves of the shading kernel, and few workgroups, because a PCIe-bound kernel needs few -- 8 move a 4K frame in
 * 0.8 ms without slowing the next frame's kernels, 64 reach the PCIe rate (0.45 ms).  The runtime's own
 * device-to-host copy is a blit kernel with a wave on every slot of the chip: measured on the 4K pool frame, queued
 * frames delivered with it take 2.5-3.2 ms each depending on how the copies fall, 2.44-2.55 ms with this kernel
 * (2.40-2.45 ms with the frames left on the device).                                                             */
int rt_deliver_rows(rt_handle* h, void* host_dst, const void* dev_src, const int32_t* dev_src_row, int64_t rows,
                    int64_t row_bytes, int workgroups, void* stream);

/* Number of rows selected by p (size of the output in rows).                       */
int rt_rows_selected(const rt_render_params* p);

/* Parity probe: runs only the ray-tree part for the rows selected by p and returns,
 * for every nearest-hit query (scene_settings.py:19), what find_nearest_intersection
 * returns in the reference: surface index (-1 = miss), t, hit point, normal -- in
 * float64, so tests can require bit equality -- plus the own-surface flags handed to
 * the shading kernel.  Arrays hold `capacity` records (P, n: 3 doubles each); *count
 * receives the number of queries (records beyond capacity are dropped).             */
int rt_debug_hits(rt_handle* h, const rt_render_params* p, uint32_t capacity, uint32_t* pixel,
                  uint32_t* node, int32_t* surf, double* t, double* P, double* n, uint32_t* flags,
                  uint32_t* count);

/* Jitter stream contract, exported so tests can pin host == device == oracle.
 * Writes the two 8-bit integers of sample s = i*n + j of the n x n grid (seed,pixel,node,light):
 * the reference's two np.random.random() draws of that sample (light.py:53-54) are u/256, v/256. */
void rt_jitter_bits(uint32_t seed, uint32_t pixel, uint32_t node, uint32_t light, uint32_t n, uint32_t s,
                    uint32_t* u8, uint32_t* v8);

#ifdef __cplusplus
}
#endif
#endif /* RT_AMD_H */
