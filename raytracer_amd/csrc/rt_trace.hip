// rt_trace.hip -- the float64 ray-tree kernel (gfx950).
//
// One launch per recursion depth.  Each lane owns one ray of the current queue:
// it finds the nearest hit by brute force over the surface list (staged in LDS,
// every lane of a wave reads the same 64-byte record => broadcast reads), turns the
// hit into (a) one record of the shading queue and (b) up to two child rays in the
// next queue, both appended with a wave-wide ballot / prefix count and ONE atomic
// per wave so that lanes stay dense, or adds throughput * background to the pixel
// when the ray leaves the scene or the child would exceed the recursion limit.
//
// Everything geometric here is float64 and follows the reference operation by
// operation (its image depends on the rounding of the hit point: shadow rays start
// exactly on the surface).  Rounding conventions being restated:
//   np.dot / np.linalg.norm (BLAS ddot)  -> fma(z,z', fma(y,y', x*x'))   [dotb]
//   np.sum(a*b, axis=1)                  -> (x*x' + y*y') + z*z'          [dots]
//   everything else                      -> one IEEE operation, as written
// so this file is compiled with floating-point contraction OFF and calls fma()
// only where the reference's BLAS does.
//
// Reference: camera.py:18-44 (camera ray), scene_settings.py:19-46 (nearest hit),
// surfaces/sphere.py:11-65, infinite_plane.py:12-44, cube.py:11-99 (scalar
// intersections), ray_tracer.py:242-294 (trace_ray), :174-239 (children),
// material.py:26-30 (blend, here as throughput weights).
#include "rt_device.h"
#include "../../include/rt_amd.h"
#include <hip/amd_detail/amd_hip_unsafe_atomics.h>

#pragma clang fp contract(off)

#define TRACE_BLOCK RT_TRACE_BLOCK
#define SURF_TILE RT_SURF_TILE

struct d3 { double x, y, z; };
__device__ __forceinline__ d3 D3(double x, double y, double z) { d3 r; r.x = x; r.y = y; r.z = z; return r; }
__device__ __forceinline__ d3 sub(d3 a, d3 b) { return D3(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ d3 add(d3 a, d3 b) { return D3(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ d3 scl(d3 a, double s) { return D3(a.x * s, a.y * s, a.z * s); }
__device__ __forceinline__ d3 divs(d3 a, double s) { return D3(a.x / s, a.y / s, a.z / s); }
__device__ __forceinline__ double dotb(d3 a, d3 b) { return fma(a.z, b.z, fma(a.y, b.y, a.x * b.x)); }
__device__ __forceinline__ double dots(d3 a, d3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
__device__ __forceinline__ d3 normalize(d3 v) {  // utils.py:9-22
    double n = sqrt(dotb(v, v));
    if (n == 0) return v;
    return divs(v, n);
}

// ---- scalar intersections: t only; the normal is derived for the winner ----------
__device__ __forceinline__ bool sphere_t(const SurfD& s, d3 o, d3 d, double& t) {  // sphere.py:33-59
    d3 oc = sub(o, D3(s.d[0], s.d[1], s.d[2]));
    double a = dotb(d, d);
    double b = 2.0 * dotb(d, oc);
    double c = dotb(oc, oc) - s.d[4];
    double disc = b * b - 4 * a * c;
    if (disc < 0) return false;
    double sq = sqrt(disc);
    double t1 = (-b - sq) / (2.0 * a);
    if (t1 > 0) { t = t1; return true; }
    double t2 = (-b + sq) / (2.0 * a);
    if (t2 > 0) { t = t2; return true; }
    return false;
}
__device__ __forceinline__ bool plane_t(const SurfD& s, d3 o, d3 d, double& t) {  // infinite_plane.py:31-44
    d3 n = D3(s.d[0], s.d[1], s.d[2]);
    double denom = dotb(d, n);
    if (fabs(denom) < RT_EPS) return false;
    double tt = (s.d[3] - dotb(o, n)) / denom;
    if (tt > 0) { t = tt; return true; }
    return false;
}
// cube.py:36-99.  ncode: 0 = zero normal, else 1 + axis*2 + (sign > 0)
__device__ __forceinline__ bool box_t(const SurfD& s, d3 o, d3 d, double& t, int& ncode) {
    double t_near = -INFINITY, t_far = INFINITY;
    int axis_n = 0, sign_pos = 1;
    const double oo[3] = {o.x, o.y, o.z}, dd[3] = {d.x, d.y, d.z};
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        double lo = s.d[a], hi = s.d[3 + a];
        if (fabs(dd[a]) < RT_EPS) {
            if (oo[a] < lo || oo[a] > hi) return false;
        } else {
            double t1 = (lo - oo[a]) / dd[a], t2 = (hi - oo[a]) / dd[a];
            if (t1 > t2) { double w = t1; t1 = t2; t2 = w; }
            if (t1 > t_near) { t_near = t1; axis_n = a; sign_pos = dd[a] > 0 ? 0 : 1; }
            if (t2 < t_far) t_far = t2;
            if (t_near > t_far) return false;
        }
    }
    if (t_far < 0) return false;
    if (t_near > 0) {
        t = t_near;
        ncode = 1 + axis_n * 2 + sign_pos;
    } else {
        t = t_far;
        ncode = 0;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            double h = oo[a] + t * dd[a];
            if (fabs(h - s.d[a]) < RT_EPS) { ncode = 1 + a * 2 + 0; break; }
            else if (fabs(h - s.d[3 + a]) < RT_EPS) { ncode = 1 + a * 2 + 1; break; }
        }
    }
    return true;
}

// dense append: returns the slot of this lane (valid only where pred); all 64 lanes must call it
__device__ __forceinline__ uint32_t wave_append(uint32_t* counter, bool pred) {
    unsigned long long m = __ballot(pred);
    uint32_t n = (uint32_t)__popcll(m);
    uint32_t prefix = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
    uint32_t base = 0;
    if ((threadIdx.x & 63) == 0 && n) base = atomicAdd(counter, n);
    base = __shfl(base, 0);
    return base + prefix;
}

__device__ __forceinline__ unsigned long long wave_sum(unsigned int v) {
    unsigned long long s = v;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o);
    return s;
}

// find_nearest_intersection (scene_settings.py:31-46) over tile[0..ns): file order, strict '<'
__device__ __forceinline__ void nearest_in_tile(const SurfD* tile, int ns, int s0, d3 o, d3 d, double& best, int& bs, int& bcode) {
    for (int j = 0; j < ns; ++j) {
        const SurfD& sf = tile[j];
        double t; int code = 0; bool hit;
        if (sf.type == RT_SURF_SPHERE) hit = sphere_t(sf, o, d, t);
        else if (sf.type == RT_SURF_PLANE) hit = plane_t(sf, o, d, t);
        else hit = box_t(sf, o, d, t, code);
        if (hit && t > RT_EPS && t < best) { best = t; bs = s0 + j; bcode = code; }
    }
}

// The same query through the uniform grid.  Candidates may arrive in any order and more than once, so
// the file-order rule is explicit: smaller t wins, equal t goes to the lower surface index -- exactly
// what the strict '<' of the ascending loop yields.  The walk stops once the best hit lies before the
// exit of the current cell: every surface that could be hit earlier, or at the same t, touches a cell
// already visited (cell lists are built from inflated boxes, which covers rounding at cell borders).
__device__ __forceinline__ void grid_candidate(const SurfD& sf, int j, d3 o, d3 d, double& best, int& bs, int& bcode) {
    double t; int code = 0; bool hit;
    if (sf.type == RT_SURF_SPHERE) hit = sphere_t(sf, o, d, t);
    else if (sf.type == RT_SURF_PLANE) hit = plane_t(sf, o, d, t);
    else hit = box_t(sf, o, d, t, code);
    if (hit && t > RT_EPS && (t < best || (t == best && j < bs))) { best = t; bs = j; bcode = code; }
}
__device__ __forceinline__ void nearest_in_grid(const SceneGrid& g, const SurfD* surf, d3 o, d3 d, double& best, int& bs, int& bcode) {
    for (int k = 0; k < g.n_global; ++k) { const int j = (int)g.global_items[k]; grid_candidate(surf[j], j, o, d, best, bs, bcode); }
    const double oo[3] = {o.x, o.y, o.z}, dd[3] = {d.x, d.y, d.z};
    double t0 = 0.0, t1 = INFINITY;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const double lo = g.lo[a], hi = g.lo[a] + g.cell[a] * (double)g.n[a];
        if (dd[a] == 0.0) { if (oo[a] < lo || oo[a] > hi) return; }
        else {
            double ta = (lo - oo[a]) / dd[a], tb = (hi - oo[a]) / dd[a];
            if (ta > tb) { const double w = ta; ta = tb; tb = w; }
            t0 = ta > t0 ? ta : t0; t1 = tb < t1 ? tb : t1;
        }
    }
    if (t0 > t1 || best < t0) return;
    // per-axis state in scalars (a dynamically indexed array would live in scratch memory)
    int ix, iy, iz, sx, sy, sz;
    double mx, my, mz, ex, ey, ez;
    auto setup = [&](int a, int& i, int& st, double& tm, double& td) {
        const double p = oo[a] + dd[a] * t0;
        i = (int)floor((p - g.lo[a]) * g.inv_cell[a]);
        i = i < 0 ? 0 : (i >= g.n[a] ? g.n[a] - 1 : i);
        st = dd[a] > 0.0 ? 1 : -1;
        if (dd[a] == 0.0) { tm = INFINITY; td = INFINITY; }
        else {
            tm = (g.lo[a] + g.cell[a] * (double)(i + (dd[a] > 0.0 ? 1 : 0)) - oo[a]) / dd[a];
            td = g.cell[a] / fabs(dd[a]);
        }
    };
    setup(0, ix, sx, mx, ex); setup(1, iy, sy, my, ey); setup(2, iz, sz, mz, ez);
    const int max_steps = g.n[0] + g.n[1] + g.n[2] + 3;
    // FP32 pre-test per candidate (the shadow walk of the shading kernels has the same one): the ray against the list
    // entry's bounding sphere, from a local origin where the ray enters the cell.  A sure miss costs a dozen FP32
    // operations instead of the dependent load of the 64-byte surface record and its float64 test.  Directions are unit
    // vectors up to rounding (camera rays are normalised, children reflect or continue them), which the margin
    // proportional to |c - o| covers.  The first three entries of a cell come with the cell's record (one round trip).
    const float wx = (float)d.x, wy = (float)d.y, wz = (float)d.z;
    double t_enter = t0;
    // The record of the NEXT cell is requested before the current cell's entries are tested (where the walk goes next
    // does not depend on them), so its round trip runs beside their tests.
    uint32_t c = (uint32_t)((iz * g.n[1] + iy) * g.n[0] + ix);
    uint4 hd, r1, r2, r3;
    { const uint4* rec = g.cell_rec + (size_t)c * 4; hd = rec[0]; r1 = rec[1]; r2 = rec[2]; r3 = rec[3]; }
    for (int it = 0; it < max_steps; ++it) {
        const double t_exit = fmin(mx, fmin(my, mz));
        const bool go_x = mx <= my && mx <= mz, go_y = !go_x && my <= mz, go_z = !go_x && !go_y;
        const int nx = ix + (go_x ? sx : 0), ny = iy + (go_y ? sy : 0), nz = iz + (go_z ? sz : 0);
        const bool next_in = nx >= 0 && nx < g.n[0] && ny >= 0 && ny < g.n[1] && nz >= 0 && nz < g.n[2];
        const uint32_t cn = next_in ? (uint32_t)((nz * g.n[1] + ny) * g.n[0] + nx) : c;
        uint4 hn, n1, n2, n3;
        { const uint4* rec = g.cell_rec + (size_t)cn * 4; hn = rec[0]; n1 = rec[1]; n2 = rec[2]; n3 = rec[3]; }
        const uint32_t n = hd.x;
        const float olx = (float)fma(d.x, t_enter, o.x), oly = (float)fma(d.y, t_enter, o.y), olz = (float)fma(d.z, t_enter, o.z);
        const float eps_cell = 3e-7f * (fabsf(olx) + fabsf(oly) + fabsf(olz)) + 1e-7f;
        uint32_t k0 = 0;
        if (n > 3) k0 = g.cell_start[c];
        for (uint32_t e = 0; e < n; ++e) {
            uint4 rb = e == 0 ? r1 : (e == 1 ? r2 : r3);
            uint32_t ju = e == 0 ? hd.y : (e == 1 ? hd.z : hd.w);
            if (e > 2) { rb = ((const uint4*)g.item_bounds)[k0 + e]; ju = g.items[k0 + e]; }
            const float4 bnd = make_float4(__uint_as_float(rb.x), __uint_as_float(rb.y), __uint_as_float(rb.z), __uint_as_float(rb.w));
            const float cx = bnd.x - olx, cy = bnd.y - oly, cz = bnd.z - olz;
            const float bb = fmaf(cz, wz, fmaf(cy, wy, cx * wx));
            const float qx = fmaf(-wx, bb, cx), qy = fmaf(-wy, bb, cy), qz = fmaf(-wz, bb, cz);
            const float rm = bnd.w + eps_cell + 2e-6f * (fabsf(cx) + fabsf(cy) + fabsf(cz));
            if (fmaf(qz, qz, fmaf(qy, qy, qx * qx)) > rm * rm * 1.0001f) continue;   // (NaN: not skipped)
            const int j = (int)ju;
            grid_candidate(surf[j], j, o, d, best, bs, bcode);
        }
        if (best < t_exit - 1e-9 * (1.0 + fabs(t_exit))) return;
        if (!next_in) return;
        t_enter = t_exit;
        ix = nx; iy = ny; iz = nz;
        mx += go_x ? ex : 0.0; my += go_y ? ey : 0.0; mz += go_z ? ez : 0.0;
        c = cn; hd = hn; r1 = n1; r2 = n2; r3 = n3;
    }
}

// hit point, normal and own-surface flags of the winning surface
struct HitFrame { d3 P, n; uint32_t flags; int mat; };
__device__ __forceinline__ HitFrame hit_frame(const SurfD& sf, d3 o, d3 d, double best, int bcode) {
    HitFrame f;
    f.n = D3(0, 0, 0);
    f.flags = 0;
    f.mat = sf.mat;
    f.P = add(o, scl(d, best));  // scene_settings.py:45
    const d3 P = f.P;
    if (sf.type == RT_SURF_SPHERE) {
        d3 c = D3(sf.d[0], sf.d[1], sf.d[2]);
        d3 oc = sub(P, c);
        f.n = divs(oc, sf.d[3]);                                   // sphere.py:62-63
        if (dots(oc, oc) - sf.d[5] > 0) f.flags |= HF_CBATCH_POS;  // sphere.py:93
        if (dotb(oc, oc) - sf.d[4] > 0) f.flags |= HF_CDOT_POS;    // sphere.py:38
        if (f.flags) f.flags |= HF_EXACT;
    } else if (sf.type == RT_SURF_PLANE) {
        f.n = D3(sf.d[0], sf.d[1], sf.d[2]);
    } else {
        if (bcode == 0) f.flags |= HF_NO_OWN;
        else {
            int ax = (bcode - 1) >> 1; bool pos = (bcode - 1) & 1;
            double sg = pos ? 1.0 : -1.0;
            f.n = D3(ax == 0 ? sg : 0.0, ax == 1 ? sg : 0.0, ax == 2 ? sg : 0.0);
            double pa = ax == 0 ? P.x : (ax == 1 ? P.y : P.z);
            double face = pos ? (ax == 0 ? sf.d[3] : (ax == 1 ? sf.d[4] : sf.d[5]))
                              : (ax == 0 ? sf.d[0] : (ax == 1 ? sf.d[1] : sf.d[2]));
            if (pos ? (pa > face) : (pa < face)) f.flags |= HF_BOX_OUTSIDE;
        }
    }
    return f;
}

__device__ __forceinline__ void store_shade(const ShadeQueue& q, uint32_t slot, d3 d, const HitFrame& f, float tr, float tg, float tb,
                                            float transp, uint32_t lpix, uint32_t gpix, uint32_t node, int bs) {
    // view = normalize(origin - hit) (ray_tracer.py:275) = -d up to a few float64 ulps, because every ray direction
    // is unit length by construction (normalised camera ray, reflection of a unit vector, unchanged direction);
    // it is only consumed in FP32, where the two agree, so the float64 sqrt and three divides are not spent on it
    d3 view = D3(-d.x, -d.y, -d.z);
    float om = (float)(1.0 - (double)transp);   // material.py:28
    q.r0[slot] = make_float4((float)f.P.x, (float)f.P.y, (float)f.P.z, (float)f.n.x);
    q.r1[slot] = make_float4((float)f.n.y, (float)f.n.z, (float)view.x, (float)view.y);
    q.r2[slot] = make_float4((float)view.z, tr * om, tg * om, tb * om);
    q.r3[slot] = make_uint4(lpix, gpix, node, (uint32_t)bs | (f.flags << 24));
    if (f.flags & HF_EXACT) { q.p64[(size_t)slot * 3 + 0] = f.P.x; q.p64[(size_t)slot * 3 + 1] = f.P.y; q.p64[(size_t)slot * 3 + 2] = f.P.z; }
}

__device__ __forceinline__ void store_debug(const HitDebug& g, uint32_t slot, uint32_t gpix, uint32_t node, int bs, bool hit, double best,
                                            const HitFrame& f) {
    g.gpixel[slot] = gpix; g.node[slot] = node; g.surf[slot] = bs;
    g.t[slot] = hit ? best : 0.0; g.flags[slot] = f.flags;
    g.P[slot * 3 + 0] = f.P.x; g.P[slot * 3 + 1] = f.P.y; g.P[slot * 3 + 2] = f.P.z;
    g.n[slot * 3 + 0] = f.n.x; g.n[slot * 3 + 1] = f.n.y; g.n[slot * 3 + 2] = f.n.z;
}

// throughput * background (ray_tracer.py:261-262, 269-270)
__device__ __forceinline__ void add_background(const TraceArgs& a, uint32_t lpix, float r, float g, float b) {
    unsafeAtomicAdd(&a.accum[(size_t)0 * a.npix + lpix], (double)(r * a.fp.bg[0]));
    unsafeAtomicAdd(&a.accum[(size_t)1 * a.npix + lpix], (double)(g * a.fp.bg[1]));
    unsafeAtomicAdd(&a.accum[(size_t)2 * a.npix + lpix], (double)(b * a.fp.bg[2]));
}

// Camera.get_ray (camera.py:31-44) for band pixel idx: global pixel index, origin, direction
__device__ __forceinline__ void camera_ray(const TraceArgs& a, uint32_t idx, uint32_t& gpix, d3& o, d3& d) {
    const int W = a.fp.width;
    int r = (int)(idx / (uint32_t)W), px = (int)(idx % (uint32_t)W);
    int py = a.fp.row_y[r];
    gpix = (uint32_t)py * (uint32_t)W + (uint32_t)px;
    const CameraD& c = a.cam;
    double sh = c.swidth * ((double)a.fp.height / (double)W);
    double nx = -((px + 0.5) / W - 0.5);
    double ny = (py + 0.5) / a.fp.height - 0.5;
    d3 cp = D3(c.pos[0], c.pos[1], c.pos[2]);
    d3 centre = add(cp, scl(D3(c.fwd[0], c.fwd[1], c.fwd[2]), c.sdist));
    d3 pt = add(add(centre, scl(D3(c.right[0], c.right[1], c.right[2]), nx * c.swidth)),
                scl(D3(c.up[0], c.up[1], c.up[2]), -ny * sh));
    d = normalize(sub(pt, cp));
    o = cp;
}

// Camera rays are enumerated in 8 x 8 pixel tiles, one tile per wave: the 64 hits of a wave (and of
// the 64-hit chunks the shading kernel works on) then lie close together on the surfaces, so their
// per-light classification and undecided sample tiles agree far more often than for a 64 x 1 strip.
// Slots of tiles that overhang the band's right or bottom edge stay idle.
__device__ __forceinline__ uint32_t primary_slots(const TraceArgs& a) {
    return (((uint32_t)a.fp.width + 7u) >> 3) * (((uint32_t)a.fp.band_rows + 7u) >> 3) * 64u;
}
__device__ __forceinline__ bool primary_pixel(const TraceArgs& a, uint32_t idx, uint32_t& lpix) {
    const uint32_t tiles_x = ((uint32_t)a.fp.width + 7u) >> 3;
    const uint32_t t = idx >> 6, within = idx & 63u;
    const uint32_t ty = t / tiles_x, tx = t - ty * tiles_x;
    const uint32_t r = ty * 8u + (within >> 3), px = tx * 8u + (within & 7u);
    lpix = r * (uint32_t)a.fp.width + px;
    return r < (uint32_t)a.fp.band_rows && px < (uint32_t)a.fp.width;
}

template <bool PRIMARY, bool GRID>
__global__ __launch_bounds__(TRACE_BLOCK) void k_trace(TraceArgs a) {
    __shared__ SurfD tile[SURF_TILE];
    __shared__ uint32_t s_cnt[3][TRACE_BLOCK / 64];
    __shared__ uint32_t s_base[3];
    const int S = a.fp.n_surfaces;
    const int W = a.fp.width;
    uint32_t count;
    if (PRIMARY) count = primary_slots(a);
    else { count = *a.in.count; if (count > a.in.capacity) count = a.in.capacity; }

    constexpr bool use_grid = GRID;  // lanes walk their own cells: surfaces come from global memory (L2)
    const bool single_tile = S <= SURF_TILE && !use_grid;
    if (single_tile) {
        // the whole surface list is staged once per workgroup
        const double* src = (const double*)a.surf;
        double* dst = (double*)tile;
        for (int i = threadIdx.x; i < S * 8; i += TRACE_BLOCK) dst[i] = src[i];
        __syncthreads();
    }
    const bool child_cut = (a.depth + 1) >= a.fp.max_rec;
    unsigned int n_traced = 0, n_cut = 0, n_shaded = 0, ovf_ray = 0, ovf_shade = 0;

    const uint32_t stride = gridDim.x * TRACE_BLOCK;
    const uint32_t rounded = (count + TRACE_BLOCK - 1) / TRACE_BLOCK * TRACE_BLOCK;
    for (uint32_t base = blockIdx.x * TRACE_BLOCK; base < rounded; base += stride) {
        const uint32_t idx = base + threadIdx.x;
        bool active = idx < count;
        d3 o = D3(0, 0, 0), d = D3(0, 0, 1);
        float tr = 1.f, tg = 1.f, tb = 1.f;
        uint32_t lpix = 0, node = 1, gpix = 0;
        if (PRIMARY) active = active && primary_pixel(a, idx, lpix);
        if (active) {
            if (PRIMARY) {
                camera_ray(a, lpix, gpix, o, d);
            } else {
                const double2 qa = a.in.oa[idx], qb = a.in.ob[idx], qc = a.in.oc[idx];
                const uint4 qt = a.in.tp[idx];
                o = D3(qa.x, qa.y, qb.x);
                d = D3(qb.y, qc.x, qc.y);
                tr = __uint_as_float(qt.x); tg = __uint_as_float(qt.y); tb = __uint_as_float(qt.z);
                lpix = qt.w; node = a.in.node[idx];
                int r = (int)(lpix / (uint32_t)W), px = (int)(lpix % (uint32_t)W);
                gpix = (uint32_t)a.fp.row_y[r] * (uint32_t)W + (uint32_t)px;
            }
            n_traced += 1;
        }
        if (PRIMARY && a.fp.max_rec <= 0) {
            // depth 0 >= max_recursions: every pixel is the background, nothing is traced (ray_tracer.py:261)
            if (active) {
                n_traced -= 1; n_cut += 1;
                a.accum[(size_t)0 * a.npix + lpix] = a.fp.bg[0]; a.accum[(size_t)1 * a.npix + lpix] = a.fp.bg[1]; a.accum[(size_t)2 * a.npix + lpix] = a.fp.bg[2];
            }
            if (idx < count) a.shade.r3[idx] = make_uint4(0u, 0u, 0u, RT_NULL_HIT);
            continue;
        }

        // ---- find_nearest_intersection (scene_settings.py:31-46): file order, strict '<' ----
        double best = INFINITY;
        int bs = -1, bcode = 0;
        if (use_grid && active) nearest_in_grid(a.grid, a.surf, o, d, best, bs, bcode);
        for (int s0 = 0; s0 < S && !use_grid; s0 += SURF_TILE) {
            int ns = S - s0 < SURF_TILE ? S - s0 : SURF_TILE;
            if (!single_tile) {
                __syncthreads();
                const double* src = (const double*)(a.surf + s0);
                double* dst = (double*)tile;
                for (int i = threadIdx.x; i < ns * 8; i += TRACE_BLOCK) dst[i] = src[i];
                __syncthreads();
            }
            if (active) nearest_in_tile(tile, ns, s0, o, d, best, bs, bcode);
        }

        // ---- hit point, normal, own-surface flags --------------------------------------
        const bool hit = active && bs >= 0;
        HitFrame f;
        f.P = D3(0, 0, 0); f.n = D3(0, 0, 0); f.flags = 0; f.mat = 0;
        if (hit) { f = hit_frame(a.surf[bs], o, d, best, bcode); if (a.fp.all_p64) f.flags |= HF_EXACT; }
        const d3 P = f.P, n = f.n;
        const MatF m = a.mat[f.mat];

        if (a.dbg.capacity) {
            uint32_t slot = wave_append(a.dbg.count, active);
            if (active && slot < a.dbg.capacity) store_debug(a.dbg, slot, gpix, node, bs, hit, best, f);
        }

        // ---- ray left the scene: throughput * background (ray_tracer.py:269-270) ----------
        if (active && !hit) add_background(a, lpix, tr, tg, tb);

        // ---- reserve queue slots: ONE atomic per workgroup and queue ------------------------------
        // (a single counter word sustains only ~90 atomics/us: per-wave reservations of three
        //  queues would serialise the whole kernel on those three words)
        const bool want_r = hit && m.has_refl;
        const bool want_t = hit && m.transp > 0.f;
        // camera rays own the shade slot of their enumeration index: nothing to reserve
        const bool push_s = hit && !PRIMARY, push_r = want_r && !child_cut, push_t = want_t && !child_cut;
        const unsigned long long bs_ = __ballot(push_s), br_ = __ballot(push_r), bt_ = __ballot(push_t);
        const uint32_t lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
        if (lane == 0) { s_cnt[0][wv] = (uint32_t)__popcll(bs_); s_cnt[1][wv] = (uint32_t)__popcll(br_); s_cnt[2][wv] = (uint32_t)__popcll(bt_); }
        __syncthreads();
        if (threadIdx.x < 3) {
            const int q = threadIdx.x;
            uint32_t total = 0;
#pragma unroll
            for (int w = 0; w < TRACE_BLOCK / 64; ++w) { const uint32_t c = s_cnt[q][w]; s_cnt[q][w] = total; total += c; }
            uint32_t* counter = q == 0 ? a.shade.count : a.out.count;
            // reflection and transparency children share the output queue: two disjoint reservations
            s_base[q] = total ? atomicAdd(counter, total) : 0u;
        }
        __syncthreads();
        const uint32_t slot_s = a.shade.base + s_base[0] + s_cnt[0][wv] + __builtin_amdgcn_mbcnt_hi((uint32_t)(bs_ >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bs_, 0u));
        const uint32_t slot_r = s_base[1] + s_cnt[1][wv] + __builtin_amdgcn_mbcnt_hi((uint32_t)(br_ >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)br_, 0u));
        const uint32_t slot_t = s_base[2] + s_cnt[2][wv] + __builtin_amdgcn_mbcnt_hi((uint32_t)(bt_ >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bt_, 0u));
        __syncthreads();  // s_cnt / s_base are rewritten by the next iteration

        // ---- shading record -----------------------------------------------------------
        if (PRIMARY) {
            if (idx < count) {  // every slot of the camera rays is written: a record or the null mark
                if (hit) { store_shade(a.shade, idx, d, f, tr, tg, tb, m.transp, lpix, gpix, node, bs); n_shaded += 1; }
                else a.shade.r3[idx] = make_uint4(0u, 0u, 0u, RT_NULL_HIT);
            }
        } else if (push_s) {
            if (slot_s < a.shade.capacity) {
                store_shade(a.shade, slot_s, d, f, tr, tg, tb, m.transp, lpix, gpix, node, bs);
                n_shaded += 1;
            } else ovf_shade += 1;
        }

        // ---- reflection child (ray_tracer.py:192-207) ---------------------------------------
        {
            const float cr = tr * m.kr[0], cg = tg * m.kr[1], cb = tb * m.kr[2];
            if (want_r && child_cut) {  // depth limit: the child returns the background untraced (:261-262)
                add_background(a, lpix, cr, cg, cb);
                n_cut += 1;
            }
            if (push_r) {
                if (slot_r < a.out.capacity) {
                    const uint32_t slot = slot_r;
                    d3 rd = sub(d, scl(n, 2 * dotb(d, n)));  // ray_tracer.py:126
                    d3 ro = add(P, scl(n, RT_EPS));          // ray_tracer.py:198
                    a.out.oa[slot] = make_double2(ro.x, ro.y); a.out.ob[slot] = make_double2(ro.z, rd.x);
                    a.out.oc[slot] = make_double2(rd.y, rd.z);
                    a.out.tp[slot] = make_uint4(__float_as_uint(cr), __float_as_uint(cg), __float_as_uint(cb), lpix);
                    a.out.node[slot] = node * 2u;
                } else ovf_ray += 1;
            }
        }
        // ---- transparency child (ray_tracer.py:226-239) -------------------------------------
        {
            const float cr = tr * m.transp, cg = tg * m.transp, cb = tb * m.transp;
            if (want_t && child_cut) {
                add_background(a, lpix, cr, cg, cb);
                n_cut += 1;
            }
            if (push_t) {
                if (slot_t < a.out.capacity) {
                    const uint32_t slot = slot_t;
                    d3 to = add(P, scl(d, RT_EPS));  // ray_tracer.py:230
                    a.out.oa[slot] = make_double2(to.x, to.y); a.out.ob[slot] = make_double2(to.z, d.x);
                    a.out.oc[slot] = make_double2(d.y, d.z);
                    a.out.tp[slot] = make_uint4(__float_as_uint(cr), __float_as_uint(cg), __float_as_uint(cb), lpix);
                    a.out.node[slot] = node * 2u + 1u;
                } else ovf_ray += 1;
            }
        }
    }

    // ---- counters: one atomic per wave and counter -------------------------------------------
    unsigned long long s_tr = wave_sum(n_traced), s_cut = wave_sum(n_cut), s_sh = wave_sum(n_shaded);
    unsigned long long s_or = wave_sum(ovf_ray), s_os = wave_sum(ovf_shade);
    if ((threadIdx.x & 63) == 0) {
        if (s_tr) atomicAdd(rt_counter(a.counters, PRIMARY ? CNT_PRIMARY : CNT_SECONDARY), s_tr);
        if (s_cut) atomicAdd(rt_counter(a.counters, CNT_DEPTH_CUT), s_cut);
        if (s_sh) atomicAdd(rt_counter(a.counters, CNT_SHADED), s_sh);
        if (s_or) atomicAdd(rt_counter(a.counters, CNT_OVERFLOW_RAY), s_or);
        if (s_os) atomicAdd(rt_counter(a.counters, CNT_OVERFLOW_SHADE), s_os);
    }
}

// ---------------------------------------------------------------------------------------
// Tail of the ray tree: every lane takes one ray of the input queue and follows it to the
// recursion limit inside this launch.  Deep levels hold few rays, so one launch per level is
// bound by the latency of a launch and of one dependent float64 chain; here the levels of one
// lane follow each other without queue round trips.  A hit with two children continues with
// the reflection child and parks the transparency child on the lane's stack (one slot per
// level, laid out [level][thread] so a wave's accesses stay contiguous).  All rays of a lane's
// subtree belong to one pixel.  The workgroup steps through the levels together and hands out
// shade slots from a block it reserved with one atomic (reservations per wave and step would
// serialise on the counter word).  Needs the whole surface list in one LDS tile (512-thread
// workgroups), or the grid (single-wave workgroups: every wave takes its tiles alone).
// ---------------------------------------------------------------------------------------
template <bool PRIMARY, bool GRID>
__global__ __launch_bounds__(GRID ? RT_TAIL_BLOCK_GRID : RT_TAIL_BLOCK, GRID ? RT_TAIL_WAVES_GRID : RT_TAIL_WAVES) void k_trace_tail(TraceArgs a) {
    constexpr int TAIL_BLOCK = GRID ? RT_TAIL_BLOCK_GRID : RT_TAIL_BLOCK;
    __shared__ SurfD tile[SURF_TILE];
    __shared__ uint32_t s_cnt[TAIL_BLOCK / 64];
    __shared__ uint32_t s_base;
    // Slots for appended hits are reserved a.tail_reserve at a time and handed out from LDS: returning atomics on one word
    // complete at ~90 M/s on this chip whoever issues them, and one per workgroup and step is more than that allows.
    // [s_free, s_end) is what the workgroup still holds; what a reservation leaves over, and the rest at the end, is
    // marked as holding no hit (the shading kernels skip such slots like the camera rays that left the scene).
    __shared__ uint32_t s_free, s_end, s_hole0, s_hole1;
    if (threadIdx.x == 0) { s_free = 0; s_end = 0; s_hole0 = 0; s_hole1 = 0; }
    __syncthreads();
    const int S = a.fp.n_surfaces, W = a.fp.width;
    uint32_t count;
    if (PRIMARY) count = primary_slots(a);
    else { count = *a.in.count; if (count > a.in.capacity) count = a.in.capacity; }
    constexpr bool use_grid = GRID;
    if (!use_grid) {
        const double* src = (const double*)a.surf;
        double* dst = (double*)tile;
        for (int i = threadIdx.x; i < S * 8; i += TAIL_BLOCK) dst[i] = src[i];
        __syncthreads();
    }
    unsigned int n_primary = 0, n_traced = 0, n_cut = 0, n_shaded = 0, ovf_ray = 0, ovf_shade = 0;
    const uint32_t n_threads = gridDim.x * TAIL_BLOCK;
    const uint32_t tid = blockIdx.x * TAIL_BLOCK + threadIdx.x;
    const uint32_t lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    // Work is handed out per WAVE in tiles of 64 rays (for camera rays: one 8 x 8 pixel tile): wave wv of workgroup b
    // takes tiles b*8 + wv, + gridDim*8, ...  In grid scenes a wave whose 64 ray trees are finished takes its next
    // tile at once (config #4, 11 001 surfaces: 1.76 -> 1.25 ms); only the steps are taken together (the barrier
    // serves the shared slot reservation below).  In small scenes the whole workgroup finishes its eight tiles before
    // it takes the next eight: letting waves (or single lanes) run ahead was measured -- the ray trees get up to 13 %
    // faster, but the appended hits then arrive in an order that costs the register-resident shading kernel more.
    // There, the eights after the first come from a shared counter (ray trees differ in length by an order of magnitude
    // between regions of a frame, so equal shares of tiles are unequal shares of work).  The counter is read one refill
    // ahead: the atomic's round trip overlaps the ray trees of the current tiles.
    constexpr uint32_t WAVES = TAIL_BLOCK / 64;
    __shared__ uint32_t s_batch;
    uint32_t next_tile = blockIdx.x * WAVES + wv;
    // (grid scenes: a wave takes its tiles alone, one atomic per tile would be too many; frames with many eights per
    // workgroup are balanced well enough by equal shares, and the launch then passes no counter)
    const bool DYNAMIC = !GRID && a.work != nullptr;
    uint32_t ahead = 0;        // thread 0: counter value fetched for the workgroup's next refill
    bool exhausted = false;    // DYNAMIC: the last refill found the tiles used up
    uint32_t idx = 0;
    bool have = false;
    d3 o = D3(0, 0, 0), d = D3(0, 0, 1);
    float tr = 1.f, tg = 1.f, tb = 1.f;
    uint32_t lpix = 0, node = 1, gpix = 0;
    int depth = a.depth;
    uint32_t sp = 0;      // parked children of this lane
    bool first = false;   // first step of a camera ray: its shade slot is its enumeration index
    {
        auto refill = [&]() {
            idx = next_tile * 64u + lane;
            if (DYNAMIC) {   // (the caller sets next_tile from the counter value fetched here)
                exhausted = (uint64_t)(next_tile - wv) * 64u >= count;
                if (threadIdx.x == 0) ahead = atomicAdd(a.work, 1u);
            } else next_tile += gridDim.x * WAVES;
            have = idx < count;
            tr = 1.f; tg = 1.f; tb = 1.f;
            lpix = 0; node = 1; gpix = 0;
            depth = a.depth;
            sp = 0;
            if (PRIMARY) have = have && primary_pixel(a, idx, lpix);
            if (have && PRIMARY) {
                camera_ray(a, lpix, gpix, o, d);
            } else if (have) {
                const double2 qa = a.in.oa[idx], qb = a.in.ob[idx], qc = a.in.oc[idx];
                const uint4 qt = a.in.tp[idx];
                o = D3(qa.x, qa.y, qb.x);
                d = D3(qb.y, qc.x, qc.y);
                tr = __uint_as_float(qt.x); tg = __uint_as_float(qt.y); tb = __uint_as_float(qt.z);
                lpix = qt.w; node = a.in.node[idx];
                int r = (int)(lpix / (uint32_t)W), px = (int)(lpix % (uint32_t)W);
                gpix = (uint32_t)a.fp.row_y[r] * (uint32_t)W + (uint32_t)px;
            }
            first = PRIMARY;
            if (PRIMARY && !have && idx < count) a.shade.r3[idx] = make_uint4(0u, 0u, 0u, RT_NULL_HIT);  // slot outside the band
        };
        if (DYNAMIC) refill();   // the workgroup's first eight tiles are its own
        for (;;) {
            // (a tile may hold no ray at all: pixels outside the band)
            if (GRID) {
                while (!__any(have) && (uint64_t)next_tile * 64u < count) refill();
            }
            if (!__syncthreads_or(have)) {
                // small scenes: the workgroup moves on as a whole (all its waves advance together, so the test is uniform)
                if (GRID) break;
                if (DYNAMIC) {
                    if (exhausted) break;
                    if (threadIdx.x == 0) s_batch = gridDim.x + ahead;
                    __syncthreads();
                    next_tile = s_batch * WAVES + wv;
                } else if ((uint64_t)(next_tile - wv) * 64u >= count) break;
                refill();
            } else {
            double best = INFINITY;
            int bs = -1, bcode = 0;
            if (have) {
                if (PRIMARY && depth == 0) n_primary += 1; else n_traced += 1;
                if (use_grid) nearest_in_grid(a.grid, a.surf, o, d, best, bs, bcode);
                else nearest_in_tile(tile, S, 0, o, d, best, bs, bcode);
            }
            const bool hit = have && bs >= 0;
            HitFrame f;
            f.P = D3(0, 0, 0); f.n = D3(0, 0, 0); f.flags = 0; f.mat = 0;
            if (hit) { f = hit_frame(use_grid ? a.surf[bs] : tile[bs], o, d, best, bcode); if (a.fp.all_p64) f.flags |= HF_EXACT; }
            const MatF m = a.mat[f.mat];
            if (a.dbg.capacity) {
                uint32_t slot = wave_append(a.dbg.count, have);
                if (have && slot < a.dbg.capacity) store_debug(a.dbg, slot, gpix, node, bs, hit, best, f);
            }
            if (have && !hit) add_background(a, lpix, tr, tg, tb);
            const bool push_s = hit && !first;
            if (first && have) {
                if (hit) { store_shade(a.shade, idx, d, f, tr, tg, tb, m.transp, lpix, gpix, node, bs); n_shaded += 1; }
                else a.shade.r3[idx] = make_uint4(0u, 0u, 0u, RT_NULL_HIT);
            }
            first = false;
            const unsigned long long bs_ = __ballot(push_s);
            if (lane == 0) s_cnt[wv] = (uint32_t)__popcll(bs_);
            __syncthreads();
            if (threadIdx.x == 0) {
                uint32_t total = 0;
#pragma unroll
                for (int w = 0; w < TAIL_BLOCK / 64; ++w) { const uint32_t c = s_cnt[w]; s_cnt[w] = total; total += c; }
                uint32_t fr = s_free, en = s_end;
                s_hole0 = fr; s_hole1 = fr;
                if (total > en - fr) {
                    s_hole1 = en;   // too small for this step: given up
                    const uint32_t want = total > a.tail_reserve ? ((total + 63u) & ~63u) : a.tail_reserve;
                    fr = atomicAdd(a.shade.count, want);
                    en = fr + want;
                }
                s_base = fr; s_free = fr + total; s_end = en;
            }
            __syncthreads();
            for (uint32_t hs = s_hole0 + threadIdx.x; hs < s_hole1; hs += TAIL_BLOCK)
                if (a.shade.base + hs < a.shade.capacity) a.shade.r3[a.shade.base + hs] = make_uint4(0u, 0u, 0u, RT_NULL_HIT);
            const uint32_t slot_s = a.shade.base + s_base + s_cnt[wv] + __builtin_amdgcn_mbcnt_hi((uint32_t)(bs_ >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bs_, 0u));
            if (push_s) {
                if (slot_s < a.shade.capacity) {
                    store_shade(a.shade, slot_s, d, f, tr, tg, tb, m.transp, lpix, gpix, node, bs);
                    n_shaded += 1;
                } else ovf_shade += 1;
            }
            const bool cut = (depth + 1) >= a.fp.max_rec;
            const bool want_r = hit && m.has_refl, want_t = hit && m.transp > 0.f;
            if (want_r && cut) { add_background(a, lpix, tr * m.kr[0], tg * m.kr[1], tb * m.kr[2]); n_cut += 1; }
            if (want_t && cut) { add_background(a, lpix, tr * m.transp, tg * m.transp, tb * m.transp); n_cut += 1; }
            const bool go_r = want_r && !cut, go_t = want_t && !cut;
            if (go_r && go_t) {
                // park the transparency child (ray_tracer.py:226-239)
                if (sp < a.stack_levels) {
                    const size_t slot = (size_t)sp * n_threads + tid;
                    d3 to = add(f.P, scl(d, RT_EPS));
                    a.stack.oa[slot] = make_double2(to.x, to.y); a.stack.ob[slot] = make_double2(to.z, d.x);
                    a.stack.oc[slot] = make_double2(d.y, d.z);
                    a.stack.tp[slot] = make_uint4(__float_as_uint(tr * m.transp), __float_as_uint(tg * m.transp),
                                                  __float_as_uint(tb * m.transp), (uint32_t)(depth + 1));
                    a.stack.node[slot] = node * 2u + 1u;
                    sp += 1;
                } else ovf_ray += 1;
            }
            if (go_r) {  // reflection child (ray_tracer.py:192-207)
                d3 rd = sub(d, scl(f.n, 2 * dotb(d, f.n)));  // ray_tracer.py:126
                o = add(f.P, scl(f.n, RT_EPS));              // ray_tracer.py:198
                d = rd;
                tr *= m.kr[0]; tg *= m.kr[1]; tb *= m.kr[2];
                node = node * 2u; depth += 1;
            } else if (go_t) {  // transparency child, same direction (ray_tracer.py:230)
                o = add(f.P, scl(d, RT_EPS));
                tr *= m.transp; tg *= m.transp; tb *= m.transp;
                node = node * 2u + 1u; depth += 1;
            } else if (have && sp > 0) {
                sp -= 1;
                const size_t slot = (size_t)sp * n_threads + tid;
                const double2 qa = a.stack.oa[slot], qb = a.stack.ob[slot], qc = a.stack.oc[slot];
                const uint4 qt = a.stack.tp[slot];
                o = D3(qa.x, qa.y, qb.x);
                d = D3(qb.y, qc.x, qc.y);
                tr = __uint_as_float(qt.x); tg = __uint_as_float(qt.y); tb = __uint_as_float(qt.z);
                depth = (int)qt.w; node = a.stack.node[slot];
            } else have = false;
            }
        }
    }
    for (uint32_t hs = s_free + threadIdx.x; hs < s_end; hs += TAIL_BLOCK)   // (the loop is left by the whole workgroup, after a barrier)
        if (a.shade.base + hs < a.shade.capacity) a.shade.r3[a.shade.base + hs] = make_uint4(0u, 0u, 0u, RT_NULL_HIT);
    unsigned long long s_tr = wave_sum(n_traced), s_cut = wave_sum(n_cut), s_sh = wave_sum(n_shaded);
    unsigned long long s_or = wave_sum(ovf_ray), s_os = wave_sum(ovf_shade);
    unsigned long long s_pr = wave_sum(n_primary);
    if (lane == 0) {
        if (s_pr) atomicAdd(rt_counter(a.counters, CNT_PRIMARY), s_pr);
        if (s_tr) atomicAdd(rt_counter(a.counters, CNT_SECONDARY), s_tr);
        if (s_cut) atomicAdd(rt_counter(a.counters, CNT_DEPTH_CUT), s_cut);
        if (s_sh) atomicAdd(rt_counter(a.counters, CNT_SHADED), s_sh);
        if (s_or) atomicAdd(rt_counter(a.counters, CNT_OVERFLOW_RAY), s_or);
        if (s_os) atomicAdd(rt_counter(a.counters, CNT_OVERFLOW_SHADE), s_os);
    }
}

void rt_launch_trace_tail(const TraceArgs& a, int blocks, hipStream_t s) {
    const dim3 g(blocks), b(a.grid.cell_start ? RT_TAIL_BLOCK_GRID : RT_TAIL_BLOCK);
    if (a.grid.cell_start) {
        if (a.depth == 0) hipLaunchKernelGGL((k_trace_tail<true, true>), g, b, 0, s, a);
        else hipLaunchKernelGGL((k_trace_tail<false, true>), g, b, 0, s, a);
    } else {
        if (a.depth == 0) hipLaunchKernelGGL((k_trace_tail<true, false>), g, b, 0, s, a);
        else hipLaunchKernelGGL((k_trace_tail<false, false>), g, b, 0, s, a);
    }
}

void rt_launch_trace(const TraceArgs& a, int blocks, hipStream_t s) {
    const dim3 g(blocks), b(TRACE_BLOCK);
    if (a.grid.cell_start) {
        if (a.depth == 0) hipLaunchKernelGGL((k_trace<true, true>), g, b, 0, s, a);
        else hipLaunchKernelGGL((k_trace<false, true>), g, b, 0, s, a);
    } else {
        if (a.depth == 0) hipLaunchKernelGGL((k_trace<true, false>), g, b, 0, s, a);
        else hipLaunchKernelGGL((k_trace<false, false>), g, b, 0, s, a);
    }
}
