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
#define SURF_TILE 512  // 512 x 64 B = 32 KiB of LDS per workgroup

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

template <bool PRIMARY>
__global__ __launch_bounds__(TRACE_BLOCK) void k_trace(TraceArgs a) {
    __shared__ SurfD tile[SURF_TILE];
    __shared__ uint32_t s_cnt[3][TRACE_BLOCK / 64];
    __shared__ uint32_t s_base[3];
    const int S = a.fp.n_surfaces;
    const int W = a.fp.width;
    uint32_t count;
    if (PRIMARY) count = (uint32_t)a.fp.band_rows * (uint32_t)W;
    else { count = *a.in.count; if (count > a.in.capacity) count = a.in.capacity; }

    const bool single_tile = S <= SURF_TILE;
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
        const bool active = idx < count;
        d3 o = D3(0, 0, 0), d = D3(0, 0, 1);
        float tr = 1.f, tg = 1.f, tb = 1.f;
        uint32_t lpix = 0, node = 1, gpix = 0;
        if (active) {
            if (PRIMARY) {
                // Camera.get_ray (camera.py:31-44)
                lpix = idx;
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
            continue;
        }

        // ---- find_nearest_intersection (scene_settings.py:31-46): file order, strict '<' ----
        double best = INFINITY;
        int bs = -1, bcode = 0;
        for (int s0 = 0; s0 < S; s0 += SURF_TILE) {
            int ns = S - s0 < SURF_TILE ? S - s0 : SURF_TILE;
            if (!single_tile) {
                __syncthreads();
                const double* src = (const double*)(a.surf + s0);
                double* dst = (double*)tile;
                for (int i = threadIdx.x; i < ns * 8; i += TRACE_BLOCK) dst[i] = src[i];
                __syncthreads();
            }
            if (active) {
                for (int j = 0; j < ns; ++j) {
                    const SurfD& sf = tile[j];
                    double t; int code = 0; bool hit;
                    if (sf.type == RT_SURF_SPHERE) hit = sphere_t(sf, o, d, t);
                    else if (sf.type == RT_SURF_PLANE) hit = plane_t(sf, o, d, t);
                    else hit = box_t(sf, o, d, t, code);
                    if (hit && t > RT_EPS && t < best) { best = t; bs = s0 + j; bcode = code; }
                }
            }
        }

        // ---- hit point, normal, own-surface flags --------------------------------------
        const bool hit = active && bs >= 0;
        d3 P = D3(0, 0, 0), n = D3(0, 0, 0);
        uint32_t flags = 0;
        int mat = 0;
        if (hit) {
            const SurfD sf = a.surf[bs];
            mat = sf.mat;
            P = add(o, scl(d, best));  // scene_settings.py:45
            if (sf.type == RT_SURF_SPHERE) {
                d3 c = D3(sf.d[0], sf.d[1], sf.d[2]);
                d3 oc = sub(P, c);
                n = divs(oc, sf.d[3]);                               // sphere.py:62-63
                if (dots(oc, oc) - sf.d[5] > 0) flags |= HF_CBATCH_POS;  // sphere.py:93
                if (dotb(oc, oc) - sf.d[4] > 0) flags |= HF_CDOT_POS;    // sphere.py:38
            } else if (sf.type == RT_SURF_PLANE) {
                n = D3(sf.d[0], sf.d[1], sf.d[2]);
            } else {
                if (bcode == 0) flags |= HF_NO_OWN;
                else {
                    int ax = (bcode - 1) >> 1; bool pos = (bcode - 1) & 1;
                    double sg = pos ? 1.0 : -1.0;
                    n = D3(ax == 0 ? sg : 0.0, ax == 1 ? sg : 0.0, ax == 2 ? sg : 0.0);
                    double pa = ax == 0 ? P.x : (ax == 1 ? P.y : P.z);
                    double face = pos ? (ax == 0 ? sf.d[3] : (ax == 1 ? sf.d[4] : sf.d[5]))
                                      : (ax == 0 ? sf.d[0] : (ax == 1 ? sf.d[1] : sf.d[2]));
                    if (pos ? (pa > face) : (pa < face)) flags |= HF_BOX_OUTSIDE;
                }
            }
        }
        const MatF m = a.mat[mat];

        if (a.dbg.capacity) {
            uint32_t slot = wave_append(a.dbg.count, active);
            if (active && slot < a.dbg.capacity) {
                a.dbg.gpixel[slot] = gpix; a.dbg.node[slot] = node; a.dbg.surf[slot] = bs;
                a.dbg.t[slot] = hit ? best : 0.0; a.dbg.flags[slot] = flags;
                a.dbg.P[slot * 3 + 0] = P.x; a.dbg.P[slot * 3 + 1] = P.y; a.dbg.P[slot * 3 + 2] = P.z;
                a.dbg.n[slot * 3 + 0] = n.x; a.dbg.n[slot * 3 + 1] = n.y; a.dbg.n[slot * 3 + 2] = n.z;
            }
        }

        // ---- ray left the scene: throughput * background (ray_tracer.py:269-270) ----------
        if (active && !hit) {
            unsafeAtomicAdd(&a.accum[(size_t)0 * a.npix + lpix], (double)(tr * a.fp.bg[0]));
            unsafeAtomicAdd(&a.accum[(size_t)1 * a.npix + lpix], (double)(tg * a.fp.bg[1]));
            unsafeAtomicAdd(&a.accum[(size_t)2 * a.npix + lpix], (double)(tb * a.fp.bg[2]));
        }

        // ---- reserve queue slots: ONE atomic per workgroup and queue ------------------------------
        // (a single counter word sustains only ~90 atomics/us: per-wave reservations of three
        //  queues would serialise the whole kernel on those three words)
        const bool want_r = hit && m.has_refl;
        const bool want_t = hit && m.transp > 0.f;
        const bool push_s = hit, push_r = want_r && !child_cut, push_t = want_t && !child_cut;
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
        const uint32_t slot_s = s_base[0] + s_cnt[0][wv] + __builtin_amdgcn_mbcnt_hi((uint32_t)(bs_ >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bs_, 0u));
        const uint32_t slot_r = s_base[1] + s_cnt[1][wv] + __builtin_amdgcn_mbcnt_hi((uint32_t)(br_ >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)br_, 0u));
        const uint32_t slot_t = s_base[2] + s_cnt[2][wv] + __builtin_amdgcn_mbcnt_hi((uint32_t)(bt_ >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bt_, 0u));
        __syncthreads();  // s_cnt / s_base are rewritten by the next iteration

        // ---- shading record -----------------------------------------------------------
        if (push_s) {
            if (slot_s < a.shade.capacity) {
                const uint32_t slot = slot_s;
                d3 view = normalize(sub(o, P));  // ray_tracer.py:275
                float om = (float)(1.0 - (double)m.transp);  // material.py:28
                a.shade.r0[slot] = make_float4((float)P.x, (float)P.y, (float)P.z, (float)n.x);
                a.shade.r1[slot] = make_float4((float)n.y, (float)n.z, (float)view.x, (float)view.y);
                a.shade.r2[slot] = make_float4((float)view.z, tr * om, tg * om, tb * om);
                a.shade.r3[slot] = make_uint4(lpix, gpix, node, (uint32_t)bs | (flags << 24));
                n_shaded += 1;
            } else ovf_shade += 1;
        }

        // ---- reflection child (ray_tracer.py:192-207) ---------------------------------------
        {
            const float cr = tr * m.kr[0], cg = tg * m.kr[1], cb = tb * m.kr[2];
            if (want_r && child_cut) {  // depth limit: the child returns the background untraced (:261-262)
                unsafeAtomicAdd(&a.accum[(size_t)0 * a.npix + lpix], (double)(cr * a.fp.bg[0]));
                unsafeAtomicAdd(&a.accum[(size_t)1 * a.npix + lpix], (double)(cg * a.fp.bg[1]));
                unsafeAtomicAdd(&a.accum[(size_t)2 * a.npix + lpix], (double)(cb * a.fp.bg[2]));
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
                unsafeAtomicAdd(&a.accum[(size_t)0 * a.npix + lpix], (double)(cr * a.fp.bg[0]));
                unsafeAtomicAdd(&a.accum[(size_t)1 * a.npix + lpix], (double)(cg * a.fp.bg[1]));
                unsafeAtomicAdd(&a.accum[(size_t)2 * a.npix + lpix], (double)(cb * a.fp.bg[2]));
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
        if (s_tr) atomicAdd(&a.counters[PRIMARY ? CNT_PRIMARY : CNT_SECONDARY], s_tr);
        if (s_cut) atomicAdd(&a.counters[CNT_DEPTH_CUT], s_cut);
        if (s_sh) atomicAdd(&a.counters[CNT_SHADED], s_sh);
        if (s_or) atomicAdd(&a.counters[CNT_OVERFLOW_RAY], s_or);
        if (s_os) atomicAdd(&a.counters[CNT_OVERFLOW_SHADE], s_os);
    }
}

void rt_launch_trace(const TraceArgs& a, int blocks, hipStream_t s) {
    if (a.depth == 0) hipLaunchKernelGGL(k_trace<true>, dim3(blocks), dim3(TRACE_BLOCK), 0, s, a);
    else hipLaunchKernelGGL(k_trace<false>, dim3(blocks), dim3(TRACE_BLOCK), 0, s, a);
}
