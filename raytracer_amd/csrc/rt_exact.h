// rt_exact.h -- float64 evaluation of the one shadow-ray decision FP32 cannot reproduce by rule.
//
// A shadow ray that starts on a sphere (the hit's own surface) with c = |oc|^2 - r^2 > 0 and points into that
// sphere has the roots t1 = c/|b| (~1e-16) and t2 ~ |b|.  The reference picks `t1 if t1 > 0 else t2`
// (sphere.py:54-59 / :119) and then requires t > 1e-6: if its float64 arithmetic still resolves t1 > 0 the
// sphere does not block, otherwise the far root does.  Which one happens depends on sqrt(fl(b*b - 4ac)) < |b|,
// i.e. on the last bits of b, because c is a rounding residue of one or two quanta.  The functions below repeat
// the reference's float64 operations for one sample in the reference's order (NumPy conventions: np.dot and
// np.linalg.norm are FMA chains, np.sum(x*y, axis=1) is ((p0+p1)+p2) with rounded products, np.cross uses
// rounded products), so the decision is the reference's.
//
// Reference: ray_tracer.py:15-85 (shadow rays), light.py:13-65 (basis, samples), utils.py:9-22 (normalize),
// surfaces/sphere.py:11-65 (scalar), :67-125 (batch).
#pragma once
#include "rt_device.h"

#pragma clang fp contract(off)

namespace rtx {

struct e3 { double x, y, z; };
__device__ __forceinline__ e3 E3(double x, double y, double z) { e3 r; r.x = x; r.y = y; r.z = z; return r; }
__device__ __forceinline__ e3 sub(e3 a, e3 b) { return E3(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ e3 add(e3 a, e3 b) { return E3(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ e3 scl(e3 a, double s) { return E3(a.x * s, a.y * s, a.z * s); }
__device__ __forceinline__ e3 divs(e3 a, double s) { return E3(a.x / s, a.y / s, a.z / s); }
__device__ __forceinline__ double dotb(e3 a, e3 b) { return fma(a.z, b.z, fma(a.y, b.y, a.x * b.x)); }
__device__ __forceinline__ double dots(e3 a, e3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
__device__ __forceinline__ e3 cross(e3 a, e3 b) { return E3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
__device__ __forceinline__ e3 normalize(e3 v) {
    const double n = sqrt(dotb(v, v));
    if (n == 0) return v;
    return divs(v, n);
}

// what stays in registers over the samples of one (hit, light) pair
struct OwnSphere {
    e3 P;          // float64 hit point
    e3 right, up;  // light basis (light.py:24-30)
};

__device__ __forceinline__ OwnSphere own_sphere_setup(const double* p64, const double* light) {
    OwnSphere o;
    o.P = E3(p64[0], p64[1], p64[2]);
    const e3 dir = normalize(sub(E3(light[0], light[1], light[2]), o.P));   // ray_tracer.py:33-34
    const e3 perp = fabs(dir.x) > 0.1 ? E3(0, 1, 0) : E3(1, 0, 0);
    o.right = normalize(cross(dir, perp));
    o.up = normalize(cross(o.right, dir));
    return o;
}

// sample (i, j) of the N x N grid with jitter bits (ub, vb): does the hit's own sphere block it? (multi-sample path)
// Float64 divisions and square roots dominate the cost, so those whose result cannot matter are not performed:
// x / N is x * (1/N) bit for bit when N is a power of two (inv_n != 0), only the root the reference selects is
// divided out, and not even that when the comparison with 1e-6 / dist is decided by a margin no rounding bridges.
__device__ __forceinline__ bool own_sphere_blocks_sample(const OwnSphere& o, const double* light, const SurfD& sph, int N, double inv_n,
                                                         int i, int j, uint32_t ub, uint32_t vb) {
    const double un = (double)i + (double)ub * (1.0 / 256.0), vn = (double)j + (double)vb * (1.0 / 256.0);
    double u = inv_n != 0.0 ? un * inv_n : un / (double)N;               // light.py:53-57
    double v = inv_n != 0.0 ? vn * inv_n : vn / (double)N;
    u = u - 0.5; v = v - 0.5;
    const double radius = light[3];
    const e3 off = add(scl(o.right, u * radius), scl(o.up, v * radius));
    const e3 smp = add(E3(light[0], light[1], light[2]), off);            // light.py:60-63
    const e3 rd = sub(smp, o.P);                                          // ray_tracer.py:57-59
    const double dist = sqrt(dots(rd, rd));
    const e3 d = divs(rd, dist);
    const e3 oc = sub(o.P, E3(sph.d[0], sph.d[1], sph.d[2]));             // sphere.py:88-107
    const double a = dots(d, d);
    const double b = 2.0 * dots(d, oc);
    const double c = dots(oc, oc) - sph.d[5];
    const double disc = b * b - 4 * a * c;
    if (!(disc >= 0)) return false;
    const double sq = sqrt(disc);
    const double two_a = 2.0 * a;
    // t1 = (-b - sq) / (2a), t2 = (-b + sq) / (2a), tn = t1 if t1 > 0 else t2 (sphere.py:110-119); blocked iff
    // 1e-6 < tn < dist (sphere.py:122, ray_tracer.py:71).  2a > 0, so t1 > 0 iff its numerator is.
    const double num1 = -b - sq;
    const double num = num1 > 0 ? num1 : -b + sq;
    if (num < 5e-7 * two_a) return false;                                 // tn < 1e-6 whatever the rounding
    if (num > 2e-6 * two_a && num < dist * two_a * (1.0 - 1e-12)) return true;
    if (num > dist * two_a * (1.0 + 1e-12)) return false;
    const double tn = num / two_a;
    return tn > RT_EPS && tn < dist;
}

// the single ray to the light centre (ray_tracer.py:36-48 with Sphere.intersect, sphere.py:33-59)
__device__ __forceinline__ bool own_sphere_blocks_centre(const double* p64, const double* light, const SurfD& sph) {
    const e3 P = E3(p64[0], p64[1], p64[2]);
    const e3 to_light = sub(E3(light[0], light[1], light[2]), P);
    const e3 d = normalize(to_light);
    const double dist = sqrt(dotb(to_light, to_light));                   // np.linalg.norm, ray_tracer.py:38
    const e3 oc = sub(P, E3(sph.d[0], sph.d[1], sph.d[2]));
    const double a = dotb(d, d);
    const double b = 2.0 * dotb(d, oc);
    const double c = dotb(oc, oc) - sph.d[4];
    const double disc = b * b - 4 * a * c;
    if (disc < 0) return false;
    const double sq = sqrt(disc);
    const double t1 = (-b - sq) / (2.0 * a);
    const double t2 = (-b + sq) / (2.0 * a);
    double t;
    if (t1 > 0) t = t1; else if (t2 > 0) t = t2; else return false;
    return RT_EPS < t && t < dist;
}

// ---- any surface, float64, for lights that lie on a surface (ShadeArgs::special) ------------------------------

// direction and length of the shadow ray of sample (i, j), as the reference forms them (light.py:51-63, ray_tracer.py:56-59)
__device__ __forceinline__ void sample_ray(const OwnSphere& o, const double* light, int N, double inv_n, int i, int j,
                                           uint32_t ub, uint32_t vb, e3& d, double& dist) {
    const double un = (double)i + (double)ub * (1.0 / 256.0), vn = (double)j + (double)vb * (1.0 / 256.0);
    double u = inv_n != 0.0 ? un * inv_n : un / (double)N;
    double v = inv_n != 0.0 ? vn * inv_n : vn / (double)N;
    u = u - 0.5; v = v - 0.5;
    const double radius = light[3];
    const e3 off = add(scl(o.right, u * radius), scl(o.up, v * radius));
    const e3 smp = add(E3(light[0], light[1], light[2]), off);
    const e3 rd = sub(smp, o.P);
    dist = sqrt(dots(rd, rd));
    d = divs(rd, dist);
}

__device__ __forceinline__ double comp(e3 a, int i) { return i == 0 ? a.x : (i == 1 ? a.y : a.z); }

// intersect_batch of the three surface kinds for one ray: t, or +inf (sphere.py:67-125, infinite_plane.py:46-81,
// cube.py:101-173)
__device__ __forceinline__ double batch_t(const SurfD& s, e3 o, e3 d) {
    if (s.type == RT_SURF_SPHERE) {
        const e3 oc = sub(o, E3(s.d[0], s.d[1], s.d[2]));
        const double a = dots(d, d);
        const double b = 2.0 * dots(d, oc);
        const double c = dots(oc, oc) - s.d[5];
        const double disc = b * b - 4 * a * c;
        if (!(disc >= 0)) return INFINITY;
        const double sq = sqrt(disc);
        const double t1 = (-b - sq) / (2.0 * a);
        const double t2 = (-b + sq) / (2.0 * a);
        const double tn = t1 > 0 ? t1 : t2;
        return tn > RT_EPS ? tn : INFINITY;
    }
    if (s.type == RT_SURF_PLANE) {
        const e3 n = E3(s.d[0], s.d[1], s.d[2]);
        const double denom = dots(d, n);
        const double odn = dots(o, n);
        const double t = fabs(denom) >= RT_EPS ? (s.d[3] - odn) / denom : INFINITY;
        return t > RT_EPS ? t : INFINITY;
    }
    double t_near = -INFINITY, t_far = INFINITY;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const double da = comp(d, a), oa = comp(o, a), lo = s.d[a], hi = s.d[3 + a];
        if (fabs(da) < RT_EPS) {
            if (oa < lo || oa > hi) t_far = -INFINITY;
        } else {
            const double inv = 1.0 / da;
            const double t1 = (lo - oa) * inv, t2 = (hi - oa) * inv;
            t_near = fmax(t_near, fmin(t1, t2));
            t_far = fmin(t_far, fmax(t1, t2));
        }
    }
    if (t_near > t_far || t_far < 0) return INFINITY;
    return t_near > RT_EPS ? t_near : (t_far > RT_EPS ? t_far : INFINITY);
}

// the scalar intersect of the three kinds (sphere.py:33-59, infinite_plane.py:31-44, cube.py:36-86): hit and t
__device__ __forceinline__ bool scalar_t(const SurfD& s, e3 o, e3 d, double& t) {
    if (s.type == RT_SURF_SPHERE) {
        const e3 oc = sub(o, E3(s.d[0], s.d[1], s.d[2]));
        const double a = dotb(d, d);
        const double b = 2.0 * dotb(d, oc);
        const double c = dotb(oc, oc) - s.d[4];
        const double disc = b * b - 4 * a * c;
        if (disc < 0) return false;
        const double sq = sqrt(disc);
        const double t1 = (-b - sq) / (2.0 * a);
        if (t1 > 0) { t = t1; return true; }
        const double t2 = (-b + sq) / (2.0 * a);
        if (t2 > 0) { t = t2; return true; }
        return false;
    }
    if (s.type == RT_SURF_PLANE) {
        const e3 n = E3(s.d[0], s.d[1], s.d[2]);
        const double denom = dotb(d, n);
        if (fabs(denom) < RT_EPS) return false;
        const double tt = (s.d[3] - dotb(o, n)) / denom;
        if (tt > 0) { t = tt; return true; }
        return false;
    }
    double t_near = -INFINITY, t_far = INFINITY;
    bool miss = false;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const double da = comp(d, a), oa = comp(o, a), lo = s.d[a], hi = s.d[3 + a];
        if (fabs(da) < RT_EPS) {
            if (oa < lo || oa > hi) miss = true;
        } else {
            double t1 = (lo - oa) / da, t2 = (hi - oa) / da;
            if (t1 > t2) { const double w = t1; t1 = t2; t2 = w; }
            if (t1 > t_near) t_near = t1;
            if (t2 < t_far) t_far = t2;
            if (t_near > t_far) miss = true;
        }
    }
    if (miss || t_far < 0) return false;
    t = t_near > 0 ? t_near : t_far;
    return true;
}

}  // namespace rtx

#pragma clang fp contract(fast)
