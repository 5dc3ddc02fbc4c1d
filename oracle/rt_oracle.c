/*
 * rt_oracle.c -- CPU restatement of the reference's render() path, float64.
 *
 * TEST INFRASTRUCTURE.  Only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py may build, load or call this file; the product
 * (raytracer_amd + librt_amd.so) never does.  It restates, in plain C, what the
 * reference computes per pixel, operation by operation and in the reference's
 * order of evaluation, so that hit points come out bit-identical to NumPy's:
 *
 *   np.dot / np.linalg.norm on 3-vectors -> BLAS ddot  = fma(z,z', fma(y,y', x*x'))
 *   np.sum(a*b, axis=1), norm(axis=1)    -> (x*x' + y*y') + z*z'  (rounded products)
 *   np.cross, + - * /, np.sqrt           -> one IEEE operation each, as written
 *   r ** 2 (Python float)                -> libm pow(r, 2.0)
 *   np.mean                              -> NumPy's pairwise sum / n
 *
 * (conventions measured on numpy 2.2.6 + scipy-openblas 0.3.29 in the build
 * container; tests/test_oracle_vs_reference.py re-checks them whenever the
 * reference tree is present, tests/golden pins them when it is not).
 * Build with -ffp-contract=off: every fma() below is deliberate.
 *
 * Parity status: PINNED -- against golden vectors produced by running the
 * unmodified reference in the build container (tests/golden/, generator
 * oracle/make_goldens.py); the reference itself ships no tests or vectors.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif
#include "../include/rt_amd.h"

#define EPS 1e-6 /* /root/reference/utils.py:6 */

typedef struct { double x, y, z; } v3;

typedef struct {
    int type, mat;
    v3 c;          /* centre, or unit normal for planes */
    double p3;     /* radius | offset | edge */
    double rr;     /* r*r        (sphere.py:38) */
    double rpow;   /* r ** 2     (sphere.py:93) */
    v3 bmin, bmax; /* cube.py:32-34 */
} surf_t;

typedef struct { v3 kd, ks, kr; double shin, transp; int has_refl; } mat_t;
typedef struct { v3 pos, col; double spec, shadow, radius; } light_t;

typedef struct {
    v3 cam_pos, fwd, right, up;
    double sdist, swidth;
    v3 bg;
    int N, R, S, M, L;
    surf_t* surf; mat_t* mat; light_t* light;
    uint32_t seed;
} scene_t;

typedef struct {
    uint64_t primary, secondary, depth_cut, shaded, shadow;
} counters_t;

/* one record per find_nearest_intersection call, for chain tests */
typedef struct rto_hit {
    uint32_t pixel, node; int32_t surf; int32_t pad; double t, P[3], n[3];
} rto_hit;

typedef struct {
    counters_t cnt;
    uint32_t pixel;
    rto_hit* hits; int64_t hit_cap, n_hits;
    double* fac; /* N*N shadow factors */
    v3* sdir; double* sdist;
} ctx_t;

/* ---- vector helpers with the reference's rounding ---------------------------- */
static inline v3 V(double x, double y, double z) { v3 r = {x, y, z}; return r; }
static inline v3 sub(v3 a, v3 b) { return V(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline v3 add(v3 a, v3 b) { return V(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline v3 scl(v3 a, double s) { return V(a.x * s, a.y * s, a.z * s); }
static inline v3 divs(v3 a, double s) { return V(a.x / s, a.y / s, a.z / s); }
static inline v3 mulv(v3 a, v3 b) { return V(a.x * b.x, a.y * b.y, a.z * b.z); }
static inline v3 neg(v3 a) { return V(-a.x, -a.y, -a.z); }
/* np.dot -> ddot */
static inline double dotb(v3 a, v3 b) { return fma(a.z, b.z, fma(a.y, b.y, a.x * b.x)); }
/* np.sum(a*b, axis=1) */
static inline double dots(v3 a, v3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
static inline v3 cross(v3 a, v3 b) {
    return V(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
/* utils.normalize (utils.py:9-22) */
static inline v3 normalize(v3 v) {
    double n = sqrt(dotb(v, v));
    if (n == 0) return v;
    return divs(v, n);
}
static inline double comp(v3 a, int i) { return i == 0 ? a.x : (i == 1 ? a.y : a.z); }

/* ---- jitter stream (same contract as oracle/jitter.py and the HIP kernels) ---- */
static inline uint32_t mix32(uint32_t x) {
    x ^= x >> 16; x *= 0x7FEB352Du; x ^= x >> 15; x *= 0x846CA68Bu; x ^= x >> 16; return x;
}
static inline uint32_t pair_word(uint32_t pair) { return mix32(pair * 0x9E3779B9u + 0x5BF03635u); }
static inline uint32_t stream_key(uint32_t seed, uint32_t pixel, uint32_t node, uint32_t light) {
    uint32_t k = mix32(seed + 0x9E3779B9u);
    k = mix32(k ^ pixel);
    k = mix32(k ^ (node * 0x9E3779B1u));
    k = mix32(k ^ ((light + 1u) * 0x85EBCA6Bu));
    return k;
}
/* sample s = i*n + j of an n x n grid: one hash per two neighbours of a row, one byte per coordinate */
static inline void sample_bits(uint32_t key, uint32_t n, uint32_t s, uint32_t* u, uint32_t* v) {
    uint32_t i = s / n, j = s % n;
    uint32_t m = pair_word(i * ((n + 1u) >> 1) + (j >> 1)) ^ key;
    if (j & 1u) { *u = (m >> 8) & 0xFFu; *v = m & 0xFFu; }
    else { *u = m >> 24; *v = (m >> 16) & 0xFFu; }
}
void rto_jitter_bits(uint32_t seed, uint32_t pixel, uint32_t node, uint32_t light, uint32_t n, uint32_t s,
                     uint32_t* u8, uint32_t* v8) {
    sample_bits(stream_key(seed, pixel, node, light), n, s, u8, v8);
}

/* ---- scalar intersections ----------------------------------------------------- */
/* Sphere.intersect (surfaces/sphere.py:11-65). returns 1 on hit. */
static int sphere_hit(const surf_t* s, v3 o, v3 d, double* t_out, v3* n_out) {
    v3 oc = sub(o, s->c);
    double a = dotb(d, d);
    double b = 2.0 * dotb(d, oc);
    double c = dotb(oc, oc) - s->rr;
    double disc = b * b - 4 * a * c;
    if (disc < 0) return 0;
    double sq = sqrt(disc);
    double t1 = (-b - sq) / (2.0 * a);
    double t2 = (-b + sq) / (2.0 * a);
    double t;
    if (t1 > 0) t = t1; else if (t2 > 0) t = t2; else return 0;
    v3 hit = add(o, scl(d, t));
    *n_out = divs(sub(hit, s->c), s->p3);
    *t_out = t;
    return 1;
}
/* InfinitePlane.intersect (surfaces/infinite_plane.py:12-44) */
static int plane_hit(const surf_t* s, v3 o, v3 d, double* t_out, v3* n_out) {
    double denom = dotb(d, s->c);
    if (fabs(denom) < EPS) return 0;
    double t = (s->p3 - dotb(o, s->c)) / denom;
    if (t > 0) { *t_out = t; *n_out = s->c; return 1; }
    return 0;
}
/* Cube.intersect (surfaces/cube.py:11-99) */
static int box_hit(const surf_t* s, v3 o, v3 d, double* t_out, v3* n_out) {
    double t_near = -INFINITY, t_far = INFINITY;
    int axis_n = 0; double sign_n = 1;
    for (int a = 0; a < 3; ++a) {
        double da = comp(d, a), oa = comp(o, a), lo = comp(s->bmin, a), hi = comp(s->bmax, a);
        if (fabs(da) < EPS) {
            if (oa < lo || oa > hi) return 0;
        } else {
            double t1 = (lo - oa) / da, t2 = (hi - oa) / da;
            if (t1 > t2) { double w = t1; t1 = t2; t2 = w; }
            if (t1 > t_near) { t_near = t1; axis_n = a; sign_n = da > 0 ? -1 : 1; }
            if (t2 < t_far) t_far = t2;
            if (t_near > t_far) return 0;
        }
    }
    if (t_far < 0) return 0;
    double nn[3] = {0, 0, 0};
    double t;
    if (t_near > 0) {
        t = t_near; nn[axis_n] = sign_n;
    } else {
        t = t_far;
        v3 hit = add(o, scl(d, t));
        for (int a = 0; a < 3; ++a) {
            if (fabs(comp(hit, a) - comp(s->bmin, a)) < EPS) { nn[a] = -1; break; }
            else if (fabs(comp(hit, a) - comp(s->bmax, a)) < EPS) { nn[a] = 1; break; }
        }
    }
    *t_out = t; *n_out = V(nn[0], nn[1], nn[2]);
    return 1;
}
static inline int surf_hit(const surf_t* s, v3 o, v3 d, double* t, v3* n) {
    switch (s->type) {
        case RT_SURF_SPHERE: return sphere_hit(s, o, d, t, n);
        case RT_SURF_PLANE: return plane_hit(s, o, d, t, n);
        default: return box_hit(s, o, d, t, n);
    }
}

/* ---- batched (shadow) intersections: one ray of the N*N batch ----------------- */
/* Sphere.intersect_batch (surfaces/sphere.py:67-125) */
static inline double sphere_batch(const surf_t* s, v3 o, v3 d) {
    v3 oc = sub(o, s->c);
    double a = dots(d, d);
    double b = 2.0 * dots(d, oc);
    double c = dots(oc, oc) - s->rpow;
    double disc = b * b - 4 * a * c;
    if (!(disc >= 0)) return INFINITY;
    double sq = sqrt(disc);
    double t1 = (-b - sq) / (2.0 * a);
    double t2 = (-b + sq) / (2.0 * a);
    double tn = t1 > 0 ? t1 : t2;
    return tn > EPS ? tn : INFINITY;
}
/* InfinitePlane.intersect_batch (surfaces/infinite_plane.py:46-81) */
static inline double plane_batch(const surf_t* s, v3 o, v3 d) {
    double denom = dots(d, s->c);
    double odn = dots(o, s->c);
    double t = fabs(denom) >= EPS ? (s->p3 - odn) / denom : INFINITY;
    return t > EPS ? t : INFINITY;
}
/* Cube.intersect_batch (surfaces/cube.py:101-173) */
static inline double box_batch(const surf_t* s, v3 o, v3 d) {
    double t_near = -INFINITY, t_far = INFINITY;
    for (int a = 0; a < 3; ++a) {
        double da = comp(d, a), oa = comp(o, a), lo = comp(s->bmin, a), hi = comp(s->bmax, a);
        if (fabs(da) < EPS) {
            if (oa < lo || oa > hi) t_far = -INFINITY;
        } else {
            double inv = 1.0 / da;
            double t1 = (lo - oa) * inv, t2 = (hi - oa) * inv;
            double tmin = fmin(t1, t2), tmax = fmax(t1, t2);
            t_near = fmax(t_near, tmin);
            t_far = fmin(t_far, tmax);
        }
    }
    if (t_near > t_far || t_far < 0) return INFINITY;
    return t_near > EPS ? t_near : (t_far > EPS ? t_far : INFINITY);
}

/* NumPy's pairwise summation (used by np.mean -> add.reduce) */
static double pairwise_sum(const double* a, int n) {
    if (n < 8) {
        double r = 0.;
        for (int i = 0; i < n; ++i) r += a[i];
        return r;
    } else if (n <= 128) {
        double r[8];
        for (int j = 0; j < 8; ++j) r[j] = a[j];
        int i;
        for (i = 8; i < n - (n % 8); i += 8)
            for (int j = 0; j < 8; ++j) r[j] += a[i + j];
        double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < n; ++i) res += a[i];
        return res;
    } else {
        int n2 = n / 2;
        n2 -= n2 % 8;
        return pairwise_sum(a, n2) + pairwise_sum(a + n2, n - n2);
    }
}

/* compute_shadow_ray_ratio_vectorized (ray_tracer.py:15-85) */
static double shadow_ratio(const scene_t* sc, ctx_t* cx, v3 hit, int li, uint32_t node) {
    const light_t* lt = &sc->light[li];
    v3 to_light = sub(lt->pos, hit);
    v3 dir = normalize(to_light);
    int N = sc->N;
    if (lt->shadow == 0 || N == 1) {
        cx->cnt.shadow += 1;
        double factor = 1.0;
        double dist = sqrt(dotb(to_light, to_light));
        for (int s = 0; s < sc->S; ++s) {
            double t; v3 n;
            if (surf_hit(&sc->surf[s], hit, dir, &t, &n) && EPS < t && t < dist) {
                factor *= sc->mat[sc->surf[s].mat].transp;
                if (factor == 0.0) break;
            }
        }
        return factor;
    }
    cx->cnt.shadow += (uint64_t)N * N;
    /* Light.create_basis (light.py:13-32) */
    v3 perp = fabs(dir.x) > 0.1 ? V(0, 1, 0) : V(1, 0, 0);
    v3 right = normalize(cross(dir, perp));
    v3 up = normalize(cross(right, dir));
    /* Light.generate_samples (light.py:35-65) + ray set-up (ray_tracer.py:56-59) */
    uint32_t key = stream_key(sc->seed, cx->pixel, node, (uint32_t)li);
    int n2 = N * N;
    for (int i = 0; i < N; ++i)
        for (int j = 0; j < N; ++j) {
            int s = i * N + j;
            uint32_t ub, vb; sample_bits(key, (uint32_t)N, (uint32_t)s, &ub, &vb);
            double u = (i + ub * (1.0 / 256.0)) / N;
            double v = (j + vb * (1.0 / 256.0)) / N;
            u = u - 0.5; v = v - 0.5;
            v3 off = add(scl(right, u * lt->radius), scl(up, v * lt->radius));
            v3 smp = add(lt->pos, off);
            v3 rd = sub(smp, hit);
            double dist = sqrt(dots(rd, rd));
            cx->sdir[s] = divs(rd, dist);
            cx->sdist[s] = dist;
            cx->fac[s] = 1.0;
        }
    for (int s = 0; s < sc->S; ++s) {
        int all_zero = 1;
        for (int k = 0; k < n2; ++k) if (cx->fac[k] != 0.0) { all_zero = 0; break; }
        if (all_zero) return 0.0;
        const surf_t* sf = &sc->surf[s];
        double tr = sc->mat[sf->mat].transp;
        for (int k = 0; k < n2; ++k) {
            double t;
            switch (sf->type) {
                case RT_SURF_SPHERE: t = sphere_batch(sf, hit, cx->sdir[k]); break;
                case RT_SURF_PLANE: t = plane_batch(sf, hit, cx->sdir[k]); break;
                default: t = box_batch(sf, hit, cx->sdir[k]); break;
            }
            if (t > EPS && t < cx->sdist[k]) {
                if (tr == 0.0) cx->fac[k] = 0.0; else cx->fac[k] *= tr;
            }
        }
    }
    return pairwise_sum(cx->fac, n2) / n2;
}

/* Python's  max(0, x) ** p  (ray_tracer.py:164): int 0 when x <= 0 */
static inline double pow_max0(double x, double p) {
    if (!(x > 0)) return p == 0 ? 1.0 : 0.0;
    return pow(x, p);
}

static v3 trace(const scene_t* sc, ctx_t* cx, v3 o, v3 d, int depth, uint32_t node);

/* calculate_phong_shading (ray_tracer.py:129-171) */
static v3 phong(const scene_t* sc, ctx_t* cx, v3 hit, v3 n, v3 view, const mat_t* m, uint32_t node) {
    v3 color = V(0, 0, 0);
    cx->cnt.shaded += 1;
    for (int li = 0; li < sc->L; ++li) {
        const light_t* lt = &sc->light[li];
        v3 ldir = normalize(sub(lt->pos, hit));
        double ratio = shadow_ratio(sc, cx, hit, li, node);
        double inten = (1.0 - lt->shadow) + lt->shadow * ratio; /* ray_tracer.py:105 */
        if (inten == 0) continue;
        double ndl = dotb(n, ldir);
        double df = ndl > 0 ? ndl : 0.0;
        v3 diffuse = scl(mulv(m->kd, lt->col), df);
        v3 inc = neg(ldir);
        v3 rdir = sub(inc, scl(n, 2 * dotb(inc, n))); /* ray_tracer.py:126 */
        double sf = pow_max0(dotb(rdir, view), m->shin);
        v3 spec = scl(scl(mulv(m->ks, lt->col), sf), lt->spec);
        color = add(color, scl(add(diffuse, spec), inten));
    }
    return color;
}

/* trace_ray (ray_tracer.py:242-294) */
static v3 trace(const scene_t* sc, ctx_t* cx, v3 o, v3 d, int depth, uint32_t node) {
    if (depth >= sc->R) { cx->cnt.depth_cut += 1; return sc->bg; }
    if (depth == 0) cx->cnt.primary += 1; else cx->cnt.secondary += 1;
    /* find_nearest_intersection (scene_settings.py:19-46) */
    double best = INFINITY; v3 bn = V(0, 0, 0); int bs = -1;
    for (int s = 0; s < sc->S; ++s) {
        double t; v3 n;
        if (surf_hit(&sc->surf[s], o, d, &t, &n) && t > EPS && t < best) { best = t; bn = n; bs = s; }
    }
    v3 hit = V(0, 0, 0);
    if (bs >= 0) hit = add(o, scl(d, best));
    if (cx->hits && cx->n_hits < cx->hit_cap) {
        rto_hit* h = &cx->hits[cx->n_hits++];
        h->pixel = cx->pixel; h->node = node; h->surf = bs; h->pad = 0; h->t = bs >= 0 ? best : 0.0;
        h->P[0] = hit.x; h->P[1] = hit.y; h->P[2] = hit.z; h->n[0] = bn.x; h->n[1] = bn.y; h->n[2] = bn.z;
    }
    if (bs < 0) return sc->bg;
    const mat_t* m = &sc->mat[sc->surf[bs].mat];
    v3 view = normalize(sub(o, hit));
    v3 local = phong(sc, cx, hit, bn, view, m, node);
    v3 refl = V(0, 0, 0), trans = V(0, 0, 0);
    if (m->has_refl) { /* ray_tracer.py:192-207 */
        v3 rd = sub(d, scl(bn, 2 * dotb(d, bn)));
        v3 ro = add(hit, scl(bn, EPS));
        refl = mulv(trace(sc, cx, ro, rd, depth + 1, node * 2u), m->kr);
    }
    if (m->transp > 0) { /* ray_tracer.py:226-239 */
        v3 to = add(hit, scl(d, EPS));
        trans = scl(trace(sc, cx, to, d, depth + 1, node * 2u + 1u), m->transp);
    }
    /* Material.blend_colors (material.py:26-30) */
    return add(add(trans, scl(local, 1.0 - m->transp)), refl);
}

/* ---- scene set-up --------------------------------------------------------------- */
static int build_scene(const rt_scene_desc* d, scene_t* sc) {
    if (!d || d->n_surfaces < 0 || d->n_materials < 0 || d->n_lights < 0 || d->n_shadow < 1) return -1;
    memset(sc, 0, sizeof *sc);
    sc->S = d->n_surfaces; sc->M = d->n_materials; sc->L = d->n_lights;
    sc->N = d->n_shadow; sc->R = d->max_rec;
    sc->bg = V(d->background[0], d->background[1], d->background[2]);
    sc->cam_pos = V(d->cam_pos[0], d->cam_pos[1], d->cam_pos[2]);
    v3 look = V(d->cam_look_at[0], d->cam_look_at[1], d->cam_look_at[2]);
    v3 uph = V(d->cam_up[0], d->cam_up[1], d->cam_up[2]);
    /* Camera.__init__ (camera.py:14-16) */
    sc->fwd = normalize(sub(look, sc->cam_pos));
    sc->right = normalize(cross(sc->fwd, uph));
    sc->up = normalize(cross(sc->right, sc->fwd));
    sc->sdist = d->cam_screen_dist; sc->swidth = d->cam_screen_width;
    sc->surf = (surf_t*)calloc((size_t)(sc->S > 0 ? sc->S : 1), sizeof(surf_t));
    sc->mat = (mat_t*)calloc((size_t)(sc->M > 0 ? sc->M : 1), sizeof(mat_t));
    sc->light = (light_t*)calloc((size_t)(sc->L > 0 ? sc->L : 1), sizeof(light_t));
    for (int s = 0; s < sc->S; ++s) {
        surf_t* f = &sc->surf[s];
        f->type = d->surf_type[s]; f->mat = d->surf_material[s];
        if (f->mat < 0 || f->mat >= sc->M || f->type < 0 || f->type > 3) return -1;
        v3 p = V(d->surf_param[0 * sc->S + s], d->surf_param[1 * sc->S + s], d->surf_param[2 * sc->S + s]);
        f->p3 = d->surf_param[3 * sc->S + s];
        f->c = p;
        if (f->type == RT_SURF_PLANE_RAW) {
            f->c = divs(p, sqrt(dotb(p, p))); /* infinite_plane.py:8, offset untouched */
            f->type = RT_SURF_PLANE;
        } else if (f->type == RT_SURF_PLANE) {
            /* unit normal supplied by the caller */
        } else if (f->type == RT_SURF_SPHERE) {
            f->rr = f->p3 * f->p3; f->rpow = pow(f->p3, 2.0);
        } else {
            double half = f->p3 / 2.0;
            f->bmin = V(p.x - half, p.y - half, p.z - half);
            f->bmax = V(p.x + half, p.y + half, p.z + half);
        }
    }
    for (int m = 0; m < sc->M; ++m) {
        const double* q = d->mat_param; int M = sc->M; mat_t* t = &sc->mat[m];
        t->kd = V(q[0 * M + m], q[1 * M + m], q[2 * M + m]);
        t->ks = V(q[3 * M + m], q[4 * M + m], q[5 * M + m]);
        t->kr = V(q[6 * M + m], q[7 * M + m], q[8 * M + m]);
        t->shin = q[9 * M + m]; t->transp = q[10 * M + m];
        t->has_refl = t->kr.x > 0 || t->kr.y > 0 || t->kr.z > 0;
    }
    for (int l = 0; l < sc->L; ++l) {
        const double* q = d->light_param; int L = sc->L; light_t* t = &sc->light[l];
        t->pos = V(q[0 * L + l], q[1 * L + l], q[2 * L + l]);
        t->col = V(q[3 * L + l], q[4 * L + l], q[5 * L + l]);
        t->spec = q[6 * L + l]; t->shadow = q[7 * L + l]; t->radius = q[8 * L + l];
    }
    return 0;
}
static void free_scene(scene_t* sc) { free(sc->surf); free(sc->mat); free(sc->light); }

/* Camera.get_ray (camera.py:18-44) */
static void camera_ray(const scene_t* sc, int px, int py, int W, int H, v3* o, v3* d) {
    double sh = sc->swidth * ((double)H / (double)W);
    double nx = -((px + 0.5) / W - 0.5);
    double ny = (py + 0.5) / H - 0.5;
    v3 centre = add(sc->cam_pos, scl(sc->fwd, sc->sdist));
    v3 pt = add(add(centre, scl(sc->right, nx * sc->swidth)), scl(sc->up, -ny * sh));
    *d = normalize(sub(pt, sc->cam_pos));
    *o = sc->cam_pos;
}

/*
 * Render the listed pixels (x[i], y[i]) of a W x H frame.
 * out_rgb: n*3 doubles = clip(colour,0,1)*255 (ray_tracer.py:330-331).
 * counters: 5 uint64 (primary, secondary, depth_cut, shaded, shadow) or NULL.
 * hits/hit_cap/n_hits: optional record of every nearest-hit query (single thread only).
 * Returns 0, or -1 for a bad descriptor.
 */
int rto_render_pixels(const rt_scene_desc* desc, int W, int H, const int32_t* xs, const int32_t* ys,
                      int64_t n, uint32_t seed, int threads, double* out_rgb, uint64_t* counters,
                      rto_hit* hits, int64_t hit_cap, int64_t* n_hits) {
    scene_t sc;
    if (build_scene(desc, &sc) != 0) { free_scene(&sc); return -1; }
    sc.seed = seed;
    if (hits) threads = 1;
    if (threads < 1) threads = 1;
    counters_t total; memset(&total, 0, sizeof total);
    int64_t hits_written = 0;
    int n2 = sc.N * sc.N;
#ifdef _OPENMP
#pragma omp parallel num_threads(threads)
#endif
    {
        ctx_t cx; memset(&cx, 0, sizeof cx);
        cx.fac = (double*)malloc(sizeof(double) * (size_t)n2);
        cx.sdist = (double*)malloc(sizeof(double) * (size_t)n2);
        cx.sdir = (v3*)malloc(sizeof(v3) * (size_t)n2);
        cx.hits = hits; cx.hit_cap = hit_cap;
#ifdef _OPENMP
#pragma omp for schedule(dynamic, 16)
#endif
        for (int64_t i = 0; i < n; ++i) {
            v3 o, d;
            cx.pixel = (uint32_t)(ys[i] * W + xs[i]);
            camera_ray(&sc, xs[i], ys[i], W, H, &o, &d);
            v3 c = trace(&sc, &cx, o, d, 0, 1u);
            double ch[3] = {c.x, c.y, c.z};
            for (int k = 0; k < 3; ++k) {
                double v = ch[k];
                v = v < 0 ? 0 : (v > 1 ? 1 : v); /* np.clip */
                out_rgb[i * 3 + k] = v * 255;
            }
        }
#ifdef _OPENMP
#pragma omp critical
#endif
        {
            total.primary += cx.cnt.primary; total.secondary += cx.cnt.secondary;
            total.depth_cut += cx.cnt.depth_cut; total.shaded += cx.cnt.shaded; total.shadow += cx.cnt.shadow;
            hits_written += cx.n_hits;
        }
        free(cx.fac); free(cx.sdist); free(cx.sdir);
    }
    if (counters) {
        counters[0] = total.primary; counters[1] = total.secondary; counters[2] = total.depth_cut;
        counters[3] = total.shaded; counters[4] = total.shadow;
    }
    if (n_hits) *n_hits = hits_written;
    free_scene(&sc);
    return 0;
}

/* Derived camera basis, for known-answer tests: out[9] = forward, right, up. */
int rto_camera_basis(const rt_scene_desc* desc, double* out) {
    scene_t sc;
    if (build_scene(desc, &sc) != 0) { free_scene(&sc); return -1; }
    out[0] = sc.fwd.x; out[1] = sc.fwd.y; out[2] = sc.fwd.z;
    out[3] = sc.right.x; out[4] = sc.right.y; out[5] = sc.right.z;
    out[6] = sc.up.x; out[7] = sc.up.y; out[8] = sc.up.z;
    free_scene(&sc);
    return 0;
}

/* Scalar / batched intersection of one surface, for the per-primitive known-answer tables.
 * type/p[4] as in rt_scene_desc. scalar: returns hit flag, t, n[3]. batch: returns t (inf = none). */
int rto_intersect_scalar(int type, const double* p, const double* o, const double* d, double* t, double* n) {
    rt_scene_desc ds; memset(&ds, 0, sizeof ds);
    int32_t ty = type, mt = 0; double mat[RT_MAT_STRIDE] = {0};
    ds.n_surfaces = 1; ds.n_materials = 1; ds.n_shadow = 1; ds.surf_type = &ty; ds.surf_param = p;
    ds.surf_material = &mt; ds.mat_param = mat; ds.cam_look_at[2] = -1; ds.cam_up[1] = 1;
    scene_t sc; if (build_scene(&ds, &sc) != 0) { free_scene(&sc); return -1; }
    v3 nn = V(0, 0, 0); double tt = 0;
    int h = surf_hit(&sc.surf[0], V(o[0], o[1], o[2]), V(d[0], d[1], d[2]), &tt, &nn);
    *t = tt; n[0] = nn.x; n[1] = nn.y; n[2] = nn.z;
    free_scene(&sc);
    return h;
}
double rto_intersect_batch(int type, const double* p, const double* o, const double* d) {
    rt_scene_desc ds; memset(&ds, 0, sizeof ds);
    int32_t ty = type, mt = 0; double mat[RT_MAT_STRIDE] = {0};
    ds.n_surfaces = 1; ds.n_materials = 1; ds.n_shadow = 1; ds.surf_type = &ty; ds.surf_param = p;
    ds.surf_material = &mt; ds.mat_param = mat; ds.cam_look_at[2] = -1; ds.cam_up[1] = 1;
    scene_t sc; if (build_scene(&ds, &sc) != 0) { free_scene(&sc); return NAN; }
    v3 O = V(o[0], o[1], o[2]), D = V(d[0], d[1], d[2]); double t;
    switch (sc.surf[0].type) {
        case RT_SURF_SPHERE: t = sphere_batch(&sc.surf[0], O, D); break;
        case RT_SURF_PLANE: t = plane_batch(&sc.surf[0], O, D); break;
        default: t = box_batch(&sc.surf[0], O, D); break;
    }
    free_scene(&sc);
    return t;
}
