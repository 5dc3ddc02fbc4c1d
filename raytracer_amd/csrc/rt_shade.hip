// rt_shade.hip -- Phong shading with N x N jittered soft shadows (gfx950, FP32 VALU).
//
// One lane per shaded hit.  For every light the lane generates the N*N sample
// points of the light's area, and for each sample builds the shadow ray and tests
// it against every surface IN REGISTERS -- shadow rays are never written to memory.
// All lanes of a wave walk the same (light, sample, surface) sequence, so surface
// and light records are wave-uniform (scalar loads / broadcasts) and the loop has
// no divergence apart from the per-lane selects of the own-surface rule.
//
// Shadow rays start exactly on the hit primitive (ray_tracer.py:56).  The reference
// relies on float64 to make the root on that primitive fall below 1e-6; FP32 cannot
// (1e-6 is below its resolution at scene scale), so the hit primitive is resolved by
// rule, using flags the float64 ray-tree kernel computed:
//   plane : its own root is ~0 with either sign            -> never blocks
//   sphere: sample leaves the surface (w.oc >= 0)          -> never blocks
//           sample enters it: c = |P-C|^2 - r^2 > 0 in f64  -> tiny positive near root is
//           selected and rejected => NOT blocked; else the far root t = -2 w.oc blocks
//   box   : sample leaves through the hit face             -> never blocks
//           sample enters: far slab exit blocks (batched path, cube.py:167-168);
//           single-ray path: not blocked when the f64 hit point lies outside the face
// Everything else is the ordinary test with the literal window (1e-6, distance).
//
// Two decisions go back to float64 (rt_exact.h, k_shade_exact), because the reference takes them by the last bits
// of its arithmetic: the "c > 0" case of the sphere rule for samples that enter the sphere steeply (the rule then
// holds only if the float64 quadratic still resolves the near root), and every surface a light lies ON (a ray
// towards that light meets the surface at t = distance up to rounding).
//
// Reference: ray_tracer.py:15-85 (shadow ratio, both paths), :88-106 (intensity),
// :129-171 (Phong), light.py:13-65 (basis + samples), surfaces/*.py intersect_batch
// (multi-sample path) and intersect (single-ray path).
#include "rt_device.h"
#include "../../include/rt_amd.h"
#include <hip/amd_detail/amd_hip_unsafe_atomics.h>

#include "rt_exact.h"

#define SHADE_BLOCK 256

struct f3 { float x, y, z; };
__device__ __forceinline__ f3 F3(float x, float y, float z) { f3 r; r.x = x; r.y = y; r.z = z; return r; }
__device__ __forceinline__ f3 operator-(f3 a, f3 b) { return F3(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ f3 operator+(f3 a, f3 b) { return F3(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ f3 operator*(f3 a, float s) { return F3(a.x * s, a.y * s, a.z * s); }
__device__ __forceinline__ float dot(f3 a, f3 b) { return fmaf(a.z, b.z, fmaf(a.y, b.y, a.x * b.x)); }
__device__ __forceinline__ f3 cross(f3 a, f3 b) {
    return F3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
__device__ __forceinline__ f3 normalize(f3 v) {
    float l2 = dot(v, v);
    if (l2 == 0.f) return v;
    return v * __frsqrt_rn(l2);
}
// x ** y for x > 0 (the Phong lobe, ray_tracer.py:166): exp2(y * log2 x) on the hardware transcendentals.
// Both are accurate to about 1 ulp, so the relative error is ~1e-7 * |y log2 x| -- below 1e-6 wherever the
// result still reaches one 8-bit step; the library powf spends ~100 instructions on the last bit and on
// special cases that cannot occur here.
__device__ __forceinline__ float pow_pos(float x, float y) { return __builtin_amdgcn_exp2f(y * __builtin_amdgcn_logf(x)); }
__device__ __forceinline__ float comp(f3 a, int i) { return i == 0 ? a.x : (i == 1 ? a.y : a.z); }

// Quarter discriminant of the unit-direction quadratic, b^2 - c with c = |oc|^2 - r^2, evaluated as
// r^2 - |oc - b w|^2 (squared distance of the centre from the ray): the literal form cancels two numbers
// of size |oc|^2 and is noise in FP32 once the hit point is a few hundred radii away (a ground plane seen
// near the horizon), this one stays accurate to ~1e-7 |oc| in the distance.
__device__ __forceinline__ float sphere_disc(f3 oc, f3 w, float b, float r2) {
    const f3 q = F3(fmaf(-b, w.x, oc.x), fmaf(-b, w.y, oc.y), fmaf(-b, w.z, oc.z));
    return r2 - dot(q, q);
}

// ---------------------------------------------------------------------------------------
// multi-sample path: semantics of intersect_batch + mask (ray_tracer.py:70-73)
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ bool sphere_blocks_b(const SurfF& s, f3 P, f3 w, float dist, bool own, uint32_t flags) {
    f3 oc = P - F3(s.a.x, s.a.y, s.a.z);
    float b = dot(w, oc);                    // half of the reference's b (unit direction, a = 1)
    float disc = sphere_disc(oc, w, b, s.a.w);
    float sq = __fsqrt_rn(fmaxf(disc, 0.f));
    float t1 = -b - sq, t2 = -b + sq;
    float tn = t1 > 0.f ? t1 : t2;           // sphere.py:119
    bool valid = disc >= 0.f;
    if (own) {                               // start point lies on this sphere
        tn = -2.f * b;
        valid = b < 0.f && !(flags & HF_CBATCH_POS);
    }
    return valid && tn > RT_EPSF && tn < dist;
}
__device__ __forceinline__ bool plane_blocks_b(const SurfF& s, f3 P, f3 w, float dist, bool own) {
    f3 n = F3(s.a.x, s.a.y, s.a.z);
    float denom = dot(w, n);
    float t = (s.a.w - dot(P, n)) / denom;   // infinite_plane.py:70-74
    return !own && fabsf(denom) >= RT_EPSF && t > RT_EPSF && t < dist;
}
__device__ __forceinline__ bool box_blocks_b(const SurfF& s, f3 P, f3 w, float dist, bool own, f3 nrm) {
    float t_near = -INFINITY, t_far = INFINITY;
    const float oo[3] = {P.x, P.y, P.z}, dd[3] = {w.x, w.y, w.z};
    const float lo[3] = {s.a.x, s.a.y, s.a.z}, hi[3] = {s.b.x, s.b.y, s.b.z};
#pragma unroll
    for (int a = 0; a < 3; ++a) {            // cube.py:127-153
        if (fabsf(dd[a]) < RT_EPSF) {
            if (oo[a] < lo[a] || oo[a] > hi[a]) t_far = -INFINITY;
        } else {
            float inv = 1.0f / dd[a];
            float t1 = (lo[a] - oo[a]) * inv, t2 = (hi[a] - oo[a]) * inv;
            t_near = fmaxf(t_near, fminf(t1, t2));
            t_far = fminf(t_far, fmaxf(t1, t2));
        }
    }
    if (own) {
        // leaving through the hit face never blocks; entering blocks at the far slab exit
        bool inward = dot(w, nrm) < 0.f;
        return inward && t_far > RT_EPSF && t_far < dist;
    }
    if (t_near > t_far || t_far < 0.f) return false;
    float t = t_near > RT_EPSF ? t_near : (t_far > RT_EPSF ? t_far : INFINITY);  // cube.py:167-168
    return t > RT_EPSF && t < dist;
}

// ---------------------------------------------------------------------------------------
// single-ray path: semantics of the scalar intersect + caller's window (ray_tracer.py:42-43)
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ bool sphere_blocks_a(const SurfF& s, f3 P, f3 w, float dist, bool own, uint32_t flags) {
    f3 oc = P - F3(s.a.x, s.a.y, s.a.z);
    float b = dot(w, oc);
    float disc = sphere_disc(oc, w, b, s.a.w);
    if (own) {
        float t = -2.f * b;
        return b < 0.f && !(flags & HF_CDOT_POS) && t > RT_EPSF && t < dist;
    }
    if (disc < 0.f) return false;
    float sq = __fsqrt_rn(disc);
    float t1 = -b - sq, t2 = -b + sq;
    float t;
    if (t1 > 0.f) t = t1; else if (t2 > 0.f) t = t2; else return false;  // sphere.py:54-59
    return t > RT_EPSF && t < dist;
}
__device__ __forceinline__ bool plane_blocks_a(const SurfF& s, f3 P, f3 w, float dist, bool own) {
    f3 n = F3(s.a.x, s.a.y, s.a.z);
    float denom = dot(w, n);
    if (own || fabsf(denom) < RT_EPSF) return false;
    float t = (s.a.w - dot(P, n)) / denom;
    return t > 0.f && t > RT_EPSF && t < dist;
}
__device__ __forceinline__ bool box_blocks_a(const SurfF& s, f3 P, f3 w, float dist, bool own, f3 nrm, uint32_t flags) {
    float t_near = -INFINITY, t_far = INFINITY;
    const float oo[3] = {P.x, P.y, P.z}, dd[3] = {w.x, w.y, w.z};
    const float lo[3] = {s.a.x, s.a.y, s.a.z}, hi[3] = {s.b.x, s.b.y, s.b.z};
    bool miss = false;
#pragma unroll
    for (int a = 0; a < 3; ++a) {            // cube.py:45-72
        if (fabsf(dd[a]) < RT_EPSF) {
            if (oo[a] < lo[a] || oo[a] > hi[a]) miss = true;
        } else {
            float t1 = (lo[a] - oo[a]) / dd[a], t2 = (hi[a] - oo[a]) / dd[a];
            t_near = fmaxf(t_near, fminf(t1, t2));
            t_far = fminf(t_far, fmaxf(t1, t2));
        }
    }
    if (own) {
        bool inward = dot(w, nrm) < 0.f;
        return inward && !miss && !(flags & HF_BOX_OUTSIDE) && t_far > RT_EPSF && t_far < dist;
    }
    if (miss || t_near > t_far || t_far < 0.f) return false;
    float t = t_near > 0.f ? t_near : t_far;  // cube.py:79-86
    return t > RT_EPSF && t < dist;
}

__device__ __forceinline__ float3 ld3(const float* p) { return make_float3(p[0], p[1], p[2]); }

// Read-only scene data at a wave-uniform address (a surface, a light): loaded through the constant address space, so the
// compiler may use the scalar unit -- one s_load into SGPRs for the whole wave -- instead of a vector load that returns
// the same 64 bytes to every lane's VGPRs.  (It cannot decide that on its own: the kernels also store to global memory,
// and nothing tells it that those stores never alias the scene.)  The scene is uploaded before the launch.
template <typename T>
__device__ __forceinline__ T uniform_load(const T* p) {
    static_assert(sizeof(T) % 4 == 0, "whole words");
    typedef const uint32_t __attribute__((address_space(4))) * const_words;
    const const_words q = (const_words)(uintptr_t)p;
    uint32_t w[sizeof(T) / 4];
#pragma unroll
    for (int i = 0; i < (int)(sizeof(T) / 4); ++i) w[i] = q[i];
    T t;
    __builtin_memcpy(&t, w, sizeof(T));
    return t;
}

// two FP32 values in an aligned register pair: arithmetic on them is one packed instruction (v_pk_fma_f32 ...)
typedef float f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f2 F2(float x, float y) { f2 r; r.x = x; r.y = y; return r; }
__device__ __forceinline__ f2 pk_fma(f2 a, f2 b, f2 c) { return __builtin_elementwise_fma(a, b, c); }
// Packed FMA forms that read one half of an operand pair for BOTH results (op_sel / op_sel_hi), so that a per-lane
// scalar is used as it lies in its register pair: the compiler has no such form and would keep a (x, x) copy of
// every scalar, which costs the registers the kernel's occupancy depends on.
//   op_sel[k]    : which half of source k feeds the LOW result  (0 = low half, 1 = high half)
//   op_sel_hi[k] : which half of source k feeds the HIGH result
__device__ __forceinline__ f2 pk_fma_bLcL(f2 a, f2 b, f2 c) {   // a * b.x + c.x
    f2 d; asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(d) : "v"(a), "v"(b), "v"(c)); return d;
}
__device__ __forceinline__ f2 pk_fma_bLcH(f2 a, f2 b, f2 c) {   // a * b.x + c.y
    f2 d; asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,1] op_sel_hi:[1,0,1]" : "=v"(d) : "v"(a), "v"(b), "v"(c)); return d;
}
__device__ __forceinline__ f2 pk_fma_bL(f2 a, f2 b, f2 c) {     // a * b.x + c
    f2 d; asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[1,0,1]" : "=v"(d) : "v"(a), "v"(b), "v"(c)); return d;
}
__device__ __forceinline__ f2 pk_fma_bH(f2 a, f2 b, f2 c) {     // a * b.y + c
    f2 d; asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "=v"(d) : "v"(a), "v"(b), "v"(c)); return d;
}
__device__ __forceinline__ f2 pk_fma_cH(f2 a, f2 b, f2 c) {     // a * b + c.y
    f2 d; asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,1] op_sel_hi:[1,1,1]" : "=v"(d) : "v"(a), "v"(b), "v"(c)); return d;
}
__device__ __forceinline__ f2 pk_add_bH(f2 a, f2 b) {           // a + b.y
    f2 d; asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,1]" : "=v"(d) : "v"(a), "v"(b)); return d;
}
// byte B of a word as a float: one instruction (v_cvt_f32_ubyteB)
template <int B> __device__ __forceinline__ float ubyte_f32(uint32_t m) { return (float)((m >> (8 * B)) & 0xFFu); }

struct HitRec {
    f3 P, nrm, view, wgt;
    uint32_t lpix, gpix, node, flags;
    int own_id;
    bool valid;  // false: the slot holds no hit (RT_NULL_HIT)
};

__device__ __forceinline__ HitRec load_hit(const ShadeQueue& q, uint32_t li) {
    const float4 a0 = q.r0[li], a1 = q.r1[li], a2 = q.r2[li];
    const uint4 a3 = q.r3[li];
    HitRec h;
    h.P = F3(a0.x, a0.y, a0.z);
    h.nrm = F3(a0.w, a1.x, a1.y);
    h.view = F3(a1.z, a1.w, a2.x);
    h.wgt = F3(a2.y, a2.z, a2.w);
    h.lpix = a3.x; h.gpix = a3.y; h.node = a3.z;
    h.valid = a3.w != RT_NULL_HIT;
    h.flags = a3.w >> 24;
    h.own_id = (int)(a3.w & 0xFFFFFFu);
    return h;
}

// slots in use: the camera rays' own slots, then the appended hits
__device__ __forceinline__ uint32_t queue_slots(const ShadeQueue& q) {
    const uint32_t appended = *q.count;
    const uint32_t room = q.capacity - q.base;
    return q.base + (appended < room ? appended : room);
}

// The 64 records of the chunk starting at `base`, one per lane.  `active` marks lanes that hold a hit;
// the others get a copy of the chunk's first hit so that wave-wide decisions see nothing foreign (their
// results are dropped).  Returns false when the chunk holds no hit at all.
__device__ __forceinline__ bool load_chunk(const ShadeQueue& q, uint32_t base, uint32_t count, uint32_t lane, HitRec& h, bool& active) {
    const uint32_t idx = base + lane;
    const bool in = idx < count;
    h = load_hit(q, in ? idx : base);
    active = in && h.valid;
    const unsigned long long m = __ballot(active);
    if (!m) return false;
    if (~m) {  // a partly filled chunk (wave-uniform branch)
        const HitRec first = load_hit(q, base + (uint32_t)__builtin_ctzll(m));
        if (!active) h = first;
    }
    return true;
}

// Appends the (shade slot, light word) pairs of the lanes with `pred` to the list of k_shade_exact (all 64 lanes call
// it).  Entries beyond the capacity are only counted: the host then enlarges the list and renders the band again.
// light_word (wave-uniform) with RT_PAIR_FULL: the entries go to the END of the list, downwards (their own counter).
__device__ __forceinline__ void list_pairs(const ShadeArgs& a, bool pred, uint32_t slot, uint32_t light_word) {
    const unsigned long long em = __ballot(pred);
    if (!em) return;
    const bool full = (light_word & RT_PAIR_FULL) != 0u;
    uint32_t e0 = 0;
    if ((threadIdx.x & 63) == 0) e0 = atomicAdd(full ? a.full_count : a.exact_count, (uint32_t)__popcll(em));
    e0 = __builtin_amdgcn_readfirstlane(e0);
    const uint32_t pre = __builtin_amdgcn_mbcnt_hi((uint32_t)(em >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)em, 0u));
    if (pred && e0 + pre < a.exact_capacity) a.exact[full ? a.exact_capacity - 1u - (e0 + pre) : e0 + pre] = make_uint2(slot, light_word);
}

// Light factor of one shadow ray (product of the transparencies of the surfaces it crosses,
// ray_tracer.py:42-46 / :70-78) for scenes of any size.  BATCH picks the multi-sample semantics.
// Without a grid every surface is tested.  With one, the planes are tested first and the segment then
// walks its cells (the walk itself in float64; the surface tests stay FP32); a surface listed in several
// cells must count once: the cells that list a surface form one contiguous run along a ray, so a surface
// also present in the previously visited cell has been counted already.
template <bool BATCH>
__device__ __forceinline__ bool surface_blocks(const SurfF& sf, f3 P, f3 w, float dist, bool own, f3 nrm, uint32_t flags) {
    if (BATCH) {
        if (sf.type == RT_SURF_SPHERE) return sphere_blocks_b(sf, P, w, dist, own, flags);
        if (sf.type == RT_SURF_PLANE) return plane_blocks_b(sf, P, w, dist, own);
        return box_blocks_b(sf, P, w, dist, own, nrm);
    }
    if (sf.type == RT_SURF_SPHERE) return sphere_blocks_a(sf, P, w, dist, own, flags);
    if (sf.type == RT_SURF_PLANE) return plane_blocks_a(sf, P, w, dist, own);
    return box_blocks_a(sf, P, w, dist, own, nrm, flags);
}

template <bool BATCH>
__device__ __forceinline__ float shadow_factor(const ShadeArgs& a, f3 P, f3 w, float dist, int own_id, f3 nrm, uint32_t flags,
                                               unsigned long long reach_mask, const uint32_t* skip = nullptr, uint32_t n_skip = 0) {
    const SceneGrid& g = a.grid;
    float f = 1.f;
    // surfaces with the same geometry as the hit's own one (SurfF::pad names the first of them) start on the ray
    // origin just like it and follow the same rule
    const int own_grp = own_id >= 0 ? a.surf[own_id].pad : -1;
    // surfaces the caller tests itself (scenes of more than 64 surfaces; smaller ones clear their bits in reach_mask)
    auto skipped = [&](uint32_t j) -> bool {
        bool sk = false;
        for (uint32_t q = 0; q < n_skip; ++q) sk = sk || skip[q] == j;
        return sk;
    };
    if (!g.cell_start || a.fp.n_surfaces <= 64) {   // (a grid over so few surfaces only exists when a test forces it)
        const int S = a.fp.n_surfaces;
        if (S <= 64) {   // only the surfaces some lane of the wave can reach from this light (wave-uniform mask)
            for (unsigned long long m = reach_mask & (S == 64 ? ~0ull : (1ull << S) - 1ull); m; m &= m - 1ull) {
                const int s = (int)__builtin_ctzll(m);
                const SurfF sf = uniform_load(a.surf + s);
                f = surface_blocks<BATCH>(sf, P, w, dist, sf.pad == own_grp, nrm, flags) ? f * sf.transp : f;
            }
            return f;
        }
        for (int s = 0; s < S; ++s) {
            if (skipped((uint32_t)s)) continue;
            const SurfF sf = uniform_load(a.surf + s);
            f = surface_blocks<BATCH>(sf, P, w, dist, sf.pad == own_grp, nrm, flags) ? f * sf.transp : f;
        }
        return f;
    }
    if (own_id >= 0) {  // the hit's own surface, by rule, once
        const SurfF sf = a.surf[own_id];
        f = surface_blocks<BATCH>(sf, P, w, dist, true, nrm, flags) ? f * sf.b.w : f;   // b.w: product over its twins
    }
    for (int k = 0; k < g.n_global; ++k) {
        const int j = (int)g.global_items[k];
        const SurfF sf = a.surf[j];
        if (sf.pad == own_grp || skipped((uint32_t)j)) continue;
        f = surface_blocks<BATCH>(sf, P, w, dist, false, nrm, flags) ? f * sf.transp : f;
    }
    if (f == 0.f) return f;
    const double oo[3] = {(double)P.x, (double)P.y, (double)P.z}, dd[3] = {(double)w.x, (double)w.y, (double)w.z};
    double t0 = 0.0, t1 = (double)dist;
#pragma unroll
    for (int ax = 0; ax < 3; ++ax) {
        const double lo = g.lo[ax], hi = g.lo[ax] + g.cell[ax] * (double)g.n[ax];
        if (dd[ax] == 0.0) { if (oo[ax] < lo || oo[ax] > hi) return f; }
        else {
            double ta = (lo - oo[ax]) / dd[ax], tb = (hi - oo[ax]) / dd[ax];
            if (ta > tb) { const double sw = ta; ta = tb; tb = sw; }
            t0 = ta > t0 ? ta : t0; t1 = tb < t1 ? tb : t1;
        }
    }
    if (t0 > t1) return f;
    int idx[3], step[3];
    double t_max[3], t_delta[3];
#pragma unroll
    for (int ax = 0; ax < 3; ++ax) {
        const double p = oo[ax] + dd[ax] * t0;
        int i = (int)floor((p - g.lo[ax]) * g.inv_cell[ax]);
        i = i < 0 ? 0 : (i >= g.n[ax] ? g.n[ax] - 1 : i);
        idx[ax] = i;
        step[ax] = dd[ax] > 0.0 ? 1 : -1;
        if (dd[ax] == 0.0) { t_max[ax] = INFINITY; t_delta[ax] = INFINITY; }
        else {
            t_max[ax] = (g.lo[ax] + g.cell[ax] * (double)(i + (dd[ax] > 0.0 ? 1 : 0)) - oo[ax]) / dd[ax];
            t_delta[ax] = g.cell[ax] / fabs(dd[ax]);
        }
    }
    uint32_t p0 = 0, p1 = 0;  // item range of the previously visited cell
    const int max_steps = g.n[0] + g.n[1] + g.n[2] + 3;
    float t_enter = (float)t0;
    for (int it = 0; it < max_steps; ++it) {
        const uint32_t c = (uint32_t)((idx[2] * g.n[1] + idx[1]) * g.n[0] + idx[0]);
        const uint32_t k0 = g.cell_start[c], k1 = g.cell_start[c + 1];
        // FP32 pre-test per candidate: the ray against the entry's bounding sphere, from a local origin at the cell's
        // entry; a sure miss spares the load of the 64-byte surface record and its test
        const f3 ol = F3(fmaf(w.x, t_enter, P.x), fmaf(w.y, t_enter, P.y), fmaf(w.z, t_enter, P.z));
        const float eps_cell = 3e-7f * (fabsf(ol.x) + fabsf(ol.y) + fabsf(ol.z) + t_enter) + 1e-7f;
        for (uint32_t k = k0; k < k1; ++k) {
            const float4 bnd = g.item_bounds[k];
            const f3 cc = F3(bnd.x, bnd.y, bnd.z) - ol;
            const float bb = dot(cc, w);
            const f3 qq = cc - w * bb;
            const float rm = bnd.w + eps_cell + 2e-6f * (fabsf(cc.x) + fabsf(cc.y) + fabsf(cc.z));
            if (dot(qq, qq) > rm * rm * 1.0001f) continue;   // (NaN: not skipped)
            const uint32_t j = g.items[k];
            bool seen = false;
            for (uint32_t q = p0; q < p1; ++q) seen = seen || g.items[q] == j;
            if (seen) continue;
            const SurfF sf = a.surf[j];
            if (sf.pad == own_grp || skipped(j)) continue;
            f = surface_blocks<BATCH>(sf, P, w, dist, false, nrm, flags) ? f * sf.transp : f;
        }
        if (f == 0.f) return f;
        p0 = k0; p1 = k1;
        const int ax = t_max[0] <= t_max[1] ? (t_max[0] <= t_max[2] ? 0 : 2) : (t_max[1] <= t_max[2] ? 1 : 2);
        if (t_max[ax] > t1) return f;   // the segment ends (or leaves the grid) inside this cell
        t_enter = (float)t_max[ax];
        idx[ax] += step[ax];
        if (idx[ax] < 0 || idx[ax] >= g.n[ax]) return f;
        t_max[ax] += t_delta[ax];
    }
    return f;
}

// Work distribution of the persistent shading kernels.  Work item = one 64-hit chunk with all its lights (splitting chunks
// by light was measured slower, for all chunks and for the last few thousand of a launch alike: the hit is loaded, the
// chunk pre-tested and the accumulators updated once per part, and more items mean more traffic on the work counters).
// The cost per item is uneven (classification), so the waves pull items instead of owning them.  One counter word
// sustains only 50..90 atomics/us, which a small frame (one rank's share of a frame split over 8 GPUs: ~25k items,
// single grabs) would be bound by: the items are dealt into RT_WORK_QUEUES queues, each with a counter on a cache line
// of its own (queue q owns the groups of 4 consecutive items q, q + Q, q + 2Q, ...: every queue covers the whole frame).
// A wave starts on the queue of its workgroup and moves on to the next queue when it finds one empty; a queue never
// refills, so a wave that has seen all of them empty is done.  All values are wave-uniform.
struct ItemQueues {
    uint32_t n_items, n_groups, single_below, queue, queues_left, grab, grab_end;
    __device__ __forceinline__ void init(uint32_t items, uint32_t n_waves) {
        n_items = items;
        n_groups = (items + 3u) >> 2;
        single_below = 16u * (n_waves / RT_WORK_QUEUES + 1u);   // single grabs once a queue holds fewer items than this
        queue = blockIdx.x % RT_WORK_QUEUES;
        queues_left = RT_WORK_QUEUES;
        grab = 0; grab_end = 0;   // positions within the queue
    }
    // the next item of this wave; false: every queue is empty
    __device__ __forceinline__ bool next(uint32_t* work, uint32_t lane, uint32_t& item) {
        for (;;) {
            if (grab >= grab_end) {
                bool got = false;
                while (queues_left) {
                    const uint32_t q_len = ((n_groups + RT_WORK_QUEUES - 1u - queue) / RT_WORK_QUEUES) << 2;
                    uint32_t g0 = 0xFFFFFFFFu, g1 = 0;
                    if (lane == 0 && q_len) {
                        uint32_t* ctr = work + queue * RT_WORK_QUEUE_STRIDE;
                        // (a plain look first: waves arriving at an empty queue do not queue up behind each other's atomics)
                        const uint32_t seen = __hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        if (seen < q_len) {
                            const uint32_t batch = (seen + single_below < q_len) ? 4u : 1u;
                            g0 = atomicAdd(ctr, batch);
                            g1 = g0 + batch;
                        }
                    }
                    g0 = __builtin_amdgcn_readfirstlane(g0);
                    g1 = __builtin_amdgcn_readfirstlane(g1);
                    if (g0 < q_len) { grab = g0; grab_end = g1 < q_len ? g1 : q_len; got = true; break; }
                    queue = queue + 1u == RT_WORK_QUEUES ? 0u : queue + 1u;
                    queues_left -= 1u;
                }
                if (!got) return false;
            }
            item = (((grab >> 2) * RT_WORK_QUEUES + queue) << 2) | (grab & 3u);
            grab += 1u;
            if (item < n_items) return true;   // (the frame's last group may be short)
        }
    }
};

__device__ __forceinline__ void phong_add(f3& color, f3 nrm, f3 ldir, f3 view, const MatF& m, const LightF& lt, float ratio);
__device__ __forceinline__ bool linear_wave_ratio(const ShadeArgs& a, const HitRec& h, const SurfF& own_sf, bool own_valid, const LightF& lt,
                                                  f3 lrel, float lc2, float linv, f3 right, f3 up, bool lit, unsigned long long reach_mask,
                                                  uint32_t key, float& ratio, uint32_t& n_gen, int& n_rows);
__device__ __forceinline__ bool chunk_may_reach(const ShadeArgs& a, f3 Lp, float light_radius, int s, f3 Pc, float Rw, bool own_uniform, int own_grp);

// LINEAR: scenes without a grid (17..64 surfaces) also try the linear rows of k_shade_fast per (wave, light); the grid
// scenes' instance leaves that code out and keeps three waves per SIMD.
template <bool LINEAR>
__global__ __launch_bounds__(SHADE_BLOCK) void k_shade_general(ShadeArgs a) {
    const uint32_t count = queue_slots(a.q);
    const int L = a.fp.n_lights, N = a.fp.n_shadow;
    const float invN = 1.0f / (float)N;
    const float inv_n2 = 1.0f / (float)(N * N);
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t n_waves = (gridDim.x * SHADE_BLOCK) >> 6;
    unsigned long long n_samples = 0;   // samples generated x lanes holding a hit (wave-uniform)
    uint32_t wl_n[7] = {0, 0, 0, 0, 0, 0, 0};   // (wave, light) pairs by class, as in k_shade_fast

    ItemQueues items;
    items.init((count + 63u) >> 6, n_waves);
    uint32_t item;
    while (items.next(a.work, lane, item)) {
        const uint32_t base = item << 6;
        HitRec hr;
        bool active;
        if (!load_chunk(a.q, base, count, lane, hr, active)) continue;
        const f3 P = hr.P, nrm = hr.nrm, view = hr.view;
        const uint32_t gpix = hr.gpix, node = hr.node, flags = hr.flags;
        const uint32_t meta = (uint32_t)hr.own_id;
        const int own_id = (flags & HF_NO_OWN) ? -1 : hr.own_id;
        const SurfF own_sf = a.surf[meta & 0xFFFFFFu];
        const MatF m = a.mat[own_sf.mat];
        // own sphere with c > 0: inward shadow rays are decided in float64 (rt_exact.h)
        const bool exact = active && own_id >= 0 && (flags & HF_EXACT);
        const f3 ov = P - F3(own_sf.a.x, own_sf.a.y, own_sf.a.z);
        const float inward_limit = -0.2f * own_sf.c.x;   // see k_shade_deferred
        const double* p64 = a.q.p64 + (size_t)(base + lane) * 3;

        const unsigned long long n_active = (unsigned long long)__popcll(__ballot(active));
        // bounding sphere of the chunk's hits (lanes without a hit hold a copy of one): centroid and largest distance
        f3 Pc = P;
        float Rw = 0.f;
        if (LINEAR && !a.grid.cell_start && a.fp.n_surfaces <= 64 && !a.no_cull) {
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) { Pc.x += __shfl_xor(Pc.x, o); Pc.y += __shfl_xor(Pc.y, o); Pc.z += __shfl_xor(Pc.z, o); }
            Pc = F3(__builtin_amdgcn_readfirstlane(Pc.x) * 0.015625f, __builtin_amdgcn_readfirstlane(Pc.y) * 0.015625f,
                    __builtin_amdgcn_readfirstlane(Pc.z) * 0.015625f);
            const f3 dP = P - Pc;
            float rw2 = dot(dP, dP);
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) rw2 = fmaxf(rw2, __shfl_xor(rw2, o));
            Rw = __builtin_amdgcn_sqrtf(rw2) * 1.0001f;
        }
        // Every light adds (double)(weight * term) -- the same value whichever kernel shades that (hit, light) pair --
        // and the lights are summed in float64: which pairs a wave defers depends on which hits share the wave, and
        // with FP32 sums the pixel would carry that in its last FP32 bit (a byte flips when it sits on an integer).
        double cx = 0.0, cy = 0.0, cz = 0.0;
        for (int l = 0; l < L; ++l) {
            if (a.special && a.special[l] != a.special[l + 1]) {   // a light on a surface: the whole pair is shaded by k_shade_exact
                list_pairs(a, active, base + lane, (uint32_t)l | RT_PAIR_FULL);
                continue;
            }
            const LightF lt = uniform_load(a.light + l);
            const f3 lrel = F3(lt.pos[0], lt.pos[1], lt.pos[2]) - P;
            const float lc2 = dot(lrel, lrel);
            const float linv = lc2 > 0.f ? __frsqrt_rn(lc2) : 0.f;
            const f3 ldir = lrel * linv;
            float ratio = 1.f;
            if (lt.point_path) {
                // ---- one ray to the light centre (ray_tracer.py:36-48)
                n_samples += n_active;
                if (LINEAR) wl_n[6] += 1;
                const float dist = lc2 * linv;
                ratio = shadow_factor<false>(a, P, ldir, dist, own_id, nrm, flags, ~0ull);
                if (exact && (flags & HF_CDOT_POS) && dot(ldir, ov) < inward_limit &&
                    rtx::own_sphere_blocks_centre(p64, a.light64 + (size_t)l * 4, a.surf64[own_id]))
                    ratio *= own_sf.b.w;
            } else {
                // ---- light basis (light.py:24-30) and the N x N grid (light.py:51-63)
                const f3 perp = fabsf(ldir.x) > 0.1f ? F3(0.f, 1.f, 0.f) : F3(1.f, 0.f, 0.f);
                const f3 right = normalize(cross(ldir, perp));
                const f3 up = normalize(cross(right, ldir));
                const uint32_t key = rt_grid_key(a.fp.seed_key, gpix, node, (uint32_t)l);
                // Scenes of 17..64 surfaces (no grid): which surfaces can any of the wave's 64 hits reach from this light at
                // all?  All N*N samples lie within rho of the light centre, so a surface whose bounding sphere stays clear of
                // the widened segment hit->light is skipped for the whole wave -- and a wave that reaches nothing skips the
                // sample loop altogether (ratio = 1).  A light whose Phong terms vanish for a hit needs no surfaces for it.
                unsigned long long reach_mask = ~0ull;
                const int S = a.fp.n_surfaces;
                bool linear_ok = false, linear_done = false, lit_lane = false;
                if (LINEAR && !a.grid.cell_start && S <= 64 && !a.no_cull) {
                    reach_mask = 0ull;
                    linear_ok = N < 256;
                    const float ndl0 = dot(nrm, ldir);
                    const float rv0 = 2.f * ndl0 * dot(nrm, view) - dot(ldir, view);
                    const bool lit = active && (ndl0 > 0.f || (rv0 > 0.f || m.shin == 0.f));
                    lit_lane = lit;
                    // one lane per surface tests the chunk's bounding sphere against the widened segment to the light
                    // (chunk_may_reach; the hits' own surfaces are tested like the others: a hit lies on its surface)
                    const bool maybe = (int)lane < S && chunk_may_reach(a, F3(lt.pos[0], lt.pos[1], lt.pos[2]), lt.radius, (int)lane, Pc, Rw, false, -1);
                    // ... and the hits themselves test the surfaces that survive (a chunk of deeper rays' hits can be spread
                    // over the whole scene: its bounding sphere then says little)
                    const float llen = lc2 * linv;
                    const float rho = lt.radius * 0.70710678f * 1.0002f + 1e-6f + 4e-7f * (fabsf(P.x) + fabsf(P.y) + fabsf(P.z) + llen);
                    const float inv_lc2 = linv * linv;
                    for (unsigned long long mm = __any(lit) ? __ballot(maybe) : 0ull; mm; mm &= mm - 1ull) {
                        const int s = (int)__builtin_ctzll(mm);
                        const SurfF sf = uniform_load(a.surf + s);
                        bool reach;
                        if (sf.type == RT_SURF_PLANE) {
                            const f3 n = F3(sf.a.x, sf.a.y, sf.a.z);
                            const float hh = sf.a.w - dot(P, n);
                            const float ah = fabsf(hh), dn0 = dot(lrel, n) * (hh < 0.f ? -1.f : 1.f);
                            reach = !((ah - dn0) > rho * 1.0002f + 1e-5f * (ah + fabsf(dn0)));
                        } else {
                            const bool sph = sf.type == RT_SURF_SPHERE;
                            const f3 oc = P - (sph ? F3(sf.a.x, sf.a.y, sf.a.z) : F3(sf.c.x, sf.c.y, sf.c.z));
                            const float rad = sph ? sf.c.x : sf.c.w;
                            const float t = fminf(fmaxf(-dot(oc, lrel) * inv_lc2, 0.f), 1.f);
                            const f3 q = oc + lrel * t;
                            const float rr = rad + rho;
                            reach = !(dot(q, q) > rr * rr * 1.0002f);
                        }
                        if (__any(lit && (reach || s == own_id))) reach_mask |= 1ull << s;
                    }
                }
                {   // own sphere with c > 0 and a sample that may enter it steeply: the loop below applies the rule ("never
                    // blocks") and k_shade_exact adds the float64 correction (same predicate as in k_shade_fast)
                    const float llen = lc2 * linv;
                    bool list = exact && (flags & HF_CBATCH_POS) && reach_mask != 0ull &&
                                dot(lrel, ov) - (0.5002f * lt.radius + 1e-6f) * (fabsf(dot(right, ov)) + fabsf(dot(up, ov))) <
                                    own_sf.c.x * (-0.1998f * llen + 1e-5f * (llen + lt.radius));
                    if (list) {
                        const float ndl0 = dot(nrm, ldir);
                        if (!(ndl0 > 0.f)) {
                            const float rv0 = 2.f * ndl0 * dot(nrm, view) - dot(ldir, view);
                            const float sp = rv0 > 0.f ? pow_pos(rv0, m.shin) : (m.shin == 0.f ? 1.f : 0.f);
                            const float cmax = fmaxf(fmaxf(m.ks[0] * lt.col[0] * hr.wgt.x, m.ks[1] * lt.col[1] * hr.wgt.y), m.ks[2] * lt.col[2] * hr.wgt.z);
                            list = sp * lt.spec * lt.shadow * cmax > a.exact_min;
                        }
                    }
                    const unsigned long long em = __ballot(list);
                    if (em) {
                        uint32_t e0 = 0;
                        if (lane == 0) e0 = atomicAdd(a.exact_count, (uint32_t)__popcll(em));
                        e0 = __builtin_amdgcn_readfirstlane(e0);
                        const uint32_t pre = __builtin_amdgcn_mbcnt_hi((uint32_t)(em >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)em, 0u));
                        if (list && e0 + pre < a.exact_capacity) a.exact[e0 + pre] = make_uint2(base + lane, (uint32_t)l);
                    }
                }
                // 17..64 surfaces: the few surfaces in reach, as linear rows with sample-tile skipping (see k_shade_fast)
                uint32_t n_gen = 0;
                int n_rows = 0;
                if (LINEAR && linear_ok && reach_mask && linear_wave_ratio(a, hr, own_sf, own_id >= 0, lt, lrel, lc2, linv, right, up, lit_lane,
                                                                 reach_mask, key, ratio, n_gen, n_rows)) {
                    n_samples += (unsigned long long)n_gen * n_active;
                    reach_mask = 0ull;   // (done: the generic loop below is skipped, ratio is set)
                    linear_done = true;
                    wl_n[n_rows <= 4 ? n_rows : 4] += 1;
                } else if (LINEAR) wl_n[reach_mask ? 5 : 0] += 1;
                float sum = 0.f;
                uint32_t s_idx = 0;
                for (int i = 0; i < N && reach_mask; ++i) {
                    for (int j = 0; j < N; ++j, ++s_idx) {
                        uint32_t ub, vb;
                        rt_sample_bits(key, (uint32_t)N, (uint32_t)i, (uint32_t)j, ub, vb);
                        const float u = ((float)i + (float)ub * RT_JITTER_INV) * invN - 0.5f;
                        const float v = ((float)j + (float)vb * RT_JITTER_INV) * invN - 0.5f;
                        const f3 sv = lrel + right * (u * lt.radius) + up * (v * lt.radius);
                        const float d2 = dot(sv, sv);
                        const float dinv = __frsqrt_rn(d2);
                        const f3 w = sv * dinv;
                        const float dist = d2 * dinv;
                        sum += shadow_factor<true>(a, P, w, dist, own_id, nrm, flags, reach_mask);
                    }
                }
                if (!linear_done) ratio = reach_mask ? sum * inv_n2 : 1.f;
                if (reach_mask) n_samples += (unsigned long long)(N * N) * n_active;
            }
            // ---- intensity (ray_tracer.py:105) and Phong terms (:157-170)
            f3 term = F3(0.f, 0.f, 0.f);
            phong_add(term, nrm, ldir, view, m, lt, ratio);
            cx += (double)(hr.wgt.x * term.x); cy += (double)(hr.wgt.y * term.y); cz += (double)(hr.wgt.z * term.z);
        }
        if (active) {
            const uint32_t lp = hr.lpix;
            unsafeAtomicAdd(&a.accum[(size_t)0 * a.npix + lp], cx);
            unsafeAtomicAdd(&a.accum[(size_t)1 * a.npix + lp], cy);
            unsafeAtomicAdd(&a.accum[(size_t)2 * a.npix + lp], cz);
        }
    }
    if (lane == 0) {
        if (n_samples) atomicAdd(rt_counter(a.counters, CNT_SAMPLES), n_samples);
#pragma unroll
        for (int c = 0; c < 7; ++c)
            if (LINEAR && wl_n[c]) atomicAdd(rt_counter(a.counters, CNT_WL_NONE + c), (unsigned long long)wl_n[c]);
    }
}


// =======================================================================================
// Small scenes (<= 16 surfaces; template parameter SM = 8 or 16 surface slots): two kernels.
//
// k_shade_fast      classifies, per (hit, light), every surface against the light's sample
//                   square (all samples lie within rho = radius/sqrt(2) of the light centre L):
//       NEVER   : the surface cannot touch any segment P->sample (distance from the segment
//                 P->L to the sphere centre exceeds r + rho; all samples on P's side of a
//                 plane; a box's bounding sphere out of reach) -> no test at all
//       LINEAR  : P outside the sphere and every sample farther away than the tangent
//                 length sqrt(c): "segment hits sphere" == "direction inside the tangent
//                 cone" == sv.(-oc/sqrt(c)) >= |sv|.  With sv = lrel + ur*right + vr*up that
//                 is  A + ur*B + vr*C >= len  with A,B,C fixed per (hit, light, sphere) and
//                 len = sqrt(|lrel|^2 + ur^2 + vr^2) (right, up are orthogonal to lrel).
//                 Planes: segment crosses the plane  <=>  A + ur*B + vr*C > 0.
//                 The hit's own sphere: sample enters it  <=>  same form with -oc.
//                 => 3 FMA per (sample, surface); no shadow ray is ever built.
//       GENERAL : touching / containing geometry, light nearer than the surface, boxes in
//                 reach, single-ray lights.
//     The decision is taken per wavefront (ballot).  A wave whose lanes need no surface for a
//     light gets ratio = 1 without generating a sample; otherwise the surfaces some lane needs
//     (at most KMAX) are compacted into registers and the sample loop runs only those.  A
//     (wave, light) pair with a GENERAL lane, or more than KMAX candidates, is appended to a
//     deferred list.  Margins (2e-4 relative) keep NEVER / LINEAR strictly inside the region
//     where they agree with the generic test.
//
// k_shade_deferred  processes the deferred (64-hit chunk, light) pairs with the generic loop:
//     every shadow ray is built and tested against every surface, with everything that
//     depends only on (hit, surface) hoisted into registers (sphere: oc and -(|oc|^2 - r^2);
//     plane: normal and offset - P.n; box: corners - P).  One (sample, sphere) test is 3 FMA
//     (w.oc) + FMA (disc) + sqrt + bit-select + sub + 2 compares: the root choice
//     `t1 if t1 > 0 else t2` (sphere.py:119) depends only on sign(c), so the sign bit of -c
//     picks the root.  The hit's own surface is inert in that loop (c = +inf, h = NaN, box at
//     +1e30) and resolved by the own-surface rule.
//
// Light contributions are additive, so the two kernels simply add into the same pixel
// accumulators.
// =======================================================================================
#define SMALL_MAX 8            // surfaces the kernels handle at full occupancy (SM = 8); SM = 16 serves 9..16
#define KMAX 8                 // linear tests the compact sample loop keeps in registers
#define NBMAX 3                // boxes it can test beside them
#define PLANE_SCALE 1152921504606846976.0f  // 2^60

// intensity (ray_tracer.py:105) and Phong terms (:157-170) of one light
__device__ __forceinline__ void phong_add(f3& color, f3 nrm, f3 ldir, f3 view, const MatF& m, const LightF& lt, float ratio) {
    const float inten = (1.0f - lt.shadow) + lt.shadow * ratio;
    if (inten != 0.f) {
        const float ndl = fmaxf(0.f, dot(nrm, ldir));
        const float k2 = 2.f * dot(ldir, nrm);
        const f3 rdir = nrm * k2 - ldir;  // reflect(-L, n)
        const float rv = dot(rdir, view);
        const float sp = rv > 0.f ? pow_pos(rv, m.shin) : (m.shin == 0.f ? 1.f : 0.f);
        const float spi = sp * lt.spec;
        color.x += (m.kd[0] * lt.col[0] * ndl + m.ks[0] * lt.col[0] * spi) * inten;
        color.y += (m.kd[1] * lt.col[1] * ndl + m.ks[1] * lt.col[1] * spi) * inten;
        color.z += (m.kd[2] * lt.col[2] * ndl + m.ks[2] * lt.col[2] * spi) * inten;
    }
}

// N x N jittered samples of one light against K compacted linear tests.  Entry e blocks the
// sample iff  lin_e = A + ur*B + vr*C >= len  (tangent cone of a sphere; a plane crossing
// "lin >= 0" is brought to the same form by scaling its row with PLANE_SCALE, which moves its
// threshold from 0 to len * 2^-60).  lin >= len  <=>  lin*|lin| - len^2 >= 0, so no square
// root is taken.  All-opaque scenes only need "any entry blocks" = that sign for the maximum
// of the lin_e, shifted into a bit mask and counted once per row.
// Returns the sum of the light factors of the N*N samples.
//
// NB > 0 adds boxes, tested per sample with slabs on the unnormalised sample direction
// sv = lrel + ur*right + vr*up (the sample sits at parameter 1): entry = max of the per-axis minima,
// exit = min of the maxima, blocked  <=>  entry <= exit, exit >= 0 and entry < 1, i.e.
// min(exit - entry, exit, 1 - entry) >= 0; the three reciprocals are shared by all boxes.  Only boxes
// the hit point keeps clear of are handled here, so the 1e-6 window of cube.py:167-168 cannot bind.
struct BoxSet {
    f3 lrel, right, up;
    float lo[NBMAX][3], hi[NBMAX][3];  // corners relative to the hit point
    float tr[NBMAX];
    // Bounds for whole sample tiles, in the form of the linear rows (A, B, C): the tangent cone of the box's bounding
    // sphere (a segment that misses it misses the box) and of its inscribed sphere (a segment that meets it meets the
    // box).  A row whose cone form does not hold for this hit is +inf (outer: may always block) / -inf (inner: never
    // known to block); a box this lane does not use has outer = -inf.
    float ro[NBMAX][3], ri[NBMAX][3];
};
template <int NB>
__device__ __forceinline__ void box_values(const BoxSet& bx, float ur, float vr, float (&val)[NB > 0 ? NB : 1]) {
    const float sx = fmaf(vr, bx.up.x, fmaf(ur, bx.right.x, bx.lrel.x));
    const float sy = fmaf(vr, bx.up.y, fmaf(ur, bx.right.y, bx.lrel.y));
    const float sz = fmaf(vr, bx.up.z, fmaf(ur, bx.right.z, bx.lrel.z));
    const float ix = __builtin_amdgcn_rcpf(sx), iy = __builtin_amdgcn_rcpf(sy), iz = __builtin_amdgcn_rcpf(sz);
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        const float x1 = bx.lo[b][0] * ix, x2 = bx.hi[b][0] * ix;
        const float y1 = bx.lo[b][1] * iy, y2 = bx.hi[b][1] * iy;
        const float z1 = bx.lo[b][2] * iz, z2 = bx.hi[b][2] * iz;
        const float t_in = fmaxf(fmaxf(fminf(x1, x2), fminf(y1, y2)), fminf(z1, z2));
        const float t_out = fminf(fminf(fmaxf(x1, x2), fmaxf(y1, y2)), fmaxf(z1, z2));
        val[b] = fminf(fminf(t_out - t_in, t_out), 1.0f - t_in);
    }
}

// The linear rows of one (wave, light) pair as the packed sample loop wants them: (B, C) of a row in one register pair,
// the A of rows 2e and 2e + 1 in another.  A lane that does not need a row holds an inert one (A = -inf, B = C = 0).
#define RT_COUNT_SCALE 1073741824.0f   // 2^30: see compact_samples
struct Rows {
    f2 BC[KMAX], AA[KMAX / 2];
    float tr[KMAX];   // transparency of the row's surface (scenes with transparent surfaces)
    __device__ __forceinline__ void clear() {
#pragma unroll
        for (int e = 0; e < KMAX; ++e) { BC[e] = F2(0.f, 0.f); tr[e] = 1.f; }
#pragma unroll
        for (int e = 0; e < KMAX / 2; ++e) AA[e] = F2(-INFINITY, -INFINITY);
    }
    // what compact_samples<K, 0, true, ...> (opaque scene, no boxes) expects of the rows it is given
    __device__ __forceinline__ void scale_for_counting() {
#pragma unroll
        for (int e = 0; e < KMAX / 2; ++e) AA[e] = AA[e] * RT_COUNT_SCALE;
    }
    template <int E>
    __device__ __forceinline__ void set(float a, float b, float c, float t) {
        BC[E] = F2(b, c);
        if (E & 1) AA[E >> 1].y = a; else AA[E >> 1].x = a;
        tr[E] = t;
    }
};

// One instruction per maximum (the compiler's fmaxf first canonicalises operands that come out of inline assembly: two
// extra v_max_f32 per value), and the three-operand form for two further rows at once.
__device__ __forceinline__ float vmax2(float a, float b) { float d; asm("v_max_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b)); return d; }
__device__ __forceinline__ float vmax3(float a, float b, float c) { float d; asm("v_max3_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c)); return d; }
__device__ __forceinline__ float vmax3_zero(float a, float b) { float d; asm("v_max3_f32 %0, %1, %2, 0" : "=v"(d) : "v"(a), "v"(b)); return d; }
__device__ __forceinline__ f2 pk_sq_minus(f2 a, f2 c) {         // a * a - c
    f2 d; asm("v_pk_fma_f32 %0, %1, %1, %2 neg_lo:[0,0,1] neg_hi:[0,0,1]" : "=v"(d) : "v"(a), "v"(c)); return d;
}
// "nothing blocks" as a number: clamp(len^2 - m * m) and clamp(len^2 - m * |m|), 1 where the difference is >= 1 and 0 where
// it is <= 0 -- exact for the sample loop's operands, which are scaled so that no difference lies in between
__device__ __forceinline__ f2 pk_open_sq(f2 m, f2 len2) {       // m >= 0: both samples in one instruction
    f2 d; asm("v_pk_fma_f32 %0, %1, %1, %2 neg_lo:[1,0,0] neg_hi:[1,0,0] clamp" : "=v"(d) : "v"(m), "v"(len2)); return d;
}
__device__ __forceinline__ float open_abs(float m, float len2) {
    float d; asm("v_fma_f32 %0, -%1, |%1|, %2 clamp" : "=v"(d) : "v"(m), "v"(len2)); return d;
}
struct JitterWords4 { uint32_t w[4]; };   // the table words of the (up to) four sample pairs of one tile row

// TABLE: the pair words come from the table in memory through the scalar unit (k_shade_fast: three waves per SIMD hide
// the load, and a pair costs one vector instruction); otherwise they are computed on the scalar unit (the any-size
// kernel runs two waves per SIMD, where waiting for the load costs more than the nine scalar instructions: a
// 40-surface scene at 1080p, N = 8, 1.44 against 1.15 ms).
template <int K, int NB, bool ALL_OPAQUE, bool TABLE>
__device__ __forceinline__ float compact_samples(int N, uint32_t key, const uint32_t* jitter_words, float ku, float koff, float lc2,
                                                 const Rows& R, const BoxSet& bx, uint32_t& n_generated) {
    // Opaque scenes without boxes count the samples nothing blocks as numbers: the sign test  max(lin) * |max(lin)| - len^2 < 0
    // becomes clamp(len^2 - max * |max|), added up in packed FP32 -- one instruction fewer per pair than shifting sign
    // bits into a mask, and no popcount per row.  For the clamp to give exactly 0 or 1 every length is scaled by 2^30
    // (the rows' A -- by the caller, Rows::scale_for_counting --, the cell edges and the jitter step; |lrel|^2 by 2^60):
    // scaling by a power of two is exact, so every
    // decision is the one the unscaled operands give, and the smallest non-zero difference the FMA can return -- one
    // unit in the last place of the 48-bit product m * m, 2^-46 of len^2 -- is at least 1 for len^2 >= 2^-14 (the
    // classification leaves lights nearer than 2^-7 to the generic kernel).  A plane's row (scaled by 2^60 already) overflows to an
    // infinity in the product, which clamps to the right side.
    constexpr bool COUNTED = ALL_OPAQUE && NB == 0;
    constexpr float SCALE = COUNTED ? RT_COUNT_SCALE : 1.0f;
    ku *= SCALE; koff *= SCALE; lc2 *= SCALE * SCALE;
    // The grid is walked in tiles of T x T cells.  Every entry is affine in (ur, vr) minus len, so its range over a
    // tile's rectangle follows from the corners; when, for every lane of the wave, either no entry can block anywhere
    // in the tile or (opaque scenes) one entry blocks everywhere, the tile's samples are not generated at all (the
    // jitter stream is keyed by the sample index, so skipping samples does not shift the others).  With transparent
    // surfaces only "nothing can block" decides a tile: a blocked sample's factor depends on which entries block it.
    // (A second, finer tile level was measured slower: more registers than it saves samples.)
    constexpr int T = 8;
    const float cell = 256.0f * ku;    // width of one grid cell in ur / vr units
    uint32_t open_samples = 0;         // opaque scenes: samples nothing blocks
    float sum = 0.f;                   // otherwise: sum of the samples' light factors
    const f2 (&BC)[KMAX] = R.BC;
    const f2 (&AA)[KMAX / 2] = R.AA;
    const float (&tr)[KMAX] = R.tr;
    // row e for a pair of samples: A + ur * B + vr * C
    auto row_pair = [&](int e, f2 ur, f2 vr) -> f2 {
        const f2 t = (e & 1) ? pk_fma_bLcH(ur, BC[e], AA[e >> 1]) : pk_fma_bLcL(ur, BC[e], AA[e >> 1]);
        return pk_fma_bH(vr, BC[e], t);
    };
    // decided(i0,i1,j0,j1): false = undecided for some lane, true = decided for the whole wave (and accounted for)
    auto tile_decided = [&](int i0, int i1, int j0, int j1) -> bool {
        const float ua = fmaf((float)i0, cell, koff), ub = fmaf((float)i1, cell, koff);
        const float va = fmaf((float)j0, cell, koff), vb = fmaf((float)j1, cell, koff);
        const float u2lo = ua * ub <= 0.f ? 0.f : fminf(ua * ua, ub * ub), u2hi = fmaxf(ua * ua, ub * ub);
        const float v2lo = va * vb <= 0.f ? 0.f : fminf(va * va, vb * vb), v2hi = fmaxf(va * va, vb * vb);
        const float len_lo = __builtin_amdgcn_sqrtf(lc2 + u2lo + v2lo) * 0.99999f;
        const float len_hi = __builtin_amdgcn_sqrtf(lc2 + u2hi + v2hi) * 1.00001f;
        bool can_block = false, must_block = false;
#pragma unroll
        for (int e = 0; e < K; ++e) {
            const float Ae = (e & 1) ? AA[e >> 1].y : AA[e >> 1].x;
            const float bu0 = ua * BC[e].x, bu1 = ub * BC[e].x, cv0 = va * BC[e].y, cv1 = vb * BC[e].y;
            const float hi = Ae + fmaxf(bu0, bu1) + fmaxf(cv0, cv1) - len_lo;
            const float lo = Ae + fminf(bu0, bu1) + fminf(cv0, cv1) - len_hi;
            // (an inert row, A = -inf, must come out as "cannot block": its margin stays finite)
            const float margin = 1e-5f * (fminf(fabsf(Ae), 1e36f) + fmaxf(fabsf(bu0), fabsf(bu1)) + fmaxf(fabsf(cv0), fabsf(cv1)) + len_hi);
            can_block = can_block || !(hi < -margin);   // (also true for NaN: never skip on doubt)
            must_block = must_block || lo > margin;
        }
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            {   // bounding sphere: can the box block anywhere in the tile?
                const float bu0 = ua * bx.ro[b][1], bu1 = ub * bx.ro[b][1], cv0 = va * bx.ro[b][2], cv1 = vb * bx.ro[b][2];
                const float hi = bx.ro[b][0] + fmaxf(bu0, bu1) + fmaxf(cv0, cv1) - len_lo;
                const float margin = 1e-5f * (fminf(fabsf(bx.ro[b][0]), 1e30f) + fmaxf(fabsf(bu0), fabsf(bu1)) + fmaxf(fabsf(cv0), fabsf(cv1)) + len_hi);
                can_block = can_block || !(hi < -margin);
            }
            if (ALL_OPAQUE) {   // inscribed sphere: does it block everywhere in the tile?
                const float bu0 = ua * bx.ri[b][1], bu1 = ub * bx.ri[b][1], cv0 = va * bx.ri[b][2], cv1 = vb * bx.ri[b][2];
                const float lo = bx.ri[b][0] + fminf(bu0, bu1) + fminf(cv0, cv1) - len_hi;
                const float margin = 1e-5f * (fminf(fabsf(bx.ri[b][0]), 1e30f) + fmaxf(fabsf(bu0), fabsf(bu1)) + fmaxf(fabsf(cv0), fabsf(cv1)) + len_hi);
                must_block = must_block || lo > margin;
            }
        }
        if (ALL_OPAQUE) {
            if (__any(can_block && !must_block)) return false;
            open_samples += must_block ? 0u : (uint32_t)((i1 - i0) * (j1 - j0));
        } else {
            if (__any(can_block)) return false;
            sum += (float)((i1 - i0) * (j1 - j0));
        }
        return true;
    };
    const uint32_t stride = rt_pair_stride((uint32_t)N);
    const f2 KU = F2(ku, 2.f * cell);    // (jitter step, two columns)
    for (int i0 = 0; i0 < N; i0 += T) {
        const int i1 = i0 + T < N ? i0 + T : N;
        for (int j0 = 0; j0 < N; j0 += T) {
            const int j1 = j0 + T < N ? j0 + T : N;
            if (tile_decided(i0, i1, j0, j1)) continue;
            n_generated += (uint32_t)((i1 - i0) * (j1 - j0));   // (wave-uniform)
            // Two neighbouring samples of a row share one jitter word (rt_device.h) and are evaluated together in packed
            // FP32 (v_pk_fma_f32: both samples' FMA in one instruction): .x = sample j, .y = sample j + 1.  A tile
            // starts at an even column, so pairs never straddle tiles; only the last pair of a row of an odd grid has
            // no second sample.  The words of a tile row -- up to four pairs -- are wave-uniform: one scalar load,
            // issued a row ahead (the table has a spare row and spare words at its end), and one XOR with the lane's key
            // per pair.
            const int n_pairs = (j1 - j0 + 1) >> 1;
            const bool odd_tail = ((j1 - j0) & 1) != 0;
            const f2 vcol0 = F2(fmaf((float)j0, cell, koff), fmaf((float)(j0 + 1), cell, koff));   // left edges of columns j0, j0 + 1
            const uint32_t* wp = jitter_words + (uint32_t)i0 * stride + ((uint32_t)j0 >> 1);
            JitterWords4 words{}, words_next{};
            if (TABLE) words = uniform_load((const JitterWords4*)wp);
            f2 open2 = F2(0.f, 0.f);   // COUNTED: open samples of the tile (.x even columns, .y odd ones)
            for (int i = i0; i < i1; ++i) {
                wp += stride;
                if (TABLE) words_next = uniform_load((const JitterWords4*)wp);
                uint32_t mask = 0;
                f2 last_open = F2(0.f, 0.f);
                const f2 UL = F2(fmaf((float)i, cell, koff), lc2);  // (left edge of row i, |lrel|^2)
                f2 vcol = vcol0;
                float row_sum = 0.f;
#pragma unroll
                for (int q = 0; q < T / 2; ++q) {
                    if (q < n_pairs) {   // (wave-uniform)
                        const uint32_t mm = (TABLE ? words.w[q] : rt_pair_word((uint32_t)i * stride + ((uint32_t)j0 >> 1) + (uint32_t)q)) ^ key;
                        const f2 ub = F2(ubyte_f32<3>(mm), ubyte_f32<1>(mm));
                        const f2 vb = F2(ubyte_f32<2>(mm), ubyte_f32<0>(mm));
                        const f2 ur = pk_fma_bLcL(ub, KU, UL);
                        const f2 vr = pk_fma_bL(vb, KU, vcol);
                        vcol = pk_add_bH(vcol, KU);
                        const f2 len2 = pk_fma(ur, ur, pk_fma_cH(vr, vr, UL));
                        if (COUNTED) {
                            // any entry blocks  <=>  the largest lin reaches len.  With an even number of rows the last
                            // maximum takes a zero along for free, and a value that is not negative needs no |.|: both
                            // samples in one packed instruction.
                            f2 vm = row_pair(0, ur, vr);
#pragma unroll
                            for (int e = 1; e + 1 < K - (K % 2 == 0 ? 1 : 0); e += 2) {
                                const f2 l1 = row_pair(e, ur, vr), l2 = row_pair(e + 1, ur, vr);
                                vm.x = vmax3(vm.x, l1.x, l2.x); vm.y = vmax3(vm.y, l1.y, l2.y);
                            }
                            f2 open;
                            if (K % 2 == 0) {
                                const f2 l1 = row_pair(K - 1, ur, vr);
                                vm.x = vmax3_zero(vm.x, l1.x); vm.y = vmax3_zero(vm.y, l1.y);
                                open = pk_open_sq(vm, len2);
                            } else {
                                open.x = open_abs(vm.x, len2.x); open.y = open_abs(vm.y, len2.y);
                            }
                            open2 += open;
                            last_open = open;
                        } else if (ALL_OPAQUE) {
                            // (with boxes: the sign test lin * |lin| - len^2 >= 0, the boxes' values beside it)
                            f2 res = F2(-INFINITY, -INFINITY);
                            if (K > 0) {
                                f2 vm = row_pair(0, ur, vr);
#pragma unroll
                                for (int e = 1; e + 1 < K - (K % 2 == 0 ? 1 : 0); e += 2) {
                                    const f2 l1 = row_pair(e, ur, vr), l2 = row_pair(e + 1, ur, vr);
                                    vm.x = vmax3(vm.x, l1.x, l2.x); vm.y = vmax3(vm.y, l1.y, l2.y);
                                }
                                if (K % 2 == 0) {
                                    const f2 l1 = row_pair(K - 1, ur, vr);
                                    vm.x = vmax3_zero(vm.x, l1.x); vm.y = vmax3_zero(vm.y, l1.y);
                                    res = pk_sq_minus(vm, len2);
                                } else {
                                    res.x = fmaf(vm.x, fabsf(vm.x), -len2.x); res.y = fmaf(vm.y, fabsf(vm.y), -len2.y);
                                }
                            }
                            if (NB > 0) {
                                float val[NB > 0 ? NB : 1];
                                box_values<NB>(bx, ur.x, vr.x, val);
#pragma unroll
                                for (int b = 0; b < NB; ++b) res.x = __builtin_fmaxf(res.x, val[b]);
                                box_values<NB>(bx, ur.y, vr.y, val);
#pragma unroll
                                for (int b = 0; b < NB; ++b) res.y = __builtin_fmaxf(res.y, val[b]);
                            }
                            // sign bit set <=> nothing blocks this sample
                            mask = __builtin_amdgcn_alignbit(mask, __float_as_uint(res.x), 31);
                            mask = __builtin_amdgcn_alignbit(mask, __float_as_uint(res.y), 31);
                        } else {
                            float f0 = 1.f, f1 = 1.f;
#pragma unroll
                            for (int e = 0; e < K; ++e) {
                                const f2 lin = row_pair(e, ur, vr);
                                f0 = fmaf(lin.x, fabsf(lin.x), -len2.x) >= 0.f ? f0 * tr[e] : f0;
                                f1 = fmaf(lin.y, fabsf(lin.y), -len2.y) >= 0.f ? f1 * tr[e] : f1;
                            }
                            if (NB > 0) {
                                float val[NB > 0 ? NB : 1];
                                box_values<NB>(bx, ur.x, vr.x, val);
#pragma unroll
                                for (int b = 0; b < NB; ++b) f0 = val[b] >= 0.f ? f0 * bx.tr[b] : f0;
                                box_values<NB>(bx, ur.y, vr.y, val);
#pragma unroll
                                for (int b = 0; b < NB; ++b) f1 = val[b] >= 0.f ? f1 * bx.tr[b] : f1;
                            }
                            row_sum += f0 + ((odd_tail && q == n_pairs - 1) ? 0.f : f1);
                        }
                    }
                }
                if (COUNTED) {
                    if (odd_tail) open2.y -= last_open.y;   // (wave-uniform) the second sample of the row's last pair does not exist
                } else if (ALL_OPAQUE) {
                    if (odd_tail) mask &= ~1u;   // the bit shifted in last belongs to no sample
                    open_samples += (uint32_t)__popc(mask);
                } else sum += row_sum;
                if (TABLE) words = words_next;
            }
            if (COUNTED) open_samples += (uint32_t)(open2.x + open2.y);   // (at most 64: exact)
        }
    }
    return ALL_OPAQUE ? (float)open_samples : sum;
}

// Wave-level pre-test of one (light, surface) combination, evaluated by ONE lane for the whole 64-hit chunk: can the
// surface matter for ANY hit of the chunk?  The hits lie within Rw of Pc, so every segment hit -> sample lies within Rw
// of the corresponding segment from Pc, and the per-hit reach tests of the classification (same forms, same margins)
// hold for all of them once Rw and the bounds of the hit-dependent allowances are added.  false: the classification of
// that combination would find "never" in every lane, and is skipped.  Errs only towards true.
__device__ __forceinline__ bool chunk_may_reach(const ShadeArgs& a, f3 Lp, float light_radius, int s, f3 Pc, float Rw, bool own_uniform, int own_grp);
__device__ __forceinline__ bool chunk_may_reach(const ShadeArgs& a, int l, int s, f3 Pc, float Rw, bool own_uniform, int own_grp) {
    const LightF* lt = a.light + l;
    return chunk_may_reach(a, F3(lt->pos[0], lt->pos[1], lt->pos[2]), lt->radius, s, Pc, Rw, own_uniform, own_grp);
}
__device__ __forceinline__ bool chunk_may_reach(const ShadeArgs& a, f3 Lp, float light_radius, int s, f3 Pc, float Rw, bool own_uniform, int own_grp) {
    const SurfF* sf = a.surf + s;
    const f3 lrel = Lp - Pc;
    const float lc2 = dot(lrel, lrel);
    const float llen = __builtin_amdgcn_sqrtf(lc2);
    const float pabs = fabsf(Pc.x) + fabsf(Pc.y) + fabsf(Pc.z);
    const float rho = light_radius * 0.70710678f * 1.0002f + 1e-6f + 4e-7f * (llen + pabs + 4.f * Rw);   // >= every lane's rho
    const int type = sf->type;
    const bool own_all = own_uniform && sf->pad == own_grp;   // every hit of the chunk lies on this surface (or a twin)
    if (type == RT_SURF_PLANE) {
        if (own_all) return false;
        const f3 n = F3(sf->a.x, sf->a.y, sf->a.z);
        const float hh = sf->a.w - dot(Pc, n);          // the hits' side of the plane ...
        const float hl = sf->a.w - dot(Lp, n);          // ... and the light's: |ah - dn0| of the classification is sg * hl
        const float ah = fabsf(hh), dn = fabsf(dot(lrel, n));
        const float fuzz = 2e-6f * (fabsf(sf->a.w) + pabs + fabsf(Lp.x) + fabsf(Lp.y) + fabsf(Lp.z) + Rw);
        const bool one_side = ah > Rw * 1.001f + fuzz;
        const float sg = hh < 0.f ? -1.f : 1.f;
        const bool never = one_side && sg * hl > rho * 1.0003f + 1e-5f * (ah + dn + 2.f * Rw) + fuzz;
        return !never;
    }
    const bool sph = type == RT_SURF_SPHERE;
    if (sph && own_all) return false;                    // the own sphere has its own entry in the classification
    if (!sph && own_all) return true;                    // the own box: a sample may enter it
    const f3 oc = Pc - (sph ? F3(sf->a.x, sf->a.y, sf->a.z) : F3(sf->c.x, sf->c.y, sf->c.z));
    const float rad = sph ? sf->c.x : sf->c.w;
    const float t = lc2 > 0.f ? fminf(fmaxf(-dot(oc, lrel) / lc2, 0.f), 1.f) : 0.f;
    const f3 q = oc + lrel * t;
    // the segments fan out from the hits: at parameter s they are within Rw + s * rho of the axis Pc -> L, and a contact
    // needs |s - t| * llen <= rad + Rw + s * rho, which bounds s (the classification's s_hi, with Rw added to the radius)
    const float radw = rad + Rw * 1.001f;
    const float s_hi = llen > 1.5f * rho + Rw ? 1.0003f * (radw + t * llen) / (llen - rho - Rw) : 1e30f;
    const float lim = radw + rho * fminf(1.f, s_hi) * 1.0003f + 2e-6f * (fabsf(oc.x) + fabsf(oc.y) + fabsf(oc.z) + llen);
    return !(dot(q, q) > lim * lim * 1.0003f);           // (NaN: true)
}

// The (wave, light) classification of k_shade_fast for scenes whose surfaces do not fit its register slots
// (k_shade_general, no grid): the candidates are the surfaces of the wave's reach mask instead of all S.  With at
// most KMAX of them, none a box some lane can reach, and every lane's tests NEVER or LINEAR, the wave's shadow ratio
// comes from the compact sample loop (tile skipping included) and the function returns true; otherwise false, and
// the caller runs its generic loop.  Conditions and margins are those of k_shade_fast, kept identical on purpose:
// both were validated against the generic tests together.
__device__ __forceinline__ bool linear_wave_ratio(const ShadeArgs& a, const HitRec& h, const SurfF& own_sf, bool own_valid, const LightF& lt,
                                                  f3 lrel, float lc2, float linv, f3 right, f3 up, bool lit, unsigned long long reach_mask,
                                                  uint32_t key, float& ratio, uint32_t& n_gen, int& n_rows) {
    const int n_reach = __popcll(reach_mask);
    if (n_reach > KMAX) return false;
    const int N = a.fp.n_shadow;
    const f3 P = h.P;
    const float llen = lc2 * linv;
    const float rho_base = lt.radius * 0.70710678f * 1.0002f + 1e-6f;
    const float rho = rho_base + fmaf(4e-7f, llen, 4e-7f * (fabsf(P.x) + fabsf(P.y) + fabsf(P.z)));
    const float inv_lc2 = linv * linv;
    const float s_hi_scale = llen > 1.5f * rho ? 1.0002f / (llen - rho) : 1e30f;
    const bool own_sphere = own_valid && own_sf.type == RT_SURF_SPHERE;
    const f3 ov = P - F3(own_sf.a.x, own_sf.a.y, own_sf.a.z);
    float A[KMAX + 1], B[KMAX + 1], C[KMAX + 1], tr[KMAX + 1];
    bool gen = llen < 0.0078125f;   // (as in k_shade_fast: the counted sample loop wants the light at least 2^-7 away)
    uint32_t lin = 0;
    unsigned long long rem = reach_mask;
#pragma unroll
    for (int e = 0; e < KMAX; ++e) {
        A[e] = -INFINITY; B[e] = 0.f; C[e] = 0.f; tr[e] = 1.f;
        if (e >= n_reach) continue;   // (wave-uniform)
        const int s = (int)__builtin_ctzll(rem);
        rem &= rem - 1ull;
        const SurfF sf = uniform_load(a.surf + s);
        tr[e] = sf.transp;
        const bool own = own_valid && sf.pad == own_sf.pad;   // the own surface or a twin of it
        if (sf.type == RT_SURF_SPHERE) {
            const f3 oc = P - F3(sf.a.x, sf.a.y, sf.a.z);
            const float c = dot(oc, oc) - sf.a.w;
            const float r = sf.c.x;
            const float t = fminf(fmaxf(-dot(oc, lrel) * inv_lc2, 0.f), 1.f);
            const f3 q = oc + lrel * t;
            const float reach = r + rho * fminf(1.f, s_hi_scale * (r + t * llen));
            const bool never = own || !lit || dot(q, q) > reach * reach * 1.0002f;
            if (!never) {
                const float lmin = llen - rho;
                const bool linear_ok = c >= 1e-3f * r * r && c < 1e4f * r * r && lmin > 0.f && lmin * lmin > c * 1.0002f;
                if (!linear_ok) gen = true;
                else {
                    const float k = -__builtin_amdgcn_rsqf(c);
                    A[e] = dot(lrel, oc) * k; B[e] = dot(right, oc) * k; C[e] = dot(up, oc) * k;
                    lin |= 1u << e;
                }
            }
        } else if (sf.type == RT_SURF_PLANE) {
            const f3 n = F3(sf.a.x, sf.a.y, sf.a.z);
            const float hh = sf.a.w - dot(P, n);
            const float ah = fabsf(hh);
            const float sg = hh < 0.f ? -1.f : 1.f;
            const float dn0 = dot(lrel, n) * sg;
            const bool never = own || !lit || (ah - dn0) > rho * 1.0002f + 1e-5f * (ah + fabsf(dn0));
            if (!never) {
                if (!(ah > 1e-4f && ah > RT_EPSF * 1.01f * (llen + rho) && ah > 64.f * (rho - rho_base))) gen = true;
                else {
                    A[e] = (dn0 - ah) * PLANE_SCALE; B[e] = dot(right, n) * (sg * PLANE_SCALE); C[e] = dot(up, n) * (sg * PLANE_SCALE);
                    lin |= 1u << e;
                }
            }
        } else if (lit) {
            if (own) {
                if (dot(lrel, h.nrm) - rho < 0.f) gen = true;   // some sample may enter the box
            } else {
                const f3 oc = P - F3(sf.c.x, sf.c.y, sf.c.z);   // bounding sphere
                const float t = fminf(fmaxf(-dot(oc, lrel) * inv_lc2, 0.f), 1.f);
                const f3 q = oc + lrel * t;
                const float reach = sf.c.w + rho;
                if (!(dot(q, q) > reach * reach * 1.0002f)) gen = true;   // (a box in reach: the generic loop)
            }
        }
    }
    A[KMAX] = -INFINITY; B[KMAX] = 0.f; C[KMAX] = 0.f; tr[KMAX] = own_sf.b.w;
    if (own_sphere && !(h.flags & HF_CBATCH_POS) && lit) {   // the own sphere blocks a sample that enters it
        const float r = own_sf.c.x;
        const float dl = dot(lrel, ov);
        if (dl - rho * r * 1.0002f < 0.f) {
            if (llen - rho > 2.0002f * r) {
                const float k = -2.0e6f;
                A[KMAX] = dl * k; B[KMAX] = dot(right, ov) * k; C[KMAX] = dot(up, ov) * k;
                lin |= 1u << KMAX;
            } else gen = true;
        }
    }
    if (__any(gen)) return false;
    uint32_t cand = 0;
#pragma unroll
    for (int s = 0; s < KMAX + 1; ++s)
        if (__ballot((lin >> s) & 1u)) cand |= 1u << s;
    const int k = __popc(cand);
    if (k > KMAX) return false;
    ratio = 1.f;
    n_rows = k;
    if (k == 0) return true;
    Rows R;
    R.clear();
    uint32_t crem = cand;
#pragma unroll
    for (int e = 0; e < KMAX; ++e) {
        const int sl = crem ? (int)__builtin_ctz(crem) : -1;
        crem &= crem - 1u;
        if (e >= k) continue;
        float ra = -INFINITY, rb = 0.f, rc = 0.f, rt = 1.f;
#pragma unroll
        for (int t = 0; t < KMAX + 1; ++t)
            if (sl == t) { ra = A[t]; rb = B[t]; rc = C[t]; rt = tr[t]; }
        switch (e) {   // (e is a constant after unrolling)
            case 0: R.set<0>(ra, rb, rc, rt); break;
            case 1: R.set<1>(ra, rb, rc, rt); break;
            case 2: R.set<2>(ra, rb, rc, rt); break;
            case 3: R.set<3>(ra, rb, rc, rt); break;
            case 4: R.set<4>(ra, rb, rc, rt); break;
            case 5: R.set<5>(ra, rb, rc, rt); break;
            case 6: R.set<6>(ra, rb, rc, rt); break;
            default: R.set<7>(ra, rb, rc, rt); break;
        }
    }
    const float ku = RT_JITTER_INV * (1.0f / (float)N) * lt.radius;
    const float koff = -0.5f * lt.radius;
    float sum;
    BoxSet bx;   // (no boxes here: a wave that reaches one takes the generic loop -- box rows were measured, and the
                 //  registers of their sample loops cost this kernel more than the generic loop costs the few waves)
    const int kk = k <= 1 ? 1 : (k <= 2 ? 2 : (k <= 4 ? 4 : 8));   // (spare rows are inert)
    if (a.all_opaque) {
        R.scale_for_counting();
        switch (kk) {
            case 1: sum = compact_samples<1, 0, true, false>(N, key, a.jitter_words, ku, koff, lc2, R, bx, n_gen); break;
            case 2: sum = compact_samples<2, 0, true, false>(N, key, a.jitter_words, ku, koff, lc2, R, bx, n_gen); break;
            case 4: sum = compact_samples<4, 0, true, false>(N, key, a.jitter_words, ku, koff, lc2, R, bx, n_gen); break;
            default: sum = compact_samples<8, 0, true, false>(N, key, a.jitter_words, ku, koff, lc2, R, bx, n_gen); break;
        }
    } else {
        switch (kk) {
            case 1: sum = compact_samples<1, 0, false, false>(N, key, a.jitter_words, ku, koff, lc2, R, bx, n_gen); break;
            case 2: sum = compact_samples<2, 0, false, false>(N, key, a.jitter_words, ku, koff, lc2, R, bx, n_gen); break;
            case 4: sum = compact_samples<4, 0, false, false>(N, key, a.jitter_words, ku, koff, lc2, R, bx, n_gen); break;
            default: sum = compact_samples<8, 0, false, false>(N, key, a.jitter_words, ku, koff, lc2, R, bx, n_gen); break;
        }
    }
    ratio = sum * (1.0f / (float)(N * N));
    return true;
}

template <bool ALL_OPAQUE, bool HAS_BOX, int SM>
#ifndef RT_LB_NOBOX
#define RT_LB_NOBOX 3   // waves per SIMD the register budget is set for: 4 would spill ~250 bytes per lane and cost 25 % (measured)
#endif
#ifndef RT_LB_BOX
#define RT_LB_BOX 2
#endif

__global__ __launch_bounds__(SHADE_BLOCK, SM > 8 ? 2 : (HAS_BOX ? RT_LB_BOX : RT_LB_NOBOX)) void k_shade_fast(ShadeArgs a) {
    constexpr uint32_t MASK_SHIFT = 32 - SM;   // a deferred item packs (light | surfaces-in-reach mask << MASK_SHIFT)
    const uint32_t count = queue_slots(a.q);
    const int S = a.fp.n_surfaces, L = a.fp.n_lights, N = a.fp.n_shadow;
    const float invN = 1.0f / (float)N;
    const float inv_n2 = 1.0f / (float)(N * N);
    const uint32_t lane = threadIdx.x & 63;
    const bool classify = !a.no_cull && N < 256;
    const uint32_t n_waves = (gridDim.x * SHADE_BLOCK) >> 6;

    // wave-uniform: 4 type bits per surface in one word
    unsigned long long tword = 0;
#pragma unroll
    for (int j = 0; j < SM; ++j) tword |= (unsigned long long)(j < S ? (uint32_t)a.surf[j].type : 0xFu) << (4 * j);
    tword = (unsigned long long)__builtin_amdgcn_readfirstlane((uint32_t)tword) |
            ((unsigned long long)__builtin_amdgcn_readfirstlane((uint32_t)(tword >> 32)) << 32);
#define STYPE(j) ((uint32_t)(tword >> (4 * (j))) & 0xFu)

    uint32_t wl_n[7] = {0, 0, 0, 0, 0, 0, 0};
    unsigned long long n_samples = 0;   // samples generated x lanes holding a hit (wave-uniform)
    ItemQueues items;
    items.init((count + 63u) >> 6, n_waves);
    for (;;) {
        uint32_t item;
        if (!items.next(a.work, lane, item)) break;
        const int l_first = 0, l_last = L;
        const uint32_t chunk = item;
        const uint32_t base = chunk << 6;
        HitRec h;
        bool active;
        if (!load_chunk(a.q, base, count, lane, h, active)) continue;
        const f3 P = h.P;
        const SurfF own_sf = a.surf[h.own_id];
        const MatF m = a.mat[own_sf.mat];
        const bool own_valid = !(h.flags & HF_NO_OWN);
        const bool own_sphere = own_valid && own_sf.type == RT_SURF_SPHERE;
        const f3 ov = P - F3(own_sf.a.x, own_sf.a.y, own_sf.a.z);  // own sphere: oc
        const bool own_sph_blocks = own_sphere && !(h.flags & HF_CBATCH_POS);

        const float pfuzz_hit = 4e-7f * (fabsf(P.x) + fabsf(P.y) + fabsf(P.z));
        const uint32_t n_active = (uint32_t)__popcll(__ballot(active));

        // ---- which (light, surface) combinations can matter for this chunk at all?  One lane per combination
        //      (chunk_may_reach), all lights at once; bit l * S + s of premask.  Most chunks are an 8 x 8 pixel tile on
        //      one surface, a few hundredths of a scene unit across, and reach one surface per light or none.
        unsigned long long premask = ~0ull;
        const bool prefilter = classify && L * S <= 64;
        const bool own_sphere_any = __any(own_sphere);
        if (prefilter) {
            // bounding sphere of the chunk's hits: centroid (lanes without a hit hold a copy of one) and largest distance
            f3 sp = P;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) { sp.x += __shfl_xor(sp.x, o); sp.y += __shfl_xor(sp.y, o); sp.z += __shfl_xor(sp.z, o); }
            const f3 Pc = F3(__builtin_amdgcn_readfirstlane(sp.x) * 0.015625f, __builtin_amdgcn_readfirstlane(sp.y) * 0.015625f,
                             __builtin_amdgcn_readfirstlane(sp.z) * 0.015625f);
            const f3 dP = P - Pc;
            float rw2 = dot(dP, dP);
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) rw2 = fmaxf(rw2, __shfl_xor(rw2, o));
            const float Rw = __builtin_amdgcn_sqrtf(rw2) * 1.0001f;
            const int grp0 = __builtin_amdgcn_readfirstlane(own_sf.pad);
            const bool own_uniform = !__any(!own_valid || own_sf.pad != grp0);
            const int cl = (int)lane / S, cs = (int)lane - cl * S;
            const bool maybe = (int)lane < L * S && chunk_may_reach(a, cl, cs, Pc, Rw, own_uniform, grp0);
            premask = __ballot(maybe);
        }
        double cx = 0.0, cy = 0.0, cz = 0.0;   // lights summed in float64 (see k_shade_general)
        for (int l = l_first; l < l_last; ++l) {
            if (a.special && a.special[l] != a.special[l + 1]) {   // a light on a surface: the whole pair is shaded by k_shade_exact
                list_pairs(a, active, base + lane, (uint32_t)l | RT_PAIR_FULL);
                continue;
            }
            const LightF lt = uniform_load(a.light + l);
            bool defer = lt.point_path || !classify;
            uint32_t defer_mask = defer ? (1u << SM) - 1u : 0u;  // surfaces the generic loop has to test for this (wave, light)
            bool need_exact = false;                   // this lane's pair also goes to k_shade_exact (correction)
            float ratio = 1.f;
            const f3 lrel = F3(lt.pos[0], lt.pos[1], lt.pos[2]) - P;
            const float lc2 = dot(lrel, lrel);
            const float linv = lc2 > 0.f ? __builtin_amdgcn_rsqf(lc2) : 0.f;
            const f3 ldir = lrel * linv;
            // No surface of this light survived the chunk pre-test and no hit of the chunk lies on a sphere (whose own entry
            // and float64 listing do not go through the pre-test): ratio = 1 without the light basis or any classification.
            const bool nothing_to_do = prefilter && !own_sphere_any && ((premask >> (l * S)) & ((1ull << S) - 1ull)) == 0ull;
            if (!defer && nothing_to_do) wl_n[0] += 1;
            if (!defer && !nothing_to_do) {
                const f3 perp = fabsf(ldir.x) > 0.1f ? F3(0.f, 1.f, 0.f) : F3(1.f, 0.f, 0.f);  // light.py:24-27
                const f3 right = normalize(cross(ldir, perp));
                const f3 up = normalize(cross(right, ldir));
                const float llen = lc2 * linv;
                // radius of the sample square's circumcircle, plus the rounding allowance of the reach tests: positions
                // carry ~1e-7 of their magnitude, which matters for hit points far away (a ground plane near the horizon)
                const float rho_base = lt.radius * 0.70710678f * 1.0002f + 1e-6f;
                const float rho = rho_base + fmaf(4e-7f, llen, pfuzz_hit);
                const float inv_lc2 = linv * linv;
                const float s_hi_scale = llen > 1.5f * rho ? 1.0002f / (llen - rho) : 1e30f;
                // a light whose Phong terms vanish for this hit (back-facing and no specular lobe
                // towards the viewer) contributes exactly 0 whatever its shadow ratio: no surface
                // is relevant for such a lane
                const float ndl0 = dot(h.nrm, ldir);
                const f3 view0 = h.view;
                const float rv0 = 2.f * ndl0 * dot(h.nrm, view0) - dot(ldir, view0);
                const bool lit = ndl0 > 0.f || (rv0 > 0.f || m.shin == 0.f);
                const bool any_lit = __any(lit && active);   // a light behind every hit of the chunk: nothing to classify
                // The rows some lane needs go straight into the compact arrays of the sample loop (slot k, wave-uniform;
                // surfaces in ascending order, the own sphere last): a lane that does not need a row holds an inert one.
                Rows R;
                R.clear();
                int k = 0;
                auto append_row = [&](float ra, float rb, float rc, float rt) {
                    switch (k) {
                        case 0: R.set<0>(ra, rb, rc, rt); break;
                        case 1: R.set<1>(ra, rb, rc, rt); break;
                        case 2: R.set<2>(ra, rb, rc, rt); break;
                        case 3: R.set<3>(ra, rb, rc, rt); break;
                        case 4: R.set<4>(ra, rb, rc, rt); break;
                        case 5: R.set<5>(ra, rb, rc, rt); break;
                        case 6: R.set<6>(ra, rb, rc, rt); break;
                        case 7: R.set<7>(ra, rb, rc, rt); break;
                        default: break;   // more than KMAX rows: the pair is deferred (k > KMAX below)
                    }
                    k += 1;
                };
                // (a light within 2^-7 of the hit: the counted sample loop wants |sample - hit|^2 >= 2^-19, see compact_samples)
                bool gen = llen < 0.0078125f;
                uint32_t reach_bits = 0, box_bits = 0;
#pragma unroll
                for (int j = 0; j < SM; ++j) {
                    if (j < S && any_lit && (!prefilter || ((premask >> (l * S + j)) & 1ull))) {   // (wave-uniform)
                        float ra = -INFINITY, rb = 0.f, rc = 0.f;
                        bool lin_j = false;
                        const SurfF sf = uniform_load(a.surf + j);
                        const bool own = own_valid && sf.pad == own_sf.pad;   // the own surface or a twin of it
                        if (STYPE(j) == RT_SURF_SPHERE) {
                            const f3 oc = P - F3(sf.a.x, sf.a.y, sf.a.z);
                            const float c = dot(oc, oc) - sf.a.w;
                            const float r = sf.c.x;
                            // closest approach of the axis P->L to the centre, at parameter t; the sample
                            // segments fan out linearly (deviation s*rho at parameter s), and a point of
                            // contact needs |s - t|*llen <= r + s*rho, which bounds s by s_hi
                            const float t = fminf(fmaxf(-dot(oc, lrel) * inv_lc2, 0.f), 1.f);
                            const f3 q = oc + lrel * t;
                            const float reach = r + rho * fminf(1.f, s_hi_scale * (r + t * llen));
                            const bool never = own || !lit || dot(q, q) > reach * reach * 1.0002f;
                            if (__any(!never)) {   // (wave-uniform; the lanes differ by selects only: no divergent branches)
                                const float lmin = llen - rho;
                                // (beyond ~100 radii the cone's opening is below what an FP32 cosine resolves)
                                const bool linear_ok = c >= 1e-3f * r * r && c < 1e4f * r * r && lmin > 0.f && lmin * lmin > c * 1.0002f;
                                const bool use = !never && linear_ok;
                                const float kk = -__builtin_amdgcn_rsqf(c);
                                reach_bits |= never ? 0u : 1u << j;
                                gen = gen || (!never && !linear_ok);
                                ra = use ? dot(lrel, oc) * kk : -INFINITY;
                                rb = use ? dot(right, oc) * kk : 0.f;
                                rc = use ? dot(up, oc) * kk : 0.f;
                                lin_j = use;
                            }
                        } else if (STYPE(j) == RT_SURF_PLANE) {
                            const f3 n = F3(sf.a.x, sf.a.y, sf.a.z);
                            const float hh = sf.a.w - dot(P, n);
                            const float ah = fabsf(hh);
                            const float sg = hh < 0.f ? -1.f : 1.f;
                            const float dn0 = dot(lrel, n) * sg;  // > |h|  <=>  L is on the far side
                            const bool never = own || !lit || (ah - dn0) > rho * 1.0002f + 1e-5f * (ah + fabsf(dn0));
                            if (__any(!never)) {
                                const bool linear_ok = ah > 1e-4f && ah > RT_EPSF * 1.01f * (llen + rho) && ah > 64.f * (rho - rho_base);
                                const bool use = !never && linear_ok;
                                reach_bits |= never ? 0u : 1u << j;
                                gen = gen || (!never && !linear_ok);
                                ra = use ? (dn0 - ah) * PLANE_SCALE : -INFINITY;
                                rb = use ? dot(right, n) * (sg * PLANE_SCALE) : 0.f;
                                rc = use ? dot(up, n) * (sg * PLANE_SCALE) : 0.f;
                                lin_j = use;
                            }
                        } else {
                            if (!lit) {
                            } else if (own) {
                                if (dot(lrel, h.nrm) - rho < 0.f) {   // some sample may enter the box
                                    // An inward sample ray leaves the box again -- later than EPSILON when the hit keeps clear
                                    // of its face's edges, and before it reaches the light when every sample lies outside the
                                    // box.  Then "blocked" is "inward", sv.n < 0: a row of the plane kind.
                                    const f3 lcb = F3(lt.pos[0], lt.pos[1], lt.pos[2]) - F3(sf.c.x, sf.c.y, sf.c.z);
                                    const float keep = sf.c.w + rho;
                                    float edge_d = INFINITY;
                                    if (fabsf(h.nrm.x) < 0.5f) edge_d = fminf(edge_d, fminf(P.x - sf.a.x, sf.b.x - P.x));
                                    if (fabsf(h.nrm.y) < 0.5f) edge_d = fminf(edge_d, fminf(P.y - sf.a.y, sf.b.y - P.y));
                                    if (fabsf(h.nrm.z) < 0.5f) edge_d = fminf(edge_d, fminf(P.z - sf.a.z, sf.b.z - P.z));
                                    if (dot(lcb, lcb) > keep * keep * 1.0002f && edge_d > 1e-5f + 16.f * pfuzz_hit) {
                                        ra = -dot(lrel, h.nrm) * PLANE_SCALE; rb = -dot(right, h.nrm) * PLANE_SCALE; rc = -dot(up, h.nrm) * PLANE_SCALE;
                                        lin_j = true;
                                    } else gen = true;
                                }
                            } else {
                                const f3 oc = P - F3(sf.c.x, sf.c.y, sf.c.z);  // bounding sphere
                                const float t = fminf(fmaxf(-dot(oc, lrel) * inv_lc2, 0.f), 1.f);
                                const f3 q = oc + lrel * t;
                                const float reach = sf.c.w + rho;
                                if (!(dot(q, q) > reach * reach * 1.0002f)) {
                                    reach_bits |= 1u << j;
                                    // clear of the box by more than rounding: the slab test of the sample loop is exact
                                    const float gx = fmaxf(fmaxf(sf.a.x - P.x, P.x - sf.b.x), 0.f);
                                    const float gy = fmaxf(fmaxf(sf.a.y - P.y, P.y - sf.b.y), 0.f);
                                    const float gz = fmaxf(fmaxf(sf.a.z - P.z, P.z - sf.b.z), 0.f);
                                    const float clear2 = fmaf(gx, gx, fmaf(gy, gy, gz * gz));
                                    if (HAS_BOX && clear2 > 1e-6f * sf.c.w * sf.c.w) box_bits |= 1u << j;
                                    else gen = true;
                                }
                            }
                        }
                        if (__any(lin_j)) append_row(ra, rb, rc, ALL_OPAQUE ? 1.f : sf.transp);
                    }
                }
                // own sphere with c > 0 and a sample that may enter it steeply: k_shade_exact decides those samples in float64.
                // Such a light lies behind the tangent plane (no diffuse term); when its specular term is below 1e-4 of
                // full scale (0.03 LSB) for every channel, whatever the shadow ratio, the float64 answer cannot show.
                // (samples steeper than cos 0.2 only: see inward_limit in k_shade_deferred)
                // The samples span a square of edge `radius` in (right, up): sv.ov is smallest in a corner, |sv| >= llen.
                if (own_sphere && (h.flags & HF_CBATCH_POS) && lit && active &&
                    dot(lrel, ov) - (0.5002f * lt.radius + 1e-6f) * (fabsf(dot(right, ov)) + fabsf(dot(up, ov))) <
                        own_sf.c.x * (-0.1998f * llen + 1e-5f * (llen + rho))) {
                    bool shows = ndl0 > 0.f;
                    if (!shows) {
                        const float shin0 = m.shin;
                        const f3 wgt0 = h.wgt;
                        const float sp = rv0 > 0.f ? pow_pos(rv0, shin0) : (shin0 == 0.f ? 1.f : 0.f);
                        const float cmax = fmaxf(fmaxf(m.ks[0] * lt.col[0] * wgt0.x, m.ks[1] * lt.col[1] * wgt0.y), m.ks[2] * lt.col[2] * wgt0.z);
                        shows = sp * lt.spec * lt.shadow * cmax > a.exact_min;
                    }
                    need_exact = shows;
                }
                float own_a = -INFINITY, own_b = 0.f, own_c = 0.f;
                bool own_lin = false;
                if (own_sph_blocks && lit) {
                    const float r = own_sf.c.x;
                    const float dl = dot(lrel, ov);
                    if (dl - rho * r * 1.0002f < 0.f) {  // some sample may enter the own sphere
                        if (llen - rho > 2.0002f * r) {
                            const float kk = -2.0e6f;     // far root -2 w.oc > 1e-6  <=>  -sv.oc * 2e6 > len
                            own_a = dl * kk; own_b = dot(right, ov) * kk; own_c = dot(up, ov) * kk;
                            own_lin = true;
                        } else gen = true;
                    }
                }
                if (__any(own_lin)) append_row(own_a, own_b, own_c, ALL_OPAQUE ? 1.f : own_sf.b.w);
                uint32_t bcand = 0;
                if (HAS_BOX) {
#pragma unroll
                    for (int s = 0; s < SM; ++s)
                        if (STYPE(s) == RT_SURF_BOX && __ballot((box_bits >> s) & 1u)) bcand |= 1u << s;
                }
                const int nb = __popc(bcand);
                defer = __any(gen) || k > KMAX || nb > NBMAX;
                if (defer) need_exact = false;   // the generic kernel lists those lanes itself
                const unsigned long long em = __ballot(need_exact);
                if (em) {
                    // hand the lanes' (slot, light) pairs over
                    const uint32_t n = (uint32_t)__popcll(em);
                    uint32_t e0 = 0;
                    if (lane == 0) e0 = atomicAdd(a.exact_count, n);
                    e0 = __builtin_amdgcn_readfirstlane(e0);
                    if (need_exact) {   // entries beyond the capacity are only counted: the host then enlarges the list
                        const uint32_t pre = __builtin_amdgcn_mbcnt_hi((uint32_t)(em >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)em, 0u));
                        if (e0 + pre < a.exact_capacity) a.exact[e0 + pre] = make_uint2(base + lane, (uint32_t)l);
                    }
                }
                if (defer) {
#pragma unroll
                    for (int s = 0; s < SM; ++s)
                        if (__ballot((reach_bits >> s) & 1u)) defer_mask |= 1u << s;
                }
                if (!defer) {
                    wl_n[k + nb <= 4 ? k + nb : 4] += 1;
                    if (k + nb > 0) {
                        const uint32_t key = rt_grid_key(a.fp.seed_key, h.gpix, h.node, (uint32_t)l);
                        const float ku = RT_JITTER_INV * invN * lt.radius;  // 2^-8 / N * radius
                        const float koff = -0.5f * lt.radius;
                        BoxSet bx;
                        if (HAS_BOX && nb > 0) {
                            bx.lrel = lrel; bx.right = right; bx.up = up;
                            uint32_t brem = bcand;
#pragma unroll
                            for (int b = 0; b < NBMAX; ++b) {
                                const int sl = brem ? (int)__builtin_ctz(brem) : 0;  // wave-uniform surface index
                                // per lane: only a box this lane found in reach and keeps clear of (its own box and
                                // boxes out of reach stay inert: an empty box far away)
                                const bool used = brem != 0u && ((box_bits >> sl) & 1u);
                                brem &= brem - 1u;
                                const SurfF sf = uniform_load(a.surf + sl);
                                bx.lo[b][0] = used ? sf.a.x - P.x : 1e30f; bx.hi[b][0] = used ? sf.b.x - P.x : 1e30f;
                                bx.lo[b][1] = used ? sf.a.y - P.y : 1e30f; bx.hi[b][1] = used ? sf.b.y - P.y : 1e30f;
                                bx.lo[b][2] = used ? sf.a.z - P.z : 1e30f; bx.hi[b][2] = used ? sf.b.z - P.z : 1e30f;
                                bx.tr[b] = used ? sf.transp : 1.f;
                                // tile bounds: tangent cones of the bounding and of the inscribed sphere (same conditions as
                                // for a sphere's linear row: hit point outside, every sample beyond the tangent length)
                                const f3 ocb = P - F3(sf.c.x, sf.c.y, sf.c.z);
                                const float d2b = dot(ocb, ocb), lminb = llen - rho;
                                const float la = dot(lrel, ocb), lb = dot(right, ocb), lcc = dot(up, ocb);
#pragma unroll
                                for (int w = 0; w < 2; ++w) {
                                    const float rr = w == 0 ? sf.c.w : sf.c.w * 0.5773f;   // |edge| * sqrt(3)/2 (plus margin), |edge| / 2 (minus margin)
                                    const float cb = d2b - rr * rr;
                                    const bool ok = used && cb >= 1e-3f * rr * rr && cb < 1e4f * rr * rr && lminb > 0.f && lminb * lminb > cb * 1.0002f;
                                    const float kk = -__builtin_amdgcn_rsqf(cb);
                                    float* row = w == 0 ? bx.ro[b] : bx.ri[b];
                                    row[0] = ok ? la * kk : ((w == 0 && used) ? INFINITY : -INFINITY);
                                    row[1] = ok ? lb * kk : 0.f;
                                    row[2] = ok ? lcc * kk : 0.f;
                                }
                            }
                        }
                        float sum;
                        uint32_t n_gen = 0;
                        if (!HAS_BOX || nb == 0) {
                            if (ALL_OPAQUE) R.scale_for_counting();
                            switch (k) {
                                case 1: sum = compact_samples<1, 0, ALL_OPAQUE, true>(N, key, a.jitter_words, ku, koff, lc2, R, bx, n_gen); break;
                                case 2: sum = compact_samples<2, 0, ALL_OPAQUE, true>(N, key, a.jitter_words, ku, koff, lc2, R, bx, n_gen); break;
                                case 3: sum = compact_samples<3, 0, ALL_OPAQUE, true>(N, key, a.jitter_words, ku, koff, lc2, R, bx, n_gen); break;
                                case 4: sum = compact_samples<4, 0, ALL_OPAQUE, true>(N, key, a.jitter_words, ku, koff, lc2, R, bx, n_gen); break;
                                case 5: sum = compact_samples<5, 0, ALL_OPAQUE, true>(N, key, a.jitter_words, ku, koff, lc2, R, bx, n_gen); break;
                                case 6: sum = compact_samples<6, 0, ALL_OPAQUE, true>(N, key, a.jitter_words, ku, koff, lc2, R, bx, n_gen); break;
                                case 7: sum = compact_samples<7, 0, ALL_OPAQUE, true>(N, key, a.jitter_words, ku, koff, lc2, R, bx, n_gen); break;
                                default: sum = compact_samples<8, 0, ALL_OPAQUE, true>(N, key, a.jitter_words, ku, koff, lc2, R, bx, n_gen); break;
                            }
                        } else {
                            // boxes: linear rows rounded up to 0 / 2 / 4 / 8 (spare rows are inert), boxes to 1 or NBMAX
                            const int kk = k == 0 ? 0 : (k <= 2 ? 2 : (k <= 4 ? 4 : 8));
                            if (nb == 1) {
                                switch (kk) {
                                    case 0: sum = compact_samples<0, 1, ALL_OPAQUE, true>(N, key, a.jitter_words, ku, koff, lc2, R, bx, n_gen); break;
                                    case 2: sum = compact_samples<2, 1, ALL_OPAQUE, true>(N, key, a.jitter_words, ku, koff, lc2, R, bx, n_gen); break;
                                    case 4: sum = compact_samples<4, 1, ALL_OPAQUE, true>(N, key, a.jitter_words, ku, koff, lc2, R, bx, n_gen); break;
                                    default: sum = compact_samples<8, 1, ALL_OPAQUE, true>(N, key, a.jitter_words, ku, koff, lc2, R, bx, n_gen); break;
                                }
                            } else {
                                switch (kk) {
                                    case 0: sum = compact_samples<0, NBMAX, ALL_OPAQUE, true>(N, key, a.jitter_words, ku, koff, lc2, R, bx, n_gen); break;
                                    case 2: sum = compact_samples<2, NBMAX, ALL_OPAQUE, true>(N, key, a.jitter_words, ku, koff, lc2, R, bx, n_gen); break;
                                    case 4: sum = compact_samples<4, NBMAX, ALL_OPAQUE, true>(N, key, a.jitter_words, ku, koff, lc2, R, bx, n_gen); break;
                                    default: sum = compact_samples<8, NBMAX, ALL_OPAQUE, true>(N, key, a.jitter_words, ku, koff, lc2, R, bx, n_gen); break;
                                }
                            }
                        }
                        ratio = sum * inv_n2;
                        n_samples += (unsigned long long)n_gen * n_active;
                    }
                }
            }
            if (defer) {
                if (lane == 0) {
                    const uint32_t slot = atomicAdd(a.deferred_count, 1u);
                    a.deferred[slot] = make_uint2(base, (uint32_t)l | (defer_mask << MASK_SHIFT));
                }
                wl_n[lt.point_path ? 6 : 5] += 1;
            } else {
                f3 term = F3(0.f, 0.f, 0.f);
                const f3 wgt1 = h.wgt;
                phong_add(term, h.nrm, ldir, h.view, m, lt, ratio);   // need_exact lanes: k_shade_exact adds the correction
                cx += (double)(wgt1.x * term.x); cy += (double)(wgt1.y * term.y); cz += (double)(wgt1.z * term.z);
            }
        }
        if (active && (cx != 0.0 || cy != 0.0 || cz != 0.0)) {
            const uint32_t lp = h.lpix;
            unsafeAtomicAdd(&a.accum[(size_t)0 * a.npix + lp], cx);
            unsafeAtomicAdd(&a.accum[(size_t)1 * a.npix + lp], cy);
            unsafeAtomicAdd(&a.accum[(size_t)2 * a.npix + lp], cz);
        }
    }
    if (lane == 0) {
#pragma unroll
        for (int c = 0; c < 7; ++c)
            if (wl_n[c]) atomicAdd(rt_counter(a.counters, CNT_WL_NONE + c), (unsigned long long)wl_n[c]);
        if (n_samples) atomicAdd(rt_counter(a.counters, CNT_SAMPLES), n_samples);
    }
}

template <bool HAS_BOX, bool ALL_OPAQUE, int SM>
__global__ __launch_bounds__(SHADE_BLOCK) void k_shade_deferred(ShadeArgs a) {
    constexpr uint32_t MASK_SHIFT = 32 - SM;
    const uint32_t count = queue_slots(a.q);
    const uint32_t n_items = *a.deferred_count;
    const int S = a.fp.n_surfaces, N = a.fp.n_shadow;
    const float invN = 1.0f / (float)N;
    const float inv_n2 = 1.0f / (float)(N * N);
    const uint32_t lane = threadIdx.x & 63;

    unsigned long long tword = 0;
    float str[SM];
#pragma unroll
    for (int j = 0; j < SM; ++j) {
        str[j] = 0.f;
        uint32_t ty = 0xFu;
        if (j < S) { ty = (uint32_t)a.surf[j].type; if (!ALL_OPAQUE) str[j] = a.surf[j].transp; }
        tword |= (unsigned long long)ty << (4 * j);
    }
    tword = (unsigned long long)__builtin_amdgcn_readfirstlane((uint32_t)tword) |
            ((unsigned long long)__builtin_amdgcn_readfirstlane((uint32_t)(tword >> 32)) << 32);

    // One item per WORKGROUP: its four waves hold the same 64 hits and split the sample rows
    // (wave w takes rows w, w+4, ...); partial sums meet in LDS.  Items are few and long, so the
    // latency of one item decides how long this launch takes.
    __shared__ float s_sum[SHADE_BLOCK / 64][64];
    const int wv = (int)(threadIdx.x >> 6);
    constexpr int NWV = SHADE_BLOCK / 64;
    // Items are dealt out by index (workgroup b takes b, b + gridDim, ...): they cost about the same, and a shared counter
    // would be one returning atomic on one word per item -- 11 ns each, one after the other, which for the 12 600 items of
    // the 1080p transparency scene was the whole duration of this launch.
    for (uint32_t item = blockIdx.x; item < n_items; item += gridDim.x) {
        const uint2 it = a.deferred[item];
        const uint32_t base = __builtin_amdgcn_readfirstlane(it.x);
        const int l = (int)(__builtin_amdgcn_readfirstlane(it.y) & ((1u << MASK_SHIFT) - 1u));
        const uint32_t smask = __builtin_amdgcn_readfirstlane(it.y) >> MASK_SHIFT;  // surfaces in reach of some lane
        HitRec h;
        bool active;
        if (!load_chunk(a.q, base, count, lane, h, active)) continue;  // (same chunk in all four waves)
        const f3 P = h.P;
        const uint32_t flags = h.flags;
        const int own_id = (flags & HF_NO_OWN) ? -1 : h.own_id;
        const SurfF own_sf = a.surf[h.own_id];
        const MatF m = a.mat[own_sf.mat];
        const LightF lt = uniform_load(a.light + l);
        const f3 lrel = F3(lt.pos[0], lt.pos[1], lt.pos[2]) - P;
        const float lc2 = dot(lrel, lrel);
        const float linv = lc2 > 0.f ? __frsqrt_rn(lc2) : 0.f;
        const f3 ldir = lrel * linv;
        // own sphere with c > 0: inward shadow rays are decided in float64 (rt_exact.h)
        const bool exact = active && own_id >= 0 && (flags & HF_EXACT);
        // Shallow inward rays need no float64: the near root is resolved (t1 > 0, no blocker) whenever
        // 4ac >= 2^-50 b^2, and c > 0 means c >= one quantum of r^2 >= 2^-54 r^2, so b^2 <= r^2 / 4 is safe: with
        // b = 2 w.oc that is |w.oc| <= r / 4.  Steeper samples (w.oc < -0.2 r) are evaluated in float64.
        const float inward_limit = -0.2f * own_sf.c.x;
        const double* p64 = a.q.p64 + (size_t)(base + lane) * 3;
        float ratio;
        if (lt.point_path) {
            // ---- one ray to the light centre (ray_tracer.py:36-48), literal form (every wave computes it)
            const float dist = lc2 * linv;
            float f = 1.f;
            for (int s = 0; s < S; ++s) {
                const SurfF sf = uniform_load(a.surf + s);
                const bool own = own_id >= 0 && sf.pad == own_sf.pad;
                bool blk;
                if (sf.type == RT_SURF_SPHERE) blk = sphere_blocks_a(sf, P, ldir, dist, own, flags);
                else if (sf.type == RT_SURF_PLANE) blk = plane_blocks_a(sf, P, ldir, dist, own);
                else blk = box_blocks_a(sf, P, ldir, dist, own, h.nrm, flags);
                f = blk ? f * sf.transp : f;
            }
            if (exact && (flags & HF_CDOT_POS) && dot(ldir, P - F3(own_sf.a.x, own_sf.a.y, own_sf.a.z)) < inward_limit &&
                rtx::own_sphere_blocks_centre(p64, a.light64 + (size_t)l * 4, a.surf64[own_id]))
                f *= own_sf.b.w;
            ratio = f;
        } else {
            // ---- per-(hit, surface) constants
            float g[SM][HAS_BOX ? 6 : 5];
#pragma unroll
            for (int j = 0; j < SM; ++j) {
#pragma unroll
                for (int c = 0; c < (HAS_BOX ? 6 : 4); ++c) g[j][c] = 0.f;
                if (j < S) {
                    const SurfF sf = uniform_load(a.surf + j);
                    const bool own = own_id >= 0 && sf.pad == own_sf.pad;
                    if (STYPE(j) == RT_SURF_SPHERE) {
                        const f3 oc = P - F3(sf.a.x, sf.a.y, sf.a.z);
                        g[j][0] = oc.x; g[j][1] = oc.y; g[j][2] = oc.z;
                        g[j][3] = own ? -INFINITY : 0.f - (dot(oc, oc) - sf.a.w);
                        g[j][4] = own ? -1.f : sf.a.w;  // r^2; own sphere: never a root here (resolved by rule below)
                    } else if (STYPE(j) == RT_SURF_PLANE) {
                        g[j][0] = sf.a.x; g[j][1] = sf.a.y; g[j][2] = sf.a.z;
                        g[j][3] = own ? NAN : sf.a.w - dot(P, F3(sf.a.x, sf.a.y, sf.a.z));
                    } else if (HAS_BOX) {
                        g[j][0] = own ? 1e30f : sf.a.x - P.x; g[j][1] = own ? 1e30f : sf.a.y - P.y; g[j][2] = own ? 1e30f : sf.a.z - P.z;
                        g[j][3] = own ? 1e30f : sf.b.x - P.x; g[j][4] = own ? 1e30f : sf.b.y - P.y; g[j][5] = own ? 1e30f : sf.b.z - P.z;
                    }
                }
            }
            // ---- own-surface rule operands
            const bool own_sphere = own_id >= 0 && own_sf.type == RT_SURF_SPHERE;
            const bool own_box = HAS_BOX && own_id >= 0 && own_sf.type == RT_SURF_BOX;
            // sphere: oc; box: outward normal.  w . ov < 0  <=>  the sample enters the body
            const f3 ov = own_sphere ? P - F3(own_sf.a.x, own_sf.a.y, own_sf.a.z) : h.nrm;
            const bool own_sph_blocks_b = own_sphere && !(flags & HF_CBATCH_POS);
            const bool exact_b = exact && own_sphere && (flags & HF_CBATCH_POS);
            const float own_tr = own_sf.b.w;   // product over the own surface and its twins
            float ob[6];
#pragma unroll
            for (int c = 0; c < 6; ++c) ob[c] = 0.f;
            if (HAS_BOX && own_box) {
                ob[0] = own_sf.a.x - P.x; ob[1] = own_sf.a.y - P.y; ob[2] = own_sf.a.z - P.z;
                ob[3] = own_sf.b.x - P.x; ob[4] = own_sf.b.y - P.y; ob[5] = own_sf.b.z - P.z;
            }
            const f3 perp = fabsf(ldir.x) > 0.1f ? F3(0.f, 1.f, 0.f) : F3(1.f, 0.f, 0.f);
            const f3 right = normalize(cross(ldir, perp));
            const f3 up = normalize(cross(right, ldir));
            const uint32_t key = rt_grid_key(a.fp.seed_key, h.gpix, h.node, (uint32_t)l);
            const float ku = RT_JITTER_INV * invN * lt.radius;  // 2^-8 / N * radius
            if (wv == 0) {
                // own sphere with c > 0 and a sample that may enter it steeply: the loop below applies the rule ("never
                // blocks") and k_shade_exact adds the float64 correction (same predicate as in k_shade_fast)
                const float llen = lc2 * linv;
                bool list = exact_b && dot(lrel, ov) - (0.5002f * lt.radius + 1e-6f) * (fabsf(dot(right, ov)) + fabsf(dot(up, ov))) <
                                           own_sf.c.x * (-0.1998f * llen + 1e-5f * (llen + lt.radius));
                if (list) {   // ... and only if the pair can show: front-lit, or a specular term above exact_min
                    const float ndl0 = dot(h.nrm, ldir);
                    if (!(ndl0 > 0.f)) {
                        const float rv0 = 2.f * ndl0 * dot(h.nrm, h.view) - dot(ldir, h.view);
                        const float sp = rv0 > 0.f ? pow_pos(rv0, m.shin) : (m.shin == 0.f ? 1.f : 0.f);
                        const float cmax = fmaxf(fmaxf(m.ks[0] * lt.col[0] * h.wgt.x, m.ks[1] * lt.col[1] * h.wgt.y), m.ks[2] * lt.col[2] * h.wgt.z);
                        list = sp * lt.spec * lt.shadow * cmax > a.exact_min;
                    }
                }
                const unsigned long long em = __ballot(list);
                if (em) {
                    uint32_t e0 = 0;
                    if (lane == 0) e0 = atomicAdd(a.exact_count, (uint32_t)__popcll(em));
                    e0 = __builtin_amdgcn_readfirstlane(e0);
                    const uint32_t pre = __builtin_amdgcn_mbcnt_hi((uint32_t)(em >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)em, 0u));
                    if (list && e0 + pre < a.exact_capacity) a.exact[e0 + pre] = make_uint2(base + lane, (uint32_t)l);
                }
            }
            float sum = 0.f;
            for (int i = wv; i < N; i += NWV) {
                const float cu = ((float)i * invN - 0.5f) * lt.radius;
                uint32_t mm = 0;
                for (int j = 0; j < N; ++j) {
                    if (!(j & 1)) mm = rt_pair_word((uint32_t)i * rt_pair_stride((uint32_t)N) + ((uint32_t)j >> 1)) ^ key;   // one word per two samples
                    const uint32_t ub = (j & 1) ? (mm >> 8) & 0xFFu : mm >> 24;
                    const uint32_t vb = (j & 1) ? mm & 0xFFu : (mm >> 16) & 0xFFu;
                    const float cv = ((float)j * invN - 0.5f) * lt.radius;
                    const float ur = fmaf((float)ub, ku, cu);
                    const float vr = fmaf((float)vb, ku, cv);
                    const f3 sv = F3(fmaf(up.x, vr, fmaf(right.x, ur, lrel.x)), fmaf(up.y, vr, fmaf(right.y, ur, lrel.y)),
                                     fmaf(up.z, vr, fmaf(right.z, ur, lrel.z)));
                    const float d2 = dot(sv, sv);
                    const float dinv = __builtin_amdgcn_rsqf(d2);
                    const f3 w = sv * dinv;
                    const float dist = d2 * dinv;
                    float iw[3] = {0.f, 0.f, 0.f};
                    if (HAS_BOX) {
                        // |w_a| < 1e-6 is "parallel" (cube.py:129); clamping the magnitude keeps the slab logic
                        // branch-free: inside the slab both plane distances become huge with opposite signs
                        iw[0] = __builtin_amdgcn_rcpf(copysignf(fmaxf(fabsf(w.x), RT_EPSF), w.x));
                        iw[1] = __builtin_amdgcn_rcpf(copysignf(fmaxf(fabsf(w.y), RT_EPSF), w.y));
                        iw[2] = __builtin_amdgcn_rcpf(copysignf(fmaxf(fabsf(w.z), RT_EPSF), w.z));
                    }
                    bool any = false;
                    float f = 1.f;
#pragma unroll
                    for (int s = 0; s < SM; ++s) {
                        if (s < S && ((smask >> s) & 1u)) {
                            bool blk;
                            if (STYPE(s) == RT_SURF_SPHERE) {
                                const float b = fmaf(w.z, g[s][2], fmaf(w.y, g[s][1], w.x * g[s][0]));
                                const float disc = sphere_disc(F3(g[s][0], g[s][1], g[s][2]), w, b, g[s][4]);
                                const float sq = __builtin_amdgcn_sqrtf(disc);  // NaN when the ray misses
                                const float ssq = copysignf(sq, g[s][3]);       // outside: near root, inside: far root
                                const float tn = ssq - b;
                                blk = tn > RT_EPSF && tn < dist;
                            } else if (STYPE(s) == RT_SURF_PLANE) {
                                const float denom = fmaf(w.z, g[s][2], fmaf(w.y, g[s][1], w.x * g[s][0]));
                                const float t = g[s][3] * __builtin_amdgcn_rcpf(denom);
                                blk = fabsf(denom) >= RT_EPSF && t > RT_EPSF && t < dist;
                            } else if (HAS_BOX) {
                                const float t1x = g[s][0] * iw[0], t2x = g[s][3] * iw[0];
                                const float t1y = g[s][1] * iw[1], t2y = g[s][4] * iw[1];
                                const float t1z = g[s][2] * iw[2], t2z = g[s][5] * iw[2];
                                const float tnear = fmaxf(fmaxf(fminf(t1x, t2x), fminf(t1y, t2y)), fminf(t1z, t2z));
                                const float tfar = fminf(fminf(fmaxf(t1x, t2x), fmaxf(t1y, t2y)), fmaxf(t1z, t2z));
                                const float t = tnear > RT_EPSF ? tnear : tfar;   // cube.py:167-168
                                blk = tnear <= tfar && t > RT_EPSF && t < dist;
                            } else blk = false;
                            if (ALL_OPAQUE) any = any || blk;
                            else f = blk ? f * str[s] : f;
                        }
                    }
                    // ---- the hit's own surface, by rule (see the header of this file)
                    {
                        const float bo = fmaf(w.z, ov.z, fmaf(w.y, ov.y, w.x * ov.x));
                        const float t2 = -2.f * bo;
                        bool blk = own_sph_blocks_b && bo < 0.f && t2 > RT_EPSF && t2 < dist;
                        if (HAS_BOX) {
                            const float t2x = fmaxf(ob[0] * iw[0], ob[3] * iw[0]);
                            const float t2y = fmaxf(ob[1] * iw[1], ob[4] * iw[1]);
                            const float t2z = fmaxf(ob[2] * iw[2], ob[5] * iw[2]);
                            const float tfar = fminf(fminf(t2x, t2y), t2z);
                            blk = blk || (own_box && bo < 0.f && tfar > RT_EPSF && tfar < dist);
                        }
                        if (ALL_OPAQUE) any = any || blk;
                        else f = blk ? f * own_tr : f;
                    }
                    if (ALL_OPAQUE) sum += any ? 0.f : 1.f;
                    else sum += f;
                }
            }
            s_sum[wv][lane] = sum;
            __syncthreads();
            sum = 0.f;
#pragma unroll
            for (int w = 0; w < NWV; ++w) sum += s_sum[w][lane];
            __syncthreads();
            ratio = sum * inv_n2;
        }
        const unsigned long long n_active = (unsigned long long)__popcll(__ballot(active));
        if (threadIdx.x == 0) atomicAdd(rt_counter(a.counters, CNT_SAMPLES), (unsigned long long)(lt.point_path ? 1 : N * N) * n_active);
        f3 color = F3(0.f, 0.f, 0.f);
        phong_add(color, h.nrm, ldir, h.view, m, lt, ratio);
        if (wv == 0 && active && (color.x != 0.f || color.y != 0.f || color.z != 0.f)) {
            const uint32_t lp = h.lpix;
            unsafeAtomicAdd(&a.accum[(size_t)0 * a.npix + lp], (double)(h.wgt.x * color.x));
            unsafeAtomicAdd(&a.accum[(size_t)1 * a.npix + lp], (double)(h.wgt.y * color.y));
            unsafeAtomicAdd(&a.accum[(size_t)2 * a.npix + lp], (double)(h.wgt.z * color.z));
        }
    }
}

// (hit, light) pairs of the fast kernel whose own sphere has to be evaluated in float64 (rt_exact.h).  The fast
// kernel has shaded the pair with the own sphere taken as "never blocks" (the rule for c > 0); this kernel adds the
// correction: for every sample that enters the sphere steeply and that the float64 quadratic finds blocked, the
// light factor of the other surfaces (generic FP32 forms) times (transparency - 1).  One wave per pair, the lanes
// share the N x N samples; most pairs have few such samples (a wide light low over the horizon of the hit) or none.
// One (hit, light) pair whose light lies on a surface (ShadeArgs::special), shaded completely by ONE LANE (every hit of
// the frame has such a pair, so there are plenty): the surfaces the light lies on are tested in float64 with the
// reference's own operations where FP32 cannot tell, the others with the generic FP32 forms, the hit's own sphere as
// in the correction path below.
__device__ __forceinline__ void shade_pair_full(const ShadeArgs& a, uint32_t slot, int l, int N, float invN, double inv_n64) {
    const HitRec h = load_hit(a.q, slot);
    const f3 P = h.P;
    const uint32_t flags = h.flags;
    const uint32_t* sp = a.special + a.fp.n_lights + 1 + a.special[l];   // the surfaces this light lies on
    const uint32_t n_sp = a.special[l + 1] - a.special[l];
    unsigned long long others = ~0ull;   // reach mask of the FP32 tests (scenes of up to 64 surfaces): all but those
    bool on_own = false;
    for (uint32_t k = 0; k < n_sp; ++k) {
        if (sp[k] < 64u) others &= ~(1ull << sp[k]);
        on_own = on_own || sp[k] == (uint32_t)h.own_id;
    }
    const bool own_valid = !(flags & HF_NO_OWN);
    const bool own_special = own_valid && on_own;
    const int own_id = (own_valid && !own_special) ? h.own_id : -1;   // a special own surface is tested like any other, in float64
    const SurfF own_sf = a.surf[h.own_id];
    const MatF m = a.mat[own_sf.mat];
    const LightF lt = a.light[l];
    const double* light64 = a.light64 + (size_t)l * 4;
    const double* p64 = a.q.p64 + (size_t)slot * 3;
    const f3 ov = P - F3(own_sf.a.x, own_sf.a.y, own_sf.a.z);
    const float inward_limit = -0.2f * own_sf.c.x;
    const bool own_exact = own_id >= 0 && (flags & HF_EXACT) && own_sf.type == RT_SURF_SPHERE;
    const f3 lrel = F3(lt.pos[0], lt.pos[1], lt.pos[2]) - P;
    const float lc2 = dot(lrel, lrel);
    const float linv = lc2 > 0.f ? __frsqrt_rn(lc2) : 0.f;
    const f3 ldir = lrel * linv;
    float ratio;
    if (lt.point_path) {
        // ---- one ray to the light centre (ray_tracer.py:36-48)
        float f = shadow_factor<false>(a, P, ldir, lc2 * linv, own_id, h.nrm, flags, others, sp, n_sp);
        if (own_exact && (flags & HF_CDOT_POS) && dot(ldir, ov) < inward_limit &&
            rtx::own_sphere_blocks_centre(p64, light64, a.surf64[own_id]))
            f *= own_sf.b.w;
        const rtx::e3 P64 = rtx::E3(p64[0], p64[1], p64[2]);
        const rtx::e3 to_light = rtx::sub(rtx::E3(light64[0], light64[1], light64[2]), P64);
        const rtx::e3 d64 = rtx::normalize(to_light);
        const double dist64 = sqrt(rtx::dotb(to_light, to_light));
        for (uint32_t k = 0; k < n_sp && f != 0.f; ++k) {
            const int s = (int)sp[k];
            double t;
            if (rtx::scalar_t(a.surf64[s], P64, d64, t) && RT_EPS < t && t < dist64) f *= a.surf[s].transp;
        }
        ratio = f;
    } else {
        const f3 perp = fabsf(ldir.x) > 0.1f ? F3(0.f, 1.f, 0.f) : F3(1.f, 0.f, 0.f);
        const f3 right = normalize(cross(ldir, perp));
        const f3 up = normalize(cross(right, ldir));
        const uint32_t key = rt_grid_key(a.fp.seed_key, h.gpix, h.node, (uint32_t)l);
        rtx::OwnSphere os;        // float64 hit point and light basis: set up when the first sample needs them
        bool have_os = false;
        float sum = 0.f;
        for (uint32_t s_idx = 0; s_idx < (uint32_t)(N * N); ++s_idx) {
            const int i = (int)(s_idx / (uint32_t)N), j = (int)(s_idx % (uint32_t)N);
            uint32_t ub, vb;
            rt_sample_bits(key, (uint32_t)N, (uint32_t)i, (uint32_t)j, ub, vb);
            const float u = ((float)i + (float)ub * RT_JITTER_INV) * invN - 0.5f;
            const float v = ((float)j + (float)vb * RT_JITTER_INV) * invN - 0.5f;
            const f3 sv = lrel + right * (u * lt.radius) + up * (v * lt.radius);
            const float d2 = dot(sv, sv);
            const float dinv = __frsqrt_rn(d2);
            const f3 w = sv * dinv;
            const float dist = d2 * dinv;
            float f = shadow_factor<true>(a, P, w, dist, own_id, h.nrm, flags, others, sp, n_sp);
            if (own_exact && (flags & HF_CBATCH_POS) && dot(w, ov) < inward_limit) {
                if (!have_os) { os = rtx::own_sphere_setup(p64, light64); have_os = true; }
                if (rtx::own_sphere_blocks_sample(os, light64, a.surf64[own_id], N, inv_n64, i, j, ub, vb)) f *= own_sf.b.w;
            }
            // The surfaces the light lies on: a sample of a wide light usually ends clear of them (the sample square
            // is tilted against the surface), and then the FP32 test says the same for a slightly shorter and a
            // slightly longer ray; only where those two differ is the end point close enough for float64 to matter.
            // (A hit ON such a surface always takes the float64 test: FP32 has no rule for it here.)
            bool have_ray = false;
            rtx::e3 d64 = rtx::E3(0, 0, 1); double dist64 = 0;
            for (uint32_t k = 0; k < n_sp && f != 0.f; ++k) {
                const int s = (int)sp[k];
                const SurfF sf = a.surf[s];
                const bool own_here = own_special && sf.pad == own_sf.pad;
                const bool b_short = surface_blocks<true>(sf, P, w, dist * 0.9998f, false, h.nrm, flags);
                const bool b_long = surface_blocks<true>(sf, P, w, dist * 1.0002f, false, h.nrm, flags);
                bool blk = b_short;
                if (own_here || b_short != b_long) {
                    if (!have_os) { os = rtx::own_sphere_setup(p64, light64); have_os = true; }
                    if (!have_ray) { rtx::sample_ray(os, light64, N, inv_n64, i, j, ub, vb, d64, dist64); have_ray = true; }
                    const double t = rtx::batch_t(a.surf64[s], os.P, d64);
                    blk = t > RT_EPS && t < dist64;
                }
                if (blk) f *= sf.transp;
            }
            sum += f;
        }
        ratio = sum / (float)(N * N);
    }
    {
        f3 color = F3(0.f, 0.f, 0.f);
        phong_add(color, h.nrm, ldir, h.view, m, lt, ratio);
        if (color.x != 0.f || color.y != 0.f || color.z != 0.f) {
            unsafeAtomicAdd(&a.accum[(size_t)0 * a.npix + h.lpix], (double)(h.wgt.x * color.x));
            unsafeAtomicAdd(&a.accum[(size_t)1 * a.npix + h.lpix], (double)(h.wgt.y * color.y));
            unsafeAtomicAdd(&a.accum[(size_t)2 * a.npix + h.lpix], (double)(h.wgt.z * color.z));
        }
    }
}

__global__ __launch_bounds__(SHADE_BLOCK) void k_shade_exact(ShadeArgs a) {
    const uint32_t listed = *a.exact_count, listed_full = *a.full_count;
    const uint32_t n_items = listed < a.exact_capacity ? listed : a.exact_capacity;
    const uint32_t n_full = listed_full < a.exact_capacity ? listed_full : a.exact_capacity;
    const int N = a.fp.n_shadow;
    const float invN = 1.0f / (float)N;
    const double inv_n64 = (N & (N - 1)) == 0 ? 1.0 / (double)N : 0.0;   // exact reciprocal of a power of two (rt_exact.h)
    const uint32_t lane = threadIdx.x & 63;
    // the pairs asked for, listed or not: the host compares it with the capacity
    if (blockIdx.x == 0 && threadIdx.x == 0 && (listed || listed_full))
        atomicAdd(rt_counter(a.counters, CNT_EXACT), (unsigned long long)listed + (unsigned long long)listed_full);
    // a wave takes a batch of pairs at a time, the first batch by its own index (a wave with nothing to do leaves
    // without touching the shared counter: usually there are fewer batches than waves), the following ones from the
    // counter.  Two pairs per batch when pairs are few; when every hit is listed (a light on a surface) about a quarter
    // of a wave's share, because one counter word sustains only ~90 atomics/us.
    const uint32_t wave = (blockIdx.x * SHADE_BLOCK + threadIdx.x) >> 6;
    const uint32_t n_waves = (gridDim.x * SHADE_BLOCK) >> 6;
    const uint32_t BATCH = max(2u, min(4096u, n_items / (n_waves * 4u) + 1u));
    auto next_batch = [&]() {
        uint32_t v = 0;
        if (lane == 0) v = atomicAdd(a.work3, BATCH);
        return (uint32_t)__builtin_amdgcn_readfirstlane(v) + n_waves * BATCH;
    };
    for (uint32_t b0 = wave * BATCH; b0 < n_items; b0 = next_batch())
    for (uint32_t it = b0; it < b0 + BATCH && it < n_items; ++it) {
        const uint2 e = a.exact[it];
        const uint32_t slot = __builtin_amdgcn_readfirstlane(e.x);
        const uint32_t lword = __builtin_amdgcn_readfirstlane(e.y);
        const int l = (int)(lword & ~RT_PAIR_FULL);
        if (slot != RT_NULL_HIT && !(lword & RT_PAIR_FULL)) {
            const HitRec h = load_hit(a.q, slot);
            const f3 P = h.P;
            const int own_id = h.own_id;
            const SurfF own_sf = a.surf[own_id];
            const LightF lt = a.light[l];
            const f3 ov = P - F3(own_sf.a.x, own_sf.a.y, own_sf.a.z);
            const float inward_limit = -0.2f * own_sf.c.x;   // see k_shade_deferred
            const f3 lrel = F3(lt.pos[0], lt.pos[1], lt.pos[2]) - P;
            const float lc2 = dot(lrel, lrel);
            const float linv = lc2 > 0.f ? __frsqrt_rn(lc2) : 0.f;
            const f3 ldir = lrel * linv;
            const f3 perp = fabsf(ldir.x) > 0.1f ? F3(0.f, 1.f, 0.f) : F3(1.f, 0.f, 0.f);
            const f3 right = normalize(cross(ldir, perp));
            const f3 up = normalize(cross(right, ldir));
            const uint32_t key = rt_grid_key(a.fp.seed_key, h.gpix, h.node, (uint32_t)l);
            auto sample = [&](uint32_t s_idx, int& i, int& j, uint32_t& ub, uint32_t& vb, f3& w, float& dist) {
                i = (int)(s_idx / (uint32_t)N); j = (int)(s_idx % (uint32_t)N);
                rt_sample_bits(key, (uint32_t)N, (uint32_t)i, (uint32_t)j, ub, vb);
                const float u = ((float)i + (float)ub * RT_JITTER_INV) * invN - 0.5f;
                const float v = ((float)j + (float)vb * RT_JITTER_INV) * invN - 0.5f;
                const f3 sv = lrel + right * (u * lt.radius) + up * (v * lt.radius);
                const float d2 = dot(sv, sv);
                const float dinv = __frsqrt_rn(d2);
                w = sv * dinv;
                dist = d2 * dinv;
            };
            // pass 1: does any sample enter steeply at all?
            bool steep = false;
            for (uint32_t s_idx = lane; s_idx < (uint32_t)(N * N); s_idx += 64u) {
                int i, j; uint32_t ub, vb; f3 w; float dist;
                sample(s_idx, i, j, ub, vb, w, dist);
                steep = steep || dot(w, ov) < inward_limit;
            }
            if (__any(steep)) {
                const double* light64 = a.light64 + (size_t)l * 4;
                const rtx::OwnSphere os = rtx::own_sphere_setup(a.q.p64 + (size_t)slot * 3, light64);
                float delta = 0.f;
                for (uint32_t s_idx = lane; s_idx < (uint32_t)(N * N); s_idx += 64u) {
                    int i, j; uint32_t ub, vb; f3 w; float dist;
                    sample(s_idx, i, j, ub, vb, w, dist);
                    if (dot(w, ov) < inward_limit &&
                        rtx::own_sphere_blocks_sample(os, light64, a.surf64[own_id], N, inv_n64, i, j, ub, vb))
                        delta += shadow_factor<true>(a, P, w, dist, own_id, h.nrm, h.flags, ~0ull) * (own_sf.b.w - 1.f);
                }
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) delta += __shfl_down(delta, o);
                if (lane == 0 && delta != 0.f) {
                    const MatF m = a.mat[own_sf.mat];
                    f3 c1 = F3(0.f, 0.f, 0.f), c0 = F3(0.f, 0.f, 0.f);
                    phong_add(c1, h.nrm, ldir, h.view, m, lt, delta / (float)(N * N));   // affine in the ratio:
                    phong_add(c0, h.nrm, ldir, h.view, m, lt, 0.f);                       // the difference is the correction
                    const f3 color = c1 - c0;
                    unsafeAtomicAdd(&a.accum[(size_t)0 * a.npix + h.lpix], (double)(h.wgt.x * color.x));
                    unsafeAtomicAdd(&a.accum[(size_t)1 * a.npix + h.lpix], (double)(h.wgt.y * color.y));
                    unsafeAtomicAdd(&a.accum[(size_t)2 * a.npix + h.lpix], (double)(h.wgt.z * color.z));
                }
            }
        }
    }
    // ---- pairs to be shaded completely (a light on a surface): one lane each, 64 consecutive entries per wave,
    //      taken from the end of the list; groups of entries are handed out like the batches above
    const uint32_t n_groups = (n_full + 63u) >> 6;
    const uint32_t GROUPS = max(1u, min(256u, n_groups / (n_waves * 4u) + 1u));
    auto next_groups = [&]() {
        uint32_t v = 0;
        if (lane == 0) v = atomicAdd(a.work4, GROUPS);
        return (uint32_t)__builtin_amdgcn_readfirstlane(v) + n_waves * GROUPS;
    };
    for (uint32_t g0 = wave * GROUPS; g0 < n_groups; g0 = next_groups())
    for (uint32_t g = g0; g < g0 + GROUPS && g < n_groups; ++g) {
        const uint32_t i = g * 64u + lane;
        if (i < n_full) {
            const uint2 e = a.exact[a.exact_capacity - 1u - i];
            if (e.x != RT_NULL_HIT && (e.y & RT_PAIR_FULL)) shade_pair_full(a, e.x, (int)(e.y & ~RT_PAIR_FULL), N, invN, inv_n64);
        }
    }
}

// workgroups of a persistent kernel that fit one CU at a time (the kernels pull their work from a shared counter, so
// more would only queue behind the resident ones)
template <typename K>
static int resident_blocks_per_cu(K kernel) {
    int n = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, kernel, SHADE_BLOCK, 0) != hipSuccess || n < 1) n = 1;
    return n > 8 ? 8 : n;
}

template <int SM>
static void launch_small(const ShadeArgs& a, int n_cu, dim3 g, dim3 b, hipStream_t s, hipEvent_t after_main) {
    static int occ[2][2] = {{0, 0}, {0, 0}};   // [has_box][all_opaque], filled on first use
    int& o = occ[a.has_box ? 1 : 0][a.all_opaque ? 1 : 0];
    if (a.has_box) {
        if (a.all_opaque) { if (!o) o = resident_blocks_per_cu(k_shade_fast<true, true, SM>); hipLaunchKernelGGL((k_shade_fast<true, true, SM>), dim3(n_cu * o), b, 0, s, a); }
        else { if (!o) o = resident_blocks_per_cu(k_shade_fast<false, true, SM>); hipLaunchKernelGGL((k_shade_fast<false, true, SM>), dim3(n_cu * o), b, 0, s, a); }
    } else {
        if (a.all_opaque) { if (!o) o = resident_blocks_per_cu(k_shade_fast<true, false, SM>); hipLaunchKernelGGL((k_shade_fast<true, false, SM>), dim3(n_cu * o), b, 0, s, a); }
        else { if (!o) o = resident_blocks_per_cu(k_shade_fast<false, false, SM>); hipLaunchKernelGGL((k_shade_fast<false, false, SM>), dim3(n_cu * o), b, 0, s, a); }
    }
    if (after_main) (void)hipEventRecord(after_main, s);
    if (a.has_box) {
        if (a.all_opaque) hipLaunchKernelGGL((k_shade_deferred<true, true, SM>), g, b, 0, s, a);
        else hipLaunchKernelGGL((k_shade_deferred<true, false, SM>), g, b, 0, s, a);
    } else {
        if (a.all_opaque) hipLaunchKernelGGL((k_shade_deferred<false, true, SM>), g, b, 0, s, a);
        else hipLaunchKernelGGL((k_shade_deferred<false, false, SM>), g, b, 0, s, a);
    }
}

// Scenes of up to 8 surfaces use the register-resident kernels at full occupancy, 9..16 surfaces the same kernels
// with twice the per-surface registers (two waves per SIMD; a deferred item then has 16 bits for the light), larger
// scenes the any-size kernel.
void rt_launch_shade(const ShadeArgs& a, int n_cu, hipStream_t s, hipEvent_t after_main) {
    const dim3 g(n_cu * 4), b(SHADE_BLOCK);
    const int S = a.fp.n_surfaces;
    if (a.variant == SHADE_GENERAL || S > 16 || (S > SMALL_MAX && a.fp.n_lights >= 65536)) {
        if (!a.grid.cell_start && S <= 64 && !a.no_cull) hipLaunchKernelGGL(k_shade_general<true>, g, b, 0, s, a);
        else hipLaunchKernelGGL(k_shade_general<false>, g, b, 0, s, a);
        if (after_main) (void)hipEventRecord(after_main, s);
    } else if (S > SMALL_MAX) {
        launch_small<16>(a, n_cu, g, b, s, after_main);
    } else {
        launch_small<SMALL_MAX>(a, n_cu, g, b, s, after_main);
    }
    hipLaunchKernelGGL(k_shade_exact, g, b, 0, s, a);
}

// ---------------------------------------------------------------------------------------
// resolve: clip(colour, 0, 1) * 255 (ray_tracer.py:330-331) -> float32 image and its
// uint8 truncation (np.uint8, ray_tracer.py:438)
// ---------------------------------------------------------------------------------------
// accum is planar ([3][npix]: a wave's accumulator updates of one channel then touch neighbouring
// addresses); the images are interleaved RGB.  Every accumulator read is set back to zero, so the
// next band starts from cleared accumulators without a separate fill.
__global__ __launch_bounds__(256) void k_resolve(double* __restrict__ accum, int64_t npix, float* out_f32, uint8_t* out_u8) {
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (; i < npix; i += stride) {
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) {
            double c = accum[ch * npix + i];
            accum[ch * npix + i] = 0.0;
            c = c < 0.0 ? 0.0 : (c > 1.0 ? 1.0 : c);
            const float v = (float)(c * 255.0);
            if (out_f32) out_f32[3 * i + ch] = v;
            if (out_u8) out_u8[3 * i + ch] = (uint8_t)v;  // same value np.uint8() gives for the float image
        }
    }
}

void rt_launch_resolve(double* accum, int64_t n_pixels, float* out_f32, uint8_t* out_u8, hipStream_t s) {
    int64_t blocks = (n_pixels + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(k_resolve, dim3((unsigned)blocks), dim3(256), 0, s, accum, n_pixels, out_f32, out_u8);
}
